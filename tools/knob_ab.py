"""Same-process A/B of one debug knob on BASELINE configs[1]: python tools/knob_ab.py KEY [VAL] -- alternates the knob
between 0 and VAL (default 1) in blocks of 20 training steps and prints the mean step time of each setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wavenets_amd import WaveNet, Adam, _lib
from wavenets_amd.data import synthetic_waveforms
L = _lib.lib()
key = int(sys.argv[1]); val = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device('cuda', 0)
m = WaveNet(**bench.CFG2, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
x = synthetic_waveforms(8, 16001, seed=1, device=dev)
for _ in range(4):
  m.train_step(x)
tot = {0: [], val: []}
for rnd in range(6):
  for v in (0, val):
    L.wn_debug_set(key, v)
    m.train_step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      m.train_step(x)
    torch.cuda.synchronize()
    tot[v].append((time.perf_counter() - t0) / 20 * 1e3)
L.wn_debug_set(key, 0)
for v, ts in tot.items():
  print(f'knob {key} = {v}: mean {sum(ts) / len(ts):.3f} ms/step  (' + ' '.join(f'{t:.3f}' for t in ts) + ')')
