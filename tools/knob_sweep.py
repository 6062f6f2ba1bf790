import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wavenets_amd import WaveNet, Adam, _lib
from wavenets_amd.data import synthetic_waveforms
L = _lib.lib()
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2:]]
dev = torch.device('cuda', 0)
x = synthetic_waveforms(8, 16001, seed=1, device=dev)
L.wn_debug_set(key, max(vals))
m = WaveNet(**bench.CFG2, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
for _ in range(4): m.train_step(x)
tot = {v: [] for v in vals}
for rnd in range(4):
  for v in vals:
    L.wn_debug_set(key, v)
    m.train_step(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): m.train_step(x)
    torch.cuda.synchronize()
    tot[v].append((time.perf_counter() - t0) / 20 * 1e3)
L.wn_debug_set(key, 0)
for v, ts in tot.items(): print(f'knob {key} = {v}: mean {sum(ts)/len(ts):.3f}  (' + ' '.join(f'{t:.3f}' for t in ts) + ')')
