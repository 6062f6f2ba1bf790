"""generation speed of configs[3] (128 channels): relay vs one workgroup (knob 2), several batch sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wavenets_amd import WaveNet, _lib
dev = torch.device('cuda', 0)
kw = bench.OTHER_CONFIGS['configs[3]'][0]
m = WaveNet(**kw, device=dev, seed=0)
L = _lib.lib()
for B in (1, 8, 32, 64):
  w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
  for knob in (0, 1):
    L.wn_debug_set(2, knob)
    m.generate(20, sample=w, use_queues=True, deterministic=False)
    torch.cuda.synchronize()
    n = 400
    ts = []
    for _ in range(3):
      t0 = time.perf_counter()
      m.generate(n, sample=w, use_queues=True, deterministic=False)
      torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    # subtract the priming pass measured by a 1-sample call
    t1 = []
    for _ in range(3):
      t0 = time.perf_counter(); m.generate(1, sample=w, use_queues=True, deterministic=False); torch.cuda.synchronize(); t1.append(time.perf_counter() - t0)
    p = sorted(t1)[1]
    print(f'B={B} {"one workgroup" if knob else "relay"}: {(t - p) / (n - 1) * 1e6:.1f} us/step = {(n - 1) / (t - p):.0f} samples/s per utterance', flush=True)
  L.wn_debug_set(2, 0)
