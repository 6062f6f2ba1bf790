"""soak: relay vs one-workgroup generation on configs[3], many steps, with a competing stream"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wavenets_amd import WaveNet, _lib
dev = torch.device('cuda', 0)
m = WaveNet(**bench.OTHER_CONFIGS['configs[3]'][0], device=dev, seed=1)
L = _lib.lib()
for B, n in ((8, 30000), (40, 12000), (1, 20000)):
  w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(B)) * 2 - 1).to(dev)
  L.wn_debug_set(2, 1)
  ref = m.generate(n, sample=w, use_queues=True, deterministic=False)
  L.wn_debug_set(2, 0)
  torch.cuda.synchronize()
  side = torch.cuda.Stream()
  big = torch.empty(1 << 28, device=dev)
  a = torch.randn(4096, 4096, device=dev)
  with torch.cuda.stream(side):
    for i in range(400):
      big.mul_(1.0001)
      if i % 4 == 0: a = (a @ a).clamp_(-1, 1)
  t0 = time.perf_counter()
  got = m.generate(n, sample=w, use_queues=True, deterministic=False)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  eq = torch.equal(ref, got)
  print(f'B={B} n={n}: equal={eq} ({dt:.1f} s under load)', flush=True)
  if not eq:
    d = (ref != got).nonzero()
    print('first mismatch at', d[0].tolist(), 'count', len(d))
    sys.exit(1)
print('soak ok')
