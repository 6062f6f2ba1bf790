"""Generation speed (the reference prints 'Speed of generation was ... samples/s', train.py:253-261):
naive sliding window vs queued ring buffers on BASELINE configs[1] weights, batch B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
# second argument 'cfg3': BASELINE configs[3] weights (128 residual channels, MoL-10) instead of configs[1]
kw = bench.OTHER_CONFIGS['configs[3]'][0] if len(sys.argv) > 2 and sys.argv[2] == 'cfg3' else bench.CFG2
m = WaveNet(**kw, device=dev)
# further arguments: KEY VAL pairs of debug knobs (wavenets_amd/csrc/wn_error.cpp)
from wavenets_amd import _lib
for k_, v_ in zip(sys.argv[3::2], sys.argv[4::2]):
  _lib.lib().wn_debug_set(int(k_), int(v_))
w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
for name, queued, n, det in (('naive', False, 20, True), ('queued', True, 400, True), ('queued, stochastic draws', True, 400, False)):
  m.generate(3, sample=w, use_queues=queued, deterministic=det)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  out = m.generate(n, sample=w, use_queues=queued, deterministic=det)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print(f'{name}: B={B} {n} samples/utterance in {dt:.3f} s -> {n / dt:.1f} samples/s per utterance, '
        f'{B * n / dt:.1f} samples/s aggregate, {dt / n * 1e3:.3f} ms/step')
