"""A/B: step time of configs[1] and configs[3] for the library in WN_HIP_LIB"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wavenets_amd import WaveNet, Adam, MeanSquaredError
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
out = []
for name, kw in (('configs1', bench.CFG2), ('configs3', bench.OTHER_CONFIGS['configs[3]'][0])):
  m = WaveNet(**kw, device=dev, seed=0)
  m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
  x = synthetic_waveforms(8, 16001, seed=99, device=dev)
  for _ in range(5): m.train_step(x)
  ts = []
  for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): m.train_step(x)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 15 * 1e3)
  out.append(f'{name} {sorted(ts)[1]:.3f} ms')
  del m; torch.cuda.empty_cache()
print(os.path.basename(os.environ.get('WN_HIP_LIB', 'libwn_hip.so')), ' | '.join(out), flush=True)
