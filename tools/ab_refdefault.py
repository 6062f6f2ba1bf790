"""Same-box A/B of one debug knob (default 18) on the reference's default network: python tools/ab_refdefault.py [KEY]"""
import time, torch, sys
sys.path.insert(0, ".")
import bench
from wavenets_amd import WaveNet, Adam, MeanSquaredError, _lib
from wavenets_amd.data import synthetic_waveforms
dev = torch.device("cuda", 0)
kw, B, desc, T, ncond = bench.OTHER_CONFIGS["reference_default"]
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 18
for knob in (0, 1, 0, 1):
  _lib.lib().wn_debug_set(KEY, knob)
  m = WaveNet(**kw, device=dev, seed=0)
  m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
  x = synthetic_waveforms(B, T + 1, seed=99, device=dev)
  spk = torch.randint(0, ncond, (B,), generator=torch.Generator().manual_seed(1))
  data = (x, torch.nn.functional.one_hot(spk, ncond).float().to(dev))
  for _ in range(3): logs = m.train_step(data)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(10): logs = m.train_step(data)
  torch.cuda.synchronize()
  print("knob", KEY, knob, (time.perf_counter() - t0) / 10 * 1e3, "ms/step loss", logs["loss"])
  del m
