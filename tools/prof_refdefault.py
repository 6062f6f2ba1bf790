"""Profiling driver: a few train steps of the reference's default network (train.py:22-50) at batch 64 x 8000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wavenets_amd import WaveNet, Adam, MeanSquaredError
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
kw, B, desc, T, ncond = bench.OTHER_CONFIGS['reference_default']
m = WaveNet(**kw, device=dev, seed=0)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
x = synthetic_waveforms(B, T + 1, seed=99, device=dev)
spk = torch.randint(0, ncond, (B,), generator=torch.Generator().manual_seed(1))
data = (x, torch.nn.functional.one_hot(spk, ncond).float().to(dev))
for _ in range(4):
  m.train_step(data)
torch.cuda.synchronize()
