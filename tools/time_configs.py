"""Train-step time of the other BASELINE.json configurations at full size on one GPU:
configs[3] (30-layer MoL-10 head, 128 residual channels) and configs[4] (global conditioning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, Adam
from wavenets_amd.data import synthetic_waveforms

dev = torch.device('cuda', 0)
CFGS = {
    'cfg4_mol128': dict(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                        activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16),
    'cfg5_cond': dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                      activation='leaky_relu', bits=8, conditioning='global', mapping_layers=[8, 16, 32],
                      mapping_activation='leaky_relu'),
    'cfg1': dict(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[], bits=8),
}
B, T = 8, 16000
x = synthetic_waveforms(B, T + 1, seed=5, device=dev)
for name, kw in CFGS.items():
  Bc = 1 if name == 'cfg1' else B
  m = WaveNet(**kw, device=dev)
  m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
  data = x[:Bc]
  if kw.get('conditioning'):
    spk = torch.randint(0, 110, (Bc,), generator=torch.Generator().manual_seed(1))
    data = (x[:Bc], torch.nn.functional.one_hot(spk, 110).float().to(dev))
  for _ in range(2):
    logs = m.train_step(data)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  n = 5
  for _ in range(n):
    logs = m.train_step(data)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / n
  print(f'{name}: {dt * 1e3:.2f} ms/step, {Bc * T / dt / 1e6:.2f} M samples/s, loss {logs["loss"]:.1f}, '
        f'params {m.flat_params.numel()}')
  del m
  torch.cuda.empty_cache()
