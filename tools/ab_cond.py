import os, sys, time
sys.path.insert(0, '.')
import torch
from wavenets_amd import WaveNet, Adam, _lib
from wavenets_amd.data import synthetic_waveforms
L = _lib.lib()
dev = torch.device('cuda', 0)
kw = dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
          activation='leaky_relu', bits=8, conditioning='global', mapping_layers=[8, 16, 32], mapping_activation='leaky_relu')
m = WaveNet(**kw, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
spk = torch.randint(0, 110, (8,), generator=torch.Generator().manual_seed(1))
c = torch.nn.functional.one_hot(spk, 110).float().to(dev)
for _ in range(3): m.train_step((x, c))
for rnd in range(3):
  for v in (0, 1):
    L.wn_debug_set(33, v)
    m.train_step((x, c)); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): logs = m.train_step((x, c))
    torch.cuda.synchronize()
    print('knob33', v, (time.perf_counter() - t0) / 20 * 1e3, 'ms/step', logs['loss'])
L.wn_debug_set(33, 0)
