import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, Adam
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
kw = dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
          activation='leaky_relu', bits=8, conditioning='global', mapping_layers=[8, 16, 32], mapping_activation='leaky_relu')
m = WaveNet(**kw, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
spk = torch.randint(0, 110, (8,), generator=torch.Generator().manual_seed(1))
c = torch.nn.functional.one_hot(spk, 110).float().to(dev)
for _ in range(3):
  m.train_step((x, c))
torch.cuda.synchronize()
