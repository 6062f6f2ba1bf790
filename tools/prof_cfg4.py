"""Profiling driver: a few train steps of BASELINE configs[3] (128 residual channels, MoL-10 head)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, Adam
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
m = WaveNet(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
for _ in range(4):
  m.train_step(x)
torch.cuda.synchronize()
