"""Training-step time of BASELINE configs[3] (128 channels, MoL-10) under debug knobs: python tools/knob_cfg4.py [KEY VAL ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, Adam, _lib
from wavenets_amd.data import synthetic_waveforms
L = _lib.lib()
args = [int(a) for a in sys.argv[1:]]
for k, v in zip(args[0::2], args[1::2]):
  L.wn_debug_set(k, v)
dev = torch.device('cuda', 0)
m = WaveNet(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
for _ in range(3):
  logs = m.train_step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
  logs = m.train_step(x)
torch.cuda.synchronize()
print(f'cfg4 knobs {args}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step  loss {float(logs["loss"]):.5f}')
