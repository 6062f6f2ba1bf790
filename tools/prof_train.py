"""Profiler driver: a few training steps of BASELINE configs[1] under debug knobs (rocprofv3 -- python tools/prof_train.py [KEY VAL ...])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wavenets_amd import WaveNet, Adam, MeanSquaredError, _lib
from wavenets_amd.data import synthetic_waveforms
L = _lib.lib()
args = [int(a) for a in sys.argv[1:]]
for k, v in zip(args[0::2], args[1::2]):
  L.wn_debug_set(k, v)
dev = torch.device('cuda', 0)
m = WaveNet(**bench.CFG2, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0), metrics=[MeanSquaredError()])
x = synthetic_waveforms(8, 16001, seed=1, device=dev)
for _ in range(8):
  m.train_step(x)
torch.cuda.synchronize()
