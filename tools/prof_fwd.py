"""Profiling driver: N forward passes (and optionally train steps) of BASELINE configs[1] so that
rocprofv3 counter runs see the fused residual-block forward kernel in isolation."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, Adam
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--fwd', type=int, default=3)
ap.add_argument('--train', type=int, default=0)
ap.add_argument('--batch', type=int, default=8)
args = ap.parse_args()
dev = torch.device('cuda', 0)
m = WaveNet(**bench.CFG2, device=dev)
m.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
g = torch.Generator().manual_seed(0)
x = (torch.rand(args.batch, 16001, 1, generator=g) * 2 - 1).to(dev)
for _ in range(args.fwd):
  m(x[:, :-1])
for _ in range(args.train):
  m.train_step(x)
torch.cuda.synchronize()
print('done')
