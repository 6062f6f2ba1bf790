"""Profiling driver: queued generation on the configs[3] network (30 blocks of 128 channels, MoL-10 draws), batch 8, 300
samples after the priming pass -- for rocprofv3 --kernel-trace --stats (tools/collect_profiles.sh, profiles/<tag>_gen128_*).
argv[1]: batch (default 8); argv[2] = 1: the one-workgroup chain (knob 2) instead of the relay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from wavenets_amd import WaveNet, _lib
dev = torch.device('cuda', 0)
m = WaveNet(**bench.OTHER_CONFIGS['configs[3]'][0], device=dev, seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == '1':
  _lib.lib().wn_debug_set(2, 1)
w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
m.generate(300, sample=w, use_queues=True, deterministic=False)
torch.cuda.synchronize()
