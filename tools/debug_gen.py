"""Debug: queued generation, fused step vs per-block launches vs sliding window (continuous MoL outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, _lib
dev = torch.device('cuda', 0)
kw = dict(blocks=4, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[32],
          activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16)
if len(sys.argv) > 1 and sys.argv[1] == 'r64':
  kw = dict(blocks=6, channels=64, skip_channels=256, dilation_bound=16, final_layers_channels=[32],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16)
m = WaveNet(**kw, device=dev, seed=7)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
  for n, t in zip(m.variable_names, m.trainable_variables):
    if n.endswith('bias'):
      t.copy_(((torch.rand(t.shape, generator=g) * 2 - 1) * 0.3).to(dev))
B, n = 3, 40
w = (torch.rand(B, m.receptive_field, 1, generator=g) * 2 - 1).to(dev)
L = _lib.lib()
naive = m.generate(n, sample=w, use_queues=False, deterministic=True)
f1 = m.generate(n, sample=w, use_queues=True, deterministic=True)
f2 = m.generate(n, sample=w, use_queues=True, deterministic=True)
L.wn_debug_set(6, 1)
unf = m.generate(n, sample=w, use_queues=True, deterministic=True)
L.wn_debug_set(6, 0)
L.wn_debug_set(6, 2)
f3 = m.generate(n, sample=w, use_queues=True, deterministic=True)
L.wn_debug_set(6, 0)
print('fused without skip waves == naive:', torch.equal(f3, naive))
print('fused deterministic:', torch.equal(f1, f2))
print('unfused == naive:', torch.equal(unf, naive))
d = (f1 - naive).abs().flatten(1)
print('fused vs naive: max', d.max().item(), 'first differing step per utterance', [(r.nonzero()[0].item() if r.any() else -1) for r in d])
# one fused step vs one unfused step: where do the workspaces differ?
def run(knob):
  L.wn_debug_set(6, knob)
  m._ws['gen'].fill_(float(knob) * 1000.0)
  out = m.generate(2, sample=w, use_queues=True, deterministic=True)
  L.wn_debug_set(6, 0)
  torch.cuda.synchronize()
  return m._ws['gen'].clone() if hasattr(m, '_ws') else None, out
try:
  L.wn_debug_set(7, 1)
  wa, oa = run(1)
  L.wn_debug_set(7, 0)
  wb, ob = run(2)
  diff = (wa != wb) & ~(torch.isnan(wa) & torch.isnan(wb))
  idx = diff.nonzero().flatten()
  print('workspace floats', wa.numel(), 'differing', idx.numel())
  if idx.numel():
    # contiguous runs
    brk = (idx[1:] - idx[:-1] > 64).nonzero().flatten()
    starts = torch.cat([idx[:1], idx[brk + 1]]); ends = torch.cat([idx[brk], idx[-1:]])
    for s_, e_ in list(zip(starts.tolist(), ends.tolist()))[:40]:
      seg = slice(s_, e_ + 1)
      print(f'  [{s_}, {e_}] from_end={wa.numel() - s_} maxdiff={(wa[seg] - wb[seg]).abs().max().item():.3e} n={int(diff[seg].sum())}')
except Exception as e:
  print('ws compare failed', repr(e))
