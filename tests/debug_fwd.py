"""Debug helper (not a test; it lives here because only tests/ may use the oracle): forward of a small
model case vs the oracle under debug knobs:  python tests/debug_fwd.py <case> [KEY VAL ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from wavenets_amd import _lib
import test_gpu_parity as T
L = _lib.lib()
name = sys.argv[1]
for k, v in zip(sys.argv[2::2], sys.argv[3::2]):
  L.wn_debug_set(int(k), int(v))
kw = dict(T.MODEL_CASES[name])
ocfg, params, model = T.make_pair(seed=3, **kw)
x, cond = T._inputs(kw, 3, 333)
inp = (x.to(T.dev()), cond.to(T.dev())) if cond is not None else x.to(T.dev())
got = model.logits(inp)
_, inter = T.O.model_forward(x.double(), [p.double() for p in params], ocfg, cond.double() if cond is not None else None, return_intermediates=True)
ref = inter['logits']
d = (got.cpu().double() - ref).abs()
print(name, sys.argv[2:], 'max err', d.max().item(), 'nan', torch.isnan(got).sum().item(), 'first bad t', (d.amax(dim=(0, 2)) > 1e-3).nonzero().flatten()[:8].tolist())
