"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py with the oracle):
the CPU leg pins the oracle against drift; the GPU leg checks the HIP path against the same
fixed vectors (bit-exact quantiser indices, 1e-4 activations, gradients relative to scale)."""
import os

import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
import sys
sys.path.insert(0, HERE)
from make_golden import CASES  # noqa: E402


def _load(name):
  return np.load(os.path.join(HERE, name + '.npz'))


def _params(d):
  ps, gs, i = [], [], 0
  while f'p{i}' in d:
    ps.append(torch.from_numpy(d[f'p{i}'])); gs.append(torch.from_numpy(d[f'g{i}'])); i += 1
  return ps, gs


def test_oracle_reproduces_quantiser_fixture():
  d = _load('quantiser')
  x = torch.from_numpy(d['x'])
  for b in (8, 16):
    assert np.array_equal(O.quantize(x, b).numpy().astype(np.int32), d[f'idx{b}'])
  assert np.allclose(O.mu_law(torch.from_numpy(d['mu_in']).double()).float().numpy(), d['mu_out'], atol=1e-7)


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_reproduces_model_fixture(name):
  d = _load(name)
  cfg = O.OracleConfig(**CASES[name]['cfg'])
  ps, gs = _params(d)
  x = torch.from_numpy(d['x'])
  cond = torch.from_numpy(d['cond']).double() if 'cond' in d else None
  loss, _, grads, pred = O.loss_and_grads(x.double(), [p.double() for p in ps], cfg, cond)
  assert abs(loss.item() - float(d['loss'])) < 1e-9 * abs(float(d['loss']))
  assert np.allclose(pred.float().numpy()[:, -64:, :], d['pred'], atol=1e-7)
  for g, r in zip(grads, gs):
    assert torch.allclose(g.float(), r, atol=1e-6 * max(1.0, r.abs().max().item()))


@pytest.mark.gpu
def test_hip_quantiser_matches_fixture():
  from wavenets_amd import ops
  d = _load('quantiser')
  x = torch.from_numpy(d['x']).cuda()
  for b in (8, 16):
    assert np.array_equal(ops.quantize(x, b).cpu().numpy(), d[f'idx{b}'])
  assert np.allclose(ops.mu_law(torch.from_numpy(d['mu_in']).cuda()).cpu().numpy(), d['mu_out'], atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_hip_matches_model_fixture(name):
  from wavenets_amd import WaveNet
  d = _load(name)
  kw = dict(CASES[name]['cfg'])
  cond_inputs = kw.pop('cond_inputs', 0)
  ps, gs = _params(d)
  dev = torch.device('cuda', 0)
  model = WaveNet(**kw, device=dev)
  if cond_inputs:
    model.build([(1, 8, 1), (1, cond_inputs)])
  model.set_weights([p.numpy() for p in ps])
  x = torch.from_numpy(d['x']).to(dev)
  cond = torch.from_numpy(d['cond']).to(dev) if 'cond' in d else None
  data = (x, cond) if cond is not None else x
  inp = (x[:, :-1], cond) if cond is not None else x[:, :-1]
  pred = model(inp)
  assert (pred[:, -64:, :].cpu().numpy() - d['pred']).__abs__().max() < 1e-4
  lg = model.logits(inp)
  assert np.abs(lg[:, -64:, :].cpu().numpy() - d['logits_tail']).max() < 1e-4
  loss, _, _ = model.loss_and_grads(data)
  assert abs(loss[0].item() - float(d['loss'])) < 2e-5 * abs(float(d['loss']))
  for n, g, r in zip(model.variable_names, model.gradients(), gs):
    scale = max(r.abs().max().item(), 1e-6)
    assert (g.cpu() - r).abs().max().item() < 1e-4 * scale + 1e-7, n
