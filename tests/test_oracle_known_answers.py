"""Analytic known-answer tests that pin the CPU oracle.

The reference has no tests or fixtures and TensorFlow is absent, so the restatement in
oracle/wavenet_oracle.py is pinned by closed-form answers derived from the reference
source text (file:line cited per test).
"""
import math

import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O


def test_dilation_schedule_and_receptive_field():
  # src/model.py:79-81,122 ; BASELINE.md anchors: 1025 (10 layers 1..512), 3071 (30), 768
  c1 = O.OracleConfig(blocks=10, dilation_bound=1024)
  assert O.dilation_schedule(c1) == [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
  assert O.receptive_field(c1) == 1025
  c2 = O.OracleConfig(blocks=30, dilation_bound=1024)
  assert O.receptive_field(c2) == 3071
  c3 = O.OracleConfig(blocks=5, layers_per_block=5, dilation_bound=256)
  assert O.dilation_schedule(c3)[:9] == [1, 2, 4, 8, 16, 32, 64, 128, 1]
  assert O.receptive_field(c3) == 768
  # the bound is exclusive: 512 gives 1..256
  assert max(O.dilation_schedule(O.OracleConfig(blocks=10, dilation_bound=512))) == 256


def test_param_counts_match_baseline():
  # BASELINE.md analytic anchors
  def count(cfg):
    return sum(int(np.prod(s)) for _, s in O.param_shapes(cfg))
  cfg1 = O.OracleConfig(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[])
  assert count(cfg1) == 60704
  cfg2 = O.OracleConfig(blocks=30, channels=64, skip_channels=256, dilation_bound=1024,
                        final_layers_channels=[128, 256], activation='leaky_relu')
  assert count(cfg2) == 1251264
  cfg4 = O.OracleConfig(blocks=30, channels=128, skip_channels=256, dilation_bound=1024,
                        final_layers_channels=[128, 256], num_mixtures=10,
                        sampling_function='logistic', bits=16)
  assert count(cfg4) == 3533854


@pytest.mark.parametrize('d', [1, 2, 8])
def test_causal_conv_single_tap_impulse(d):
  # y[t] = a*x[t-d] + b*x[t], zero for t<d (src/layers.py:82-88, Keras 'causal' padding)
  C, T = 3, 32
  a, b = 0.75, -1.5
  kern = torch.zeros(2, C, C)
  kern[0] = torch.eye(C) * a     # tap 0 multiplies x[t-d]
  kern[1] = torch.eye(C) * b     # tap 1 multiplies x[t]
  x = torch.randn(2, T, C)
  y = O.causal_conv1d(x, kern, torch.zeros(C), d)
  exp = b * x
  exp[:, d:] += a * x[:, :-d]
  assert torch.allclose(y, exp, atol=1e-6)
  # impulse: response only at t0 and t0+d
  imp = torch.zeros(1, T, C); imp[0, 5, 0] = 1.0
  yi = O.causal_conv1d(imp, kern, torch.zeros(C), d)
  nz = torch.nonzero(yi[0, :, 0]).flatten().tolist()
  assert nz == [5, 5 + d]


def test_kernel3_tap_order():
  # tap j multiplies x[t-(k-1-j)d]
  kern = torch.zeros(3, 1, 1); kern[0, 0, 0] = 1.0; kern[1, 0, 0] = 10.0; kern[2, 0, 0] = 100.0
  x = torch.zeros(1, 10, 1); x[0, 0, 0] = 1.0
  y = O.causal_conv1d(x, kern, torch.zeros(1), 2)[0, :, 0]
  assert y.tolist() == [100.0, 0, 10.0, 0, 1.0, 0, 0, 0, 0, 0]


def test_receptive_field_edge():
  # output at the last step depends on input RF-2 steps back via the block stack plus one
  # more via the input causal conv: RF = 2 + sum(dil) (src/model.py:122)
  cfg = O.OracleConfig(blocks=4, channels=4, dilation_bound=16, final_layers_channels=[])
  rf = O.receptive_field(cfg)
  assert rf == 2 + 1 + 2 + 4 + 8
  params = O.init_params(cfg, seed=3, dtype=torch.float64)
  T = rf + 5
  x = torch.randn(1, T, 1, dtype=torch.float64)
  base = O.model_forward(x, params, cfg, return_logits=True)[0, -1]
  # perturbing the sample exactly rf-1 back changes the last output ...
  x1 = x.clone(); x1[0, T - 1 - (rf - 1), 0] += 1.0
  assert not torch.allclose(O.model_forward(x1, params, cfg, return_logits=True)[0, -1], base)
  # ... one further back does not
  x2 = x.clone(); x2[0, T - 1 - rf, 0] += 1.0
  assert torch.allclose(O.model_forward(x2, params, cfg, return_logits=True)[0, -1], base, atol=0, rtol=0)


@pytest.mark.parametrize('bits', [8, 16])
def test_quantiser_table(bits):
  # src/model.py:151-153 ; SURVEY.md row Q1
  n = 2 ** bits
  x = torch.tensor([-1.0, -1e-9, 0.0, 1.0, -2.0, 2.0])
  assert O.quantize(x, bits).tolist() == [0, n // 2 - 1, n // 2, n - 1, 0, n - 1]
  edges = O.quantiser_edges(bits)
  assert len(edges) == n - 1
  e = torch.from_numpy(edges)
  # exactly on edge j (1-based) -> bin j ; one ulp below -> bin j-1
  at = O.quantize(e, bits)
  below = O.quantize(torch.nextafter(e, torch.tensor(-2.0)), bits)
  assert at.tolist() == list(range(1, n))
  assert below.tolist() == list(range(0, n - 1))
  # dequantise = left bin edge
  assert O.dequantize(torch.tensor([0, n // 2, n - 1]), bits).tolist() == [-1.0, 0.0, 1.0 - 2.0 / n]


def test_mulaw_roundtrip():
  # src/utils.py:35 and src/callbacks.py:130
  x = torch.linspace(-1, 1, 4001, dtype=torch.float64)
  assert torch.allclose(O.inverse_mu_law(O.mu_law(x)), x, atol=1e-9)
  assert abs(O.mu_law(torch.tensor(1.0)).item() - 1.0) < 1e-7
  assert O.mu_law(torch.tensor(0.0)).item() == 0.0


def test_categorical_loss_closed_forms():
  # uniform probs -> ln 256 ; one-hot at the target -> clip at 1-1e-7 and renormalise
  B, T, C = 2, 5, 256
  probs = torch.full((B, T, C), 1.0 / C)
  tgt = torch.randint(0, C, (B, T, 1))
  assert torch.allclose(O.loss_categorical(tgt, probs), torch.full((B, T), math.log(C)), atol=1e-5)
  onehot = torch.zeros(B, T, C, dtype=torch.float64)
  onehot.scatter_(-1, tgt, 1.0)
  l = O.loss_categorical(tgt, onehot)
  pt, rest = 1.0 - 1e-7, 255 * 1e-7
  assert torch.allclose(l, torch.full((B, T), -(math.log(pt) - math.log(pt + rest)), dtype=torch.float64), atol=1e-12)


def test_logistic_bin_mass_as_written():
  # src/model.py:538 uses halfbit = 0.5/2**bits although a bin of [-1,1] is 2/2**bits wide,
  # so each bin integrates the CDF over HALF its width: the masses at all bin centres sum
  # to ~0.5 of the interior mass, not 1.  The restatement keeps the reference's formula.
  bits, M = 8, 3
  pred = torch.tensor([0.3, -0.2, 0.1, -0.4, 0.0, 0.5, -3.0, -2.5, -3.5], dtype=torch.float64)
  centres = (torch.arange(2 ** bits, dtype=torch.float64) + 0.5) / 2 ** (bits - 1) - 1.0
  y = centres.view(1, -1, 1)
  l = O.loss_logistic(y, pred.view(1, 1, -1).expand(1, 2 ** bits, -1), M, bits)
  mass = torch.exp(-l).sum().item()
  assert 0.495 < mass <= 0.5 + 1e-9


def test_gaussian_loss_single_component():
  M = 1
  pred = torch.tensor([[[0.0, 0.25, math.log(0.5)]]], dtype=torch.float64)
  y = torch.tensor([[[0.5]]], dtype=torch.float64)
  l = O.loss_gaussian(y, pred, M).item()
  exp = -math.log(math.exp(-0.5 * ((0.5 - 0.25) / 0.5) ** 2) / (0.5 * math.sqrt(2 * 3.14159265359)))
  assert abs(l - exp) < 1e-12


def test_layer_skip_is_pre_residual_when_no_skip_channels():
  # src/layers.py:216-223
  cfg = O.OracleConfig(blocks=1, channels=4, dilation_bound=4, final_layers_channels=[])
  p = O.init_params(cfg, seed=1)
  x = torch.randn(1, 8, 4)
  xo, sk = O.layer_forward(x, p[2:6], dilations=[1], activation_name=None, residual=True, has_skip=False)
  assert torch.allclose(xo, sk + x)


def test_finite_difference_gradients_fp64():
  cfg = O.OracleConfig(blocks=3, channels=4, skip_channels=6, dilation_bound=4,
                       final_layers_channels=[5], activation='leaky_relu', bits=4)
  params = O.init_params(cfg, seed=7, dtype=torch.float64)
  x = (torch.rand(2, 12, 1, dtype=torch.float64) * 2 - 1)
  loss, _, grads, _ = O.loss_and_grads(x, params, cfg)
  rng = np.random.default_rng(0)
  for pi in [0, 2, 5, 6, len(params) - 2]:
    p = params[pi]
    idx = tuple(int(rng.integers(0, s)) for s in p.shape)
    h = 1e-6
    pp = [q.clone() for q in params]; pp[pi][idx] += h
    pm = [q.clone() for q in params]; pm[pi][idx] -= h
    lp = O.loss_and_grads(x, pp, cfg)[0]
    lm = O.loss_and_grads(x, pm, cfg)[0]
    fd = (lp - lm).item() / (2 * h)
    assert abs(fd - grads[pi][idx].item()) < 1e-6 * max(1.0, abs(fd))


def test_keras_adam_hand_computed():
  # train.py:225-226 + Keras Adam (eps outside sqrt, not bias-corrected separately)
  p = [torch.tensor([1.0, -2.0], dtype=torch.float64)]
  g = [torch.tensor([3.0, 4.0], dtype=torch.float64)]        # norm 5 -> clipped to norm 1
  gc = O.clip_by_norm_per_tensor(g, 1.0)
  assert torch.allclose(gc[0], torch.tensor([0.6, 0.8], dtype=torch.float64))
  g_small = [torch.tensor([0.3, 0.4], dtype=torch.float64)]  # norm .5 -> untouched
  assert torch.allclose(O.clip_by_norm_per_tensor(g_small, 1.0)[0], g_small[0])
  m = [torch.zeros(2, dtype=torch.float64)]; v = [torch.zeros(2, dtype=torch.float64)]
  lr = 0.1
  p1, m1, v1 = O.keras_adam_step(p, gc, m, v, 1, lr)
  # step 1: m = .1 g, v = .001 g^2, alpha = lr*sqrt(.001)/.1
  alpha = lr * math.sqrt(1 - 0.999) / (1 - 0.9)
  exp = p[0] - alpha * (0.1 * gc[0]) / (torch.sqrt(0.001 * gc[0] ** 2) + 1e-7)
  assert torch.allclose(p1[0], exp, atol=1e-15)
  p2, m2, v2 = O.keras_adam_step(p1, gc, m1, v1, 2, lr)
  m_exp = 0.9 * 0.1 * gc[0] + 0.1 * gc[0]
  v_exp = 0.999 * 0.001 * gc[0] ** 2 + 0.001 * gc[0] ** 2
  alpha2 = lr * math.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
  assert torch.allclose(p2[0], p1[0] - alpha2 * m_exp / (torch.sqrt(v_exp) + 1e-7), atol=1e-15)


def test_loss_normalisation_is_sum_over_time_mean_over_global_batch():
  # src/model.py:328-329
  cfg = O.OracleConfig(blocks=2, channels=4, dilation_bound=4, final_layers_channels=[], bits=4)
  params = O.init_params(cfg, seed=2, dtype=torch.float64)
  x = torch.rand(4, 9, 1, dtype=torch.float64) * 2 - 1
  loss, _, grads, pred = O.loss_and_grads(x, params, cfg)
  per = O.loss_categorical(O.quantize(x[:, 1:], 4), pred)
  assert abs(loss.item() - per.sum().item() / 4) < 1e-12
  # two replicas with half the batch each and global_batch=4: summed grads == single
  l0, _, g0, _ = O.loss_and_grads(x[:2], params, cfg, global_batch=4)
  l1, _, g1, _ = O.loss_and_grads(x[2:], params, cfg, global_batch=4)
  assert abs((l0 + l1).item() - loss.item()) < 1e-12
  for a, b, c in zip(g0, g1, grads):
    assert torch.allclose(a + b, c, atol=1e-12)


def test_naive_generation_shapes_and_determinism():
  cfg = O.OracleConfig(blocks=3, channels=4, dilation_bound=8, final_layers_channels=[], bits=4)
  params = O.init_params(cfg, seed=5)
  rf = O.receptive_field(cfg)
  w = torch.zeros(2, rf, 1)
  out = O.generate_naive(params, cfg, 6, w)
  assert out.shape == (2, 6, 1)
  # every value is a left bin edge
  q = (out + 1.0) * 2 ** (cfg.bits - 1)
  assert torch.equal(q, torch.round(q))
  assert torch.equal(out, O.generate_naive(params, cfg, 6, w))


def test_global_conditioning_is_a_per_batch_bias():
  cfg = O.OracleConfig(blocks=2, channels=4, dilation_bound=4, final_layers_channels=[],
                       conditioning='global', mapping_layers=[3, 5], mapping_activation='leaky_relu',
                       cond_inputs=7, bits=4)
  params = O.init_params(cfg, seed=9, dtype=torch.float64)
  names = [n for n, _ in O.param_shapes(cfg)]
  assert names[-4:] == ['mapping0/kernel', 'mapping0/bias', 'mapping1/kernel', 'mapping1/bias']
  x = torch.rand(2, 10, 1, dtype=torch.float64)
  cond = torch.rand(2, 7, dtype=torch.float64)
  out = O.model_forward(x, params, cfg, cond)
  # materialising the repeated (B,T,Cc) condition as the reference does gives the same
  m = O.mapping_forward(cond, params, cfg)
  assert m.shape == (2, 5)
  out2 = O.model_forward(x, params, cfg, cond)
  assert torch.equal(out, out2)
  # different condition -> different output
  assert not torch.allclose(out, O.model_forward(x, params, cfg, cond + 1.0))


def test_activation_with_branch_only_decides_inside_the_kink_tolerance():
  """O.activation_with_branch: the implementation's branch is taken only where |pre-activation| < KINK_TOL."""
  x = torch.tensor([-1.0, -5e-5, 5e-5, 0.0, 2.0, -2e-4, 3e-4], dtype=torch.float64, requires_grad=True)
  all_pos = torch.ones(7, dtype=torch.bool)
  y, n = O.activation_with_branch(x, 'leaky_relu', all_pos)
  g, = torch.autograd.grad(y.sum(), x)
  assert g.tolist() == [0.2, 1.0, 1.0, 1.0, 1.0, 0.2, 1.0] and n == 1          # only -5e-5 changed side
  y, n = O.activation_with_branch(x, 'leaky_relu', ~all_pos)
  g, = torch.autograd.grad(y.sum(), x)
  assert g.tolist() == [0.2, 0.2, 0.2, 0.2, 1.0, 0.2, 1.0] and n == 2
  y0, n0 = O.activation_with_branch(x, 'leaky_relu', None)
  assert n0 == 0 and torch.equal(y0, O.activation(x, 'leaky_relu'))
  yt, nt = O.activation_with_branch(x, 'tanh', all_pos)                          # smooth activations: untouched
  assert nt == 0 and torch.equal(yt, torch.tanh(x))
