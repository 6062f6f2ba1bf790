"""GPU parity tests: the HIP path (through the C-ABI, via the Python mirror classes) against
the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): 1e-4 absolute per fp32 activation, bit-exact quantisation
indices.  Gradients are compared relative to each tensor's scale.
"""
import math

import numpy as np
import pytest
import torch

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu

ATOL_ACT = 1e-4          # per-activation tolerance stated by north_star


def dev():
  return torch.device('cuda', 0)


@pytest.fixture(params=['split', 'fp32'])
def math_mode(request):
  """Both contraction modes: 'split' = fp16 hi/lo 3-product MFMA (default), 'fp32' = exact-fp32 MFMA
  (debug knob 1).  The tolerance is the same 1e-4 / bit-exact bar for both."""
  from wavenets_amd import _lib
  _lib.lib().wn_debug_set(1, 1 if request.param == 'fp32' else 0)
  yield request.param
  _lib.lib().wn_debug_set(1, 0)


def _rel(a, b):
  a, b = a.double().cpu(), b.double().cpu()
  return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def make_pair(seed=0, bias_range=0.1, **kw):
  """(oracle cfg, oracle params, HIP model with the same weights)."""
  from wavenets_amd import WaveNet
  cond_inputs = kw.pop('cond_inputs', 0)
  ocfg = O.OracleConfig(**kw, cond_inputs=cond_inputs)
  params = O.init_params(ocfg, seed=seed, bias_range=bias_range)
  mkw = dict(kw)
  model = WaveNet(**mkw, device=dev())
  if kw.get('conditioning') is not None:
    model.build([(1, 8, 1), (1, cond_inputs)])
  model.set_weights([p.numpy() for p in params])
  return ocfg, params, model


# ------------------------------------------------------------------------------------------
# integer / elementwise boundary
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('bits', [4, 8, 16])
def test_quantize_bit_exact(bits):
  from wavenets_amd import ops
  g = torch.Generator().manual_seed(bits)
  x = torch.rand(200000, generator=g) * 2.2 - 1.1
  edges = torch.from_numpy(O.quantiser_edges(bits))
  pick = edges[torch.randint(0, len(edges), (20000,), generator=g)]
  special = torch.tensor([-1.0, 1.0, 0.0, -0.0, -1e-9, 1e-9, -1e-30, 1e-30, 2.0, -2.0, 0.999999, -0.999999])
  x = torch.cat([x, pick, torch.nextafter(pick, torch.tensor(2.0)), torch.nextafter(pick, torch.tensor(-2.0)), special])
  ref = O.quantize(x, bits)
  got = ops.quantize(x.to(dev()), bits).cpu().long()
  assert torch.equal(got, ref)
  # dequantise: exact left bin edges
  back = ops.dequantize(got.to(dev()).int(), bits).cpu()
  assert torch.equal(back, O.dequantize(ref, bits))


def test_mulaw_and_inverse():
  from wavenets_amd import ops
  x = torch.linspace(-1, 1, 100001)
  y = ops.mu_law(x.to(dev())).cpu()
  assert (y - O.mu_law(x)).abs().max() < 2e-6
  z = ops.inverse_mu_law(y.to(dev())).cpu()
  assert (z - x).abs().max() < 5e-6
  assert (ops.inverse_mu_law(y.to(dev())).cpu() - O.inverse_mu_law(y)).abs().max() < 2e-6


# ------------------------------------------------------------------------------------------
# standalone residual block
# ------------------------------------------------------------------------------------------
def _layer_pair(R, D, S, dil, k=2, act=None, residual=True, cin=None, cond_c=0, seed=0):
  from wavenets_amd import WaveNetLayer
  layer = WaveNetLayer(kernel=k, dilation_rate=dil, activation=act, channels=R, residual=residual,
                       dilation_channels=D, skip_channels=S, condition=cond_c > 0, device=dev(), seed=seed)
  cin = R if cin is None else cin
  layer.build([(2, 16, cin), (2, 16, cond_c)] if cond_c else (2, 16, cin))
  g = torch.Generator().manual_seed(seed + 1)
  flat = (torch.rand(layer.flat_params.numel(), generator=g) * 2 - 1) * 0.2
  with torch.no_grad():
    layer.flat_params.copy_(flat.to(dev()))
  # oracle parameter list in the same order
  ps = []
  dl = dil if isinstance(dil, list) else [dil]
  for c in layer.dilated_stack:
    ps += [c.kernel.detach().cpu(), c.bias.detach().cpu()]
  ps += [layer.conv1.kernel.detach().cpu(), layer.conv1.bias.detach().cpu()]
  if S is not None:
    ps += [layer.conv_skip.kernel.detach().cpu(), layer.conv_skip.bias.detach().cpu()]
  if cond_c:
    ps += [layer.conv_cond.kernel.detach().cpu(), layer.conv_cond.bias.detach().cpu()]
  return layer, ps, dl


LAYER_CASES = [
    # R, D, S, dilation, k, act, residual, T  -- fused shapes
    (32, 32, None, 1, 2, None, True, 100),
    (32, 32, 64, 8, 2, None, True, 257),
    (64, 64, 256, 4, 2, None, True, 300),
    (64, 64, 256, 512, 2, None, True, 1100),
    (64, 64, None, 2, 3, None, False, 97),
    (128, 128, 256, 16, 2, None, True, 130),
    # composed path: odd sizes, depth > 1
    (8, 12, 20, 2, 2, None, True, 70),
    (6, 6, None, 3, 3, None, True, 65),
    (32, 32, 48, [1, 2, 4], 2, 'leaky_relu', True, 90),
    (16, 24, 40, [2, 1], 2, 'tanh', True, 77),
]


@pytest.mark.parametrize('R,D,S,dil,k,act,residual,T', LAYER_CASES)
def test_layer_forward_parity(R, D, S, dil, k, act, residual, T, math_mode):
  layer, ps, dl = _layer_pair(R, D, S, dil, k, act, residual)
  x = torch.randn(3, T, R, generator=torch.Generator().manual_seed(5)) * 0.7
  xo_ref, sk_ref = O.layer_forward(x.double(), [p.double() for p in ps], dilations=dl, activation_name=act,
                                   residual=residual, has_skip=S is not None)
  with torch.no_grad():
    xo, sk = layer(x.to(dev()))
  assert (xo.cpu().double() - xo_ref).abs().max() < ATOL_ACT
  assert (sk.cpu().double() - sk_ref).abs().max() < ATOL_ACT


def test_layer_forward_with_time_varying_condition():
  layer, ps, dl = _layer_pair(32, 32, 64, 2, cond_c=5)
  x = torch.randn(2, 80, 32, generator=torch.Generator().manual_seed(1))
  c = torch.randn(2, 80, 5, generator=torch.Generator().manual_seed(2))
  xo_ref, sk_ref = O.layer_forward(x.double(), [p.double() for p in ps], dilations=dl, activation_name=None,
                                   residual=True, has_skip=True, cond=c.double())
  with torch.no_grad():
    xo, sk = layer((x.to(dev()), c.to(dev())))
  assert (xo.cpu().double() - xo_ref).abs().max() < ATOL_ACT
  assert (sk.cpu().double() - sk_ref).abs().max() < ATOL_ACT


@pytest.mark.parametrize('R,D,S,dil,k,act,residual,T', [LAYER_CASES[i] for i in (1, 2, 4, 6, 8, 9)])
def test_layer_backward_parity(R, D, S, dil, k, act, residual, T):
  layer, ps, dl = _layer_pair(R, D, S, dil, k, act, residual)
  g = torch.Generator().manual_seed(11)
  x = torch.randn(2, T, R, generator=g) * 0.7
  gx = torch.randn(2, T, R, generator=g)
  gs = torch.randn(2, T, S or R, generator=g)
  # oracle (fp64 autograd)
  xr = x.double().requires_grad_(True)
  pr = [p.double().requires_grad_(True) for p in ps]
  xo, sk = O.layer_forward(xr, pr, dilations=dl, activation_name=act, residual=residual, has_skip=S is not None)
  obj = (xo * gx.double()).sum() + (sk * gs.double()).sum()
  ref = torch.autograd.grad(obj, [xr] + pr)
  # HIP
  xd = x.to(dev()).requires_grad_(True)
  xo_d, sk_d = layer(xd)
  obj_d = (xo_d * gx.to(dev())).sum() + (sk_d * gs.to(dev())).sum()
  obj_d.backward()
  assert _rel(xd.grad, ref[0]) < 2e-5
  flat_ref = torch.cat([r.reshape(-1) for r in ref[1:]])
  got = layer.flat_params.grad.cpu()
  # compare tensor by tensor relative to each tensor's scale
  off = 0
  for r in ref[1:]:
    n = r.numel()
    assert _rel(got[off:off + n], r.reshape(-1)) < 5e-5
    off += n
  assert off == flat_ref.numel() == got.numel()


def test_layer_generate_single_step():
  # WaveNetLayer.generate (src/layers.py:226-290): gathered [x[t-d], x[t]] -> one step
  layer, ps, dl = _layer_pair(32, 32, 64, 4)
  x = torch.randn(2, 40, 32, generator=torch.Generator().manual_seed(3))
  with torch.no_grad():
    full_x, full_s = layer(x.to(dev()))
    t = 17
    gathered = torch.stack([x[:, t - 4], x[:, t]], dim=1).to(dev())
    gx, gs = layer.generate(gathered)
  assert gx.shape == (2, 1, 32) and gs.shape == (2, 1, 64)
  assert (gx[:, 0] - full_x[:, t]).abs().max() < 1e-5
  assert (gs[:, 0] - full_s[:, t]).abs().max() < 1e-5


@pytest.mark.parametrize('R,D,S,dil,k,cond_c', [(32, 32, 64, 4, 2, 0), (64, 64, 256, 512, 2, 0), (32, 32, None, 1, 2, 0),
                                                (32, 32, 64, 2, 2, 5), (64, 64, None, 3, 3, 0)])
def test_layer_generate_matches_oracle(R, D, S, dil, k, cond_c, math_mode):
  """G3 against the oracle's restatement of src/layers.py:226-290 (not against the layer's own forward)."""
  layer, ps, dl = _layer_pair(R, D, S, dil, k=k, cond_c=cond_c, seed=2)
  g = torch.Generator().manual_seed(11)
  gathered = torch.randn(3, k, R, generator=g) * 0.8
  cond = torch.randn(3, 1, cond_c, generator=g) if cond_c else None
  ref_x, ref_s = O.layer_generate(gathered.double(), [p.double() for p in ps], has_skip=S is not None,
                                  cond=cond.double() if cond is not None else None)
  with torch.no_grad():
    gx, gs = layer.generate((gathered.to(dev()), cond.to(dev())) if cond_c else gathered.to(dev()))
  assert gx.shape == (3, 1, R) and gs.shape == (3, 1, S if S is not None else R)
  assert (gx.cpu().double() - ref_x).abs().max() < ATOL_ACT
  assert (gs.cpu().double() - ref_s).abs().max() < ATOL_ACT
  # and the oracle's single step is the oracle's full forward at that step (the two restatements agree)
  d = dl[0]
  x = torch.randn(3, (k - 1) * d + 3, R, generator=g).double()
  t = x.shape[1] - 1
  taps = torch.stack([x[:, t - (k - 1 - j) * d] for j in range(k)], dim=1)
  cfull = cond.double().expand(3, x.shape[1], cond_c) if cond_c else None
  fx, fs = O.layer_forward(x, [p.double() for p in ps], dilations=dl, activation_name=None, residual=True,
                           has_skip=S is not None, cond=cfull)
  sx, ss = O.layer_generate(taps, [p.double() for p in ps], has_skip=S is not None,
                            cond=cond.double() if cond is not None else None)
  assert (sx[:, 0] - fx[:, t]).abs().max() < 1e-12 and (ss[:, 0] - fs[:, t]).abs().max() < 1e-12


def test_layer_errors():
  from wavenets_amd import WaveNetLayer
  l = WaveNetLayer(channels=32, device=dev())
  with pytest.raises(ValueError, match='Layer is not built'):
    l.compute_output_shape((1, 10, 32))
  with pytest.raises(ValueError, match='Residual connection must have the same shape as input'):
    l.build((1, 10, 16))
  lc = WaveNetLayer(channels=32, condition=True, device=dev())
  with pytest.raises(ValueError, match='Condition tensor must have the same length as input'):
    lc.build([(1, 10, 32), (1, 9, 4)])
  l2 = WaveNetLayer(channels=32, skip_channels=48, device=dev())
  l2.build((2, 10, 32))
  assert l2.compute_output_shape((2, 10, 32)) == ((2, 10, 32), (2, 10, 48))


# ------------------------------------------------------------------------------------------
# whole model: forward
# ------------------------------------------------------------------------------------------
MODEL_CASES = {
    'cat_small_fused': dict(blocks=6, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[48, 40],
                            activation='leaky_relu', bits=8),
    'cat_noskipch': dict(blocks=5, channels=32, dilation_bound=16, final_layers_channels=[], bits=8),
    'cat_r64': dict(blocks=4, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
                    activation='leaky_relu', bits=8),
    # 32-channel blocks under a 128-wide first head conv: the folded skip path's M = Z^T dL/da as ONE transposed-read job
    # over five of its eight z segments (wn_wgrad_tr kind 8; 'cat_r64' fills a kind-7 job, the 30-block nets end on a
    # two-segment job)
    'cat_r32_f128': dict(blocks=5, channels=32, skip_channels=96, dilation_bound=16, final_layers_channels=[128, 48],
                         activation='leaky_relu', bits=8),
    'cat_r128': dict(blocks=3, channels=128, skip_channels=64, dilation_bound=8, final_layers_channels=[64],
                     activation='leaky_relu', bits=8),
    'cat_odd_composed': dict(blocks=4, channels=12, dilation_channels=10, skip_channels=20, dilation_bound=4,
                             final_layers_channels=[9], activation='relu', bits=5),
    'cat_lpb2_odd': dict(blocks=3, layers_per_block=2, kernel_size=3, channels=12, dilation_channels=10, skip_channels=20,
                         dilation_bound=9, final_layers_channels=[9], activation='relu', bits=5),
    'cat_lpb3': dict(blocks=3, layers_per_block=3, channels=32, skip_channels=32, dilation_bound=8,
                     final_layers_channels=[32], activation='tanh', bits=6),
    # deeper stacks on the split-precision training path (deep16): 64 channels (unpadded images, INNER<2> weight gradients)
    # and kernel size 3 (inner gradients' max-abs slots feeding the generic job table)
    'cat_lpb2_r64': dict(blocks=3, layers_per_block=2, channels=64, skip_channels=64, dilation_bound=8,
                         final_layers_channels=[64], activation='leaky_relu', bits=8),
    'cat_lpb2_k3_r32': dict(blocks=3, layers_per_block=2, kernel_size=3, channels=32, skip_channels=32, dilation_bound=9,
                            final_layers_channels=[32], activation='relu', bits=6),
    'cat_k3': dict(blocks=4, kernel_size=3, channels=32, skip_channels=32, dilation_bound=9,
                   final_layers_channels=[], bits=6),
    'cat_noskip_nores': dict(blocks=3, channels=32, skip_channels=32, dilation_bound=4, use_skip=False,
                             use_residual=False, final_layers_channels=[16], activation='elu', bits=6),
    'mol': dict(blocks=4, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[32],
                activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16),
    'mol_r128_k3': dict(blocks=5, kernel_size=3, channels=128, skip_channels=96, dilation_bound=81,
                        final_layers_channels=[64], activation='leaky_relu', num_mixtures=3,
                        sampling_function='logistic', bits=16),
    'gauss': dict(blocks=4, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[32],
                  activation='leaky_relu', num_mixtures=8, sampling_function='gaussian', bits=16),
    'cond': dict(blocks=4, channels=32, skip_channels=64, dilation_bound=8, final_layers_channels=[32],
                 activation='leaky_relu', conditioning='global', mapping_layers=[8, 16, 32],
                 mapping_activation='leaky_relu', bits=8, cond_inputs=11),
    'cond_nomap': dict(blocks=3, channels=32, dilation_bound=8, final_layers_channels=[],
                       conditioning='global', bits=6, cond_inputs=4),
}


def _inputs(kw, B, T, seed=3):
  x = O.synthetic_waveform(B, T, seed=seed)
  cond = None
  if kw.get('conditioning') is not None:
    cond = torch.rand(B, kw['cond_inputs'], generator=torch.Generator().manual_seed(seed))
  return x, cond


@pytest.mark.parametrize('name', list(MODEL_CASES))
def test_model_forward_parity(name, math_mode):
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=1, **kw)
  B, T = 3, 333
  x, cond = _inputs(kw, B, T)
  ref, inter = O.model_forward(x.double(), [p.double() for p in params], ocfg,
                               cond.double() if cond is not None else None, return_intermediates=True)
  inp = (x.to(dev()), cond.to(dev())) if cond is not None else x.to(dev())
  out = model(inp)
  assert out.shape == ref.shape
  assert (out.cpu().double() - ref).abs().max() < ATOL_ACT
  lg = model.logits(inp)
  assert (lg.cpu().double() - inter['logits']).abs().max() < ATOL_ACT


def test_model_forward_cfg1_full_size(math_mode):
  # BASELINE configs[0]: 10-layer mu-law-256, dilations 1..512, 32 residual ch, batch 1 x 16000
  kw = dict(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[], bits=8)
  ocfg, params, model = make_pair(seed=2, **kw)
  x = O.synthetic_waveform(1, 16000, seed=9)
  ref = O.model_forward(x, params, ocfg)                  # fp32 oracle at full size
  out = model(x.to(dev()))
  assert (out.cpu() - ref).abs().max() < ATOL_ACT
  # bit-exact arg-max indices wherever the oracle's top-2 margin exceeds the tolerance
  top2 = torch.topk(ref, 2, dim=-1).values
  clear = (top2[..., 0] - top2[..., 1]) > 1e-4
  assert torch.equal(out.cpu().argmax(-1)[clear], ref.argmax(-1)[clear])


# ------------------------------------------------------------------------------------------
# training step: loss, gradients, optimizer
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', list(MODEL_CASES))
def test_loss_and_gradients_parity(name, math_mode):
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=4, **kw)
  B, T = 2, 150
  x, cond = _inputs(kw, B, T + 1, seed=8)
  loss_ref, _, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg,
                                               cond.double() if cond is not None else None)
  data = (x.to(dev()), cond.to(dev())) if cond is not None else x.to(dev())
  loss, _, _ = model.loss_and_grads(data)
  assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))
  names = model.variable_names
  for n, g, r in zip(names, model.gradients(), grads_ref):
    scale = max(r.abs().max().item(), 1e-6)
    err = (g.cpu().double() - r).abs().max().item()
    assert err < 1e-4 * scale + 1e-7, (n, err, scale)


def test_l2_regulariser():
  kw = dict(MODEL_CASES['cat_small_fused'], l2_reg_factor=0.01)
  ocfg, params, model = make_pair(seed=4, **kw)
  x, _ = _inputs(kw, 2, 101, seed=2)
  loss_ref, reg_ref, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg)
  loss, _, _ = model.loss_and_grads(x.to(dev()))
  assert abs(loss[1].item() - reg_ref.item()) < 1e-5 * max(1.0, reg_ref.item())
  for g, r in zip(model.gradients(), grads_ref):
    assert (g.cpu().double() - r).abs().max().item() < 1e-4 * max(r.abs().max().item(), 1e-6) + 1e-7


def test_global_batch_scaling_for_data_parallel():
  # a replica holding half of a global batch of 4: loss and grads scaled by 1/4 (src/model.py:328-329)
  kw = dict(MODEL_CASES['cat_small_fused'])
  ocfg, params, model = make_pair(seed=4, **kw)
  x, _ = _inputs(kw, 4, 120, seed=5)
  _, _, g_all, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg)
  acc = None
  for half in (x[:2], x[2:]):
    model.loss_and_grads(half.to(dev()), global_batch=4, n_replicas=2)
    acc = model.flat_grads.clone() if acc is None else acc + model.flat_grads
  ref = torch.cat([g.reshape(-1) for g in g_all])
  assert (acc.cpu().double() - ref).abs().max() < 1e-4 * ref.abs().max()


@pytest.mark.parametrize('name', ['cat_small_fused', 'mol'])
def test_three_train_steps_match_oracle(name):
  from wavenets_amd import Adam
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=6, **kw)
  model.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
  x, _ = _inputs(kw, 2, 129, seed=1)
  p = [q.double() for q in params]
  m = [torch.zeros_like(q) for q in p]
  v = [torch.zeros_like(q) for q in p]
  for step in range(1, 4):
    loss_ref, p, m, v = O.train_step(x.double(), p, m, v, step, ocfg, lr=5e-4, clipnorm=1.0)
    logs = model.train_step(x.to(dev()))
    assert set(logs) == {'loss'}
  for n, got, ref in zip(model.variable_names, model.trainable_variables, p):
    assert (got.cpu().double() - ref).abs().max().item() < 2e-6 + 1e-4 * 5e-4 * 3, n
  assert model.optimizer.iterations == 3


def test_adam_clip_active_and_inactive():
  # per-tensor clipnorm: big gradient clipped, small one untouched (train.py:225-226)
  from wavenets_amd import Adam
  kw = dict(MODEL_CASES['cat_noskipch'])
  ocfg, params, model = make_pair(seed=6, **kw)
  opt = Adam(learning_rate=1e-2, clipnorm=1.0)
  model.compile(optimizer=opt)
  g = torch.Generator().manual_seed(0)
  grads = []
  for i, p in enumerate(params):
    gr = torch.randn(p.shape, generator=g, dtype=torch.float64)
    gr = gr / gr.norm() * (5.0 if i % 2 == 0 else 0.3)
    grads.append(gr)
  model.flat_grads.copy_(torch.cat([q.reshape(-1) for q in grads]).float().to(dev()))
  before = [q.double() for q in params]
  ref, _, _ = O.keras_adam_step(before, O.clip_by_norm_per_tensor(grads, 1.0),
                                [torch.zeros_like(q) for q in before], [torch.zeros_like(q) for q in before], 1, 1e-2)
  opt.apply_gradients(model)
  for got, r in zip(model.trainable_variables, ref):
    assert (got.cpu().double() - r).abs().max() < 1e-6


def test_test_step_and_loss_fn():
  kw = dict(MODEL_CASES['cat_small_fused'])
  ocfg, params, model = make_pair(seed=4, **kw)
  model.compile(optimizer=None)
  x, _ = _inputs(kw, 2, 90, seed=5)
  out = model.test_step(x.to(dev()))
  pred = O.model_forward(x[:, :-1], params, ocfg)
  tgt = O.quantize(x[:, 1:], 8)
  per = O.loss_categorical(tgt, pred)
  assert abs(out['loss'] - per.sum().item() / 2) < 1e-4 * per.sum().item()
  # WaveNet.loss_fn(target, pred) on materialised probabilities
  got = model.loss_fn(tgt.to(dev()), pred.to(dev()))
  assert (got.cpu() - per).abs().max() < 1e-5
  # prepare_target is the bit-exact quantiser
  assert torch.equal(model.prepare_target(x[:, 1:].to(dev())).cpu().long(), tgt)


@pytest.mark.parametrize('name', ['mol', 'gauss'])
def test_mixture_loss_fn_and_samplers(name):
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=4, **kw)
  g = torch.Generator().manual_seed(0)
  pred = torch.randn(2, 50, 3 * kw['num_mixtures'], generator=g)
  y = torch.rand(2, 50, 1, generator=g) * 2 - 1
  # fp64 oracle: at bits=16 the logistic bin mass sigmoid(a)-sigmoid(b) cancels ~5 digits, so an
  # fp32 evaluation of the reference formula is itself only good to ~1e-3; the kernel evaluates the
  # row in double from the fp32 inputs
  ref = O.loss_fn(y.double(), pred.double(), ocfg)
  got = model.loss_fn(y.to(dev()), pred.to(dev()))
  assert (got.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
  s = model.sample_waveform(pred.to(dev()), deterministic=True)
  assert torch.equal(s.cpu(), O.sample_waveform_deterministic(pred, ocfg))
  r = model.sample_waveform(pred.to(dev()), deterministic=False)
  assert r.shape == (2, 50, 1) and r.abs().max() <= 1.0


@pytest.mark.parametrize('name', ['mol', 'gauss'])
def test_mixture_stochastic_sampler_distribution(name):
  """S1, non-deterministic mixture draws (src/model.py:423-443, 463-483): the empirical CDF of n draws from one
  prediction row against the analytic law (component ~ softmax(w); mu + e^s * logistic / normal noise; clipped to
  [-1, 1]).  Dvoretzky-Kiefer-Wolfowitz: P(sup|F_n - F| > eps) <= 2 exp(-2 n eps^2) = 1.1e-6 at these sizes."""
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=4, **kw)
  M = kw['num_mixtures']
  g = torch.Generator().manual_seed(3)
  row = torch.cat([torch.randn(M, generator=g), torch.rand(M, generator=g) * 1.6 - 0.8,
                   torch.rand(M, generator=g) * 3.0 - 4.0])          # scales e^-4 .. e^-1: both atoms get mass
  n = 200000
  eps = math.sqrt(math.log(2 / 1.1e-6) / (2 * n))
  big = row.expand(1, n, 3 * M).contiguous()
  draws = model.sample_waveform(big.to(dev()), deterministic=False).cpu().reshape(-1).double()
  assert draws.abs().max() <= 1.0
  v = torch.linspace(-1.0, 0.9999, 400, dtype=torch.float64)
  Fn = (draws.unsqueeze(0) <= v.unsqueeze(1)).double().mean(1)
  F = O.mixture_sample_cdf(row, ocfg, v)
  assert (Fn - F).abs().max().item() < eps, (Fn - F).abs().max().item()
  # the component pick follows softmax(w): with tiny scales every draw sits on its component's mean
  tight = torch.cat([row[:M], row[M:2 * M], torch.full((M,), -12.0)])
  d2 = model.sample_waveform(tight.expand(1, n, 3 * M).contiguous().to(dev()), deterministic=False).cpu().reshape(-1)
  comp = (d2.unsqueeze(1) - row[M:2 * M].unsqueeze(0)).abs().argmin(1)
  counts = torch.bincount(comp, minlength=M).double()
  w = torch.softmax(row[:M].double(), -1)
  chi2 = (((counts - n * w) ** 2) / (n * w)).sum().item()
  assert chi2 < 70.0, chi2                                # M-1 <= 9 dof: P(chi2 > 70) < 1e-10
  # a second call draws a different stream (per-call offset), rows within a call are not all alike
  d3 = model.sample_waveform(big.to(dev()), deterministic=False).cpu().reshape(-1).double()
  assert not torch.equal(d3, draws) and draws.unique().numel() > n // 4


def test_weights_io_round_trip_with_optimizer_state(tmp_path):
  """train.py:149-154,237-238 convention: save -> load into a fresh model restores weights (+ Adam moments and
  step, which the reference drops) so that the next training step is bit-identical."""
  from wavenets_amd import WaveNet, Adam, io
  kw = dict(MODEL_CASES['cat_small_fused'])
  x, _ = _inputs(kw, 2, 200, seed=5)
  a = WaveNet(**kw, device=dev(), seed=1)
  a.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
  for _ in range(3):
    a.train_step(x.to(dev()))
  path = str(tmp_path / io.checkpoint_name(3, 5e-4))
  io.save_weights(a, path, optimizer=a.optimizer)
  b = WaveNet(**kw, device=dev(), seed=2)
  b.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
  assert not torch.equal(a.flat_params.data, b.flat_params.data)
  io.load_weights(b, path, optimizer=b.optimizer)
  assert torch.equal(a.flat_params.data, b.flat_params.data)
  assert b.optimizer.iterations == a.optimizer.iterations == 3
  assert torch.equal(a.optimizer.m, b.optimizer.m) and torch.equal(a.optimizer.v, b.optimizer.v)
  la, lb = a.train_step(x.to(dev())), b.train_step(x.to(dev()))
  assert torch.equal(a.flat_params.data, b.flat_params.data)
  assert io.find_resume(str(tmp_path))[1:] == (3, 5e-4)
  # weights only (what the reference stores): the moments restart from zero
  c = WaveNet(**kw, device=dev(), seed=3)
  io.load_weights(c, path)
  with np.load(path) as d:
    assert all(np.array_equal(w, d[f'w{i:03d}']) for i, w in enumerate(c.get_weights()))


@pytest.mark.parametrize('name', ['cat_small_fused', 'cond', 'cat_lpb3'])
def test_keras_weights_h5_round_trip(tmp_path, name):
  """`.weights.h5` exchange files (train.py:149-154): export from one model, import into another -> identical
  variables and identical logits; the tree carries the Keras-3 attribute paths."""
  from wavenets_amd import WaveNet, io, h5
  kw = dict(MODEL_CASES[name])
  cond_inputs = kw.pop('cond_inputs', 0)
  a = WaveNet(**kw, device=dev(), seed=11)
  b = WaveNet(**kw, device=dev(), seed=12)
  if cond_inputs:
    a.build([(1, 8, 1), (1, cond_inputs)])
    b.build([(1, 8, 1), (1, cond_inputs)])
  path = str(tmp_path / io.checkpoint_name(4, 0.0005, 'h5'))
  io.save_weights(a, path)
  tree = h5.read_h5(path)
  assert tree['causal']['vars']['0'].shape == (kw.get('kernel_size', 2), 1, kw['channels'])
  assert len(tree['wavenet_blocks']) == kw['blocks']
  assert len(tree['wavenet_blocks']['wave_net_layer']['dilated_stack']) == kw.get('layers_per_block', 1)
  assert ('mapping' in tree) == (cond_inputs > 0)
  io.load_weights(b, path)
  assert torch.equal(a.flat_params.data, b.flat_params.data)
  x, cond = _inputs(dict(kw, cond_inputs=cond_inputs), 2, 90, seed=1)
  inp = (x.to(dev()), cond.to(dev())) if cond is not None else x.to(dev())
  assert torch.equal(a.logits(inp), b.logits(inp))
  assert io.find_resume(str(tmp_path)) == (path, 4, 0.0005)


def test_categorical_samplers():
  kw = dict(MODEL_CASES['cat_small_fused'])
  ocfg, params, model = make_pair(seed=4, **kw)
  g = torch.Generator().manual_seed(0)
  probs = torch.softmax(torch.randn(3, 40, 256, generator=g) * 2, -1)
  s = model.sample_waveform(probs.to(dev()), deterministic=True)
  assert torch.equal(s.cpu(), O.sample_waveform_deterministic(probs, ocfg))
  # stochastic draw: chi-square of bin counts against the probabilities (one row repeated)
  row = torch.softmax(torch.randn(16, generator=g), -1)
  p16 = torch.zeros(256); p16[:16] = row
  n = 200000
  big = p16.expand(1, n, 256).contiguous()
  draws = model.sample_waveform(big.to(dev()), deterministic=False).cpu().reshape(-1)
  idx = torch.round((draws + 1.0) * 128).long()
  assert idx.max() < 16
  counts = torch.bincount(idx, minlength=16).double()[:16]
  chi2 = (((counts - n * row.double()) ** 2) / (n * row.double())).sum().item()
  assert chi2 < 60.0, chi2          # 15 dof: P(chi2 > 60) ~ 2e-7


# ------------------------------------------------------------------------------------------
# generation
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['cat_noskipch', 'mol'])
def test_naive_generation_matches_oracle(name):
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=7, bias_range=0.3, **kw)
  rf = O.receptive_field(ocfg)
  assert model.receptive_field == rf
  w = O.synthetic_waveform(2, rf, seed=4)
  ref = O.generate_naive(params, ocfg, 12, w)
  out = model.generate(12, sample=w.to(dev()), deterministic=True)
  assert out.shape == (2, 12, 1)
  if name == 'mol':
    assert (out.cpu() - ref).abs().max() < 1e-4
  else:
    # indices: compare exactly; a mismatch is tolerated only where the oracle's own top-2
    # probabilities are closer than the activation tolerance (then later steps may diverge)
    same = torch.equal(out.cpu(), ref)
    if not same:
      first = (out.cpu() != ref).nonzero()[0]
      pytest.fail(f'generated indices differ first at {first.tolist()}')


@pytest.mark.parametrize('queued', [False, True])
@pytest.mark.parametrize('name', ['cat_r64', 'cat_small_fused'])
def test_generation_range_guard_repeats_in_exact_fp32(name, queued):
  """The split-precision kernels cast activations to fp16 hi | lo unscaled (|.| < 65504).  A residual stream of ~1e5
  (huge input-conv bias) must trip the generate call's guard slot -- priming pass AND the per-step kernels, fused chain
  included -- and the call must come back as the exact-fp32 result (src/model.py:258-307)."""
  from wavenets_amd import _lib
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=11, bias_range=0.1, **kw)
  names = model.variable_names
  ws = [p.clone() for p in params]
  ws[names.index('causal/bias')] = torch.full_like(ws[names.index('causal/bias')], 1.0e5)
  model.set_weights([w.numpy() for w in ws])
  w0 = O.synthetic_waveform(2, model.receptive_field, seed=3).to(dev())
  out = model.generate(6, sample=w0, deterministic=True, use_queues=queued)
  assert model.generation_guard_trips == 1
  with model.exact_fp32():
    ref = model.generate(6, sample=w0, deterministic=True, use_queues=queued)
  assert model.generation_guard_trips == 1            # (exact mode never trips)
  assert torch.isfinite(out).all() and torch.equal(out, ref)
  # and a well-scaled net leaves the slot far below the limit
  ocfg2, params2, model2 = make_pair(seed=11, bias_range=0.1, **kw)
  model2.generate(6, sample=w0, deterministic=True, use_queues=queued)
  L = _lib.lib()
  slot = L.wn_generate_guard_slot(model2._plan, 2, int(queued))
  v = float(model2._ws['gen'][slot])
  assert 0.0 < v < 100.0 and getattr(model2, 'generation_guard_trips', 0) == 0


@pytest.mark.parametrize('name', ['cat_small_fused', 'cat_r64', 'cat_r128'])
def test_generation_range_guard_catches_an_overflow_that_only_a_generated_sample_causes(name):
  """The priming pass stays in range (an all-zero window: the residual stream is the input conv's bias), but the first
  GENERATED sample is non-zero and a huge input-conv kernel turns it into a residual stream of ~1e4..1e5 from queued step 1
  on -- steps the chain kernel used to publish only every 32nd time.  The guard slot must be raised in that very step
  and the call repeated in exact fp32 (src/model.py:296-305)."""
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=11, bias_range=0.1, **kw)
  names = model.variable_names
  ws = [p.clone() for p in params]
  ck = names.index('causal/kernel')
  ws[ck] = torch.full_like(ws[ck], 6.0e5)             # (k, 1, R): samples are multiples of 1/128 -> |H[0]| >= 4.7e3 * k
  model.set_weights([w.numpy() for w in ws])
  w0 = torch.zeros(2, model.receptive_field, 1, device=dev())
  n = 7                                               # steps 1..6 are queued steps; none is a multiple of 32, 6 is the last
  with model.exact_fp32():
    ref = model.generate(n, sample=w0, deterministic=True, use_queues=True)
  assert model.generation_guard_trips == 0
  if float(ref[:, :n - 2].abs().max()) < 0.06:
    pytest.skip('the first samples of this seed are (almost) zero: no overflow to provoke')
  out = model.generate(n, sample=w0, deterministic=True, use_queues=True)
  assert model.generation_guard_trips == 1
  assert torch.isfinite(out).all() and torch.equal(out, ref)


def test_generate_errors():
  from wavenets_amd import WaveNet
  m = WaveNet(blocks=2, channels=32, dilation_bound=4, final_layers_channels=[], conditioning='global',
              mapping_layers=[4], device=dev())
  with pytest.raises(ValueError, match='Conditioning must be provided'):
    m.generate(3)
  with pytest.raises(ValueError, match='same batch size'):
    m.generate(3, condition=torch.zeros(2, 5), sample=torch.zeros(3, m.receptive_field, 1))
  m2 = WaveNet(blocks=2, channels=32, dilation_bound=4, final_layers_channels=[], device=dev())
  with pytest.raises(ValueError, match='Loss must be set in the model init function'):
    m2.compile(loss='mse')
  assert abs(m2.compute_receptive_field(16000) - m2.receptive_field / 16000) < 1e-12


# ------------------------------------------------------------------------------------------
# queued (ring-buffer) generation == sliding window, index for index (README.md:16 TODO of the
# reference; A/B hook src/callbacks.py:58-68)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,det', [('cat_noskipch', True), ('cat_small_fused', True), ('cat_r64', True),
                                      ('mol', True), ('cond', True), ('cat_odd_composed', True),
                                      ('cat_k3', True), ('cat_noskip_nores', True), ('cat_small_fused', False),
                                      ('cat_r128', True)])
def test_queued_generation_equals_naive(name, det):
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=7, bias_range=0.3, **kw)
  rf = model.receptive_field
  B, n = 3, 25
  w = O.synthetic_waveform(B, rf, seed=4).to(dev())
  cond = None
  if kw.get('conditioning') is not None:
    cond = torch.rand(B, kw['cond_inputs'], generator=torch.Generator().manual_seed(2)).to(dev())
  naive = model.generate(n, condition=cond, sample=w, use_queues=False, deterministic=det)
  queued = model.generate(n, condition=cond, sample=w, use_queues=True, deterministic=det)
  assert queued.shape == (B, n, 1)
  assert torch.equal(naive, queued), (naive - queued).abs().max()


def test_queued_generation_kernel_size_3_keeps_full_history():
  """kernel_size = 3: the reference's receptive_field (src/model.py:122, 1 + sum(d) (k - 1) + 1) is k - 2
  samples shorter than the network's true reach (sum(d) (k - 1) + k), so its sliding window zero-pads one
  sample the ring buffers still hold.  The sliding window here reproduces the reference (truncated window);
  the queued sampler is the untruncated recurrence.  Both are checked against the oracle."""
  kw = dict(MODEL_CASES['mol_r128_k3'])
  ocfg, params, model = make_pair(seed=7, bias_range=0.3, **kw)
  rf = O.receptive_field(ocfg)
  assert model.receptive_field == rf and (rf - 2) % 2 == 0      # 1 + sum(d) (k - 1) + 1
  w = O.synthetic_waveform(2, rf, seed=4)
  n = 6
  naive = model.generate(n, sample=w.to(dev()), use_queues=False, deterministic=True)
  assert (naive.cpu() - O.generate_naive(params, ocfg, n, w)).abs().max() < 1e-4
  queued = model.generate(n, sample=w.to(dev()), use_queues=True, deterministic=True)
  x = w.clone()
  full = []
  with torch.no_grad():
    for _ in range(n):                                   # growing sequence: nothing is dropped on the left
      pred = O.model_forward(x, params, ocfg)[:, -1:, :]
      full.append(O.sample_waveform_deterministic(pred, ocfg))
      x = torch.cat([x, full[-1].to(x.dtype)], dim=1)
  assert (queued.cpu() - torch.cat(full, dim=1)).abs().max() < 1e-4
  assert torch.equal(queued[:, 0], naive[:, 0])          # the first sample sees the same window


def test_queued_generation_with_stacked_dilated_convs():
  """layers_per_block > 1 is the blocker the reference names for its queue TODO (README.md:16,
  src/layers.py:226-290): here every dilated conv of the stack has its own input ring."""
  for name in ('cat_lpb3', 'cat_lpb2_odd'):
    kw = dict(MODEL_CASES[name])
    ocfg, params, model = make_pair(seed=7, bias_range=0.3, **kw)
    w = O.synthetic_waveform(2, model.receptive_field, seed=5).to(dev())
    naive = model.generate(20, sample=w, use_queues=False, deterministic=True)
    queued = model.generate(20, sample=w, use_queues=True, deterministic=True)
    assert torch.equal(naive, queued), name


# ------------------------------------------------------------------------------------------
# dropout (reference default config: dropout 0.1, train.py:39) -- stateless hash mask restated by the oracle
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['cat_small_fused', 'cat_lpb3', 'cat_odd_composed', 'cat_noskipch'])
def test_dropout_training_step_parity(name, math_mode):
  from wavenets_amd import WaveNet, _lib
  kw = dict(MODEL_CASES[name])
  rate, seed = 0.25, 77
  ocfg = O.OracleConfig(**kw)
  params = O.init_params(ocfg, seed=4)
  model = WaveNet(**kw, dropout=rate, device=dev(), seed=seed)
  model.set_weights([p.numpy() for p in params])
  B, T = 2, 140
  x, _ = _inputs(kw, B, T + 1, seed=8)
  for step in (1, 2):                     # the library pre-increments its step counter per call
    loss_ref, _, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg,
                                                 dropout=(rate, seed, step))
    loss, _, _ = model.loss_and_grads(x.to(dev()))
    assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))
    for n, g, r in zip(model.variable_names, model.gradients(), grads_ref):
      scale = max(r.abs().max().item(), 1e-6)
      assert (g.cpu().double() - r).abs().max().item() < 1e-4 * scale + 1e-7, (n, step)
  # masks differ between steps and the keep rate is right
  k1 = O.dropout_keep_mask(200000, O.dropout_key(seed, 0, 1), rate)
  k2 = O.dropout_keep_mask(200000, O.dropout_key(seed, 0, 2), rate)
  assert not torch.equal(k1, k2)
  assert abs(k1.float().mean().item() - (1 - rate)) < 5e-3
  assert _lib.lib().wn_dropout_key_for(seed, 3, 9) == O.dropout_key(seed, 3, 9)
  # inference ignores dropout
  ref = O.model_forward(x[:, :-1].double(), [p.double() for p in params], ocfg)
  assert (model(x[:, :-1].to(dev())).cpu().double() - ref).abs().max() < ATOL_ACT


def test_call_training_true_applies_dropout(math_mode):
  """WaveNet.call(inputs, training=True) (src/model.py:233 -> src/layers.py:195-196): a stochastic forward whose mask
  is that of the model's next training call; the oracle restates the mask hash."""
  from wavenets_amd import WaveNet
  kw = dict(MODEL_CASES['cat_small_fused'])
  rate, seed = 0.3, 5
  ocfg = O.OracleConfig(**kw)
  params = O.init_params(ocfg, seed=2)
  model = WaveNet(**kw, dropout=rate, device=dev(), seed=seed)
  model.set_weights([p.numpy() for p in params])
  x, _ = _inputs(kw, 2, 200, seed=4)
  pd = [p.double() for p in params]
  for step in (1, 2):
    got = model(x.to(dev()), training=True)
    ref = O.model_forward(x.double(), pd, ocfg, dropout=(rate, seed, step))
    assert (got.cpu().double() - ref).abs().max() < ATOL_ACT, step
  plain = O.model_forward(x.double(), pd, ocfg)
  assert (ref - plain).abs().max() > 1e-4                      # the mask does something (probabilities are ~4e-3)
  assert (model(x.to(dev())).cpu().double() - plain).abs().max() < ATOL_ACT       # training=False: no dropout
  # a training step after two stochastic forwards uses mask number 3
  loss_ref, _, _, _ = O.loss_and_grads(torch.cat([x, x[:, :1]], 1).double(), pd, ocfg, dropout=(rate, seed, 3))
  loss, _, _ = model.loss_and_grads(torch.cat([x, x[:, :1]], 1).to(dev()))
  assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))


@pytest.mark.parametrize('name', ['cat_small_fused', 'cat_r64', 'cat_r128'])
def test_forward_range_guard_falls_back_to_exact_fp32(name):
  """The split-precision kernels cast activations to fp16 hi|lo unscaled: a residual stream beyond 65504 would become
  inf -> NaN.  The forward pass publishes the running max-abs of every tensor that feeds such a kernel; a tripped pass is
  repeated with the exact-fp32 kernels (WaveNet.call / test_step / train_step) and the optimizer update of the tripped
  attempt is skipped on the device.  Here the input conv's bias puts the residual stream at +-1e5."""
  from wavenets_amd import Adam, _lib
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=6, **kw)
  big = torch.where(torch.arange(params[1].numel()) % 2 == 0, 1.0e5, -1.0e5).to(params[1].dtype)
  params[1] = params[1] + big                                     # causal/bias
  model.set_weights([p.numpy() for p in params])
  B, T = 2, 120
  x, _ = _inputs(kw, B, T + 1, seed=3)
  pd = [p.double() for p in params]
  ref, inter = O.model_forward(x[:, :-1].double(), pd, ocfg, return_intermediates=True)
  assert inter['h'][1].abs().max() > 9e4                           # the stream really is out of fp16 range
  # the raw split-precision pass reports the overflow instead of hiding it
  loss, _, _ = model.loss_and_grads(x.to(dev()))
  assert loss[2].item() == 1.0
  # guarded entry points: finite and equal to the oracle at fp32 resolution of a 1e5 stream (ulp 8e-3)
  lg = model.logits(x[:, :-1].to(dev())).cpu().double()
  assert torch.isfinite(lg).all() and (lg - inter['logits']).abs().max() < 5e-2
  out = model(x[:, :-1].to(dev())).cpu().double()
  assert (out - ref).abs().max() < 5e-3
  model.compile(optimizer=None)
  ev = model.test_step(x.to(dev()))
  loss_ref, _, _, _ = O.loss_and_grads(x.double(), pd, ocfg)
  assert abs(ev['loss'] - loss_ref.item()) < 2e-3 * abs(loss_ref.item())
  # training: the tripped attempt must not touch the parameters or the moments; the repeated step is the real one
  model.compile(optimizer=Adam(learning_rate=5e-4, clipnorm=1.0))
  before = model.flat_params.data.clone()
  logs = model.train_step(x.to(dev()))
  assert model.optimizer.iterations == 1 and abs(logs['loss'] - loss_ref.item()) < 2e-3 * abs(loss_ref.item())
  after = model.flat_params.data
  assert torch.isfinite(after).all() and torch.isfinite(model.optimizer.m).all() and torch.isfinite(model.optimizer.v).all()
  step = (after - before).abs()
  assert step.max() <= 5e-4 * 1.001 + 1e-2 * 8e-3 and step.max() > 1e-4      # one Adam step of size lr (fp32 at 1e5: ulp 8e-3)
  assert _lib.lib().wn_debug_value(1) == 0                       # the exact-fp32 switch is restored


def test_forward_range_guard_quiet_for_tiny_activations():
  """A residual stream of 1e-6 loses nothing that matters (absolute error of the fp16 lo part <= 3e-8): no fallback."""
  kw = dict(MODEL_CASES['cat_r64'])
  ocfg, params, model = make_pair(seed=6, **kw)
  for i in (0, 1):
    params[i] = params[i] * 1e-6                                   # causal kernel and bias
  model.set_weights([p.numpy() for p in params])
  x, _ = _inputs(kw, 2, 121, seed=3)
  ref, inter = O.model_forward(x[:, :-1].double(), [p.double() for p in params], ocfg, return_intermediates=True)
  assert inter['h'][0].abs().max() < 1e-5
  loss, _, _ = model.loss_and_grads(x.to(dev()))
  assert loss[2].item() == 0.0
  assert (model.logits(x[:, :-1].to(dev())).cpu().double() - inter['logits']).abs().max() < ATOL_ACT


def test_layer_dropout_training_mode():
  from wavenets_amd import WaveNetLayer
  layer = WaveNetLayer(channels=32, skip_channels=32, dilation_rate=2, dropout=0.5, device=dev())
  x = torch.randn(2, 64, 32, device=dev())
  torch.manual_seed(0)
  xo, sk = layer(x, training=True)
  xe, se = layer(x, training=False)
  assert xo.shape == xe.shape and not torch.allclose(xo, xe)
  # the residual is not dropped: with zero conv weights the block is the identity in both modes
  with torch.no_grad():
    layer.flat_params.zero_()
  xo, _ = layer(x, training=True)
  assert torch.allclose(xo, x, atol=1e-6)


def test_sound_callback_writes_ab_files(tmp_path):
  """src/callbacks.py:51-118 with use_fast='both': queued and sliding-window audio side by side."""
  import numpy as np
  from wavenets_amd import WaveNet
  from wavenets_amd.callbacks import SoundCallback
  m = WaveNet(blocks=3, channels=32, skip_channels=32, dilation_bound=4, final_layers_channels=[], bits=8, device=dev())
  cb = SoundCallback(str(tmp_path), 16000, 300, True, epoch_frequency=2, use_fast='both', model=m)
  assert cb.on_epoch_end(0) is None                     # logged every 2nd epoch only
  cb.on_epoch_end(1)
  d = tmp_path / 'epoch_0001'
  import wave
  from wavenets_amd.callbacks import create_spectrogram
  for key in ('fast', 'standard'):
    wav = np.load(d / f'generated_{key}.npy')
    assert wav.shape == (5, 300, 1) and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
    # apply_mulaw=True: what is logged went through inverse_mu_law (src/callbacks.py:70-72,126-131): every value is the
    # expansion of a left bin edge i / 128 - 1 of the 8-bit quantiser (src/model.py:411)
    comp = np.sign(wav) * np.log1p(255.0 * np.abs(wav)) / np.log(256.0)
    q = (comp + 1.0) * 128.0
    assert np.abs(q - np.round(q)).max() < 1e-3 and q.min() >= -1e-3 and q.max() <= 255.001
    spec = np.load(d / f'generated_spectrogram_{key}.npy')
    assert spec.shape == (5, 129, 1, 1) and np.array_equal(spec, create_spectrogram(wav, 16000))
    assert spec.min() == 0.0 and spec.max() == 1.0                       # min-max scaled over the batch (src/callbacks.py:151-157)
    with wave.open(str(d / f'generated_{key}_4.wav'), 'rb') as f:
      assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 16000, 300)
      pcm = np.frombuffer(f.readframes(300), dtype='<i2').astype(np.float32) / 32768.0
    assert np.abs(pcm - wav[4, :, 0]).max() <= 1.0 / 32768.0 + 1e-7
  # a primed run (initial_sample given, src/callbacks.py:83-101): deterministic continuation is the same queued or not
  x = O.synthetic_waveform(8, m.receptive_field, seed=2).to(dev())
  a = m.generate(20, sample=x, use_queues=True, deterministic=True)
  b = m.generate(20, sample=x, use_queues=False, deterministic=True)
  assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------
# edge cases of the tiled kernels: utterances shorter than the dilation (every shifted tap is causal
# padding), lengths that are not multiples of the 16 / 32 step tiles, a single utterance, and
# generation batches that do not fill a 32-utterance tile
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('B,T', [(1, 7), (2, 40), (3, 97)])
def test_gradients_short_utterances_large_dilations(B, T, math_mode):
  kw = dict(blocks=9, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', bits=8)             # dilations 1 .. 256: most of them exceed T
  ocfg, params, model = make_pair(seed=11, **kw)
  x, _ = _inputs(kw, B, T + 1, seed=12)
  loss_ref, _, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg)
  loss, _, _ = model.loss_and_grads(x.to(dev()))
  assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))
  for n, g, r in zip(model.variable_names, model.gradients(), grads_ref):
    scale = max(r.abs().max().item(), 1e-6)
    assert (g.cpu().double() - r).abs().max().item() < 1e-4 * scale + 1e-7, n


@pytest.mark.parametrize('B', [33, 70])
def test_queued_generation_ragged_utterance_tiles(B):
  kw = dict(MODEL_CASES['cat_r64'])
  ocfg, params, model = make_pair(seed=7, bias_range=0.3, **kw)
  w = O.synthetic_waveform(B, model.receptive_field, seed=6).to(dev())
  naive = model.generate(6, sample=w, use_queues=False, deterministic=True)
  queued = model.generate(6, sample=w, use_queues=True, deterministic=True)
  assert torch.equal(naive, queued)
  # rows are independent: the first utterances do not depend on how many follow
  few = model.generate(6, sample=w[:5], use_queues=True, deterministic=True)
  assert torch.equal(few, queued[:5])


@pytest.mark.parametrize('skip_channels', [32, 96, 160])
def test_queued_generation_skip_wave_counts(skip_channels):
  """1, 3 and 5 skip waves in the fused generation chain (they also issue the chain's LDS-DMA)."""
  kw = dict(blocks=5, channels=32, skip_channels=skip_channels, dilation_bound=8, final_layers_channels=[64],
            activation='leaky_relu', num_mixtures=4, sampling_function='logistic', bits=16)
  ocfg, params, model = make_pair(seed=9, bias_range=0.3, **kw)
  w = O.synthetic_waveform(4, model.receptive_field, seed=3).to(dev())
  naive = model.generate(12, sample=w, use_queues=False, deterministic=True)
  queued = model.generate(12, sample=w, use_queues=True, deterministic=True)
  assert torch.equal(naive, queued)          # continuous (mixture) outputs: equality is bit for bit


@pytest.mark.parametrize('name', ['cat_small_fused', 'cond', 'mol'])
def test_train_step_in_two_calls_equals_one_call(name):
  """wn_plan_set_train_phases: forward + loss (+ step sample, L2 loss term, range flag) as one call and the backward half as
  a second call -- with foreign work queued in between -- give the loss, the sample and every gradient of the single call
  bit for bit; and train_step, which reads its scalars back in that gap, leaves the same parameters as the late read-back."""
  from wavenets_amd import Adam, MeanSquaredError
  kw = dict(MODEL_CASES[name])
  kw['l2_reg_factor'] = 1e-3
  B, T = 3, 401
  res = []
  for early in (False, True):
    ocfg, params, model = make_pair(seed=5, bias_range=0.3, **kw)
    x = O.synthetic_waveform(B, T + 1, seed=31).to(dev())
    data = x
    if kw.get('conditioning') is not None:
      data = (x, torch.rand(B, kw['cond_inputs'], generator=torch.Generator().manual_seed(2)).to(dev()))
    seen = {}

    def between(loss, sample, y_true):
      seen['loss'] = loss.clone()                       # complete here: loss, reg_loss, range flag
      torch.zeros(1 << 20, device=dev()).add_(1.0)      # somebody else's kernels in the gap
    model._sample_calls = 3
    loss, samp, _ = model.loss_and_grads(data, want_sample=True, _between=between if early else None)
    if early:
      assert torch.equal(seen['loss'], loss)
    grads = model.flat_grads.clone()
    model.compile(optimizer=Adam(learning_rate=1e-3, clipnorm=1.0), metrics=[MeanSquaredError()])
    model.early_logs = early
    logs = [model.train_step(data) for _ in range(3)]
    res.append((loss.clone(), samp.clone(), grads, model.flat_params.clone(), logs))
  for a_, b_ in zip(res[0][:4], res[1][:4]):
    assert torch.equal(a_, b_)
  for la, lb in zip(res[0][4], res[1][4]):
    assert la.keys() == lb.keys()
    for k_ in la:
      assert abs(la[k_] - lb[k_]) <= 1e-6 * max(1.0, abs(la[k_])), (k_, la[k_], lb[k_])


@pytest.mark.parametrize('name,det', [('cat_r64', True), ('cat_r64', False), ('cat_small_fused', False), ('mol', False)])
def test_queued_generation_draws_the_sliding_windows_samples(name, det):
  """The queued sampler (chain launch + head launch with the sampling tail, or sampler + emit in one launch for mixture
  heads) draws the samples of the sliding window, stochastic draws included (same Philox counters), at 5 utterances."""
  kw = dict(MODEL_CASES[name])
  ocfg, params, model = make_pair(seed=11, bias_range=0.3, **kw)
  B, n = 5, 14
  w = O.synthetic_waveform(B, model.receptive_field, seed=12).to(dev())
  naive = model.generate(n, sample=w, use_queues=False, deterministic=det)
  queued = model.generate(n, sample=w, use_queues=True, deterministic=det)
  assert torch.equal(naive, queued), (naive - queued).abs().max()


@pytest.mark.parametrize('skip_channels,finals', [(160, [48]), (256, [256, 64]), (96, [32]), (192, [])])
def test_queued_generation_unfolded_skip_contraction(skip_channels, finals):
  """Heads the fold does not apply to (first head conv narrower than 64 or not a multiple of 32 columns, as wide as the
  skip sum, or no hidden head layer at all): the chain kernel carries the reference's full-width skip contraction -- 5, 8,
  3 and 6 column tiles, i.e. two to four skip waves of two tiles each on the two-phase schedule (the folded default needs at
  most four tiles and runs both in phase C)."""
  kw = dict(blocks=5, channels=64 if skip_channels == 256 else 32, skip_channels=skip_channels, dilation_bound=8,
            final_layers_channels=finals, activation='leaky_relu', bits=8)
  ocfg, params, model = make_pair(seed=13, bias_range=0.3, **kw)
  assert 'folded' not in model.kernel_report()
  w = O.synthetic_waveform(5, model.receptive_field, seed=14).to(dev())
  naive = model.generate(12, sample=w, use_queues=False, deterministic=True)
  queued = model.generate(12, sample=w, use_queues=True, deterministic=True)
  queued_s = model.generate(12, sample=w, use_queues=True, deterministic=False)
  naive_s = model.generate(12, sample=w, use_queues=False, deterministic=False)
  assert torch.equal(naive, queued)
  assert torch.equal(naive_s, queued_s)


def test_queued_generation_ring_wraparound():
  """More steps than the deepest ring has slots (dilation 32 -> 33 slots): every ring wraps at least twice."""
  kw = dict(blocks=6, channels=32, skip_channels=64, dilation_bound=64, final_layers_channels=[32],
            activation='leaky_relu', bits=8)
  ocfg, params, model = make_pair(seed=5, bias_range=0.3, **kw)
  w = O.synthetic_waveform(2, model.receptive_field, seed=8).to(dev())
  naive = model.generate(80, sample=w, use_queues=False, deterministic=True)
  queued = model.generate(80, sample=w, use_queues=True, deterministic=True)
  assert torch.equal(naive, queued)


@pytest.mark.parametrize('form', ['relay', 'one_workgroup'])
@pytest.mark.parametrize('B,cond,finals,skip', [(3, False, [128, 64], True), (40, False, [128, 64], True), (5, True, [128, 64], True),
                                                (33, False, [64, 64], True), (4, False, [128, 64], False)])
def test_queued_generation_128_channel_chain(B, cond, finals, skip, form):
  """128-channel blocks: every block of a step, the input conv and the folded skip contraction in one launch -- as a relay
  over one workgroup per block whose rows travel as tagged granules (wn_gen_relay128_kernel, the default) and inside one
  workgroup per utterance tile (wn_gen_chain128_kernel, knob 2).  One and two utterance tiles, rings that wrap, with and
  without global conditioning, a skip sum the chain cannot fold (64 columns) and no skip connections at all: each form
  draws the sliding window's samples."""
  from wavenets_amd import _lib
  kw = dict(blocks=5, channels=128, skip_channels=256, dilation_bound=16, final_layers_channels=finals,
            activation='leaky_relu', bits=8, use_skip=skip)
  if cond:
    kw.update(conditioning='global', mapping_layers=[8, 16], mapping_activation='leaky_relu', cond_inputs=7)
  ocfg, params, model = make_pair(seed=17, bias_range=0.3, **kw)
  w = O.synthetic_waveform(B, model.receptive_field, seed=3).to(dev())
  c = torch.rand(B, 7, generator=torch.Generator().manual_seed(4)).to(dev()) if cond else None
  args = dict(sample=w, deterministic=False)
  if cond:
    args['condition'] = c
  naive = model.generate(40, use_queues=False, **args)
  _lib.lib().wn_debug_set(2, 1 if form == 'one_workgroup' else 0)
  try:
    queued = model.generate(40, use_queues=True, **args)
    again = model.generate(40, use_queues=True, **args)      # the granule tags of the first call are still in the workspace
  finally:
    _lib.lib().wn_debug_set(2, 0)
  assert torch.equal(naive, queued), (naive - queued).abs().max()
  assert torch.equal(queued, again)


@pytest.mark.parametrize('det', [True, False])
@pytest.mark.parametrize('channels,sampler,mix,finals', [(64, 'logistic', 10, [128, 256]), (32, 'gaussian', 8, [64, 64]),
                                                         (128, 'logistic', 5, [128, 128])])
def test_queued_generation_mixture_head_in_one_launch(channels, sampler, mix, finals, det):
  """Mixture heads in queued generation: the split-precision hidden layers, the exact-fp32 output layer (3 x mixtures
  columns, the rows of wn_gemm_rows_kernel<1>) and the mixture sampler in ONE launch (wn_gen_head_kernel with f32_K > 0,
  tail 3 / 4) behind the 64-channel chain, the 32-channel chain and the 128-channel relay: the sliding window's samples,
  bit for bit, for arg-max and for Philox draws (src/model.py:423-503)."""
  kw = dict(blocks=6, channels=channels, skip_channels=256, dilation_bound=32, final_layers_channels=finals,
            activation='leaky_relu', num_mixtures=mix, sampling_function=sampler, bits=16)
  ocfg, params, model = make_pair(seed=23, bias_range=0.3, **kw)
  w = O.synthetic_waveform(5, model.receptive_field, seed=6).to(dev())
  naive = model.generate(30, sample=w, use_queues=False, deterministic=det)
  queued = model.generate(30, sample=w, use_queues=True, deterministic=det)
  assert torch.equal(naive, queued), (naive - queued).abs().max()


def test_generation_relay_watchdog_reports_a_missing_hand_over():
  """Every wait of the relay is bounded: when a block never hands its rows on (fault injection, knob 3), its successor
  gives up after 2^21 polls (a fraction of a second), records itself in the watchdog word behind the guard slot, the launch ENDS, and
  WaveNet.generate raises instead of returning the samples; the next call on the same workspace is good again."""
  from wavenets_amd import _lib
  kw = dict(blocks=6, channels=128, skip_channels=256, dilation_bound=16, final_layers_channels=[128, 64],
            activation='leaky_relu', bits=8)
  ocfg, params, model = make_pair(seed=17, bias_range=0.3, **kw)
  w = O.synthetic_waveform(3, model.receptive_field, seed=3).to(dev())
  good = model.generate(6, sample=w, use_queues=True, deterministic=True)
  _lib.lib().wn_debug_set(3, 3)                     # block 2 withholds its rows
  try:
    with pytest.raises(RuntimeError, match='gave up waiting'):
      model.generate(3, sample=w, use_queues=True, deterministic=True)
  finally:
    _lib.lib().wn_debug_set(3, 0)
  again = model.generate(6, sample=w, use_queues=True, deterministic=True)
  assert torch.equal(good, again)


def test_generation_relay_under_load_and_long_runs():
  """The relay's hand-offs with the GPU busy on another stream (a large copy kernel loop competing for the CUs and the
  fabric) and over many steps: 600 samples at 30 blocks of 128 channels, bit-identical to the one-workgroup form."""
  from wavenets_amd import WaveNet, _lib
  kw = dict(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16)
  model = WaveNet(**kw, device=dev(), seed=2)
  w = O.synthetic_waveform(8, model.receptive_field, seed=5).to(dev())
  _lib.lib().wn_debug_set(2, 1)
  try:
    ref = model.generate(600, sample=w, use_queues=True, deterministic=False)
  finally:
    _lib.lib().wn_debug_set(2, 0)
  side = torch.cuda.Stream()
  big = torch.empty(1 << 28, device=dev())
  with torch.cuda.stream(side):
    for _ in range(200):
      big.mul_(1.0001)
  got = model.generate(600, sample=w, use_queues=True, deterministic=False)
  torch.cuda.synchronize()
  assert torch.equal(ref, got)


def test_plan_caches_survive_shape_changes():
  """One model object through training and generation at changing batch sizes / lengths: the plan's
  cached device tables (weight-gradient jobs, generation block table) are keyed by shape."""
  from wavenets_amd import Adam
  kw = dict(MODEL_CASES['cat_r64'])
  ocfg, params, model = make_pair(seed=3, bias_range=0.2, **kw)
  model.compile(optimizer=Adam(learning_rate=1e-3, clipnorm=1.0))
  losses = []
  for B, T in ((2, 200), (3, 97), (2, 200)):
    x = O.synthetic_waveform(B, T + 1, seed=B * 100 + T).to(dev())
    losses.append(model.train_step(x)['loss'])
    for gb in (3, 5):
      w = O.synthetic_waveform(gb, model.receptive_field, seed=gb).to(dev())
      a = model.generate(5, sample=w, use_queues=True, deterministic=True)
      b = model.generate(5, sample=w, use_queues=False, deterministic=True)
      assert torch.equal(a, b)
  assert all(math.isfinite(v) for v in losses)


@pytest.mark.parametrize('B,T', [(1, 20), (3, 97)])
def test_wide_blocks_gradients_ragged(B, T, math_mode):
  """128-channel blocks (BASELINE configs[3] width): the forward is two split-precision contractions with the
  gate in the first one's epilogue (fp32 mode: the exact one-kernel forward).  Utterances shorter than the
  larger dilations and lengths off the 32-step tile."""
  kw = dict(blocks=7, channels=128, skip_channels=256, dilation_bound=128, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16)
  ocfg, params, model = make_pair(seed=13, **kw)
  x, _ = _inputs(kw, B, T + 1, seed=14)
  loss_ref, _, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg)
  loss, _, _ = model.loss_and_grads(x.to(dev()))
  assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item()))
  for n, g, r in zip(model.variable_names, model.gradients(), grads_ref):
    scale = max(r.abs().max().item(), 1e-6)
    assert (g.cpu().double() - r).abs().max().item() < 1e-4 * scale + 1e-7, n


def test_wide_blocks_forward_paths_agree():
  """The streamed split-precision forward of 128-channel blocks against the exact-fp32 kernels on the same weights:
  activations within the 1e-4 bar at a length with many tiles."""
  kw = dict(MODEL_CASES['cat_r128'])
  ocfg, params, model = make_pair(seed=21, **kw)
  x = O.synthetic_waveform(2, 4100, seed=22).to(dev())
  with model.exact_fp32():
    ref = model.logits(x).clone()
  out = model.logits(x)
  assert not torch.equal(out, ref)                        # two different kernels really ran
  assert (out - ref).abs().max().item() < ATOL_ACT


def test_wide_skip_contraction_matches_oracle():
  """The unfolded skip sum (K = blocks * D >= 512, N = 256: exact-fp32 mode aside, a head whose first conv is as wide as
  the skip sum) runs on the 256-column streamed kernel from 2048 row tiles on (enough to fill the chip with one column
  block) and on the 128-column kernel below that: both against the oracle.  T is off the 32-step tile and the tile count
  does not fill the last workgroup."""
  kw = dict(blocks=9, channels=64, skip_channels=256, dilation_bound=16, final_layers_channels=[256, 64],
            activation='leaky_relu', bits=8)
  ocfg, params, model = make_pair(seed=31, **kw)
  assert 'folded' not in model.kernel_report()
  x = O.synthetic_waveform(9, 7301, seed=32)          # 9 x 229 = 2061 row tiles, the last of each utterance holds 5 rows
  out = model.logits(x.to(dev())).clone()
  small = model.logits(x[:2].to(dev())).clone()        # 458 row tiles: the 128-column kernel
  _, inter = O.model_forward(x[:2].double(), [p.double() for p in params], ocfg, None, return_intermediates=True)
  assert (out[:2].cpu().double() - inter['logits']).abs().max() < ATOL_ACT      # (oracle on two of the utterances)
  assert (small.cpu().double() - inter['logits']).abs().max() < ATOL_ACT
  assert torch.equal(out[:2], small)                   # same per-element MFMA sequence in both kernels


@pytest.mark.parametrize('name', ['cat_r64', 'mol', 'gauss'])
def test_train_step_metric_sample_drawn_inside_the_step(name):
  """train.py:227 compiles MeanSquaredError: every step draws sample_waveform(pred) (src/model.py:338).  The
  library draws it from the logits inside the step; the result must be the draw sample_waveform makes from
  that step's pred (same Philox counter), so the metric is identical."""
  from wavenets_amd import Adam, MeanSquaredError
  kw = dict(MODEL_CASES[name])
  x = O.synthetic_waveform(3, 301, seed=41).to(dev())
  logs = []
  for fused in (True, False):
    ocfg, params, model = make_pair(seed=40, **kw)
    model.compile(optimizer=Adam(learning_rate=1e-3, clipnorm=1.0), metrics=[MeanSquaredError()])
    model._fused_step_sample = fused
    for _ in range(2):
      out = model.train_step(x)
    logs.append(out)
    # the sample itself, not only its mean square
    _, samp, y = model.loss_and_grads(x, want_sample=True)
    assert samp.shape == (3, 300, 1) and float(samp.abs().max()) <= 1.0
    logs[-1]['_samp'] = samp.clone()
  if name == 'cat_r64':
    # 256 classes: the in-step draw and the loss are the epilogue of the head's last conv (the logits are never written), the
    # comparison run goes through pred + sample_waveform.  Same probabilities up to the last bits of their running sums, same
    # Philox numbers: a drawn class may differ only where the uniform number falls within rounding of a CDF edge.
    assert abs(logs[0]['loss'] - logs[1]['loss']) <= 1e-6 * abs(logs[1]['loss'])
    a_, b_ = logs[0]['_samp'], logs[1]['_samp']
    diff = (a_ != b_)
    assert int(diff.sum()) <= max(1, a_.numel() // 200), int(diff.sum())
    assert float((a_ - b_).abs().max()) <= 2.0 / 256 + 1e-7                     # an adjacent class at most
    assert abs(logs[0]['mean_squared_error'] - logs[1]['mean_squared_error']) <= 1e-3 * logs[1]['mean_squared_error']
  else:
    assert logs[0]['mean_squared_error'] == logs[1]['mean_squared_error']
    assert logs[0]['loss'] == logs[1]['loss']
    assert torch.equal(logs[0]['_samp'], logs[1]['_samp'])


@pytest.mark.parametrize('bits', [9, 11])
def test_train_step_metric_sample_wide_categorical_heads(bits):
  """512 classes: the draw comes from the logits in its own kernel (the row does not fit the loss kernel's
  registers); 2048 classes: the library declines and the step falls back to pred + sample_waveform.  Either way
  the sample is the one sample_waveform draws from the step's pred."""
  from wavenets_amd import Adam, MeanSquaredError
  kw = dict(blocks=3, channels=32, skip_channels=32, dilation_bound=4, final_layers_channels=[32], activation='relu', bits=bits)
  x = O.synthetic_waveform(2, 130, seed=51).to(dev())
  res = []
  for fused in (True, False):
    ocfg, params, model = make_pair(seed=50, **kw)
    model.compile(optimizer=Adam(learning_rate=1e-3, clipnorm=1.0), metrics=[MeanSquaredError()])
    model._fused_step_sample = fused
    logs = model.train_step(x)
    _, samp, _ = model.loss_and_grads(x, want_sample=True)
    res.append((logs, samp.clone()))
  assert res[0][0] == res[1][0]
  assert torch.equal(res[0][1], res[1][1])
  # sample values are left bin edges i / 2^(bits-1) - 1
  q = (res[0][1] + 1.0) * float(1 << (bits - 1))
  assert torch.equal(q, q.round()) and float(q.max()) < (1 << bits)


def test_head_weight_gradients_on_their_own_time_split():
  """At longer utterances the head's weight gradients are accumulated over a finer time split than the blocks' (own
  compact slab): every gradient tensor against the oracle."""
  kw = dict(MODEL_CASES['cat_r64'], blocks=10, dilation_bound=32)
  ocfg, params, model = make_pair(seed=61, **kw)
  x = O.synthetic_waveform(8, 3001, seed=62)      # 8 x 3000, 68 jobs: the blocks split each utterance 10-fold, the head 12-fold
  _, _, grads_ref, _ = O.loss_and_grads(x.double(), [p.double() for p in params], ocfg)
  model.loss_and_grads(x.to(dev()))
  for n, a, r in zip(model.variable_names, model.gradients(), grads_ref):
    scale = max(r.abs().max().item(), 1e-6)
    assert (a.cpu().double() - r).abs().max().item() < 1e-4 * scale + 1e-7, n
