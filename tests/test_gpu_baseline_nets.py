"""The exact BASELINE.json networks at their full depth against the fp64 oracle.

configs[1] (30-layer mu-law-256, 64 residual / 256 skip channels), configs[3] (30-layer
mixture-of-logistics head with 10 mixtures, 128 residual channels, 16 bits) and configs[4] (configs[1] with
global conditioning: 110-way one-hot speaker id through the mapping net [8, 16, 32]) are built with their 30
blocks / dilations 1..512 (x3) and run on B = 2 utterances of T = 3500 samples (longer than the receptive
field, 3071): logits, loss and EVERY gradient tensor are compared with the oracle in both contraction modes
at north_star's bar -- 1e-4 absolute per activation, gradients 1e-4 relative to each tensor's scale.
SURVEY.md section 7 hard part 2 ("1e-4 through 30 residual layers + softmax") is what this file checks.
Reference maths: src/model.py:79-122,213-239, src/layers.py:178-224 (conditioning add: :203-204).
"""
import pytest
import torch

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu

ATOL_ACT = 1e-4

NETS = {
    # configs[0] at its FULL size (batch 1 x 16000): 10 blocks, dilations 1..512, 32 residual channels, skip_channels=None
    # (skip = the pre-residual conv1 output, src/layers.py:216-219), head [], no activation -- the
    # two-launch backward family and the S == 0 gradient path of the exact network, not of a 5-block stand-in
    'configs0_cat_r32': dict(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[], bits=8),
    'configs1_cat_r64': dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024,
                             final_layers_channels=[128, 256], activation='leaky_relu', bits=8),
    'configs3_mol10_r128': dict(blocks=30, channels=128, skip_channels=256, dilation_bound=1024,
                                final_layers_channels=[128, 256], activation='leaky_relu', num_mixtures=10,
                                sampling_function='logistic', bits=16),
    'configs4_globalcond_r64': dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024,
                                    final_layers_channels=[128, 256], activation='leaky_relu', bits=8,
                                    conditioning='global', mapping_layers=[8, 16, 32],
                                    mapping_activation='leaky_relu', cond_inputs=110),
}
SIZES = {'configs0_cat_r32': (1, 16000)}          # (B, T); default: 2 x 3500, longer than the receptive field 3071
GEOMETRY = {'configs0_cat_r32': (1025, 10, [(0, 1), (9, 512)])}      # receptive field, blocks, (conv index, dilation)


def _size(name):
  return SIZES.get(name, (2, 3500))


def dev():
  return torch.device('cuda', 0)


_ORACLE = {}


def _oracle(name):
  """Inputs and fp64 oracle forward of one network, computed once for both math modes."""
  if name in _ORACLE:
    return _ORACLE[name]
  kw = dict(NETS[name])
  cond_inputs = kw.pop('cond_inputs', 0)
  ocfg = O.OracleConfig(**kw, cond_inputs=cond_inputs)
  params = O.init_params(ocfg, seed=21, bias_range=0.1)
  B, T = _size(name)
  x = O.synthetic_waveform(B, T + 1, seed=31)
  cond = None
  if cond_inputs:
    ids = torch.randint(0, cond_inputs, (B,), generator=torch.Generator().manual_seed(5))
    cond = torch.nn.functional.one_hot(ids, cond_inputs).float()         # speaker id -> one-hot
  pd = [p.double() for p in params]
  cd = cond.double() if cond is not None else None
  torch.set_num_threads(16)
  _, inter = O.model_forward(x[:, :-1].double(), pd, ocfg, cd, return_intermediates=True)
  _ORACLE[name] = (ocfg, params, x, cond, inter)
  return _ORACLE[name]


def _oracle_grads(name, head_branch):
  """fp64 loss and gradients (parameters, skip sum, every block input).  ``head_branch``: the side of the
  leaky_relu kink the product took for each head activation -- used by the oracle only where its own
  pre-activation is within 1e-4 of the kink (O.activation_with_branch): a 30-block net at this size has a few
  such elements in every run, and there the derivative (1 vs 0.2) is undecided within the activation tolerance."""
  ocfg, params, x, cond, _ = _oracle(name)
  ps = [p.double().requires_grad_(True) for p in params]
  cd = cond.double() if cond is not None else None
  pred, inter = O.model_forward(x[:, :-1].double(), ps, ocfg, cd, return_intermediates=True, head_branch=head_branch)
  target = O.prepare_target(x[:, 1:].double(), ocfg)
  B = x.shape[0]
  loss = O.loss_fn(target, pred, ocfg).sum() / B
  want = [inter['skip_sum']] + inter['h'][:-1]            # nothing flows into the last block output (skip head)
  gr = torch.autograd.grad(loss, ps + want, allow_unused=True)
  grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(gr[:len(ps)], ps)]
  return loss.detach(), grads, gr[len(ps)], gr[len(ps) + 1:], inter['kink_overrides']


@pytest.fixture(params=['split', 'fp32'])
def math_mode(request):
  from wavenets_amd import _lib
  _lib.lib().wn_debug_set(1, 1 if request.param == 'fp32' else 0)
  yield request.param
  _lib.lib().wn_debug_set(1, 0)


def _model(name, params):
  from wavenets_amd import WaveNet
  kw = dict(NETS[name])
  cond_inputs = kw.pop('cond_inputs', 0)
  model = WaveNet(**kw, device=dev())
  if cond_inputs:
    model.build([(1, 8, 1), (1, cond_inputs)])
  model.set_weights([p.numpy() for p in params])
  return model


@pytest.mark.parametrize('name', list(NETS))
def test_30_block_activations_loss_and_all_gradients(name, math_mode):
  from wavenets_amd import _lib
  ocfg, params, x, cond, inter = _oracle(name)
  model = _model(name, params)
  B, T = _size(name)
  rf, nblk, dils = GEOMETRY.get(name, (3071, 30, [(0, 1), (9, 512), (10, 1), (29, 512)]))
  assert model.receptive_field == rf and len(model.wavenet_blocks) == nblk
  assert [_lib.lib().wn_plan_dilation(model._plan, b) for b, _ in dils] == [d for _, d in dils]
  inp = (x[:, :-1].to(dev()), cond.to(dev())) if cond is not None else x[:, :-1].to(dev())
  lg = model.logits(inp).cpu().double()
  err = (lg - inter['logits']).abs().max().item()
  assert err < ATOL_ACT, (name, math_mode, 'logits', err)
  data = (x.to(dev()), cond.to(dev())) if cond is not None else x.to(dev())
  loss, _, _ = model.loss_and_grads(data)
  torch.cuda.synchronize()

  def ws(what, idx, shape):
    return model.training_intermediate(what, idx, B, T).cpu().double().reshape(shape)

  # ---- every activation the training pass keeps: block inputs H[0..30], skip sum, head activations ----
  worst_act = 0.0
  for b, h in enumerate(inter['h']):
    worst_act = max(worst_act, (ws(0, b, h.shape) - h).abs().max().item())
    assert worst_act < ATOL_ACT, (name, math_mode, f'H[{b}]', worst_act)
  try:
    assert (ws(3, 0, inter['skip_sum'].shape) - inter['skip_sum']).abs().max().item() < ATOL_ACT
    folded = False
  except ValueError:
    # split-precision training passes fold the skip path into the head's first conv (a = sum_b (W_s(b) W_f0)^T z_b + b'):
    # the skip sum is never formed; its consumer, the first head activation, is compared below
    folded = True
  branch = []
  for i, a in enumerate(inter['head_pre']):
    ha = ws(4, i, a.shape)
    assert (ha - O.activation(a, ocfg.activation)).abs().max().item() < ATOL_ACT, (name, f'head activation {i}')
    branch.append(ha >= 0)                               # leaky_relu: the product's side of the kink
  # ---- loss, data gradients at the skip sum and at every block input, every parameter gradient ----
  loss_ref, grads_ref, g_skip_ref, g_h_ref, n_kink = _oracle_grads(name, branch)
  # The oracle may take the product's side of a leaky_relu kink only where ITS OWN fp64 pre-activation lies within
  # KINK_TOL (= the activation tolerance) of 0; that must stay a handful of elements (0-2 of 2.7 M in practice), or
  # a sign bug near zero could hide behind it.  Re-derived here from the oracle's pre-activations, not trusted.
  n_head = sum(a.numel() for a in inter['head_pre'])
  n_over, worst_over = 0, 0.0
  for a, br in zip(inter['head_pre'], branch):
    over = (a >= 0) != br
    n_over += int(over.sum())
    if over.any():
      worst_over = max(worst_over, a[over].abs().max().item())
  assert n_over == n_kink, (n_over, n_kink)                 # every disagreement is one the oracle resolved as a kink ...
  assert worst_over < O.KINK_TOL, worst_over                # ... i.e. none lies outside the undecided band
  assert n_kink <= 8 and n_kink <= 1e-5 * n_head, (name, math_mode, n_kink, n_head)
  assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item())), (loss[0].item(), loss_ref.item())
  if folded:
    # d loss / d skip sum = (d loss / d a) W_f0^T, a = pre-activation of the first head conv (kept in the workspace)
    names = model.variable_names
    w_f0 = params[names.index('final0/kernel')].double()[0]                       # (S, F0)
    g_skip = ws(6, 0, inter['head_pre'][0].shape) @ w_f0.T
  else:
    g_skip = ws(7, 0, g_skip_ref.shape)
  e = (g_skip - g_skip_ref).abs().max().item()
  assert e < 1e-4 * g_skip_ref.abs().max().item(), (name, math_mode, 'd loss / d skip sum', e)
  assert folded == (math_mode == 'split' and len(inter['head_pre']) > 0)      # (a head without hidden layers cannot fold)
  for b, r in enumerate(g_h_ref):
    e = (ws(9, b, r.shape) - r).abs().max().item()
    assert e < 1e-4 * r.abs().max().item(), (name, math_mode, f'd loss / d H[{b}]', e)
  worst = ('', 0.0)
  for n, g, r in zip(model.variable_names, model.gradients(), grads_ref):
    scale = max(r.abs().max().item(), 1e-6)
    e = (g.cpu().double() - r).abs().max().item()
    if e / scale > worst[1]:
      worst = (n, e / scale)
    assert e < 1e-4 * scale + 1e-7, (name, math_mode, n, e, scale)
  print(f'{name} [{math_mode}]: logits max|err| {err:.2e}, block inputs max|err| {worst_act:.2e}, worst gradient '
        f'{worst[0]} rel {worst[1]:.2e}, {n_kink} leaky_relu kink decisions taken from the product')


@pytest.mark.parametrize('name', ['configs1_cat_r64', 'configs3_mol10_r128'])
def test_30_block_probabilities_and_samples(name):
  """Model output as call() returns it (softmax probabilities / mixture parameters) and the deterministic
  sample drawn from it (src/model.py:415-418,487-498), 30 blocks deep."""
  ocfg, params, x, cond, _ = _oracle(name)
  model = _model(name, params)
  out = model(x[:, :-1].to(dev()))
  ref = O.model_forward(x[:, :-1].double(), [p.double() for p in params], ocfg)
  assert (out.cpu().double() - ref).abs().max().item() < ATOL_ACT
  got = model.sample_waveform(out, deterministic=True).cpu()
  want = O.sample_waveform_deterministic(ref.float(), ocfg)
  if ocfg.sampling_function == 'categorical':
    top2 = torch.topk(ref, 2, dim=-1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 2e-4).unsqueeze(-1)
    assert torch.equal(got[clear], want[clear])
  else:
    w = torch.softmax(ref[..., :ocfg.num_mixtures], -1)
    top2 = torch.topk(w, 2, dim=-1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-3).unsqueeze(-1)
    assert (got[clear] - want[clear]).abs().max().item() < ATOL_ACT


@pytest.mark.parametrize('name', ['configs1_cat_r64', 'configs3_mol10_r128'])
@pytest.mark.parametrize('queued', [False, True])
def test_30_block_generation_matches_oracle(name, queued):
  """generate (src/model.py:258-307, intended semantics) on the exact BASELINE networks, receptive field 3071, B = 2,
  8 samples, deterministic draws, for both samplers -- the sliding window and the queued one (README.md:16) -- against
  the oracle's sliding window in fp64.  Categorical: the emitted left bin edges must agree exactly wherever the
  oracle's own top-2 probabilities are further apart than the activation tolerance; a step that is not clear ends the
  comparison (later inputs may legitimately differ).  Mixture: clipped means within 1e-4."""
  ocfg, params, x, cond, _ = _oracle(name)
  model = _model(name, params)
  rf = model.receptive_field
  assert rf == 3071
  n = 8
  w = O.synthetic_waveform(2, rf, seed=77)
  pd = [p.double() for p in params]
  # the oracle's sliding window, keeping every step's prediction to judge how clear each decision was
  xw = w.double().clone()
  ref, clear = [], []
  with torch.no_grad():
    for _ in range(n):
      pred = O.model_forward(xw, pd, ocfg)[:, -1:, :]
      smp = O.sample_waveform_deterministic(pred.float(), ocfg).double()
      if ocfg.sampling_function == 'categorical':
        top2 = torch.topk(pred, 2, dim=-1).values
        clear.append(bool(((top2[..., 0] - top2[..., 1]) > 2e-4).all()))
      else:
        wts = torch.softmax(pred[..., :ocfg.num_mixtures], -1)
        top2 = torch.topk(wts, 2, dim=-1).values
        clear.append(bool(((top2[..., 0] - top2[..., 1]) > 1e-3).all()))
      ref.append(smp)
      xw = torch.cat([xw[:, 1:], smp], dim=1)
  ref = torch.cat(ref, dim=1)
  out = model.generate(n, sample=w.to(dev()), deterministic=True, use_queues=queued).cpu().double()
  assert out.shape == (2, n, 1)
  assert getattr(model, 'generation_guard_trips', 0) == 0
  upto = n if all(clear) else clear.index(False)
  assert upto >= 1, 'not even the first decision of the oracle is clear: change the seed'
  if ocfg.sampling_function == 'categorical':
    assert torch.equal(out[:, :upto], ref[:, :upto]), (name, queued, upto)
  else:
    assert (out[:, :upto] - ref[:, :upto]).abs().max().item() < ATOL_ACT, (name, queued, upto)


# the reference's own default configuration (train.py:22-50): 5 WaveNetLayers of 5 stacked dilated convs each (only the
# last one gated), 32 channels, gaussian mixture with 8 components on 16-bit audio, global conditioning through the
# mapping net [8, 16, 32], dropout 0.1, dilation_bound 256, leaky_relu, head [128, 256], no separate skip width
REFERENCE_DEFAULT = dict(blocks=5, layers_per_block=5, channels=32, dilation_bound=256, num_mixtures=8,
                         sampling_function='gaussian', bits=16, conditioning='global', mapping_layers=[8, 16, 32],
                         mapping_activation='leaky_relu', activation='leaky_relu', final_layers_channels=[128, 256],
                         kernel_size=2)


def test_reference_default_network_with_dropout(math_mode):
  """train.py:22-50 as shipped: receptive field 768, 25 dilated convs, dropout ON (the mask is the product's stateless
  hash of (seed, block, step, element), restated by the oracle -- TF's stream cannot be matched), one-hot(gender, 2)
  condition (src/utils.py:46-49).  Forward (inference), loss and every parameter gradient of a training pass vs fp64."""
  from wavenets_amd import WaveNet
  kw = dict(REFERENCE_DEFAULT)
  rate, seed = 0.1, 123
  ocfg = O.OracleConfig(**kw, cond_inputs=2)
  assert O.receptive_field(ocfg) == 768
  params = O.init_params(ocfg, seed=9, bias_range=0.1)
  model = WaveNet(**kw, dropout=rate, device=dev(), seed=seed)
  model.build([(1, 8, 1), (1, 2)])
  assert model.receptive_field == 768 and len(model.wavenet_blocks) == 5
  model.set_weights([p.numpy() for p in params])
  Bq, Tq = 3, 1200
  x = O.synthetic_waveform(Bq, Tq + 1, seed=41)
  cond = torch.nn.functional.one_hot(torch.tensor([0, 1, 1]), 2).float()
  pd = [p.double() for p in params]
  ref = O.model_forward(x[:, :-1].double(), pd, ocfg, cond.double())
  out = model((x[:, :-1].to(dev()), cond.to(dev()))).cpu().double()
  assert (out - ref).abs().max().item() < ATOL_ACT
  n_head = n_inner = 0
  for step in (1, 2):
    loss, _, _ = model.loss_and_grads((x.to(dev()), cond.to(dev())))
    torch.cuda.synchronize()
    # leaky_relu's derivative jumps at 0: where the ORACLE's own fp64 pre-activation lies within KINK_TOL of 0 it takes the
    # side the product took (O.activation_with_branch) -- for the two head activations and the 20 non-gated dilated convs
    # (src/layers.py:66-74, src/model.py:105-111).  The overrides are counted and bounded below.
    head_branch = [model.training_intermediate(4, i, Bq, Tq).cpu().reshape(Bq, Tq, -1) >= 0 for i in range(2)]
    inner_branch = [[model.training_intermediate(11, b * 4 + i, Bq, Tq).cpu().reshape(Bq, Tq, -1) >= 0 for i in range(4)]
                    for b in range(5)]
    klog = []
    loss_ref, _, grads_ref, _ = O.loss_and_grads(x.double(), pd, ocfg, cond.double(), dropout=(rate, seed, step),
                                                 head_branch=head_branch, inner_branch=inner_branch, kink_log=klog)
    assert len(klog) == 20
    for cnt, worst, nov in klog:
      assert cnt == nov and worst < O.KINK_TOL, (cnt, nov, worst)      # every disagreement lies inside the undecided band
    n_inner += sum(c for c, _, _ in klog)
    with torch.no_grad():
      _, inter = O.model_forward(x[:, :-1].double(), pd, ocfg, cond.double(), return_intermediates=True,
                                 dropout=(rate, seed, step), head_branch=head_branch, inner_branch=inner_branch)
    n_head += inter['kink_overrides']
    assert abs(loss[0].item() - loss_ref.item()) < 2e-5 * max(1.0, abs(loss_ref.item())), (step, loss[0].item(), loss_ref.item())
    for n, g, r in zip(model.variable_names, model.gradients(), grads_ref):
      scale = max(r.abs().max().item(), 1e-6)
      e = (g.cpu().double() - r).abs().max().item()
      assert e < 1e-4 * scale + 1e-7, (n, step, e, scale)
  assert n_inner <= 8 and n_head <= 8, (n_inner, n_head)
