"""CPU oracle for the WaveNet hot path of jirsat/wavenets  --  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* (written from scratch, PyTorch-CPU, fp32 or fp64) of the
arithmetic that the TF2/Keras reference performs on its training / generation path.  It is
the checker the HIP kernels are compared with; it is never imported by the product
package ``wavenets_amd`` (only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it).

PARITY UNPINNED: TensorFlow/Keras are not installed in the build image and the reference
ships no tests, fixtures or golden vectors, so this restatement cannot
be executed against the reference itself.  It is pinned instead by analytic known-answer
tests (tests/test_oracle_known_answers.py): impulse responses of the causal dilated
convolution, the receptive-field edge, the quantiser table, the mu-law round trip, closed
form losses, fp64 finite-difference gradient checks and hand-computed Adam steps.

Every function cites the reference file:line (relative to /root/reference) it follows.
Layout conventions are the reference's: activations channels-last ``(B, T, C)``, Conv1D
kernels ``(k, C_in, C_out)``, Dense kernels ``(in, out)``  (src/layers.py:134).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

# src/model.py:9  -- the reference builds sqrt(2*pi) from a truncated pi, in fp32
_PI_REF = 3.14159265359
KERAS_EPSILON = 1e-7          # keras.backend.epsilon()
LEAKY_SLOPE = 0.2             # keras 'leaky_relu' activation string: negative_slope=0.2


# --------------------------------------------------------------------------------------
# configuration (src/model.py:14-34 constructor keywords, verbatim)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
  kernel_size: int = 2
  channels: int = 32
  blocks: int = 10
  layers_per_block: int = 1
  activation: Optional[str] = None
  conditioning: Optional[str] = None
  mapping_layers: Optional[Sequence[int]] = None
  mapping_activation: Optional[str] = None
  dilation_bound: int = 512
  num_mixtures: Optional[int] = None
  sampling_function: str = 'categorical'
  bits: int = 8
  skip_channels: Optional[int] = None
  dilation_channels: Optional[int] = None
  use_residual: bool = True
  use_skip: bool = True
  final_layers_channels: Sequence[int] = field(default_factory=list)
  l2_reg_factor: float = 0.0
  cond_inputs: int = 0          # width of the raw condition vector (global conditioning)

  @property
  def D(self) -> int:           # src/layers.py:49-50
    return self.channels if self.dilation_channels is None else self.dilation_channels

  @property
  def out_channels(self) -> int:  # src/model.py:115
    return 3 * self.num_mixtures if self.num_mixtures is not None else 2 ** self.bits


def dilation_schedule(cfg: OracleConfig) -> List[int]:
  """src/model.py:79-81."""
  max_power = int(math.log(cfg.dilation_bound, cfg.kernel_size))
  return [cfg.kernel_size ** (i % max_power)
          for i in range(cfg.layers_per_block * cfg.blocks)]


def receptive_field(cfg: OracleConfig) -> int:
  """src/model.py:122."""
  return 1 + sum(dilation_schedule(cfg)) * (cfg.kernel_size - 1) + 1


def mapping_widths(cfg: OracleConfig) -> List[int]:
  """src/model.py:125-130: None -> [], int -> [int]."""
  m = cfg.mapping_layers
  if m is None:
    return []
  if isinstance(m, int):
    return [m]
  return list(m)


def cond_channels(cfg: OracleConfig) -> int:
  """Width of the mapped condition fed to every block's conv_cond (src/model.py:141-148)."""
  w = mapping_widths(cfg)
  return w[-1] if w else cfg.cond_inputs


# --------------------------------------------------------------------------------------
# parameters, in Keras creation order
# --------------------------------------------------------------------------------------
def param_shapes(cfg: OracleConfig) -> List[Tuple[str, Tuple[int, ...]]]:
  """Names and shapes of every trainable variable in Keras creation order.

  causal (src/model.py:84-88); per block: dilated stack, conv1, conv_skip, conv_cond
  (src/layers.py:62-120); final convs (src/model.py:105-119); mapping Dense stack
  (src/model.py:141-147).
  """
  k, R, D, S = cfg.kernel_size, cfg.channels, cfg.D, cfg.skip_channels
  out = [('causal/kernel', (k, 1, R)), ('causal/bias', (R,))]
  cc = cond_channels(cfg)
  for b in range(cfg.blocks):
    cin = R
    for i in range(cfg.layers_per_block):
      cout = 2 * D if i == cfg.layers_per_block - 1 else D
      out.append((f'block{b}/dil{i}/kernel', (k, cin, cout)))
      out.append((f'block{b}/dil{i}/bias', (cout,)))
      cin = cout
    out.append((f'block{b}/conv1/kernel', (1, D, R)))
    out.append((f'block{b}/conv1/bias', (R,)))
    if S is not None:
      out.append((f'block{b}/conv_skip/kernel', (1, D, S)))
      out.append((f'block{b}/conv_skip/bias', (S,)))
    if cfg.conditioning is not None:
      out.append((f'block{b}/conv_cond/kernel', (1, cc, 2 * D)))
      out.append((f'block{b}/conv_cond/bias', (2 * D,)))
  cprev = (S if S is not None else R) if cfg.use_skip else R
  for i, ch in enumerate(list(cfg.final_layers_channels) + [cfg.out_channels]):
    out.append((f'final{i}/kernel', (1, cprev, ch)))
    out.append((f'final{i}/bias', (ch,)))
    cprev = ch
  if cfg.conditioning == 'global':
    cin = cfg.cond_inputs
    for j, w in enumerate(mapping_widths(cfg)):
      out.append((f'mapping{j}/kernel', (cin, w)))
      out.append((f'mapping{j}/bias', (w,)))
      cin = w
  return out


def init_params(cfg: OracleConfig, seed: int = 0, bias_range: float = 0.1,
                dtype=torch.float32) -> List[torch.Tensor]:
  """Glorot-uniform kernels (Keras default); biases U(-bias_range, bias_range).

  Keras initialises biases to zero; parity tests use non-zero biases so that bias bugs
  are visible.
  """
  g = torch.Generator().manual_seed(seed)
  params = []
  for name, shape in param_shapes(cfg):
    if name.endswith('kernel'):
      if len(shape) == 3:
        fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
      else:
        fan_in, fan_out = shape
      lim = math.sqrt(6.0 / (fan_in + fan_out))
      p = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim
    else:
      p = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bias_range
    params.append(p.to(dtype))
  return params


# --------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------
def activation(x: torch.Tensor, name: Optional[str]) -> torch.Tensor:
  """Keras activation strings used by the reference configs (train.py:35,38)."""
  if name is None or name == 'linear':
    return x
  if name == 'relu':
    return torch.relu(x)
  if name == 'leaky_relu':
    return torch.where(x >= 0, x, x * LEAKY_SLOPE)
  if name == 'tanh':
    return torch.tanh(x)
  if name == 'sigmoid':
    return torch.sigmoid(x)
  if name == 'elu':
    return torch.where(x > 0, x, torch.expm1(x))
  raise NotImplementedError(name)


KINK_TOL = 1e-4   # = the per-activation tolerance of north_star: inside it the branch of a kinked activation is undecided


def activation_with_branch(x: torch.Tensor, name: Optional[str], branch_pos: Optional[torch.Tensor]):
  """``activation`` for gradient comparisons through relu / leaky_relu.  Their derivative jumps at 0, so where a
  pre-activation lies within KINK_TOL of the kink an implementation that is accurate to the stated tolerance may
  legitimately sit on either side.  ``branch_pos`` (bool, same shape) says which side the implementation under
  test took (post-activation >= 0 / > 0); it is used ONLY for those undecided elements, every other element takes
  the branch of its own sign.  Returns (y, number of elements decided by branch_pos)."""
  if branch_pos is None or name not in ('relu', 'leaky_relu'):
    return activation(x, name), 0
  own = (x >= 0) if name == 'leaky_relu' else (x > 0)
  near = x.detach().abs() < KINK_TOL
  pos = torch.where(near, branch_pos.to(torch.bool), own)
  slope = LEAKY_SLOPE if name == 'leaky_relu' else 0.0
  return torch.where(pos, x, x * slope), int((near & (pos != own)).sum())


def causal_conv1d(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor,
                  dilation: int = 1) -> torch.Tensor:
  """Keras Conv1D(padding='causal'): left zero-pad d*(k-1), VALID dilated correlation.

  y[b,t,:] = sum_j kernel[j]^T x[b, t-(k-1-j)*d, :] + bias     (src/layers.py:66-88,
  src/model.py:84-88).  x: (B,T,Cin); kernel: (k,Cin,Cout).
  """
  k = kernel.shape[0]
  B, T, _ = x.shape
  y = bias.expand(B, T, -1).clone()
  for j in range(k):
    s = (k - 1 - j) * dilation
    if s >= T:
      continue
    xs = torch.zeros_like(x)
    xs[:, s:, :] = x[:, :T - s, :]
    y = y + xs @ kernel[j]
  return y


def conv1x1(x: torch.Tensor, kernel: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
  """Conv1D(kernel_size=1): per-timestep matmul (src/layers.py:92-104)."""
  return x @ kernel[0] + bias


# --------------------------------------------------------------------------------------
# block and model forward
# --------------------------------------------------------------------------------------
class _ParamCursor:
  def __init__(self, params):
    self.p, self.i = params, 0

  def take(self, n=1):
    out = self.p[self.i:self.i + n]
    self.i += n
    return out


def _hash32(x: np.ndarray) -> np.ndarray:
  x = x.astype(np.uint64)
  m = np.uint64(0xFFFFFFFF)
  x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
  x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
  x ^= x >> np.uint64(16)
  return x


def dropout_key(seed: int, block: int, step: int) -> int:
  """Restates wn_dropout_key (wavenets_amd/csrc/wn_elem.hip); all arithmetic mod 2^32."""
  M = 0xFFFFFFFF
  k = ((seed & M) * 0x9E3779B9 + (seed >> 32)) & M
  k ^= ((block & M) * 0x85EBCA6B + 0x1234567) & M
  k ^= (((step & M) * 0xC2B2AE35) + ((step >> 32) & M) * 0x27D4EB2F) & M
  return k


def dropout_keep_mask(n: int, key: int, rate: float) -> torch.Tensor:
  """Keep-mask of the product's stateless dropout hash over element indices 0..n-1 (the TF
  random stream of tf.keras.layers.Dropout, src/layers.py:108-111, is not reproducible)."""
  idx = np.arange(n, dtype=np.uint64)
  lo = idx & np.uint64(0xFFFFFFFF)
  hi = idx >> np.uint64(32)
  h = _hash32(lo ^ _hash32((hi + np.uint64(key)) & np.uint64(0xFFFFFFFF)))
  u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
  return torch.from_numpy(u >= np.float32(rate))


def layer_forward(x: torch.Tensor, layer_params: Sequence[torch.Tensor], *,
                  dilations: Sequence[int], activation_name: Optional[str],
                  residual: bool, has_skip: bool, cond: Optional[torch.Tensor] = None,
                  drop: Optional[Tuple[float, int]] = None,
                  inner_branch: Optional[Sequence[torch.Tensor]] = None, kink_log: Optional[list] = None
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
  """WaveNetLayer.call, src/layers.py:178-224 (dropout omitted: rate 0 in parity runs).

  layer_params order: [dil kernels/biases ...], conv1 k/b, [conv_skip k/b], [conv_cond k/b].
  cond: (B, T, Cc) or (B, 1, Cc) (broadcast over T), already mapped.
  """
  cur = _ParamCursor(list(layer_params))
  res = x
  h = x
  if drop is not None:                             # src/layers.py:192-196: conv input only
    rate, key = drop
    keep = dropout_keep_mask(x.numel(), key, rate).view(x.shape).to(x.dtype)
    h = x * keep * (1.0 / (1.0 - np.float32(rate)).astype(np.float32).item())
  n = len(dilations)
  for i, d in enumerate(dilations):
    kern, b = cur.take(2)
    h = causal_conv1d(h, kern, b, d)
    if i < n - 1:                                 # src/layers.py:66-74
      # (inner_branch: gradient comparisons only -- see activation_with_branch; kink_log collects (count, worst |pre|))
      br_i = inner_branch[i] if inner_branch is not None else None
      pre = h
      h, nov = activation_with_branch(pre, activation_name, br_i)
      if kink_log is not None and br_i is not None and activation_name in ('relu', 'leaky_relu'):
        own = (pre >= 0) if activation_name == 'leaky_relu' else (pre > 0)
        over = own != br_i.to(torch.bool)
        kink_log.append((int(over.sum()), float(pre.detach()[over].abs().max()) if bool(over.any()) else 0.0, nov))
  kr, br = cur.take(2)
  if has_skip:
    ks, bs = cur.take(2)
  if cond is not None:
    kc, bc = cur.take(2)
    h = h + conv1x1(cond, kc, bc)                  # src/layers.py:203-204
  D = h.shape[-1] // 2
  t, s = h[..., :D], h[..., D:]                    # src/layers.py:208
  z = torch.tanh(t) * torch.sigmoid(s)             # src/layers.py:210
  x_out = conv1x1(z, kr, br)                       # src/layers.py:213
  skip = conv1x1(z, ks, bs) if has_skip else x_out  # src/layers.py:216-219
  if residual:
    x_out = x_out + res                            # src/layers.py:222-223
  return x_out, skip


def layer_generate(gathered: torch.Tensor, layer_params: Sequence[torch.Tensor], *,
                   has_skip: bool, cond: Optional[torch.Tensor] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
  """WaveNetLayer.generate, src/layers.py:226-290: the single-step form of a depth-1 block for a
  queued sampler.  ``gathered`` (B, k, R) already holds the taps the dilated conv would read,
  oldest first ([x[t-d], x[t]] for k = 2); the conv runs undilated with VALID padding (:257-260),
  then conv_cond (:263-267), the gate (:271-273), conv1 (:276-278), conv_skip or the pre-residual
  alias (:281-286) and the residual with the LAST tap (:251, :289).  Returns (B,1,R), (B,1,S)."""
  cur = _ParamCursor(list(layer_params))
  kern, b = cur.take(2)                            # (k, R, 2D)
  kr, br = cur.take(2)
  if has_skip:
    ks, bs = cur.take(2)
  k = kern.shape[0]
  if gathered.shape[1] != k:
    raise ValueError('gathered input must hold exactly kernel-size taps')
  residual = gathered[:, -1:, :]                   # src/layers.py:251
  h = sum(gathered[:, j:j + 1, :] @ kern[j] for j in range(k)) + b
  if cond is not None:
    kc, bc = cur.take(2)
    h = h + conv1x1(cond, kc, bc)
  D = h.shape[-1] // 2
  z = torch.tanh(h[..., :D]) * torch.sigmoid(h[..., D:])
  x_out = conv1x1(z, kr, br)
  skip = conv1x1(z, ks, bs) if has_skip else x_out
  return x_out + residual, skip


def _params_per_block(cfg: OracleConfig) -> int:
  n = 2 * cfg.layers_per_block + 2
  if cfg.skip_channels is not None:
    n += 2
  if cfg.conditioning is not None:
    n += 2
  return n


def mapping_forward(cond: torch.Tensor, params: Sequence[torch.Tensor],
                    cfg: OracleConfig) -> torch.Tensor:
  """Global-conditioning mapping net: Dense stack + Identity (src/model.py:141-148)."""
  npb = _params_per_block(cfg)
  nfinal = 2 * (len(cfg.final_layers_channels) + 1)
  off = 2 + cfg.blocks * npb + nfinal
  m = cond
  for j in range(len(mapping_widths(cfg))):
    m = activation(m @ params[off + 2 * j] + params[off + 2 * j + 1], cfg.mapping_activation)
  return m


def model_forward(x: torch.Tensor, params: Sequence[torch.Tensor], cfg: OracleConfig,
                  cond: Optional[torch.Tensor] = None, return_logits: bool = False,
                  return_intermediates: bool = False, dropout: Optional[Tuple[float, int, int]] = None,
                  head_branch: Optional[Sequence[torch.Tensor]] = None,
                  inner_branch: Optional[Sequence[Sequence[torch.Tensor]]] = None, kink_log: Optional[list] = None):
  """WaveNet.call, src/model.py:213-239.  x: (B,T,1); cond: (B, cond_inputs) or None.

  Returns probabilities for the categorical head (softmax activation on the last conv,
  src/model.py:113-119) or the linear 3*M mixture parameters.  ``return_logits`` returns
  the pre-softmax values instead (used by the fused loss checks).
  """
  if cfg.conditioning == 'local':
    raise NotImplementedError('local conditioning is broken in the reference '
                              '(src/model.py:136-137) and is not restated')
  dil = dilation_schedule(cfg)
  cur = _ParamCursor(list(params))
  ck, cb = cur.take(2)
  c = None
  if cfg.conditioning == 'global':
    c = mapping_forward(cond, params, cfg).unsqueeze(1)   # (B,1,Cc), broadcast = tf.repeat
  h = causal_conv1d(x, ck, cb, 1)                          # src/model.py:228
  skips, inter = [], {'h': [h]}
  npb = _params_per_block(cfg)
  lpb = cfg.layers_per_block
  for b in range(cfg.blocks):
    lp = cur.take(npb)
    drop = None
    if dropout is not None and dropout[0] > 0:          # (rate, seed, step): training-mode dropout
      drop = (dropout[0], dropout_key(dropout[1], b, dropout[2]))
    h, sk = layer_forward(h, lp, dilations=dil[b * lpb:(b + 1) * lpb],
                          activation_name=cfg.activation, residual=cfg.use_residual,
                          has_skip=cfg.skip_channels is not None, cond=c, drop=drop,
                          inner_branch=inner_branch[b] if inner_branch is not None else None, kink_log=kink_log)
    skips.append(sk)
    inter['h'].append(h)
  if cfg.use_skip:
    h = skips[0]
    for sk in skips[1:]:
      h = h + sk                                           # src/model.py:235-236
  inter['skip_sum'] = h
  nf = len(cfg.final_layers_channels)
  inter['head_pre'], inter['kink_overrides'] = [], 0
  for i in range(nf):
    fk, fb = cur.take(2)
    a = conv1x1(h, fk, fb)
    inter['head_pre'].append(a)
    # src/model.py:105-111 (head_branch: see activation_with_branch -- gradient comparisons only)
    h, nov = activation_with_branch(a, cfg.activation, head_branch[i] if head_branch is not None else None)
    inter['kink_overrides'] += nov
  fk, fb = cur.take(2)
  logits = conv1x1(h, fk, fb)
  inter['logits'] = logits
  if cfg.num_mixtures is None and not return_logits:
    out = torch.softmax(logits, dim=-1)                    # src/model.py:116
  else:
    out = logits
  if return_intermediates:
    return out, inter
  return out


# --------------------------------------------------------------------------------------
# quantiser / companding  (integer path: bit-exact)
# --------------------------------------------------------------------------------------
def quantiser_edges(bits: int) -> np.ndarray:
  """src/model.py:151-153: np.linspace(-1, 1, 2**bits+1)[1:-1] (as Python floats ->
  the Keras Discretization layer stores them as float32)."""
  return np.asarray(np.linspace(-1, 1, num=2 ** bits + 1).tolist()[1:-1], dtype=np.float32)


def quantize(x: torch.Tensor, bits: int) -> torch.Tensor:
  """Keras Discretization == tf Bucketize: index = number of edges <= x (upper bound).

  src/model.py:151-153, SURVEY.md row Q1.  Returns int64 of x's shape.
  """
  edges = torch.from_numpy(quantiser_edges(bits))
  return torch.bucketize(x.to(torch.float32).contiguous(), edges, right=True)


def dequantize(idx: torch.Tensor, bits: int) -> torch.Tensor:
  """src/model.py:411,418: i / 2**(bits-1) - 1  (left bin edge)."""
  return idx.to(torch.float32) / 2.0 ** (bits - 1) - 1.0


def mu_law(x: torch.Tensor) -> torch.Tensor:
  """src/utils.py:34-35."""
  return torch.sign(x) * (torch.log(1.0 + 255.0 * torch.abs(x)) / math.log(256.0))


def inverse_mu_law(y: torch.Tensor) -> torch.Tensor:
  """src/callbacks.py:126-131."""
  return torch.sign(y) * (torch.pow(torch.tensor(256.0, dtype=y.dtype), torch.abs(y)) - 1.0) / 255.0


# --------------------------------------------------------------------------------------
# losses (per (b,t), shape (B,T))  src/model.py:505-551
# --------------------------------------------------------------------------------------
def loss_categorical(target_idx: torch.Tensor, probs: torch.Tensor) -> torch.Tensor:
  """keras.losses.sparse_categorical_crossentropy(target, probs), from_logits=False.

  p = clip(probs, eps, 1-eps); loss = -log_softmax(log p)[target]   (SURVEY.md row O1).
  target_idx: (B,T,1) or (B,T) integer.
  """
  if target_idx.dim() == 3:
    target_idx = target_idx[..., 0]
  p = torch.clamp(probs, KERAS_EPSILON, 1.0 - KERAS_EPSILON)
  lp = torch.log(p)
  lsm = lp - torch.logsumexp(lp, dim=-1, keepdim=True)
  return -torch.gather(lsm, -1, target_idx.long().unsqueeze(-1))[..., 0]


def loss_logistic(target: torch.Tensor, pred: torch.Tensor, num_mixtures: int,
                  bits: int) -> torch.Tensor:
  """Discretised mixture of logistics as written in src/model.py:533-547."""
  M = num_mixtures
  w, mu, ls = pred[..., :M], pred[..., M:2 * M], pred[..., 2 * M:]
  y = target.expand(*target.shape[:-1], M)
  w = torch.softmax(w, dim=-1)
  halfbit = 0.5 * 1 / (2 ** bits)
  ls = torch.clamp(ls, min=-7.0)
  inv = torch.exp(-1.0 * ls)
  lik = torch.sum(w * (torch.sigmoid((y - mu + halfbit) * inv)
                       - torch.sigmoid((y - mu - halfbit) * inv)), dim=-1)
  return -1.0 * torch.log(lik)


def loss_gaussian(target: torch.Tensor, pred: torch.Tensor, num_mixtures: int) -> torch.Tensor:
  """Mixture of gaussians as written in src/model.py:517-532."""
  M = num_mixtures
  w, mu, ls = pred[..., :M], pred[..., M:2 * M], pred[..., 2 * M:]
  y = target.expand(*target.shape[:-1], M)
  w = torch.softmax(w, dim=-1)
  ls = torch.clamp(ls, min=-7.0)
  sc = torch.exp(ls)
  sqrt2pi = math.sqrt(2.0 * _PI_REF)
  xx = torch.clamp((y - mu) / sc, max=1e8)
  lik = torch.sum(w * (torch.exp(-0.5 * xx * xx) / (sc * sqrt2pi)), dim=-1)
  return -1.0 * torch.log(lik)


def loss_fn(target: torch.Tensor, pred: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
  """WaveNet.loss_fn dispatch, src/model.py:515-549."""
  if cfg.sampling_function == 'categorical':
    return loss_categorical(target, pred)
  if cfg.sampling_function == 'logistic':
    return loss_logistic(target, pred, cfg.num_mixtures, cfg.bits)
  if cfg.sampling_function == 'gaussian':
    return loss_gaussian(target, pred, cfg.num_mixtures)
  raise NotImplementedError(cfg.sampling_function)


def prepare_target(x: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
  """src/model.py:151-155."""
  return quantize(x, cfg.bits) if cfg.num_mixtures is None else x


# --------------------------------------------------------------------------------------
# deterministic samplers  (src/model.py:393-503; stochastic draws use TF's RNG -> not
# reproducible; only the deterministic branches are restated)
# --------------------------------------------------------------------------------------
def sample_waveform_deterministic(pred: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
  """(B,T,C_out) -> (B,T,1).  categorical: argmax -> i/2^(bits-1)-1 (src/model.py:415-418,
  with the intended trailing axis); mixtures: mean of the arg-max-weight
  component clipped to [-1,1] (src/model.py:447-458, 487-498)."""
  if cfg.sampling_function == 'categorical':
    idx = torch.argmax(pred, dim=-1)
    return dequantize(idx, cfg.bits).unsqueeze(-1)
  M = cfg.num_mixtures
  w, mu = pred[..., :M], pred[..., M:2 * M]
  sel = torch.argmax(torch.softmax(w, dim=-1), dim=-1, keepdim=True)
  return torch.clamp(torch.gather(mu, -1, sel), -1.0, 1.0)


# --------------------------------------------------------------------------------------
# training step  (src/model.py:309-348, train.py:225-226)
# --------------------------------------------------------------------------------------
def mixture_sample_cdf(pred_row: torch.Tensor, cfg: OracleConfig, v: torch.Tensor) -> torch.Tensor:
  """P(sample <= v), -1 <= v < 1, of the NON-deterministic mixture samplers for one prediction row
  (3M,): component m is drawn with probability softmax(w)_m, then x = mu_m + e^{s_m} eps with
  eps = ln z - ln(1-z), z ~ U(0,1) (standard logistic; src/model.py:463-483) or eps ~ N(0,1)
  (src/model.py:423-443), and x is clipped to [-1, 1] (atoms at both ends; no floor on s here, unlike
  the losses).  TF's random stream cannot be matched, so the product's draws are tested against this law."""
  M = cfg.num_mixtures
  p = pred_row.double()
  w = torch.softmax(p[:M], -1)
  mu, sc = p[M:2 * M], torch.exp(p[2 * M:])
  a = (v.double().unsqueeze(-1) - mu) / sc
  if cfg.sampling_function == 'logistic':
    F = torch.sigmoid(a)
  else:
    F = 0.5 * (1.0 + torch.erf(a / math.sqrt(2.0)))
  return (F * w).sum(-1)


def l2_penalty(params: Sequence[torch.Tensor], cfg: OracleConfig) -> torch.Tensor:
  """sum over every kernel of l2 * sum(W^2)  (kernel_regularizer=L2(l2) on every conv and
  dense, src/layers.py:74,88,96,104,120; src/model.py:88,110,118,146)."""
  tot = params[0].new_zeros(())
  for (name, _), p in zip(param_shapes(cfg), params):
    if name.endswith('kernel'):
      tot = tot + cfg.l2_reg_factor * torch.sum(p * p)
  return tot


def loss_and_grads(x: torch.Tensor, params: Sequence[torch.Tensor], cfg: OracleConfig,
                   cond: Optional[torch.Tensor] = None, global_batch: Optional[int] = None,
                   n_replicas: int = 1, dropout: Optional[Tuple[float, int, int]] = None,
                   head_branch: Optional[Sequence[torch.Tensor]] = None,
                   inner_branch: Optional[Sequence[Sequence[torch.Tensor]]] = None, kink_log: Optional[list] = None):
  """Forward + loss + reverse-mode gradients of one replica's share of a train step.

  x: (B, T+1, 1).  inputs = x[:, :-1], target = prepare_target(x[:, 1:]) (src/model.py:
  319-321); loss = sum_{b,t} l[b,t] / B_global (tf.nn.compute_average_loss,
  src/model.py:328-329); + l2/n_replicas (scale_regularization_loss, :331-334).
  Returns (loss, reg_loss, grads list, pred).
  """
  ps = [p.detach().clone().requires_grad_(True) for p in params]
  inputs, y_true = x[:, :-1, :], x[:, 1:, :]
  target = prepare_target(y_true, cfg)
  pred = model_forward(inputs, ps, cfg, cond, dropout=dropout, head_branch=head_branch, inner_branch=inner_branch,
                       kink_log=kink_log)
  per = loss_fn(target, pred, cfg)                  # (B,T)
  Bg = x.shape[0] if global_batch is None else global_batch
  loss = per.sum() / Bg
  reg = l2_penalty(ps, cfg) / n_replicas if cfg.l2_reg_factor > 0 else loss.new_zeros(())
  total = loss + reg
  grads = torch.autograd.grad(total, ps, allow_unused=True)
  grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, ps)]
  return loss.detach(), reg.detach(), grads, pred.detach()


def clip_by_norm_per_tensor(grads: Sequence[torch.Tensor], clipnorm: float) -> List[torch.Tensor]:
  """Keras optimizer clipnorm: each gradient tensor is clipped independently,
  g * clipnorm / max(||g||, clipnorm)  (train.py:225-226)."""
  out = []
  for g in grads:
    n = torch.sqrt(torch.sum(g * g))
    out.append(g * (clipnorm / torch.maximum(n, torch.tensor(clipnorm, dtype=g.dtype))))
  return out


def keras_adam_step(params, grads, m, v, step: int, lr: float,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7):
  """One Keras Adam update (epsilon added to sqrt(v), outside the bias correction).

  alpha = lr * sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2);
  p -= alpha*m/(sqrt(v)+eps).  step is 1-based.
  """
  alpha = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
  np_, nm, nv = [], [], []
  for p, g, mi, vi in zip(params, grads, m, v):
    mi = mi + (g - mi) * (1.0 - beta1)
    vi = vi + (g * g - vi) * (1.0 - beta2)
    p = p - alpha * mi / (torch.sqrt(vi) + eps)
    np_.append(p); nm.append(mi); nv.append(vi)
  return np_, nm, nv


def train_step(x, params, m, v, step, cfg: OracleConfig, lr: float = 5e-4,
               clipnorm: float = 1.0, cond=None, global_batch=None, n_replicas=1):
  """One full single-replica train step: grads -> per-tensor clip -> Keras Adam."""
  loss, reg, grads, pred = loss_and_grads(x, params, cfg, cond, global_batch, n_replicas)
  grads = clip_by_norm_per_tensor(grads, clipnorm)
  params, m, v = keras_adam_step(params, grads, m, v, step, lr)
  return loss, params, m, v


# --------------------------------------------------------------------------------------
# naive sliding-window generation  (intended semantics of src/model.py:241-307)
# --------------------------------------------------------------------------------------
def generate_naive(params, cfg: OracleConfig, length: int, window: torch.Tensor,
                   cond: Optional[torch.Tensor] = None) -> torch.Tensor:
  """window: (B, RF, 1) initial samples.  Each step: full forward over the window, keep the
  last step, deterministic sample, slide (src/model.py:296-305).  Returns (B, length, 1)."""
  x = window.clone()
  out = []
  with torch.no_grad():
    for _ in range(length):
      pred = model_forward(x, params, cfg, cond)[:, -1:, :]
      s = sample_waveform_deterministic(pred, cfg)         # (B,1,1)
      out.append(s)
      x = torch.cat([x[:, 1:], s.to(x.dtype)], dim=1)
  return torch.cat(out, dim=1)


# --------------------------------------------------------------------------------------
# synthetic benchmark input  
# --------------------------------------------------------------------------------------
def synthetic_waveform(batch: int, length: int, seed: int = 1234) -> torch.Tensor:
  """(B, length, 1) fp32 mu-law companded two-tone + noise at 16 kHz."""
  g = torch.Generator().manual_seed(seed)
  n = torch.arange(length, dtype=torch.float64)[None, :]
  f0 = 80.0 + 320.0 * torch.rand((batch, 1), generator=g, dtype=torch.float64)
  ph = 2 * math.pi * torch.rand((batch, 1), generator=g, dtype=torch.float64)
  x = (0.6 * torch.sin(2 * math.pi * f0 * n / 16000.0 + ph)
       + 0.2 * torch.sin(2 * math.pi * 3 * f0 * n / 16000.0)
       + 0.02 * torch.randn((batch, length), generator=g, dtype=torch.float64))
  x = torch.clamp(x, -1.0, 1.0)
  return mu_law(x).to(torch.float32).unsqueeze(-1)
