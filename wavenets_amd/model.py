"""WaveNet model: host-side mirror of src/model.py::WaveNet over the gfx950 HIP library.

Same constructor keywords, method names, argument meaning, tensor layouts
(channels-last ``(B, T, C)`` fp32) and exception types as the reference class; everything
underneath is ``libwn_hip.so`` (no TensorFlow, no CPU fallback).  PyTorch is used only for
device memory, streams and ``torch.distributed`` (RCCL).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import spec as _spec
from .layers import WaveNetLayer
from .optim import Adam


class _ConvHandle:
  """Read-only handle on one conv / dense of the flat parameter buffer (``.kernel``/``.bias``)."""

  def __init__(self, model, name):
    self._model, self._name = model, name

  @property
  def kernel(self):
    return self._model._tensor(self._name + '/kernel')

  @property
  def bias(self):
    return self._model._tensor(self._name + '/bias')

  @property
  def weights(self):
    return [self.kernel, self.bias]


class MeanSquaredError:
  """Minimal stand-in for tf.keras.metrics.MeanSquaredError (train.py:227)."""
  name = 'mean_squared_error'

  def __init__(self):
    self._scratch = None
    self.reset_state()

  def reset_state(self):
    self.total, self.count = 0.0, 0

  def update_state(self, y_true, y_pred):
    self.total += float(self.update_state_device(y_true, y_pred))
    self.count += 1

  # two-phase form used by train_step / test_step: the reduction is queued behind the step's kernels (one library launch
  # pair, no torch arithmetic) and its value comes back in the step's single device-to-host read.  ``out``: a one-float
  # device view to write into (the gradient bucket's tail in a training step); ``scale`` = 1 / replicas makes the SUM
  # over the replicas the mean Keras reports.
  def update_state_device(self, y_true, y_pred, out=None, scale=1.0):
    n = y_true.numel()
    if out is None:
      out = torch.empty(1, dtype=torch.float32, device=y_true.device)
    if self._scratch is None or self._scratch.device != y_true.device:
      self._scratch = torch.empty(2048, dtype=torch.float32, device=y_true.device)
    _lib.check(_lib.lib().wn_sum_squared_error(_lib.ptr(y_true.contiguous()), _lib.ptr(y_pred.contiguous()), n,
                                               float(scale) / n, _lib.ptr(out), _lib.ptr(self._scratch),
                                               _lib.stream_ptr()))
    return out

  def commit(self, value):
    self.total += float(value)
    self.count += 1

  def result(self):
    return self.total / max(self.count, 1)


class _Mean:
  def __init__(self, name):
    self.name = name
    self.reset_state()

  def reset_state(self):
    self.total, self.count = 0.0, 0

  def update_state(self, v):
    self.total += float(v)
    self.count += 1

  def result(self):
    return self.total / max(self.count, 1)


class WaveNet(torch.nn.Module):
  """WaveNet model class (src/model.py:11-556)."""
  _TAIL_METRICS = 5      # device-reduced metric slots behind {loss, reg_loss, range_flag} in the gradient bucket

  def __init__(self,
               kernel_size: int = 2,
               channels: int = 32,
               blocks: int = 10,
               layers_per_block: int = 1,
               activation=None,
               conditioning=None,
               mapping_layers=None,
               mapping_activation=None,
               dropout: float = 0,
               dilation_bound: int = 512,
               num_mixtures=None,
               sampling_function: str = 'categorical',
               bits=8,
               skip_channels=None,
               dilation_channels=None,
               use_residual=True,
               use_skip=True,
               final_layers_channels=None,
               l2_reg_factor: float = 0,
               device=None,
               seed: int = 0):
    super().__init__()
    s = _spec.validate(kernel_size, channels, blocks, layers_per_block, activation, conditioning,
                       mapping_layers, mapping_activation, dropout, dilation_bound, num_mixtures,
                       sampling_function, bits, skip_channels, dilation_channels, use_residual,
                       use_skip, final_layers_channels, l2_reg_factor)
    self.spec = s
    self.regularization = s.l2_reg_factor > 0
    self.num_mixtures = num_mixtures
    self.use_skip = use_skip
    self.sampling_function = sampling_function
    self.bits = bits
    self.conditioning = conditioning
    self.dropout = s.dropout
    self.receptive_field = s.receptive_field            # src/model.py:122
    self._device = torch.device(device) if device is not None else torch.device('cuda', 0)
    self._seed = seed
    self._plan = None
    self._cond_inputs = None
    self._names: List[str] = []
    self._ws = {}
    self.built = False
    self.optimizer = None
    self._metrics_from_compilation = []
    self.loss_tracker = _Mean('loss')
    self.reg_loss = _Mean('reg_loss') if self.regularization else None
    self._sample_calls = 0
    # steps / evaluation passes / generate calls that left the range of the split-precision kernels and were repeated
    # in exact fp32 (each costs a second, slower pass; train.py logs them per epoch)
    self.train_guard_trips = 0
    self.test_guard_trips = 0
    self.generation_guard_trips = 0
    # None = automatic: a single replica reads its step scalars back between forward and backward; data-parallel replicas
    # take them from the gradient bucket's tail after the ONE all-reduce of the step (no second collective per step)
    self.early_logs = None
    # call() / logits() read the forward range guard back after every pass (one blocking device-to-host copy) and repeat
    # the pass in exact fp32 when it tripped.  False skips the read -- for latency-critical inference on weights known to
    # stay in range; range_tripped_last_forward() checks the same slot later.
    self.range_check = True
    self._last_fwd = None
    self._log_mirror, self._log_event = None, None
    self._drop_step = 0                   # training calls made so far (dropout mask counter; saved by io.save_weights)
    self._fused_step_sample = True        # train_step draws its metric sample inside the library
    # structure handles (attribute names of the reference)
    dil = s.dilations
    lpb = s.layers_per_block
    self.causal = _ConvHandle(self, 'causal')
    self.wavenet_blocks = [
        WaveNetLayer(kernel=s.kernel_size, channels=s.channels,
                     dilation_rate=dil[b * lpb:(b + 1) * lpb], activation=s.activation,
                     dilation_channels=s.dilation_channels, residual=s.use_residual,
                     skip_channels=s.skip_channels, l2_reg_factor=s.l2_reg_factor, dropout=s.dropout,
                     condition=conditioning is not None, _owner=(self, b))
        for b in range(s.blocks)]
    self.final = [_ConvHandle(self, f'final{i}') for i in range(len(s.final_layers_channels) + 1)]
    self.mapping = [_ConvHandle(self, f'mapping{j}') for j in range(len(s.mapping_layers))] \
        if conditioning == 'global' else None
    if conditioning is None:
      self.build(None)

  # ------------------------------------------------------------------ build
  def build(self, input_shape):
    """Create the plan and the parameters (src/model.py:171-211).  With conditioning the
    condition width is only known from the first input, as in Keras."""
    if self.built:
      return
    s = self.spec
    cond_inputs = 0
    if self.conditioning is not None:
      cond_inputs = int(input_shape[1][-1])
    self._cond_inputs = cond_inputs
    cfg = _lib.WnConfig()
    cfg.kernel_size, cfg.channels, cfg.blocks = s.kernel_size, s.channels, s.blocks
    cfg.layers_per_block = s.layers_per_block
    cfg.activation = _lib.ACTIVATIONS[s.activation]
    cfg.dilation_bound = s.dilation_bound
    cfg.num_mixtures = s.num_mixtures or 0
    cfg.head = _lib.HEADS[s.sampling_function]
    cfg.bits = s.bits
    cfg.skip_channels = s.skip_channels or 0
    cfg.dilation_channels = s.dilation_channels or 0
    cfg.use_residual, cfg.use_skip = int(s.use_residual), int(s.use_skip)
    if len(s.final_layers_channels) > _lib.WN_MAX_FINAL or len(s.mapping_layers) > _lib.WN_MAX_MAPPING:
      raise NotImplementedError('too many final / mapping layers')
    cfg.n_final = len(s.final_layers_channels)
    for i, c in enumerate(s.final_layers_channels):
      cfg.final_channels[i] = c
    cfg.cond_inputs = cond_inputs
    cfg.n_mapping = len(s.mapping_layers)
    for i, c in enumerate(s.mapping_layers):
      cfg.mapping_channels[i] = c
    cfg.mapping_activation = _lib.ACTIVATIONS[s.mapping_activation]
    cfg.l2_reg_factor = s.l2_reg_factor
    L = _lib.lib()
    plan = L.wn_plan_create(C.byref(cfg))
    if not plan:
      raise ValueError(L.wn_last_error_string().decode())
    self._plan = C.c_void_p(plan)
    if s.dropout > 0:
      if s.dropout >= 1:
        raise ValueError('Dropout must be between 0 and 1.')
      _lib.check(L.wn_plan_set_dropout(self._plan, s.dropout, self._seed, 0))
    shapes = s.param_shapes(cond_inputs)
    n = L.wn_plan_num_tensors(self._plan)
    assert n == len(shapes), (n, len(shapes))
    self._names = [nm for nm, _ in shapes]
    self._offsets, self._shapes = [], []
    off, ln = C.c_int64(), C.c_int64()
    nd, isk = C.c_int32(), C.c_int32()
    sh = (C.c_int64 * 3)()
    for i, (nm, shp) in enumerate(shapes):
      _lib.check(L.wn_plan_tensor_info(self._plan, i, C.byref(off), C.byref(ln), C.byref(nd), sh, C.byref(isk)))
      assert tuple(sh[:nd.value]) == tuple(shp), (nm, tuple(sh[:nd.value]), shp)
      self._offsets.append(off.value)
      self._shapes.append(tuple(shp))
    total = L.wn_plan_param_count(self._plan)
    # Keras defaults: glorot-uniform kernels, zero biases (no initializer passed anywhere)
    g = torch.Generator().manual_seed(self._seed)
    flat = torch.zeros(total, dtype=torch.float32)
    for nm, shp, o in zip(self._names, self._shapes, self._offsets):
      if nm.endswith('kernel'):
        if len(shp) == 3:
          fan_in, fan_out = shp[0] * shp[1], shp[0] * shp[2]
        else:
          fan_in, fan_out = shp
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        cnt = int(np.prod(shp))
        flat[o:o + cnt] = (torch.rand(cnt, generator=g) * 2 - 1) * lim
    self.flat_params = torch.nn.Parameter(flat.to(self._device), requires_grad=False)
    # gradient bucket = [flat gradient | loss, reg_loss, range_flag, device-reduced metrics]: the data-parallel exchange
    # is ONE all-reduce
    self._grad_bucket = torch.zeros(self.flat_params.numel() + 3 + self._TAIL_METRICS, dtype=torch.float32,
                                    device=self._device)
    self.flat_grads = self._grad_bucket[:self.flat_params.numel()]
    self.built = True
    if self.optimizer is not None:
      self.optimizer.build(self)

  def __del__(self):
    try:
      if self._plan is not None:
        _lib.lib().wn_plan_destroy(self._plan)
        self._plan = None
    except Exception:  # interpreter shutdown
      pass

  def kernel_report(self) -> str:
    """Which kernel family each phase selects for this network (the fast paths are shape-specialised)."""
    buf = C.create_string_buffer(2048)
    _lib.check(_lib.lib().wn_plan_describe(self._plan, buf, 2048))
    return buf.value.decode()

  def _log_kernels_once(self):
    import os
    import sys
    if not getattr(self, '_kernels_logged', False) and os.environ.get('WN_LOG_KERNELS'):
      self._kernels_logged = True
      print('[wavenets_amd] ' + self.kernel_report(), file=sys.stderr)

  # ------------------------------------------------------------------ weights
  def _tensor(self, name):
    i = self._names.index(name)
    o, shp = self._offsets[i], self._shapes[i]
    return self.flat_params.data[o:o + int(np.prod(shp))].view(*shp)

  @property
  def trainable_variables(self):
    """Views on the flat buffer in Keras creation order."""
    return [self._tensor(n) for n in self._names]

  @property
  def variable_names(self):
    return list(self._names)

  def get_weights(self):
    return [t.detach().cpu().numpy().copy() for t in self.trainable_variables]

  def set_weights(self, weights):
    if len(weights) != len(self._names):
      raise ValueError(f'expected {len(self._names)} arrays, got {len(weights)}')
    for t, w in zip(self.trainable_variables, weights):
      w = torch.as_tensor(np.asarray(w), dtype=torch.float32)
      if tuple(w.shape) != tuple(t.shape):
        raise ValueError(f'shape mismatch {tuple(w.shape)} vs {tuple(t.shape)}')
      t.copy_(w)

  def gradients(self):
    """Gradient views matching trainable_variables (after train_step / loss_and_grads)."""
    return [self.flat_grads[o:o + int(np.prod(s))].view(*s) for o, s in zip(self._offsets, self._shapes)]

  # ------------------------------------------------------------------ compile
  def compile(self, **kwargs):
    """src/model.py:157-169."""
    if 'loss' in kwargs:
      raise ValueError('Loss must be set in the model init function.')
    self._metrics_from_compilation = []
    if 'metrics' in kwargs and kwargs['metrics'] is not None:
      for metric in kwargs['metrics']:
        self._metrics_from_compilation.append(metric)
    self.loss_tracker = _Mean('loss')
    if self.regularization:
      self.reg_loss = _Mean('reg_loss')
    self.optimizer = kwargs.get('optimizer')
    if self.optimizer is not None and self.built:
      self.optimizer.build(self)

  @property
  def metrics(self):
    # the reference appends to the same list on every access (src/model.py:350-360); not reproduced
    m = list(self._metrics_from_compilation) + [self.loss_tracker]
    if self.regularization:
      m.append(self.reg_loss)
    return m

  # ------------------------------------------------------------------ helpers
  def _workspace(self, key, floats):
    ws = self._ws.get(key)
    if ws is None or ws.numel() < floats:
      ws = torch.empty(int(floats), dtype=torch.float32, device=self._device)
      self._ws[key] = ws
    return ws

  def _split_inputs(self, inputs):
    if self.conditioning is not None:
      x, cond = inputs
      cond = torch.as_tensor(cond, dtype=torch.float32, device=self._device).contiguous()
    else:
      x, cond = inputs, None
    x = torch.as_tensor(x, dtype=torch.float32, device=self._device).contiguous()
    if x.dim() != 3 or x.shape[-1] != 1:
      raise ValueError('input must have shape (batch, samples, 1)')
    if not self.built:
      self.build([tuple(x.shape), tuple(cond.shape)] if cond is not None else tuple(x.shape))
    if cond is not None and (cond.dim() != 2 or cond.shape[0] != x.shape[0] or cond.shape[1] != self._cond_inputs):
      raise ValueError('condition must have shape (batch, n_cond)')
    return x, cond

  # ------------------------------------------------------------------ call
  def call(self, inputs, training=False):
    """src/model.py:213-239: probabilities (categorical) or linear mixture parameters.  (Reads the pass's range-guard
    float back -- one host sync -- and repeats the pass with the exact-fp32 kernels when an activation left the range
    of the split-precision ones.)"""
    x, cond = self._split_inputs(inputs)
    return self._forward_guarded(x, cond, want_probs=True, training=bool(training and self.dropout > 0))

  def _forward_guarded(self, x, cond, want_probs, training):
    B, T = x.shape[0], x.shape[1]
    L = _lib.lib()
    ws = self._workspace('train' if training else 'fwd', L.wn_plan_workspace_floats(self._plan, B, T, int(training)))
    out = torch.empty(B, T, self.spec.out_channels, dtype=torch.float32, device=self._device)
    fn = L.wn_forward_training if training else L.wn_forward
    if training:
      # the Dropout layers are active (src/layers.py:195-196): a fresh mask per call, as in a training step
      self._arm_dropout()

    def run():
      _lib.check(fn(self._plan, _lib.ptr(self.flat_params), _lib.ptr(x), _lib.ptr(cond), B, T,
                    _lib.ptr(out) if want_probs else None, None if want_probs else _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                    _lib.stream_ptr()))
    run()
    # the pass's guard float, copied out of the shared workspace on the stream (a later pass on the same workspace -- another
    # call, a test_step -- overwrites the slot; this one-float copy is what range_tripped_last_forward() reads)
    slot = L.wn_plan_range_slot(self._plan, B, T, int(training))
    self._last_fwd = (ws[slot:slot + 1].clone(), training)
    if self.range_check and L.wn_debug_value(1) != 1 and self._guard_over(*self._last_fwd):
      with self.exact_fp32():
        run()
    return out

  def range_tripped_last_forward(self) -> bool:
    """Deferred form of the guard check of call() / logits() (range_check = False): did the last call() / logits() pass
    leave the range of the split-precision kernels?  (Repeat it inside `with model.exact_fp32():` if so.  With
    range_check = False nothing else checks: an unchecked pass may hold inf / NaN.)"""
    if self._last_fwd is None:
      return False
    return bool(self._guard_over(*self._last_fwd))

  def forward(self, inputs, training=False):
    return self.call(inputs, training=training)

  def logits(self, inputs):
    """Pre-softmax head output (parity checks)."""
    x, cond = self._split_inputs(inputs)
    return self._forward_guarded(x, cond, want_probs=False, training=False)

  # ------------------------------------------------------------------ training
  @staticmethod
  def _world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
      return dist.get_world_size()
    return 1

  @staticmethod
  def _rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
      return dist.get_rank()
    return 0

  def training_intermediate(self, what: int, idx: int, B: int, T: int):
    """View on an intermediate the last loss_and_grads call (B utterances, T predicted samples) left in the
    training workspace (wn_debug_ws_region; parity tests compare these with the oracle's autograd values)."""
    off, ln = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().wn_debug_ws_region(self._plan, B, T, what, idx, C.byref(off), C.byref(ln)))
    return self._ws['train'][off.value:off.value + ln.value]

  def exact_fp32(self):
    """Context manager: the calling thread's launches use the exact-fp32 MFMA kernels (no fp16 hi|lo operand split)."""
    import contextlib

    @contextlib.contextmanager
    def _cm():
      L = _lib.lib()
      prev = L.wn_debug_value(1)
      L.wn_debug_set(1, 1)
      try:
        yield
      finally:
        L.wn_debug_set(1, prev)
    return _cm()

  def _guard_over(self, guard, training):
    """Reads a pass's copied range-guard float (one host sync) against the limit of the split-precision kernels."""
    m = float(guard)
    # with dropout the split kernels read H * mask / (1 - rate) while H is what the slot records
    limit = _lib.lib().wn_range_limit() * ((1.0 - self.dropout) if (training and self.dropout > 0) else 1.0)
    return not (m < limit)

  def set_drop_step(self, n: int):
    """Training calls already made (resume): the next dropout mask is that of call n + 1."""
    self._drop_step = int(n)

  def _arm_dropout(self):
    # every replica draws its own mask each step (MirroredStrategy runs an independent Dropout per replica):
    # mask counter = call_index * world + rank + 1; single process: 1, 2, 3, ...
    if self.dropout > 0:
      step = self._drop_step * self._world() + self._rank() + 1
      _lib.check(_lib.lib().wn_plan_set_dropout(self._plan, self.dropout, self._seed, step))
      self._drop_step += 1

  def loss_and_grads(self, data, global_batch=None, n_replicas=None, want_pred=False, want_sample=False,
                     _loss_in_bucket=False, _between=None):
    """Forward + loss + backward of this replica's rows (src/model.py:319-335).

    Fills ``self.flat_grads`` with d(sum_local l / B_global)/d(theta); returns
    (loss tensor[3] = {loss, reg_loss, range_flag}, pred or None, y_true).  range_flag = 1: a forward activation left
    the range of the split-precision kernels and the results are not valid (train_step then repeats the step with the
    exact-fp32 kernels; a direct caller checks loss[2] and does the same through ``exact_fp32()``).  ``want_sample``: the step also draws
    ``sample_waveform(pred)`` (src/model.py:338) from the logits inside the library -- same draw, no (B,T,C)
    probability tensor -- and returns it in place of pred; when the library cannot (deterministic / more than
    1024 classes) pred is returned and the caller samples from it."""
    x, cond = self._split_inputs(data)
    B, T = x.shape[0], x.shape[1] - 1
    if T < 1:
      raise ValueError('training data must hold at least 2 samples per utterance')
    world = self._world()
    if global_batch is None:
      global_batch = B * world                         # compute_average_loss, src/model.py:328-329
    if n_replicas is None:
      n_replicas = world
    L = _lib.lib()
    self._log_kernels_once()
    ws = self._workspace('train', L.wn_plan_workspace_floats(self._plan, B, T, 1))
    # train_step keeps {loss, reg_loss} in the gradient bucket's tail (one all-reduce); other callers get their own tensor
    loss = self._grad_bucket[self.flat_params.numel():] if _loss_in_bucket else \
        torch.empty(3, dtype=torch.float32, device=self._device)
    sample = None
    if want_sample and self._fused_step_sample:
      sample = torch.empty(B, T, 1, dtype=torch.float32, device=self._device)
      if L.wn_plan_arm_step_sample(self._plan, _lib.ptr(sample), 0, 0x0402, self._sample_calls + 1) == 0:
        self._sample_calls += 1
      else:
        sample = None
    want_pred = want_pred or (want_sample and sample is None)
    pred = torch.empty(B, T, self.spec.out_channels, dtype=torch.float32, device=self._device) if want_pred else None
    self._arm_dropout()

    def run():
      _lib.check(L.wn_train_fwd_bwd(self._plan, _lib.ptr(self.flat_params), _lib.ptr(x), _lib.ptr(cond), B, T,
                                    int(global_batch), int(n_replicas), _lib.ptr(self.flat_grads),
                                    _lib.ptr(loss), _lib.ptr(pred), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    try:
      if _between is None:
        run()
        out = (sample if sample is not None else self.sample_waveform(pred)) if want_sample else pred
      else:
        # the step as two calls: forward + loss, the caller's own work (its read-back of loss / metrics), backward
        try:
          L.wn_plan_set_train_phases(self._plan, 1)
          run()
          out = (sample if sample is not None else self.sample_waveform(pred)) if want_sample else pred
          _between(loss, out, x[:, 1:, :])
          L.wn_plan_set_train_phases(self._plan, 2)
          run()
        finally:
          L.wn_plan_set_train_phases(self._plan, 3)
    finally:
      if sample is not None:
        L.wn_plan_arm_step_sample(self._plan, None, 0, 0, 0)      # never leave a stale pointer armed
    return loss, out, x[:, 1:, :]

  def train_step(self, data):
    """src/model.py:309-348.  Data-parallel: the flat gradient (and the step's scalars) are
    SUM-all-reduced over RCCL before the (replicated, deterministic) optimizer step.

    Range guard: the split-precision kernels need |activation| < 65504.  The step's range flag rides in the bucket's
    tail through the all-reduce, the optimizer kernel skips the update on every replica when it is set, and the step is
    then repeated here with the exact-fp32 kernels (same dropout mask, same optimizer iteration); ``train_guard_trips``
    counts those repeats (a tripped step costs a second, ~2.7x slower one)."""
    if self.optimizer is None:
      raise RuntimeError('compile(optimizer=...) first')
    lv, pending, y_true, sample = self._train_step_once(data)
    if lv[2] > 0:                                    # tripped on some replica -> on all of them after the SUM
      self.train_guard_trips += 1
      self.optimizer.iterations -= 1
      self._drop_step -= 1 if self.dropout > 0 else 0
      with self.exact_fp32():
        lv, pending, y_true, sample = self._train_step_once(data)
    for m, v in zip(pending, lv[3:]):
      m.commit(v)
    for metric in self.metrics:
      if metric.name == 'loss':
        metric.update_state(lv[0])
      elif metric.name == 'reg_loss':
        metric.update_state(lv[1])
      elif not hasattr(metric, 'update_state_device'):
        metric.update_state(y_true, sample)
    return {m.name: m.result() for m in self.metrics}

  def _mirror(self, vec):
    """Queues the device-to-host copy of the step's scalars into pinned memory and records the event the host waits on."""
    if self._log_mirror is None or self._log_mirror.numel() != vec.numel():
      self._log_mirror = torch.empty(vec.numel(), dtype=torch.float32, pin_memory=True)
      self._log_event = torch.cuda.Event()
    self._log_mirror.copy_(vec, non_blocking=True)
    self._log_event.record()

  def _train_step_once(self, data):
    """One pass of the step.  The scalars a step reports -- {loss, reg_loss, range flag, device-reduced metrics} -- live in
    the gradient bucket's tail, each already scaled so that the SUM over the replicas is the reported value.

    * single replica (no process group; ``early_logs`` True): they are final right after the loss kernels, so their
      pinned copy is queued THERE, between the forward and the backward half, and the host waits for that copy only: it
      returns with ~4 ms of the step still queued and has the next step's launches out before the GPU runs dry.
    * data parallel (a process group exists; ``early_logs`` False): ONE collective per step -- the bucket's SUM
      all-reduce carries gradients and tail together (train.py:203, src/model.py:328-336); the pinned copy of the reduced
      tail is queued right behind it, the optimizer behind that, and the host waits for the copy's event (no blocking
      ``tolist()`` at the end of the step).  ``early_logs = True`` under a process group instead all-reduces a copy of
      the few scalars between forward and backward (a second, tiny collective) and reads them early."""
    from . import dp
    if not self.built:
      self._split_inputs(data)                          # Keras builds on the first call (the condition width fixes the shapes)
    compiled = self._metrics_from_compilation
    want_metric = len(compiled) > 0
    dev_metrics = [m for m in compiled if hasattr(m, 'update_state_device')][:self._TAIL_METRICS]
    early = self.early_logs if self.early_logs is not None else not dp.initialized()
    world = self._world()
    n = self.flat_params.numel()
    tail = self._grad_bucket[n:]
    nt = 3 + len(dev_metrics)

    def queue_metrics(sample, y_true):
      for i, m in enumerate(dev_metrics):
        m.update_state_device(y_true, sample, out=tail[3 + i:4 + i], scale=1.0 / world)

    if early:
      def between(loss, sample, y_true):
        queue_metrics(sample, y_true)
        vec = tail[:nt]
        if dp.initialized():
          vec = vec.clone()
          dp.allreduce_bucket(vec)
        self._mirror(vec)
      loss, sample, y_true = self.loss_and_grads(data, want_sample=want_metric, _loss_in_bucket=True, _between=between)
      dp.allreduce_bucket(self._grad_bucket)            # gradients + tail; no-op without a process group
      self.optimizer.apply_gradients(self, skip_flag=tail[2:3])
    else:
      loss, sample, y_true = self.loss_and_grads(data, want_sample=want_metric, _loss_in_bucket=True)
      queue_metrics(sample, y_true)
      dp.allreduce_bucket(self._grad_bucket)            # the step's ONE collective: gradients + {loss, reg_loss, flag, metrics}
      self._mirror(tail[:nt])
      self.optimizer.apply_gradients(self, skip_flag=tail[2:3])
    self._log_event.synchronize()
    return self._log_mirror.tolist(), dev_metrics, y_true, sample

  def test_step(self, data):
    """src/model.py:362-391 (range guard as in train_step: a tripped pass is repeated with the exact-fp32 kernels and
    counted in ``test_guard_trips``)."""
    from . import dp
    x, cond = self._split_inputs(data)
    B, T = x.shape[0], x.shape[1] - 1
    world = self._world()
    L = _lib.lib()
    ws = self._workspace('fwd', L.wn_plan_workspace_floats(self._plan, B, T, 0))
    compiled = self._metrics_from_compilation
    want_metric = len(compiled) > 0
    dev_metrics = [m for m in compiled if hasattr(m, 'update_state_device')][:self._TAIL_METRICS]
    vec = torch.zeros(3 + len(dev_metrics), dtype=torch.float32, device=self._device)
    pred = torch.empty(B, T, self.spec.out_channels, dtype=torch.float32, device=self._device) if want_metric else None

    def once():
      _lib.check(L.wn_eval_loss(self._plan, _lib.ptr(self.flat_params), _lib.ptr(x), _lib.ptr(cond), B, T,
                                B * world, _lib.ptr(vec), _lib.ptr(pred), _lib.ptr(ws), ws.numel(),
                                _lib.stream_ptr()))
      sample = self.sample_waveform(pred) if want_metric else None
      # as in train_step: device-side metric reductions into the same vector, ONE collective (loss sums over the replicas;
      # reg_loss unused; flag: any replica; metrics: replica means), ONE device-to-host read
      for i, m in enumerate(dev_metrics):
        m.update_state_device(x[:, 1:, :], sample, out=vec[3 + i:4 + i], scale=1.0 / world)
      dp.allreduce_bucket(vec)
      return vec.tolist(), sample
    lv, sample = once()
    if lv[2] > 0:
      self.test_guard_trips += 1
      with self.exact_fp32():
        lv, sample = once()
    for m, v in zip(dev_metrics, lv[3:]):
      m.commit(v)
    for metric in self.metrics:
      if metric.name == 'loss':
        metric.update_state(lv[0])
      elif metric.name == 'reg_loss':
        continue
      elif not hasattr(metric, 'update_state_device'):
        metric.update_state(x[:, 1:, :], sample)
    return {m.name: m.result() for m in self.metrics if m.name != 'reg_loss'}

  # ------------------------------------------------------------------ sampling / loss
  def prepare_target(self, x):
    """src/model.py:151-155: Discretization (bit-exact bucket indices) or identity."""
    if self.num_mixtures is not None:
      return x
    x = torch.as_tensor(x, dtype=torch.float32, device=self._device).contiguous()
    idx = torch.empty(x.shape, dtype=torch.int32, device=self._device)
    _lib.check(_lib.lib().wn_quantize(_lib.ptr(x), _lib.ptr(idx), x.numel(), self.bits, _lib.stream_ptr()))
    return idx

  def sample_waveform(self, inputs, deterministic=False):
    """src/model.py:393-503: (B,T,C_out) -> (B,T,1).  Stochastic draws use Philox keyed by the
    reference's seed (4,2) -> 0x0402 plus a per-call offset (TF's stream is not reproducible)."""
    pred = torch.as_tensor(inputs, dtype=torch.float32, device=self._device).contiguous()
    if pred.dim() != 3:
      raise ValueError('prediction must have shape (batch, samples, channels)')
    B, T, Cc = pred.shape
    out = torch.empty(B, T, 1, dtype=torch.float32, device=self._device)
    self._sample_calls += 1
    _lib.check(_lib.lib().wn_sample_waveform(_lib.HEADS[self.sampling_function], _lib.ptr(pred), B * T, Cc,
                                              self.num_mixtures or 0, self.bits, int(bool(deterministic)),
                                              0x0402, self._sample_calls, _lib.ptr(out), _lib.stream_ptr()))
    return out

  def loss_fn(self, target, pred):
    """src/model.py:505-551: per-(b,t) loss, shape (B,T)."""
    pred = torch.as_tensor(pred, dtype=torch.float32, device=self._device).contiguous()
    B, T, Cc = pred.shape
    if self.sampling_function == 'categorical':
      tgt = torch.as_tensor(target, device=self._device).to(torch.int32).contiguous()
    else:
      tgt = torch.as_tensor(target, dtype=torch.float32, device=self._device).contiguous()
    if tgt.numel() != B * T:
      raise ValueError('target must have shape (batch, samples, 1)')
    out = torch.empty(B, T, dtype=torch.float32, device=self._device)
    _lib.check(_lib.lib().wn_loss_fn(_lib.HEADS[self.sampling_function], _lib.ptr(tgt), _lib.ptr(pred), B * T, Cc,
                                      self.num_mixtures or 0, self.bits, _lib.ptr(out), _lib.stream_ptr()))
    return out

  # ------------------------------------------------------------------ generation
  def generate(self, length, batch_size: int = 1, condition=None, sample=None,
               use_queues=False, deterministic=False):
    """src/model.py:258-307 (intended semantics; the reference's kwarg / rank bugs are not
    reproduced).  Returns (B, length, 1)."""
    if self.conditioning is not None and condition is None:
      raise ValueError('Conditioning must be provided.')
    if condition is not None:
      condition = torch.as_tensor(condition, dtype=torch.float32, device=self._device).contiguous()
    if sample is not None:
      sample = torch.as_tensor(sample, dtype=torch.float32, device=self._device).contiguous()
    if condition is not None and sample is not None:
      if condition.shape[0] != sample.shape[0]:
        raise ValueError('Condition and sample must have same batch size.')
    if condition is not None:
      batch_size = condition.shape[0]
    if sample is not None:
      batch_size = sample.shape[0]
    rf = self.receptive_field
    if sample is None:
      if deterministic:
        sample = torch.zeros(batch_size, rf, 1, device=self._device)
      else:
        g = torch.Generator(device='cpu').manual_seed(0x0402)
        sample = torch.randn(batch_size, rf, 1, generator=g).to(self._device)
    if sample.shape[1] != rf:
      if sample.shape[1] < rf:
        raise ValueError('sample must hold at least receptive_field samples')
      sample = sample[:, -rf:, :].contiguous()
    if not self.built:
      self.build([tuple(sample.shape), tuple(condition.shape)] if condition is not None else tuple(sample.shape))
    L = _lib.lib()
    ws = self._workspace('gen', L.wn_generate_workspace_floats(self._plan, batch_size, int(bool(use_queues))))
    out = torch.empty(batch_size, int(length), 1, dtype=torch.float32, device=self._device)
    def run():
      _lib.check(L.wn_generate(self._plan, _lib.ptr(self.flat_params), _lib.ptr(sample), _lib.ptr(condition),
                               batch_size, int(length), int(bool(deterministic)), int(bool(use_queues)), 0x0402,
                               _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    run()
    # range guard (one host read; the caller reads the samples anyway): an activation beyond the fp16 range of the
    # split-precision kernels -> the whole call again on the exact-fp32 kernels (same seed: same draws)
    if length > 0 and L.wn_debug_value(1) != 1:
      slot = L.wn_generate_guard_slot(self._plan, batch_size, int(bool(use_queues)))
      guard, watchdog = ws[slot:slot + 2].cpu()
      if int(watchdog.view(torch.int32)) != 0:
        raise RuntimeError(f'wn_generate: a workgroup of the generation relay gave up waiting for its predecessor '
                           f'(code {int(watchdog.view(torch.int32))}); the samples of this call are invalid')
      if not (float(guard) < L.wn_range_limit()):
        self.generation_guard_trips += 1
        with self.exact_fp32():
          run()
    return out

  def compute_receptive_field(self, sampling_frequency):
    """src/model.py:553-556."""
    return self.receptive_field / sampling_frequency
