"""WaveNet residual block: host-side mirror of src/layers.py::WaveNetLayer.

Same constructor keywords, ``build`` / ``compute_output_shape`` / ``call`` / ``generate``
methods, return tuples and exceptions as the reference layer; the arithmetic runs in
``libwn_hip.so`` (``wn_layer_fwd`` / ``wn_layer_bwd``).  ``call`` is differentiable through a
``torch.autograd.Function`` whose backward is the hand-written HIP backward.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from . import spec as _spec


class _LayerFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, layer, x, cond, flat):
    B, T = x.shape[0], x.shape[1]
    d = layer._desc
    L = _lib.lib()
    saved = torch.empty(int(L.wn_layer_saved_floats(C.byref(d), B, T)), dtype=torch.float32, device=x.device)
    ws = layer._workspace(B, T)
    x_out = torch.empty(B, T, layer.channels, dtype=torch.float32, device=x.device)
    skip = torch.empty(B, T, layer.skip_channels or layer.channels, dtype=torch.float32, device=x.device)
    _lib.check(L.wn_layer_fwd(C.byref(d), _lib.ptr(flat), _lib.ptr(x), _lib.ptr(cond), B, T, _lib.ptr(x_out),
                              _lib.ptr(skip), _lib.ptr(saved), _lib.ptr(ws), _lib.stream_ptr()))
    ctx.layer = layer
    ctx.save_for_backward(x, cond if cond is not None else torch.empty(0, device=x.device), flat, saved)
    ctx.has_cond = cond is not None
    return x_out, skip

  @staticmethod
  def backward(ctx, g_xout, g_skip):
    layer = ctx.layer
    x, cond, flat, saved = ctx.saved_tensors
    cond = cond if ctx.has_cond else None
    B, T = x.shape[0], x.shape[1]
    d = layer._desc
    L = _lib.lib()
    ws = layer._workspace(B, T)
    g_xout = g_xout.contiguous() if g_xout is not None else None
    g_skip = g_skip.contiguous() if g_skip is not None else None
    g_x = torch.empty_like(x)
    g_cond = torch.empty_like(cond) if cond is not None else None
    g_flat = torch.zeros_like(flat)
    _lib.check(L.wn_layer_bwd(C.byref(d), _lib.ptr(flat), _lib.ptr(x), _lib.ptr(cond), _lib.ptr(saved),
                              _lib.ptr(g_xout), _lib.ptr(g_skip), B, T, _lib.ptr(g_x), _lib.ptr(g_cond),
                              _lib.ptr(g_flat), _lib.ptr(ws), _lib.stream_ptr()))
    return None, g_x, g_cond, g_flat


class _Conv:
  """``.kernel`` / ``.bias`` / ``.weights`` of one conv inside the flat layer buffer."""

  def __init__(self, layer, koff, kshape, boff):
    self._layer, self._koff, self._kshape, self._boff = layer, koff, kshape, boff

  @property
  def kernel(self):
    f = self._layer._flat()
    return f[self._koff:self._koff + int(np.prod(self._kshape))].view(*self._kshape)

  @property
  def bias(self):
    f = self._layer._flat()
    return f[self._boff:self._boff + self._kshape[-1]]

  @property
  def weights(self):
    return [self.kernel, self.bias]


class WaveNetLayer(torch.nn.Module):
  """WaveNet layer (src/layers.py:4-290).

  As in the reference, the layer is agnostic to global vs local conditioning: when
  ``condition`` is True the input is a tuple ``(x, cond)`` with ``cond`` of shape
  ``(batch, samples, cond_channels)``.
  """

  def __init__(self, kernel=2,
               dilation_rate=1,
               activation=None,
               channels=32,
               residual=True,
               dilation_channels=None,
               skip_channels=None,
               l2_reg_factor=None,
               condition=False,
               dropout=0,
               device=None,
               seed: int = 0,
               _owner=None):
    super().__init__()
    if activation not in _spec.SUPPORTED_ACTIVATIONS:
      raise NotImplementedError(f'activation {activation!r} is not supported')
    if dilation_channels is None:                      # src/layers.py:49-50
      dilation_channels = channels
    if not isinstance(dilation_rate, list):            # src/layers.py:52-53
      dilation_rate = [dilation_rate]
    self.input_dilation = dilation_rate[0]
    self.depth = len(dilation_rate)
    self.dilation_rates = list(dilation_rate)
    self.kernel_size = kernel
    self.channels = channels
    self.residual = residual
    self.dilation_channels = dilation_channels
    self.skip_channels = skip_channels
    self.activation = activation
    self.condition = condition
    self.l2_reg_factor = 0 if l2_reg_factor is None else l2_reg_factor
    self.dropout_rate = dropout
    self.dropout = torch.nn.Dropout(dropout) if dropout > 0 else None
    self._device = torch.device(device) if device is not None else torch.device('cuda', 0)
    self._seed = seed
    self._owner = _owner
    self._ws = None
    self.built = False
    self._output_shape = None
    self._desc = None
    self.flat_params = None

  # ------------------------------------------------------------------ structure
  def _make_desc(self, in_channels, cond_channels):
    if self.depth > 16:
      raise NotImplementedError('at most 16 dilated convs per block')
    d = _lib.WnLayerDesc()
    d.kernel_size, d.channels, d.dilation_channels = self.kernel_size, self.channels, self.dilation_channels
    d.skip_channels = self.skip_channels or 0
    d.depth = self.depth
    for i, r in enumerate(self.dilation_rates):
      d.dilations[i] = int(r)
    d.activation = _lib.ACTIVATIONS[self.activation]
    d.residual = int(bool(self.residual))
    d.cond_channels = cond_channels
    d.in_channels = in_channels
    return d

  def _param_layout(self, in_channels, cond_channels):
    k, R, D, S = self.kernel_size, self.channels, self.dilation_channels, self.skip_channels
    out, off, cin = [], 0, in_channels
    for i in range(self.depth):
      cout = 2 * D if i == self.depth - 1 else D
      out.append((f'dil{i}', off, (k, cin, cout), off + k * cin * cout))
      off += k * cin * cout + cout
      cin = cout
    out.append(('conv1', off, (1, D, R), off + D * R)); off += D * R + R
    if S is not None:
      out.append(('conv_skip', off, (1, D, S), off + D * S)); off += D * S + S
    if cond_channels > 0:
      out.append(('conv_cond', off, (1, cond_channels, 2 * D), off + cond_channels * 2 * D))
      off += cond_channels * 2 * D + 2 * D
    return out, off

  def _flat(self):
    if self._owner is not None:
      model, b = self._owner
      names = model.variable_names
      first = names.index(f'block{b}/dil0/kernel')
      start = model._offsets[first]
      return model.flat_params.data[start:start + self._nparams]
    return self.flat_params.data

  def build(self, input_shape):
    """src/layers.py:122-164: shape checks + parameter creation."""
    if self.condition:
      x_shape, cond_shape = input_shape
      if x_shape[1] != cond_shape[1]:
        raise ValueError('Condition tensor must have the same length as input')
      cond_channels = int(cond_shape[-1])
    else:
      x_shape = input_shape
      cond_channels = 0
    in_channels = int(x_shape[-1])
    if self.residual and in_channels != self.channels:
      raise ValueError('Residual connection must have the same shape as input')
    self._desc = self._make_desc(in_channels, cond_channels)
    layout, total = self._param_layout(in_channels, cond_channels)
    assert total == _lib.lib().wn_layer_param_count(C.byref(self._desc))
    self._nparams = total
    self.dilated_stack, self.conv_skip, self.conv_cond = [], None, None
    for name, koff, kshape, boff in layout:
      h = _Conv(self, koff, kshape, boff)
      if name.startswith('dil'):
        self.dilated_stack.append(h)
      else:
        setattr(self, name, h)
    if self._owner is None:
      g = torch.Generator().manual_seed(self._seed)
      flat = torch.zeros(total, dtype=torch.float32)
      for name, koff, kshape, boff in layout:
        fan_in, fan_out = kshape[0] * kshape[1], kshape[0] * kshape[2]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        cnt = int(np.prod(kshape))
        flat[koff:koff + cnt] = (torch.rand(cnt, generator=g) * 2 - 1) * lim
      self.flat_params = torch.nn.Parameter(flat.to(self._device))
    self.built = True
    x_out_shape = (x_shape[0], x_shape[1], self.channels)
    skip_shape = (x_shape[0], x_shape[1], self.skip_channels or self.channels)
    self._output_shape = (x_out_shape, skip_shape)

  def compute_output_shape(self, input_shape):
    """src/layers.py:166-176."""
    if not self.built:
      raise ValueError('Layer is not built')
    return self._output_shape

  def _workspace(self, B, T):
    need = int(_lib.lib().wn_layer_workspace_floats(C.byref(self._desc), B, T))
    if self._ws is None or self._ws.numel() < need:
      self._ws = torch.empty(need, dtype=torch.float32, device=self._device)
    return self._ws

  # ------------------------------------------------------------------ call
  def call(self, inputs, training=False):
    """src/layers.py:178-224: returns (x_out, skip)."""
    if self.condition:
      x, cond = inputs
      cond = torch.as_tensor(cond, dtype=torch.float32, device=self._device).contiguous()
    else:
      x, cond = inputs, None
    x = torch.as_tensor(x, dtype=torch.float32, device=self._device).contiguous()
    if not self.built:
      self.build([tuple(x.shape), tuple(cond.shape)] if self.condition else tuple(x.shape))
    if self.condition and cond.shape[1] != x.shape[1]:
      raise ValueError('Condition tensor must have the same length as input')
    flat = self.flat_params if self._owner is None else self._flat()
    if training and self.dropout is not None:
      # Dropout on the conv input only, not on the residual (src/layers.py:192-196):
      # block(x) = convpath(dropout(x)) + x  =  fn(x_d) - x_d + x
      x_d = self.dropout(x)
      x_out, skip = _LayerFn.apply(self, x_d, cond, flat)
      if self.residual:
        x_out = x_out - x_d + x
      return x_out, skip
    return _LayerFn.apply(self, x, cond, flat)

  def forward(self, inputs, training=False):
    return self.call(inputs, training=training)

  def generate(self, inputs):
    """src/layers.py:226-290: single-step block for a queued sampler.  ``inputs`` already holds
    the ``kernel`` gathered samples ``[x[t-(k-1)d], ..., x[t]]`` as (B, k, R); the convolution is
    undilated and VALID, the output is one step.  Depth-1 stacks only (README.md:16)."""
    if self.depth != 1:
      raise NotImplementedError('generate() supports depth-1 dilated stacks only')
    if self.condition:
      x, cond = inputs
    else:
      x, cond = inputs, None
    x = torch.as_tensor(x, dtype=torch.float32, device=self._device).contiguous()
    if x.shape[1] != self.kernel_size:
      raise ValueError('generate() expects exactly kernel_size gathered samples')
    if not self.built:
      raise ValueError('Layer is not built')
    saved_rates, saved_desc = self.dilation_rates, self._desc
    try:
      self.dilation_rates = [1]
      self._desc = self._make_desc(saved_desc.in_channels, saved_desc.cond_channels)
      if cond is not None:
        cond = torch.as_tensor(cond, dtype=torch.float32, device=self._device)
        cond = cond.expand(x.shape[0], x.shape[1], cond.shape[-1]).contiguous()
      with torch.no_grad():
        x_out, skip = self.call((x, cond) if self.condition else x)
    finally:
      self.dilation_rates, self._desc = saved_rates, saved_desc
    return x_out[:, -1:, :], skip[:, -1:, :]
