"""Keras-semantics Adam with per-tensor clipnorm, fused on the GPU (train.py:225-226).

``Adam(learning_rate=5e-4, clipnorm=1.0)`` mirrors ``tf.keras.optimizers.Adam``: epsilon
(1e-7) is added to sqrt(v) outside the bias correction, each gradient tensor is clipped to
``clipnorm`` independently (tf.clip_by_norm) before the moment update.
"""
from __future__ import annotations

import torch

from . import _lib


class Adam:
  def __init__(self, learning_rate: float = 0.001, beta_1: float = 0.9, beta_2: float = 0.999,
               epsilon: float = 1e-7, clipnorm=None):
    self.learning_rate = float(learning_rate)
    self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
    self.clipnorm = clipnorm
    self.iterations = 0
    self.m = self.v = self._scratch = None

  def build(self, model):
    """Allocate the moment buffers (src/model.py:211 optimizer.build)."""
    if self.m is None or self.m.numel() != model.flat_params.numel():
      self.m = torch.zeros_like(model.flat_params.data)
      self.v = torch.zeros_like(model.flat_params.data)
      self._scratch = torch.zeros(len(model.variable_names) + 8, dtype=torch.float32,
                                  device=model.flat_params.device)

  def apply_gradients(self, model, skip_flag=None):
    """One update of model.flat_params from model.flat_grads (src/model.py:336).  ``skip_flag``: a device float;
    when it is non-zero the kernel leaves parameters and moments untouched (the range guard of the split-precision
    mode tripped and the caller repeats the step with the exact-fp32 kernels, see WaveNet.train_step)."""
    self.build(model)
    self.iterations += 1
    _lib.check(_lib.lib().wn_adam_step_guarded(
        model._plan, _lib.ptr(model.flat_params), _lib.ptr(model.flat_grads), _lib.ptr(self.m),
        _lib.ptr(self.v), self.iterations, self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
        float(self.clipnorm) if self.clipnorm else 0.0, _lib.ptr(self._scratch), _lib.ptr(skip_flag),
        _lib.stream_ptr()))
