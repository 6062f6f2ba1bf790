"""Elementwise pieces of the boundary as free functions on CUDA tensors.

mu_law / inverse_mu_law: src/utils.py:34-35, src/callbacks.py:126-131;
quantize / dequantize: src/model.py:151-153, :411,418.
"""
from __future__ import annotations

import torch

from . import _lib


def _f32(x):
  if not x.is_cuda:
    raise ValueError('wavenets_amd.ops work on GPU tensors only (no CPU fallback)')
  return x.to(torch.float32).contiguous()


def mu_law(x: torch.Tensor) -> torch.Tensor:
  x = _f32(x)
  y = torch.empty_like(x)
  _lib.check(_lib.lib().wn_mulaw(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_ptr()))
  return y


def inverse_mu_law(y: torch.Tensor) -> torch.Tensor:
  y = _f32(y)
  x = torch.empty_like(y)
  _lib.check(_lib.lib().wn_inv_mulaw(_lib.ptr(y), _lib.ptr(x), y.numel(), _lib.stream_ptr()))
  return x


def quantize(x: torch.Tensor, bits: int = 8) -> torch.Tensor:
  x = _f32(x)
  idx = torch.empty(x.shape, dtype=torch.int32, device=x.device)
  _lib.check(_lib.lib().wn_quantize(_lib.ptr(x), _lib.ptr(idx), x.numel(), bits, _lib.stream_ptr()))
  return idx


def dequantize(idx: torch.Tensor, bits: int = 8) -> torch.Tensor:
  if not idx.is_cuda:
    raise ValueError('wavenets_amd.ops work on GPU tensors only (no CPU fallback)')
  idx = idx.to(torch.int32).contiguous()
  x = torch.empty(idx.shape, dtype=torch.float32, device=idx.device)
  _lib.check(_lib.lib().wn_dequantize(_lib.ptr(idx), _lib.ptr(x), idx.numel(), bits, _lib.stream_ptr()))
  return x
