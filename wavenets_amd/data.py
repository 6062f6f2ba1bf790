"""Data step of the reference (src/utils.py:22-85) on GPU tensors: int16 -> float, mu-law companding,
framing into ``recording_length + 1`` windows with hop ``recording_length`` and the validity filter.
Dataset I/O itself (tfds, VCTK) is out of scope; this operates on waveforms already in memory."""
from __future__ import annotations

import math

import torch

from . import ops


def normalise_int16(x: torch.Tensor) -> torch.Tensor:
  """src/utils.py:52-55: speech / 2**15."""
  return x.to(torch.float32) / 2.0 ** 15


def frame(x: torch.Tensor, recording_length: int) -> torch.Tensor:
  """tf.signal.frame(x, frame_length=L+1, frame_step=L, axis=0), no end padding (src/utils.py:36-38).
  x: (N,) or (N, 1) -> (n_frames, L+1, 1)."""
  x = x.reshape(-1)
  L = int(recording_length)
  if x.numel() < L + 1:
    return x.new_zeros((0, L + 1, 1))
  return x.unfold(0, L + 1, L).unsqueeze(-1).contiguous()


def preprocess_waveform(x: torch.Tensor, recording_length: int, apply_mulaw: bool) -> torch.Tensor:
  """convert_and_split + filter of src/utils.py:32-38,57-70: frames that are finite and inside [-1, 1]."""
  x = x.to(torch.float32).reshape(-1)
  if apply_mulaw:
    x = ops.mu_law(x) if x.is_cuda else torch.sign(x) * (torch.log1p(255.0 * x.abs()) / math.log(256.0))
  frames = frame(x, recording_length)
  if frames.shape[0] == 0:
    return frames
  ok = torch.isfinite(frames).all(dim=(1, 2)) & (frames >= -1).all(dim=(1, 2)) & (frames <= 1).all(dim=(1, 2))
  return frames[ok]


def preprocess_with_condition(x: torch.Tensor, label: int, n_classes: int, recording_length: int, apply_mulaw: bool):
  """The ``condition=True`` branch of preprocess_dataset (src/utils.py:42-50,57-63): every frame of an utterance
  carries the utterance's one-hot class (the reference uses one_hot(gender, 2)); the validity filter looks at the
  frame only.  Returns (frames (n, L+1, 1), cond (n, n_classes))."""
  frames = preprocess_waveform(x, recording_length, apply_mulaw)
  if not 0 <= int(label) < n_classes:
    raise ValueError('label outside [0, n_classes)')
  cond = torch.zeros(frames.shape[0], n_classes, dtype=torch.float32, device=frames.device)
  cond[:, int(label)] = 1.0
  return frames, cond


def synthetic_waveforms(n: int, length: int, seed: int = 1234, device=None) -> torch.Tensor:
  """Benchmark input of SURVEY.md section 8d: two-tone + noise at 16 kHz, clipped, mu-law companded."""
  g = torch.Generator().manual_seed(seed)
  t = torch.arange(length, dtype=torch.float64)[None, :]
  f0 = 80.0 + 320.0 * torch.rand((n, 1), generator=g, dtype=torch.float64)
  ph = 2 * math.pi * torch.rand((n, 1), generator=g, dtype=torch.float64)
  x = (0.6 * torch.sin(2 * math.pi * f0 * t / 16000.0 + ph) + 0.2 * torch.sin(2 * math.pi * 3 * f0 * t / 16000.0)
       + 0.02 * torch.randn((n, length), generator=g, dtype=torch.float64)).clamp(-1.0, 1.0)
  x = torch.sign(x) * (torch.log1p(255.0 * x.abs()) / math.log(256.0))
  x = x.to(torch.float32).unsqueeze(-1)
  return x.to(device) if device is not None else x
