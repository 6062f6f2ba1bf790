"""Host-side training policies the reference takes from Keras (train.py:167-176): same defaults."""
from __future__ import annotations

import math


class ReduceLROnPlateau:
  """tf.keras.callbacks.ReduceLROnPlateau(monitor='loss', factor=0.2, patience=5, min_lr=2e-8, min_delta=10)."""

  def __init__(self, factor=0.2, patience=5, min_lr=2e-8, min_delta=10.0):
    self.factor, self.patience, self.min_lr, self.min_delta = factor, patience, min_lr, min_delta
    self.best, self.wait = math.inf, 0

  def on_epoch_end(self, loss, optimizer) -> bool:
    if loss < self.best - self.min_delta:
      self.best, self.wait = loss, 0
      return False
    self.wait += 1
    if self.wait >= self.patience and optimizer.learning_rate > self.min_lr:
      optimizer.learning_rate = max(optimizer.learning_rate * self.factor, self.min_lr)
      self.wait = 0
      return True
    return False


class EarlyStopping:
  """EarlyStopping(monitor='loss', patience=15, min_delta=10, restore_best_weights=True)."""

  def __init__(self, patience=15, min_delta=10.0, restore_best_weights=True):
    self.patience, self.min_delta, self.restore = patience, min_delta, restore_best_weights
    self.best, self.wait, self.best_weights = math.inf, 0, None

  def on_epoch_end(self, loss, model) -> bool:
    """True = stop training."""
    if loss < self.best - self.min_delta:
      self.best, self.wait = loss, 0
      if self.restore:
        self.best_weights = model.flat_params.data.clone()
      return False
    self.wait += 1
    if self.wait >= self.patience:
      if self.restore and self.best_weights is not None:
        model.flat_params.data.copy_(self.best_weights)
      return True
    return False


class TerminateOnNaN:
  def on_batch_end(self, loss) -> bool:
    return not math.isfinite(loss)


# --------------------------------------------------------------------------------------------
# Observability (src/callbacks.py:4-159, train.py:253-270).  TensorBoard is replaced by plain files:
# <log_dir>/epoch_<e>/<key>_<i>.wav, <key>.npy (waveforms) and <key>_spectrogram.npy.
# --------------------------------------------------------------------------------------------
class AddLRToLogs:
  """src/callbacks.py:120-124."""

  def on_epoch_end(self, logs, optimizer):
    logs.update({'lr': float(optimizer.learning_rate)})
    return logs


def inverse_mu_law(y):
  """sign(y) * (256^|y| - 1) / 255 (src/callbacks.py:126-131); runs in libwn_hip.so."""
  from . import ops
  return ops.inverse_mu_law(y)


def create_spectrogram(data, sample_rate=None):
  """Log-magnitude STFT image batch (src/callbacks.py:133-159): frames of 256 samples, hop 128, periodic
  Hann window, 256-point real FFT (tf.signal.stft defaults), log(|.| + 1e-5), laid out
  (batch, 129 bins, frames, 1) and min-max scaled over the whole batch.  Host-side (numpy): logging
  is not on the hot path.  Returns a numpy array."""
  import numpy as np
  del sample_rate
  x = np.asarray(data.detach().cpu() if hasattr(data, 'detach') else data, dtype=np.float32)
  x = np.squeeze(x)
  if x.ndim == 1:
    x = x[None, :]
  n_frames = 1 + (x.shape[-1] - 256) // 128 if x.shape[-1] >= 256 else 0
  idx = np.arange(256)[None, :] + 128 * np.arange(n_frames)[:, None]
  window = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(256) / 256.0)).astype(np.float32)
  frames = x[:, idx] * window                                   # (B, frames, 256)
  spec = np.log(np.abs(np.fft.rfft(frames, n=256, axis=-1)).astype(np.float32) + 1e-5)
  spec = np.transpose(spec[..., None], (0, 2, 1, 3))            # (B, 129, frames, 1)
  spec = spec - spec.min()
  return spec / spec.max()


def write_wav(path, waveform, sampling_frequency):
  """16-bit PCM wav of one (T,) or (T, 1) float waveform in [-1, 1] (tf.audio.encode_wav, train.py:268-270)."""
  import numpy as np
  import wave
  x = np.asarray(waveform.detach().cpu() if hasattr(waveform, 'detach') else waveform, dtype=np.float32).reshape(-1)
  pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype('<i2')
  with wave.open(path, 'wb') as f:
    f.setnchannels(1)
    f.setsampwidth(2)
    f.setframerate(int(sampling_frequency))
    f.writeframes(pcm.tobytes())


class SoundCallback:
  """src/callbacks.py:4-118: generate at the end of every `epoch_frequency`-th epoch, from noise and (if
  given) from an initial sample; `use_fast='both'` produces the queued and the sliding-window result
  side by side (the reference's A/B hook for its queue TODO)."""

  def __init__(self, log_dir, sampling_frequency: int, samples: int, apply_mulaw: bool, epoch_frequency: int = 1,
               condition=None, use_fast=False, initial_sample=None, model=None):
    if use_fast not in ['both', True, False]:
      raise ValueError('use_fast must be one of True, False, "both"')
    if epoch_frequency < 1:
      raise ValueError('epoch_frequency must be greater than 0')
    self.log_dir = log_dir
    self.sampling_frequency = sampling_frequency
    self.log_freq = epoch_frequency
    self.samples = samples
    self.condition = condition
    self.apply_mulaw = apply_mulaw
    self.initial_sample = initial_sample
    self.use_fast = use_fast
    self.model = model

  def set_model(self, model):
    self.model = model

  def on_epoch_end(self, epoch, logs=None):
    import os
    import numpy as np
    del logs
    if epoch % self.log_freq != self.log_freq - 1:
      return None
    modes = [('fast', True), ('standard', False)] if self.use_fast == 'both' else [('standard', self.use_fast)]
    generated = {}
    for key, queued in modes:
      generated[key] = self.model.generate(self.samples, batch_size=5, condition=self.condition, use_queues=queued)
    if self.initial_sample is not None:
      if self.condition is not None:
        wave_, cond = self.initial_sample
        wave_, cond = wave_[:8, :, :], cond[:8, :]
      else:
        wave_, cond = self.initial_sample[:8, :, :], None
      for key, queued in modes:
        name = 'with_initial' + ('_fast' if (key == 'fast') else '')
        generated[name] = self.model.generate(self.samples, batch_size=5, condition=cond, sample=wave_,
                                              use_queues=queued)
    out_dir = os.path.join(self.log_dir, f'epoch_{epoch:04d}')
    os.makedirs(out_dir, exist_ok=True)
    for key, batch in generated.items():
      if self.apply_mulaw:
        batch = inverse_mu_law(batch)
      np.save(os.path.join(out_dir, f'generated_{key}.npy'), batch.detach().cpu().numpy())
      np.save(os.path.join(out_dir, f'generated_spectrogram_{key}.npy'), create_spectrogram(batch, self.sampling_frequency))
      for i in range(min(8, batch.shape[0])):
        write_wav(os.path.join(out_dir, f'generated_{key}_{i}.wav'), batch[i], self.sampling_frequency)
    return generated
