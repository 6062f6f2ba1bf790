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
