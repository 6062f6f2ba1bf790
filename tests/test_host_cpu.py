"""Host-side pieces (no GPU): framing / validity filter (src/utils.py:36-38,57-70), checkpoint naming
and resume parsing (train.py:68-86,149-154), training policies (train.py:167-176)."""
import math
import os

import numpy as np
import torch

from wavenets_amd import callbacks, data, io


def test_frame_matches_tf_signal_frame_semantics():
  x = torch.arange(25, dtype=torch.float32)
  f = data.frame(x, 8)                     # frame_length 9, hop 8, no end padding
  assert f.shape == (3, 9, 1)
  assert f[0, :, 0].tolist() == list(range(0, 9))
  assert f[1, :, 0].tolist() == list(range(8, 17))      # consecutive frames overlap by one sample
  assert f[2, :, 0].tolist() == list(range(16, 25))
  assert data.frame(torch.zeros(5), 8).shape == (0, 9, 1)


def test_preprocess_filters_invalid_frames():
  x = torch.linspace(-0.5, 0.5, 41)
  x[12] = 1.5                               # outside [-1, 1] -> the frames holding it are dropped
  f = data.preprocess_waveform(x, 10, apply_mulaw=False)
  assert f.shape == (3, 11, 1)              # 4 frames, frame 1 (samples 10..20) is dropped
  x2 = torch.linspace(-0.5, 0.5, 41); x2[25] = float('nan')
  assert data.preprocess_waveform(x2, 10, False).shape[0] == 3
  y = data.preprocess_waveform(torch.tensor([0.0, 1.0, -1.0, 0.5] * 6), 5, apply_mulaw=True)
  assert y.abs().max() <= 1.0 and abs(y[0, 1, 0].item() - 1.0) < 1e-6
  assert (data.normalise_int16(torch.tensor([-32768, 16384], dtype=torch.int16)) == torch.tensor([-1.0, 0.5])).all()


def test_checkpoint_naming_and_resume(tmp_path):
  assert io.checkpoint_name(7, 0.0005) == 'weights-e0007-lr0.0005.weights.npz'
  assert io.find_resume(str(tmp_path / 'missing')) is None
  for e, lr in ((1, 0.0005), (12, 0.0001), (3, 0.0005)):
    (tmp_path / io.checkpoint_name(e, lr)).write_bytes(b'')
  f, epoch, lr = io.find_resume(str(tmp_path))
  assert os.path.basename(f) == 'weights-e0012-lr0.0001.weights.npz' and epoch == 12 and lr == 0.0001


class _Opt:
  learning_rate = 1e-3


def test_training_policies():
  opt = _Opt()
  r = callbacks.ReduceLROnPlateau(factor=0.2, patience=2, min_lr=2e-8, min_delta=10)
  assert not r.on_epoch_end(1000.0, opt)
  assert not r.on_epoch_end(995.0, opt)     # improvement < min_delta: wait = 1
  assert r.on_epoch_end(994.0, opt)         # wait = 2 -> reduce
  assert abs(opt.learning_rate - 2e-4) < 1e-12
  t = callbacks.TerminateOnNaN()
  assert t.on_batch_end(float('nan')) and t.on_batch_end(math.inf) and not t.on_batch_end(3.0)

  class M:
    class P:
      data = torch.zeros(3)
    flat_params = P()
  m = M()
  e = callbacks.EarlyStopping(patience=2, min_delta=10)
  m.flat_params.data = torch.ones(3)
  assert not e.on_epoch_end(100.0, m)
  m.flat_params.data = torch.full((3,), 2.0)
  assert not e.on_epoch_end(95.0, m)
  assert e.on_epoch_end(96.0, m)            # stop, best weights restored
  assert torch.equal(m.flat_params.data, torch.ones(3))
