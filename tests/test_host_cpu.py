"""Host-side pieces (no GPU): framing / validity filter (src/utils.py:36-38,57-70), checkpoint naming
and resume parsing (train.py:68-86,149-154), training policies (train.py:167-176)."""
import math
import os

import numpy as np
import pytest
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
  import numpy as np
  for e, lr in ((1, 0.0005), (12, 0.0001), (3, 0.0005)):
    np.savez(str(tmp_path / io.checkpoint_name(e, lr)), names=np.array(['causal/kernel']))
  # a write that was killed half-way sorts last but does not open: resume must skip it, not crash
  (tmp_path / io.checkpoint_name(13, 0.0001)).write_bytes(b'PK\x03\x04 truncated')
  (tmp_path / io.checkpoint_name(14, 0.0001)).write_bytes(b'')
  f, epoch, lr = io.find_resume(str(tmp_path))
  assert os.path.basename(f) == 'weights-e0012-lr0.0001.weights.npz' and epoch == 12 and lr == 0.0001


class _FakeModel:
  variable_names = ['causal/kernel', 'causal/bias']
  dropout = 0.1
  _drop_step = 41

  def __init__(self):
    import numpy as np
    self.w = [np.arange(6, dtype=np.float32).reshape(2, 1, 3), np.ones(3, np.float32)]

  def get_weights(self):
    return [w.copy() for w in self.w]

  def set_weights(self, ws):
    self.w = [w.copy() for w in ws]

  def set_drop_step(self, n):
    self._drop_step = n


def test_save_weights_is_atomic_and_round_trips(tmp_path):
  import numpy as np
  a, b = _FakeModel(), _FakeModel()
  b.w = [w * 0 for w in b.w]
  b._drop_step = 0
  path = str(tmp_path / io.checkpoint_name(2, 0.0005))
  io.save_weights(a, path)
  assert sorted(os.listdir(tmp_path)) == [io.checkpoint_name(2, 0.0005)]       # no temp file left behind
  io.load_weights(b, path)
  assert all(np.array_equal(x, y) for x, y in zip(a.w, b.w)) and b._drop_step == 41
  b.variable_names = ['other']
  import pytest
  with pytest.raises(ValueError):
    io.load_weights(b, path)


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


def test_create_spectrogram_matches_definition():
  """src/callbacks.py:133-159: frame 256 / hop 128 / periodic Hann / log(|.|+1e-5), (B,129,frames,1), min-max."""
  import numpy as np
  from wavenets_amd.callbacks import create_spectrogram
  rng = np.random.default_rng(0)
  x = rng.standard_normal((3, 1000, 1)).astype(np.float32) * 0.1
  got = create_spectrogram(torch.from_numpy(x))
  frames = 1 + (1000 - 256) // 128
  assert got.shape == (3, 129, frames, 1)
  assert got.min() == 0.0 and abs(got.max() - 1.0) < 1e-6
  # direct DFT of one frame
  b, f, k = 1, 2, 17
  w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(256) / 256)
  seg = x[b, 128 * f:128 * f + 256, 0] * w
  mag = abs(np.sum(seg * np.exp(-2j * np.pi * k * np.arange(256) / 256)))
  # undo the min-max scaling with two reference points
  raw = np.log(mag + 1e-5)
  allraw = []
  for bb in range(3):
    for ff in range(frames):
      s = x[bb, 128 * ff:128 * ff + 256, 0] * w
      allraw.append(np.log(np.abs(np.fft.rfft(s)) + 1e-5))
  allraw = np.array(allraw)
  want = (raw - allraw.min()) / (allraw.max() - allraw.min())
  assert abs(got[b, k, f, 0] - want) < 1e-4


def test_write_wav_roundtrip(tmp_path):
  import numpy as np
  import wave
  from wavenets_amd.callbacks import write_wav
  x = np.sin(np.linspace(0, 20, 800)).astype(np.float32) * 0.5
  p = str(tmp_path / 'a.wav')
  write_wav(p, torch.from_numpy(x)[:, None], 16000)
  with wave.open(p, 'rb') as f:
    assert f.getframerate() == 16000 and f.getnchannels() == 1 and f.getsampwidth() == 2
    pcm = np.frombuffer(f.readframes(f.getnframes()), dtype='<i2')
  assert pcm.shape[0] == 800 and np.abs(pcm / 32768.0 - x).max() < 1e-4


def test_sound_callback_argument_checks():
  from wavenets_amd.callbacks import SoundCallback
  with pytest.raises(ValueError, match='use_fast'):
    SoundCallback('x', 16000, 10, True, use_fast='maybe')
  with pytest.raises(ValueError, match='epoch_frequency'):
    SoundCallback('x', 16000, 10, True, epoch_frequency=0)


def test_condition_frames_follow_the_utterance_label():
  """src/utils.py:42-50: every frame of an utterance carries its one-hot class; the validity filter drops frames, and
  their condition rows with them."""
  import torch
  x = torch.linspace(-0.9, 0.9, 1000)
  x[450] = 1.5                                       # one frame leaves [-1, 1] -> filtered out
  frames, cond = data.preprocess_with_condition(x, 2, 4, 99, False)
  assert frames.shape == (9, 100, 1) and cond.shape == (9, 4)
  assert torch.equal(cond, torch.tensor([0.0, 0.0, 1.0, 0.0]).expand(9, 4))
  import pytest
  with pytest.raises(ValueError):
    data.preprocess_with_condition(x, 4, 4, 99, False)


def test_generation_chain_prefetch_registers_are_untouched_until_their_wait():
  """wn_gen.hip fetches weights with inline-asm loads and hand-placed s_waitcnt vmcnt(N) (the compiler's own placement
  drains vmcnt at every use in a loop).  Nothing tells the compiler those registers are in flight, so a phi copy or a
  reused temporary before the wait would read or clobber them: tools/check_gen_isa.py walks the kernel's control-flow
  graph and must find no such instruction."""
  import importlib.util, os, shutil
  if not os.path.exists('/opt/rocm/bin/hipcc'):
    pytest.skip('no hipcc')
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_gen_isa.py')
  spec = importlib.util.spec_from_file_location('check_gen_isa', path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  asm = mod.compile_asm()
  lines = open(asm).read().split('\n')
  import re
  kernels, i = 0, 0
  while i < len(lines):
    m = re.match(r'^(_Z\d+wn_gen_chain3_kernel\w+):', lines[i])
    if m:
      j = i
      while not lines[j].startswith('.Lfunc_end'):
        j += 1
      nloads, bad = mod.check_kernel(m.group(1), lines[i + 1:j])
      assert nloads > 0 and bad == 0, (m.group(1), nloads, bad)
      kernels += 1
      i = j
    i += 1
  assert kernels >= 3
