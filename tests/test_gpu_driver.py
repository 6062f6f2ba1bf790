"""train.py end to end on the GPU (the host driver of SURVEY.md 8f-3): the reference's YAML keys, global
conditioning as (frames, one-hot) tuples (train.py:98-124, src/utils.py:46-49), best-only checkpoints with the
reference's file-name convention and resume-from-filename (train.py:68-86,149-154), timed generation dump
(train.py:253-270)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = dict(lr=0.002, recording_length=400, batch_size=4, apply_mulaw=True, dataset='synthetic', kernel_size=2, channels=32,
            blocks=4, layers_per_block=1, activation='leaky_relu', dropout=0.1, dilation_bound=16, num_mixtures=None,
            sampling_function='categorical', bits=8, skip_channels=64, final_layers_channels=[32], synthetic_utterances=8,
            preview_length=24)


def _run(tmp_path, cfg, epochs):
  cfg = dict(cfg, results_dir=str(tmp_path / 'results'))
  path = tmp_path / 'run.yaml'
  path.write_text(yaml.safe_dump(cfg))
  res = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--configfile', str(path), '--epochs', str(epochs)],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
  assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
  losses = [float(m) for m in re.findall(r'- loss: ([0-9.eE+-]+)', res.stdout)]
  return res.stdout, losses, tmp_path / 'results' / 'run'


@pytest.mark.parametrize('conditioning', [None, 'global'])
def test_train_driver_trains_checkpoints_resumes_and_generates(tmp_path, conditioning):
  cfg = dict(BASE, conditioning=conditioning, mapping_layers=[4, 8], mapping_activation='leaky_relu', condition_classes=3)
  out, losses, run_dir = _run(tmp_path, cfg, 3)
  assert len(losses) == 3 and losses[-1] < losses[0] and all(np.isfinite(losses))
  ckpts = sorted(f for f in os.listdir(run_dir) if f.endswith('.weights.npz'))
  assert ckpts and re.fullmatch(r'weights-e\d{4}-lr[0-9.e+-]+\.weights\.npz', ckpts[-1])
  with np.load(run_dir / ckpts[-1]) as d:
    names = [str(n) for n in d['names']]
    assert ('mapping0/kernel' in names) == (conditioning == 'global')
    assert ('block0/conv_cond/kernel' in names) == (conditioning == 'global')
    if conditioning == 'global':
      assert d[f'w{names.index("mapping0/kernel"):03d}'].shape == (3, 4)     # 3 one-hot classes -> mapping [4, 8]
    assert 'adam_m' in d and int(d['drop_step']) > 0                           # optimizer state + dropout counter
  gen = np.load(run_dir / 'samples' / 'samples.npy')
  assert gen.shape == (4, 24, 1) and np.isfinite(gen).all() and np.abs(gen).max() <= 1.0
  assert 'Speed of generation was' in out
  # resume: epoch and lr parsed back out of the newest file name, training continues from there
  out2, losses2, _ = _run(tmp_path, cfg, 5)
  m = re.search(r'resuming from .*weights-e(\d+)-lr', out2)
  assert m and int(m.group(1)) == int(re.search(r'weights-e(\d+)', ckpts[-1]).group(1))
  assert 1 <= len(losses2) <= 5 - int(m.group(1)) and losses2[0] < losses[0]
