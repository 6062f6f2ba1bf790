"""Checkpoint files with the reference's naming / resume convention (train.py:68-86,149-154).

The reference writes Keras ``weights-e{epoch:04d}-lr{lr}.weights.h5`` files (weights only, best
only) and resumes by parsing epoch and lr back out of the LAST file name.  HDF5 is not available
here (no h5py), so the container is ``.weights.npz``: arrays ``w000, w001, ...`` in Keras variable
order plus their names -- the order a ``.weights.h5`` importer would need.
Unlike the reference the optimizer state can be stored as well.
"""
from __future__ import annotations

import os
import re
from typing import Optional, Tuple

import numpy as np
import torch

_NAME = re.compile(r'weights-e(\d+)-lr([0-9.eE+-]+)\.weights\.npz$')


def checkpoint_name(epoch: int, lr: float) -> str:
  return f'weights-e{epoch:04d}-lr{lr}.weights.npz'


def save_weights(model, path: str, optimizer=None) -> None:
  arrays = {f'w{i:03d}': w for i, w in enumerate(model.get_weights())}
  arrays['names'] = np.array(model.variable_names)
  if optimizer is not None and optimizer.m is not None:
    arrays['adam_m'] = optimizer.m.detach().cpu().numpy()
    arrays['adam_v'] = optimizer.v.detach().cpu().numpy()
    arrays['adam_iterations'] = np.int64(optimizer.iterations)
  if getattr(model, 'dropout', 0) > 0:
    arrays['drop_step'] = np.int64(getattr(model, '_drop_step', 0))
  # write beside the target and rename: a kill mid-write must not leave a truncated file that sorts last and
  # is then picked by find_resume (the temp name does not match the checkpoint pattern)
  tmp = os.path.join(os.path.dirname(os.path.abspath(path)), f'.tmp-{os.getpid()}-' + os.path.basename(path))
  if not tmp.endswith('.npz'):
    tmp += '.npz'                     # np.savez appends .npz otherwise
  try:
    np.savez(tmp, **arrays)
    os.replace(tmp, path if path.endswith('.npz') else path + '.npz')
  finally:
    if os.path.exists(tmp):
      os.remove(tmp)


def load_weights(model, path: str, optimizer=None) -> None:
  with np.load(path, allow_pickle=False) as d:
    names = [str(n) for n in d['names']]
    if names != model.variable_names:
      raise ValueError('checkpoint variables do not match the model (different architecture?)')
    model.set_weights([d[f'w{i:03d}'] for i in range(len(names))])
    if optimizer is not None and 'adam_m' in d:
      optimizer.build(model)
      optimizer.m.copy_(torch.from_numpy(d['adam_m']).to(optimizer.m.device))
      optimizer.v.copy_(torch.from_numpy(d['adam_v']).to(optimizer.v.device))
      optimizer.iterations = int(d['adam_iterations'])
    if 'drop_step' in d and hasattr(model, 'set_drop_step'):
      model.set_drop_step(int(d['drop_step']))


def find_resume(run_dir: str) -> Optional[Tuple[str, int, float]]:
  """Last checkpoint of a run directory -> (file, initial_epoch, lr), as train.py:68-86 does."""
  if not os.path.isdir(run_dir):
    return None
  files = sorted(f for f in os.listdir(run_dir) if _NAME.search(f))
  for name in reversed(files):          # newest first; a file that does not open (truncated write) is skipped
    full = os.path.join(run_dir, name)
    try:
      with np.load(full, allow_pickle=False) as d:
        d['names']
    except Exception:                   # zipfile.BadZipFile, OSError, KeyError, ValueError ...
      continue
    m = _NAME.search(name)
    return full, int(m.group(1)), float(m.group(2))
  return None
