"""Checkpoint files with the reference's naming / resume convention (train.py:68-86,149-154).

The reference writes Keras ``weights-e{epoch:04d}-lr{lr}.weights.h5`` files (weights only, best
only) and resumes by parsing epoch and lr back out of the LAST file name.  Two containers are supported, chosen by the
file name:

* ``.weights.npz`` (training checkpoints of this driver): arrays ``w000, w001, ...`` in Keras variable order plus their
  names; unlike the reference the optimizer state and the dropout counter can be stored as well;
* ``.weights.h5`` (exchange with the reference): an HDF5 file in the layout Keras 3's ``saving_lib`` gives a subclassed
  model -- one group per tracked attribute path, variables as ``<path>/vars/<i>`` (kernel = 0, bias = 1), members of list
  attributes named after their class in snake case with ``_<n>`` suffixes:
      causal/vars/{0,1}
      wavenet_blocks/wave_net_layer[_b]/dilated_stack/conv1d[_i]/vars/{0,1}
      wavenet_blocks/wave_net_layer[_b]/{conv1,conv_skip,conv_cond}/vars/{0,1}
      final/conv1d[_i]/vars/{0,1}            mapping/layers/dense[_j]/vars/{0,1}
  (``mapping`` is a ``keras.Sequential``, src/model.py:142-148: saving_lib keeps a Sequential's children under an extra
  ``layers`` group; the trailing ``Identity`` layer has no variables)
  written and read by ``wavenets_amd/h5.py`` (no h5py in the image).  Weights only, like the reference.  The importer
  identifies list members by their numeric suffix, not by the exact spelling of the class name, ignores groups it does
  not know (metrics, seed generators, empty ``vars`` groups) and accepts the mapping network with or without the
  ``layers`` level.  TensorFlow / Keras are not installed here, so this layout is derived from the
  Keras 3 sources' description, not validated against a file Keras wrote.
"""
from __future__ import annotations

import os
import re
from typing import Optional, Tuple

import numpy as np
import torch

_NAME = re.compile(r'weights-e(\d+)-lr([0-9.eE+-]+)\.weights\.(npz|h5)$')


def checkpoint_name(epoch: int, lr: float, ext: str = 'npz') -> str:
  return f'weights-e{epoch:04d}-lr{lr}.weights.{ext}'


# ------------------------------------------------------------------ Keras .weights.h5 layout
def _suffix(name: str, i: int) -> str:
  return name if i == 0 else f'{name}_{i}'


def keras_path(var_name: str) -> str:
  """``block3/dil0/kernel`` -> ``wavenet_blocks/wave_net_layer_3/dilated_stack/conv1d/vars/0`` etc."""
  parts = var_name.split('/')
  idx = {'kernel': '0', 'bias': '1'}[parts[-1]]
  head = parts[0]
  if head == 'causal':
    path = ['causal']
  elif head.startswith('block'):
    path = ['wavenet_blocks', _suffix('wave_net_layer', int(head[5:]))]
    sub = parts[1]
    path += ['dilated_stack', _suffix('conv1d', int(sub[3:]))] if sub.startswith('dil') else [sub]
  elif head.startswith('final'):
    path = ['final', _suffix('conv1d', int(head[5:]))]
  elif head.startswith('mapping'):
    path = ['mapping', 'layers', _suffix('dense', int(head[7:]))]
  else:
    raise ValueError(f'unknown variable {var_name}')
  return '/'.join(path + ['vars', idx])


def _members(group: dict):
  """Members of a list-attribute group in list order: sorted by the numeric suffix of their names."""
  def key(n):
    m = re.search(r'_(\d+)$', n)
    return int(m.group(1)) if m else 0
  return [group[n] for n in sorted((n for n in group if isinstance(group[n], dict)), key=key)]


def _with_vars(members):
  """Only the members that own variables (a Sequential also lists Identity / Dropout children)."""
  return [g for g in members if isinstance(g.get('vars'), dict) and len(g['vars']) > 0]


def _tree_from_model(model) -> dict:
  tree: dict = {}
  for name, w in zip(model.variable_names, model.get_weights()):
    node = tree
    parts = keras_path(name).split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = np.asarray(w, dtype=np.float32)
  return tree


def _weights_from_tree(model, tree: dict):
  """Arrays in the model's variable order, looked up structurally (containers by member index)."""
  def var(group, idx):
    try:
      return np.asarray(group['vars'][str(idx)], dtype=np.float32)
    except (KeyError, TypeError):
      raise ValueError('checkpoint variables do not match the model (different architecture?)')
  blocks = _members(tree.get('wavenet_blocks', {}))
  finals = _members(tree.get('final', {}))
  mapping_group = tree.get('mapping', {})
  if isinstance(mapping_group.get('layers'), dict):        # keras.Sequential: children live under 'layers'
    mapping_group = mapping_group['layers']
  mapping = _with_vars(_members(mapping_group))
  out = []
  for name in model.variable_names:
    parts = name.split('/')
    i = {'kernel': 0, 'bias': 1}[parts[-1]]
    head = parts[0]
    try:
      if head == 'causal':
        g = tree['causal']
      elif head.startswith('block'):
        blk = blocks[int(head[5:])]
        g = _members(blk['dilated_stack'])[int(parts[1][3:])] if parts[1].startswith('dil') else blk[parts[1]]
      elif head.startswith('final'):
        g = finals[int(head[5:])]
      else:
        g = mapping[int(head[7:])]
    except (KeyError, IndexError):
      raise ValueError('checkpoint variables do not match the model (different architecture?)')
    out.append(var(g, i))
  return out


def save_weights(model, path: str, optimizer=None) -> None:
  if path.endswith('.h5'):                    # Keras exchange format: weights only, as the reference stores them
    from . import h5
    tmp = os.path.join(os.path.dirname(os.path.abspath(path)), f'.tmp-{os.getpid()}-' + os.path.basename(path))
    try:
      h5.write_h5(tmp, _tree_from_model(model))
      os.replace(tmp, path)
    finally:
      if os.path.exists(tmp):
        os.remove(tmp)
    return
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
  if path.endswith('.h5'):
    from . import h5
    model.set_weights(_weights_from_tree(model, h5.read_h5(path)))
    return
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
      if name.endswith('.h5'):
        from . import h5
        h5.read_h5(full)
      else:
        with np.load(full, allow_pickle=False) as d:
          d['names']
    except Exception:                   # zipfile.BadZipFile, OSError, KeyError, ValueError ...
      continue
    m = _NAME.search(name)
    return full, int(m.group(1)), float(m.group(2))
  return None
