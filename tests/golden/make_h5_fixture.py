"""Builds tests/golden/tiny.weights.h5: a Keras-3-layout weight file of a 2-block conditioned WaveNet (4 channels) whose
every variable is filled with small integers that encode (tensor index, element index), written by wavenets_amd/h5.py.
TensorFlow / Keras / h5py are not installed in the build image, so the file cannot come from Keras itself; the tests pin
(a) the byte-level structure against the HDF5 specification (signature, superblock fields, object and node signatures)
and (b) the reader and the Keras path mapping against the known contents.

  python tests/golden/make_h5_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wavenets_amd import h5, io  # noqa: E402

# variable names and shapes of WaveNet(blocks=2, channels=4, skip_channels=8, dilation_bound=4, final_layers_channels=[6],
#                                    conditioning='global', mapping_layers=[3], bits=4) with a 5-way condition
SHAPES = [('causal/kernel', (2, 1, 4)), ('causal/bias', (4,))]
for b in range(2):
  SHAPES += [(f'block{b}/dil0/kernel', (2, 4, 8)), (f'block{b}/dil0/bias', (8,)), (f'block{b}/conv1/kernel', (1, 4, 4)),
             (f'block{b}/conv1/bias', (4,)), (f'block{b}/conv_skip/kernel', (1, 4, 8)), (f'block{b}/conv_skip/bias', (8,)),
             (f'block{b}/conv_cond/kernel', (1, 3, 8)), (f'block{b}/conv_cond/bias', (8,))]
SHAPES += [('final0/kernel', (1, 8, 6)), ('final0/bias', (6,)), ('final1/kernel', (1, 6, 16)), ('final1/bias', (16,)),
           ('mapping0/kernel', (5, 3)), ('mapping0/bias', (3,))]


def value(i, shape):
  return (1000.0 * i + np.arange(int(np.prod(shape)), dtype=np.float32)).reshape(shape)


class Model:
  variable_names = [n for n, _ in SHAPES]

  def get_weights(self):
    return [value(i, s) for i, (_, s) in enumerate(SHAPES)]


if __name__ == '__main__':
  out = os.path.join(ROOT, 'tests', 'golden', 'tiny.weights.h5')
  tree = io._tree_from_model(Model())
  tree['loss_tracker'] = {'vars': {'0': np.float32(1.5), '1': np.float32(2.0)}}     # a metric's state: ignored on import
  tree['prepare_target'] = {'vars': {}}                                            # a layer without variables
  tree['mapping']['layers']['identity'] = {'vars': {}}                             # the Sequential's trailing Identity
  h5.write_h5(out, tree)
  print(out, os.path.getsize(out), 'bytes')
