"""Keras `.weights.h5` exchange (train.py:149-154, 237-238): the HDF5 subset written / read by
wavenets_amd/h5.py, the Keras-3 path layout of wavenets_amd/io.py, and the committed fixture."""
import os
import struct

import numpy as np
import pytest

from wavenets_amd import h5, io

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, 'golden', 'tiny.weights.h5')


def _load_fixture_module():
  import importlib.util
  spec = importlib.util.spec_from_file_location('make_h5_fixture', os.path.join(HERE, 'golden', 'make_h5_fixture.py'))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  return m


def test_fixture_bytes_follow_the_hdf5_specification():
  d = open(FIXTURE, 'rb').read()
  assert d[:8] == b'\x89HDF\r\n\x1a\n'                              # format signature
  assert d[8] == 0 and d[13] == 8 and d[14] == 8                     # superblock v0, 8-byte offsets and lengths
  leaf_k, internal_k = struct.unpack_from('<HH', d, 16)
  assert (leaf_k, internal_k) == (16, 16)
  base, _free, eof, _drv = struct.unpack_from('<QQQQ', d, 24)
  assert base == 0 and eof == len(d)                                 # end-of-file address = file size
  _name, root_hdr, cache, _r = struct.unpack_from('<QQII', d, 56)
  btree, heap = struct.unpack_from('<QQ', d, 80)
  assert cache == 1 and d[btree:btree + 4] == b'TREE' and d[heap:heap + 4] == b'HEAP'
  # root object header: version 1, one message of type 0x0011 (symbol table) pointing at the same B-tree / heap
  ver, _res, nmsg, refcnt, size = struct.unpack_from('<BBHII', d, root_hdr)
  mtype, msize = struct.unpack_from('<HH', d, root_hdr + 16)
  assert (ver, nmsg, refcnt, mtype, msize) == (1, 1, 1, 0x0011, 16) and size == 24
  assert struct.unpack_from('<QQ', d, root_hdr + 24) == (btree, heap)
  # the B-tree's first child is a symbol table node holding the root's members in name order
  used = struct.unpack_from('<H', d, btree + 6)[0]
  snod = struct.unpack_from('<Q', d, btree + 24 + 8)[0]
  assert used == 1 and d[snod:snod + 4] == b'SNOD'
  heap_data = struct.unpack_from('<Q', d, heap + 24)[0]
  n = struct.unpack_from('<H', d, snod + 6)[0]
  names = []
  for i in range(n):
    off = struct.unpack_from('<Q', d, snod + 8 + 40 * i)[0]
    names.append(d[heap_data + off:d.index(b'\0', heap_data + off)].decode())
  assert names == sorted(names) == ['causal', 'final', 'loss_tracker', 'mapping', 'prepare_target', 'wavenet_blocks']


def test_fixture_reads_back_and_maps_onto_the_model_variables():
  m = _load_fixture_module()
  tree = h5.read_h5(FIXTURE)
  assert set(tree['wavenet_blocks']) == {'wave_net_layer', 'wave_net_layer_1'}
  assert set(tree['final']) == {'conv1d', 'conv1d_1'}
  assert set(tree['mapping']) == {'layers'} and set(tree['mapping']['layers']) == {'dense', 'identity'}
  k = tree['wavenet_blocks']['wave_net_layer_1']['dilated_stack']['conv1d']['vars']['0']
  assert k.dtype == np.float32 and k.shape == (2, 4, 8)
  assert tree['prepare_target']['vars'] == {}
  model = m.Model()
  got = io._weights_from_tree(model, tree)
  for i, ((name, shape), w) in enumerate(zip(m.SHAPES, got)):
    assert w.shape == shape and np.array_equal(w, m.value(i, shape)), name
  # regenerating the fixture reproduces the committed bytes (the writer is deterministic)
  regenerated = io._tree_from_model(model)
  regenerated['loss_tracker'] = {'vars': {'0': np.float32(1.5), '1': np.float32(2.0)}}
  regenerated['prepare_target'] = {'vars': {}}
  regenerated['mapping']['layers']['identity'] = {'vars': {}}
  assert h5._Writer().finish(regenerated) == open(FIXTURE, 'rb').read()


def test_keras_paths():
  assert io.keras_path('causal/bias') == 'causal/vars/1'
  assert io.keras_path('block0/dil0/kernel') == 'wavenet_blocks/wave_net_layer/dilated_stack/conv1d/vars/0'
  assert io.keras_path('block12/dil2/bias') == 'wavenet_blocks/wave_net_layer_12/dilated_stack/conv1d_2/vars/1'
  assert io.keras_path('block3/conv_skip/kernel') == 'wavenet_blocks/wave_net_layer_3/conv_skip/vars/0'
  assert io.keras_path('final2/kernel') == 'final/conv1d_2/vars/0'
  assert io.keras_path('mapping1/bias') == 'mapping/layers/dense_1/vars/1'


def test_mapping_network_loads_with_and_without_the_sequential_layers_level():
  # keras.Sequential (src/model.py:142-148) saves its children under 'layers'; a flat 'mapping/dense' tree (what round 2
  # of this build wrote) must still load, and variable-free children (Identity) must not shift the member index
  m = _load_fixture_module()
  model = m.Model()
  tree = io._tree_from_model(model)
  assert 'layers' in tree['mapping']
  tree['mapping']['layers']['identity'] = {'vars': {}}
  a = io._weights_from_tree(model, tree)
  flat = dict(tree)
  flat['mapping'] = {k: v for k, v in tree['mapping']['layers'].items()}
  b = io._weights_from_tree(model, flat)
  assert all(np.array_equal(x, y) for x, y in zip(a, b))
  assert np.array_equal(a[-2], m.value(len(m.SHAPES) - 2, (5, 3)))


def test_h5_round_trip_large_groups_dtypes_and_errors(tmp_path):
  rng = np.random.default_rng(0)
  tree = {'many': {f'layer_{i:03d}': {'vars': {'0': rng.standard_normal((3, 2)).astype(np.float32)}} for i in range(100)},
          'misc': {'i64': np.arange(5, dtype=np.int64), 'i32': np.arange(3, dtype=np.int32), 'f64': np.array(3.5),
                   'empty': np.zeros((0, 3), np.float32), 'big': rng.standard_normal((64, 257)).astype(np.float32)}}
  p = str(tmp_path / 'x.h5')
  h5.write_h5(p, tree)                                   # 100 members: four symbol-table nodes under one B-tree node

  def same(a, b):
    if isinstance(a, dict):
      return set(a) == set(b) and all(same(a[k], b[k]) for k in a)
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
  assert same(tree, h5.read_h5(p))
  (tmp_path / 'bad.h5').write_bytes(b'not hdf5 at all')
  with pytest.raises(ValueError):
    h5.read_h5(str(tmp_path / 'bad.h5'))
  with pytest.raises(TypeError):
    h5.write_h5(p, {'x': np.array(['a'])})


def test_io_dispatches_on_the_file_name(tmp_path):
  m = _load_fixture_module()

  class M(m.Model):
    def set_weights(self, ws):
      self.loaded = [np.asarray(w) for w in ws]
  a, b = M(), M()
  path = str(tmp_path / io.checkpoint_name(7, 0.001, 'h5'))
  assert path.endswith('weights-e0007-lr0.001.weights.h5')          # the reference's file name, train.py:150
  io.save_weights(a, path)
  io.load_weights(b, path)
  assert all(np.array_equal(x, y) for x, y in zip(a.get_weights(), b.loaded))
  assert io.find_resume(str(tmp_path)) == (path, 7, 0.001)
  b.variable_names = b.variable_names + ['block2/conv1/kernel']
  with pytest.raises(ValueError, match='do not match'):
    io.load_weights(b, path)


def _tiny_file_bytes():
  return bytearray(h5._Writer().finish({'v': np.arange(6, dtype=np.float32).reshape(2, 3)}))


def _dataset_messages(d):
  """(message type, offset of the 8-byte message header) of the one dataset's object header, by walking the file the way
  the reader does: root header -> symbol table -> B-tree -> symbol node -> the dataset's object header."""
  root_hdr = struct.unpack_from('<Q', d, 64)[0]
  btree, _heap = struct.unpack_from('<QQ', d, root_hdr + 24)
  snod = struct.unpack_from('<Q', d, btree + 24 + 8)[0]
  hdr = struct.unpack_from('<Q', d, snod + 8 + 8)[0]
  _ver, _res, nmsg, _ref, _size = struct.unpack_from('<BBHII', d, hdr)
  out, p = [], hdr + 16
  for _ in range(nmsg):
    mtype, msize = struct.unpack_from('<HH', d, p)
    out.append((mtype, p))
    p += 8 + msize
  return hdr, out


@pytest.mark.parametrize('feature', ['superblock_v2', 'offset_size_4', 'object_header_v2', 'new_style_group', 'big_endian',
                                     'string_datatype', 'layout_v4', 'chunked', 'filtered'])
def test_reader_names_every_unsupported_feature(tmp_path, feature):
  """libhdf5 / h5py can emit more than the subset written here (v2 object headers and link messages with
  libver='latest', chunked or compressed datasets, other datatypes).  Each such feature must end in a
  NotImplementedError that names it -- never in a silent misread.  The files are a valid tiny file with exactly the
  bytes of that feature changed."""
  d = _tiny_file_bytes()
  assert np.array_equal(h5._Reader(bytes(d)).read(struct.unpack_from('<Q', d, 64)[0])['v'], np.arange(6, dtype=np.float32).reshape(2, 3))
  hdr, msgs = _dataset_messages(d)
  pos = dict((t, p) for t, p in msgs)
  if feature == 'superblock_v2':
    d[8] = 2; match = 'superblock version 2'
  elif feature == 'offset_size_4':
    d[13] = 4; match = 'offsets / lengths'
  elif feature == 'object_header_v2':
    d[hdr] = 2; match = 'object header version 2'
  elif feature == 'new_style_group':                      # the root's symbol-table message becomes a link-info message
    root_hdr = struct.unpack_from('<Q', d, 64)[0]
    struct.pack_into('<H', d, root_hdr + 16, 0x0002); match = 'new-style group'
  elif feature == 'big_endian':
    d[pos[0x0003] + 8 + 1] |= 1; match = 'big-endian'
  elif feature == 'string_datatype':
    d[pos[0x0003] + 8] = 0x13; match = 'datatype class 3'   # version 1, class 3 (string)
  elif feature == 'layout_v4':
    d[pos[0x0008] + 8] = 4; match = 'data layout message version 4'
  elif feature == 'chunked':
    d[pos[0x0008] + 8 + 1] = 2; match = 'chunked dataset'
  else:                                                    # the fill-value message becomes a filter-pipeline message
    struct.pack_into('<H', d, pos[0x0005], 0x000B); match = 'filtered'
  f = tmp_path / 'x.h5'
  f.write_bytes(bytes(d))
  with pytest.raises(NotImplementedError, match=match):
    h5.read_h5(str(f))
