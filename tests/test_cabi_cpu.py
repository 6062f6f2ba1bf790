"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol declared in
include/wn_hip.h, and its parameter layout (Keras creation order) agrees with the pure-Python
spec.  No compute entry point is called (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from wavenets_amd import _lib, spec
from oracle import wavenet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
  if not os.path.exists(_lib.LIB_PATH):
    _lib.build_library()
  return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
  hdr = open(os.path.join(ROOT, 'include', 'wn_hip.h')).read()
  declared = set(re.findall(r'\b(wn_[a-z_0-9]+)\s*\(', hdr))
  declared -= {'wn_plan', 'wn_config', 'wn_layer_desc'}
  assert declared, 'no declarations parsed'
  for name in sorted(declared):
    assert hasattr(lib, name), f'{name} declared in include/wn_hip.h but not exported'
  assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def _cfg(**kw):
  d = dict(kernel_size=2, channels=32, blocks=10, layers_per_block=1, activation=None, conditioning=None,
           mapping_layers=None, mapping_activation=None, dropout=0, dilation_bound=512, num_mixtures=None,
           sampling_function='categorical', bits=8, skip_channels=None, dilation_channels=None,
           use_residual=True, use_skip=True, final_layers_channels=[], l2_reg_factor=0)
  d.update(kw)
  return spec.validate(**d)


def _plan(lib, s, cond_inputs=0):
  cfg = _lib.WnConfig()
  cfg.kernel_size, cfg.channels, cfg.blocks, cfg.layers_per_block = s.kernel_size, s.channels, s.blocks, s.layers_per_block
  cfg.activation = _lib.ACTIVATIONS[s.activation]
  cfg.dilation_bound = s.dilation_bound
  cfg.num_mixtures = s.num_mixtures or 0
  cfg.head = _lib.HEADS[s.sampling_function]
  cfg.bits = s.bits
  cfg.skip_channels = s.skip_channels or 0
  cfg.dilation_channels = s.dilation_channels or 0
  cfg.use_residual, cfg.use_skip = int(s.use_residual), int(s.use_skip)
  cfg.n_final = len(s.final_layers_channels)
  for i, c in enumerate(s.final_layers_channels):
    cfg.final_channels[i] = c
  cfg.cond_inputs = cond_inputs
  cfg.n_mapping = len(s.mapping_layers)
  for i, c in enumerate(s.mapping_layers):
    cfg.mapping_channels[i] = c
  cfg.mapping_activation = _lib.ACTIVATIONS[s.mapping_activation]
  return lib.wn_plan_create(C.byref(cfg))


@pytest.mark.parametrize('kw,cond_inputs,count,rf', [
    (dict(blocks=10, channels=32, dilation_bound=1024, final_layers_channels=[]), 0, 60704, 1025),
    (dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
          activation='leaky_relu'), 0, 1251264, 3071),
    (dict(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
          num_mixtures=10, sampling_function='logistic', bits=16), 0, 3533854, 3071),
    (dict(blocks=30, channels=64, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
          activation='leaky_relu', conditioning='global', mapping_layers=[8, 16, 32],
          mapping_activation='leaky_relu'), 110, None, 3071),
    (dict(blocks=5, layers_per_block=5, channels=32, dilation_bound=256, num_mixtures=8,
          sampling_function='gaussian', bits=16, final_layers_channels=[128, 256]), 0, None, 768),
])
def test_plan_layout_matches_spec(lib, kw, cond_inputs, count, rf):
  s = _cfg(**kw)
  p = C.c_void_p(_plan(lib, s, cond_inputs))
  assert p.value, lib.wn_last_error_string()
  try:
    shapes = s.param_shapes(cond_inputs)
    assert lib.wn_plan_num_tensors(p) == len(shapes)
    total = sum(int(np.prod(sh)) for _, sh in shapes)
    assert lib.wn_plan_param_count(p) == total
    if count is not None:
      assert total == count                       # BASELINE.md anchors
    assert lib.wn_plan_receptive_field(p) == rf == s.receptive_field
    assert lib.wn_plan_out_channels(p) == s.out_channels
    off, ln, nd, isk = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    sh = (C.c_int64 * 3)()
    run = 0
    for i, (name, shp) in enumerate(shapes):
      assert lib.wn_plan_tensor_info(p, i, C.byref(off), C.byref(ln), C.byref(nd), sh, C.byref(isk)) == 0
      assert off.value == run and ln.value == int(np.prod(shp)) and tuple(sh[:nd.value]) == tuple(shp)
      assert bool(isk.value) == name.endswith('kernel')
      run += ln.value
    for i, d in enumerate(s.dilations):
      assert lib.wn_plan_dilation(p, i) == d
    assert lib.wn_plan_workspace_floats(p, 2, 4000, 1) > lib.wn_plan_workspace_floats(p, 2, 4000, 0) > 0
    # the oracle's own enumeration agrees too
    oc = O.OracleConfig(**{k: v for k, v in kw.items() if k not in ('mapping_layers',)},
                        mapping_layers=kw.get('mapping_layers'), cond_inputs=cond_inputs)
    assert [n for n, _ in O.param_shapes(oc)] == [n for n, _ in shapes]
  finally:
    lib.wn_plan_destroy(p)


def test_plan_rejects_bad_config(lib):
  s = _cfg()
  cfg_ok = _plan(lib, s)
  assert cfg_ok
  lib.wn_plan_destroy(C.c_void_p(cfg_ok))
  cfg = _lib.WnConfig()
  cfg.kernel_size, cfg.channels, cfg.blocks, cfg.layers_per_block = 2, 32, 3, 1
  cfg.dilation_bound = 500            # not a power of kernel_size (src/model.py:56-57)
  cfg.bits = 8
  assert not lib.wn_plan_create(C.byref(cfg))
  assert b'power of kernel_size' in lib.wn_last_error_string()


def test_spec_validation_messages():
  with pytest.raises(ValueError, match='Kernel size must be at least 2'):
    _cfg(kernel_size=1)
  with pytest.raises(ValueError, match='power of kernel_size'):
    _cfg(dilation_bound=500)
  with pytest.raises(ValueError, match='Categorical sampling cannot be used with mixtures'):
    _cfg(num_mixtures=3)
  with pytest.raises(ValueError, match="Conditioning must be 'global', 'local' or None"):
    _cfg(conditioning='speaker')
  with pytest.raises(ValueError, match='Dropout must be between 0 and 1'):
    _cfg(dropout=1.5)
  with pytest.raises(ValueError, match='Blocks must be at least 1'):
    _cfg(blocks=0)


def test_layer_desc_param_count(lib):
  d = _lib.WnLayerDesc()
  d.kernel_size, d.channels, d.dilation_channels, d.skip_channels, d.depth = 2, 64, 64, 256, 1
  d.dilations[0] = 4
  d.in_channels = 64
  # cfg2 weights/layer: 37 312
  assert lib.wn_layer_param_count(C.byref(d)) == 37312
  assert lib.wn_layer_workspace_floats(C.byref(d), 2, 1000) > 0
  assert lib.wn_layer_saved_floats(C.byref(d), 2, 1000) >= 2 * 1000 * (64 + 64)   # sigmoid + gated activation per row


def test_plan_describe_names_the_kernel_family_of_every_phase():
  """The fast paths are shape-specialised; wn_plan_describe says which family a plan takes (no silent slow path)."""
  lib = _lib.lib()

  def describe(s, **kw):
    p = C.c_void_p(_plan(lib, s, **kw))
    buf = C.create_string_buffer(2048)
    assert lib.wn_plan_describe(p, buf, 2048) == 0
    lib.wn_plan_destroy(p)
    return buf.value.decode()
  base = dict(blocks=30, dilation_bound=1024, skip_channels=256, final_layers_channels=[128, 256], activation='leaky_relu')
  c1 = describe(_cfg(channels=64, **base))                                   # BASELINE configs[1]
  assert 'wn_layer_fwd_f16_kernel' in c1 and 'wn_bwd_pair_kernel' in c1 and 'wn_wgrad_layer_kernel' in c1 and 'folded' in c1
  c3 = describe(_cfg(channels=128, num_mixtures=10, sampling_function='logistic', bits=16, **base))     # configs[3]
  assert 'wn_layer_fwd_s128_kernel' in c3 and 'wn_bwd_s128_kernel' in c3 and 'wn_wgrad_tr_kernel' in c3 and 'folded' in c3
  d = describe(_cfg(blocks=5, layers_per_block=5, dilation_bound=256, num_mixtures=8, sampling_function='gaussian', bits=16,
                    final_layers_channels=[128, 256], activation='leaky_relu'))   # the reference's default (train.py:22-50)
  assert 'layers_per_block > 1' in d
  lib.wn_debug_set(1, 1)
  try:
    assert 'exact fp32' in describe(_cfg(channels=64, **base))
  finally:
    lib.wn_debug_set(1, 0)
