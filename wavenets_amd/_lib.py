"""ctypes binding of libwn_hip.so (C-ABI declared in include/wn_hip.h).

The HIP library is the product: there is no CPU fallback.  Importing this module without a
built library raises immediately; calling a compute entry point without a GPU raises from
the HIP runtime.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# WN_HIP_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get('WN_HIP_LIB') or os.path.join(_HERE, 'libwn_hip.so')

WN_OK, WN_E_INVALID, WN_E_UNSUPPORTED, WN_E_HIP = 0, -1, -2, -3
ACTIVATIONS = {None: 0, 'linear': 0, 'relu': 1, 'leaky_relu': 2, 'tanh': 3, 'sigmoid': 4, 'elu': 5}
HEADS = {'categorical': 0, 'logistic': 1, 'gaussian': 2}
WN_MAX_FINAL = 8
WN_MAX_MAPPING = 8


class WnConfig(C.Structure):
  """struct wn_config (include/wn_hip.h) == WaveNet constructor keywords, src/model.py:14-34."""
  _fields_ = [
      ('kernel_size', C.c_int32), ('channels', C.c_int32), ('blocks', C.c_int32),
      ('layers_per_block', C.c_int32), ('activation', C.c_int32), ('dilation_bound', C.c_int32),
      ('num_mixtures', C.c_int32), ('head', C.c_int32), ('bits', C.c_int32),
      ('skip_channels', C.c_int32), ('dilation_channels', C.c_int32), ('use_residual', C.c_int32),
      ('use_skip', C.c_int32), ('n_final', C.c_int32), ('final_channels', C.c_int32 * WN_MAX_FINAL),
      ('cond_inputs', C.c_int32), ('n_mapping', C.c_int32),
      ('mapping_channels', C.c_int32 * WN_MAX_MAPPING), ('mapping_activation', C.c_int32),
      ('l2_reg_factor', C.c_float),
  ]


class WnLayerDesc(C.Structure):
  """struct wn_layer_desc == WaveNetLayer constructor keywords, src/layers.py:10-20."""
  _fields_ = [
      ('kernel_size', C.c_int32), ('channels', C.c_int32), ('dilation_channels', C.c_int32),
      ('skip_channels', C.c_int32), ('depth', C.c_int32), ('dilations', C.c_int32 * 16),
      ('activation', C.c_int32), ('residual', C.c_int32), ('cond_channels', C.c_int32),
      ('in_channels', C.c_int32),
  ]


def build_library(verbose: bool = False) -> str:
  """Compile wavenets_amd/csrc/*.hip for gfx950 into wavenets_amd/libwn_hip.so (in-tree)."""
  cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j8']
  res = subprocess.run(cmd, capture_output=not verbose, text=True)
  if res.returncode != 0:
    raise RuntimeError('building libwn_hip.so failed:\n' + (res.stdout or '') + (res.stderr or ''))
  return LIB_PATH


_P = C.c_void_p
_SIGS = {
    'wn_last_error_string': (C.c_char_p, []),
    'wn_plan_create': (_P, [C.POINTER(WnConfig)]),
    'wn_plan_destroy': (None, [_P]),
    'wn_plan_param_count': (C.c_int64, [_P]),
    'wn_plan_num_tensors': (C.c_int32, [_P]),
    'wn_plan_tensor_info': (C.c_int, [_P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    'wn_plan_receptive_field': (C.c_int32, [_P]),
    'wn_plan_out_channels': (C.c_int32, [_P]),
    'wn_plan_dilation': (C.c_int32, [_P, C.c_int32]),
    'wn_plan_workspace_floats': (C.c_int64, [_P, C.c_int32, C.c_int32, C.c_int32]),
    'wn_debug_set': (C.c_int, [C.c_int, C.c_int]),
    'wn_exec_create': (_P, [_P]),
    'wn_exec_destroy': (None, [_P]),
    'wn_exec_bind': (C.c_int, [_P]),
    'wn_debug_value': (C.c_int, [C.c_int]),
    'wn_debug_gen_ts': (C.c_int, [_P]),
    'wn_plan_set_dropout': (C.c_int, [_P, C.c_float, C.c_uint64, C.c_uint64]),
    'wn_dropout_key_for': (C.c_uint32, [C.c_uint64, C.c_int32, C.c_uint64]),
    'wn_plan_describe': (C.c_int, [_P, C.c_char_p, C.c_int32]),
    'wn_prof_enable': (C.c_int, [_P, C.c_int32]),
    'wn_prof_read': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    'wn_debug_ws_region': (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]),
    'wn_stack_prof_enable': (C.c_int, [_P, C.c_int32]),
    'wn_stack_prof_read': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    'wn_stack_prof_read_foldprep': (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    'wn_phase_enable': (C.c_int, [_P, C.c_int32]),
    'wn_phase_read': (C.c_int, [_P, C.POINTER(C.c_float)]),
    'wn_forward': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'wn_forward_training': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'wn_train_fwd_bwd': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                   _P, C.c_int64, _P]),
    'wn_eval_loss': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    'wn_adam_step': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, _P, _P]),
    'wn_adam_step_guarded': (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, _P, _P, _P]),
    'wn_plan_range_slot': (C.c_int64, [_P, C.c_int32, C.c_int32, C.c_int32]),
    'wn_range_limit': (C.c_float, []),
    'wn_generate': (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _P,
                              _P, C.c_int64, _P]),
    'wn_generate_workspace_floats': (C.c_int64, [_P, C.c_int32, C.c_int32]),
    'wn_generate_guard_slot': (C.c_int64, [_P, C.c_int32, C.c_int32]),
    'wn_layer_saved_floats': (C.c_int64, [C.POINTER(WnLayerDesc), C.c_int32, C.c_int32]),
    'wn_layer_workspace_floats': (C.c_int64, [C.POINTER(WnLayerDesc), C.c_int32, C.c_int32]),
    'wn_layer_param_count': (C.c_int64, [C.POINTER(WnLayerDesc)]),
    'wn_layer_fwd': (C.c_int, [C.POINTER(WnLayerDesc), _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    'wn_layer_bwd': (C.c_int, [C.POINTER(WnLayerDesc), _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P,
                               _P, _P, _P]),
    'wn_quantize': (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    'wn_dequantize': (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    'wn_mulaw': (C.c_int, [_P, _P, C.c_int64, _P]),
    'wn_inv_mulaw': (C.c_int, [_P, _P, C.c_int64, _P]),
    'wn_loss_fn': (C.c_int, [C.c_int32, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    'wn_plan_arm_step_sample': (C.c_int, [_P, _P, C.c_int32, C.c_uint64, C.c_uint64]),
    'wn_plan_set_train_phases': (C.c_int, [_P, C.c_int32]),
    'wn_sum_squared_error': (C.c_int, [_P, _P, C.c_int64, C.c_float, _P, _P, _P]),
    'wn_sample_waveform': (C.c_int, [C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_uint64, C.c_uint64, _P, _P]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib() -> C.CDLL:
  """Load libwn_hip.so once; raise loudly when it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
          '(wavenets_amd has no CPU fallback)')
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
      fn = getattr(l, name)
      fn.restype = res
      fn.argtypes = args
    _lib = l
  return _lib


class WnError(RuntimeError):
  pass


def check(rc: int) -> None:
  """Map C-ABI return codes onto the exception types of the reference."""
  if rc == WN_OK:
    return
  msg = lib().wn_last_error_string().decode()
  if rc == WN_E_INVALID:
    raise ValueError(msg)
  if rc == WN_E_UNSUPPORTED:
    raise NotImplementedError(msg)
  raise WnError(f'HIP error: {msg}')


def ptr(t):
  """Device pointer of a torch tensor (or None)."""
  return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
  import torch
  return C.c_void_p(torch.cuda.current_stream().cuda_stream)
