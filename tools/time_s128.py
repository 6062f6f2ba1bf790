"""Per-launch time of the streamed 128-channel block forward (training form) under its timing ablations
(library built with -DWN_S128_DIAG; knob 29): python tools/time_s128.py [DIAG ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, _lib
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
m = WaveNet(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16, device=dev)
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
L = _lib.lib()
names = {0: 'full', 1: 'no stores', 2: 'no x requests', 4: 'no products', 8: 'no gate', 16: 'no residual loads',
         32: 'no weight requests', 19: 'no activation traffic (1|2|16)', 12: 'no arithmetic (4|8)',
         51: 'weights + compute only (1|2|16|32: no memory at all)', 63: 'nothing but the loop'}
variants = [int(v) for v in (sys.argv[1:] or ['0', '1', '2', '4', '8', '16', '32', '19', '12', '51', '63'])]
for rnd in range(2):
  for var in variants:
    L.wn_debug_set(29, var)
    m.loss_and_grads(x)
    _lib.check(L.wn_prof_enable(m._plan, 30 * 3))
    for _ in range(3):
      m.loss_and_grads(x)
    torch.cuda.synchronize()
    n, ms = C.c_int32(), C.c_float()
    _lib.check(L.wn_prof_read(m._plan, C.byref(n), C.byref(ms)))
    print(f'round {rnd} diag {var:3d} {names.get(var, ""):55s}: {ms.value * 1e3:7.1f} us/launch over {n.value} launches', flush=True)
L.wn_debug_set(29, 0)
