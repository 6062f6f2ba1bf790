"""Phase stamps (s_memtime) of the streamed 128-channel block forward, timing build (-DWN_S128_DIAG, knob 29 = 64)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wavenets_amd import WaveNet, _lib
from wavenets_amd.data import synthetic_waveforms
dev = torch.device('cuda', 0)
m = WaveNet(blocks=30, channels=128, skip_channels=256, dilation_bound=1024, final_layers_channels=[128, 256],
            activation='leaky_relu', num_mixtures=10, sampling_function='logistic', bits=16, device=dev)
x = synthetic_waveforms(8, 16001, seed=5, device=dev)
L = _lib.lib()
raw = C.CDLL(_lib.LIB_PATH)
L.wn_debug_set(29, 64)
for _ in range(3):
  m.loss_and_grads(x)
torch.cuda.synchronize()
n = 512 * 4 * 2 * 10
buf = (C.c_ulonglong * n)()
assert raw.wn_debug_s128_ts(buf, n) == 0
ts = np.array(buf, dtype=np.float64).reshape(512, 4, 2, 10)      # the LAST block launch of the last pass
names = ['tile start', 'step 0 ready', 'conv step 8', 'conv done', 'gate done', 'sig+z stores issued', 'residual issued',
         '1x1 step 0 ready', '1x1 done', 'x_out stores issued']
for ps in (0, 1):
  t = ts[:, :, ps, :]
  ok = (t[:, :, 9] > t[:, :, 0]) & (t[:, :, 0] > 0)
  d = np.diff(t, axis=2)[ok]
  print(f'pass {ps}: {ok.sum()} waves; mean clocks per phase (100 MHz-independent shader clocks):')
  for i in range(9):
    print(f'   {names[i]:24s} -> {names[i + 1]:24s} {d[:, i].mean():9.0f}  (p10 {np.percentile(d[:, i], 10):8.0f} p90 {np.percentile(d[:, i], 90):8.0f})')
  print(f'   tile total {(t[:, :, 9] - t[:, :, 0])[ok].mean():9.0f}')
t0 = ts[:, :, 0, 0]; t1 = ts[:, :, 1, 9]
ok = (t0 > 0) & (t1 > t0)
print('first tile start -> second tile end:', (t1 - t0)[ok].mean(), 'gap between tiles:', (ts[:, :, 1, 0] - ts[:, :, 0, 9])[ok].mean())
L.wn_debug_set(29, 0)
