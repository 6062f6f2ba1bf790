import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, bench
from wavenets_amd import WaveNet, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
m = WaveNet(**bench.OTHER_CONFIGS['configs[3]'][0], device=dev, seed=0)
w = (torch.rand(B, m.receptive_field, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
L = C.CDLL(_lib.LIB_PATH)
m.generate(50, sample=w, use_queues=True, deterministic=True)
_lib.lib().wn_debug_set(24, 1)
m.generate(50, sample=w, use_queues=True, deterministic=True)
torch.cuda.synchronize()
out = np.zeros(30 * 8, dtype=np.uint64)
L.wn_debug_relay_ts.argtypes = [C.c_void_p, C.c_int]
print('rc', L.wn_debug_relay_ts(out.ctypes.data, 30))
ts = out.reshape(30, 8).astype(np.int64)
t0 = ts[0, 0]
print('block: entry staged oldtap x_in z_ready x_out | sacc_in sacc_out   (s_memtime ticks relative to block 0 entry)')
for b in range(30):
  print(b, ' '.join(f'{int(v - t0):7d}' if v else '      -' for v in ts[b]))
