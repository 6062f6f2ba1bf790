"""Times the fused residual-block forward launches (HIP events inside the library) for the
layer-forward kernel variants selectable through wn_debug_set(0, variant)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavenets_amd import WaveNet, _lib
import bench

dev = torch.device('cuda', 0)
m = WaveNet(**bench.CFG2, device=dev)
x = (torch.rand(8, 16000, 1, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
L = _lib.lib()
variants = [int(v) for v in (sys.argv[1:] or ['0', '1'])]   # 100 = exact-fp32 kernel (knob 1), else fp16-split
for rnd in range(3):
  for var in variants:
    L.wn_debug_set(1, 1 if var >= 100 else 0)
    L.wn_debug_set(0, var % 100)
    m(x)
    _lib.check(L.wn_prof_enable(m._plan, 30 * 4))
    for _ in range(4):
      m(x)
    torch.cuda.synchronize()
    n, ms = C.c_int32(), C.c_float()
    _lib.check(L.wn_prof_read(m._plan, C.byref(n), C.byref(ms)))
    print(f'round {rnd} variant {var}: {ms.value * 1e3:.1f} us/launch over {n.value} launches '
          f'-> {196.608e6 / (ms.value * 1e-3) / 1e12:.2f} TB/s algorithmic')
