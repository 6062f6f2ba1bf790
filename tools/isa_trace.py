"""Compact memory / matrix instruction trace of a kernel's ISA: L = global load, D = LDS-DMA, M = MFMA, S = global store,
wN = s_waitcnt vmcnt(N), B = s_barrier -- shows at a glance whether hipcc kept a kernel's loads ahead of their use or sank
them next to it with vmcnt(0) behind each (found that way in round 4: the fused forward block kernel's conv walked six
exposed global round trips per tile).   usage: python tools/isa_trace.py file.hip kernel_substring [max_chars]"""
import re, subprocess, sys, tempfile, os
src, pat = os.path.abspath(sys.argv[1]), sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 2400
out = tempfile.mktemp(suffix='.s')
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=on', '-S', '--cuda-device-only', '-o', out, src],
               check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
txt = open(out).read()
for m in re.finditer(r'^(_Z\w*%s\w*):' % re.escape(pat), txt, re.M):
  a = m.end(); b = txt.index('.end_amdhsa_kernel', a)
  seq = []
  for l in txt[a:b].split('\n'):
    mm = re.match(r'\s+(\S+)', l)
    if not mm: continue
    op = mm.group(1)
    if op.startswith('global_load_lds'): seq.append('D')
    elif op.startswith(('global_load', 'buffer_load')): seq.append('L')
    elif op.startswith('v_mfma'): seq.append('M')
    elif op.startswith('s_waitcnt') and 'vmcnt' in l: seq.append('w' + re.search(r'vmcnt\((\d+)\)', l).group(1))
    elif op.startswith(('global_store', 'buffer_store')): seq.append('S')
    elif op.startswith('s_barrier'): seq.append('B')
    elif op.startswith('scratch_'): seq.append('X')
  # run-length encode
  rl = []
  for t in seq:
    if rl and rl[-1][0] == t: rl[-1][1] += 1
    else: rl.append([t, 1])
  print(m.group(1)); print(' '.join(t if n == 1 else f'{t}x{n}' for t, n in rl)[:lim]); print()
