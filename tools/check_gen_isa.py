"""Static check of the generation chain kernel's hand-waited prefetches.

wn_gen.hip issues its weight prefetches as inline-asm global loads and waits for them with hand-placed
s_waitcnt vmcnt(N).  The compiler does not know those registers are in flight, so nothing stops it from
reading or overwriting one (a phi copy, a spill, a reused temporary) before the wait.  This script
disassembles the kernel and walks the control-flow graph from every asm load: on every path, the first
instruction that touches a destination register must come after a hand-placed wait.

usage: python tools/check_gen_isa.py [path/to/wn_gen.s]   (default: compiles wavenets_amd/csrc/wn_gen.hip)
"""
import os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, '..', 'wavenets_amd', 'csrc', 'wn_gen.hip')


def compile_asm():
  out = '/tmp/wn_gen_check.s'
  flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '--cuda-device-only']
  subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-S', SRC, '-o', out])
  # -S does not assemble the inline asm: a bad operand only shows when the object is produced
  subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-c', SRC, '-o', '/tmp/wn_gen_check.o'])
  return out


def regs(tok):
  out = []
  for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
    if m.group(1):
      out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    else:
      out.append(int(m.group(3)))
  return out


def check_kernel(name, body):
  # instructions with flags: in_asm (inside ASMSTART/ASMEND)
  ins, labels, in_asm = [], {}, False
  for l in body:
    t = l.strip()
    if t.startswith(';;#ASMSTART'):
      in_asm = True; continue
    if t.startswith(';;#ASMEND'):
      in_asm = False; continue
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
      continue
    if t.startswith('.LBB'):
      labels[t.split(':')[0]] = len(ins); continue
    t = t.split(';')[0].strip()
    if t:
      ins.append((t, in_asm))
  def succ(i):
    t = ins[i][0]
    op = t.split()[0]
    if op == 's_endpgm':
      return []
    if op == 's_branch':
      return [labels[t.split()[1]]]
    if op.startswith('s_cbranch'):
      return [labels[t.split()[1]], i + 1]
    return [i + 1] if i + 1 < len(ins) else []
  bad = 0
  nloads = 0
  for i, (t, a) in enumerate(ins):
    if not (a and t.startswith('global_load_dwordx4')):
      continue
    nloads += 1
    dest = set(regs(t.split(',')[0]))
    seen, stack, pred = set(), [], {}
    for n in succ(i):
      pred.setdefault(n, i); stack.append(n)
    while stack:
      j = stack.pop()
      if j in seen:
        continue
      seen.add(j)
      tj, aj = ins[j]
      if aj and tj.startswith('s_waitcnt') and 'vmcnt' in tj:
        continue                      # hand-placed wait: this path is safe from here on
      if tj.startswith('s_waitcnt') and 'vmcnt(0)' in tj:
        continue                      # a compiler drain is as good
      if aj and tj.startswith('global_load_dwordx4') and set(regs(tj.split(',')[0])) == dest:
        # the same registers fetched again before any use (tail dummy fetch after the last block): fine, keep walking
        pass
      elif dest & set(regs(tj)):
        path, q = [], j
        while q != i:
          q = pred[q]
          if ins[q][0].split()[0] in ('s_branch',) or ins[q][0].startswith('s_cbranch'):
            path.append(f'{q}:{ins[q][0]}')
        print(f'{name}: instruction {j} "{tj}" touches {sorted(dest & set(regs(tj)))} of the load at {i} "{t}" before a wait'
              f' (branches on the way: {" <- ".join(path[:12])})')
        bad += 1
        continue
      for n in succ(j):
        if n not in seen:
          pred.setdefault(n, j); stack.append(n)
  return nloads, bad


def main():
  path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
  lines = open(path).read().split('\n')
  total_bad, total_loads = 0, 0
  i = 0
  while i < len(lines):
    m = re.match(r'^(_Z\d+wn_gen_chain3_kernel\w+):', lines[i])
    if m:
      j = i
      while not lines[j].startswith('.Lfunc_end'):
        j += 1
      n, b = check_kernel(m.group(1), lines[i + 1:j])
      print(f'{m.group(1)}: {n} hand-waited loads, {b} unsafe touches')
      total_loads += n; total_bad += b
      i = j
    i += 1
  if total_loads == 0:
    print('no hand-waited loads found'); return 2
  return 1 if total_bad else 0


if __name__ == '__main__':
  sys.exit(main())
