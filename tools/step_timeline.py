"""Per-kernel timeline of ONE training step from a rocprofv3 --kernel-trace database (rocprofv3 ... -- python3 tools/prof_train.py):
   python tools/step_timeline.py results.db   -> kernels of the last full step in start order, with stream, duration and the
   gap to the previous kernel's end on the same stream."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# a step starts at wn_shift_split_kernel (first kernel of wn_train_fwd_bwd) or the input conv forward
starts = [i for i, r in enumerate(rows) if 'wn_inconv_fwd_kernel' in r[0]]
if len(starts) < 3:
  print('no steps found'); sys.exit(1)
a, b = starts[-2], starts[-1]
t0 = rows[a][1]
last_end = {}
tot = {}
for name, st, en, stream, q in rows[a:b]:
  short = name.split('(')[0].replace('void ', '')[:60]
  gap = (st - last_end[q]) / 1000 if q in last_end else 0.0
  last_end[q] = en
  tot[short] = tot.get(short, 0) + (en - st) / 1000
  print(f'{(st - t0) / 1000:9.1f} us  q{q}  {(en - st) / 1000:8.1f} us  gap {gap:7.1f}  {short}')
print('step span', (rows[b][1] - t0) / 1000, 'us')
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
  print(f'{v:9.1f} us  {k}')
