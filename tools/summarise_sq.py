"""profiles/<tag>_train_pmc_sq.csv from a rocprofv3 `--pmc SQ_...` csv under gpurun_out/final_sq (per-kernel averages;
mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES / 4 as in the round-1 summary: four SIMDs per CU)."""
import collections, csv, glob, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob('gpurun_out/final_sq/*counter_collection.csv')[0])):
  k = r['Kernel_Name'].split('(')[0]
  if 'wn_' in k:
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
cols = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES',
        'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS']
rows = []
for k, d in acc.items():
  n = max(len(v) for v in d.values())
  avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else 0.0) for c in cols}
  wc = avg['SQ_WAVE_CYCLES'] or 1.0
  rows.append((avg['SQ_WAVE_CYCLES'], k, n, avg, avg['SQ_WAIT_ANY'] / wc, avg['SQ_VALU_MFMA_BUSY_CYCLES'] / wc / 4))
rows.sort(reverse=True)
with open(f'profiles/{tag}_train_pmc_sq.csv', 'w') as f:
  f.write('kernel,calls,' + ','.join(cols) + ',wait_any_frac,mfma_busy_per_wave_cycle\n')
  for _, k, n, avg, wf, mf in rows:
    f.write(f'"{k}",{n},' + ','.join(f'{avg[c]:.0f}' for c in cols) + f',{wf:.3f},{mf:.3f}\n')
print(open(f'profiles/{tag}_train_pmc_sq.csv').read())
