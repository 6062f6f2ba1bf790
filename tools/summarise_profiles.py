"""Turns the rocprofv3 outputs merged under gpurun_out/ into the small tracked summaries of profiles/:
   <tag>_bench_kernel_stats.csv   (rocprofv3 --kernel-trace --stats of `python bench.py`)
   <tag>_train_pmc_hbm_traffic.csv (separate --pmc FETCH_SIZE / WRITE_SIZE passes, per-launch averages;
                                    FETCH_SIZE doubled: gfx950 reports half of wide coalesced reads)"""
import collections, csv, glob, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ks = (glob.glob('gpurun_out/final_ks/*/*kernel_stats.csv') + glob.glob('gpurun_out/final_ks/*kernel_stats.csv'))
if ks:
  ks.sort(key=os.path.getmtime)
  shutil.copy(ks[-1], f'profiles/{tag}_bench_kernel_stats.csv')


def agg(path, cname):
  rows = list(csv.DictReader(open(glob.glob(path)[0])))
  a = collections.defaultdict(list)
  for r in rows:
    if r['Counter_Name'] == cname:
      a[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
  return a


f = agg('gpurun_out/final_f/*counter_collection.csv', 'FETCH_SIZE')
w = agg('gpurun_out/final_w/*counter_collection.csv', 'WRITE_SIZE')
out = ['kernel,calls,FETCH_SIZE_KB_avg_raw,FETCH_bytes_x2_corrected,WRITE_SIZE_KB_avg,WRITE_bytes,total_bytes_corrected']
for k in f:
  if 'wn_' not in k:
    continue
  fa = sum(f[k]) / len(f[k])
  wa = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
  out.append(f'"{k}",{len(f[k])},{fa:.0f},{fa * 2048:.0f},{wa:.0f},{wa * 1024:.0f},{fa * 2048 + wa * 1024:.0f}')
open(f'profiles/{tag}_train_pmc_hbm_traffic.csv', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
