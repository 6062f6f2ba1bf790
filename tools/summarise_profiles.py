"""profiles/<tag>_*.csv from the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_<tag>/):
   <tag>_bench_kernel_stats.csv                    kernel trace of `python bench.py` (configs[1])
   <tag>_{train,cfg4}_kernel_stats.csv             per-kernel summary of the profiling drivers (4 training steps each)
   <tag>_gen128_kernel_stats.csv                   queued generation on the configs[3] network (tools/prof_gen128.py: 300 samples, batch 8)
   <tag>_{train,cfg4}_pmc_hbm_traffic.csv          FETCH_SIZE x 2 (gfx950: wide coalesced reads report half) + WRITE_SIZE, per launch
   <tag>_{train,cfg4}_pmc_sq.csv                   SQ counters per launch and three derived columns:
       wait_any_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES (both in quad-cycles, summed over waves): share of wave lifetime parked
       mfma_pipe_util  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 matrix pipes x SQ_BUSY_CYCLES / 32 shader engines): busy share of
                         the chip's 256 x 4 matrix pipes over the kernel's duration (MFMA_BUSY counts cycles, 32 per
                         v_mfma_f32_32x32x16_f16; SQ_BUSY_CYCLES is summed over the 32 shader engines)
       mfma_busy_per_wave_quadcycle = SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES / 4: the same cycles per WAVE lifetime; with
                         w waves per SIMD the pipe's utilisation is about w times this (the round-3 files carried only this
                         column, as "mfma_busy_per_wave_cycle")"""
import collections, csv, glob, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
out = os.path.join(root, 'gpurun_out', f'prof_{tag}')
dest = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, 'profiles')      # (on the GPU box: a directory under gpurun_out/)
os.makedirs(dest, exist_ok=True)


def find(d, pat):
  fs = glob.glob(os.path.join(out, d, '**', pat), recursive=True)
  return sorted(fs, key=os.path.getmtime)[-1] if fs else None


def agg(path, names):
  a = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] in names:
      a[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
  return a


for name, dst in (('bench_ks', 'bench'), ('train_ks', 'train'), ('cfg4_ks', 'cfg4'), ('gen128_ks', 'gen128')):
  f = find(name, '*kernel_stats.csv')
  if f:
    shutil.copy(f, os.path.join(dest, f'{tag}_{dst}_kernel_stats.csv'))
    print('kernel stats', name, '->', f'profiles/{tag}_{dst}_kernel_stats.csv')
for cfg in ('train', 'cfg4'):
  ff, fw, fs = find(cfg + '_f', '*counter_collection.csv'), find(cfg + '_w', '*counter_collection.csv'), find(cfg + '_sq', '*counter_collection.csv')
  if ff and fw:
    f, w = agg(ff, {'FETCH_SIZE'}), agg(fw, {'WRITE_SIZE'})
    rows = ['kernel,calls,FETCH_SIZE_KB_avg_raw,FETCH_bytes_x2_corrected,WRITE_SIZE_KB_avg,WRITE_bytes,total_bytes_corrected']
    tot = 0.0
    for k in sorted(f, key=lambda k: -sum(f[k]['FETCH_SIZE'])):
      if 'wn_' not in k:
        continue
      fa = sum(f[k]['FETCH_SIZE']) / len(f[k]['FETCH_SIZE'])
      wv = w.get(k, {}).get('WRITE_SIZE', [0.0])
      wa = sum(wv) / max(1, len(wv))
      n = len(f[k]['FETCH_SIZE'])
      tot += n * (fa * 2048 + wa * 1024)
      rows.append(f'"{k}",{n},{fa:.0f},{fa * 2048:.0f},{wa:.0f},{wa * 1024:.0f},{fa * 2048 + wa * 1024:.0f}')
    rows.append(f'"TOTAL bytes over all profiled launches (4 training steps + warm-up)",,,,,,{tot:.0f}')
    open(os.path.join(dest, f'{tag}_{cfg}_pmc_hbm_traffic.csv'), 'w').write('\n'.join(rows) + '\n')
    print('\n'.join(rows[:14]))
  if fs:
    cols = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES',
            'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS']
    a = agg(fs, set(cols))
    rows = []
    for k, d in a.items():
      if 'wn_' not in k:
        continue
      n = max(len(v) for v in d.values())
      avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else 0.0) for c in cols}
      wc = avg['SQ_WAVE_CYCLES'] or 1.0
      pipe = avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * avg['SQ_BUSY_CYCLES'] / 32.0) if avg['SQ_BUSY_CYCLES'] else 0.0
      rows.append((avg['SQ_WAVE_CYCLES'] * n, k, n, avg, avg['SQ_WAIT_ANY'] / wc, pipe, avg['SQ_VALU_MFMA_BUSY_CYCLES'] / wc / 4))
    rows.sort(reverse=True)
    with open(os.path.join(dest, f'{tag}_{cfg}_pmc_sq.csv'), 'w') as fo:
      fo.write('kernel,calls,' + ','.join(cols) + ',wait_any_frac,mfma_pipe_util,mfma_busy_per_wave_quadcycle\n')
      for _, k, n, avg, wf, pu, mf in rows:
        fo.write(f'"{k}",{n},' + ','.join(f'{avg[c]:.0f}' for c in cols) + f',{wf:.3f},{pu:.3f},{mf:.3f}\n')
    print(open(os.path.join(dest, f'{tag}_{cfg}_pmc_sq.csv')).read()[:1500])
