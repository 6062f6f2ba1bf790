"""Per-kernel averages of the counters in a rocprofv3 --pmc csv (…counter_collection.csv)."""
import collections, csv, glob, sys
path = glob.glob(sys.argv[1])[0]
pat = sys.argv[2] if len(sys.argv) > 2 else 'wn_'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
  k = r['Kernel_Name'].split('(')[0]
  if pat in k:
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', kv[1].get('GRBM_GUI_ACTIVE', [0])))):
  n = max(len(v) for v in d.values())
  print(k[:70], 'calls', n)
  print('   ', '  '.join(f'{c}={sum(v) / len(v):.4g}' for c, v in sorted(d.items())))
