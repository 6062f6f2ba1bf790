"""Per-kernel summary of a rocprofv3 rocpd database (gpurun_out/<dir>/<name>_results.db), next to the
committed profiles/r01_bench_kernel_stats.csv averages."""
import csv, glob, os, sqlite3, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(root, 'gpurun_out/*/*_results.db')), key=os.path.getmtime)[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc limit ?", (top,)).fetchall()
old = {}
ref = os.path.join(root, 'profiles/r01_bench_kernel_stats.csv')
if os.path.exists(ref):
  for r in csv.DictReader(open(ref)):
    old[r['Name'][:60]] = float(r['AverageNs']) / 1e3
print(db)
for name, n, avg, tot in rows:
  o = old.get(name[:60])
  print(name[:60].ljust(60), str(n).rjust(5), f'{avg:9.1f} us {tot:8.2f} ms', f'| committed avg {o:.1f}' if o else '')
