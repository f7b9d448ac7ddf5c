"""Print the top rows of a rocprofv3 *_kernel_stats.csv.  python tools/kstats.py <dir-or-csv> [steps] [n]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[-1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{p}: {len(rows)} kernels, {tot/1e6:.2f} ms total, {tot/1e6/steps:.3f} ms/step, {sum(int(r['Calls']) for r in rows)/steps:.0f} launches/step")
for r in rows[:n]:
    print(f"{r['Name'][:96]:96s} n/step={int(r['Calls'])/steps:7.1f} avg_us={float(r['AverageNs'])/1e3:8.2f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} {float(r['Percentage']):5.2f}%")
