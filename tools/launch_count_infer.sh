#!/bin/bash
# kernel launches per captured inference launch (8 clips): difference of two kernel traces
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in 10 30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lci_$n -- python3 bench.py --mode infer --steps $n --warmup 0 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("lci_10"), load("lci_30")
rows = []
for k in b:
    n = (b[k][0] - a.get(k, (0, 0))[0]) / 20.0
    t = (b[k][1] - a.get(k, (0, 0))[1]) / 20.0 / 1e3
    if n > 0:
        rows.append((t, n, k[:120]))
print(f"launches: {sum(r[1] for r in rows):.1f}; kernel time {sum(r[0] for r in rows):.1f} us")
with open("gpurun_out/launch_breakdown_infer.txt", "w") as f:
    for t, n, k in sorted(rows, reverse=True):
        f.write(f"{n:6.1f} launches  {t:8.1f} us/step  {k}\n")
PY
rm -rf gpurun_out/lci_10 gpurun_out/lci_30
