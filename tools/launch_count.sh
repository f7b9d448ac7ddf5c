#!/bin/bash
# exact kernel launches per captured training step: difference of two kernel traces (10 vs 30 timed steps)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for n in 10 30; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lc_$n -- python3 bench.py --steps $n --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("lc_10"), load("lc_30")
dl = sum(b[k][0] - a.get(k, (0, 0))[0] for k in b) / 20.0
dt = sum(b[k][1] - a.get(k, (0, 0))[1] for k in b) / 20.0 / 1e6
aten = sum(b[k][0] - a.get(k, (0, 0))[0] for k in b if k.startswith("void at::") or "rocclr" in k) / 20.0
print(f"launches per step: {dl:.1f} (ATen / runtime copies: {aten:.1f}); kernel time per step: {dt:.3f} ms")
import re
rows = []
for k in b:
    n = (b[k][0] - a.get(k, (0, 0))[0]) / 20.0
    t = (b[k][1] - a.get(k, (0, 0))[1]) / 20.0 / 1e3
    if n > 0:
        m = re.search(r"(conv_\w+_kernel<[^>]*>|[\w:]+_kernel\w*|__amd_rocclr_\w+|\w+Functor\w*|\w+)", k)
        rows.append((n, t, k[:110]))
with open("gpurun_out/launch_breakdown.txt", "w") as f:
    for n, t, k in sorted(rows, reverse=True):
        f.write(f"{n:6.1f} launches  {t:8.1f} us/step  {k}\n")
PY
rm -rf gpurun_out/lc_10 gpurun_out/lc_30
