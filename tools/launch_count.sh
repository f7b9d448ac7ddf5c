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
PY
rm -rf gpurun_out/lc_10 gpurun_out/lc_30
