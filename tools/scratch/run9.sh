cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_e
python tools/launch_floor.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_e/floor_plain.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_e/floortrace -- python3 $GRAFT_REPO_ROOT/tools/launch_floor.py 2>&1 | grep "us per launch" | tee $GRAFT_REPO_ROOT/gpurun_out/r03_e/floor_prof.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03_e/floortrace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("scale_rows", "ln_fwd", "keep_from", "maxpool", "mask_pyramid")):
        print(r["Name"][:50], "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"])
PY
rm -rf gpurun_out/r03_e/floortrace
