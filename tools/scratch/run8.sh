set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_e
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_e/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_e/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py gpurun_out/r03_e/trace 8 > gpurun_out/r03_e/timeline.txt
rm -rf gpurun_out/r03_e/trace
grep "^# launches" gpurun_out/r03_e/timeline.txt
grep -n "fpn_\|gap_concat" gpurun_out/r03_e/timeline.txt
