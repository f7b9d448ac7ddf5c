set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_a
python -m pytest tests -m gpu -x -q > gpurun_out/r03_a/pytest.log 2>&1 || { tail -30 gpurun_out/r03_a/pytest.log; exit 1; }
tail -3 gpurun_out/r03_a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_a/bench.log 2>&1
tail -1 gpurun_out/r03_a/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_a/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 14 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_a/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_timeline.py gpurun_out/r03_a/trace 8 > gpurun_out/r03_a/timeline.txt
rm -rf gpurun_out/r03_a/trace
tail -40 gpurun_out/r03_a/timeline.txt
