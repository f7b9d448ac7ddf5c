set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b
timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | tee gpurun_out/r03_b/pw_clock.log
