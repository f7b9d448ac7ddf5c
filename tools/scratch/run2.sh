set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c
timeout -k 10 600 python tools/pwgemm_bench.py --b32 --quick > gpurun_out/r03_c/pw_fwd_b32.log 2>&1 || { tail -30 gpurun_out/r03_c/pw_fwd_b32.log; exit 1; }
cat gpurun_out/r03_c/pw_fwd_b32.log
