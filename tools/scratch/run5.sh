set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c
AVDF_PW_WPC=1 timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | tee gpurun_out/r03_c/pw_clock3_wpc1.log
timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | tee gpurun_out/r03_c/pw_clock3.log
timeout -k 10 600 python tools/pwgemm_bench.py --quick > gpurun_out/r03_c/pw_fwd5.log 2>&1 || { tail -30 gpurun_out/r03_c/pw_fwd5.log; exit 1; }
cat gpurun_out/r03_c/pw_fwd5.log
