set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_g
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "pipelined_pointwise" 2>&1 | tail -3
AVDF_PW_WPC=1 timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_g/pw_clock_pipe_wpc1.log
timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_g/pw_clock_pipe.log
timeout -k 10 600 python tools/pwgemm_bench.py > gpurun_out/r03_g/pw_fwd_pipe.log 2>&1 || { tail -30 gpurun_out/r03_g/pw_fwd_pipe.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r03_g/pw_fwd_pipe.log | cut -c1-420
