set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_f
python -m pytest tests -m gpu -x -q > gpurun_out/r03_f/pytest.log 2>&1 || { tail -40 gpurun_out/r03_f/pytest.log; exit 1; }
tail -2 gpurun_out/r03_f/pytest.log
timeout -k 10 900 python tools/pw_insitu.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_f/pw_insitu.log | grep -v "^  ("
