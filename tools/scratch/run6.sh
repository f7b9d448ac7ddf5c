set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c/pytest.log 2>&1 || { tail -40 gpurun_out/r03_c/pytest.log; exit 1; }
tail -3 gpurun_out/r03_c/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_c/bench.log 2>&1
tail -1 gpurun_out/r03_c/bench.log | cut -c1-250
FUSE=0 python - <<'PY' > gpurun_out/r03_c/bench_nofuse.log 2>&1
import runpy, sys
from audio_visual_deepfake_detection_amd import ops
ops.FUSE_MLP = False
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
runpy.run_path("bench.py", run_name="__main__")
PY
tail -1 gpurun_out/r03_c/bench_nofuse.log | cut -c1-250
timeout -k 10 300 python tools/pwgemm_bench.py --clock 2>&1 | tee gpurun_out/r03_c/pw_clock.log
