set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c
python tools/launch_floor.py 2>&1 | tee gpurun_out/r03_c/launch_floor.log
