set -e -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_h
timeout -k 10 600 python tools/scratch/pw_combo.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_h/pw_combo.log
