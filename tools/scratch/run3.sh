set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_b
python tools/launch_floor.py 2>&1 | tee gpurun_out/r03_b/launch_floor.log
echo "--- with DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python tools/launch_floor.py 2>&1 | tee -a gpurun_out/r03_b/launch_floor.log
echo "--- with HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=1 python tools/launch_floor.py 2>&1 | tee -a gpurun_out/r03_b/launch_floor.log
