cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_c
run() { echo "== $1"; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*, "unit": "clips/s".\{0,60\}' ; }
run "X=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_ENABLE_INTERRUPT=0"
run "X=2"
