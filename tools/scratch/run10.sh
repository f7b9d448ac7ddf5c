cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_f
timeout -k 10 1100 python tools/plan_search.py --auto 6 --out gpurun_out/r03_f/plan_search_bwd.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_f/plan_search_bwd.log | grep -v "^  (" 
