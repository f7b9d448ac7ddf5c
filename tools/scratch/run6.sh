set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_e
python -m pytest tests -m gpu -x -q > gpurun_out/r03_e/pytest.log 2>&1 || { tail -40 gpurun_out/r03_e/pytest.log; exit 1; }
tail -3 gpurun_out/r03_e/pytest.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_e/bench_a$i.log 2>&1
echo "fused-fpn $(tail -1 gpurun_out/r03_e/bench_a$i.log | cut -c1-90)"
AVDF_FUSE_FPN=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_e/bench_b$i.log 2>&1
echo "per-level $(tail -1 gpurun_out/r03_e/bench_b$i.log | cut -c1-90)"
done
python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
AVDF_FUSE_FPN=0 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
