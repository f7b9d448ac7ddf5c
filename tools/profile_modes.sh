#!/bin/bash
# rocprofv3 kernel trace + stats of the inference and raw-media modes (kernel_stats summaries -> gpurun_out/prof_modes/)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/prof_modes; mkdir -p $OUT; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/infer -- python3 bench.py --mode infer --steps 40 --warmup 10 --no-cpu-baseline > $OUT/infer_bench_line.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python3 bench.py --mode e2e --steps 3 --warmup 1 --no-cpu-baseline > $OUT/e2e_bench_line.log 2>&1
for m in infer e2e; do
  S=$(find $OUT/$m -name "*kernel_stats.csv" | head -1)
  cp $S $OUT/${m}_kernel_stats.csv
  python3 tools/kstats.py $S 50 30 > $OUT/${m}_kernel_stats.md || true
  grep "^{" $OUT/${m}_bench_line.log | tail -1 > $OUT/${m}_bench_line.json
  rm -rf $OUT/$m
done
ls -la $OUT
