#!/bin/bash
# kernel-stats of two bench configurations side by side: tools/kstats_ab.sh <tag> "<args A>" "<args B>"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1
OUT=$R/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $2 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $3 > $OUT/b.log 2>&1
cp $(find $OUT/a -name "*kernel_stats.csv" | head -1) $OUT/a_kernel_stats.csv
cp $(find $OUT/b -name "*kernel_stats.csv" | head -1) $OUT/b_kernel_stats.csv
rm -rf $OUT/a $OUT/b
tail -1 $OUT/a.log; tail -1 $OUT/b.log
