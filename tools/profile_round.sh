#!/bin/bash
# Collect the rocprofv3 evidence of one bench configuration on the GPU box (run from the repo root):
#   tools/profile_round.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/{stats,fetch,write,mfma}
# kernel trace + stats (un-instrumented durations), then three separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; counters are never combined with a trace domain other than the kernel trace).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also "$@" > $OUT/bench_line.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also "$@" > /dev/null 2>&1
STATS=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
# forward + backward passes in the trace = calls of the optimiser kernel (one per bench step: warm-up, timed and roofline
# passes) + 3 (TrainStep's probe step and the two capture warm-ups run the whole step except the optimiser).  Exact
# per-step launch counts come from tools/launch_count.sh (difference of two traces).
NSTEP=$(python3 -c "import csv,sys; print(3 + max([int(r['Calls']) for r in csv.DictReader(open(sys.argv[1])) if 'adamw_ema_seg_kernel' in r['Name']] or [1]))" $STATS)
python3 tools/kstats.py $STATS $NSTEP 60 > $OUT/kernel_stats.md
python3 tools/pmc_summary.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 tools/pmc_mfma.py $OUT/mfma $STATS $OUT/pmc_mfma_util.json > $OUT/pmc_mfma_util.txt
cp $STATS $OUT/kernel_stats.csv
python3 tools/hbm_table.py $OUT/ > $OUT/hbm_kernels.md
tail -1 $OUT/bench_line.log | cut -c1-300
# the raw per-dispatch CSVs are large: keep the summaries only
grep "^{" $OUT/bench_line.log | tail -1 > $OUT/bench_line.json
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma
ls -la $OUT
