#!/bin/bash
# Kernel trace only (the first step of tools/profile_round.sh): per-kernel statistics and one iteration's timeline of a
# configuration under rocprofv3 --kernel-trace.   gpurun -- 'bash tools/kernel_trace.sh r03x c2'
#   -> gpurun_out/<tag>_<cfg>_{kernel_stats,iteration_timeline}.txt
TAG=$1; CFG=$2
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
BENCH="python $R/bench.py --steps 40 --warmup 10 --config $CFG --also= --no-cpu-baseline --no-floor"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_kt -o p -- $BENCH > $OUT/${TAG}_${CFG}_bench_under_trace.log 2>&1
DB=$(ls /tmp/prof_kt/*.db | head -1)
(cd $R && python tools/rocprof_summary.py $DB > $OUT/${TAG}_${CFG}_kernel_stats.txt && python tools/iter_trace.py $DB 3 15 > $OUT/${TAG}_${CFG}_iteration_timeline.txt)
tail -n 1 $OUT/${TAG}_${CFG}_iteration_timeline.txt
