#!/bin/bash
# kernel-trace timeline of one replayed iteration in the one-rank forced sharded mode (MGGAN_FORCE_DIST=1):
#   gpurun -- 'bash tools/trace_forced_dist.sh c3'  -> gpurun_out/forced_<cfg>_{kernel_stats,iteration_timeline}.txt
CFG=${1:-c2}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_fd
MGGAN_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_fd -o p -- python $R/bench.py --steps 30 --warmup 10 --config $CFG --also= --no-cpu-baseline --no-floor > $OUT/forced_${CFG}_bench.log 2>&1
DB=$(ls /tmp/prof_fd/*.db | head -1)
cd $R && python tools/rocprof_summary.py $DB > $OUT/forced_${CFG}_kernel_stats.txt && python tools/iter_trace.py $DB 3 15 > $OUT/forced_${CFG}_iteration_timeline.txt
tail -2 $OUT/forced_${CFG}_iteration_timeline.txt
