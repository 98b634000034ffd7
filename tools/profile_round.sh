#!/bin/bash
# The measurement recipe behind profiles/: run on the GPU box from the repo root, e.g.
#   gpurun -- 'bash tools/profile_round.sh r01'
# writes gpurun_out/<tag>_{kernel_stats,iteration_timeline,time_marks}.txt and gpurun_out/hbm_traffic.json
# (copy them to profiles/).  --kernel-trace and the two --pmc passes are separate runs.
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
BENCH="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_kt -o p -- $BENCH > $OUT/${TAG}_bench_under_trace.log 2>&1
DB=$(ls /tmp/prof_kt/*.db | head -1)
(cd $R && python tools/rocprof_summary.py $DB > $OUT/${TAG}_kernel_stats.txt && python tools/iter_trace.py $DB 3 15 > $OUT/${TAG}_iteration_timeline.txt)
PMCB="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/prof_f -o p -- $PMCB > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/prof_w -o p -- $PMCB > /dev/null 2>&1
(cd $R && python tools/hbm_traffic.py $(ls /tmp/prof_f/*.db | head -1) $(ls /tmp/prof_w/*.db | head -1) > $OUT/hbm_traffic.json)
cd $R
python tools/time_marks.py > $OUT/${TAG}_time_marks.txt 2>&1
tail -n 3 $OUT/${TAG}_kernel_stats.txt; tail -n 2 $OUT/${TAG}_iteration_timeline.txt; head -c 300 $OUT/hbm_traffic.json; tail -n 5 $OUT/${TAG}_time_marks.txt
