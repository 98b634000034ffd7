#!/bin/bash
# The measurement recipe behind profiles/: run on the GPU box from the repo root, e.g.
#   gpurun -- 'bash tools/profile_round.sh r02 c2; bash tools/profile_round.sh r02 c3'
# writes gpurun_out/<tag>_<cfg>_{kernel_stats,iteration_timeline,time_marks,pmc_mfma,pmc_lds}.txt,
# gpurun_out/hbm_traffic_<cfg>.json and gpurun_out/mfma_util_<cfg>.json (copy them to profiles/).
# --kernel-trace and every --pmc pass are separate runs (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass).
TAG=${1:-r02}
CFG=${2:-c2}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w /tmp/prof_m /tmp/prof_l
COMMON="--config $CFG --also= --no-cpu-baseline --no-floor"
BENCH="python $R/bench.py --steps 40 --warmup 10 $COMMON"
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_kt -o p -- $BENCH > $OUT/${TAG}_${CFG}_bench_under_trace.log 2>&1
DB=$(ls /tmp/prof_kt/*.db | head -1)
(cd $R && python tools/rocprof_summary.py $DB > $OUT/${TAG}_${CFG}_kernel_stats.txt && python tools/iter_trace.py $DB 3 15 > $OUT/${TAG}_${CFG}_iteration_timeline.txt)
PMCB="python $R/bench.py --steps 6 --warmup 2 $COMMON"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d /tmp/prof_f -o p -- $PMCB > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d /tmp/prof_w -o p -- $PMCB > /dev/null 2>&1
(cd $R && python tools/hbm_traffic.py $(ls /tmp/prof_f/*.db | head -1) $(ls /tmp/prof_w/*.db | head -1) > $OUT/hbm_traffic_${CFG}.json)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format rocpd -d /tmp/prof_m -o p -- $PMCB > $OUT/${TAG}_${CFG}_pmc_mfma.log 2>&1
(cd $R && python tools/pmc_table.py $(ls /tmp/prof_m/*.db | head -1) > $OUT/${TAG}_${CFG}_pmc_mfma.txt; python tools/mfma_util.py $(ls /tmp/prof_m/*.db | head -1) > $OUT/mfma_util_${CFG}.json)
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS --output-format rocpd -d /tmp/prof_l -o p -- $PMCB > $OUT/${TAG}_${CFG}_pmc_lds.log 2>&1
(cd $R && python tools/pmc_table.py $(ls /tmp/prof_l/*.db | head -1) > $OUT/${TAG}_${CFG}_pmc_lds.txt)
cd $R
case $CFG in c2) SH="64 20 4";; c3) SH="256 32 8";; *) SH="32 3 1";; esac
python tools/time_marks.py $SH > $OUT/${TAG}_${CFG}_time_marks.txt 2>&1
tail -n 3 $OUT/${TAG}_${CFG}_kernel_stats.txt; tail -n 2 $OUT/${TAG}_${CFG}_iteration_timeline.txt; head -c 300 $OUT/hbm_traffic_${CFG}.json; tail -n 5 $OUT/${TAG}_${CFG}_time_marks.txt
