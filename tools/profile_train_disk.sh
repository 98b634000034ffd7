#!/bin/bash
# Kernel statistics of train() on an on-disk dataset (bench.py: train_loop_disk_leg): the augmented crop kernels beside the
# replayed iterations.  Run on the GPU box from the repo root:
#   gpurun -- 'bash tools/profile_train_disk.sh r06'   -> gpurun_out/<tag>_train_disk_kernel_stats.txt
TAG=${1:-r06}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_td
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_td -o p -- python -c "
import sys
sys.path.insert(0, '$R'); sys.path.insert(0, '$R/mg-gan_amd')
import torch, bench
class A: pass
r = bench.train_loop_disk_leg(A(), torch.device('cuda:0'), 1.32, frames=980, epochs=5)
print(r['ms_per_step'], r['ms_per_step_by_epoch'])
" > $OUT/${TAG}_train_disk_under_trace.log 2>&1
DB=$(ls /tmp/prof_td/*.db | head -1)
(cd $R && python tools/rocprof_summary.py $DB > $OUT/${TAG}_train_disk_kernel_stats.txt)
head -n 14 $OUT/${TAG}_train_disk_kernel_stats.txt | cut -c1-150; grep -a "crop_patches\|pad_batch" $OUT/${TAG}_train_disk_kernel_stats.txt | cut -c1-150
