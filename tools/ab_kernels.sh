#!/bin/bash
# A/B of kernel variants on ONE box: the in-tree library against other builds of it (MGGAN_HIP_LIB), same bench command
# under rocprofv3 --kernel-trace, per-kernel averages side by side (two boxes differ by 5-10 %, two runs on one by 0.2 us).
#   gpurun -- 'bash tools/ab_kernels.sh "scratch/libvariant.so [more.so ...]" c3 "conv1_wgrad_kernel|image_gram"'
# A variant: copy a .hip file, edit, `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -c` it and link it
# with the other objects of mg-gan_amd/csrc/build/ into a .so under scratch/ (untracked, travels with gpurun).
R=$(pwd); cd /tmp && export TMPDIR=/tmp
for v in base $1; do
  rm -rf /tmp/prof_ab
  if [ $v = base ]; then unset MGGAN_HIP_LIB; else export MGGAN_HIP_LIB=$R/$v; fi
  rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_ab -o p -- python $R/bench.py --steps 30 --warmup 8 --config ${2:-c3} --also= --no-cpu-baseline --no-floor > /tmp/ab_$(basename $v).log 2>&1
  DB=$(ls /tmp/prof_ab/*.db | head -1)
  echo "== $v"; (cd $R && python tools/rocprof_summary.py $DB | grep -E "${3:-kernel}" | cut -c1-150)
done
