#!/bin/bash
# Iteration-level A/B on ONE box: bench.py (headline + configs[2]) alternating between the in-tree library and other builds
# of it (same C ABI, MGGAN_HIP_LIB), REPS times each:  gpurun -- 'bash tools/ab_iter.sh "scratch/lib_head.so" 3'
REPS=${2:-3}
for r in $(seq $REPS); do
  for v in base $1; do
    if [ $v = base ]; then unset MGGAN_HIP_LIB; else export MGGAN_HIP_LIB=$(pwd)/$v; fi
    python bench.py --no-floor --no-cpu-baseline --steps 40 ${3} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', [(c['workload'], c['ms_per_step']) for c in d['configs']])"
  done
done
