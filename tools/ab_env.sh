#!/bin/bash
# Same-box A/B of environment knobs on the single-GPU bench (headline + configs[2]), alternating rounds:
#   gpurun -- 'bash tools/ab_env.sh 3 "MGGAN_PIPELINE=0" "MGGAN_PIPELINE=1" "MGGAN_PIPELINE=1 MGGAN_PIPE_AT=pm_bwd"'
#   -> gpurun_out/ab_env.txt   (one line per run: the knobs, ms per iteration of c2 and c3)
REPS=$1; shift
OUT=gpurun_out/ab_env.txt; mkdir -p gpurun_out; : > $OUT
for rep in $(seq 1 $REPS); do
  for knobs in "$@"; do
    env $knobs python bench.py --no-floor --no-cpu-baseline --no-profile --steps ${STEPS:-60} --warmup 10 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knobs', [(c['workload'], c['ms_per_step']) for c in d['configs']])" >> $OUT
  done
done
cat $OUT
