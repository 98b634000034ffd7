#!/bin/bash
# Sweep one environment knob over values on ONE box:  bash tools/sweep_env.sh MGGAN_CNN_GRID "0 512 768 1024" [c2|c3]
K=$1; CFG=${3:-c3}
for v in $2; do
  if [ "$v" = "-" ]; then unset $K; else export $K=$v; fi
  echo "$K=$v $(python bench.py --no-floor --no-cpu-baseline --steps 40 --config $CFG --also= 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print([(c['workload'], c['ms_per_step']) for c in d['configs']])")"
done
unset $K
