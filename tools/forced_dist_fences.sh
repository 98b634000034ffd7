#!/bin/bash
# One-rank cost of the sharded mode (peer-mapped transport) with the formal fence protocol and with MGGAN_COMM_FENCES=light,
# against the single-GPU graph, alternating on one box:  gpurun -- 'bash tools/forced_dist_fences.sh' -> gpurun_out/forced_dist_fences.txt
OUT=gpurun_out/forced_dist_fences.txt; mkdir -p gpurun_out; : > $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', [(c['workload'], c['ms_per_step']) for c in d['configs']])"; }
B="python bench.py --no-floor --no-cpu-baseline --no-profile --steps 60 --warmup 10"
for rep in 1 2 3; do
  $B 2>/dev/null | line "single-graph          " >> $OUT
  export MASTER_PORT=$((20000 + RANDOM % 20000)); MGGAN_FORCE_DIST=1 $B 2>/dev/null | line "peer-mapped, strict   " >> $OUT
  export MASTER_PORT=$((20000 + RANDOM % 20000)); MGGAN_FORCE_DIST=1 MGGAN_COMM_FENCES=light $B 2>/dev/null | line "peer-mapped, light    " >> $OUT
done
cat $OUT
