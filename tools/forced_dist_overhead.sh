#!/bin/bash
# One-rank cost of the sharded launch mode (the collective hooks forced on with a single rank, MGGAN_FORCE_DIST=1) against
# the plain single-GPU graph, on ONE box:  gpurun -- 'bash tools/forced_dist_overhead.sh' -> gpurun_out/forced_dist.txt
OUT=gpurun_out/forced_dist.txt
mkdir -p gpurun_out
: > $OUT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', [(c['workload'], c['ms_per_step']) for c in d['configs']], d['config'].get('collective'), [c.get('exchanges_per_step') for c in d['configs']], d['config'].get('launch'))"; }
for rep in 1 2; do
  export MASTER_PORT=$((20000 + RANDOM % 20000))  # (a fixed port is still in TIME_WAIT when the next process binds it)
  python bench.py --no-floor --no-cpu-baseline --no-profile --steps 40 2>/dev/null | line "single-graph          " >> $OUT
  MGGAN_FORCE_DIST=1 python bench.py --no-floor --no-cpu-baseline --no-profile --steps 40 2>/dev/null | line "sharded,peer-mapped   " >> $OUT
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  MGGAN_FORCE_DIST=1 MGGAN_DEVICE_COMM=0 python bench.py --no-floor --no-cpu-baseline --no-profile --steps 40 2>/dev/null | line "sharded,rccl-graph    " >> $OUT
  export MASTER_PORT=$((20000 + RANDOM % 20000))
  MGGAN_FORCE_DIST=1 MGGAN_DEVICE_COMM=0 MGGAN_RCCL_GRAPH=0 python bench.py --no-floor --no-cpu-baseline --no-profile --steps 40 2>/dev/null | line "sharded,rccl-segments " >> $OUT
done
cat $OUT
