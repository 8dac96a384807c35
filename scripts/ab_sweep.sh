#!/bin/bash
# A/B of the C5 sweep on one box: bash scripts/ab_sweep.sh "<env settings>" ...  (each argument = one variant; 1 and 2 loops in flight)
O=gpurun_out/ab_sweep.txt
: > $O
for v in "$@"; do
  for conc in 1 2; do
    env $v python bench.py --workload sweep64 --sweep-concurrent $conc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v conc=$conc', round(d['value']/1e8,3), 'e8 DOF-updates/s', round(d['config']['wall_s'],3), 's', d['config']['pcg_iters_per_step_mean'])" >> $O || exit 1
  done
done
cat $O
