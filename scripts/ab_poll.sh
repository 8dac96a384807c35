#!/bin/bash
# A/B on one box: polled one-test-ahead loops (default) against bursts + copy-back (HEATFLOW_POLL=0)
# usage: bash scripts/ab_poll.sh  -> gpurun_out/ab_poll.txt
O=gpurun_out/ab_poll.txt
: > $O
for rep in 1 2; do
  for poll in 1 0; do
    HEATFLOW_POLL=$poll python bench.py --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 poll=$poll', round(d['ms_per_step'],4), 'ms/step', d['config']['pcg_iters_per_step_mean'])" >> $O || exit 1
  done
done
for conc in 1 2; do
  for poll in 1 0; do
    HEATFLOW_POLL=$poll python bench.py --workload sweep64 --sweep-concurrent $conc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sweep64 conc=$conc poll=$poll', d['value'], d['config'].get('wall_s'))" >> $O || exit 1
  done
done
cat $O
