#!/bin/bash
# A/B of multigrid set-up parameters on one box: bash scripts/ab_amg.sh "<env settings>" ...   (each argument = one variant)
O=gpurun_out/ab_amg.txt
: > $O
for v in "$@"; do
  env $v python bench.py --device-warmup-s 1 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v', '| ms/step', round(d['ms_per_step'],4), 'iters', round(c['pcg_iters_per_step_mean'],2), 'max', c['pcg_iters_per_step_max'], 'setup', c['rank0_setup_s'])" >> $O || exit 1
done
cat $O
