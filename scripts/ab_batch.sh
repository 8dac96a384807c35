#!/bin/bash
# usage (GPU box, repo root): bash scripts/ab_batch.sh "<env assignments>" ...   e.g.  bash scripts/ab_batch.sh "" "HEATFLOW_BATCH_LDS=0"
# The C5 sweep (64 kappa points, one GPU) per environment variant, with 1 and 2 batched loops in flight.
for v in "$@"; do
  for c in 1 2; do
    env $v python bench.py --workload sweep64 --sweep-concurrent $c --cpu-farm-points 0 --batch-roofline 0 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%-40s loops in flight %d: %.3e DOF-updates/s, wall %.3f s, whole call %.3f s, iters/step %.2f' % (sys.argv[1] or '(default)', int(sys.argv[2]), d['value'], c['wall_s'], c['whole_call_s'], c['pcg_iters_per_step_mean']))" "$v" $c
  done
done
