#!/bin/bash
# usage (on the GPU box): scripts/amg_variants.sh name "ENV=.." ...  -> iterations and ms/step of a 64-step C3 run per variant
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  echo "== $name ($envs)"
  ( export $envs; HEATFLOW_DEBUG=1 timeout -k 10 300 python3 scripts/gpu_probe.py 0.43 64 0 1 2>&1 | grep "^amg\|run 64\|per PCG\|level 1 A\|level 1 GP\|level 0 P " | sed 's/np.int32(\([0-9]*\))/\1/g' | cut -c1-330 ) || exit 1
done
