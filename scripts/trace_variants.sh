#!/bin/bash
# usage (on the GPU box): scripts/trace_variants.sh name "ENV=.. ENV=.." ...   -> iteration breakdown per variant
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  out=$GRAFT_REPO_ROOT/gpurun_out/var_$name
  ( export $envs; timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_probe.py 0.43 16 0 1 > $out.log 2>&1 ) || exit 1
  echo "== $name ($envs)"; grep "ms/step" $out.log | cut -c1-60
  python3 $GRAFT_REPO_ROOT/scripts/iter_breakdown.py $out/run_kernel_trace.csv
done
