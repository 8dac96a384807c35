#!/bin/bash
# usage (GPU box, repo root): bash scripts/trace_batch.sh <tag> "<env assignments>" ...
# One batch of 8 kappa points on the stock mesh (scripts/batch_probe.py, 30 steps) under rocprofv3 --kernel-trace per
# environment variant -> the launches of one batched multigrid-PCG iteration with median durations.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  D=$O/trace_${tag}_$i
  for kv in $v; do export "$kv"; done
  timeout -k 10 300 rocprofv3 --kernel-trace -d $D -o run --output-format csv -- python3 $R/scripts/batch_probe.py 1.0 ${NV:-8} 30 1 > $O/trace_${tag}_$i.log 2>&1 || { echo "variant '$v' failed"; tail -5 $O/trace_${tag}_$i.log; exit 1; }
  for kv in $v; do unset "${kv%%=*}"; done
  echo "== ${v:-(default)}"
  python3 $R/scripts/batch_breakdown.py $(find $D -name 'run_kernel_trace.csv' | head -1)
  rm -rf $D
done
