#!/bin/bash
# usage (GPU box, repo root): bash scripts/profile_batch.sh <tag> [trace|pmc] ...
# Evidence of the batched time loop (BASELINE C5) kept under profiles/<tag>_batch_*:
#   trace  rocprofv3 --kernel-trace --stats of `bench.py --traffic none --workload sweep64 --sweep-concurrent 1` -> kernel stats + the launches of
#          one batched multigrid-PCG iteration (scripts/batch_breakdown.py); the raw trace is deleted (tens of MB)
#   pmc    FETCH_SIZE / WRITE_SIZE passes (separate runs) of one batch of NV (default 16) on the stock mesh (scripts/batch_probe.py, 20 steps)
#          -> HBM bytes per launch of the kb_* kernels next to their algorithmic bytes (scripts/pmc_batch_summary.py)
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    trace)
      D=$O/prof_${tag}_batch_trace
      timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $D -o run --output-format csv -- python3 $R/bench.py --traffic none --workload sweep64 --sweep-concurrent 1 --cpu-farm-points 0 > $O/${tag}_batch_bench_under_rocprof.json 2> $O/${tag}_batch_trace.err || exit 1
      T=$(find $D -name 'run_kernel_trace.csv' | head -1)
      S=$(find $D -name 'run_kernel_stats.csv' | head -1)
      python3 $R/scripts/batch_breakdown.py $T > $O/${tag}_batch_iteration_breakdown.txt || exit 1
      cp $S $O/${tag}_batch_kernel_stats.csv
      rm -f $T
      cat $O/${tag}_batch_iteration_breakdown.txt ;;
    pmc)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 300 rocprofv3 --pmc $c -d $O/prof_${tag}_batch_$c -o run --output-format csv -- python3 $R/scripts/batch_probe.py 1.0 ${NV:-16} 20 1 > $O/${tag}_batch_pmc_$c.log 2> $O/${tag}_batch_pmc_$c.err || exit 1
      done
      python3 $R/scripts/pmc_batch_summary.py $O/prof_${tag}_batch_FETCH_SIZE $O/prof_${tag}_batch_WRITE_SIZE $O/${tag}_batch_pmc_traffic ${NV:-16} || exit 1
      rm -rf $O/prof_${tag}_batch_FETCH_SIZE $O/prof_${tag}_batch_WRITE_SIZE ;;
  esac
done
