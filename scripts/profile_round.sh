#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag> [c3|hbm|pmc_c3|pmc_hbm|amg] ...
# Collects the evidence kept under profiles/: rocprofv3 kernel-trace + stats of bench.py at C3 and at the
# HBM-resident 16M-DOF point, the multigrid iteration breakdown, and the FETCH_SIZE / WRITE_SIZE passes.
# Output goes to gpurun_out/prof_<tag>_*; copy the summaries into profiles/ afterwards.
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
C3ARGS="--traffic file --steps 20 --warmup 5 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 3 --device-warmup-s 0.5"
HBMARGS="--traffic file --scale 0.1075 --precond jacobi --steps 2 --warmup 5 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 --profile-steps 1 --device-warmup-s 0.5"
for what in "$@"; do
  case $what in
    c3)
      timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_c3 -o run --output-format csv -- python3 $R/bench.py $C3ARGS > $O/prof_${tag}_c3.json 2> $O/prof_${tag}_c3.err || exit 1
      python3 $R/scripts/iter_breakdown.py $O/prof_${tag}_c3/run_kernel_trace.csv > $O/prof_${tag}_c3_iteration_breakdown.txt
      cat $O/prof_${tag}_c3_iteration_breakdown.txt ;;
    hbm)
      timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_hbm -o run --output-format csv -- python3 $R/bench.py $HBMARGS > $O/prof_${tag}_hbm.json 2> $O/prof_${tag}_hbm.err || exit 1
      head -12 $O/prof_${tag}_hbm/run_kernel_stats.csv ;;
    pmc_c3)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 400 rocprofv3 --pmc $c -d $O/prof_${tag}_c3_$c -o run --output-format csv -- python3 $R/bench.py --traffic none --steps 6 --warmup 5 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 3 --profile-steps 0 > /dev/null 2> $O/prof_${tag}_c3_$c.err || exit 1
      done ;;
    pmc_hbm)
      for c in FETCH_SIZE WRITE_SIZE; do
        timeout -k 10 500 rocprofv3 --pmc $c -d $O/prof_${tag}_hbm_$c -o run --output-format csv -- python3 $R/bench.py --traffic none --scale 0.1075 --precond jacobi --steps 1 --warmup 5 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 --profile-steps 0 > /dev/null 2> $O/prof_${tag}_hbm_$c.err || exit 1
      done ;;
    amg)
      HEATFLOW_DEBUG=1 timeout -k 10 200 python3 $R/scripts/gpu_probe.py 0.43 8 3 1 > $O/prof_${tag}_amg.log 2>&1 || exit 1
      grep "\[amg\]" $O/prof_${tag}_amg.log ;;
  esac
done
