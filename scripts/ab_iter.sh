#!/bin/bash
# usage: bash scripts/ab_iter.sh "<env settings>" ...   -> iteration breakdown (rocprofv3 kernel trace) of the C3 run per variant, one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/ab_iter.txt
for v in "$@"; do
  rm -rf $O/prof_abi
  env $v timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_abi -o run --output-format csv -- python3 $R/bench.py --traffic none --steps 20 --warmup 5 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 --profile-steps 0 --device-warmup-s 0.5 > /dev/null 2> $O/prof_abi.err || exit 1
  echo "== $v" >> $O/ab_iter.txt
  python3 $R/scripts/iter_breakdown.py $O/prof_abi/run_kernel_trace.csv | tail -12 >> $O/ab_iter.txt
done
rm -rf $O/prof_abi
grep "==\|start-to-start" $O/ab_iter.txt
