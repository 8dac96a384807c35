#!/bin/bash
# usage (GPU box, repo root): bash scripts/evidence_round.sh <tag>
# The non-profiler evidence kept under profiles/: default bench line, 2-rank gloo rehearsals of both workloads,
# multigrid-PCG beyond the Infinity Cache, connectivity-table sharing, robustness sweep, operator table.
tag=$1
O=gpurun_out
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || exit 1
HEATFLOW_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --sweep-points 0 --cpu-steps 0 --hbm-scale 0 --jacobi-steps 0 > $O/${tag}_bench_gloo2.json 2> $O/${tag}_bench_gloo2.err || exit 1
HEATFLOW_BENCH_BACKEND=gloo python bench.py --gpus 2 --workload sweep64 > $O/${tag}_sweep64_gloo2.json 2> $O/${tag}_sweep64_gloo2.err || exit 1
{ python scripts/gpu_probe.py 0.215 12 3 1 | grep -v "assemble mode" ; python scripts/gpu_probe.py 0.1075 12 3 1 | grep -v "assemble mode"; } > $O/${tag}_amg_large.txt 2>&1 || exit 1
python scripts/pattern_share_probe.py 0.43 > $O/${tag}_pattern_share.txt 2>&1 || exit 1
python scripts/amg_robustness.py > $O/${tag}_amg_robustness.txt 2>&1 || exit 1
HEATFLOW_DEBUG=1 python scripts/gpu_probe.py 0.43 8 3 1 2>&1 | grep "\[amg\]" | awk '!seen[$0]++' > $O/${tag}_amg_operators.txt
tail -3 $O/${tag}_amg_large.txt
