"""Per-kernel breakdown of one multigrid-PCG iteration from a rocprofv3 --kernel-trace CSV.

usage: python scripts/iter_breakdown.py <..._kernel_trace.csv> [iteration index]
Prints the launches between two consecutive k_pcg_update_amg dispatches of a mid-solve iteration and the
median duration of each position over all iterations with the same launch sequence.
"""
import csv, re, sys, statistics, collections

def short(n):
    m = re.search(r"k_\w+(<[^>]*>)?", n)
    return m.group(0) if m else n[:25]

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pcg_update_amg" in r["Kernel_Name"]]
seqs = collections.defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    key = tuple((short(r["Kernel_Name"]), r["Grid_Size_X"]) for r in rows[a:b])
    seqs[key].append((a, b))
key, spans = max(seqs.items(), key=lambda kv: len(kv[1]))
print(f"{len(spans)} iterations share the most common launch sequence ({len(key)} kernels)")
durs = [[] for _ in key]
tot = []
for a, b in spans:
    full = True
    for j, r in enumerate(rows[a:b]):
        durs[j].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
s = 0.0
for (nm, grid), d in zip(key, durs):
    med = statistics.median(d) / 1e3
    s += med
    print(f"  {nm:22s} grid {grid:>8s}  median {med:7.2f} us")
print(f"sum of medians {s:.1f} us; median start-to-start {statistics.median(tot)/1e3:.1f} us")
