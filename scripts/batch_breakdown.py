"""Launch sequence of one batched multigrid-PCG iteration from a rocprofv3 kernel trace (kb_* kernels):
median duration per position in the most common sequence between two kb_update launches.
    python scripts/batch_breakdown.py <run_kernel_trace.csv>"""
import collections, csv, re, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(k[b]?_[a-z_0-9]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:30]


names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("kb_update")]
seqs = collections.Counter(tuple(names[a:b]) for a, b in zip(idx, idx[1:]))
best, cnt = seqs.most_common(1)[0]
durs = [[] for _ in best]
spans = []
for a, b in zip(idx, idx[1:]):
    if tuple(names[a:b]) != best:
        continue
    ok = all(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]) > 3000 or names[i] == "kb_reduce" for i in range(a, b))
    if not ok:
        continue
    for k, i in enumerate(range(a, b)):
        durs[k].append((int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3)
    spans.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
print(f"{len(spans)} iterations share the most common launch sequence ({len(best)} kernels)")
tot = 0.0
for k, n in enumerate(best):
    med = statistics.median(durs[k]) if durs[k] else float("nan")
    tot += med
    print(f"  {n:40s} median {med:7.2f} us")
print(f"sum of medians {tot:.1f} us; median start-to-start {statistics.median(spans):.1f} us")
