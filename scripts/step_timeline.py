"""Timeline of one multigrid-PCG time step from a rocprofv3 kernel trace: every launch outside the steady
iteration pattern, every gap above 2 us, and the step's totals.

    python scripts/step_timeline.py <run_kernel_trace.csv> [step index]
"""
import csv, re, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(k[b]?_[a-z_0-9]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:30]


names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_proj_dots"]
steps = [(a, b) for a, b in zip(idx, idx[1:]) if any(n == "k_pcg_update_amg" for n in names[a:b])]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2
i0, i1 = steps[which]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end, busy, gaps = t0, 0, 0.0
hot = {"k_pcg_update_amg", "k_spmv_vec<64, 0, float>", "k_spmv_row<0, float>", "k_dense_mv_f32", "k_spmv<3, true, double, 8>",
       "k_spmv<0, true, float, 8>", "k_spmv<0, true, float, 4>", "k_spmv<6, true, float, 4>", "k_spmv<6, true, float, 8>", "k_spmv<4, true, double, 8>", "k_spmv<9, true, double, 8>"}
for i in range(i0, i1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gap = (s - prev_end) / 1e3
    if names[i] not in hot or gap > 2 or i - i0 < 8 or i1 - i < 20:
        print(f"{(s - t0) / 1e3:9.1f} gap {gap:6.1f} dur {(e - s) / 1e3:6.1f} {names[i]}")
    if gap > 0:
        gaps += gap
    prev_end = e
    busy += e - s
print(f"step {which}: {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, gaps {gaps:.1f} us, {i1 - i0} launches")
allsteps = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in steps[3:]]
print(f"median step {statistics.median(allsteps):.1f} us over {len(allsteps)} steps")
