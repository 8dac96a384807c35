"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    python scripts/pmc_summary.py <fetch_dir> <write_dir> <out_prefix> <n> <n_e> <nnz>

bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE counts 64 B per 128-B request
(MI355X_MICROARCH.md, HBM); the factor is calibrated here on the vector kernels, whose byte
count is known exactly (k_pcg_update reads 40 B/row)."""
import collections, csv, glob, json, re, statistics, sys

fetch_dir, write_dir, out, n, ne, nnz = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name[:30]
    targs = m.group(2) or ""
    if m.group(1) == "k_spmv":
        targs = re.sub(r",\s*(true|false)", "", targs)             # k_spmv<9, true, double> (compressed columns) -> k_spmv<9>
        targs = re.sub(r",\s*double", "", targs).replace(", float", ",f32")   # single-precision transfer operators keep a tag
        if "f32" not in targs:
            targs = re.sub(r",\s*\d+>$", ">", targs)               # stream entries per lane (4 / 8): kept for the pipelined transfer operators only
    return m.group(1) + targs


res = {}
for kind, d in (("fetch", fetch_dir), ("write", write_dir)):
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    res[kind] = acc
alg = {"k_spmv<9>": 12 * nnz + 44 * n, "k_spmv<0>": 12 * nnz + 20 * n, "k_spmv<3>": 12 * nnz + 28 * n,
       "k_spmv<4>": 12 * nnz + 36 * n, "k_pcg_update_amg": 64 * n, "k_pcg_update": 64 * n, "k_spmv<8>": 12 * nnz + 36 * n, "k_assemble_lds<false>": 16 * ne + 16 * n + 16 * nnz,
       "k_assemble_lds<true>": 16 * ne + 16 * n + 16 * nnz, "k_assemble_rows": 16 * ne + 16 * n + 16 * nnz, "k_spmv<2>": 12 * nnz + 36 * n}
kern = {}
with open(out + ".csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_p90,WRITE_SIZE_KB_p90,hbm_bytes_per_launch_corrected,algorithmic_bytes,ratio\n")
    for k in sorted(res["fetch"], key=lambda k: -sum(res["fetch"][k])):
        fv, wv = sorted(res["fetch"][k]), sorted(res["write"].get(k, [0.0]))
        fk, wk = fv[max(0, int(0.9 * len(fv)) - 1)], wv[max(0, int(0.9 * len(wv)) - 1)]   # p90: skips no-op launches
        corr = (2 * fk + wk) * 1024
        a = alg.get(k)
        f.write(f"{k},{len(fv)},{fk:.0f},{wk:.0f},{corr:.0f},{a if a else ''},{(corr / a if a else float('nan')):.3f}\n")
        if a:
            kern[k] = {"hbm_bytes": corr, "algorithmic": a}
# one multigrid-PCG iteration = the launches of scripts/iter_breakdown.py's sequence; kernels that appear twice per iteration (the
# sub-wave kernel: down leg of level 1, up leg of level 2) enter with the mean over their launches, times two.  Launches that
# returned at their first instruction (after convergence) carry next to nothing and are left out of the means.
def mean_bytes(k):
    fv, wv = res["fetch"].get(k, []), res["write"].get(k, [])
    if not fv:
        return None
    f_ok = [v for v in fv if v > 0.1 * max(fv)]
    w_ok = [v for v in wv if v > 0.1 * max(wv)] if wv and max(wv) > 0 else [0.0]
    return (2 * statistics.mean(f_ok) + statistics.mean(w_ok)) * 1024


iteration = {}
for k in sorted(res["fetch"]):
    per_iter = 2 if k.startswith("k_spmv_vec") else 1
    if k in ("k_spmv<9>", "k_spmv<4>", "k_pcg_update_amg", "k_dense_mv_f32") or k.startswith(("k_spmv<0,f32", "k_spmv<6,f32", "k_spmv<7,f32", "k_spmv_vec<", "k_spmv_row<")):
        b = mean_bytes(k)
        if b:
            iteration[k] = {"bytes_per_launch_mean": b, "launches_per_iteration": per_iter}
it_total = sum(v["bytes_per_launch_mean"] * v["launches_per_iteration"] for v in iteration.values())
json.dump({"n": n, "nnz": nnz, "note": __doc__.split("bytes =")[1].strip(), "kernels": kern,
           "multigrid_iteration": {"bytes": it_total, "kernels": iteration,
                                   "note": "sum over the launches of one multigrid-PCG iteration (iteration head, update, V-cycle), mean HBM bytes per launch"}},
          open(out + ".json", "w"), indent=1)
print(open(out + ".csv").read())
