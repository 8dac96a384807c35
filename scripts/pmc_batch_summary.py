"""Per-kernel HBM bytes per launch of the batched loop's kernels (kb_*) from two rocprofv3 PMC passes.

    python scripts/pmc_batch_summary.py <fetch_dir> <write_dir> <out_prefix> <nv> [n nnz]

bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 as in scripts/pmc_summary.py (gfx950 counts 64 B per 128-B request in
FETCH_SIZE; calibrated on the vector kernels).  n and nnz default to the `n = ... nnz = ...` line scripts/batch_probe.py
prints (its log sits next to the PMC directories).  Algorithmic bytes (SURVEY 8d conventions: f64 values, i32 indices,
every array once per pass) for the affine operator family A_j = A + d_j A1 on NV interleaved columns:
  kb_spmv<9>  (4 + 16)*nnz + 4*n + 40*n*NV   colidx + two value arrays; rowptr; per row and column z (8), Ap and p read-modify-write (32)
              (kb_spmv_lds, the LDS-staged kernel with 16-bit column positions, is held to the same i32-index formula)
  kb_spmv<4>  (4 + 16)*nnz + 4*n + 32*n*NV   z, b, dinv read, z2 written
  kb_spmv<3>  (4 + 16)*nnz + 4*n + 24*n*NV   z, b read, tmp written
  kb_spmv<0>  (4 + 8)*nnz  + 4*n + 16*n*NV   right-hand side b = M u (shared values)
  kb_update   64*n*NV                        x, r read-modify-write; p, Ap, dinv read; z written
"""
import collections, csv, glob, json, os, re, sys

fetch_dir, write_dir, out, nv = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
if len(sys.argv) > 6:
    n, nnz = int(sys.argv[5]), int(sys.argv[6])
else:
    log = open(os.path.join(os.path.dirname(out), os.path.basename(out).replace("pmc_traffic", "pmc_FETCH_SIZE") + ".log")).read()
    m = re.search(r"n = (\d+)\s+nnz = (\d+)", log)
    n, nnz = int(m.group(1)), int(m.group(2))


def short(name):
    m = re.search(r"(kb?_[a-z_0-9]+)(<[^>]*>)?", name)
    if not m:
        return name[:30]
    targs = (m.group(2) or "").replace(" ", "")
    if m.group(1) in ("kb_spmv", "kb_spmv_lds"):
        targs = re.sub(r"^<(\d+),(\d+),(\d+)(,.*)?>$", lambda q: f"<{q.group(1)},nv{q.group(2)},op{q.group(3)}{q.group(4) or ''}>", targs)
    return m.group(1) + targs


res = {}
for kind, d in (("fetch", fetch_dir), ("write", write_dir)):
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    res[kind] = acc


def algorithmic(k):
    m = re.match(r"kb_spmv(?:_lds)?<(\d+),nv(\d+),op(\d+)", k)
    if m:
        mode, v, op = int(m.group(1)), int(m.group(2)), int(m.group(3))
        mat = {0: 12, 1: 4 + 8 * v, 2: 20}[op] * nnz + 4 * n
        per = {9: 40, 4: 32, 3: 24, 0: 16, 5: 40, 2: 40, 8: 32}.get(mode)
        return mat + per * n * v if per else None
    m = re.match(r"kb_update<(\d+)", k)
    if m:
        return 64 * n * int(m.group(1))
    return None


kern = {}
with open(out + ".csv", "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KB_p90,WRITE_SIZE_KB_p90,hbm_bytes_per_launch_corrected,algorithmic_bytes,ratio\n")
    for k in sorted(res["fetch"], key=lambda k: -sum(res["fetch"][k])):
        if not k.startswith("kb_"):
            continue
        fv, wv = sorted(res["fetch"][k]), sorted(res["write"].get(k, [0.0]))
        fk, wk = fv[max(0, int(0.9 * len(fv)) - 1)], wv[max(0, int(0.9 * len(wv)) - 1)]   # p90: skips launches that return at once
        corr = (2 * fk + wk) * 1024
        a = algorithmic(k)
        f.write(f"{k},{len(fv)},{fk:.0f},{wk:.0f},{corr:.0f},{a if a else ''},{(corr / a if a else float('nan')):.3f}\n")
        kern[k] = {"hbm_bytes": corr, "algorithmic": a}
json.dump({"n": n, "nnz": nnz, "nv": nv, "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024, see scripts/pmc_summary.py", "kernels": kern}, open(out + ".json", "w"), indent=1)
print(open(out + ".csv").read())
