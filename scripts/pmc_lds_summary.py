"""Per-kernel LDS statistics from one rocprofv3 PMC pass (SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT).

    python scripts/pmc_lds_summary.py <pmc_dir> <out.csv>

bank = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (share of LDS-busy cycles lost to bank conflicts),
addr = SQ_LDS_ADDR_CONFLICT / SQ_LDS_IDX_ACTIVE (same for address conflicts of LDS atomics)."""
import collections, csv, glob, re, sys

d, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            cnt[k] += 1
with open(out, "w") as f:
    f.write("kernel,launches,lds_active_cycles_per_launch,bank_conflict_share,addr_conflict_share\n")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_LDS_IDX_ACTIVE"]):
        a = acc[k]["SQ_LDS_IDX_ACTIVE"]
        if a <= 0 or cnt[k] == 0:
            continue
        f.write(f"{k},{cnt[k]},{a / cnt[k]:.0f},{acc[k]['SQ_LDS_BANK_CONFLICT'] / a:.3f},{acc[k]['SQ_LDS_ADDR_CONFLICT'] / a:.3f}\n")
print(open(out).read())
