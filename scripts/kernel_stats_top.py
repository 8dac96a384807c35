"""Top rows of a rocprofv3 kernel_stats.csv with short kernel names: python scripts/kernel_stats_top.py <csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    nm = re.sub(r"\(.*", "", nm)
    print(f"{nm:34s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms  {r['Percentage']}%")
