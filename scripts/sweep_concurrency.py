"""C5 sweep on one GPU at several numbers of points in flight: python scripts/sweep_concurrency.py 1 2 4 8 16"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for c in sys.argv[1:]:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sweep64", "--sweep-concurrent", c],
                       capture_output=True, text=True)
    if p.returncode:
        print(c, "failed", p.stderr[-500:]); continue
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print(f"concurrent {c:>2}: wall {d['config']['wall_s']:.3f} s  value {d['value']:.3e} DOF-updates/s  "
          f"phases {d['config']['rank0_phases_s']}", flush=True)
