"""C5 sweep on one GPU at several (batch size, loops in flight): python scripts/sweep_concurrency.py 1x6 8x1 8x2 8x4"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    b, c = spec.split("x")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "sweep64", "--sweep-concurrent", c,
                        "--sweep-batch", b], capture_output=True, text=True)
    if p.returncode:
        print(spec, "failed", p.stderr[-800:]); continue
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print(f"batch {b} x {c:>2} in flight: wall {d['config']['wall_s']:.3f} s  value {d['value']:.3e} DOF-updates/s  "
          f"phases {d['config']['rank0_phases_s']}", flush=True)
