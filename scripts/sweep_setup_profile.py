"""Where the wall time of a whole C5 sweep call goes besides the point loop (cProfile around bench.run_sweep64 on one GPU):
    python scripts/sweep_setup_profile.py [batch] [concurrent]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ranks = bench.Ranks(bench.parse_args([]))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
sw = bench.run_sweep64(ranks, 64, 100, 5, conc, batch)
pr.disable()
print("whole %.3f s; value %.3e, whole-call value %.3e" % (time.perf_counter() - t0, sw["value"], sw["value_whole_call"]))
print(sw["rank0_phases_s"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
