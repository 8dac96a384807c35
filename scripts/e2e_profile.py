"""Wall-clock breakdown (cProfile) of one reference-style run: with_diamond.main() on the stock config."""
import cProfile, pstats, sys, os, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(tempfile.mkdtemp())
import with_diamond
scale = sys.argv[1] if len(sys.argv) > 1 else "1.0"
with_diamond.main(["--scale", "4.0"])          # warm-up: library load, HIP context
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
with_diamond.main(["--scale", scale])
pr.disable()
print("wall %.3f s" % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
