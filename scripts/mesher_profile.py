import cProfile, pstats, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import build_case
build_case("geballe_with_diamond", 2.0)
for scale in (1.0, 0.43):
    pr=cProfile.Profile(); pr.enable()
    t0=time.time(); cfg,stack,mesh=build_case("geballe_with_diamond", scale); print("scale", scale, time.time()-t0, mesh.stats)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(9)
