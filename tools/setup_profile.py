#!/usr/bin/env python3
"""Where the set-up of a single power flow goes (newtonRaphson(): host model, symbolic analysis, upload; first powerFlow!: graph capture):
python tools/setup_profile.py [case]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
s = jg.powerSystem(case)
an = jg.newtonRaphson(s); jg.powerFlow_(an); an.close()          # warm the process (library load, HIP context)
s = jg.powerSystem(case)
t0 = time.perf_counter(); pr = cProfile.Profile(); pr.enable()
an = jg.newtonRaphson(s)
pr.disable(); t1 = time.perf_counter()
jg.powerFlow_(an); t2 = time.perf_counter()
jg.setInitialPoint_(an); t3 = time.perf_counter(); jg.powerFlow_(an); t4 = time.perf_counter()
print(case, "CACHED plan: newtonRaphson %.1f ms, first powerFlow! %.1f ms, warm powerFlow! %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t4 - t3)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
an.close()
jg._lib.lib().jg_plan_cache_clear()
s = jg.powerSystem(case)
t0 = time.perf_counter(); pr = cProfile.Profile(); pr.enable()
an = jg.newtonRaphson(s)
pr.disable(); t1 = time.perf_counter()
jg.powerFlow_(an); t2 = time.perf_counter()
print(case, "COLD plan (warm process): newtonRaphson %.1f ms, first powerFlow! %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
t0 = time.perf_counter()
pipe = jg.ContingencyPipeline(s, 512, inflight=3, pool=256)
t1 = time.perf_counter()
print("ContingencyPipeline(512, inflight=3, pool=256) construction %.1f ms" % (1e3 * (t1 - t0)))
pipe.close()
