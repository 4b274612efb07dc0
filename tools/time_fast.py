#!/usr/bin/env python3
"""Fast Newton-Raphson (XB) timing: ms per iteration (two forward/backward sweeps, no refactorisation). python tools/time_fast.py [batch] [case]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
case = sys.argv[2] if len(sys.argv) > 2 else "case1354pegase"
s = jg.powerSystem(case)
an = jg.fastNewtonRaphsonXB(s, batch=batch)
scale = 1.0 + 0.01 * np.random.default_rng(5).standard_normal((batch, 1))
jg.setInjection_(an, s.bus.supply.active[None, :] - s.bus.demand.active[None, :] * scale,
                 s.bus.supply.reactive[None, :] - s.bus.demand.reactive[None, :] * scale)
vm0, va0 = np.atleast_2d(an.voltage.magnitude).copy(), np.atleast_2d(an.voltage.angle).copy()
ts = []
for _ in range(4):
    jg.powerflow._push_voltage(an, vm0[0], va0[0])
    t0 = time.perf_counter()
    jg.powerFlow_(an, iteration=100, fetch=False)
    ts.append(time.perf_counter() - t0)
it = np.atleast_1d(an.method.iteration)
print(case, "batch", batch, "fast XB: ms/solve %.2f, iterations max %d mean %.1f, ms per iteration %.3f, converged %d/%d"
      % (1e3 * np.median(ts), it.max(), it.mean(), 1e3 * np.median(ts) / it.max(), int(np.sum(np.atleast_1d(an.status) == 0)), batch))
nr = jg.newtonRaphson(s, batch=batch)
jg.setInjection_(nr, s.bus.supply.active[None, :] - s.bus.demand.active[None, :] * scale,
                 s.bus.supply.reactive[None, :] - s.bus.demand.reactive[None, :] * scale)
ts = []
for _ in range(4):
    jg.setInitialPoint_(nr)
    t0 = time.perf_counter()
    jg.powerFlow_(nr, fetch=False)
    ts.append(time.perf_counter() - t0)
it = np.atleast_1d(nr.method.iteration)
print(case, "batch", batch, "Newton-Raphson: ms/solve %.2f, iterations max %d" % (1e3 * np.median(ts), it.max()))
# round 5: fast Newton-Raphson under BATCHED outages (per-scenario B', B'': one factorisation per batch, then sweeps only)
labels = [int(x) for x in jg.outageList(s, batch, seed=512)]
t0 = time.perf_counter()
fo = jg.contingencyAnalysis(s, labels, method="xb")
t_make = time.perf_counter() - t0
t0 = time.perf_counter()
jg.setOutages_(fo, labels)                                   # Ybus patches + 4 + 4 edits of B', B'' per scenario + ONE factorisation of the batch
t_patch = time.perf_counter() - t0
ts = []
for _ in range(4):
    jg.setInitialPoint_(fo)
    t0 = time.perf_counter()
    jg.powerFlow_(fo, iteration=100, fetch=False)
    ts.append(time.perf_counter() - t0)
it = np.atleast_1d(fo.method.iteration)
print(case, "batch", batch, "fast XB under %d outages: analysis %.1f ms, patch + refactor %.2f ms, ms/solve %.2f, iterations max %d mean %.1f, ms per iteration %.3f, converged %d/%d"
      % (batch, 1e3 * t_make, 1e3 * t_patch, 1e3 * np.median(ts), it.max(), it.mean(), 1e3 * np.median(ts) / max(it.max(), 1), int(np.sum(np.atleast_1d(fo.status) == 0)), batch))
