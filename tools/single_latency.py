#!/usr/bin/env python3
"""Single-instance latency (ms per powerFlow!) from the case's start point: python tools/single_latency.py [case] [batch]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
s = jg.powerSystem(case)
an = jg.newtonRaphson(s, batch=batch)
ts = []
for _ in range(7):
    jg.setInitialPoint_(an)
    t0 = time.perf_counter()
    jg.powerFlow_(an, fetch=False)
    ts.append(time.perf_counter() - t0)
it = np.atleast_1d(an.method.iteration)
print(case, "batch", batch, "ms/solve %.3f" % (1e3 * np.median(ts)), "iterations", int(it[0]),
      "kernels asm %.4f fact %.4f bwd %.4f" % tuple(an.time_kernel(k, 10) for k in (0, 1, 2)))
