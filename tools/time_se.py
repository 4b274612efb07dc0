#!/usr/bin/env python3
"""HIP-event kernel timings of the config-4 state estimation handle: python tools/time_se.py [batch] [case]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
case = sys.argv[2] if len(sys.argv) > 2 else "case9241synth"
s = jg.powerSystem(case)
pf = jg.newtonRaphson(s)
jg.powerFlow_(pf, tolerance=1e-11)
mon = jg.measurement(s)
jg.addVoltmeter_(mon, pf); jg.addWattmeter_(mon, pf); jg.addVarmeter_(mon, pf)
jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
an = jg.gaussNewton(mon, batch=batch)
print(an.dims)
for _ in range(2):
    print(case, batch, "rows %.4f  gain %.4f  fact %.4f  bwd %.4f  selected-inverse %.4f ms" % tuple(an.time_kernel(k, 5) for k in (0, 1, 2, 3, 4)))
import time
jg.stateEstimation_(an)
t0 = time.perf_counter(); out = jg.residualTest_(an); t1 = time.perf_counter()
print("residualTest_ wall %.2f ms (first call builds the tables), max nres %.3g" % (1e3 * (t1 - t0), float(np.max(out.maxNormalizedResidual))))
t0 = time.perf_counter(); out = jg.residualTest_(an); t1 = time.perf_counter()
print("residualTest_ wall %.2f ms" % (1e3 * (t1 - t0)))
