#!/usr/bin/env python3
"""BASELINE configs 1-2 (single instance): ms per powerFlow! on the GPU next to the C oracle on one host core, same start point.
python tools/single_vs_oracle.py [case ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
from oracle import oracle as O  # noqa: E402  (measurement tool: the oracle is the CPU baseline here, never the product path)
for case in (sys.argv[1:] or ["case1354pegase", "case9241synth", "case_ACTIVSg10k"]):
    s = jg.powerSystem(case)
    an = jg.newtonRaphson(s)
    ts = []
    for _ in range(9):
        jg.setInitialPoint_(an)
        t0 = time.perf_counter()
        jg.powerFlow_(an, fetch=True)
        ts.append(time.perf_counter() - t0)
    if case == "case9241synth":
        t = jg.synthetic.case9241synth()
    else:
        with np.load(os.path.join(ROOT, "tests", "golden", "cases", case + ".npz")) as z:
            t = {k: z[k] for k in z.files}
    osys = O.OracleSystem(t)
    to = []
    for _ in range(5):
        o = O.OracleNR(osys)
        t0 = time.perf_counter()
        st = o.power_flow()
        to.append(time.perf_counter() - t0)
    vm, va = o.voltage()
    print("%s: GPU %.3f ms/solve (%d iterations, %.3f ms/iteration) | C oracle, 1 core: %.3f ms/solve (%d iterations) | max |dV| %.1e |dtheta| %.1e" % (
        case, 1e3 * np.median(ts), an.method.iteration, 1e3 * np.median(ts) / an.method.iteration, 1e3 * np.median(to), o.iteration,
        np.abs(an.voltage.magnitude - vm).max(), np.abs(an.voltage.angle - va).max()))
    an.close()
