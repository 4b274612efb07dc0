#!/usr/bin/env python3
"""Stress of the single-instance path (round 6): thousands of warm solves of one handle from alternating starts (the case's start point, the solution, a point half-way,
an outage and its restoration), every one checked for status, iteration count and state.  python tools/experiments/r06_single_stress.py [case] [solves]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
s = jg.powerSystem(case)
an = jg.newtonRaphson(s)
vm0, va0 = an.voltage.magnitude.copy(), an.voltage.angle.copy()
jg.powerFlow_(an)
k, vm, va = int(an.method.iteration), an.voltage.magnitude.copy(), an.voltage.angle.copy()
jg.powerflow._push_voltage(an, 0.5 * (vm0 + vm), 0.5 * (va0 + va)); jg.powerFlow_(an); k2 = int(an.method.iteration)
rng = np.random.default_rng(7)
t0 = time.perf_counter()
bad = 0
for i in range(solves):
    mode = int(rng.integers(0, 3))
    if mode == 0:
        jg.setInitialPoint_(an); want = k
    elif mode == 1:
        want = 0                                       # from the solution it stands on
    else:
        jg.powerflow._push_voltage(an, 0.5 * (vm0 + vm), 0.5 * (va0 + va)); want = k2
    jg.powerFlow_(an)
    ok = an.status == 0 and int(an.method.iteration) == want and np.abs(an.voltage.magnitude - vm).max() < 1e-8 and np.abs(an.voltage.angle - va).max() < 1e-8
    if not ok:
        bad += 1
        print("solve", i, "mode", mode, "status", an.status, "iterations", an.method.iteration, "expected", want, flush=True)
print(f"{case}: {solves} solves in {time.perf_counter() - t0:.1f} s, {bad} wrong (iterations from the start {k}, half-way {k2})")
an.close()
