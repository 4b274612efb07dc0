#!/usr/bin/env python3
"""A single instance solved repeatedly on one warm handle (round 6): ms per solve; under rocprofv3 --kernel-trace the timeline of its iteration graphs.
   python tools/r06_single_probe.py [case] [solves]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
solves = int(sys.argv[2]) if len(sys.argv) > 2 else 12
s = jg.powerSystem(case)
an = jg.newtonRaphson(s)
jg.powerFlow_(an)
ts = []
for k in range(solves):
    jg.setInitialPoint_(an)
    t0 = time.perf_counter()
    jg.powerFlow_(an, fetch=False)
    ts.append(1e3 * (time.perf_counter() - t0))
print(case, "iterations", an.method.iteration, "ms per solve", [round(x, 3) for x in ts])
an.close()
