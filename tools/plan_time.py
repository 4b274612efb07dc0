#!/usr/bin/env python3
"""Wall time of the symbolic analysis alone (device-free, C ABI jg_plan_create): python tools/plan_time.py [case] [scenarios]; JG_PLAN_TIMING=1 prints the phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import juliagrid.jl_amd as jg
case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
ld = int(sys.argv[2]) if len(sys.argv) > 2 else 64
s = jg.powerSystem(case); jg.acModel_(s); Y = s.model.ac.nodalMatrix
policy = 1 | 4 | (((47 << 16 | 127 << 24 | 12 << 4) if Y.n >= 4000 else (47 << 16 | (280 // 8) << 24 | 4 << 4)) if ld >= 256 else ((26 if Y.n >= 4000 else 24) << 16 | (384 // 8) << 24))
if ld >= 256: policy |= 1 << 50
policy |= 1 << 49
for r in range(3):
    t0 = time.perf_counter()
    plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=policy)
    print("analysis %.2f ms" % (1e3 * (time.perf_counter() - t0)))
