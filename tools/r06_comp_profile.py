#!/usr/bin/env python3
"""The shared-factor first iteration alone (round 6): one 512-lane handle of the 10k-bus grid attached to a base case; times the linear step
(jg_nr_time_kernel 4: correction + sweep pair) and the mismatch pass for a list of dense-top sizes.

  python tools/r06_comp_profile.py [top_cap ...]                                  (a table)
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r06_comp --output-format csv -- python tools/r06_comp_profile.py 0     (per-kernel durations)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

caps = [int(x) for x in sys.argv[1:]] or [0]
case = os.environ.get("JG_CASE", "case_ACTIVSg10k")
batch = int(os.environ.get("JG_BATCH", "512"))
s = jg.powerSystem(case)
single = jg.newtonRaphson(s)
jg.powerFlow_(single)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
for cap in caps:
    base = jg.BaseCase(single, top_cap=cap)
    base.attach(an)
    jg.startFromBase_(an)
    jg.powerFlow_(an, fetch=False)
    t4 = float(np.median([an.time_kernel(4, 10) for _ in range(5)]))
    t5 = float(np.median([an.time_kernel(5, 20) for _ in range(5)]))
    info = base.info
    print(f"top_cap {cap:5d}: top {info['top_pivots']:4d} pivots, {info['forward_launches']:2d} + {info['backward_launches']:2d} level launches, base {info['create_ms']:.1f} ms | "
          f"shared-factor step {t4:.4f} ms, mismatch pass {t5:.4f} ms | iterations {np.bincount(an.method.iteration).tolist()}")
    base.close()
