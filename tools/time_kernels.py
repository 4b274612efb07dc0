#!/usr/bin/env python3
"""HIP-event kernel timings (assembly / factorisation / backward) of the batched NR handle: python tools/time_kernels.py [batch] [case] [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
case = sys.argv[2] if len(sys.argv) > 2 else "case_ACTIVSg10k"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
s = jg.powerSystem(case)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
jg.powerflow._upload_branches(an)
for _ in range(3):
    print(case, batch, "asm %.4f  fact %.4f  bwd %.4f  branch-post %.4f ms" % tuple(an.time_kernel(k, reps) for k in (0, 1, 2, 3)))
an.close()
