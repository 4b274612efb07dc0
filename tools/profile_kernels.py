#!/usr/bin/env python3
"""Small driver for rocprofv3 runs (kernel trace / PMC passes): one batched N-1 power flow on the 10k-bus grid.

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 --output-format csv -- python tools/profile_kernels.py
  rocprofv3 --pmc FETCH_SIZE  --kernel-trace -d ... -- python tools/profile_kernels.py      (separate passes)
  rocprofv3 --pmc WRITE_SIZE  --kernel-trace -d ... -- python tools/profile_kernels.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
case = sys.argv[3] if len(sys.argv) > 3 else "case_ACTIVSg10k"
s = jg.powerSystem(case)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))     # ONE handle: every dispatch belongs to the batch
jg.powerFlow_(an, iteration=iters, fetch=False)
# calibration launches for the PMC passes (tools/pmc_summary.py): device-to-device copies of KNOWN size -- V and theta, n x ld x 8
# bytes each, read once and written once
an.snapshot_voltage()
an.restore_voltage()
print("dims", an.dims, "solves", iters, "iterations", int(np.sum(an.method.iteration)))
