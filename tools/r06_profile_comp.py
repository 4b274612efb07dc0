#!/usr/bin/env python3
"""Driver for the rocprofv3 passes over the compensated FIRST iteration (round 6): one 512-lane handle of the 10k-bus grid attached to a base case; R starts from the
base with an iteration limit of 1, i.e. R x (mismatch pass, correction, sweep pair on the shared factor, assembly of the next Jacobian), then the calibration copies.

  rocprofv3 --kernel-trace --stats -d ... -- python tools/r06_profile_comp.py [R] [case] [batch]
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ... -- python tools/r06_profile_comp.py      (and WRITE_SIZE in a pass of its own; tools/r06_pmc_comp.py sums them up)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
case = sys.argv[2] if len(sys.argv) > 2 else "case_ACTIVSg10k"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
s = jg.powerSystem(case)
single = jg.newtonRaphson(s)
jg.powerFlow_(single)
base = jg.BaseCase(single)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
base.attach(an)
for _ in range(R):
    jg.startFromBase_(an)
    jg.powerFlow_(an, iteration=1, fetch=False)
an.snapshot_voltage()          # calibration launches (device-to-device copies of known size), as tools/profile_kernels.py
an.restore_voltage()
print("first iterations", R, "counts", jg.firstIterationCounts(an), "base", base.info)
