#!/usr/bin/env python3
"""Per-iteration timeline of one batched N-1 run (JG_TRACE=1 prints the active-scenario counts)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JG_TRACE"] = "1"
import juliagrid.jl_amd as jg  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
case = sys.argv[2] if len(sys.argv) > 2 else "case_ACTIVSg10k"
with np.load(os.path.join(ROOT, "tests", "golden", "cases", case + ".npz")) as z:
    t = {k: z[k] for k in z.files}
s = jg.powerSystem(t)
base = jg.newtonRaphson(s, batch=1)
jg.powerFlow_(base)
vm0, va0 = base.voltage.magnitude.copy(), base.voltage.angle.copy()
base.close()
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
jg.powerflow._push_voltage(an, vm0, va0)
an.snapshot_voltage()
for rep in range(3):
    an.restore_voltage()
    t0 = time.perf_counter()
    jg.powerFlow_(an, iteration=20, tolerance=1e-8, fetch=False)
    print("run", rep, "ms", 1e3 * (time.perf_counter() - t0), "iterations", int(np.sum(an.method.iteration)),
          "hist", np.bincount(an.method.iteration).tolist(), file=sys.stderr)
