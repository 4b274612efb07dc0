#!/usr/bin/env python3
"""JG_WALK_PROFILE=1: per-level work / barrier time of one workgroup of the persistent factor walk."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JG_WALK_PROFILE"] = "1"
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with np.load(os.path.join(ROOT, "tests", "golden", "cases", "case_ACTIVSg10k.npz")) as z:
    t = {k: z[k] for k in z.files}
s = jg.powerSystem(t)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
for k in (0, 1, 2):
    print("kernel", k, "ms", an.time_kernel(k, 5), file=sys.stderr)
an.close()
