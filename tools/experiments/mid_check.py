#!/usr/bin/env python3
"""Grouped tasks below the multifrontal top (JG_MID_STRUCT, jg_symbolic.hpp) against the default plan: python tools/mid_check.py <case> <batch> <out.npz>
solves a seeded N-1 batch and writes iteration counts, status, V, theta; a second call with `--compare a.npz b.npz` prints the largest differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = a["st"] == 0
    print("iterations equal:", bool(np.array_equal(a["it"], b["it"])), "status equal:", bool(np.array_equal(a["st"], b["st"])),
          "max |dV| %.3e  max |dtheta| %.3e (converged scenarios: %d of %d)" % (np.abs(a["vm"] - b["vm"])[ok].max(), np.abs(a["va"] - b["va"])[ok].max(), ok.sum(), ok.size))
    sys.exit(0 if np.array_equal(a["it"], b["it"]) and np.abs(a["vm"] - b["vm"])[ok].max() < 1e-9 and np.abs(a["va"] - b["va"])[ok].max() < 1e-9 else 1)
import juliagrid.jl_amd as jg  # noqa: E402
case, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
s = jg.powerSystem(case)
an = jg.contingencyAnalysis(s, jg.outageList(s, batch, seed=512))
jg.powerFlow_(an, iteration=20, tolerance=1e-8)
np.savez(out, it=np.asarray(an.method.iteration), st=np.asarray(an.status), vm=np.asarray(an.voltage.magnitude), va=np.asarray(an.voltage.angle))
print(case, batch, os.environ.get("JG_MID_STRUCT"), "iterations", int(np.sum(an.method.iteration)), "failed", int(np.sum(np.asarray(an.status) != 0)))
an.close()
