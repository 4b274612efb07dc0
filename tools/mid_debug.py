#!/usr/bin/env python3
"""One Newton step (mismatch + solve) on a small grid; writes the increment: python tools/mid_debug.py <case> <batch> <out.npy>;
--compare a.npy b.npy lists the entries that differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    d[np.isnan(b)] = np.inf
    bad = np.argwhere(d > 1e-9 * np.abs(a).max())
    print("shape", a.shape, "max diff", d.max(), "bad entries", len(bad))
    rows = sorted(set(int(r[-1]) for r in bad))
    print("bad state indices:", rows[:80])
    if a.ndim == 2:
        print("bad scenarios:", sorted(set(int(r[0]) for r in bad))[:40])
    sys.exit(0)
import juliagrid.jl_amd as jg  # noqa: E402
case, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
s = jg.powerSystem(case)
an = jg.newtonRaphson(s, batch=batch)
jg.mismatch_(an)
jg.solve_(an)
inc = np.asarray(an.increment)
np.save(out, np.asarray(inc))
print(case, batch, "increment", inc.shape, "nan", int(np.isnan(inc).sum()))
an.close()
