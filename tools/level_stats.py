#!/usr/bin/env python3
"""Per-level shape of the factorisation plan (device-free): python tools/level_stats.py [case] [ld]
items / waves / workgroups / update terms per level launch, and the top launches."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "case_ACTIVSg10k"
ld = int(sys.argv[2]) if len(sys.argv) > 2 else 512
s = jg.powerSystem(case)
jg.acModel_(s)
Y = s.model.ac.nodalMatrix
policy = 1 | 4 | (((47 << 16 | 127 << 24 | 12 << 4) if Y.n >= 4000 else (47 << 16 | (280 // 8) << 24 | 4 << 4)) if ld >= 256 else ((26 if Y.n >= 4000 else 24) << 16 | (384 // 8) << 24))
plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=policy)
seg, rec = plan.replay_tables("fact")
print("level  segs  items  wgs(8 waves)  terms  max_terms_item  kinds(U/L, D, rhs, partial)")
tot = 0
for lv in np.unique(seg[:, 4]):
    sg = seg[seg[:, 4] == lv]
    items = sg[:, 6].sum()
    wgs = (sg[:, 1] * 2).sum()
    terms = 0; mx = 0; kinds = np.zeros(5, dtype=int)
    for b, c, w, r, *_ in sg:
        rr = rec[b:b + c * 16 * r]
        live = rr[rr[:, 0] >= 0]
        terms += live[:, 3].sum()
        lead = rr[::r * w] if w > 0 else rr
        per_item = rr[:, 3].reshape(-1, r * w).sum(axis=1) if w > 0 else rr[:, 3]
        mx = max(mx, per_item.max())
        for k in range(5):
            kinds[k] += (lead[:, 0] == k).sum()
    tot += terms
    print(f"{lv:5d} {len(sg):5d} {items:6d} {wgs:8d} {terms:8d} {mx:6d}   {kinds.tolist()}")
print("scheduled terms in levels:", tot)
tasks, data, launches = plan.top_tables()[:3]
print("top launches (task_begin, ntasks, cls, level, grouped, wg_begin, nwg, -):", launches.tolist() if hasattr(launches, "tolist") else launches)
for t in tasks:
    print("  task m=%d e=%d root=%d cls=%d level=%d G=%d" % (t[0], t[1], t[2], t[9], t[10], 1 << t[13]))
