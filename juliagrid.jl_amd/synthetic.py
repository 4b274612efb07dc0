"""Seeded synthetic transmission grids with the shape of the PEGASE cases (measurement input, host only).

BASELINE.json names `case9241pegase` for the rocprof / state-estimation / contingency configurations, but the
reference ships no such case (docs/src/examples/powerSystemDatasets.md:4-15; `.MISSING_LARGE_BLOBS` lists only the
25k/70k/USA grids).  SURVEY.md 8(d) therefore specifies a generator: n = 9 241 buses, 16 049 branches, 1 445
generators, seed 9241, every random draw from PCG64(seed):

  * topology  - buses are points in the unit square; a Euclidean minimum spanning tree over the 6-nearest-neighbour
                graph makes it connected and near-planar (small separators, like a real grid), the shortest remaining
                neighbour pairs become loop-closing chords (a third of the buses stay radial leaves as in
                case1354pegase, degree <= 12), ~n/12 substations are tied to their 4 nearest substations by long
                lines (the meshed backbone that produces realistic fill: ~30-wide fronts), and the remainder up to
                `nb` are parallel circuits (case1354pegase: 14 % of its branches duplicate a bus pair);
  * branches  - (r, x, g, b, tap, shift) rows bootstrapped jointly from case1354pegase's branch table, lines and
                transformers (11.8 % off-nominal taps, 0.3 % phase shifters) separately: transformers feed radial
                leaf buses only (a tap or shift inside a loop of 0.001-pu lines would circulate several pu), the
                tap side at the meshed end; parallel circuits share the parameters of their twin;
  * buses     - (Pd, Qd, Gs, Bs) rows bootstrapped jointly from case1354pegase's bus table;
  * units     - 15.6 % of the buses are PV (non-leaf buses), one of them the slack; voltage set-points vary smoothly
                over the map (1.00 .. 1.04 pu); dispatch proportional to bootstrapped |Pg|, balanced to demand + 2 %;
  * loading   - demand, dispatch and bus shunts are multiplied by `load_scale`, the one tuned constant.  With the CPU
                oracle's Newton-Raphson from the flat start (V = set-points / 1.0, theta = 0): 0.30 is the largest
                0.05-step that converges (6 iterations, 74 degrees of angle spread), 0.35 does not; the shipped value
                0.20 keeps every bus above 0.9 pu (5 iterations, 48 degrees) and leaves room for N-1 outages.
                tests/test_synthetic.py asserts the <= 10 iterations (the product code never imports the oracle).

The result is a plain table dict for `powerSystem(...)`, identical in layout to the npz fixtures.
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(os.path.dirname(_HERE), "tests", "golden", "cases", "case1354pegase.npz")

PEGASE9241 = dict(n=9241, nb=16049, ng=1445, seed=9241, load_scale=0.2)


def _topology(n, nb, rng, k=6, dup_frac=0.141, leaf_frac=0.30, hub_div=12, hub_k=4, max_deg=12):
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components, minimum_spanning_tree
    from scipy.spatial import cKDTree

    pts = rng.random((n, 2))
    tree = cKDTree(pts)
    d, idx = tree.query(pts, k=k + 1)
    rows, cols = np.repeat(np.arange(n), k), idx[:, 1:].reshape(-1)
    w = d[:, 1:].reshape(-1) + 1e-9 * rng.random(n * k)            # ties broken by the stream, not by the sort
    a, b = np.minimum(rows, cols), np.maximum(rows, cols)
    _, u = np.unique(a.astype(np.int64) * n + b, return_index=True)
    a, b, w = a[u], b[u], w[u]
    mst = minimum_spanning_tree(coo_matrix((w, (a, b)), shape=(n, n)).tocsr()).tocoo()
    edges = {(int(min(x, y)), int(max(x, y))) for x, y in zip(mst.row, mst.col)}
    while True:                                                      # the kNN graph may fall apart: bridge the pieces
        e = np.array(sorted(edges))
        nc, lab = connected_components(coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(n, n)), directed=False)
        if nc == 1:
            break
        c0, rest = np.flatnonzero(lab == 0), np.flatnonzero(lab != 0)
        dd, ii = cKDTree(pts[c0]).query(pts[rest])
        j = int(np.argmin(dd))
        x, y = int(c0[ii[j]]), int(rest[j])
        edges.add((min(x, y), max(x, y)))
    deg = np.zeros(n, dtype=int)
    for x, y in edges:
        deg[x] += 1
        deg[y] += 1
    leaves = np.flatnonzero(deg == 1)
    radial = set(rng.choice(leaves, size=min(len(leaves), int(leaf_frac * n)), replace=False).tolist())
    # meshed backbone: substations tied to their nearest substations by long lines
    hubs = rng.choice(np.setdiff1d(np.arange(n), np.fromiter(radial, dtype=int)), size=n // hub_div, replace=False)
    _, hi = cKDTree(pts[hubs]).query(pts[hubs], k=hub_k + 1)
    for i in range(len(hubs)):
        for j in hi[i, 1:]:
            x, y = int(hubs[i]), int(hubs[j])
            e = (min(x, y), max(x, y))
            if e not in edges:
                edges.add(e)
                deg[x] += 1
                deg[y] += 1
    distinct = int(round(nb * (1.0 - dup_frac)))
    for o in np.argsort(w, kind="stable"):                           # loop-closing chords, shortest first
        if len(edges) >= distinct:
            break
        x, y = int(a[o]), int(b[o])
        if (x, y) in edges or x in radial or y in radial or deg[x] >= max_deg or deg[y] >= max_deg:
            continue
        edges.add((x, y))
        deg[x] += 1
        deg[y] += 1
    e = np.array(sorted(edges), dtype=np.int64)
    if len(e) > nb:
        raise ValueError("more distinct bus pairs than branches: lower hub_div / raise nb")
    twin = rng.integers(0, len(e), nb - len(e))                      # parallel circuits: twins of existing branches
    origin = np.concatenate([np.arange(len(e)), twin])               # branch -> the distinct pair it realises
    e = e[origin]
    flip = rng.random(nb) < 0.5                                      # orientation carries no meaning for lines
    e[flip] = e[flip][:, ::-1]
    order = rng.permutation(nb)
    return e[order], origin[order], deg, pts


def pegaseShaped(n=9241, nb=16049, ng=1445, seed=9241, load_scale=0.2):
    """Table dict of a seeded synthetic PEGASE-shaped grid (see the module docstring)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    with np.load(_SRC) as z:
        src = {k: z[k] for k in z.files}
    e, origin, deg, pts = _topology(n, nb, rng)
    # one parameter row per distinct bus pair; transformers only on edges that end in a radial leaf
    is_tr = (src["br_tap"] != 1.0) | (src["br_shift"] != 0.0)
    lines, trafos = np.flatnonzero(~is_tr), np.flatnonzero(is_tr)
    npair = int(origin.max()) + 1
    pair_row = lines[rng.integers(0, lines.size, npair)]
    leaf_edge = np.zeros(npair, dtype=bool)
    leaf_edge[origin] = (deg[e[:, 0]] == 1) | (deg[e[:, 1]] == 1)
    want = int(round(trafos.size / src["br_r"].size * nb))
    cand = np.flatnonzero(leaf_edge)
    chosen = rng.choice(cand, size=min(want, cand.size), replace=False)
    pair_row[chosen] = trafos[rng.integers(0, trafos.size, chosen.size)]
    bi = pair_row[origin]
    swap = is_tr[bi] & (deg[e[:, 0]] == 1)                           # tap side (`from`) at the meshed end of the leaf edge
    e[swap] = e[swap][:, ::-1]
    ui = rng.integers(0, src["bus_pd"].size, n)
    npv = int(round(0.156 * n))
    cand = np.flatnonzero(deg >= 2)
    gen_buses = rng.choice(cand, size=npv + 1, replace=False)        # PV buses + the slack, on non-leaf buses
    slack = int(gen_buses[np.argmax(deg[gen_buses])])                # best-connected of them (ties: first drawn)
    bus_type = np.ones(n, dtype=np.int8)
    bus_type[gen_buses] = 2
    bus_type[slack] = 3
    extra = rng.choice(gen_buses, size=max(ng - gen_buses.size, 0), replace=True)
    gen_bus = np.concatenate([gen_buses, extra])[:ng]
    gi = rng.integers(0, src["gen_pg"].size, gen_bus.size)
    pd, qd = src["bus_pd"][ui] * load_scale, src["bus_qd"][ui] * load_scale
    pd[gen_buses] *= 0.25                                            # plants carry little local demand
    qd[gen_buses] *= 0.25
    wgt = np.abs(src["gen_pg"][gi]) + 0.05
    pg = wgt / wgt.sum() * 1.02 * pd.sum()
    # set-points vary smoothly over the map (1.00 .. 1.04): neighbouring plants 0.001 pu apart must not be asked to
    # hold voltages 0.1 pu apart (bootstrapped set-points do exactly that and no power flow exists)
    vg = 1.02 + 0.02 * np.cos(2.0 * np.pi * pts[gen_bus, 0]) * np.cos(2.0 * np.pi * pts[gen_bus, 1])
    cap = np.maximum(np.abs(src["gen_qmax"][gi]), 0.5) * max(load_scale, 0.25) * 4.0
    return dict(
        base_power=np.array([1.0e8]), bus_label=np.arange(1, n + 1, dtype=np.int64), bus_type=bus_type,
        bus_pd=pd, bus_qd=qd, bus_gs=src["bus_gs"][ui], bus_bs=src["bus_bs"][ui] * load_scale,
        bus_vm=np.ones(n), bus_va=np.zeros(n),
        br_from=e[:, 0] + 1, br_to=e[:, 1] + 1, br_status=np.ones(nb, dtype=np.int8),
        br_r=src["br_r"][bi], br_x=src["br_x"][bi], br_g=src["br_g"][bi], br_b=src["br_b"][bi],
        br_tap=src["br_tap"][bi], br_shift=src["br_shift"][bi],
        gen_bus=(gen_bus + 1).astype(np.int64), gen_status=np.ones(gen_bus.size, dtype=np.int8),
        gen_pg=pg, gen_qg=np.zeros(gen_bus.size), gen_vg=vg, gen_qmin=-cap, gen_qmax=cap,
    )


def case9241synth():
    """The grid standing in for case9241pegase in BASELINE.json's configurations."""
    return pegaseShaped(**PEGASE9241)


def tiledGrid(tables, copies, slack_active=None, seed=0, jitter=0.005, tie_x=0.05, star=False):
    """A grid of `copies` x n buses built from a solved case: `copies` instances of `tables`, the slack bus of instance c tied to the
    slack bus of instance c + 1 by one line (x = tie_x pu), ONE slack (instance 0; the other former slack buses become PV buses whose
    unit produces `slack_active` = the slack's active OUTPUT in the solved single case, pu -- the caller computes it, e.g. from
    power!(analysis) -- so that the ties carry next to nothing and the stored voltages stay a good start), branch reactances jittered by
    +-`jitter` (seeded) so that the instances are not bit-identical.  Stands in for the reference's 25 000 / 70 000 / 82 000-bus datasets
    (docs/src/examples/powerSystemDatasets.md:13-15: not shipped, `.MISSING_LARGE_BLOBS`): same table layout, sizes beyond 10 000 buses
    for the symbolic analysis, the int32 tables and the memory plan."""
    t = {k: np.array(v) for k, v in tables.items()}
    n, nb, ng = t["bus_type"].size, t["br_from"].size, t["gen_bus"].size
    slack = int(np.flatnonzero(t["bus_type"] == 3)[0]) + 1
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {"base_power": t["base_power"].copy()}
    for k in t:
        if k.startswith("bus_"):
            out[k] = np.tile(t[k], copies)
    for c in range(1, copies):
        out["bus_type"][c * n + slack - 1] = 2
    br = {k: np.tile(t[k], copies) for k in t if k.startswith("br_")}
    off_b = np.repeat(np.arange(copies) * n, nb)
    br["br_from"] = br["br_from"] + off_b
    br["br_to"] = br["br_to"] + off_b
    br["br_x"] = br["br_x"] * (1.0 + jitter * (2.0 * rng.random(copies * nb) - 1.0))
    ties = copies - 1
    tie = {"br_from": np.array([(0 if star else c) * n + slack for c in range(ties)], dtype=np.int64),     # star: every instance tied to instance 0, else a chain
           "br_to": np.array([(c + 1) * n + slack for c in range(ties)], dtype=np.int64),
           "br_status": np.ones(ties, dtype=np.int8), "br_r": np.full(ties, 0.1 * tie_x), "br_x": np.full(ties, tie_x),
           "br_g": np.zeros(ties), "br_b": np.zeros(ties), "br_tap": np.ones(ties), "br_shift": np.zeros(ties)}
    for k in br:
        out[k] = np.concatenate([br[k], tie[k].astype(br[k].dtype)])
    gen = {k: np.tile(t[k], copies) for k in t if k.startswith("gen_")}
    gen["gen_bus"] = gen["gen_bus"] + np.repeat(np.arange(copies) * n, ng)
    if slack_active is not None:                      # the units on a former slack bus now hold the output the slack had in the solved case
        on = np.flatnonzero((t["gen_bus"] == slack) & (t["gen_status"] == 1))
        if on.size:
            share = np.full(on.size, float(slack_active) / on.size)
            for c in range(1, copies):
                gen["gen_pg"][c * ng + on] = share
    out.update(gen)
    for k in t:
        if k not in out:
            out[k] = np.tile(t[k], copies) if np.ndim(t[k]) and t[k].size in (n, nb, ng) else t[k].copy()
    return out
