"""The tables of a SINGLE instance (round 6; jg_symbolic.hpp: SingleTables, SINGLE_FACT_LEVELS) replayed on the CPU: the factorisation below the top with a thread
per item (k_fact1_bottom / k_fact1_partial) and the backward sweep with rows as lanes (k_bwd1_top / k_bwd1_bottom).  Every operand an item reads must have been
written in an earlier level BY THE SAME WORKGROUP (or before the launch), and the solve must equal numpy's."""
import numpy as np
import pytest

from plan_emulator import Replay, dsolve


def _system(n, edges, rng):
    adj = {(i, i) for i in range(n)} | {(a, b) for a, b in edges} | {(b, a) for a, b in edges}
    rowptr, col = [0], []
    for i in range(n):
        col += sorted(j for (r, j) in adj if r == i)
        rowptr.append(len(col))
    rowptr, col = np.array(rowptr, dtype=np.int32), np.array(col, dtype=np.int32)
    dense = np.zeros((2 * n, 2 * n))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            dense[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] = rng.standard_normal((2, 2))
    dense += np.diag(np.abs(dense).sum(axis=1) + 1.0)
    A = np.array([dense[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] for i in range(n) for p in range(rowptr[i], rowptr[i + 1])])
    return rowptr, col, dense, A


def _cases(rng):
    cases = [(12, [(i, i + 1) for i in range(11)]), (9, [(0, i) for i in range(1, 9)]),
             (24, [(i, j) for i in range(24) for j in range(i + 1, 24)]),
             (20, [(i, i + 1) for i in range(0, 19)] + [(i, i + 10) for i in range(10)]),
             (40, [(i, j) for i in range(40) for j in range(i + 1, min(i + 4, 40))])]
    for _ in range(10):
        n = int(rng.integers(30, 160))
        m = int(rng.integers(n, 2 * n))
        cases.append((n, [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(m)]))
    # a transmission-like graph: a ring of rings with a few chords (deep tree, many small subtrees)
    e = []
    for r in range(12):
        base = r * 20
        e += [(base + i, base + (i + 1) % 20) for i in range(20)] + [(base + 3, (base + 20 + 7) % 240)]
        e += [(base + i, base + i + 5) for i in range(0, 15, 5)]
    cases.append((240, e))
    return cases


def _single_backward(plan, X, Y, nE):
    """Replay of k_bwd1_top + k_bwd1_bottom.  X: factor incl. Jordan rows, Y: y (bottom rows) / y' (top rows) in pivot order."""
    g = plan.get
    info = g(90)
    assert info[0] == 1
    n_top, n_lev, n_bottom, n_wg, b_levels, rows_per_wg = (int(v) for v in info[1:7])
    flat = 0
    t_row, t_ptr, t_term, t_level = g(91).reshape(-1, 4), g(92).reshape(-1, 2), g(93), g(94)
    b_wg, b_row, b_term = g(95).reshape(-1, 2), g(96).reshape(-1, 6), g(97).reshape(-1, 2)
    n = plan.n
    W = Y.copy()
    done = np.zeros(n, dtype=bool)
    xs = np.full((n_top, 2), np.nan)
    assert t_level[0] == 0 and t_level[n_lev] == n_top
    for L in range(n_lev):
        new = []
        for row in range(t_level[L], t_level[L + 1]):
            k, bus, dg, nt = (int(v) for v in t_row[row])
            jb, sb = (int(v) for v in t_ptr[row])
            y = W[k].copy()
            for t in range(nt):
                slot = int(t_term[sb + t])
                assert slot < t_level[L], "a top row reads a column of its own or a later level"
                y -= X[nE + jb + t] @ xs[slot]
            new.append((row, k, dsolve(X[dg], y)))
        for row, k, x in new:
            xs[row] = x; W[k] = x; done[k] = True
    assert not np.isnan(xs).any()
    # the same levels with the TERMS as lanes (k_bwd1_top2): a level's compact blocks follow each other in (row, column) order
    if info[7]:
        t_jb, t_cslot, t_toff = g(98).reshape(-1, 2), g(99), g(100)
        xs2 = np.full((n_top, 2), np.nan)
        covered = 0
        for L in range(n_lev):
            g0, nt = (int(v) for v in t_jb[L])
            assert nt <= info[8] and t_level[L + 1] - t_level[L] <= info[9]
            prod = np.zeros((nt, 2))
            for t in range(nt):
                slot = int(t_cslot[g0 + t])
                assert slot < t_level[L], "a term reads a column of its own or a later level"
                prod[t] = X[nE + g0 + t] @ xs2[slot]
            for row in range(t_level[L], t_level[L + 1]):
                k, bus, dg, e = (int(v) for v in t_row[row])
                o = int(t_toff[row])
                assert e == 0 or (int(t_ptr[row][0]) == g0 + o and o + e <= nt), "the row's blocks are not where its products are"
                xs2[row] = dsolve(X[dg], Y[k] - prod[o: o + e].sum(axis=0))
            covered += nt
        assert covered == len(t_cslot) or (covered == 0 and len(t_cslot) == 1)
        assert np.abs(xs2 - xs).max() <= 1e-10 * max(1.0, np.abs(xs).max())
    assert n_top + n_bottom == n
    for w in range(n_wg):
        r0, levels, r1 = int(b_wg[w, 0]), int(b_wg[w, 1]), int(b_wg[w + 1, 0])
        assert 0 < r1 - r0 <= rows_per_wg
        lx = np.full((rows_per_wg, 2), np.nan)
        for L in range(levels):
            new = []
            for t in range(r1 - r0):
                k, bus, dg, nt, tp, lev = (int(v) for v in b_row[r0 + t])
                if lev != L:
                    continue
                y = W[k].copy()
                for q in range(nt):
                    ent, c = (int(v) for v in b_term[tp + q])
                    if c >= 0:
                        assert done[c] and c not in [int(b_row[r, 0]) for r in range(r0, r1)], "a top column must be a finished top row"
                        y -= X[ent] @ W[c]
                    else:
                        assert not np.isnan(lx[-(c + 1)]).any(), "a bottom row reads a row of its workgroup that is not finished yet"
                        y -= X[ent] @ lx[-(c + 1)]
                new.append((t, k, dsolve(X[dg], y)))
            for t, k, x in new:
                lx[t] = x; W[k] = x
        assert all(int(b_row[r0 + t, 5]) < levels for t in range(r1 - r0))
    return W


@pytest.mark.parametrize("top_level,soft,mmin", [(2, 12, 8), (3, 26, 12), (5, 24, 12)])
def test_single_instance_tables_replay(jg, top_level, soft, mmin):
    rng = np.random.default_rng(1000 * top_level + soft)
    used = 0
    for ci, (n, edges) in enumerate(_cases(rng)):
        rowptr, col, dense, A = _system(n, edges, rng)
        policy = 1 | 4 | top_level << 8 | soft << 16 | 127 << 24 | 1 << 49 | mmin << 54 | 1 << 60
        plan = jg._lib.Plan(n, rowptr, col, policy=policy)
        hdr, _, _, task_of, info = plan.top_tables()
        if not (hdr.shape[0] and info[6]):
            assert plan.get(85)[0] == 0 and plan.get(90)[0] == 0       # no top tasks: nothing to build
            continue
        rhs = rng.standard_normal((n, 2))
        exact = np.linalg.solve(dense, rhs.reshape(-1))
        nE = plan.get("e_row").size
        perm = plan.get("perm")
        # ---- the factorisation through the thread-per-item tables (same top tasks), then the plain Jordan sweep
        assert plan.get(85)[0] == 1
        rs = Replay(plan, inplace=True, prefactor=True, producer=True, jordan=True, single=True)
        X, Y = rs.factor(A, rhs)
        x = rs.backsolve(X, Y)
        assert np.abs(x.reshape(-1) - exact).max() <= 1e-11
        # the level tables give the same factor up to the summation order of long lists
        rl = Replay(plan, inplace=True, prefactor=True, producer=True, jordan=True)
        Xl, Yl = rl.factor(A, rhs)
        keep = ~np.isnan(Xl).any(axis=(1, 2))
        assert np.allclose(X[keep], Xl[keep], rtol=1e-12, atol=1e-13) and np.allclose(Y, Yl, rtol=1e-12, atol=1e-13)
        # ---- the backward sweep with rows as lanes
        assert plan.get(90)[0] == 1
        Wx = _single_backward(plan, X, Y, nE)
        xo = np.zeros((n, 2)); xo[perm] = Wx
        assert np.abs(xo.reshape(-1) - exact).max() <= 1e-11
        used += 1
    assert used >= 8


def test_single_tables_on_the_headline_grid(jg):
    """ACTIVSg10k with the policy a handle of ONE scenario asks for: both sets of tables are granted, every bottom workgroup holds whole subtrees."""
    s = jg.powerSystem("case_ACTIVSg10k")
    jg.acModel_(s)
    Y = s.model.ac.nodalMatrix
    policy = 1 | 4 | (26 << 16 | 127 << 24) | 1 << 49 | 12 << 54 | 1 << 60
    plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=policy)
    finfo, binfo = plan.get(85), plan.get(90)
    assert finfo[0] == 1 and binfo[0] == 1
    assert binfo[1] + binfo[3] == Y.n and 4 <= binfo[2] <= 12 and binfo[5] <= 6
    assert binfo[7] == 1 and binfo[8] <= 6 * 1024 and binfo[9] <= 1024        # the terms-as-lanes sweep of the top is granted on the headline grid
    f1_wg = plan.get(83).reshape(-1, int(finfo[2]) + 1)
    assert f1_wg.shape[0] == finfo[1] and (np.diff(f1_wg, axis=1) >= 0).all()
    assert (f1_wg[:, -1] - f1_wg[:, 0]).max() <= max(int(finfo[3]), 64 * 8)
