"""Tables of the shared-factor solve (csrc/jg_symbolic.cpp: build_comp_tables -- what jg_comp.hip walks for the first iteration of a common-start batch),
replayed in numpy on the CPU: every row a record reads was written in an EARLIER level (all rows of a level run concurrently on the device), the dense
top takes exactly the pivots above the split, and the replayed solve -- bottom levels forward, dense inverse of the top's Schur complement, bottom levels
backward, on the compact factor the device forms (Lh D^-1 below, D^-1 U above the diagonal) -- equals numpy's solve of the oracle's Jacobian."""
import numpy as np
import pytest

from conftest import load_case
from plan_emulator import Replay, block_jacobian_from_csc, dsolve, dsolve_right


def _system(oracle, name):
    s = oracle.OracleSystem(load_case(name))
    a = oracle.OracleNR(s)
    a.mismatch()
    _, f0, _ = a.vectors()
    f0 = f0.copy()
    a.solve()
    J, _, _ = a.vectors()
    rowptr, col, A = block_jacobian_from_csc(s.n, s.colptr, s.rowval, a.type, a.pq, a.pvpq, a.jcolptr, a.jrowval, J)
    rhs = np.zeros((s.n, 2))
    for i in range(s.n):
        if a.pvpq[i]:
            rhs[i, 0] = f0[a.pvpq[i] - 1]
        if a.pq[i]:
            rhs[i, 1] = f0[a.pq[i] - 1]
    return s.n, rowptr, col, A, rhs


def _compact(plan, X):
    """what k_comp_pack stores per factor entry"""
    e_row, e_col, diag = plan.get("e_row"), plan.get("e_col"), plan.get("diag")
    M = np.zeros_like(X[:e_row.size])
    for e in range(e_row.size):
        r, c = e_row[e], e_col[e]
        if r == c:
            M[e] = X[e]
        elif r > c:
            M[e] = np.stack([dsolve_right(X[diag[c]], X[e][0]), dsolve_right(X[diag[c]], X[e][1])])
        else:
            M[e] = np.stack([dsolve(X[diag[r]], X[e][:, 0]), dsolve(X[diag[r]], X[e][:, 1])], axis=1)
    return M


def _sweep(seg, rec, M, W, written, level_base, bwd, rhs=None, perm_out=None):
    """one sweep by levels; `written[row]` = global level at which the row became final (race check)"""
    for lv in sorted(set(seg[:, 4])):
        new = {}
        for base, nchunks, wpi, rpw, level, _last, _items, _pad in seg[seg[:, 4] == lv]:
            for c in range(nchunks):
                for item in range(16 // wpi):
                    tot, target = None, -1
                    for sub in range(wpi):
                        r0 = base + (c * 16 + item * wpi + sub) * rpw
                        rr = rec[r0:r0 + rpw]
                        if rr[0, 0] < 0:
                            assert sub == 0 or target < 0
                            continue
                        k, src, dg = rr[0, 0], rr[0, 1], rr[0, 2]
                        acc = np.zeros(2)
                        if sub == 0:
                            target = k
                            if bwd:
                                assert written[k] >= 0 and written[k] < level_base + lv, f"row {k}: its forward value is not final"
                                acc = dsolve(M[dg], W[k])
                            else:
                                acc = rhs[src].copy()
                        else:
                            assert k == target
                        for q in rr:
                            for t in range(q[3]):
                                ent, row = q[4 + 2 * t], q[5 + 2 * t]
                                assert 0 <= written[row] < level_base + lv, f"level {lv}: row {row} read before it is final"
                                acc -= M[ent] @ W[row]
                        tot = acc if tot is None else tot + acc
                    if target >= 0:
                        assert target not in new, f"row {target} scheduled twice in level {lv}"
                        new[target] = tot
        for k, v in new.items():
            W[k] = v
            if bwd:
                assert written[k] < level_base + lv
            written[k] = level_base + lv
    return W


@pytest.mark.parametrize("name,top_cap", [("case14test", 0), ("case30test", 4), ("case118", 8), ("case118", -1), ("case300", 40), ("case1354pegase", 64), ("case1354pegase", 0)])
def test_shared_factor_tables_replay(jg, oracle, name, top_cap):
    n, rowptr, col, A, rhs = _system(oracle, name)
    plan = jg._lib.Plan(n, rowptr, col, policy=1)
    X, _ = Replay(plan, inplace=True).factor(A, rhs)
    M = _compact(plan, X)
    perm = plan.get("perm")
    D = np.zeros((2 * n, 2 * n))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            D[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] = A[p]
    ref = np.linalg.solve(D, rhs.reshape(-1)).reshape(n, 2)

    def solve(tables, r, Sinv=None):
        info, top, (fseg, frec), (bseg, brec) = tables
        nt = int(info[0])
        W = np.full((n + nt, 2), np.nan)
        written = np.full(n + nt, -1)
        _sweep(fseg, frec, M, W, written, 0, False, rhs=r)
        bottom = np.setdiff1d(np.arange(n), top)
        assert (written[bottom] >= 0).all() and (written[n:] >= 0).all() and (written[top] < 0).all(), "every bottom row and every partial top row is scheduled, no top row"
        if nt:
            xt = Sinv @ W[n:].reshape(-1)
            lvl = int(fseg[:, 4].max()) + 1
            for t, k in enumerate(top):
                W[k] = xt[2 * t:2 * t + 2]
                written[k] = lvl
        base = int(fseg[:, 4].max()) + 2
        done_fwd = written.copy()
        _sweep(bseg, brec, M, W, written, base, True)
        assert (written[bottom] > done_fwd[bottom]).all(), "every bottom row is solved by the backward sweep"
        x = np.zeros((n, 2))
        x[perm] = W[:n]
        return x, W

    full = plan.comp_tables(-1)
    assert full[0][0] == 0 and full[1].size == 0
    x_full, _ = solve(full, rhs)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(x_full - ref).max() <= 1e-9 * scale
    tab = plan.comp_tables(top_cap)
    info, top = tab[0], tab[1]
    nt = int(info[0])
    cap = 512 if top_cap == 0 else max(top_cap, 0)
    assert nt <= cap and nt == top.size
    if top_cap < 0:
        return
    # the top is closed under taking ancestors (the first upper entry of a pivot row is its parent in the elimination tree)
    u_ptr, u_col = plan.get("u_ptr"), plan.get("u_col")
    in_top = np.zeros(n, dtype=bool)
    in_top[top] = True
    for k in top:
        assert all(in_top[c] for c in u_col[u_ptr[k]:u_ptr[k + 1]])
    assert info[2] <= full[0][2] and info[3] <= full[0][3]
    if nt:
        assert info[2] < full[0][2], "the levels above the split are gone"
    # Sinv the way the device forms it: unit right-hand sides on the top rows through the level-only tables
    Sinv = np.zeros((2 * nt, 2 * nt))
    for t in range(nt):
        for c in range(2):
            r = np.zeros((n, 2))
            r[perm[top[t]], c] = 1.0
            _, Wf = solve(full, r)
            Sinv[:, 2 * t + c] = Wf[top].reshape(-1)
    x, _ = solve(tab, rhs, Sinv)
    assert np.abs(x - ref).max() <= 1e-9 * scale, (name, top_cap, np.abs(x - ref).max())
