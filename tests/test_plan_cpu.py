"""Static schedule of the block LU engine, replayed on the CPU in numpy (no GPU):
race-freedom of every launch/task/step, and numerical correctness of the replayed factorisation
against the oracle's pivoting LU on real Newton-Raphson Jacobians."""
import numpy as np
import pytest

from conftest import load_case
from plan_emulator import Replay, block_jacobian_from_csc


def _oracle_jacobian(oracle, name):
    s = oracle.OracleSystem(load_case(name))
    a = oracle.OracleNR(s)
    a.mismatch()
    _, f0, _ = a.vectors()
    a.solve()
    J, _, inc = a.vectors()
    return s, a, J, f0, inc


@pytest.mark.parametrize("name", ["case14test", "case30test", "case118", "case1354pegase"])
def test_schedule_replay_matches_oracle_increment(jg, oracle, name):
    s, a, J, f0, inc = _oracle_jacobian(oracle, name)
    rowptr, col, A = block_jacobian_from_csc(s.n, s.colptr, s.rowval, a.type, a.pq, a.pvpq, a.jcolptr, a.jrowval, J)
    plan = jg._lib.Plan(s.n, rowptr, col)
    rp = Replay(plan)
    rhs = np.zeros((s.n, 2))
    for i in range(s.n):
        if a.pvpq[i]:
            rhs[i, 0] = f0[a.pvpq[i] - 1]
        if a.pq[i]:
            rhs[i, 1] = f0[a.pq[i] - 1]
    X, Y = rp.factor(A, rhs)
    x = rp.backsolve(X, Y)
    got = np.zeros(a.dim)
    for i in range(s.n):
        if a.pvpq[i]:
            got[a.pvpq[i] - 1] = x[i, 0]
        else:
            assert x[i, 0] == 0.0
        if a.pq[i]:
            got[a.pq[i] - 1] = x[i, 1]
        else:
            assert x[i, 1] == 0.0
    assert np.abs(got - inc).max() <= 1e-10 * max(1.0, np.abs(inc).max())


def test_plan_structure_invariants(jg):
    s = jg.powerSystem(load_case("case_ACTIVSg10k"))
    jg.acModel_(s)
    Y = s.model.ac.nodalMatrix
    plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1)
    perm = plan.get("perm")
    assert sorted(perm) == list(range(Y.n))
    e_row, e_col, e_src, t_ptr = plan.get("e_row"), plan.get("e_col"), plan.get("e_src"), plan.get("t_ptr")
    assert (e_src >= 0).sum() == Y.nnz and np.unique(e_src[e_src >= 0]).size == Y.nnz
    # symmetric factor pattern
    pairs = set(zip(e_row.tolist(), e_col.tolist()))
    assert all((c, r) in pairs for r, c in pairs)
    # every term multiplies Lh(i,k) * Dinv(k) * U(k,j) with k < min(i,j)
    t_a, t_d, t_b = plan.get("t_a"), plan.get("t_d"), plan.get("t_b")
    ent = np.repeat(np.arange(e_row.size), np.diff(t_ptr))
    assert np.all(e_row[t_a] == e_row[ent]) and np.all(e_col[t_b] == e_col[ent])
    assert np.all(e_col[t_a] == e_row[t_b])
    assert np.all(e_row[t_d] == e_col[t_a]) and np.all(e_col[t_d] == e_col[t_a])
    assert np.all(e_col[t_a] < np.minimum(e_row[ent], e_col[ent]))
    # level monotonicity: an entry is strictly above all its sources
    lev = plan.get("e_level")
    assert np.all(lev[ent] > lev[t_a]) and np.all(lev[ent] > lev[t_b]) and np.all(lev[ent] > lev[t_d])
    # schedules cover every item exactly once; launches tile the task and item ranges; the device's
    # flattened addressing (item_begin + task*chunk) matches the task/step lists
    for kind, nitems in (("fact", e_row.size + Y.n), ("bwd", Y.n)):
        sch = plan.schedule(kind)
        assert sorted(sch["items"].tolist()) == list(range(nitems))
        L = sch["launches"]
        assert L[0, 0] == 0 and L[-1, 1] == sch["task_ptr"].size - 1
        assert np.all(L[1:, 0] == L[:-1, 1])
        assert L[0, 5] == 0 and L[-1, 6] == nitems and np.all(L[1:, 5] == L[:-1, 6])
        for t0, t1, waves, wpi, chunk, ib, ie, fused in L:
            if fused:                            # one task walking several narrow levels as steps
                assert t1 == t0 + 1 and waves == 16
                s0, s1 = sch["task_ptr"][t0], sch["task_ptr"][t1]
                assert sch["step_ptr"][s0] == ib and sch["step_ptr"][s1] == ie and s1 - s0 >= 2
                for st in range(s0, s1):
                    w, cnt = sch["step_wpi"][st], sch["step_ptr"][st + 1] - sch["step_ptr"][st]
                    assert 16 % w == 0 and cnt >= 1
                continue
            assert waves % wpi == 0 and (wpi == 1 or chunk == waves // wpi) and chunk % (waves // wpi) == 0
            for k, t in enumerate(range(t0, t1)):
                s0, s1 = sch["task_ptr"][t], sch["task_ptr"][t + 1]
                assert s1 == s0 + 1
                assert sch["step_ptr"][s0] == ib + k * chunk and sch["step_ptr"][s1] == min(ib + (k + 1) * chunk, ie)


def test_plan_rejects_unsymmetric_pattern(jg):
    rowptr = np.array([0, 2, 3], dtype=np.int32)
    col = np.array([0, 1, 1], dtype=np.int32)
    with pytest.raises(jg._lib.JGridError):
        jg._lib.Plan(2, rowptr, col)


def test_replay_is_stable_on_ill_conditioned_gain(jg, oracle):
    """Gain matrix of the squared-ammeter + weak-PMU set (cond ~7e9, pivot blocks up to cond ~2e9,
    test/stateEstimation/analysis.jl:43-49): the factored-diagonal block LU must match a dense solve
    as well as Cholesky does (an explicit 2x2 inverse loses the solution here)."""
    import scipy.sparse as sp
    from test_oracle_se import se_case14
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", bus=True, frm=False, to=False, variance=1.0, polar=True)
    oracle.add_from_power_flow(tab, osys, vm, va, "ammeter", variance=1e-4, square=True)
    gn = oracle.OracleGN(osys, tab)
    gn.increment()
    v = gn.vectors()
    n, sl = osys.n, osys.slack - 1
    H = sp.csc_matrix((v["jacobian"], gn.hrowval - 1, gn.hcolptr - 1), shape=(gn.m, 2 * n)).toarray()
    H[:, sl] = 0
    W = gn.precision_dense()
    G = H.T @ W @ H
    G[sl, sl] = 1
    rhs = H.T @ W @ v["residual"]
    rhs[sl] = 0
    dx = np.linalg.solve(G, rhs)
    assert np.linalg.cond(G) > 1e9
    perm = np.empty(2 * n, dtype=int)
    perm[0::2], perm[1::2] = np.arange(n), n + np.arange(n)
    Gb, rb = G[np.ix_(perm, perm)], rhs[perm]
    rowptr, col, A = [0], [], []
    for i in range(n):
        for j in range(n):
            blk = Gb[2 * i:2 * i + 2, 2 * j:2 * j + 2]
            if i == j or np.any(blk != 0):
                col.append(j)
                A.append(blk)
        rowptr.append(len(col))
    rp = Replay(jg._lib.Plan(n, np.array(rowptr), np.array(col)))
    X, Y = rp.factor(np.array(A), rb.reshape(n, 2))
    x = rp.backsolve(X, Y)
    xs = np.concatenate([x[:, 0], x[:, 1]])
    assert np.abs(xs - dx).max() <= 1e-6 * np.abs(dx).max()       # ~ cond * eps
    assert np.abs(xs - v["increment"]).max() <= 1e-6 * np.abs(dx).max()
