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
    X = rp.factor(A)
    rhs = np.zeros((s.n, 2))
    for i in range(s.n):
        if a.pvpq[i]:
            rhs[i, 0] = f0[a.pvpq[i] - 1]
        if a.pq[i]:
            rhs[i, 1] = f0[a.pq[i] - 1]
    x = rp.solve(X, rhs)
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
    # every term multiplies L(i,k) by U(k,j) with k < min(i,j)
    t_a, t_b = plan.get("t_a"), plan.get("t_b")
    ent = np.repeat(np.arange(e_row.size), np.diff(t_ptr))
    assert np.all(e_row[t_a] == e_row[ent]) and np.all(e_col[t_b] == e_col[ent])
    assert np.all(e_col[t_a] == e_row[t_b])
    assert np.all(e_col[t_a] < np.minimum(e_row[ent], e_col[ent]))
    # level monotonicity: an entry is strictly above all its sources
    lev = plan.get("e_level")
    assert np.all(lev[ent] > lev[t_a]) and np.all(lev[ent] > lev[t_b])
    # schedules cover every item exactly once
    for kind, nitems in (("lu", e_row.size), ("fwd", Y.n), ("bwd", Y.n)):
        sch = plan.schedule(kind)
        assert sorted(sch["items"].tolist()) == list(range(nitems))
        assert sch["launches"][0, 0] == 0 and sch["launches"][-1, 1] == sch["task_ptr"].size - 1
        assert np.all(sch["launches"][1:, 0] == sch["launches"][:-1, 1])


def test_plan_rejects_unsymmetric_pattern(jg):
    rowptr = np.array([0, 2, 3], dtype=np.int32)
    col = np.array([0, 1, 1], dtype=np.int32)
    with pytest.raises(jg._lib.JGridError):
        jg._lib.Plan(2, rowptr, col)
