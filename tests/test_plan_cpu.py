"""Static schedule of the block LU engine, replayed on the CPU in numpy (no GPU):
race-freedom of every launch/task/step, and numerical correctness of the replayed factorisation
against the oracle's pivoting LU on real Newton-Raphson Jacobians."""
import numpy as np
import pytest

from conftest import load_case
from plan_emulator import Replay, block_jacobian_from_csc


@pytest.fixture(autouse=True)
def _order_check(monkeypatch):
    """Every plan of this module is built with JG_ORDER_CHECK: each incrementally updated fill count of the min-fill ordering is compared with
    a recount (jg_symbolic.cpp: elimination_order; a disagreement fails the plan)."""
    monkeypatch.setenv("JG_ORDER_CHECK", "1")


def _oracle_jacobian(oracle, name):
    s = oracle.OracleSystem(load_case(name))
    a = oracle.OracleNR(s)
    a.mismatch()
    _, f0, _ = a.vectors()
    a.solve()
    J, _, inc = a.vectors()
    return s, a, J, f0, inc


@pytest.mark.parametrize("inplace", [False, True, "producer finishes level 0", "pre tables", "producer finishes level 0, Jordan rows",
                                     "producer finishes level 0, Jordan rows, tasks", "pre tables, tasks", "tasks", "tasks of one round"])
@pytest.mark.parametrize("name", ["case14test", "case30test", "case118", "case1354pegase"])
def test_schedule_replay_matches_oracle_increment(jg, oracle, name, inplace):
    s, a, J, f0, inc = _oracle_jacobian(oracle, name)
    rowptr, col, A = block_jacobian_from_csc(s.n, s.colptr, s.rowval, a.type, a.pq, a.pvpq, a.jcolptr, a.jrowval, J)
    pre = isinstance(inplace, str)                            # policy bit 2: level 0 by the producer / by the plan's PRE tables
    jordan = pre and "Jordan" in inplace                      # policy bit 49: what jg_nr_create asks for (granted where the plan has top tasks)
    tasks = isinstance(inplace, str) and "tasks" in inplace   # policy bit 50: the factorisation tables as TASKS (shared operands staged in LDS)
    pre = pre and ("level 0" in inplace or "pre tables" in inplace)
    plan = jg._lib.Plan(s.n, rowptr, col, policy=((1 | 4) if pre else (1 if inplace else 0)) | (1 << 49 if jordan else 0) | (1 << 50 if tasks else 0) |
                        (1 << 51 if tasks and "one round" in inplace else 0))
    assert bool(plan.top_tables()[4][8]) == tasks
    if jordan:
        jordan = bool(plan.top_tables()[4][6])
        assert jordan == (plan.top_tables()[0].shape[0] > 0) and (jordan or s.n < 100)
    rp = Replay(plan, inplace=bool(inplace), prefactor=pre, producer=pre and inplace.startswith("producer finishes level 0"), jordan=jordan)
    if tasks and s.n > 1000:                                  # sharing pays: fewer staged operands than terms, (almost) no three-operand term left
        info = plan.top_tables()[4]
        seg, rec = plan.replay_tables("fact")
        terms = (rec[rec[:, 0] & 7 != 7, 3] & 0xff).sum()
        assert 0 < info[10] < 0.6 * terms and info[11] < 0.02 * terms
    if pre:
        plain = jg._lib.Plan(s.n, rowptr, col, policy=1)
        assert plan.get("e_level").max() == plain.get("e_level").max() - 1      # every level moved down by one
        assert plan.get("pre_pivot").sum() > s.n // 4
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
    # replay tables: segments in level order, `last` closes every level, at most one segment per wpi class and
    # level, records tile exactly, every item appears once (leader wave) and every term exactly once
    src_entry = plan.get("src_entry")
    assert np.array_equal(e_src[src_entry], np.arange(Y.nnz))
    for kind, nitems, T in (("fact", e_row.size + Y.n, 4), ("bwd", Y.n, 6)):
        seg, rec = plan.replay_tables(kind)
        base, nchunks, wpi, rpw, level, last = (seg[:, k] for k in range(6))
        assert np.all(np.diff(level) >= 0) and last[-1] == 1
        assert np.all(last[:-1] == (level[1:] > level[:-1]))
        assert np.all(np.isin(wpi, [-1, 0, 1, 2, 4, 8, 16])) and (kind == "bwd" or np.all(wpi > 0))     # wpi 0 / -1 = backward chain tasks (general / small)
        for l in np.unique(level):
            assert np.unique(wpi[level == l]).size == (level == l).sum() <= 7
        count = np.where(wpi > 0, nchunks * 16 * rpw, nchunks)                                      # one record per chain task
        assert base[0] == 0 and np.all(base[1:] == base[:-1] + count[:-1])
        assert rec.shape[0] == base[-1] + count[-1]
        rows = np.concatenate([np.arange(b, b + c) for b, c, w in zip(base, count, wpi) if w > 0])
        assert np.all(rec[rows, 3] <= T) and np.all(rec[rows][rec[rows, 0] < 0, 3] == 0)
    seg, rec = plan.replay_tables("fact")
    lead = np.concatenate([np.arange(b, b + c * 16 * r, r * w) for b, c, w, r in seg[:, :4]])   # first record of each leader wave
    lead = lead[rec[lead, 0] >= 0]
    ids = np.where(rec[lead, 0] == 3, e_row.size + rec[lead, 1], rec[lead, 1])
    # multifrontal top: task-owned fill-in without a term of a bottom pivot starts from zero inside its task (no level item);
    # the terms of task pivots run inside the tasks
    hdr, tdata, launches, task_of, info = plan.top_tables()
    assert hdr.shape[0] > 10 and info[0] > 0 and (task_of >= 0).sum() > 100
    ent = np.repeat(np.arange(e_row.size), np.diff(t_ptr))
    bottom_term = task_of[e_col[t_a]] < 0
    has_bottom = np.zeros(e_row.size, dtype=bool)
    has_bottom[ent[bottom_term]] = True
    skipped = (task_of[np.minimum(e_row, e_col)] >= 0) & ~has_bottom & (e_src < 0)
    assert sorted(ids.tolist()) == sorted(np.flatnonzero(~skipped).tolist() + list(range(e_row.size, e_row.size + Y.n)))
    l_ptr, l_col = plan.get("l_ptr"), plan.get("l_col")
    lrow = np.repeat(np.arange(Y.n), np.diff(l_ptr))
    rhs_bottom = (task_of[l_col] < 0).sum()
    assert np.all(task_of[lrow[task_of[l_col] >= 0]] >= 0)                                         # a task pivot only feeds task rows
    assert rec[rec[:, 0] >= 0, 3].sum() == bottom_term.sum() + rhs_bottom
    u_ptr = plan.get("u_ptr")
    s_top = np.diff(u_ptr)[task_of >= 0].astype(np.int64)
    assert info[2] == (s_top * (s_top + 1)).sum() == (~bottom_term).sum() + (l_ptr[-1] - rhs_bottom)
    assert seg[-1, 4] <= info[0] and launches[-1, 3] + seg[-1, 4] < 40                              # 84 dependent launches before
    # no top tasks on request: the pure level schedule
    plan0 = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=255 << 8)
    seg0, rec0 = plan0.replay_tables("fact")
    assert plan0.top_tables()[0].shape[0] == 0 and seg0[-1, 4] == lev.max() and rec0[rec0[:, 0] >= 0, 3].sum() == t_ptr[-1] + l_ptr[-1]
    seg, rec = plan.replay_tables("bwd")
    rows_seg = seg[seg[:, 2] > 0]
    lead = np.concatenate([np.arange(b, b + c * 16 * r, r * w) for b, c, w, r in rows_seg[:, :4]])
    lead = lead[rec[lead, 0] >= 0]
    chain = plan.get("bwd_chain")
    in_chains = []
    for b, c in seg[seg[:, 2] <= 0][:, :2]:
        for nb, nE, off, wpr in rec[b:b + c, :4]:
            in_chains += chain[off:off + 3 * nb:3].tolist()
    assert sorted(rec[lead, 0].tolist() + in_chains) == list(range(Y.n))                            # every pivot exactly once
    assert len(in_chains) > 300 and seg[-1, 4] < 80                                                 # 157 row levels -> chain levels
    # in-place policy: entries the assembly already finalised (off-diagonal, no update terms) are not scheduled
    plan1 = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=1)
    seg1, rec1 = plan1.replay_tables("fact")
    lead1 = np.concatenate([np.arange(b, b + c * 16 * r, r * w) for b, c, w, r in seg1[:, :4]])
    lead1 = lead1[rec1[lead1, 0] >= 0]
    top_owned = task_of[np.minimum(e_row, e_col)] >= 0
    skipped = np.where(top_owned, ~has_bottom, (e_src >= 0) & (np.diff(t_ptr) == 0) & (e_row != e_col))   # task-owned: only partial sums are scheduled
    assert lead1.size == e_row.size + Y.n - skipped.sum() and skipped.sum() > 20000
    ent1 = rec1[lead1][rec1[lead1, 0] != 3]
    assert not np.any(skipped[ent1[:, 1]])
    assert not np.any(top_owned[ent1[ent1[:, 0] == 2, 1]])                   # no level item factorises a task-owned diagonal block
    assert np.all(ent1[ent1[:, 2] >= 0, 2] == ent1[ent1[:, 2] >= 0, 1])       # src names the entry itself


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
    for policy in (0, 1 << 50):                                   # wave records; tasks (a staged operand is Lh D^-1 through the stored 2x2 LU: a right solve, not an inverse)
        rp = Replay(jg._lib.Plan(n, np.array(rowptr), np.array(col), policy=policy))
        assert rp.fact_tasks == bool(policy)
        X, Y = rp.factor(np.array(A), rb.reshape(n, 2))
        x = rp.backsolve(X, Y)
        xs = np.concatenate([x[:, 0], x[:, 1]])
        assert np.abs(xs - dx).max() <= 1e-6 * np.abs(dx).max()       # ~ cond * eps
        assert np.abs(xs - v["increment"]).max() <= 1e-6 * np.abs(dx).max()


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("name", ["case14test", "case118", "case1354pegase"])
def test_selected_inverse_replay_matches_dense_inverse(jg, name, symmetric):
    """Takahashi recursion on the factor pattern (tables of the bad-data test): a random symmetric, diagonally dominant
    block matrix on the Ybus pattern; every Z block of the upper pattern + diagonal equals the dense inverse."""
    s = jg.powerSystem(load_case(name))
    jg.acModel_(s)
    Y = s.model.ac.nodalMatrix
    n = Y.n
    rowptr, col = (Y.colptr - 1).astype(np.int32), (Y.rowval - 1).astype(np.int32)
    rng = np.random.default_rng(5)
    dense = np.zeros((2 * n, 2 * n))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            j = col[p]
            if j > i:
                b = rng.standard_normal((2, 2))
                dense[2 * i:2 * i + 2, 2 * j:2 * j + 2] = b
                dense[2 * j:2 * j + 2, 2 * i:2 * i + 2] = b.T
    dense += np.diag(np.abs(dense).sum(axis=1) + 1.0)
    A = np.array([dense[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] for i in range(n) for p in range(rowptr[i], rowptr[i + 1])])
    plan = jg._lib.Plan(n, rowptr, col, policy=3 if symmetric else 1)
    rp = Replay(plan, inplace=True, symmetric=symmetric)
    rhs = rng.standard_normal((n, 2))
    X, Yf = rp.factor(A, rhs)
    xs = rp.backsolve(X, Yf)                                     # LDL' through transposed reads solves the system too
    assert np.abs(xs.reshape(-1) - np.linalg.solve(dense, rhs.reshape(-1))).max() <= 1e-12
    Z = rp.selected_inverse(X)
    inv = np.linalg.inv(dense)
    perm, e_row, e_col = plan.get("perm"), plan.get("e_row"), plan.get("e_col")
    worst = 0.0
    for e in np.flatnonzero(e_col >= e_row):
        i, j = perm[e_row[e]], perm[e_col[e]]
        worst = max(worst, np.abs(Z[e] - inv[2 * i:2 * i + 2, 2 * j:2 * j + 2]).max())
    assert worst <= 1e-12 * np.abs(inv).max()
    seg, rec = plan.replay_tables("sel")
    assert seg[-1, 4] == 2 * plan.get("bwd_level").max()


def _solve_with_plan(jg, n, edges, rng, symmetric=False, top=0, prefactor=None):
    adj = {(i, i) for i in range(n)} | {(a, b) for a, b in edges} | {(b, a) for a, b in edges}
    rowptr, col = [0], []
    for i in range(n):
        cols = sorted(j for (r, j) in adj if r == i)
        col += cols
        rowptr.append(len(col))
    rowptr, col = np.array(rowptr, dtype=np.int32), np.array(col, dtype=np.int32)
    dense = np.zeros((2 * n, 2 * n))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            j = col[p]
            if not symmetric or j >= i:
                b = rng.standard_normal((2, 2))
                dense[2 * i:2 * i + 2, 2 * j:2 * j + 2] = b
                if symmetric and j > i:
                    dense[2 * j:2 * j + 2, 2 * i:2 * i + 2] = b.T
    if symmetric:
        dense = (dense + dense.T) / 2
    dense += np.diag(np.abs(dense).sum(axis=1) + 1.0)
    A = np.array([dense[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] for i in range(n) for p in range(rowptr[i], rowptr[i + 1])])
    pre = prefactor is not None
    plan = jg._lib.Plan(n, rowptr, col, policy=(3 if symmetric else 1) | (4 if pre else 0) | top)
    assert sorted(plan.get("perm")) == list(range(n))
    rp = Replay(plan, inplace=True, symmetric=symmetric, prefactor=pre, producer=bool(prefactor))
    rhs = rng.standard_normal((n, 2))
    X, Yf = rp.factor(A, rhs)
    x = rp.backsolve(X, Yf)
    assert np.abs(x.reshape(-1) - np.linalg.solve(dense, rhs.reshape(-1))).max() <= 1e-11
    if plan.top_tables()[4][6]:                                 # a Jordan plan carries both sweeps: the same system through the Jordan rows
        rj = Replay(plan, inplace=True, symmetric=symmetric, prefactor=pre, producer=bool(prefactor), jordan=True)
        Xj, Yj = rj.factor(A, rhs)
        xj = rj.backsolve(Xj, Yj)
        assert np.abs(xj.reshape(-1) - np.linalg.solve(dense, rhs.reshape(-1))).max() <= 1e-11
        nE = plan.get("e_row").size
        keep = ~np.isnan(Xj[:nE]).any(axis=(1, 2))
        lower_or_diag = plan.get("e_row") >= plan.get("e_col")
        assert np.array_equal(Xj[:nE][lower_or_diag], X[:nE][lower_or_diag]), "Lh and D are untouched by the Jordan elimination"
        assert keep[lower_or_diag].all()
    if not (top >> 50) & 1:                                     # ... and through the factorisation TASKS of the same policy (bit 50), filled to 1, 2, 3 rounds
        rounds = 1 + (n + len(edges)) % 3
        tp = _solve_with_plan(jg, n, edges, np.random.default_rng(n), symmetric, top | 1 << 50 | rounds << 51, prefactor)
        assert int(tp.top_tables()[4][8]) == 1 and int(tp.top_tables()[4][9]) == rounds
        for name in ("perm", "e_row", "e_col", "t_ptr", "e_level"):
            assert np.array_equal(tp.get(name), plan.get(name)), "the tasks change the tables of the factorisation, not the analysis"
    return plan


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("top_level,soft,struct_min", [(1, 4, 0), (2, 8, 0), (2, 64, 0), (3, 16, 0), (6, 24, 3), (9, 47, 2)])
def test_top_tasks_on_small_and_random_graphs(jg, monkeypatch, symmetric, top_level, soft, struct_min):
    """The multifrontal top pushed down to the leaves (every pivot / every pivot above level 1 or 2 in a task; fronts cut at
    4, 8, 16 and 64 blocks), and selected by front size (every pivot with struct_min neighbours at elimination and its
    ancestors, whatever its level): tasks are connected pieces of the tree, extend-add maps, partial level items and the update
    stack must stay consistent."""
    if struct_min:
        monkeypatch.setenv("JG_TOP_STRUCT", str(struct_min))
    rng = np.random.default_rng(100 * top_level + soft)
    cases = [(1, []), (2, [(0, 1)]), (12, [(i, i + 1) for i in range(11)]), (9, [(0, i) for i in range(1, 9)]),
             (8, [(i, j) for i in range(8) for j in range(i + 1, 8)]),
             (20, [(i, i + 1) for i in range(0, 19)] + [(i, i + 10) for i in range(10)]),
             (40, [(i, j) for i in range(40) for j in range(i + 1, min(i + 4, 40))])]
    for _ in range(6):
        n = int(rng.integers(20, 90))
        m = int(rng.integers(n, 3 * n))
        cases.append((n, [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(m)]))
    ntasks = 0
    for ci, (n, edges) in enumerate(cases):
        # unsymmetric plans also with policy bit 2: level 0 finished by the producer (even cases) or by the PRE tables (odd)
        # every third case with policy bit 3: one top launch per (level, class)
        plan = _solve_with_plan(jg, n, edges, rng, symmetric, top=top_level << 8 | soft << 16 | (8 if ci % 3 == 0 else 0), prefactor=None if symmetric else ci % 2 == 0)
        ntasks += plan.top_tables()[0].shape[0]
    assert ntasks > 5              # (a prefactor plan has one level less: fewer pivots above a given top level)


@pytest.mark.parametrize("top_level,soft", [(1, 4), (1, 63), (2, 8), (3, 16), (6, 24)])
def test_jordan_rows_on_small_and_random_graphs(jg, top_level, soft):
    """Plans with policy bit 49 (jg_symbolic.hpp: Jordan rows): the top tasks eliminate above the diagonal as well, the pivots of a
    task become ONE backward level of wave records over the task's external columns, chains only exist below the top.  Replayed both
    ways (plain sweep and Jordan sweep of the same plan) against a dense solve; a symmetric plan must refuse the bit."""
    rng = np.random.default_rng(31 * top_level + soft)
    cases = [(2, [(0, 1)]), (12, [(i, i + 1) for i in range(11)]), (9, [(0, i) for i in range(1, 9)]),
             (8, [(i, j) for i in range(8) for j in range(i + 1, 8)]),
             (24, [(i, j) for i in range(24) for j in range(i + 1, 24)]),
             (20, [(i, i + 1) for i in range(0, 19)] + [(i, i + 10) for i in range(10)]),
             (40, [(i, j) for i in range(40) for j in range(i + 1, min(i + 4, 40))])]
    for _ in range(8):
        n = int(rng.integers(20, 100))
        m = int(rng.integers(n, 3 * n))
        cases.append((n, [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(m)]))
    granted = 0
    for ci, (n, edges) in enumerate(cases):
        policy = top_level << 8 | soft << 16 | (8 if ci % 3 == 0 else 0) | 1 << 49
        plan = _solve_with_plan(jg, n, edges, rng, False, top=policy, prefactor=ci % 2 == 0)
        hdr, _, _, task_of, info = plan.top_tables()
        assert int(info[6]) == (1 if hdr.shape[0] else 0)
        if not info[6]:
            continue
        granted += 1
        # every pivot of a task sits in ONE backward level of the Jordan tables, and no chain task holds a task pivot
        seg, rec = plan.replay_tables("bwdj")
        chain = plan.get("bwd_chain")
        level_of_row = {}
        for base, nchunks, wpi, rpw, level, *_ in seg:
            if wpi <= 0:
                for t in range(nchunks):
                    nb, nE, off, _ = (int(v) for v in rec[base + t][:4])
                    assert all(task_of[int(k)] < 0 for k in chain[off: off + 3 * nb: 3])
                continue
            for r in rec[base: base + nchunks * 16 * rpw]:
                if r[0] >= 0:
                    level_of_row[int(r[0])] = int(level)
        for t in range(hdr.shape[0]):
            assert len({level_of_row[int(k)] for k in np.flatnonzero(task_of == t)}) == 1
    assert granted > 8
    # symmetric plans (LDL' through transposed reads of the upper entries) carry Jordan rows as well; a plan with grouped tasks does not
    for ci, (n, edges) in enumerate(cases):
        sym = _solve_with_plan(jg, n, edges, rng, True, top=top_level << 8 | soft << 16 | (8 if ci % 3 == 0 else 0) | 1 << 49)
        assert int(sym.top_tables()[4][6]) == (1 if sym.top_tables()[0].shape[0] else 0)
    grp = _solve_with_plan(jg, 40, [(i, j) for i in range(40) for j in range(i + 1, min(i + 4, 40))], rng, False, top=2 << 8 | 16 << 16 | 1 << 49 | 1 << 32 | 1 << 40)
    assert grp.top_tables()[0].shape[0] > 0 and int(grp.top_tables()[4][6]) == 0 and int(grp.top_tables()[4][7]) == 0


@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("mid,mmin,strict,top_level,soft", [(1, 1, 0, 2, 47), (2, 3, 0, 3, 47), (3, 6, 1, 6, 40), (2, 6, 0, 255, 63), (1, 2, 1, 1, 24)])
def test_grouped_tasks_on_small_and_random_graphs(jg, symmetric, mid, mmin, strict, top_level, soft):
    """Plans with a "mid" policy (jg_symbolic.hpp): pivots with `mid` neighbours at elimination and their ancestors go to tasks, a
    task's front is capped by its geometry (15 / 31 rows at 16 / 4 scenarios per workgroup, else one scenario per workgroup), update
    blocks are interleaved for the parent's workgroup, and the grouped tasks of a level form one launch with a workgroup map.  The
    replay checks every map and solves a system through the tables."""
    rng = np.random.default_rng(7 * mid + mmin)
    cases = [(2, [(0, 1)]), (12, [(i, i + 1) for i in range(11)]), (9, [(0, i) for i in range(1, 9)]),
             (24, [(i, j) for i in range(24) for j in range(i + 1, 24)]),                 # dense: a chain of 24 pivots, fronts beyond 16
             (40, [(i, j) for i in range(40) for j in range(i + 1, 40)]),                 # ... and beyond 32: all three geometries in one chain
             (60, [(i, j) for i in range(60) for j in range(i + 1, min(i + 6, 60))])]
    for _ in range(6):
        n = int(rng.integers(30, 120))
        m = int(rng.integers(n, 4 * n))
        cases.append((n, [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(m)]))
    geoms = set()
    for ci, (n, edges) in enumerate(cases):
        policy = top_level << 8 | soft << 16 | (8 if ci % 2 else 0) | mid << 32 | mmin << 40 | strict << 48
        plan = _solve_with_plan(jg, n, edges, rng, symmetric, top=policy, prefactor=None if symmetric else ci % 2 == 0)
        hdr, _, launches = plan.top_tables()[:3]
        geoms |= set(int(v) for v in hdr[:, 13])
        for tb, ntk, cls, lvl, grouped, wgb, nwg, _ in launches:
            assert all((int(hdr[t, 13]) > 0) == bool(grouped) for t in range(tb, tb + ntk))
    assert geoms >= {0, 2, 4}


@pytest.mark.parametrize("symmetric", [False, True])
def test_ordering_and_tables_on_degenerate_graphs(jg, symmetric):
    """One bus, isolated buses, islands, a path, a star, a complete graph, a ladder, random sparse graphs: the greedy
    minimum-fill / height ordering and every replay table must stay valid (checked by replaying a solve)."""
    rng = np.random.default_rng(11)
    cases = [
        (1, []),
        (3, []),                                                             # three isolated buses
        (6, [(0, 1), (1, 2), (3, 4)]),                                       # two islands + one isolated bus
        (12, [(i, i + 1) for i in range(11)]),                               # path
        (9, [(0, i) for i in range(1, 9)]),                                  # star: the hub must go last
        (8, [(i, j) for i in range(8) for j in range(i + 1, 8)]),            # complete graph
        (20, [(i, i + 1) for i in range(0, 19)] + [(i, i + 10) for i in range(10)]),
    ]
    for _ in range(4):
        n = int(rng.integers(15, 60))
        m = int(rng.integers(n, 3 * n))
        cases.append((n, [tuple(sorted(rng.choice(n, 2, replace=False))) for _ in range(m)]))
    for n, edges in cases:
        plan = _solve_with_plan(jg, n, edges, rng, symmetric)
        if n == 9 and len(edges) == 8:
            assert plan.get("perm")[-1] == 0 and plan.get("e_row").size == 9 + 16      # no fill on the star
        if n == 12 and len(edges) == 11:                                                 # path: the height penalty trades a little fill
            assert plan.get("e_row").size <= 12 + 22 + 12                                # for a shallower tree (dissection, not a chain)
            assert plan.replay_tables("fact")[0][-1, 4] < 12


def test_task_tables_do_not_depend_on_the_threads_that_built_them(jg, monkeypatch):
    """The factorisation tasks of a plan are laid out level by level on several host threads (round 4: 48 ms on one for the 512-scenario
    plan of the 10 000-bus grid); who built which level must not show in the tables."""
    s = jg.powerSystem("case_ACTIVSg10k")
    jg.acModel_(s)
    Y = s.model.ac.nodalMatrix
    policy = 1 | 4 | (47 << 16 | 127 << 24 | 12 << 4) | 1 << 49 | 1 << 50
    tables = []
    for thr in ("1", "3", "8"):
        monkeypatch.setenv("JG_PLAN_THREADS", thr)
        plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=policy)
        seg, rec = plan.replay_tables("fact")
        tables.append((seg.copy(), rec.copy(), plan.get(74).copy()))
    for seg, rec, meta in tables[1:]:
        assert np.array_equal(seg, tables[0][0]) and np.array_equal(rec, tables[0][1]) and np.array_equal(meta, tables[0][2])
    assert len(tables[0][1]) > 10000
