"""Numeric-failure paths of the batched Newton-Raphson solver (the device keeps a STATIC pivot order; the reference's
UMFPACK / KLU pivot and raise SingularException, /root/reference/src/backend/utility.jl:470-484, and the reference's loop
treats non-convergence as a status, acPowerFlow.jl:1414-1423):

  * a bridge outage islands part of the grid: the island's Jacobian is singular, its last pivot block cancels to rounding level
    (not to an exact zero) -- the scenario must come back with status 3, the other scenarios of the batch untouched;
  * a NaN in one scenario's state must end in status 3 for that scenario only (k_check);
  * close to the nose point (load scaled up until the oracle's pivoting LU needs most of its 20 iterations) the static-pivot
    factorisation must follow the oracle iteration for iteration.
"""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def _island_bridges(jg, s, min_size=3):
    """Bridge branches whose removal cuts off at least `min_size` buses without the slack, largest island first."""
    n = s.bus.number
    f, t = s.branch.layout.from_ - 1, s.branch.layout.to - 1
    on = np.flatnonzero(s.branch.layout.status == 1)
    out = []
    for k in np.flatnonzero(jg.bridges(s)):
        adj = [[] for _ in range(n)]
        for b in on:
            if b != k:
                adj[f[b]].append(t[b])
                adj[t[b]].append(f[b])
        seen = np.zeros(n, dtype=bool)
        stack = [int(s.bus.layout.slack) - 1]
        seen[stack[0]] = True
        while stack:
            v = stack.pop()
            for u in adj[v]:
                if not seen[u]:
                    seen[u] = True
                    stack.append(int(u))
        cut = int(n - seen.sum())
        if cut >= min_size:
            out.append((cut, int(k) + 1))
    return [lab for _, lab in sorted(out, reverse=True)]


@pytest.mark.parametrize("name", ["case118", "case1354pegase", "case_ACTIVSg10k"])
def test_bridge_outage_is_reported_not_solved(jg, oracle, name):
    t = load_case(name)
    s = jg.powerSystem(t)
    bad = _island_bridges(jg, s, 2 if name == "case118" else 3)[:2]
    if not bad:
        pytest.skip("grid has no bridge that cuts off three buses")
    good = [int(x) for x in jg.outageList(s, 3, seed=7)]
    labels = [bad[0], good[0], 0, good[1]] + bad[1:] + [good[2]]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    ref = jg.contingencyAnalysis(s, [lab for lab in labels if lab not in bad])
    jg.powerFlow_(ref, iteration=20, tolerance=1e-8)
    keep = [i for i, lab in enumerate(labels) if lab not in bad]
    for i, lab in enumerate(labels):
        if lab in bad:
            assert an.status[i] == 3, f"islanding outage of branch {lab} came back with status {an.status[i]}"
    # the healthy scenarios of the batch do not notice their neighbours: bitwise the batch without them
    assert np.array_equal(an.status[keep], ref.status) and np.array_equal(an.method.iteration[keep], ref.method.iteration)
    assert np.array_equal(an.voltage.magnitude[keep], ref.voltage.magnitude) and np.array_equal(an.voltage.angle[keep], ref.voltage.angle)
    osys = oracle.OracleSystem(t)
    o = oracle.OracleNR(osys)
    ptr, dy = jg.outagePatch(s, good[0])
    for p, d in zip(ptr, dy):
        o.add_ybus(p - 1, d)
    st = o.power_flow(iteration=20, tolerance=1e-8)
    assert an.status[1] == st
    if st == 0:
        vm, va = o.voltage()
        assert an.method.iteration[1] == o.iteration
        assert np.abs(an.voltage.magnitude[1] - vm).max() <= 1e-8 and np.abs(an.voltage.angle[1] - va).max() <= 1e-8


@pytest.mark.parametrize("name,batch", [("case118", 3), ("case1354pegase", 70)])
def test_nan_state_ends_in_status_3_for_that_scenario_only(jg, name, batch):
    s = jg.powerSystem(load_case(name))
    clean = jg.newtonRaphson(s, batch=batch)
    jg.powerFlow_(clean)
    an = jg.newtonRaphson(s, batch=batch)
    vm = np.tile(an.voltage.magnitude[0], (batch, 1))
    va = np.tile(an.voltage.angle[0], (batch, 1))
    hit = [1, batch - 1] if batch > 3 else [1]
    for b in hit:
        vm[b, 5 + b % 7] = np.nan
    jg.powerflow._push_voltage(an, vm, va)
    jg.powerFlow_(an)
    ok = [b for b in range(batch) if b not in hit]
    assert all(an.status[b] == 3 for b in hit)
    assert np.array_equal(an.status[ok], clean.status[ok]) and np.array_equal(an.method.iteration[ok], clean.method.iteration[ok])
    assert np.array_equal(an.voltage.magnitude[ok], clean.voltage.magnitude[ok])


@pytest.mark.parametrize("name", ["case14", "case30test", "case118", "case300", "case1354pegase"])
def test_static_pivots_follow_the_pivoting_oracle_towards_the_nose_point(jg, oracle, name):
    """Demand and generation scaled up in steps of 5 % until the oracle stops converging; the three heaviest loadings that
    still have a solution (the Jacobian is closest to singular there) in one batch: iteration counts equal, V / theta 1e-8."""
    t = load_case(name)
    s = jg.powerSystem(t)
    osys = oracle.OracleSystem(t)

    def oracle_run(scale):
        o = oracle.OracleNR(osys)
        o.set_power(osys.ps * scale, osys.qs, osys.pd * scale, osys.qd * scale)
        return o, o.power_flow(iteration=20, tolerance=1e-8)

    scale, solved = 1.0, []
    while scale < 6.0:
        o, st = oracle_run(scale)
        if st != 0:
            break
        solved.append((scale, o))
        scale = round(scale + 0.05, 2)
    assert len(solved) >= 3
    pick = solved[-3:]
    an = jg.newtonRaphson(s, batch=len(pick))
    sc = np.array([p[0] for p in pick])[:, None]
    jg.setInjection_(an, (s.bus.supply.active[None, :] - s.bus.demand.active[None, :]) * sc,
                     s.bus.supply.reactive[None, :] - s.bus.demand.reactive[None, :] * sc)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    for b, (scl, o) in enumerate(pick):
        assert an.status[b] == 0 and an.method.iteration[b] == o.iteration, f"scale {scl}: {an.method.iteration[b]} vs {o.iteration} iterations"
        vm, va = o.voltage()
        assert np.abs(an.voltage.magnitude[b] - vm).max() <= 1e-8 and np.abs(an.voltage.angle[b] - va).max() <= 1e-8
    assert max(o.iteration for _, o in pick) >= 5          # the heaviest loading needs more iterations than the base case


def test_fast_newton_raphson_survives_post_processing(jg):
    """fastNewtonRaphsonBX -> powerFlow! -> power! (uploads the branch tables) -> setInitialPoint! -> powerFlow!: the second
    solve repeats the first (the branch upload used to free the fast solver's buffers), and the full Jacobian of a fast
    analysis is refused instead of silently overwriting the factorised B', B''."""
    s = jg.powerSystem(load_case("case30test"))
    an = jg.fastNewtonRaphsonBX(s)
    jg.powerFlow_(an, iteration=100, tolerance=1e-8)
    it1, vm1, va1 = int(an.method.iteration), an.voltage.magnitude.copy(), an.voltage.angle.copy()
    jg.power_(an)
    with pytest.raises(RuntimeError):
        an.jacobian
    jg.setInitialPoint_(an)
    jg.powerFlow_(an, iteration=100, tolerance=1e-8)
    assert an.status == 0 and int(an.method.iteration) == it1
    assert np.array_equal(an.voltage.magnitude, vm1) and np.array_equal(an.voltage.angle, va1)
    an.close()


def test_mismatch_and_increment_belong_to_their_scenario_after_compaction(jg, oracle):
    """Batch of 200 outage scenarios of case1354pegase (lanes are compacted while the batch iterates): method.mismatch is the
    mismatch of every scenario's FINAL state, method.increment the last increment of THAT scenario (oracle, same outage)."""
    t = load_case("case1354pegase")
    s = jg.powerSystem(t)
    labels = [int(x) for x in jg.outageList(s, 200, seed=11)]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    assert len(set(an.method.iteration.tolist())) > 1            # scenarios finish in different iterations
    f = an.mismatch
    conv = an.status == 0
    assert np.abs(f[conv]).max() < 1e-8
    osys = oracle.OracleSystem(t)
    inc = an.increment
    for sc in (0, 63, 64, 65, 130, 199):
        o = oracle.OracleNR(osys)
        ptr, dy = jg.outagePatch(s, labels[sc])
        for p, d in zip(ptr, dy):
            o.add_ybus(p - 1, d)
        st = o.power_flow(iteration=20, tolerance=1e-8)
        assert st == an.status[sc]
        if st == 0:
            _, f_ref, inc_ref = o.vectors()
            assert an.method.iteration[sc] == o.iteration
            assert np.abs(inc[sc] - inc_ref).max() <= 1e-9 * max(1.0, np.abs(inc_ref).max()) + 1e-12
            assert np.abs(f[sc]).max() < 1e-8 and np.abs(f_ref).max() < 1e-8      # both at their converged state (they differ by J x 1e-10)


@pytest.mark.parametrize("name", ["case118", "case1354pegase", "case_ACTIVSg10k"])
def test_refined_newton_step(jg, oracle, name):
    """jg_nr_set_refine: one step of iterative refinement behind the static-pivot solve (the reference's UMFPACK solve refines).
    The refined first Newton increment is at least as close to the pivoting oracle's as the plain one and good to 1e-11; the
    refined iteration reaches the same solution in the same number of iterations, alone and in a compacted batch."""
    t = load_case(name)
    s = jg.powerSystem(t)
    o = oracle.OracleNR(oracle.OracleSystem(t))
    o.mismatch()
    o.solve()
    _, _, inc_ref = o.vectors()
    scale = max(1.0, np.abs(inc_ref).max())
    err = {}
    for refine in (False, True):
        an = jg.newtonRaphson(jg.powerSystem(t), refine=refine)
        jg.mismatch_(an)
        jg.solve_(an)
        err[refine] = np.abs(an.increment - inc_ref).max() / scale
        an.close()
    assert err[True] <= 1e-11 and err[True] <= 2.0 * err[False] + 1e-15
    plain = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(plain)
    labels = [int(x) for x in jg.outageList(s, 150, seed=2)]
    for batch_labels in (None, labels):
        a = jg.newtonRaphson(jg.powerSystem(t), refine=True) if batch_labels is None else jg.contingencyAnalysis(s, batch_labels)
        b = None
        if batch_labels is not None:
            jg.setRefinement_(a, True)
            b = jg.contingencyAnalysis(s, batch_labels)
            jg.powerFlow_(b)
        jg.powerFlow_(a)
        if batch_labels is None:
            assert a.status == 0 and a.method.iteration == plain.method.iteration
            assert np.abs(a.voltage.magnitude - plain.voltage.magnitude).max() < 1e-9
        else:
            assert np.array_equal(a.status, b.status) and np.array_equal(a.method.iteration, b.method.iteration)
            ok = a.status == 0
            assert np.abs(a.voltage.magnitude[ok] - b.voltage.magnitude[ok]).max() < 1e-8 and np.abs(a.voltage.angle[ok] - b.voltage.angle[ok]).max() < 1e-8


@pytest.mark.parametrize("copies,batch", [(3, 6), (4, 70)])
def test_island_whose_root_pivot_sits_in_the_top_tasks(jg, copies, batch):
    """the guard of k_fact_top took its reference scale AFTER the children's update matrices had come in, so a pivot that those
    cancel -- the last pivot of an island as large as a whole sub-grid, whose root sits in the multifrontal top -- was compared with itself.
    Instances of case1354pegase tied slack to slack by single lines: the outage of a tie cuts off one or more whole instances without a slack
    bus.  Those scenarios must come back with status 3, their neighbours in the batch bitwise untouched."""
    from juliagrid.jl_amd.synthetic import tiledGrid
    t = load_case("case1354pegase")
    one = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(one)
    jg.power_(one)
    slack = int(np.flatnonzero(np.asarray(one.system.bus.layout.type) == 3)[0])
    p_slack = float(np.asarray(one.power.supply.active).reshape(-1)[slack])
    one.close()
    s = jg.powerSystem(tiledGrid(t, copies, slack_active=p_slack))
    nb = s.branch.number
    ties = [nb - (copies - 1) + c + 1 for c in range(copies - 1)]          # the tie lines are the last branches (1-based labels)
    assert all(jg.bridges(s)[k - 1] for k in ties)
    good = [int(x) for x in jg.outageList(s, batch - len(ties), seed=5)]
    labels = good[:2] + [ties[0]] + good[2:-1] + ties[1:] + good[-1:]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    ref = jg.contingencyAnalysis(s, [lab for lab in labels if lab not in ties])
    jg.powerFlow_(ref, iteration=20, tolerance=1e-8)
    keep = [i for i, lab in enumerate(labels) if lab not in ties]
    for i, lab in enumerate(labels):
        if lab in ties:
            assert an.status[i] == 3, f"the outage of tie {lab} came back with status {an.status[i]} after {an.method.iteration[i]} iterations"
    assert (ref.status == 0).sum() >= len(keep) - 2
    assert np.array_equal(an.status[keep], ref.status) and np.array_equal(an.method.iteration[keep], ref.method.iteration)
    assert np.array_equal(an.voltage.magnitude[keep], ref.voltage.magnitude) and np.array_equal(an.voltage.angle[keep], ref.voltage.angle)
    an.close(); ref.close()
