"""The first Newton iteration of a common-start batch on ONE shared factor (jgrid.h: jg_nr_base_*; csrc/jg_comp.hip) -- the compensation method in
place of the per-scenario lu! + ldiv! of the reference's N-1 loop (/root/reference/src/powerSystem/branch.jl:453-459, src/powerFlow/acPowerFlow.jl:890-897).
Checked against the oracle: the base quantities (J_0^-1 on the Ybus pattern, J_0^-1 f_0) against a dense inverse of the oracle's Jacobian, the
compensated step against the oracle's own first increment of every outage, whole power flows (iteration counts, V, theta), the islanding guard and
every condition under which jg_nr_run must fall back to the refactorising iteration by itself."""
import numpy as np
import pytest

from conftest import load_case
from plan_emulator import block_jacobian_from_csc

pytestmark = pytest.mark.gpu


def _oracle_state(oracle, osys, vm, va):
    """f_0, J (reference CSC values), first increment of the oracle at (vm, va)"""
    o = oracle.OracleNR(osys)
    o.set_voltage(vm, va)
    o.mismatch()
    _, f0, _ = o.vectors()
    f0 = f0.copy()
    o.solve()
    J, _, inc = o.vectors()
    return o, f0, J.copy(), inc.copy()


def _bus_pairs(o, n, vec):
    """reference ordering (pvpq then pq) -> [n][2] bus order, zeros where the bus has no such equation / unknown"""
    out = np.zeros((n, 2))
    for i in range(n):
        if o.pvpq[i]:
            out[i, 0] = vec[o.pvpq[i] - 1]
        if o.pq[i]:
            out[i, 1] = vec[o.pq[i] - 1]
    return out


@pytest.mark.parametrize("name,top_cap,converged", [("case14test", 0, False), ("case118", 8, False), ("case118", -1, True), ("case1354pegase", 64, True),
                                                    ("case1354pegase", 0, False)])
def test_base_quantities_match_a_dense_inverse_of_the_oracles_jacobian(jg, oracle, name, top_cap, converged):
    t = load_case(name)
    s = jg.powerSystem(t)
    single = jg.newtonRaphson(s)
    if converged:
        jg.powerFlow_(single)
    else:
        single._pull_voltage()
    vm, va = single.voltage.magnitude.copy(), single.voltage.angle.copy()
    base = jg.BaseCase(single, top_cap=top_cap)
    info = base.info
    n = s.bus.number
    assert info["top_pivots"] <= (max(top_cap, 0) if top_cap else 512)
    if top_cap < 0:
        assert info["top_pivots"] == 0 and info["forward_launches"] == info["forward_launches_no_top"]
    osys = oracle.OracleSystem(t)
    o, f0, J, inc = _oracle_state(oracle, osys, vm, va)
    rowptr, col, A = block_jacobian_from_csc(n, osys.colptr, osys.rowval, o.type, o.pq, o.pvpq, o.jcolptr, o.jrowval, J)
    D = np.zeros((2 * n, 2 * n))
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            D[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2] = A[p]
    Z = np.linalg.inv(D)
    scale = np.abs(Z).max()
    zc = base.get(0, col.size * 4).reshape(-1, 2, 2)
    worst = 0.0
    for i in range(n):
        for p in range(rowptr[i], rowptr[i + 1]):
            worst = max(worst, np.abs(zc[p] - Z[2 * i:2 * i + 2, 2 * col[p]:2 * col[p] + 2]).max())
    assert worst <= 1e-9 * scale, (worst, scale)
    f0b = base.get(2, 2 * n).reshape(n, 2)
    assert np.abs(f0b - _bus_pairs(o, n, f0)).max() <= 1e-11 * max(1.0, np.abs(s.bus.demand.active).max())
    y0 = base.get(1, 2 * n).reshape(n, 2)
    ref = _bus_pairs(o, n, inc)
    assert np.abs(y0 - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    print(f"[base {name} top_cap {top_cap}] {info}; J0^-1 on the pattern: max error {worst:.2e} (scale {scale:.2e})")
    base.close()
    single.close()


def _labels_with_bridges(jg, s, count, seed):
    ok = [int(x) for x in jg.outageList(s, count, seed=seed)]
    br = np.flatnonzero(jg.bridges(s) & (s.branch.layout.status == 1))
    return ok, [int(x) + 1 for x in br[:3]]


@pytest.mark.parametrize("name,batch,top_cap", [("case118", 70, 8), ("case1354pegase", 130, 0), ("case1354pegase", 64, -1)])
def test_compensated_first_iteration_equals_the_oracles_first_step_and_the_refactorising_path(jg, oracle, name, batch, top_cap):
    t = load_case(name)
    s = jg.powerSystem(t)
    single = jg.newtonRaphson(s)
    jg.powerFlow_(single)
    start = (single.voltage.magnitude.copy(), single.voltage.angle.copy())
    base = jg.BaseCase(single, top_cap=top_cap)
    ok, br = _labels_with_bridges(jg, s, batch - 4, seed=11)
    labels = ok[:batch - 4] + br[:2] + [0, ok[0]]                   # non-bridge outages, two islanding ones, the base case, a duplicate
    labels = labels + [0] * (batch - len(labels))
    an = jg.contingencyAnalysis(s, labels)
    base.attach(an)
    ref = jg.contingencyAnalysis(s, labels)
    jg.powerflow._push_voltage(ref, *start)
    # ---- ONE iteration: method.increment of both paths against the oracle's first increment of that outage
    jg.startFromBase_(an)
    jg.powerFlow_(an, iteration=1)
    jg.powerFlow_(ref, iteration=1)
    assert jg.firstIterationCounts(an) == (1, 0) and jg.firstIterationCounts(ref) == (0, 1)
    inc_c, inc_r = an.increment, ref.increment
    osys = oracle.OracleSystem(t)
    worst = 0.0
    nbr = len(br[:2])
    for sc in range(batch):
        o = oracle.OracleNR(osys)
        if labels[sc]:
            ptr, dy = jg.outagePatch(s, labels[sc])
            for p, d in zip(ptr, dy):
                o.add_ybus(p - 1, d)
        o.set_voltage(*start)
        o.mismatch()
        if labels[sc] in br:                                       # islanding: the 4 x 4 system of the correction is singular (the batched factorisation may need
            assert an.status[sc] == 3, (sc, labels[sc])            # another iteration before a pivot of ITS cancels below the guard)
            continue
        if not labels[sc]:
            continue                                               # the base case is converged at the start: no step is taken
        o.solve()
        _, _, inc = o.vectors()
        sc_scale = max(1e-3, np.abs(inc).max())
        worst = max(worst, np.abs(inc_c[sc] - inc).max() / sc_scale)
        assert np.abs(inc_c[sc] - inc).max() <= 1e-9 * sc_scale, (sc, labels[sc])
        assert np.abs(inc_r[sc] - inc).max() <= 1e-9 * sc_scale
    assert (an.status == 3).sum() == nbr
    print(f"[compensated step {name} x {batch}, top_cap {top_cap}] first increment against the oracle's, worst relative error {worst:.2e}")
    # ---- whole power flows: equal iteration counts and states, duplicates bitwise
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    jg.powerflow._push_voltage(ref, *start)
    jg.powerFlow_(ref)
    assert jg.firstIterationCounts(an) == (2, 0)
    isl = np.array([lab in br for lab in labels])
    assert (an.status[isl] == 3).all() and (ref.status[isl] != 0).all() and np.array_equal(an.status[~isl], ref.status[~isl])
    good = an.status == 0
    assert np.array_equal(an.method.iteration[good], ref.method.iteration[good])
    assert np.abs(an.voltage.magnitude[good] - ref.voltage.magnitude[good]).max() <= 1e-10
    assert np.abs(an.voltage.angle[good] - ref.voltage.angle[good]).max() <= 1e-10
    dup = batch - 1 if len(labels) == batch and labels[batch - 1] == ok[0] else labels.index(ok[0], 1)
    assert np.array_equal(an.voltage.magnitude[0], an.voltage.magnitude[dup]) and np.array_equal(an.voltage.angle[0], an.voltage.angle[dup])
    for sc in range(0, batch, max(1, batch // 24)):
        if not good[sc]:
            continue
        o = oracle.OracleNR(osys)
        if labels[sc]:
            ptr, dy = jg.outagePatch(s, labels[sc])
            for p, d in zip(ptr, dy):
                o.add_ybus(p - 1, d)
        o.set_voltage(*start)
        assert o.power_flow() == 0 and o.iteration == an.method.iteration[sc], (sc, labels[sc])
        vm, va = o.voltage()
        assert np.abs(an.voltage.magnitude[sc] - vm).max() <= 1e-8 and np.abs(an.voltage.angle[sc] - va).max() <= 1e-8
    an.close(); ref.close(); base.close(); single.close()


def test_the_run_falls_back_by_itself_when_the_conditions_do_not_hold(jg, oracle):
    t = load_case("case300")
    s = jg.powerSystem(t)
    single = jg.newtonRaphson(s)
    jg.powerFlow_(single)
    start = (single.voltage.magnitude.copy(), single.voltage.angle.copy())
    base = jg.BaseCase(single)
    labels = [int(x) for x in jg.outageList(s, 8, seed=2)]
    B = len(labels)
    an = jg.contingencyAnalysis(s, labels)
    jg.powerflow._push_voltage(an, *start)
    jg.powerFlow_(an)                                              # no base attached
    assert jg.firstIterationCounts(an) == (0, 1)
    it0, vm0, va0, good = an.method.iteration.copy(), an.voltage.magnitude.copy(), an.voltage.angle.copy(), an.status == 0
    assert good.sum() >= B - 2                                      # (an outage whose power flow diverges ends at the iteration limit on both paths, in another garbage state)
    base.attach(an)
    assert base.info["attached"] == 1
    jg.powerflow._push_voltage(an, *start)                         # attached, but the state was not taken from the base
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (0, 2)
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (1, 2)
    assert np.array_equal(an.method.iteration, it0) and np.array_equal(an.status == 0, good)
    assert np.abs(an.voltage.magnitude[good] - vm0[good]).max() <= 1e-10 and np.abs(an.voltage.angle[good] - va0[good]).max() <= 1e-10
    jg.powerFlow_(an)                                              # a second run without a new start: the state has moved
    assert jg.firstIterationCounts(an) == (1, 3)
    jg.setFirstIteration_(an, False)
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (1, 4)
    jg.setFirstIteration_(an, True)
    # scenarios with outages AND their own injections: the mismatch moves everywhere, the run refactorises
    rng = np.random.default_rng(5)
    scale = 1.0 + 0.01 * rng.standard_normal((B, 1))
    pd, qd = s.bus.demand.active[None, :] * scale, s.bus.demand.reactive[None, :] * scale
    jg.setInjection_(an, s.bus.supply.active[None, :] - pd, s.bus.supply.reactive[None, :] - qd)
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (1, 5)
    # ... without outages (Monte-Carlo injections on the base grid) the shared factor serves: J_s = J_0
    jg.setOutages_(an, [0] * B)
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (2, 5)
    osys = oracle.OracleSystem(t)
    for b in range(B):
        o = oracle.OracleNR(osys)
        o.set_power(osys.ps, osys.qs, pd[b], qd[b])
        o.set_voltage(*start)
        assert o.power_flow() == 0 and an.status[b] == 0 and an.method.iteration[b] == o.iteration
        vm, va = o.voltage()
        assert np.abs(an.voltage.magnitude[b] - vm).max() <= 1e-8 and np.abs(an.voltage.angle[b] - va).max() <= 1e-8
    an.close(); base.close(); single.close()


def test_set_ybus_detaches_the_base_and_destroy_order_is_free(jg):
    t = load_case("case118")
    s = jg.powerSystem(t)
    single = jg.newtonRaphson(s)
    jg.powerFlow_(single)
    base = jg.BaseCase(single)
    single.close()                                                 # the base keeps its own copies
    an = jg.contingencyAnalysis(s, [int(x) for x in jg.outageList(s, 5, seed=1)])
    base.attach(an)
    base.close()                                                   # the handle still holds a reference
    jg.startFromBase_(an)
    jg.powerFlow_(an)
    assert jg.firstIterationCounts(an) == (1, 0) and (an.status == 0).all()
    jg.powerflow._upload_ybus(an)                                  # jg_nr_set_ybus: the base was factorised for the old matrix
    with pytest.raises(jg._lib.JGridError):
        jg.startFromBase_(an)
    an.close()
