"""N3: fast Newton-Raphson BX / XB with both constant matrices factorised once on the device.
Pinned by the reference's own goldens (test/powerFlow/analysis.jl:70-142: iteration counts and V, theta of
results.h5) and compared with the oracle restatement (models element-wise 1e-12, mismatches 1e-12, increments 1e-9)."""
import numpy as np
import pytest

from conftest import load_case, load_golden

pytestmark = pytest.mark.gpu


def _make(jg, name, bx, **kw):
    f = jg.fastNewtonRaphsonBX if bx else jg.fastNewtonRaphsonXB
    return f(jg.powerSystem(load_case(name)), **kw)


@pytest.mark.parametrize("name", ["case14test", "case30test"])
@pytest.mark.parametrize("bx", [True, False])
def test_matpower_goldens(jg, name, bx):
    g = load_golden(name)
    key = "fastNewtonRaphsonBX" if bx else "fastNewtonRaphsonXB"
    an = _make(jg, name, bx)
    jg.powerFlow_(an, iteration=30)
    assert an.status == 0
    assert an.method.iteration == int(g[key + "_iteration"][0])
    for got, ref in ((an.voltage.magnitude, g[key + "_voltageMagnitude"]), (an.voltage.angle, g[key + "_voltageAngle"])):
        assert np.linalg.norm(got - ref) <= 1.5e-8 * max(np.linalg.norm(got), np.linalg.norm(ref))      # isapprox default


@pytest.mark.parametrize("name", ["case14", "case118", "case300", "case1354pegase"])
@pytest.mark.parametrize("bx", [True, False])
def test_model_mismatch_and_step_match_oracle(jg, oracle, name, bx):
    t = load_case(name)
    an = _make(jg, name, bx)
    o = oracle.OracleFastNR(oracle.OracleSystem(t), bx)
    assert np.array_equal(an.method.pq, o.pq) and np.array_equal(an.method.pvpq, o.pvpq)
    for M, R in ((an.method.active.jacobian, o.P), (an.method.reactive.jacobian, o.Q)):          # reference CSC layout
        D = M.toscipy() - R
        assert abs(D).max() <= 1e-12 * max(1.0, abs(R).max())
    dp, dq = jg.mismatch_(an)
    op, oq = o.mismatch()
    scale = max(1.0, op, oq)
    assert abs(dp - op) <= 1e-12 * scale and abs(dq - oq) <= 1e-12 * scale
    f = an.mismatch
    assert np.abs(f[:o.mismP.size] - o.mismP).max() <= 1e-12 * scale
    assert np.abs(f[o.mismP.size:] - o.mismQ).max() <= 1e-12 * scale
    jg.solve_(an)
    o.solve()
    assert np.abs(an.voltage.magnitude - o.vm).max() <= 1e-9 and np.abs(an.voltage.angle - o.va).max() <= 1e-9


@pytest.mark.parametrize("name", ["case118", "case1354pegase", "case1951rte"])
def test_power_flow_matches_oracle_and_newton(jg, oracle, name):
    t = load_case(name)
    an = _make(jg, name, True)
    jg.powerFlow_(an, iteration=100)
    o = oracle.OracleFastNR(oracle.OracleSystem(t), True)
    assert o.power_flow(iteration=100) == 0 and an.status == 0
    assert an.method.iteration == o.iteration
    assert np.abs(an.voltage.magnitude - o.vm).max() <= 1e-8 and np.abs(an.voltage.angle - o.va).max() <= 1e-8
    nr = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(nr)
    assert np.abs(an.voltage.magnitude - nr.voltage.magnitude).max() <= 1e-6          # same fixed point, looser stop


def test_batched_injections(jg, oracle):
    """Monte-Carlo load variations share the two factorised matrices."""
    t = load_case("case300")
    s = jg.powerSystem(t)
    B = 4
    an = jg.fastNewtonRaphsonXB(s, batch=B)
    scale = 1.0 + 0.01 * np.random.default_rng(5).standard_normal((B, 1))
    pd, qd = s.bus.demand.active[None, :] * scale, s.bus.demand.reactive[None, :] * scale
    jg.setInjection_(an, s.bus.supply.active[None, :] - pd, s.bus.supply.reactive[None, :] - qd)
    jg.powerFlow_(an, iteration=100)
    for b in range(B):
        o = oracle.OracleFastNR(oracle.OracleSystem(t), False)
        o.sys.pd, o.sys.qd = pd[b].copy(), qd[b].copy()
        assert o.power_flow(iteration=100) == 0 and an.status[b] == 0
        assert an.method.iteration[b] == o.iteration
        assert np.abs(an.voltage.magnitude[b] - o.vm).max() <= 1e-8


def test_large_grid_follows_the_oracle_step_by_step(jg, oracle):
    """10k-bus grid: three fast iterations (two triangular sweeps each, no refactorisation) land on the oracle's state;
    the iteration limit is reported like the reference does (status 1, iteration == limit)."""
    t = load_case("case_ACTIVSg10k")
    an = _make(jg, "case_ACTIVSg10k", False)
    jg.powerFlow_(an, iteration=3)
    o = oracle.OracleFastNR(oracle.OracleSystem(t), False)
    assert o.power_flow(iteration=3) == 1 and an.status == 1
    assert an.method.iteration == 3 == o.iteration
    assert np.abs(an.voltage.magnitude - o.vm).max() <= 1e-9 and np.abs(an.voltage.angle - o.va).max() <= 1e-9


@pytest.mark.parametrize("name,bx,count", [("case118", True, 9), ("case118", False, 9), ("case1354pegase", True, 70), ("case1354pegase", False, 6)])
def test_batched_outages_match_the_oracle_per_outage(jg, oracle, name, bx, count):
    """ Fast Newton-Raphson under BATCHED outages: scenario s = branch labels[s] out of service.  The reference edits the entries of B'
    and B'' the branch touches and refactorises (_updateBranch!, src/powerSystem/branch.jl:477; acPowerFlow.jl:476-537); here every scenario keeps the
    shared matrices plus its (at most) 4 + 4 edits and the batch is factorised ONCE.  Against the oracle's fast Newton-Raphson on the grid with that branch
    switched off, per outage: equal iteration counts, V / theta 1e-8; the base-case lane is bitwise the plain analysis."""
    t = load_case(name)
    s = jg.powerSystem(t)
    labels = [int(x) for x in jg.outageList(s, count - 1, seed=21)] + [0]
    an = jg.contingencyAnalysis(s, labels, method="bx" if bx else "xb")
    jg.powerFlow_(an, iteration=100)
    checked = 0
    for b in sorted(set(list(range(0, count, max(1, count // 6))) + [count - 1])):
        t2 = {k: np.array(v) for k, v in t.items()}
        if labels[b]:
            t2["br_status"][labels[b] - 1] = 0
        o = oracle.OracleFastNR(oracle.OracleSystem(t2), bx)
        st = o.power_flow(iteration=100)
        assert an.status[b] == st == 0, (b, labels[b], an.status[b], st)
        assert an.method.iteration[b] == o.iteration, (b, labels[b])
        assert np.abs(an.voltage.magnitude[b] - o.vm).max() <= 1e-8 and np.abs(an.voltage.angle[b] - o.va).max() <= 1e-8
        checked += 1
    assert checked >= 5
    plain = (jg.fastNewtonRaphsonBX if bx else jg.fastNewtonRaphsonXB)(jg.powerSystem(t), batch=count, max_patch=4)
    jg.powerFlow_(plain, iteration=100)
    assert np.array_equal(plain.voltage.magnitude[-1], an.voltage.magnitude[-1]) and plain.method.iteration[-1] == an.method.iteration[-1]
    # restoring a scenario (label None) brings its matrices back: the lane is bitwise the base case again
    jg.setOutage_(an, 0, None)
    jg.setInitialPoint_(an)
    jg.powerFlow_(an, iteration=100)
    assert np.array_equal(an.voltage.magnitude[0], an.voltage.magnitude[-1]) and np.array_equal(an.voltage.angle[0], an.voltage.angle[-1])
    an.close(); plain.close()


def test_batched_outages_on_the_10k_bus_grid(jg, oracle):
    """512 outages of case_ACTIVSg10k in one fast Newton-Raphson batch: ONE factorisation of the 512 edited matrix pairs, then sweeps only.  Fast decoupled
    iterations converge slowly on this grid, so -- like test_large_grid_follows_the_oracle_step_by_step -- four iterations are compared state for state with the
    oracle's on the grid with that branch switched off (status 1 = iteration limit, as the reference reports it)."""
    t = load_case("case_ACTIVSg10k")
    s = jg.powerSystem(t)
    labels = [int(x) for x in jg.outageList(s, 512, seed=512)]
    an = jg.contingencyAnalysis(s, labels, method="xb")
    jg.powerFlow_(an, iteration=4)
    assert np.all(an.status == 1) and np.all(an.method.iteration == 4)
    for b in (0, 77, 300, 511):
        t2 = {k: np.array(v) for k, v in t.items()}
        t2["br_status"][labels[b] - 1] = 0
        o = oracle.OracleFastNR(oracle.OracleSystem(t2), False)
        assert o.power_flow(iteration=4) == 1 and o.iteration == 4
        assert np.abs(an.voltage.magnitude[b] - o.vm).max() <= 1e-9 and np.abs(an.voltage.angle[b] - o.va).max() <= 1e-9, b
    # the lanes differ (their matrices do): an outage moves the state of its neighbourhood
    assert np.abs(an.voltage.angle[0] - an.voltage.angle[1]).max() > 1e-6
    an.close()


def test_islanding_outages_in_a_fast_batch_end_like_the_reference(jg, oracle):
    """A bridge outage cuts an island off the slack bus.  The island's block of B' is a Laplacian only when no branch inside it shifts the phase
    (fastNewtonJacobian! puts -B cos(shift) -+ A sin(shift) off the diagonal and B on it, acPowerFlow.jl:416-447, 476-481): then it is singular -- the
    reference's lu! raises, the device's pivot guard marks the scenario, status 3 before the first iteration -- otherwise it is merely ill conditioned
    and the reference iterates to the limit, as the device does (status 1).  Which of the two is decided by the oracle's B' (smallest singular value)."""
    t = load_case("case1354pegase")
    s = jg.powerSystem(t)
    from test_guard_gpu import _island_bridges
    bad = _island_bridges(jg, s, 3)[:3]
    good = [int(x) for x in jg.outageList(s, 2, seed=1)]
    labels = good[:1] + bad + good[1:]
    an = jg.contingencyAnalysis(s, labels, method="xb")
    jg.powerFlow_(an, iteration=30)
    assert an.status[0] == 0 and an.status[-1] == 0
    seen = set()
    for b, lab in enumerate(labels):
        if lab not in bad:
            continue
        t2 = {k: np.array(v) for k, v in t.items()}
        t2["br_status"][lab - 1] = 0
        try:
            o = oracle.OracleFastNR(oracle.OracleSystem(t2), False)
            sv = np.linalg.svd(o.P.toarray(), compute_uv=False)
            singular = sv[-1] < 1e-11 * sv[0]
        except RuntimeError:                                      # splu refused the matrix outright
            singular = True
        if singular:
            assert an.status[b] == 3 and an.method.iteration[b] == 0, (lab, an.status[b])
        else:
            assert o.power_flow(iteration=30) == 1 and an.status[b] == 1 and an.method.iteration[b] == 30 == o.iteration, (lab, an.status[b])
        seen.add(bool(singular))
    assert seen == {True, False}, "the three bridges of the fixture cover both endings"
    an.close()


@pytest.mark.parametrize("method", ["bx", "xb", "nr"])
def test_the_outage_of_a_branch_that_is_already_out_changes_nothing(jg, method):
    """A scenario whose 'outage' names a branch the grid already has out of service: neither Ybus nor B' / B'' hold anything of that branch
    (acPowerFlow.jl:416-447 skips it), so the lane is bitwise the base case."""
    t = {k: np.array(v) for k, v in load_case("case118").items()}
    off = int(jg.outageList(jg.powerSystem(t), 1, seed=9)[0])
    t["br_status"][off - 1] = 0
    s = jg.powerSystem(t)
    an = jg.contingencyAnalysis(s, [off, 0, off], method=method)
    jg.powerFlow_(an, iteration=100)
    assert np.all(np.asarray(an.status) == 0)
    for b in (0, 2):
        assert np.array_equal(an.voltage.magnitude[b], an.voltage.magnitude[1]) and np.array_equal(an.voltage.angle[b], an.voltage.angle[1])
        assert an.method.iteration[b] == an.method.iteration[1]
    an.close()
