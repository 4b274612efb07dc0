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
