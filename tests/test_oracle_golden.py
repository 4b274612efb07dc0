"""Pins the CPU oracle against the reference's own golden vectors:
test/data/results.h5 (MATPOWER) as used by test/powerFlow/analysis.jl:1-68 through
testVoltage (test/utility/utility.jl:34-40): equal iteration count, V and theta `isapprox`."""
import numpy as np
import pytest

from conftest import load_case, load_golden


@pytest.mark.parametrize("name,iters", [("case14test", 7), ("case30test", 4)])
def test_oracle_matches_matpower_goldens(oracle, name, iters):
    g = load_golden(name)
    a = oracle.OracleNR(oracle.OracleSystem(load_case(name)))
    status = a.power_flow(iteration=20, tolerance=1e-8)
    vm, va = a.voltage()
    assert status == 0
    assert a.iteration == iters == int(g["newtonRaphson_iteration"][0])
    # Julia isapprox: norm(x - y) <= sqrt(eps) * max(norm(x), norm(y)); we are far inside it
    for got, key in ((vm, "newtonRaphson_voltageMagnitude"), (va, "newtonRaphson_voltageAngle")):
        ref = g[key]
        assert np.linalg.norm(got - ref) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(got), np.linalg.norm(ref))
        assert np.abs(got - ref).max() < 1e-13


def test_oracle_slack_relocation_case30(oracle):
    """test/powerFlow/analysis.jl:59-67: slack moved to bus 3 (type swap) converges to the same solution."""
    t = load_case("case30test")
    g = load_golden("case30test")
    typ = t["bus_type"].copy()
    lab = {int(l): i for i, l in enumerate(t["bus_label"])}
    typ[lab[1]] = 2
    typ[lab[3]] = 3
    t["bus_type"] = typ
    a = oracle.OracleNR(oracle.OracleSystem(t))
    assert a.power_flow() == 0
    vm, va = a.voltage()
    # angles are referenced to the new slack: compare angle differences
    ref_m, ref_a = g["newtonRaphson_voltageMagnitude"], g["newtonRaphson_voltageAngle"]
    assert np.abs(vm - ref_m).max() < 1e-8
    assert np.abs((va - va[0]) - (ref_a - ref_a[0])).max() < 1e-8


@pytest.mark.parametrize("name,iters", [("case14", 2), ("case118", 3), ("case1354pegase", 4), ("case_ACTIVSg10k", 4)])
def test_oracle_converges_shipped_cases(oracle, name, iters):
    a = oracle.OracleNR(oracle.OracleSystem(load_case(name)))
    assert a.power_flow() == 0
    assert a.iteration == iters
    assert a.history[-1].max() < 1e-8


def test_oracle_jacobian_matches_finite_differences(oracle):
    s = oracle.OracleSystem(load_case("case14test"))
    a = oracle.OracleNR(s)
    a.mismatch()
    a.solve()                       # fills the Jacobian at the start point, then steps
    # rebuild at a known point and compare J*dx with the mismatch difference
    b = oracle.OracleNR(s)
    vm, va = b.vm.copy(), b.va.copy()
    b.mismatch()
    _, f0, _ = b.vectors()
    b.solve()
    J, _, _ = b.vectors()
    import scipy.sparse as sp
    Jm = sp.csc_matrix((J, b.jrowval - 1, b.jcolptr - 1), shape=(b.dim, b.dim))
    rng = np.random.default_rng(0)
    dx = 1e-6 * rng.standard_normal(b.dim)
    vm2, va2 = vm.copy(), va.copy()
    for i in range(s.n):
        if b.pvpq[i]:
            va2[i] += dx[b.pvpq[i] - 1]
        if b.pq[i]:
            vm2[i] += dx[b.pq[i] - 1]
    c = oracle.OracleNR(s)
    c.set_voltage(vm2, va2)
    c.mismatch()
    _, f1, _ = c.vectors()
    assert np.abs((f1 - f0) - Jm @ dx).max() < 1e-9


@pytest.mark.parametrize("name", ["case14test", "case30test"])
def test_power_restatement_hits_the_matpower_goldens(oracle, name):
    """testPower (test/utility/utility.jl:41-59) on the oracle's restatement of power! (acAnalysis.jl:30-169): the
    MATPOWER goldens of test/data/results.h5 pin injections, supply, shunt, branch flows, charging, series losses and
    the generator allocation; atol 1e-8 as in test/powerFlow/analysis.jl."""
    g = load_golden(name)
    o = oracle.OracleNR(oracle.OracleSystem(load_case(name)))
    assert o.power_flow() == 0
    vm, va = o.voltage()
    r = oracle.power_and_current(o.sys, vm, va)
    pairs = [("injection", "injectionActive", "injectionReactive"), ("supply", "supplyActive", "supplyReactive"),
             ("shunt", "shuntActive", "shuntReactive"), ("from_", "fromActive", "fromReactive"), ("to", "toActive", "toReactive"),
             ("series", "lossActive", "lossReactive"), ("generator", "generatorActive", "generatorReactive")]
    for fam, ka, kr in pairs:
        assert np.abs(r[fam][0] - g["newtonRaphson_" + ka]).max() <= 1e-8, (fam, "active")
        assert np.abs(r[fam][1] - g["newtonRaphson_" + kr]).max() <= 1e-8, (fam, "reactive")
    assert np.abs(r["charging"][1] - (g["newtonRaphson_chargingFrom"] + g["newtonRaphson_chargingTo"])).max() <= 1e-8


@pytest.mark.parametrize("name", ["case14test", "case30test"])
def test_reactive_limits_hit_the_matpower_goldens(oracle, name):
    """test/powerFlow/limits.jl:4-42 on the oracle: solve, reactiveLimit!, solve again from the container's start,
    adjustAngle! to the original slack; iteration counts add up; V, theta to 1e-8 of MATPOWER's enforce-Q-limits run."""
    g = load_golden(name)
    t = load_case(name)
    osys = oracle.OracleSystem(t)
    slack0 = osys.slack
    o = oracle.OracleNR(osys)
    assert o.power_flow() == 0
    it0 = o.iteration
    vm, va = o.voltage()
    violate = oracle.reactive_limit(osys, o.type, vm, va)
    assert np.any(violate != 0)
    o2 = oracle.OracleNR(osys)
    assert o2.power_flow() == 0
    vm2, va2 = o2.voltage()
    va2 = va2 + (t["bus_va"][slack0 - 1] - va2[slack0 - 1])          # adjustAngle!(analysis; slack = original slack)
    assert it0 + o2.iteration == int(g["reactiveLimit_newtonRaphson_iteration"][0])
    assert np.abs(vm2 - g["reactiveLimit_newtonRaphson_voltageMagnitude"]).max() <= 1e-8
    assert np.abs(va2 - g["reactiveLimit_newtonRaphson_voltageAngle"]).max() <= 1e-8


@pytest.mark.parametrize("name", ["case14test", "case30test"])
@pytest.mark.parametrize("bx", [True, False])
def test_fast_newton_raphson_hits_the_matpower_goldens(oracle, name, bx):
    """test/powerFlow/analysis.jl:70-142 on the oracle's restatement of fastNewtonRaphsonBX / XB: iteration counts
    equal the reference's, V and theta within isapprox's default (sqrt(eps) relative)."""
    g = load_golden(name)
    key = "fastNewtonRaphsonBX" if bx else "fastNewtonRaphsonXB"
    o = oracle.OracleFastNR(oracle.OracleSystem(load_case(name)), bx)
    assert o.power_flow(iteration=30) == 0
    assert o.iteration == int(g[key + "_iteration"][0])
    for got, ref in ((o.vm, g[key + "_voltageMagnitude"]), (o.va, g[key + "_voltageAngle"])):
        assert np.linalg.norm(got - ref) <= 1.5e-8 * max(np.linalg.norm(got), np.linalg.norm(ref))
