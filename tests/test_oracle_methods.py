"""Pins the oracle's restatement of the reference's two non-normal Gauss-Newton solvers -- increment! for
GaussNewton{Orthogonal} (acStateEstimation.jl:906-932) and GaussNewton{PetersWilkinson} (:934-971) -- on the reference's own
acceptance rule for them (test/stateEstimation/analysis.jl:219-232, 284-297): with noise-free measurements of all
families the estimate equals the power-flow state (IEEE 14: atol 1e-10; IEEE 30 with tolerance 1e-10: atol 1e-8), and
on the fact that all three increments solve the same least-squares problem."""
import numpy as np
import pytest

from conftest import load_case
from test_oracle_se import se_case14


def all_families(oracle, s, vm, va):
    tab = oracle.MeterTable()
    for fam in ("voltmeter", "ammeter", "wattmeter", "varmeter", "pmu"):
        oracle.add_from_power_flow(tab, s, vm, va, fam)
    return tab


def case30(oracle):
    t = load_case("case30test")
    s = oracle.OracleSystem(t)
    pf = oracle.OracleNR(s)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    s.type = pf.type.copy(); s.slack = pf.slack
    return t, s, vm, va


@pytest.mark.parametrize("method", ["increment_orthogonal", "increment_peters_wilkinson"])
def test_ieee14_known_answer(oracle, method):
    t, s, vm, va = se_case14(oracle)
    gn = oracle.OracleGN(s, all_families(oracle, s, vm, va))
    ok, it = gn.state_estimation_with(getattr(gn, method))             # defaults: 40 iterations, 1e-8 (analysis.jl:221, 229)
    v = gn.vectors()
    assert ok and it <= 10
    assert np.abs(v["magnitude"] - vm).max() <= 1e-10
    assert np.abs(v["angle"] - va).max() <= 1e-10


@pytest.mark.parametrize("method", ["increment_orthogonal", "increment_peters_wilkinson"])
def test_ieee30_known_answer(oracle, method):
    t, s, vm, va = case30(oracle)
    gn = oracle.OracleGN(s, all_families(oracle, s, vm, va))
    ok, it = gn.state_estimation_with(getattr(gn, method), tolerance=1e-10)     # analysis.jl:286, 294
    v = gn.vectors()
    assert ok
    assert np.abs(v["magnitude"] - vm).max() <= 1e-8
    assert np.abs(v["angle"] - va).max() <= 1e-8


def test_three_increments_solve_the_same_least_squares_problem(oracle):
    t, s, vm, va = se_case14(oracle)
    gn = oracle.OracleGN(s, all_families(oracle, s, vm, va))
    gn.increment()
    normal = gn.vectors()["increment"].copy()
    orth = gn.increment_orthogonal()
    pw = gn.increment_peters_wilkinson()
    scale = np.abs(orth).max()
    assert np.abs(orth - pw).max() <= 1e-9 * scale
    assert np.abs(orth - normal).max() <= 1e-7 * scale
    assert orth[s.slack - 1] == 0.0 and pw[s.slack - 1] == 0.0
