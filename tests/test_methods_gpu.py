"""gaussNewton(monitoring, Orthogonal) / gaussNewton(monitoring, PetersWilkinson) on the device (jg_gn_set_method: corrected
semi-normal equations on the engine's factor) against the oracle's dense restatement of the reference's two solvers
(acStateEstimation.jl:906-971) and the reference's acceptance rule for them (test/stateEstimation/analysis.jl:219-232, 284-297).
Tolerances: increment 1e-8 relative (another factorisation of the same least-squares problem), estimate 1e-10 (IEEE 14) /
1e-8 (IEEE 30, tolerance 1e-10), iteration counts equal."""
import numpy as np
import pytest

from conftest import load_case
from test_oracle_methods import all_families, case30
from test_oracle_se import se_case14
from test_se_gpu import _mirror, _system_like

pytestmark = pytest.mark.gpu

METHODS = [("Orthogonal", "increment_orthogonal"), ("PetersWilkinson", "increment_peters_wilkinson")]


@pytest.mark.parametrize("tag,restated", METHODS)
def test_first_increment_matches_the_restated_solver(jg, oracle, tag, restated):
    t, osys, vm, va = se_case14(oracle)
    tab = all_families(oracle, osys, vm, va)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab), getattr(jg, tag))
    gn = oracle.OracleGN(osys, tab)
    mx = jg.incrementSE_(an)
    ref = getattr(gn, restated)()
    assert np.abs(an.increment - ref).max() <= 1e-8 * np.abs(ref).max()
    assert abs(mx - np.abs(ref).max()) <= 1e-8 * np.abs(ref).max()
    assert an.increment[osys.slack - 1] == 0.0


@pytest.mark.parametrize("tag,restated", METHODS)
def test_ieee14_known_answer_and_iteration_count(jg, oracle, tag, restated):
    t, osys, vm, va = se_case14(oracle)
    tab = all_families(oracle, osys, vm, va)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab), getattr(jg, tag))
    jg.stateEstimation_(an)
    gn = oracle.OracleGN(osys, tab)
    ok, it = gn.state_estimation_with(getattr(gn, restated))
    assert an.status == 0 and ok
    assert an.method.iteration == it
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-10
    assert np.abs(an.voltage.angle - va).max() <= 1e-10


@pytest.mark.parametrize("tag,restated", METHODS)
def test_ieee30_known_answer(jg, oracle, tag, restated):
    t, osys, vm, va = case30(oracle)
    tab = all_families(oracle, osys, vm, va)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab), getattr(jg, tag))
    jg.stateEstimation_(an, tolerance=1e-10)
    assert an.status == 0
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-8
    assert np.abs(an.voltage.angle - va).max() <= 1e-8


def test_correction_pass_beats_the_plain_normal_equations_on_a_stiff_set(jg, oracle):
    """Weights spread over 12 decades (sigma^2 1e-12 PMUs next to 1 legacy meters): the increment of the Orthogonal tag stays at
    the QR solution to 1e-8 where the plain normal equations are allowed to drift."""
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "wattmeter", variance=1.0)
    oracle.add_from_power_flow(tab, osys, vm, va, "varmeter", variance=1.0)
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", bus=True, frm=False, to=False, variance=1e-12)
    s = _system_like(jg, t, osys)
    gn = oracle.OracleGN(osys, tab)
    ref = gn.increment_orthogonal()
    orth = jg.gaussNewton(_mirror(jg, s, tab), jg.Orthogonal)
    jg.incrementSE_(orth)
    plain = jg.gaussNewton(_mirror(jg, s, tab))
    jg.incrementSE_(plain)
    e_orth = np.abs(orth.increment - ref).max() / np.abs(ref).max()
    e_plain = np.abs(plain.increment - ref).max() / np.abs(ref).max()
    assert e_orth <= 1e-8
    assert e_orth <= e_plain + 1e-14


def test_batch_of_noisy_realisations_agrees_between_the_tags(jg, oracle):
    """Monte-Carlo batch: both tags minimise the same objective, so their estimates agree to the step tolerance."""
    t, osys, vm, va = se_case14(oracle)
    tab = all_families(oracle, osys, vm, va)
    s = _system_like(jg, t, osys)
    out = []
    for tag in (jg.LU, jg.Orthogonal):
        an = jg.gaussNewton(_mirror(jg, s, tab), tag, batch=5)
        jg.setNoise_(an, np.random.default_rng(7))
        jg.stateEstimation_(an, tolerance=1e-10)
        assert np.all(np.asarray(an.status) == 0)
        out.append((an.voltage.magnitude.copy(), an.voltage.angle.copy()))
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-8 and np.abs(out[0][1] - out[1][1]).max() <= 1e-8


def test_correlated_pmus_are_refused(jg, oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", correlated=True)
    with pytest.raises(Exception):
        jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab), jg.Orthogonal)


def test_a_tag_must_be_a_wls_method(jg, oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = all_families(oracle, osys, vm, va)
    with pytest.raises(TypeError):
        jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab), 3)


@pytest.mark.parametrize("tag", ["Orthogonal", "PetersWilkinson"])
@pytest.mark.parametrize("which", ["case14", "case30"])
def test_pmu_only_model_with_the_two_tags(jg, oracle, which, tag):
    """pmuStateEstimation(monitoring, Orthogonal / PetersWilkinson) (test/stateEstimation/analysis.jl:379-392, 416-428):
    exact uncorrelated PMUs => the power-flow state, and the same estimate as the oracle's linear WLS solve."""
    from test_oracle_pmu import pmu_table
    t, osys, vm, va = se_case14(oracle) if which == "case14" else case30(oracle)
    tab = pmu_table(oracle, osys, vm, va, variance_branch=None, correlated=False)
    om, oa = oracle.OraclePmuWLS(osys, tab).solve()
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab), getattr(jg, tag))
    jg.solveSE_(an)
    assert np.abs(an.voltage.magnitude - om).max() <= 1e-9 and np.abs(an.voltage.angle - oa).max() <= 1e-9
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    an.close()
