"""Parity of the HIP linear PMU state estimation (pmuStateEstimation + solve!, through the C ABI) with the CPU oracle
and the reference's acceptance rule.  Tolerances (f64):
  * coefficient pattern (colptr, rowval) and se.type ................... bit-exact
  * coefficient values, mean, precision ................................ 1e-12 relative
  * estimate vs oracle (different LU, different summation order) ....... 1e-9 absolute on |V| (pu) and angle (rad)
  * exact PMUs => power-flow state ...................................... atol 1e-10 (test/utility/utility.jl:293-297);
    1e-7 on the 1354 / 9241-bus grids (gain cond ~1e9, see the test)
"""
import numpy as np
import pytest

from conftest import load_case
from test_oracle_pmu import case30, pmu_table
from test_oracle_se import se_case14
from test_se_gpu import _mirror, _system_like

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("correlated", [False, True])
@pytest.mark.parametrize("which", ["case14", "case30"])
def test_model_and_estimate_match_the_oracle(jg, oracle, which, correlated):
    t, osys, vm, va = se_case14(oracle) if which == "case14" else case30(oracle)
    var = 1e-4 if (which == "case30" and correlated) else None               # analysis.jl:435-436
    tab = pmu_table(oracle, osys, vm, va, variance_branch=var, correlated=correlated)
    o = oracle.OraclePmuWLS(osys, tab)
    om, oa = o.solve()
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab))
    H = an.coefficient
    Ho = o.coefficient
    Ho.sort_indices()
    assert np.array_equal(H.colptr - 1, Ho.indptr) and np.array_equal(H.rowval - 1, Ho.indices)
    assert np.abs(H.nzval - Ho.data).max() <= 1e-12 * np.abs(Ho.data).max()
    assert np.abs(an.method.mean - o.mean).max() <= 1e-12
    W = o.precision.toarray()
    assert np.abs(an.precision - W).max() <= 1e-12 * np.abs(W).max()
    jg.solveSE_(an)
    assert np.abs(an.voltage.magnitude - om).max() <= 1e-9 and np.abs(an.voltage.angle - oa).max() <= 1e-9
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    an.close()


def test_noisy_batch_matches_the_oracle_per_scenario(jg, oracle):
    """64 + 6 noise realisations in one batch (second wavefront group partially filled): every scenario has its own
    mean AND precision (variancePmu depends on the readings); spot-check scenarios against the oracle."""
    t, osys, vm, va = case30(oracle)
    tab = pmu_table(oracle, osys, vm, va, variance=1e-6, correlated=True)
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab), batch=70)
    jg.setNoise_(an, np.random.default_rng(7))
    jg.stateEstimation_(an)
    z1, v1, s1, z2, v2, s2 = an._z
    rng = np.random.default_rng(7)
    n1 = z1[None, :] + np.sqrt(v1)[None, :] * rng.standard_normal((70, z1.size))
    n2 = z2[None, :] + np.sqrt(v2)[None, :] * rng.standard_normal((70, z2.size))
    rows = sorted(tab.rows, key=lambda r: r[0])
    for b in (0, 33, 63, 64, 69):
        tb = oracle.MeterTable()
        tb.rows = [r[:3] + (float(n1[b, d]),) + r[4:6] + (float(n2[b, d]),) + r[7:] for d, r in enumerate(rows)]
        om, oa = oracle.OraclePmuWLS(osys, tb).solve()
        assert np.abs(an.voltage.magnitude[b] - om).max() <= 1e-9 and np.abs(an.voltage.angle[b] - oa).max() <= 1e-9
    assert np.abs(an.voltage.magnitude - vm).max() < 0.05        # the estimates scatter around the truth
    an.close()


def test_out_of_service_pmus_and_bus_only_placement(jg, oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = pmu_table(oracle, osys, vm, va)
    rows = list(tab.rows)
    for k in (3, 20, 21):
        rows[k] = rows[k][:5] + (0,) + rows[k][6:]
    tab.rows = rows
    o = oracle.OraclePmuWLS(osys, tab)
    om, oa = o.solve()
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab))
    assert list(an.method.type[6:8]) == [0, 0]
    jg.solveSE_(an)
    assert np.abs(an.voltage.magnitude - om).max() <= 1e-9 and np.abs(an.voltage.angle - oa).max() <= 1e-9
    an.close()
    # PMUs on the buses only: H = identity rows, the estimate is the reading itself
    tab2 = oracle.MeterTable()
    oracle.add_from_power_flow(tab2, osys, vm, va, "pmu", frm=False, to=False)
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab2))
    jg.solveSE_(an)
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-12 and np.abs(an.voltage.angle - va).max() < 1e-12
    an.close()


def test_unobservable_set_raises(jg, oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    tab.add("pmu", 0, 1, vm[0], 1e-8, 1, va[0], 1e-8, 1)
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab))
    with pytest.raises(jg._lib.JGridError):
        jg.solveSE_(an)
    an.close()


@pytest.mark.parametrize("name", ["case1354pegase", "case9241synth"])
def test_large_grid_exact_pmus_return_the_power_flow(jg, oracle, name):
    """Size-independent property at BASELINE scale: exact PMUs (every bus, both ends of every live branch) => the
    power-flow state, for a whole batch; case1354pegase is also checked against the oracle estimate."""
    t = load_case(name)
    osys = oracle.OracleSystem(t)
    pf = oracle.OracleNR(osys)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    # line admittances up to 5e3 pu and weights spread over 1e4..1e5 put the gain matrix at cond ~1e9: scipy's pivoting LU
    # (the oracle) itself returns the power-flow state to 2e-10 (1354) / 1e-8 (9241) only -- tolerance 1e-7 at this scale
    tab = pmu_table(oracle, osys, vm, va, variance=1e-6, min_current=5e-2)
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, osys), tab), batch=65)
    jg.stateEstimation_(an)
    assert np.abs(an.voltage.magnitude - vm[None, :]).max() < 1e-7 and np.abs(an.voltage.angle - va[None, :]).max() < 1e-7
    assert np.array_equal(an.voltage.magnitude[0], an.voltage.magnitude[64])            # same readings => same bits in every lane
    if name == "case1354pegase":
        om, oa = oracle.OraclePmuWLS(osys, tab).solve()
        assert np.abs(an.voltage.magnitude[64] - om).max() <= 1e-7 and np.abs(an.voltage.angle[64] - oa).max() <= 1e-7
    an.close()
