"""Bad-data processing on the GPU (jg_gn_residual_test: selected inverse of the gain + normalised residuals) against the
oracle and the reference's known answers (test/stateEstimation/badData.jl).  Tolerances (f64):
  * normalised residuals vs the dense oracle ............ 1e-7 relative to the largest one (c = h G^-1 h' through two
    different inverses of a gain matrix with cond ~1e7)
  * known answers (chi threshold, objective, largest normalised residual) ... the reference's atol 1e-1
  * estimate after removal == power flow ................. atol 1e-10 (the reference's rule)
"""
import numpy as np
import pytest

from test_oracle_baddata import bad_case14, estimate, legacy_table, row_of_device, set_reading
from test_se_gpu import _mirror, _system_like

pytestmark = pytest.mark.gpu


def _check_vector(jg, an, ref):
    got = jg.normalizedResidual(an)
    assert np.abs(got - ref).max() <= 1e-7 * ref.max()


def test_one_outlier(jg, oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    pos = set_reading(tab, "varmeter", 4, mean1=10.25)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, s), tab))
    jg.stateEstimation_(an)
    chi = jg.chiTest(an)
    assert chi.detect and abs(chi.threshold - 109.7) < 1e-1 and abs(chi.objective - 3227.3) < 1e-1
    out = jg.residualTest_(an, threshold=3.0)
    assert out.detect and out.label == "Varmeter 4" and out.index == row_of_device(tab, pos) + 1
    assert abs(out.maxNormalizedResidual - 52.5) < 1e-1
    _check_vector(jg, an, oracle.gn_normalized_residuals(estimate(oracle, s, tab)))
    assert an.monitoring.varmeter.reactive.status[3] == 0 and an.method.type[out.index - 1] == 0 and an.method.iteration == 0
    jg.stateEstimation_(an)
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    out = jg.residualTest_(an, threshold=3.0)                    # clean now
    assert not out.detect and out.maxNormalizedResidual < 1e-3
    assert not jg.chiTest(an).detect
    an.close()


def test_two_outliers_with_polar_pmus(jg, oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", frm=False, to=False, variance=1e-5, polar=True)
    set_reading(tab, "varmeter", 4, mean1=10.25)
    set_reading(tab, "pmu", 10, mean1=30.0)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, s), tab))
    jg.stateEstimation_(an)
    out = jg.residualTest_(an)
    assert out.label == "PMU 10" and abs(out.maxNormalizedResidual - 7713.26) < 1e-1
    assert an.monitoring.pmu.magnitude.status[9] == 0 and an.monitoring.pmu.angle.status[9] == 1
    jg.stateEstimation_(an)
    out = jg.residualTest_(an)
    assert out.label == "Varmeter 4" and abs(out.maxNormalizedResidual - 78.3) < 1e-1
    jg.stateEstimation_(an)
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    an.close()


def test_rectangular_pmu_outlier_removes_both_rows(jg, oracle):
    """badData.jl:108-126 in spirit: legacy set + rectangular from-end PMUs, one with a wrong magnitude."""
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", bus=False, to=False, variance=1e-5)
    pos = set_reading(tab, "pmu", 7, mean1=30.0)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, s), tab))
    jg.stateEstimation_(an)
    out = jg.residualTest_(an)
    ref = oracle.gn_normalized_residuals(estimate(oracle, s, tab))
    _check_vector(jg, an, ref)
    r0 = row_of_device(tab, pos)
    assert out.label == "PMU 7" and out.index - 1 in (r0, r0 + 1) and abs(out.maxNormalizedResidual - ref.max()) <= 1e-7 * ref.max()
    assert list(an.method.type[r0:r0 + 2]) == [0, 0]
    assert an.monitoring.pmu.magnitude.status[6] == 0 and an.monitoring.pmu.angle.status[6] == 0
    jg.stateEstimation_(an, iteration=200, tolerance=1e-12)     # test/utility/utility.jl:282-286 rule for exact sets
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    an.close()


def test_pmu_model_outliers(jg, oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", variance=1e-5)
    set_reading(tab, "pmu", 2, mean1=15.0)
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, s), tab))
    jg.stateEstimation_(an)
    assert jg.chiTest(an).detect
    out = jg.residualTest_(an)
    assert out.label == "PMU 2" and abs(out.maxNormalizedResidual - 2606.8) < 1e-1
    p = oracle.OraclePmuWLS(s, tab)
    p.solve()
    _check_vector(jg, an, oracle.pmu_normalized_residuals(p))
    jg.stateEstimation_(an)
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-10 and np.abs(an.voltage.angle - va).max() < 1e-10
    assert not jg.chiTest(an).detect
    an.close()
    set_reading(tab, "pmu", 20, mean1=30.0, mean2=10 * np.pi)     # badData.jl:182-191: PMU 2 is back in service, still reading 15
    an = jg.pmuStateEstimation(_mirror(jg, _system_like(jg, t, s), tab))
    jg.stateEstimation_(an)
    out = jg.residualTest_(an)
    assert out.label == "PMU 20" and abs(out.maxNormalizedResidual - 8853.2) < 1e-1
    an.close()


def test_batch_every_scenario_finds_its_own_outlier(jg, oracle):
    """70 scenarios (two wavefront groups, the second partly filled): scenario b carries one gross error in wattmeter
    b % 30 (scenario 0 none).  Each scenario must flag its own device and return to the power flow after removal."""
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    B = 70
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, s), tab), batch=B)
    z1, v1, s1, z2, v2, s2 = an._z
    n1 = np.tile(z1, (B, 1))
    first_watt = s.n                                             # devices: 14 voltmeters, then the wattmeters
    planted = np.zeros(B, dtype=int)
    for b in range(1, B):
        planted[b] = first_watt + b % 30
        n1[b, planted[b]] += 5.0
    mean, wd, wo, _ = an._values(an.monitoring, an._devs, an._dev_row, an.dims["m"], n1, v1, s1, np.tile(z2, (B, 1)), v2, s2)
    an._upload_measurement(mean, wd, wo)
    jg.stateEstimation_(an)
    out = jg.residualTest_(an)
    assert not out.detect[0] and out.detect[1:].all()
    assert [int(i) - 1 for i in out.index[1:]] == [int(an._dev_row[p]) for p in planted[1:]]
    assert out.label[5] == f"Wattmeter {5 % 30 + 1}"
    # spot-check the whole vector of two scenarios against the oracle
    for b in (1, 69):
        tb = oracle.MeterTable()
        tb.rows = list(tab.rows)
        pos = [i for i, r in enumerate(tb.rows) if r[0] == 3][planted[b] - first_watt]
        tb.rows[pos] = tb.rows[pos][:3] + (float(n1[b, planted[b]]),) + tb.rows[pos][4:]
        ref = oracle.gn_normalized_residuals(estimate(oracle, s, tb))
        assert np.abs(jg.normalizedResidual(an)[b] - ref).max() <= 1e-7 * ref.max()
    jg.stateEstimation_(an, iteration=200, tolerance=1e-12)
    assert np.abs(an.voltage.magnitude - vm[None, :]).max() < 1e-10 and np.abs(an.voltage.angle - va[None, :]).max() < 1e-10
    chi = jg.chiTest(an)
    assert not chi.detect.any()
    an.close()


def test_config4_scale_planted_errors_are_found(jg):
    """BASELINE config 4 measurement set on the 9241-bus grid (~0.97e5 rows, ~1.5e5 factor blocks), 66 scenarios: scenario
    b >= 1 carries one gross wattmeter error at a different place of the grid.  Size-independent properties: every
    scenario flags exactly its planted row, the clean scenario flags nothing, and after removal the estimates return
    to the power-flow state."""
    s = jg.powerSystem("case9241synth")
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, tolerance=1e-11)
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf)
    jg.addWattmeter_(mon, pf)
    jg.addVarmeter_(mon, pf)
    jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
    B = 66
    an = jg.gaussNewton(mon, batch=B)
    z1, v1, s1, z2, v2, s2 = an._z
    n1 = np.tile(z1, (B, 1))
    nv, nw = mon.voltmeter.number, mon.wattmeter.number
    planted = np.zeros(B, dtype=int)
    for b in range(1, B):
        planted[b] = nv + (b * 641) % nw
        n1[b, planted[b]] += 2.0                                  # 200 MW on a 100 MVA base, sigma = 0.1
    mean, wd, wo, _ = an._values(mon, an._devs, an._dev_row, an.dims["m"], n1, v1, s1, np.tile(z2, (B, 1)), v2, s2)
    an._upload_measurement(mean, wd, wo)
    jg.stateEstimation_(an)
    assert np.all(an.status == 0)
    out = jg.residualTest_(an)
    assert not out.detect[0] and out.maxNormalizedResidual[0] < 0.5      # rounding residue over sqrt(~0) of near-critical rows
    assert out.detect[1:].all() and out.maxNormalizedResidual[1:].min() > 10.0
    assert [int(i) - 1 for i in out.index[1:]] == [int(an._dev_row[p]) for p in planted[1:]]
    jg.stateEstimation_(an, iteration=60, tolerance=1e-11)
    assert np.all(an.status == 0)
    assert np.abs(an.voltage.magnitude - pf.voltage.magnitude[None, :]).max() < 1e-8
    assert np.abs(an.voltage.angle - pf.voltage.angle[None, :]).max() < 1e-8
    assert not jg.residualTest_(an).detect.any()
    an.close()
