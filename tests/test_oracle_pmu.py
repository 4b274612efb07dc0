"""The PMU-only linear WLS oracle (oracle.OraclePmuWLS) against the reference's own acceptance rule
(test/stateEstimation/analysis.jl:347-440 + testPmuEstimation, test/utility/utility.jl:293-297):
PMUs on every bus and both ends of every branch, read from a solved AC power flow, return that power
flow's voltages to atol 1e-10 -- uncorrelated and correlated, IEEE 14 (modified) and IEEE 30."""
import numpy as np
import pytest

from conftest import load_case
from test_oracle_se import se_case14


def case30(oracle):
    t = load_case("case30test")
    s = oracle.OracleSystem(t)
    pf = oracle.OracleNR(s)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    return t, s, vm, va


def pmu_table(oracle, osys, vm, va, variance=None, min_current=0.0, variance_branch=None, **flags):
    """addPmu!(monitoring, pf): all buses, then from / to end of every in-service branch; min_current drops branch PMUs
    on (numerically) dead branches, whose angle reading makes variancePmu degenerate (1/variance ~ 1e37)."""
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", variance=variance, **flags)
    if variance_branch is not None:                              # @pmu(varianceMagnitudeFrom = ..., varianceAngleFrom = ..., ...To ...)
        tab.rows = [r if r[1] == 0 else r[:4] + (variance_branch,) + r[5:7] + (variance_branch,) + r[8:] for r in tab.rows]
    if min_current > 0.0:
        tab.rows = [r for r in tab.rows if r[1] == 0 or r[3] > min_current]
    return tab


@pytest.mark.parametrize("correlated", [False, True])
def test_case14_exact_pmus_return_the_power_flow(oracle, correlated):
    t, osys, vm, va = se_case14(oracle)
    p = oracle.OraclePmuWLS(osys, pmu_table(oracle, osys, vm, va, correlated=correlated))
    m, a = p.solve()
    assert p.coefficient.shape == (2 * (osys.n + 2 * int((osys.status == 1).sum())), 2 * osys.n)
    assert np.abs(m - vm).max() < 1e-10 and np.abs(a - va).max() < 1e-10


@pytest.mark.parametrize("correlated", [False, True])
def test_case30_exact_pmus_return_the_power_flow(oracle, correlated):
    t, osys, vm, va = case30(oracle)
    # analysis.jl:435-436: the correlated IEEE 30 set is built with 1e-4 variances on the branch PMUs
    p = oracle.OraclePmuWLS(osys, pmu_table(oracle, osys, vm, va, variance_branch=1e-4 if correlated else None, correlated=correlated))
    m, a = p.solve()
    assert np.abs(m - vm).max() < 1e-10 and np.abs(a - va).max() < 1e-10


def test_precision_block_is_the_inverse_covariance(oracle):
    """analysis.jl:318-344: precision of a correlated PMU == inv(covariance) from the polar variances."""
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    zv, zt, vv, vt = 1.3, -0.2, 1e-1, 0.2
    tab.add("pmu", 2, 2, zv, vv, 1, zt, vt, 1, correlated=True)
    p = oracle.OraclePmuWLS(osys, tab)
    cov = np.array([[vv * np.cos(zt) ** 2 + vt * (zv * np.sin(zt)) ** 2, np.cos(zt) * np.sin(zt) * (vv - vt * zv ** 2)],
                    [0.0, vv * np.sin(zt) ** 2 + vt * (zv * np.cos(zt)) ** 2]])
    cov[1, 0] = cov[0, 1]
    assert np.allclose(np.linalg.inv(cov), p.precision.toarray(), rtol=1e-12)


def test_out_of_service_pmu_keeps_its_rows_empty(oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = pmu_table(oracle, osys, vm, va)
    rows = list(tab.rows)
    rows[3] = rows[3][:5] + (0,) + rows[3][6:]                # magnitude channel of the 4th PMU out of service
    tab.rows = rows
    p = oracle.OraclePmuWLS(osys, tab)
    assert p.coefficient[6:8].nnz == 0 and p.mean[6] == 0.0 and p.mean[7] == 0.0
    m, a = p.solve()
    assert np.abs(m - vm).max() < 1e-10 and np.abs(a - va).max() < 1e-10
