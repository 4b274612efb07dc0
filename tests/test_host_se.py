"""Host-side WLS bookkeeping (juliagrid.jl_amd.measurement / stateestimation value rules) against the
oracle's acWLS restatement -- CPU only, no device call."""
import numpy as np
import pytest

from conftest import load_case


class _FakePF:
    def __init__(self, vm, va):
        from types import SimpleNamespace as NS
        self.voltage = NS(magnitude=vm, angle=va)


def _solved(jg, oracle, name):
    t = load_case(name)
    osys = oracle.OracleSystem(t)
    pf = oracle.OracleNR(osys)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    osys.type = pf.type.copy(); osys.slack = pf.slack
    s = jg.powerSystem(t)
    jg.acModel_(s)
    jg.initializeACPowerFlow(s)
    return t, s, osys, vm, va


def test_exact_quantities_match_oracle(jg, oracle):
    for name in ("case14test", "case1354pegase"):
        t, s, osys, vm, va = _solved(jg, oracle, name)
        q = jg.exactQuantities(s, vm, va)
        br, bus = oracle.exact_quantities(osys, vm, va)
        assert np.abs(q.injectionActive - bus[:, 0]).max() < 1e-11 and np.abs(q.injectionReactive - bus[:, 1]).max() < 1e-11
        for k, a in enumerate((q.fromActive, q.fromReactive, q.toActive, q.toReactive, q.fromMagnitude, q.fromAngle,
                               q.toMagnitude, q.toAngle)):
            ok = np.ones(a.size, dtype=bool)
            if k in (5, 7):                      # the angle of a (numerically) zero current is arbitrary
                ok = br[:, k - 1] > 1e-6
            assert np.abs(a - br[:, k])[ok].max() < 1e-9


@pytest.mark.parametrize("pmu_kw", [dict(), dict(polar=True), dict(correlated=True), dict(polar=True, square=True)])
def test_wls_model_values_match_oracle(jg, oracle, pmu_kw):
    t, s, osys, vm, va = _solved(jg, oracle, "case14test")
    pf = _FakePF(vm, va)
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf)
    jg.addAmmeter_(mon, pf)
    jg.addAmmeter_(mon, pf, square=True, statusTo=-1)
    jg.addWattmeter_(mon, pf)
    jg.addVarmeter_(mon, pf, varianceFrom=1e-2)
    jg.addPmu_(mon, pf, **pmu_kw)
    mon.wattmeter.active.status[3] = 0
    mon.pmu.magnitude.status[5] = 0
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "voltmeter")
    oracle.add_from_power_flow(tab, osys, vm, va, "ammeter")
    oracle.add_from_power_flow(tab, osys, vm, va, "ammeter", to=False, square=True)
    oracle.add_from_power_flow(tab, osys, vm, va, "wattmeter")
    oracle.add_from_power_flow(tab, osys, vm, va, "varmeter")
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", **pmu_kw)
    rows = tab.rows
    # same edits on the oracle table: varmeter from-variances, one wattmeter and one PMU magnitude masked
    for d, r in enumerate(rows):
        if r[0] == 4 and r[1] == 1:
            rows[d] = r[:4] + (1e-2,) + r[5:]
    w = [d for d, r in enumerate(rows) if r[0] == 3][3]
    rows[w] = rows[w][:5] + (0,) + rows[w][6:]
    pm = [d for d, r in enumerate(rows) if r[0] == 5][5]
    rows[pm] = rows[pm][:5] + (0,) + rows[pm][6:]
    gn = oracle.OracleGN(osys, tab)
    se = jg.stateestimation
    code, index, rng, corr, devs, dev_row = se._wls_layout(mon)
    mean, wdiag, woff, status = se._wls_values(mon, devs, dev_row, code.size, *se._readings(mon))
    assert np.array_equal((status * code).astype(np.int8), gn.type)
    assert np.array_equal(index, gn.index)
    assert np.array_equal(rng, gn.range)
    assert np.abs(mean - gn.mean).max() < 1e-13
    assert np.allclose(wdiag, gn.wdiag, rtol=1e-13, atol=0)
    mask = np.zeros(gn.woff.size, dtype=bool)
    mask[corr - 1] = True
    assert np.all(gn.woff[~mask] == 0.0)          # (a pair whose angle is exactly 0 has a zero off-diagonal too)
    if corr.size:
        assert np.allclose(woff, gn.woff[corr - 1], rtol=1e-12, atol=0)


def test_ems_loads_the_reference_example_files(jg):
    """ems("case14.h5", "monitoring.h5") on copies of the reference's own example data (src/data/)."""
    import os
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "golden", "h5")
    system, mon = jg.ems(os.path.join(d, "case14.h5"), os.path.join(d, "monitoring.h5"))
    assert (mon.voltmeter.number, mon.ammeter.number, mon.wattmeter.number, mon.varmeter.number, mon.pmu.number) == (14, 40, 54, 54, 54)
    assert mon.voltmeter.layout.index == list(range(1, 15))
    assert all(mon.ammeter.layout.square) and not any(mon.pmu.layout.polar)
    assert mon.ammeter.layout.from_[:4] == [True, False, True, False] and mon.ammeter.layout.to[:4] == [False, True, False, True]
    assert set(mon.wattmeter.active.variance) == {1e-4} and set(mon.pmu.angle.variance) == {1e-8}
    assert set(mon.varmeter.reactive.status) == {1}
    assert abs(mon.voltmeter.magnitude.mean[0] - 1.06018914) < 1e-8 and abs(mon.wattmeter.active.mean[0] - 2.31931565) < 1e-8
    assert sum(mon.wattmeter.layout.bus) == 14 and sum(mon.wattmeter.layout.from_) == 20 and sum(mon.wattmeter.layout.to) == 20
