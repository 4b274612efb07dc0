"""Bad-data oracle (oracle.normalized_residuals / chi_threshold) against the reference's known answers
(test/stateEstimation/badData.jl:1-82, 129-167): chi-square threshold 109.7 and objective 3227.3, largest normalised
residual 52.5 at "Varmeter 4"; with polar bus PMUs added: 7713.26 at "PMU 10", then 78.3 at "Varmeter 4";
PMU-only model: 2606.8 at "PMU 2"; after removal the estimate returns to the power flow (atol 1e-10)."""
import numpy as np

from conftest import load_case


def bad_case14(oracle):
    """IEEE 14 as modified in badData.jl:6-12: bus 1 -> PV, bus 3 -> slack with angle -0.17."""
    t = load_case("case14test")
    lab = {int(l): i for i, l in enumerate(t["bus_label"])}
    t["bus_type"] = t["bus_type"].copy(); t["bus_va"] = t["bus_va"].copy()
    t["bus_type"][lab[1]] = 2
    t["bus_type"][lab[3]] = 3
    t["bus_va"][lab[3]] = -0.17
    s = oracle.OracleSystem(t)
    pf = oracle.OracleNR(s)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    s.type = pf.type.copy(); s.slack = pf.slack
    return t, s, vm, va


def legacy_table(oracle, s, vm, va):
    """@voltmeter/@wattmeter/@varmeter(variance = 1e-2) + add*!(monitoring, pf) (badData.jl:14-21)."""
    tab = oracle.MeterTable()
    for fam in ("voltmeter", "wattmeter", "varmeter"):
        oracle.add_from_power_flow(tab, s, vm, va, fam, variance=1e-2)
    return tab


def set_reading(tab, kind, number, **kw):
    """update<Device>!(monitoring; label = "<Device> number", ...): number counts the devices of that family from 1."""
    pos = [i for i, r in enumerate(tab.rows) if r[0] == oracle_kind(kind)][number - 1]
    r = list(tab.rows[pos])
    for k, v in kw.items():
        r[{"mean1": 3, "status1": 5, "mean2": 6, "status2": 8}[k]] = v
    tab.rows[pos] = tuple(r)
    return pos


def oracle_kind(name):
    return dict(voltmeter=1, ammeter=2, wattmeter=3, varmeter=4, pmu=5)[name]


def row_of_device(tab, pos):
    """first measurement row (0-based) of the device at table position pos (PMUs own two rows)."""
    rows = sorted(range(len(tab.rows)), key=lambda i: tab.rows[i][0])
    r = 0
    for i in rows:
        if i == pos:
            return r
        r += 2 if tab.rows[i][0] == 5 else 1
    raise KeyError(pos)


def estimate(oracle, s, tab):
    gn = oracle.OracleGN(s, tab)
    assert gn.state_estimation(40, 1e-8) == 0
    gn.increment()                                              # se.residual / se.jacobian at the estimate
    return gn


def test_one_outlier_known_answers(oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    pos = set_reading(tab, "varmeter", 4, mean1=10.25)
    gn = estimate(oracle, s, tab)
    assert abs(gn.objective - 3227.3) < 1e-1
    assert abs(oracle.chi_threshold(gn.m - 2 * s.n + 1) - 109.7) < 1e-1
    nr = oracle.gn_normalized_residuals(gn)
    assert int(np.argmax(nr)) == row_of_device(tab, pos) and abs(nr.max() - 52.5) < 1e-1
    set_reading(tab, "varmeter", 4, status1=0)                   # what residualTest! does on detection
    gn = estimate(oracle, s, tab)
    v = gn.vectors()
    assert np.abs(v["magnitude"] - vm).max() < 1e-10 and np.abs(v["angle"] - va).max() < 1e-10


def test_two_outliers_known_answers(oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = legacy_table(oracle, s, vm, va)
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", frm=False, to=False, variance=1e-5, polar=True)
    pv = set_reading(tab, "varmeter", 4, mean1=10.25)
    pp = set_reading(tab, "pmu", 10, mean1=30.0)
    gn = estimate(oracle, s, tab)
    nr = oracle.gn_normalized_residuals(gn)
    assert int(np.argmax(nr)) == row_of_device(tab, pp) and abs(nr.max() - 7713.26) < 1e-1
    set_reading(tab, "pmu", 10, status1=0)                       # polar PMU: the magnitude channel only (:268-273)
    gn = estimate(oracle, s, tab)
    nr = oracle.gn_normalized_residuals(gn)
    assert int(np.argmax(nr)) == row_of_device(tab, pv) and abs(nr.max() - 78.3) < 1e-1
    set_reading(tab, "varmeter", 4, status1=0)
    gn = estimate(oracle, s, tab)
    v = gn.vectors()
    assert np.abs(v["magnitude"] - vm).max() < 1e-10 and np.abs(v["angle"] - va).max() < 1e-10


def test_pmu_model_known_answers(oracle):
    t, s, vm, va = bad_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", variance=1e-5)
    pos = set_reading(tab, "pmu", 2, mean1=15.0)
    p = oracle.OraclePmuWLS(s, tab)
    p.solve()
    nr = oracle.pmu_normalized_residuals(p)
    assert int(np.argmax(nr)) // 2 == pos and abs(nr.max() - 2606.8) < 1e-1
    x = np.concatenate([p.magnitude * np.cos(p.angle), p.magnitude * np.sin(p.angle)])
    r = p.mean - p.coefficient @ x
    assert float(r @ (p.precision @ r)) >= oracle.chi_threshold(p.m - 2 * s.n)     # chi.detect (:158-159)
    set_reading(tab, "pmu", 2, status1=0, status2=0)
    p = oracle.OraclePmuWLS(s, tab)
    m, a = p.solve()
    assert np.abs(m - vm).max() < 1e-10 and np.abs(a - va).max() < 1e-10
    # two outliers (:182-191): PMU 20 with angle 10 pi and magnitude 30
    set_reading(tab, "pmu", 2, status1=1, status2=1)
    set_reading(tab, "pmu", 20, mean1=30.0, mean2=10 * np.pi)
    p = oracle.OraclePmuWLS(s, tab)
    p.solve()
    nr = oracle.pmu_normalized_residuals(p)
    assert int(np.argmax(nr)) // 2 == 19 and abs(nr.max() - 8853.2) < 1e-1
