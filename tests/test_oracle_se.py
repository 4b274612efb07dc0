"""Pins the Gauss-Newton oracle with the reference's own known-answer rule:
noise-free measurements generated from a converged power flow => the WLS estimate equals the
power-flow state to atol 1e-10 after stateEstimation!(gn; iteration = 200, tolerance = 1e-12)
(test/utility/utility.jl:282-286; one case per measurement family as test/stateEstimation/analysis.jl:27-210),
plus the analytic PMU covariance (:300-346) and squared-current variance (:173-200) checks."""
import numpy as np
import pytest

from conftest import load_case


def se_case14(oracle):
    """The modified IEEE 14 system of test/stateEstimation/analysis.jl:7-20."""
    t = load_case("case14test")
    lab = {int(l): i for i, l in enumerate(t["bus_label"])}
    t["bus_type"] = t["bus_type"].copy(); t["bus_vm"] = t["bus_vm"].copy(); t["bus_va"] = t["bus_va"].copy(); t["br_g"] = t["br_g"].copy()
    t["bus_type"][lab[1]] = 2
    t["bus_type"][lab[3]] = 3
    t["bus_va"][lab[3]] = -0.25
    t["bus_vm"][lab[1]] = 1.0
    t["bus_vm"][lab[3]] = 1.2
    t["bus_vm"][lab[4]] = 1.0
    t["bus_vm"][lab[5]] = 1.1
    t["br_g"][2] = 0.01
    t["br_g"][5] = 0.05
    s = oracle.OracleSystem(t)
    pf = oracle.OracleNR(s)
    assert pf.power_flow() == 0
    vm, va = pf.voltage()
    # the power flow mutates bus types/slack exactly like newtonRaphson(system) does
    s.type = pf.type.copy(); s.slack = pf.slack
    return t, s, vm, va


def base_pmu(oracle, s, vm, va):
    """addPmu!(monitoring, pf; statusFrom = -1, statusTo = -1, polar = true) with unit variances (:22-24)."""
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", bus=True, frm=False, to=False, variance=1.0, polar=True)
    return tab


def check(oracle, s, tab, vm, va):
    gn = oracle.OracleGN(s, tab)
    st = gn.state_estimation(iteration=200, tolerance=1e-12)
    v = gn.vectors()
    assert st == 0, gn.history
    assert np.abs(v["magnitude"] - vm).max() <= 1e-10
    assert np.abs(v["angle"] - va).max() <= 1e-10
    return gn


FAMILIES = [
    ("voltmeter", dict(variance=1e-4)),
    ("ammeter", dict(variance=1e-2)),
    ("ammeter", dict(variance=1e-4, square=True)),
    ("wattmeter", dict(variance=1e-4, frm=False, to=False)),
    ("wattmeter", dict(variance=1e-4, bus=False)),
    ("varmeter", dict(variance=1e-4, frm=False, to=False)),
    ("varmeter", dict(variance=1e-2, bus=False)),
]


@pytest.mark.parametrize("family,kw", FAMILIES)
def test_family_recovers_power_flow_state(oracle, family, kw):
    t, s, vm, va = se_case14(oracle)
    tab = base_pmu(oracle, s, vm, va)
    oracle.add_from_power_flow(tab, s, vm, va, family, **kw)
    check(oracle, s, tab, vm, va)


def _set_status(tab, mag=None, ang=None, mag_on=(), ang_on=(), mag_off=(), ang_off=()):
    """status edits on PMU devices (1-based device numbers, like monitoring.pmu.*.status[...] in the reference tests)."""
    rows = tab.rows
    for d in range(len(rows)):
        r = list(rows[d])
        if mag is not None and r[1] != 0:
            r[5] = mag
        if ang is not None and r[1] != 0:
            r[8] = ang
        rows[d] = tuple(r)
    for lst, pos, val in ((mag_on, 5, 1), (ang_on, 8, 1), (mag_off, 5, 0), (ang_off, 8, 0)):
        for d in lst:
            r = list(rows[d - 1]); r[pos] = val; rows[d - 1] = tuple(r)


PMU_CASES = [
    # (addPmu! keywords, status edits) -- test/stateEstimation/analysis.jl:84-171
    (dict(bus=True, frm=False, to=False, variance=1e-4, correlated=True), {}),
    (dict(bus=False, variance=1e-4), {}),
    (dict(bus=False, correlated=True), {}),
    (dict(bus=False, to=False, variance=1e-2, polar=True), dict(mag_off=(2, 14, 18), ang_off=(14, 18))),
    (dict(bus=False, to=False, variance=1e-2, polar=True, square=True), dict(mag_off=(14, 18), ang_off=(14, 18))),
    (dict(bus=False, frm=False, variance=1e-2, polar=True), dict(mag=0, ang=0, mag_on=(4, 8, 12, 16, 18), ang_on=(3, 5, 8, 13, 18))),
    (dict(bus=False, frm=False, variance=1e-2, polar=True, square=True), dict(mag=0, ang=0, mag_on=(4, 8, 12, 16, 18))),
]


@pytest.mark.parametrize("kw,edits", PMU_CASES)
def test_pmu_variants_recover_power_flow_state(oracle, kw, edits):
    t, s, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", **kw)
    _set_status(tab, **edits)
    if not kw.get("bus", True):                      # addPmuBus helper of the reference tests
        oracle.add_from_power_flow(tab, s, vm, va, "pmu", bus=True, frm=False, to=False, variance=1.0, polar=True)
    check(oracle, s, tab, vm, va)


def test_masked_measurements_keep_their_rows(oracle):
    """status 0 => type 0, mean 0, H row zero, row kept; estimate unaffected."""
    t, s, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", bus=False, to=False, variance=1e-2, polar=True)
    oracle.add_from_power_flow(tab, s, vm, va, "pmu", bus=True, frm=False, to=False, variance=1.0, polar=True)
    rows = tab.rows
    for d in (1, 13, 17):
        r = list(rows[d]); r[5] = 0; rows[d] = tuple(r)        # magnitude status
    for d in (13, 17):
        r = list(rows[d]); r[8] = 0; rows[d] = tuple(r)        # angle status
    gn = check(oracle, s, tab, vm, va)
    assert gn.type[2 * 13] == 0 and gn.type[2 * 13 + 1] == 0 and gn.mean[2 * 13] == 0.0
    assert gn.type[2] == 0 and gn.type[3] != 0


def test_all_measurements_case14_and_case30(oracle):
    for name in ("case14test", "case30test"):
        t = load_case(name)
        s = oracle.OracleSystem(t)
        pf = oracle.OracleNR(s)
        assert pf.power_flow() == 0
        vm, va = pf.voltage()
        s.type = pf.type.copy(); s.slack = pf.slack
        tab = oracle.MeterTable()
        for fam in ("voltmeter", "ammeter", "wattmeter", "varmeter", "pmu"):
            oracle.add_from_power_flow(tab, s, vm, va, fam)
        gn = check(oracle, s, tab, vm, va)
        assert gn.objective < 1e-12
        # row ordering: voltmeters, ammeters, wattmeters, varmeters, PMUs (2 rows each)
        non = int((s.status == 1).sum())
        assert list(gn.range) == [1, s.n + 1, s.n + 2 * non + 1, 2 * s.n + 4 * non + 1, 3 * s.n + 6 * non + 1,
                                  3 * s.n + 6 * non + 2 * (s.n + 2 * non) + 1]


def test_pmu_covariance_matrix(oracle):
    """test/stateEstimation/analysis.jl:300-346: precision == inv(covariance) for rectangular PMUs."""
    t = dict(
        base_power=1e8, bus_type=np.array([3, 1, 1], dtype=np.int8), bus_pd=np.array([0.5, 0, 0.5]), bus_qd=np.array([0, 0.05, 0.0]),
        bus_gs=np.zeros(3), bus_bs=np.zeros(3), bus_vm=np.ones(3), bus_va=np.zeros(3),
        br_from=np.array([1, 1, 2]), br_to=np.array([2, 2, 3]), br_status=np.ones(3, dtype=np.int8), br_r=np.full(3, 0.02),
        br_x=np.array([0.05, 0.01, 0.04]), br_g=np.full(3, 1e-4), br_b=np.full(3, 0.04), br_tap=np.ones(3), br_shift=np.zeros(3),
        gen_bus=np.array([1]), gen_status=np.ones(1, dtype=np.int8), gen_pg=np.array([3.2]), gen_qg=np.array([0.2]), gen_vg=np.ones(1))
    s = oracle.OracleSystem(t)
    tab = oracle.MeterTable()
    cov = np.zeros((6, 6))
    zv, zt, vv, vt = 0.9, 0.5, 1e-2, 1.6
    tab.add("pmu", 0, 3, zv, vv, 1, zt, vt, 1, correlated=True)
    cov[0, 0] = vv * np.cos(zt) ** 2 + vt * (zv * np.sin(zt)) ** 2
    cov[1, 1] = vv * np.sin(zt) ** 2 + vt * (zv * np.cos(zt)) ** 2
    cov[0, 1] = cov[1, 0] = np.cos(zt) * np.sin(zt) * (vv - vt * zv ** 2)
    zv, zt, vv, vt = 0.8, -0.3, 0.5, 2.6
    tab.add("pmu", 1, 3, zv, vv, 1, zt, vt, 1)
    cov[2, 2] = vv * np.cos(zt) ** 2 + vt * (zv * np.sin(zt)) ** 2
    cov[3, 3] = vv * np.sin(zt) ** 2 + vt * (zv * np.cos(zt)) ** 2
    zv, zt, vv, vt = 1.3, -0.2, 1e-1, 0.2
    tab.add("pmu", 2, 2, zv, vv, 1, zt, vt, 1, correlated=True)
    cov[4, 4] = vv * np.cos(zt) ** 2 + vt * (zv * np.sin(zt)) ** 2
    cov[5, 5] = vv * np.sin(zt) ** 2 + vt * (zv * np.cos(zt)) ** 2
    cov[4, 5] = cov[5, 4] = np.cos(zt) * np.sin(zt) * (vv - vt * zv ** 2)
    gn = oracle.OracleGN(s, tab)
    assert np.allclose(np.linalg.inv(cov), gn.precision_dense(), rtol=1e-10, atol=1e-12)


def test_squared_current_precision(oracle):
    """test/stateEstimation/analysis.jl:173-189: mean z^2, variance 4 z^2 sigma^2."""
    t, s, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    tab.add("ammeter", 1, 1, 0.5, 0.01, square=True)
    tab.add("pmu", 1, 1, 0.5, 0.01, 1, 0.1, 0.02, 1, polar=True, square=True)
    gn = oracle.OracleGN(s, tab)
    var = 4 * 0.5 ** 2 * 0.01
    assert gn.mean[0] == 0.5 ** 2 and np.isclose(gn.wdiag[0], 1 / var)
    assert gn.mean[1] == 0.5 ** 2 and np.isclose(gn.wdiag[1], 1 / var)
    assert gn.wdiag[2] == 1 / 0.02
    assert list(gn.type) == [4, 4, 14]


def test_jacobian_matches_finite_differences(oracle):
    t, s, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    for fam, kw in (("voltmeter", {}), ("ammeter", {}), ("ammeter", dict(square=True)), ("wattmeter", {}), ("varmeter", {}),
                    ("pmu", dict(polar=True)), ("pmu", {})):
        oracle.add_from_power_flow(tab, s, vm, va, fam, **kw)
    rng = np.random.default_rng(1)
    x_vm = vm + 0.01 * rng.standard_normal(s.n)
    x_va = va + 0.01 * rng.standard_normal(s.n)
    gn = oracle.OracleGN(s, tab, x_vm, x_va)
    gn.increment()
    v0 = gn.vectors()
    import scipy.sparse as sp
    H = sp.csc_matrix((v0["jacobian"], gn.hrowval - 1, gn.hcolptr - 1), shape=(gn.m, 2 * s.n))
    d = 1e-6 * rng.standard_normal(2 * s.n)
    gn2 = oracle.OracleGN(s, tab, x_vm + d[s.n:], x_va + d[:s.n])
    gn2.increment()
    v1 = gn2.vectors()
    # residual = z - h(x)  =>  r0 - r1 = H d
    assert np.abs((v0["residual"] - v1["residual"]) - H @ d).max() < 5e-8    # second-order term of a 1e-6 step
