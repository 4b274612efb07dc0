"""update<Meter>!(analysis; ...) on a live Gauss-Newton analysis (the reference's "reusing" tests,
test/stateEstimation/reusing.jl + testReusing in test/utility/utility.jl): after every update the analysis must be
indistinguishable from one built from scratch on the updated Measurement container -- se.type, se.mean, se.precision
equal, same iteration count, voltages to 1e-8 -- and both must agree with the oracle."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def _table_of(oracle, mon):
    """oracle MeterTable with the same devices as a product Measurement container."""
    tab = oracle.MeterTable()
    loc = lambda lay, i: 0 if getattr(lay, "bus", None) and lay.bus[i] else (1 if lay.from_[i] else 2)
    v = mon.voltmeter
    for i in range(v.number):
        tab.add("voltmeter", 0, v.layout.index[i], v.magnitude.mean[i], v.magnitude.variance[i], v.magnitude.status[i])
    a = mon.ammeter
    for i in range(a.number):
        tab.add("ammeter", 1 if a.layout.from_[i] else 2, a.layout.index[i], a.magnitude.mean[i], a.magnitude.variance[i],
                a.magnitude.status[i], square=a.layout.square[i])
    for fam, meter, g in (("wattmeter", mon.wattmeter, mon.wattmeter.active), ("varmeter", mon.varmeter, mon.varmeter.reactive)):
        for i in range(meter.number):
            tab.add(fam, loc(meter.layout, i), meter.layout.index[i], g.mean[i], g.variance[i], g.status[i])
    p = mon.pmu
    for i in range(p.number):
        tab.add("pmu", loc(p.layout, i), p.layout.index[i], p.magnitude.mean[i], p.magnitude.variance[i], p.magnitude.status[i],
                p.angle.mean[i], p.angle.variance[i], p.angle.status[i], square=p.layout.square[i], polar=p.layout.polar[i],
                correlated=p.layout.correlated[i])
    return tab


def _setup(jg, oracle):
    t = load_case("case14test")
    s = jg.powerSystem(t)
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf)
    osys = oracle.OracleSystem(t)
    opf = oracle.OracleNR(osys)
    assert opf.power_flow() == 0
    osys.type = opf.type.copy(); osys.slack = opf.slack
    mon = jg.measurement(s)                                     # reusing.jl:17-38
    jg.addVoltmeter_(mon, bus=1, magnitude=1.06)
    jg.addAmmeter_(mon, from_=2, magnitude=0.97)
    jg.addAmmeter_(mon, to=4, magnitude=1.18)
    jg.addWattmeter_(mon, bus=3, active=-0.94)
    jg.addWattmeter_(mon, from_=10, active=0.29)
    jg.addWattmeter_(mon, to=13, active=-0.16)
    jg.addVarmeter_(mon, bus=10, reactive=-0.06)
    jg.addVarmeter_(mon, from_=4, reactive=0.22)
    jg.addVarmeter_(mon, to=6, reactive=-0.62)
    jg.addPmu_(mon, pf, statusFrom=-1, statusTo=-1, polar=True, varianceMagnitudeBus=1.0, varianceAngleBus=1.0)
    jg.addPmu_(mon, bus=2, magnitude=1.045, angle=-0.075, polar=False, varianceMagnitude=1.0, varianceAngle=1.0)
    jg.addPmu_(mon, from_=10, magnitude=0.30, angle=-0.07, polar=True)
    jg.addPmu_(mon, from_=10, magnitude=0.30, angle=-0.07, polar=False)
    jg.addPmu_(mon, to=2, magnitude=0.98, angle=2.92, polar=True)
    jg.addPmu_(mon, to=2, magnitude=0.98, angle=2.92, polar=False)
    return s, osys, mon


def _same_as_fresh(jg, oracle, s, osys, mon, wls):
    fresh = jg.gaussNewton(mon)
    assert np.array_equal(wls.method.type, fresh.method.type)
    assert np.array_equal(wls.method.mean, fresh.method.mean)
    assert np.array_equal(wls.precision, fresh.precision)
    wls.setVoltage(s.bus.voltage.magnitude, s.bus.voltage.angle)           # setInitialPoint!
    jg.stateEstimation_(wls, iteration=40, tolerance=1e-10)
    jg.stateEstimation_(fresh, iteration=40, tolerance=1e-10)
    assert wls.method.iteration == fresh.method.iteration and wls.status == fresh.status == 0
    assert np.array_equal(wls.voltage.magnitude, fresh.voltage.magnitude) and np.array_equal(wls.voltage.angle, fresh.voltage.angle)
    gn = oracle.OracleGN(osys, _table_of(oracle, mon))
    assert gn.state_estimation(40, 1e-10) == 0 and gn.iteration == wls.method.iteration
    v = gn.vectors()
    assert np.abs(wls.voltage.magnitude - v["magnitude"]).max() < 1e-8 and np.abs(wls.voltage.angle - v["angle"]).max() < 1e-8
    assert np.array_equal(wls.method.type, gn.type)
    fresh.close()


def test_updates_match_a_fresh_analysis(jg, oracle):
    s, osys, mon = _setup(jg, oracle)
    wls = jg.gaussNewton(mon)
    _same_as_fresh(jg, oracle, s, osys, mon, wls)
    steps = [
        (jg.updateVoltmeter_, dict(label=1, magnitude=2.6, variance=1e30)),          # reusing.jl:43-61
        (jg.updateVoltmeter_, dict(label=1, variance=1e-4, status=0)),
        (jg.updateVoltmeter_, dict(label=1, magnitude=1.06, status=1)),
        (jg.updateAmmeter_, dict(label=1, magnitude=3.0, variance=1e30)),            # :64-99
        (jg.updateAmmeter_, dict(label=1, variance=1e-3, status=0)),
        (jg.updateAmmeter_, dict(label=1, magnitude=0.97, status=1)),
        (jg.updateAmmeter_, dict(label=1, square=True)),
        (jg.updateAmmeter_, dict(label=1, square=False)),
        (jg.updateAmmeter_, dict(label=2, square=True)),                             # :101-136
        (jg.updateAmmeter_, dict(label=2, variance=1e-4, status=0)),
        (jg.updateAmmeter_, dict(label=2, magnitude=1.18, status=1, square=False)),
        (jg.updateWattmeter_, dict(label=1, active=5.3, variance=1e45)),             # :138 ff
        (jg.updateWattmeter_, dict(label=1, variance=1e-4, status=0)),
        (jg.updateWattmeter_, dict(label=1, active=-0.94, status=1)),
        (jg.updateWattmeter_, dict(label=3, active=-0.2, variance=1e-3)),
        (jg.updateVarmeter_, dict(label=2, reactive=0.3, status=0)),
        (jg.updateVarmeter_, dict(label=2, status=1, variance=1e-2)),
        (jg.updatePmu_, dict(label=3, magnitude=1.2, statusAngle=0)),
        (jg.updatePmu_, dict(label=3, status=1, varianceAngle=1e-3)),
        (jg.updatePmu_, dict(label=17, status=0)),
        (jg.updatePmu_, dict(label=17, magnitude=0.31, angle=-0.08, status=1)),
    ]
    for fn, kw in steps:
        fn(wls, **kw)
        _same_as_fresh(jg, oracle, s, osys, mon, wls)
    # container first, analysis afterwards without arguments (reusing.jl:55-57)
    jg.updateVoltmeter_(mon, label=1, magnitude=1.07, status=0)
    jg.updateVoltmeter_(wls, label=1)
    _same_as_fresh(jg, oracle, s, osys, mon, wls)
    with pytest.raises(jg._lib.JGridError):
        jg.updateAmmeter_(mon, label=1, square=True)
        wls.method._code = wls.method._code                                          # an illegal code switch is refused by the library
        mon.ammeter.layout.from_[0] = False
        jg.stateestimation._refresh(wls)
    wls.close()
