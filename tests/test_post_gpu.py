"""N2: power!(analysis) / current!(analysis) on the device for batched states, against the oracle's
restatement of src/postprocessing/acAnalysis.jl.  Tolerance 1e-11 relative to the largest quantity of each family
(same formulas, f64; device sincos differs from libm by <= 2 ulp)."""
import numpy as np
import pytest

from conftest import load_case, load_golden

pytestmark = pytest.mark.gpu

POWER = ["injection", "shunt", "supply", "from_", "to", "series", "charging", "generator"]
CURRENT = {"injection": "i_injection", "from_": "i_from", "to": "i_to", "series": "i_series"}


def _close(got, ref, what):
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(np.asarray(got) - ref).max() <= 1e-11 * scale, what


def _angles_close(got, ref, mag, what):
    """angles are compared as phasors: the angle of a (numerically) zero current is meaningless"""
    scale = max(1.0, float(np.abs(mag).max()))
    assert np.abs(mag * np.exp(1j * np.asarray(got)) - mag * np.exp(1j * ref)).max() <= 1e-11 * scale, what


@pytest.mark.parametrize("name", ["case14", "case30test", "case118", "case300", "case1354pegase", "case_ACTIVSg10k"])
def test_power_and_current_match_oracle(jg, oracle, name):
    t = load_case(name)
    an = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(an)
    jg.power_(an)
    jg.current_(an)
    o = oracle.OracleNR(oracle.OracleSystem(t))
    assert o.power_flow() == 0
    ref = oracle.power_and_current(o.sys, an.voltage.magnitude, an.voltage.angle)
    for fam in POWER:
        got = getattr(an.power, fam)
        _close(got.active, ref[fam][0], (name, fam, "active"))
        _close(got.reactive, ref[fam][1], (name, fam, "reactive"))
    for fam, key in CURRENT.items():
        got = getattr(an.current, fam)
        _close(got.magnitude, ref[key][0], (name, fam, "magnitude"))
        _angles_close(got.angle, ref[key][1], ref[key][0], (name, fam, "angle"))
    # power balance of the solved case: sum of injections = losses (series + charging + shunts)
    s = an.power
    assert abs(s.injection.active.sum() - (s.series.active.sum() + s.charging.active.sum() + s.shunt.active.sum())) < 1e-8


def test_batched_outages_zero_the_outaged_branch(jg, oracle):
    """Scenario s of an N-1 batch: branch labels[s] carries no flow and no current there, everything else equals the
    oracle evaluated on that scenario's own state with that branch switched off."""
    t = load_case("case118")
    s = jg.powerSystem(t)
    labels = [int(x) for x in jg.outageList(s, 5, seed=2)] + [0]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an)
    jg.power_(an)
    jg.current_(an)
    for sc, lab in enumerate(labels):
        t2 = {k: np.array(v) for k, v in t.items()}
        if lab:
            t2["br_status"][lab - 1] = 0
            assert an.power.from_.active[sc, lab - 1] == 0.0 and an.current.to.magnitude[sc, lab - 1] == 0.0
        osys = oracle.OracleSystem(t2)
        osys.type[:] = s.bus.layout.type            # the batch shares the base case's bus types
        ref = oracle.power_and_current(osys, an.voltage.magnitude[sc], an.voltage.angle[sc])
        for fam in ("injection", "from_", "to", "series", "charging"):
            got = getattr(an.power, fam)
            _close(got.active[sc], ref[fam][0], (sc, fam))
            _close(got.reactive[sc], ref[fam][1], (sc, fam))
        _close(an.current.from_.magnitude[sc], ref["i_from"][0], (sc, "i_from"))


@pytest.mark.parametrize("name", ["case14test", "case30test"])
def test_power_hits_the_matpower_goldens(jg, name):
    """testPower of the reference (test/utility/utility.jl:41-59, atol 1e-8) through the device path."""
    g = load_golden(name)
    an = jg.newtonRaphson(jg.powerSystem(load_case(name)))
    jg.powerFlow_(an, tolerance=1e-10)
    jg.power_(an)
    pw = an.power
    pairs = [(pw.injection, "injection"), (pw.supply, "supply"), (pw.shunt, "shunt"), (pw.from_, "from"), (pw.to, "to"),
             (pw.generator, "generator")]
    for got, key in pairs:
        assert np.abs(got.active - g[f"newtonRaphson_{key}Active"]).max() <= 1e-8, key
        assert np.abs(got.reactive - g[f"newtonRaphson_{key}Reactive"]).max() <= 1e-8, key
    assert np.abs(pw.series.active - g["newtonRaphson_lossActive"]).max() <= 1e-8
    assert np.abs(pw.series.reactive - g["newtonRaphson_lossReactive"]).max() <= 1e-8
    assert np.abs(pw.charging.reactive - (g["newtonRaphson_chargingFrom"] + g["newtonRaphson_chargingTo"])).max() <= 1e-8


@pytest.mark.parametrize("name", ["case14test", "case30test"])
def test_reactive_limits_hit_the_matpower_goldens(jg, name):
    """test/powerFlow/limits.jl:4-42 through the device path (newtonRaphson -> powerFlow! -> reactiveLimit! ->
    newtonRaphson -> powerFlow! -> adjustAngle!)."""
    g = load_golden(name)
    system = jg.powerSystem(load_case(name))
    slack0 = system.bus.layout.slack
    an = jg.newtonRaphson(system)
    jg.powerFlow_(an)
    it0 = int(an.method.iteration)
    violate = jg.reactiveLimit_(an)
    assert np.any(violate != 0)
    an2 = jg.newtonRaphson(system)
    jg.powerFlow_(an2)
    jg.adjustAngle_(an2, slack0)
    assert it0 + int(an2.method.iteration) == int(g["reactiveLimit_newtonRaphson_iteration"][0])
    assert np.abs(an2.voltage.magnitude - g["reactiveLimit_newtonRaphson_voltageMagnitude"]).max() <= 1e-8
    assert np.abs(an2.voltage.angle - g["reactiveLimit_newtonRaphson_voltageAngle"]).max() <= 1e-8
