"""update{Bus,Branch,Generator}!(analysis; ...) on a live power-flow analysis (the reference's reusing tests,
test/powerFlow/reusing.jl; testReusing in test/utility/utility.jl:197-220): after every update

    pf = newtonRaphson(analysis.system); powerFlow!(pf; tolerance = 1e-10)
    setInitialPoint!(analysis);          powerFlow!(analysis; tolerance = 1e-10)

must give the same iteration count and voltages to 1e-8 -- here also against the oracle built from the updated tables;
for fast Newton-Raphson the two constant matrices must equal those of the fresh analysis."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def _tables_of(s):
    """case tables of a (mutated) product PowerSystem, for the oracle."""
    inv = {v: k for k, v in s.bus.label.items()}
    return dict(
        base_power=np.array([s.base.power]), bus_label=np.array([inv[i + 1] for i in range(s.bus.number)]),
        bus_type=s.bus.layout.type.copy(), bus_pd=s.bus.demand.active.copy(), bus_qd=s.bus.demand.reactive.copy(),
        bus_gs=s.bus.shunt.conductance.copy(), bus_bs=s.bus.shunt.susceptance.copy(),
        bus_vm=s.bus.voltage.magnitude.copy(), bus_va=s.bus.voltage.angle.copy(),
        br_from=s.branch.layout.from_.copy(), br_to=s.branch.layout.to.copy(), br_status=s.branch.layout.status.copy(),
        br_r=s.branch.parameter.resistance.copy(), br_x=s.branch.parameter.reactance.copy(),
        br_g=s.branch.parameter.conductance.copy(), br_b=s.branch.parameter.susceptance.copy(),
        br_tap=s.branch.parameter.turnsRatio.copy(), br_shift=s.branch.parameter.shiftAngle.copy(),
        gen_bus=s.generator.layout.bus.copy(), gen_status=s.generator.layout.status.copy(),
        gen_pg=s.generator.output.active.copy(), gen_qg=s.generator.output.reactive.copy(),
        gen_vg=s.generator.voltage.magnitude.copy(), gen_qmin=s.generator.capability.minReactive.copy(),
        gen_qmax=s.generator.capability.maxReactive.copy())


def _reuse_check(jg, oracle, an, make, fast=None, same_pattern=True):
    fresh = make(jg.powerSystem(_tables_of(an.system)))
    jg.powerFlow_(fresh, iteration=100, tolerance=1e-10)
    jg.setInitialPoint_(an)
    jg.powerFlow_(an, iteration=100, tolerance=1e-10)
    assert an.status == fresh.status == 0 and an.method.iteration == fresh.method.iteration
    assert np.abs(an.voltage.magnitude - fresh.voltage.magnitude).max() < 1e-8
    assert np.abs(an.voltage.angle - fresh.voltage.angle).max() < 1e-8
    osys = oracle.OracleSystem(_tables_of(an.system))
    if fast is None:
        o = oracle.OracleNR(osys)
        assert o.power_flow(100, 1e-10) == 0
    else:
        if same_pattern:                                         # (after dropZeros! a fresh model stores zeros the reused one dropped)
            assert np.allclose(an.method.active.jacobian.nzval, fresh.method.active.jacobian.nzval, rtol=1e-13, atol=0)
            assert np.allclose(an.method.reactive.jacobian.nzval, fresh.method.reactive.jacobian.nzval, rtol=1e-13, atol=0)
        o = oracle.OracleFastNR(osys, bx=fast)
        assert o.power_flow(100, 1e-10) == 0
    vm, va = o.voltage() if fast is None else (o.vm, o.va)
    assert o.iteration == an.method.iteration
    assert np.abs(an.voltage.magnitude - vm).max() < 1e-8 and np.abs(an.voltage.angle - va).max() < 1e-8
    fresh.close()


BUS_STEPS = [                                                    # reusing.jl:11-31
    dict(label=14, active=0.12, reactive=0.13),
    dict(label=14, conductance=0.01, susceptance=0.02),
    dict(label=14, conductance=0.03, susceptance=0.01),
    dict(label=14, magnitude=1.02, angle=-0.17),               # the reference starts bus 14 at 1.2 pu: full NR then diverges
                                                                 # from that point on (it only compares fresh and reused runs)
    dict(label=7, active=0.15),
    dict(label=10, active=0.12, susceptance=0.05, angle=-0.2),
]
BRANCH_STEPS = [                                                 # reusing.jl:56-75 (labels B12 / B5)
    dict(label=12, resistance=0.02, status=1),
    dict(label=5, reactance=0.28, susceptance=0.001, status=1),
    dict(label=5, turnsRatio=0.99, status=0),
    dict(label=5, status=1),
    dict(label=5, status=0),
    dict(label=12, status=0),
    dict(label=12, conductance=0.01, status=1),
    dict(label=9, shiftAngle=-0.05, turnsRatio=0.97),
]
GEN_STEPS = [                                                    # reusing.jl:94-107 in spirit (case14test units)
    dict(label=2, active=0.35, reactive=0.1),
    dict(label=3, magnitude=1.02),
    dict(label=1, active=2.1),
]


@pytest.mark.parametrize("kind", ["nr", "bx", "xb"])
def test_updates_match_a_fresh_analysis(jg, oracle, kind):
    make = {"nr": jg.newtonRaphson, "bx": jg.fastNewtonRaphsonBX, "xb": jg.fastNewtonRaphsonXB}[kind]
    fast = None if kind == "nr" else (kind == "bx")
    s = jg.powerSystem(load_case("case14test"))
    an = make(s)
    _reuse_check(jg, oracle, an, make, fast)
    for kw in BUS_STEPS:
        jg.updateBus_(an, **kw)
        _reuse_check(jg, oracle, an, make, fast)
    jg.updateBusSystem_(s, label=10, magnitude=0.99, angle=-0.2, active=0.2, reactive=0.1)      # system first (:33-35)
    jg.updateBus_(an, label=10, conductance=0.01, susceptance=0.02, active=0.2)
    _reuse_check(jg, oracle, an, make, fast)
    for kw in BRANCH_STEPS:
        jg.updateBranch_(an, **kw)
        _reuse_check(jg, oracle, an, make, fast)
    for kw in GEN_STEPS:
        jg.updateGenerator_(an, **kw)
        _reuse_check(jg, oracle, an, make, fast)
    an.close()


def test_generator_status_and_type_conversion(jg, oracle):
    s = jg.powerSystem(load_case("case14test"))
    an = jg.newtonRaphson(s)
    gens_at = {}
    for k in range(s.generator.number):
        if s.generator.layout.status[k] == 1:
            gens_at.setdefault(int(s.generator.layout.bus[k]), []).append(k + 1)
    multi = [g for g in gens_at.values() if len(g) > 1]
    single = [g for b, g in gens_at.items() if len(g) == 1 and s.bus.layout.type[b - 1] == 2]
    if multi:                                                    # one of several units at a bus may leave and return
        jg.updateGenerator_(an, label=multi[0][-1], status=0)
        _reuse_check(jg, oracle, an, jg.newtonRaphson)
        jg.updateGenerator_(an, label=multi[0][-1], status=1)
        _reuse_check(jg, oracle, an, jg.newtonRaphson)
    assert single
    with pytest.raises(RuntimeError):                            # errorTypeConversion (generator.jl:398-402)
        jg.updateGenerator_(an, label=single[0][0], status=0)
    assert s.generator.layout.status[single[0][0] - 1] == 1      # refused before the system was touched
    an.close()


ADD_STEPS = [                                                    # reusing.jl:40-50: new branches, some between buses with no Ybus entry yet
    dict(from_=2, to=3, resistance=0.02, reactance=0.35),
    dict(from_=3, to=5, resistance=0.02, reactance=0.35, conductance=0.001),
    dict(from_=11, to=12, reactance=0.12, turnsRatio=0.95, shiftAngle=-0.17),
    dict(from_=16, to=7, resistance=0.01, reactance=0.23, susceptance=0.1),
]


@pytest.mark.parametrize("kind", ["nr", "bx"])
def test_pattern_changing_edits_rebuild_the_model(jg, oracle, kind):
    """addBranch!(analysis; ...) and dropZeros! on a live analysis (test/powerFlow/reusing.jl:40-50, test/powerFlow/analysis.jl:100-109;
    acPowerFlow.jl:806-811): when the Ybus pattern changes the next solve rebuilds maps, pattern and symbolic analysis, and the
    analysis keeps answering like a fresh one."""
    make = {"nr": jg.newtonRaphson, "bx": jg.fastNewtonRaphsonBX}[kind]
    fast = None if kind == "nr" else True
    s = jg.powerSystem(load_case("case14test"))
    an = make(s)
    _reuse_check(jg, oracle, an, make, fast)
    pattern = [s.model.revision.acPattern]
    for kw in ADD_STEPS:
        jg.addBranch_(an, **kw)
        pattern.append(s.model.revision.acPattern)
        _reuse_check(jg, oracle, an, make, fast)
    assert pattern[-1] > pattern[0] and len(set(pattern)) < len(pattern)      # some additions grew the pattern, some did not
    nnz_j = an.dims["nnzJ"]
    jg.updateBranch_(an, label=5, status=0)
    jg.dropZeros_(an)                                            # the stored zeros of the branches out of service leave Ybus
    _reuse_check(jg, oracle, an, make, fast, same_pattern=False)
    assert an.dims["nnzJ"] < nnz_j                               # the Jacobian pattern followed
    jg.updateBranch_(an, label=5, status=1)                      # ... and the entries come back with the branch
    _reuse_check(jg, oracle, an, make, fast, same_pattern=False)
    assert an.dims["nnzJ"] <= nnz_j                              # (a model built from scratch stores the zeros of the other branch out of service again)
    an.close()


def test_batched_analysis_survives_a_pattern_change(jg, oracle):
    """A batch with per-scenario outages and injections: after addBranch! between two unconnected buses the rebuilt handle still
    holds every scenario's outage (re-expressed on the new pattern), injections and state."""
    s = jg.powerSystem(load_case("case118"))
    labels = [int(x) for x in jg.outageList(s, 6, seed=3)] + [0]
    an = jg.contingencyAnalysis(s, labels)
    scale = np.linspace(0.9, 1.1, len(labels))[:, None]
    jg.setInjection_(an, (s.bus.supply.active - s.bus.demand.active)[None, :] * scale, (s.bus.supply.reactive - s.bus.demand.reactive)[None, :] * scale)
    f, t = int(s.branch.layout.from_[0]), 60
    Y = s.model.ac.nodalMatrix
    assert not Y.has(f, t)
    inv = {v: k for k, v in s.bus.label.items()}
    jg.addBranch_(an, from_=inv[f], to=inv[t], resistance=0.01, reactance=0.2)
    jg.powerFlow_(an, iteration=30, tolerance=1e-10)
    assert np.all(an.status == 0)
    fresh_sys = jg.powerSystem(_tables_of(s))
    fresh = jg.contingencyAnalysis(fresh_sys, labels)
    jg.setInjection_(fresh, (s.bus.supply.active - s.bus.demand.active)[None, :] * scale, (s.bus.supply.reactive - s.bus.demand.reactive)[None, :] * scale)
    jg.powerFlow_(fresh, iteration=30, tolerance=1e-10)
    assert np.abs(an.voltage.magnitude - fresh.voltage.magnitude).max() < 1e-8 and np.abs(an.voltage.angle - fresh.voltage.angle).max() < 1e-8
