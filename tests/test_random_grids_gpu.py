"""Newton-Raphson parity on seeded random grids of different sizes and shapes (juliagrid.jl_amd.synthetic.pegaseShaped):
every grid gets its own elimination order, replay tables and chain structure, so this exercises the symbolic analysis and
the level / chain kernels on structures the fixtures do not contain.  Per grid: base case + N-1 outages in one batch,
iteration counts equal to the oracle's, V and theta to 1e-8 (the reference's bar), and the WLS known answer (noise-free
measurements => power-flow state) through the symmetric engine."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GRIDS = [(72, 118, 12, 11), (160, 270, 25, 12), (777, 1300, 120, 13), (2869, 4582, 510, 14)]


@pytest.mark.parametrize("n,nb,ng,seed", GRIDS)
def test_nr_batch_matches_the_oracle(jg, oracle, n, nb, ng, seed):
    t = jg.pegaseShaped(n=n, nb=nb, ng=ng, seed=seed, load_scale=0.15)
    osys = oracle.OracleSystem(t)
    o = oracle.OracleNR(osys)
    assert o.power_flow(30, 1e-10) == 0
    s = jg.powerSystem(t)
    labels = [0] + [int(x) for x in jg.outageList(s, 5, seed=seed)]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an, iteration=30, tolerance=1e-10)
    vm, va = o.voltage()
    assert an.status[0] == 0 and an.method.iteration[0] == o.iteration
    assert np.abs(an.voltage.magnitude[0] - vm).max() < 1e-8 and np.abs(an.voltage.angle[0] - va).max() < 1e-8
    sref = jg.powerSystem(t)
    jg.acModel_(sref)
    for k, lab in enumerate(labels[1:], start=1):
        oo = oracle.OracleNR(oracle.OracleSystem(t))
        ptr, dy = jg.outagePatch(sref, lab)
        for p, dv in zip(ptr, dy):
            oo.add_ybus(p - 1, dv)
        st = oo.power_flow(30, 1e-10)
        assert an.status[k] == st
        if st == 0:
            assert an.method.iteration[k] == oo.iteration
            ovm, ova = oo.voltage()
            assert np.abs(an.voltage.magnitude[k] - ovm).max() < 1e-8 and np.abs(an.voltage.angle[k] - ova).max() < 1e-8
    an.close()


@pytest.mark.parametrize("n,nb,ng,seed", GRIDS[:3])
def test_wls_known_answer(jg, n, nb, ng, seed):
    t = jg.pegaseShaped(n=n, nb=nb, ng=ng, seed=seed, load_scale=0.15)
    s = jg.powerSystem(t)
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, iteration=30, tolerance=1e-11)
    assert pf.status == 0
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf)
    jg.addWattmeter_(mon, pf)
    jg.addVarmeter_(mon, pf)
    jg.addPmu_(mon, pf, buses=range(1, n + 1, 7), statusTo=-1, minMagnitude=1e-6)
    an = jg.gaussNewton(mon, batch=2)
    jg.stateEstimation_(an, iteration=60, tolerance=1e-11)
    assert np.all(an.status == 0)
    assert np.abs(an.voltage.magnitude[0] - pf.voltage.magnitude).max() < 1e-9
    assert np.abs(an.voltage.angle[0] - pf.voltage.angle).max() < 1e-9
    out = jg.residualTest_(an)
    assert not out.detect.any()
    an.close()
    pf.close()
