"""Grids beyond 10 000 buses: the reference lists 25 000 / 70 000 / 82 000-bus datasets
(docs/src/examples/powerSystemDatasets.md:13-15) that it does not ship.  Stand-ins of that size, both seeded:
  * 25 000 buses   juliagrid.jl_amd.synthetic.pegaseShaped(n = 25 000)           (the generator behind case9241synth)
  * 70 000 buses   juliagrid.jl_amd.synthetic.tiledGrid(case_ACTIVSg10k, 7)      (seven tied instances of the shipped 10k grid)
  * 90 000 buses   tiledGrid(case_ACTIVSg10k, 9, star = True)                    (for the 82 000-bus set: nine instances tied to the first one)
Checked: the oracle converges; single-instance NR parity on the GPU (iteration count equal, V / theta 1e-8); a 512-scenario N-1 batch
fits and runs (symbolic analysis, int32 tables, top-task caps, memory at that size), spot-checked against the oracle."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def big_tables(oracle, which):
    from juliagrid.jl_amd.synthetic import pegaseShaped, tiledGrid
    if which == "synth25k":
        t = pegaseShaped(n=25000, nb=int(25000 * 16049 / 9241), ng=int(25000 * 1445 / 9241), seed=25000, load_scale=0.1)
        return {k: np.array(v) for k, v in t.items()}
    t = load_case("case_ACTIVSg10k")
    o = oracle.OracleNR(oracle.OracleSystem(t))
    assert o.power_flow() == 0
    vm, va = o.voltage()
    slack = int(np.flatnonzero(t["bus_type"] == 3)[0])
    p_slack = oracle.exact_quantities(oracle.OracleSystem(t), vm, va)[1][slack, 0] + t["bus_pd"][slack]
    return tiledGrid(t, 9, slack_active=p_slack, star=True) if which == "tiled90k" else tiledGrid(t, 7, slack_active=p_slack)


@pytest.mark.parametrize("which,n", [("synth25k", 25000), ("tiled70k", 70000), ("tiled90k", 90000)])
def test_single_instance_and_batch_on_a_big_grid(jg, oracle, which, n):
    t = big_tables(oracle, which)
    assert t["bus_type"].size == n
    osys = oracle.OracleSystem({k: np.array(v) for k, v in t.items()})
    o = oracle.OracleNR(osys)
    assert o.power_flow(iteration=20, tolerance=1e-8) == 0 and 2 <= o.iteration <= 10
    vm, va = o.voltage()
    s = jg.powerSystem({k: np.array(v) for k, v in t.items()})
    an = jg.newtonRaphson(s)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    assert an.status == 0 and an.method.iteration == o.iteration
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-8 and np.abs(an.voltage.angle - va).max() <= 1e-8
    print(f"[{which}] n {n} dimJ {an.dims['dimJ']} lu blocks {an.dims['lu_blocks']} terms {an.dims['lu_terms']} launches {an.dims['lu_launches']} + {an.dims['solve_launches']}, "
          f"{o.iteration} iterations")
    an.close()
    # 512 N-1 scenarios from the base-case solution
    labels = jg.outageList(s, 512, seed=512)
    batch = jg.contingencyAnalysis(s, labels)
    batch.setVoltage(vm, va) if hasattr(batch, "setVoltage") else jg.powerflow._push_voltage(batch, vm, va)
    jg.powerFlow_(batch, iteration=20, tolerance=1e-8)
    st = np.asarray(batch.status)
    assert np.mean(st == 0) >= 0.9
    rng = np.random.default_rng(1)
    for b in rng.choice(np.flatnonzero(st == 0), 3, replace=False):
        ptr, dy = jg.outagePatch(s, int(labels[b]))
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, dv)
        o.set_voltage(vm, va)
        assert o.power_flow(iteration=20, tolerance=1e-8) == 0
        assert o.iteration == batch.method.iteration[b]
        v2, a2 = o.voltage()
        assert np.abs(batch.voltage.magnitude[b] - v2).max() <= 1e-8 and np.abs(batch.voltage.angle[b] - a2).max() <= 1e-8
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, -dv)
    # round 6: the same screen with its first iteration on the base case's shared factor (jg_nr_base_*: one factorisation, J0^-1 on the Ybus pattern by one solve
    # per bus and component, a dense inverse of the tree's top) -- equal iteration counts, states to rounding
    import time
    it_ref, vm_ref, va_ref = batch.method.iteration.copy(), batch.voltage.magnitude.copy(), batch.voltage.angle.copy()
    single = jg.newtonRaphson(s)
    jg.powerflow._push_voltage(single, vm, va)
    t0 = time.perf_counter()
    base = jg.BaseCase(single)
    t_base = time.perf_counter() - t0
    single.close()
    base.attach(batch)
    jg.startFromBase_(batch)
    jg.powerFlow_(batch, iteration=20, tolerance=1e-8)
    assert jg.firstIterationCounts(batch) == (1, 1)
    ok = st == 0
    assert np.array_equal(np.asarray(batch.status), st) and np.array_equal(batch.method.iteration[ok], it_ref[ok])
    assert np.abs(batch.voltage.magnitude[ok] - vm_ref[ok]).max() <= 1e-9 and np.abs(batch.voltage.angle[ok] - va_ref[ok]).max() <= 1e-9
    print(f"[{which}] base case in {1e3 * t_base:.0f} ms: {base.info}")
    base.close()
    batch.close()
