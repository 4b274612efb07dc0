"""Parity of the HIP Newton-Raphson path (through the C ABI) with the CPU oracle and the reference's
golden vectors.  Tolerances (floating point, f64 everywhere):
  * index maps pq/pvpq/pcount/jacobian colptr,rowval ........ bit-exact
  * mismatch / Jacobian entries at the same state ............ 1e-12 relative to the largest entry
    (device sincos differs from libm by <= 2 ulp; summation order is identical)
  * Newton increment ......................................... 1e-9 relative (different LU algorithm:
    static-pivot block LU on the device vs threshold-pivoting Gilbert-Peierls in the oracle)
  * converged V, theta ....................................... 1e-8 absolute (the reference's own test bar,
    test/utility/utility.jl:34-40, 197-206); iteration counts equal
"""
import numpy as np
import pytest

from conftest import load_case, load_golden

pytestmark = pytest.mark.gpu

SMALL = ["case14", "case14test", "case30test", "case118", "case300"]


def _pair(jg, oracle, name, batch=1):
    t = load_case(name)
    s = jg.powerSystem(t)
    an = jg.newtonRaphson(s, batch=batch)
    o = oracle.OracleNR(oracle.OracleSystem(t))
    return s, an, o


@pytest.mark.parametrize("name", SMALL + ["case1354pegase", "case_ACTIVSg10k", "case9241synth"])
def test_index_maps_bit_exact(jg, oracle, name):
    s, an, o = _pair(jg, oracle, name)
    assert np.array_equal(an.method.pq, o.pq)
    assert np.array_equal(an.method.pvpq, o.pvpq)
    assert np.array_equal(an.method.pcount, o.pcount)
    J = an.jacobian
    assert np.array_equal(J.colptr, o.jcolptr)
    assert np.array_equal(J.rowval, o.jrowval)
    assert an.dims["dimJ"] == o.dim and an.dims["nnzJ"] == o.nnzJ


@pytest.mark.parametrize("name", SMALL + ["case1354pegase", "case_ACTIVSg10k", "case9241synth"])
def test_mismatch_and_jacobian_elementwise(jg, oracle, name):
    s, an, o = _pair(jg, oracle, name)
    dp, dq = jg.mismatch_(an)
    op, oq = o.mismatch()
    _, f_ref, _ = o.vectors()
    f = an.mismatch
    scale = max(1.0, np.abs(f_ref).max())
    assert np.abs(f - f_ref).max() <= 1e-12 * scale
    assert abs(dp - op) <= 1e-12 * scale and abs(dq - oq) <= 1e-12 * scale
    jv = an.jacobian.nzval                       # filled by the fused kernel
    o.solve()
    j_ref, _, inc_ref = o.vectors()
    assert np.abs(jv - j_ref).max() <= 1e-12 * np.abs(j_ref).max()
    jg.solve_(an)
    inc = an.increment
    assert np.abs(inc - inc_ref).max() <= 1e-9 * max(1.0, np.abs(inc_ref).max())
    vm, va = o.voltage()
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-9
    assert np.abs(an.voltage.angle - va).max() <= 1e-9
    assert an.method.iteration == 1 == o.iteration


@pytest.mark.parametrize("name,iters", [("case14test", 7), ("case30test", 4)])
def test_matpower_goldens(jg, name, iters):
    """test/powerFlow/analysis.jl:1-44 through testVoltage: iteration count and V, theta."""
    g = load_golden(name)
    an = jg.newtonRaphson(jg.powerSystem(load_case(name)))
    jg.powerFlow_(an)
    assert an.status == 0
    assert an.method.iteration == iters == int(g["newtonRaphson_iteration"][0])
    assert np.abs(an.voltage.magnitude - g["newtonRaphson_voltageMagnitude"]).max() <= 1e-8
    assert np.abs(an.voltage.angle - g["newtonRaphson_voltageAngle"]).max() <= 1e-8


@pytest.mark.parametrize("name", ["case14", "case118", "case300", "case1354pegase", "case1951rte", "case_ACTIVSg10k",
                                  "case9241synth"])
def test_power_flow_matches_oracle(jg, oracle, name):
    s, an, o = _pair(jg, oracle, name)
    jg.powerFlow_(an)
    assert o.power_flow() == 0 and an.status == 0
    assert an.method.iteration == o.iteration
    vm, va = o.voltage()
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-8
    assert np.abs(an.voltage.angle - va).max() <= 1e-8
    # residual of the device solution evaluated by the oracle
    o.set_voltage(an.voltage.magnitude, an.voltage.angle)
    assert max(o.mismatch()) < 1e-8


@pytest.mark.parametrize("name", ["case118", "case1354pegase", "case_ACTIVSg10k"])
def test_repeated_solves_of_one_handle_keep_the_reference_accounting(jg, oracle, name):
    """Round 6: a handle of ONE scenario that has solved before runs its next solve as ONE graph of as many iterations as the last solve took (run_whole);
    a solve that needs fewer runs the rest as empty launches, one that needs more gets its Jacobian (graph J) and carries on iteration by iteration.  Whatever
    the order of easy and hard starts, every solve must report the reference's iteration count and state (acPowerFlow.jl:1389-1433)."""
    s, an, o = _pair(jg, oracle, name)
    vm0, va0 = an.voltage.magnitude.copy(), an.voltage.angle.copy()
    assert o.power_flow() == 0
    k, (vm, va) = o.iteration, o.voltage()
    assert k >= 3

    def check(iters):
        assert an.status == 0 and an.method.iteration == iters
        assert np.abs(an.voltage.magnitude - vm).max() <= 1e-8 and np.abs(an.voltage.angle - va).max() <= 1e-8
    jg.powerFlow_(an); check(k)                                   # first solve: graph per iteration
    jg.setInitialPoint_(an); jg.powerFlow_(an); check(k)          # the whole solve as one graph of k iterations
    jg.powerFlow_(an); check(0)                                   # from the solution: converged at the start verdict, k empty iterations
    jg.setInitialPoint_(an); jg.powerFlow_(an); check(k)          # expects 1 iteration, needs k: graph J, then the loop
    jg.powerflow._push_voltage(an, 0.5 * (vm0 + vm), 0.5 * (va0 + va))                    # a start half-way: fewer iterations than k
    o.set_voltage(0.5 * (vm0 + vm), 0.5 * (va0 + va)); assert o.power_flow() == 0
    k2 = o.iteration
    jg.powerFlow_(an); check(k2)
    jg.setInitialPoint_(an); jg.powerFlow_(an, iteration=2)       # the limit below the expectation: the loop of single iterations, status 1
    assert an.status == 1 and an.method.iteration == 2
    jg.setInitialPoint_(an); jg.powerFlow_(an); check(k)


def test_iteration_limit_and_loop_accounting(jg, oracle):
    """acPowerFlow.jl:1406-1420: at most `iteration` solves; status 1 when the limit hits."""
    s, an, o = _pair(jg, oracle, "case14test")
    jg.powerFlow_(an, iteration=3)
    assert o.power_flow(iteration=3) == 1
    assert an.status == 1 and an.method.iteration == 3 == o.iteration
    vm, va = o.voltage()
    assert np.abs(an.voltage.magnitude - vm).max() <= 1e-9 and np.abs(an.voltage.angle - va).max() <= 1e-9
    jg.powerFlow_(an, iteration=0)
    assert an.method.iteration == 0


def test_set_initial_point_and_rerun(jg):
    """test/powerFlow/analysis.jl:52-57: setInitialPoint! restores the start voltages exactly."""
    an = jg.newtonRaphson(jg.powerSystem(load_case("case30test")))
    vm0, va0 = an.voltage.magnitude.copy(), an.voltage.angle.copy()
    jg.powerFlow_(an)
    v1 = an.voltage.magnitude.copy()
    jg.setInitialPoint_(an)
    assert np.array_equal(an.voltage.magnitude, vm0) and np.array_equal(an.voltage.angle, va0)
    jg.powerFlow_(an)
    assert an.method.iteration == 4
    assert np.array_equal(an.voltage.magnitude, v1)          # run-to-run bitwise determinism


def test_update_branch_reuse_equals_fresh(jg):
    """test/powerFlow/reusing.jl:59-69 + test/utility/utility.jl:197-206: after status toggles a reused
    analysis (refactorization path, stored zeros) equals a freshly built one: iterations equal,
    voltages within 1e-8 at tolerance 1e-10."""
    t = load_case("case14test")
    s = jg.powerSystem(t)
    an = jg.newtonRaphson(s)
    jg.powerFlow_(an, tolerance=1e-10)
    for label, status in ((12, 0), (3, 1), (12, 1), (5, 0)):
        jg.updateBranch_(an, label, status=status)
        jg.setInitialPoint_(an)
        jg.powerFlow_(an, tolerance=1e-10)
        t2 = dict(t)
        t2["br_status"] = s.branch.layout.status.copy()
        fresh = jg.newtonRaphson(jg.powerSystem(t2))
        jg.powerFlow_(fresh, tolerance=1e-10)
        assert an.method.iteration == fresh.method.iteration
        assert np.abs(an.voltage.magnitude - fresh.voltage.magnitude).max() <= 1e-8
        assert np.abs(an.voltage.angle - fresh.voltage.angle).max() <= 1e-8


def test_stale_bus_type_raises(jg):
    """acPowerFlow.jl:802-804: a bus-type revision invalidates the analysis."""
    s = jg.powerSystem(load_case("case14"))
    an = jg.newtonRaphson(s)
    s.model.revision.type += 1
    with pytest.raises(RuntimeError):
        jg.solve_(an)


def _non_bridge_branches(t, count, seed):
    """in-service branches whose removal keeps the grid connected (parallel pairs or cycle edges)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    n = t["bus_type"].size
    on = np.flatnonzero(t["br_status"] == 1)
    rng = np.random.default_rng(seed)
    rng.shuffle(on)
    out = []
    for k in on:
        keep = on[on != k]
        g = sp.coo_matrix((np.ones(keep.size), (t["br_from"][keep] - 1, t["br_to"][keep] - 1)), shape=(n, n))
        if connected_components(g, directed=False)[0] == 1:
            out.append(int(k) + 1)
        if len(out) == count:
            break
    return out


@pytest.mark.parametrize("name,batch", [("case118", 70), ("case1354pegase", 16)])
def test_batched_outages_match_oracle(jg, oracle, name, batch):
    """Batched N-1: scenario s = base grid with one branch out, expressed as 4 Ybus
    edits; each scenario must agree with the oracle solving that outage on its own."""
    t = load_case(name)
    labels = _non_bridge_branches(t, batch - 1, seed=7)
    s = jg.powerSystem(t)
    an = jg.newtonRaphson(s, batch=batch)
    for sc, lab in enumerate(labels):
        jg.setOutage_(an, sc, lab)          # last scenario stays the base case
    jg.powerFlow_(an)
    osys = oracle.OracleSystem(t)
    for sc in range(batch):
        o = oracle.OracleNR(osys)
        if sc < len(labels):
            ptr, dy = jg.outagePatch(s, labels[sc])
            for p, d in zip(ptr, dy):
                o.add_ybus(p - 1, d)
        st = o.power_flow()
        assert an.status[sc] == st
        if st == 0:
            assert an.method.iteration[sc] == o.iteration, (sc, labels[sc] if sc < len(labels) else None)
            vm, va = o.voltage()
            assert np.abs(an.voltage.magnitude[sc] - vm).max() <= 1e-8
            assert np.abs(an.voltage.angle[sc] - va).max() <= 1e-8


def test_batched_injections_match_oracle(jg, oracle):
    """Monte-Carlo load variations: per-scenario injections, shared topology."""
    t = load_case("case300")
    s = jg.powerSystem(t)
    B = 5
    an = jg.newtonRaphson(s, batch=B)
    rng = np.random.default_rng(3)
    scale = 1.0 + 0.01 * rng.standard_normal((B, 1))
    pd = s.bus.demand.active[None, :] * scale
    qd = s.bus.demand.reactive[None, :] * scale
    jg.setInjection_(an, s.bus.supply.active[None, :] - pd, s.bus.supply.reactive[None, :] - qd)
    jg.powerFlow_(an)
    osys = oracle.OracleSystem(t)
    for b in range(B):
        o = oracle.OracleNR(osys)
        o.set_power(osys.ps, osys.qs, pd[b], qd[b])
        assert o.power_flow() == 0 and an.status[b] == 0
        assert an.method.iteration[b] == o.iteration
        vm, va = o.voltage()
        assert np.abs(an.voltage.magnitude[b] - vm).max() <= 1e-8
        assert np.abs(an.voltage.angle[b] - va).max() <= 1e-8


def test_monte_carlo_injections_at_config2_scale(jg, oracle):
    """SURVEY 8(d) config 2 / north_star "Monte-Carlo instances": B = 512 load-perturbed copies of case1354pegase in ONE handle -- every scenario its
    own demand (each bus's active and reactive demand scaled by an independent N(1, 0.05^2) draw, PCG64(1354): what a user loop over updateBus!(;
    active, reactive) does, bus.jl:286-298, :314-323), shared topology.  EVERY lane against the
    oracle solving that scenario alone: equal iteration counts, V / theta 1e-8; every scenario converges; two scenarios with the same draw are bitwise
    equal (lanes do not interact)."""
    t = load_case("case1354pegase")
    s = jg.powerSystem(t)
    B, n = 512, s.bus.number
    an = jg.newtonRaphson(s, batch=B)
    rng = np.random.Generator(np.random.PCG64(1354))
    fp = 1.0 + 0.05 * rng.standard_normal((B, n))
    fq = 1.0 + 0.05 * rng.standard_normal((B, n))
    fp[B - 1], fq[B - 1] = fp[3], fq[3]                               # a duplicate of scenario 3 in another lane group
    pd = s.bus.demand.active[None, :] * fp
    qd = s.bus.demand.reactive[None, :] * fq
    jg.setInjection_(an, s.bus.supply.active[None, :] - pd, s.bus.supply.reactive[None, :] - qd)
    jg.powerFlow_(an)
    assert (an.status == 0).all()
    assert len(set(int(i) for i in an.method.iteration)) >= 1 and an.method.iteration.max() <= 8
    assert np.array_equal(an.voltage.magnitude[3], an.voltage.magnitude[B - 1]) and np.array_equal(an.voltage.angle[3], an.voltage.angle[B - 1])
    assert np.abs(an.voltage.magnitude[0] - an.voltage.magnitude[1]).max() > 1e-6, "the scenarios are different power flows"
    osys = oracle.OracleSystem(t)
    lanes = range(B)                                                  #  every lane, not a sample
    worst = 0.0
    for b in lanes:
        o = oracle.OracleNR(osys)
        o.set_power(osys.ps, osys.qs, pd[b], qd[b])
        assert o.power_flow() == 0
        assert an.method.iteration[b] == o.iteration, b
        vm, va = o.voltage()
        worst = max(worst, np.abs(an.voltage.magnitude[b] - vm).max(), np.abs(an.voltage.angle[b] - va).max())
    print(f"[monte carlo 512 x case1354pegase] {len(list(lanes))} lanes against the oracle: max |dV|, |dtheta| {worst:.2e}; iterations {np.bincount(an.method.iteration).tolist()}")
    assert worst <= 1e-8
    an.close()


def test_full_size_batch_properties(jg, oracle):
    """BASELINE config 5 shape on one GPU (ACTIVSg10k, 128 outage scenarios): size-independent
    properties -- every converged scenario satisfies the power-flow equations when its state is
    re-evaluated by the oracle; identical scenarios give bitwise-identical results; the base-case
    scenario equals the single-instance solution."""
    t = load_case("case_ACTIVSg10k")
    B = 128
    labels = _non_bridge_branches(t, B - 2, seed=512)
    s = jg.powerSystem(t)
    an = jg.newtonRaphson(s, batch=B)
    for sc, lab in enumerate(labels):
        jg.setOutage_(an, sc, lab)
    jg.setOutage_(an, B - 2, labels[0])           # duplicate of scenario 0
    jg.powerFlow_(an)
    assert (an.status == 0).sum() >= B - 4
    assert np.array_equal(an.voltage.magnitude[0], an.voltage.magnitude[B - 2])
    assert np.array_equal(an.voltage.angle[0], an.voltage.angle[B - 2])
    # a scenario's result does not depend on the batch it is solved in -- bitwise within a plan class (65 - 255 scenarios here: Engine::create), to
    # rounding across classes (a handle for up to 32 scenarios starts its multifrontal top lower: another summation order)
    alone = jg.newtonRaphson(jg.powerSystem(t), batch=65)
    jg.powerFlow_(alone)
    assert np.array_equal(alone.voltage.magnitude[0], an.voltage.magnitude[B - 1]) and np.array_equal(alone.voltage.angle[0], an.voltage.angle[B - 1])
    assert alone.method.iteration[0] == an.method.iteration[B - 1]
    alone.close()
    single = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(single)
    assert np.abs(single.voltage.magnitude - an.voltage.magnitude[B - 1]).max() <= 1e-12 and np.abs(single.voltage.angle - an.voltage.angle[B - 1]).max() <= 1e-12
    assert single.method.iteration == an.method.iteration[B - 1]
    osys = oracle.OracleSystem(t)
    for sc in (0, 17, 63, 64, 100, B - 1):
        if an.status[sc] != 0:
            continue
        o = oracle.OracleNR(osys)
        if sc < len(labels):
            ptr, dy = jg.outagePatch(s, labels[sc])
            for p, d in zip(ptr, dy):
                o.add_ybus(p - 1, d)
        o.set_voltage(an.voltage.magnitude[sc], an.voltage.angle[sc])
        assert max(o.mismatch()) < 1e-8


def test_config5_shape_on_the_synthetic_9241_grid(jg, oracle):
    """BASELINE config 5 (case9241pegase-shaped grid, batched N-1): a 64-scenario shard (what one of 8 GPUs gets
    of the 512) from the flat start; spot-checked scenarios equal the oracle solving that outage alone."""
    t = load_case("case9241synth")
    s = jg.powerSystem(t)
    labels = jg.outageList(s, 64, seed=512)
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    assert (an.status == 0).sum() >= 60
    osys = oracle.OracleSystem(t)
    for sc in (0, 13, 31, 63):
        o = oracle.OracleNR(osys)
        ptr, dy = jg.outagePatch(s, int(labels[sc]))
        for p, d in zip(ptr, dy):
            o.add_ybus(p - 1, d)
        st = o.power_flow(iteration=20, tolerance=1e-8)
        assert an.status[sc] == st
        if st == 0:
            assert an.method.iteration[sc] == o.iteration
            vm, va = o.voltage()
            assert np.abs(an.voltage.magnitude[sc] - vm).max() <= 1e-8
            assert np.abs(an.voltage.angle[sc] - va).max() <= 1e-8


def test_pipeline_equals_one_batch_at_a_time(jg):
    """ContingencyPipeline (several batches in flight on separate handles / streams / host threads) returns, per
    scenario, exactly what one handle solving the batches one after the other returns (bitwise)."""
    s = jg.powerSystem(load_case("case1354pegase"))
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    labels = [int(x) for x in jg.outageList(s, 150, seed=3)]          # 150 scenarios in batches of 64: 64 + 64 + 22
    pipe = jg.ContingencyPipeline(s, 64, inflight=3, start=start)
    states = {}
    jobs = [labels[i:i + 64] for i in range(0, len(labels), 64)]
    res = pipe.run(jobs, on_done=lambda j, h: states.__setitem__(j, (h._pull_voltage(), h.voltage.magnitude.copy(), h.voltage.angle.copy())))
    it, st = pipe.screen(labels)
    assert np.array_equal(it, np.concatenate([r[0][:len(j)] for r, j in zip(res, jobs)]))
    one = jg.ContingencyPipeline(s, 64, inflight=1, start=start)
    for j, job in enumerate(jobs):
        ref = one.run([job], fetch=True)[0]
        assert np.array_equal(ref[0][:len(job)], res[j][0][:len(job)]) and np.array_equal(ref[1][:len(job)], res[j][1][:len(job)])
        assert np.array_equal(one.handles[0].voltage.magnitude[:len(job)], states[j][1][:len(job)])
        assert np.array_equal(one.handles[0].voltage.angle[:len(job)], states[j][2][:len(job)])
    assert (st == 0).sum() >= len(labels) - 3
    pipe.close()
    one.close()


def test_change_of_the_slack_bus(jg):
    """test/powerFlow/analysis.jl:59-67: bus 1 becomes a generator bus, bus 3 the slack; a fresh analysis hits the same goldens."""
    s = jg.powerSystem(load_case("case30test"))
    with pytest.raises(RuntimeError):
        jg.updateBusSystem_(s, label=3, type=3)                   # bus.jl:190-195: the old slack must be reassigned first
    jg.updateBusSystem_(s, label=1, type=2)
    assert s.bus.layout.slack == 0
    with pytest.raises(RuntimeError):
        jg.newtonRaphson(s)                                       # "The slack bus is missing."
    jg.updateBusSystem_(s, label=3, type=3)
    an = jg.newtonRaphson(s)
    jg.powerFlow_(an)
    g = load_golden("case30test")
    assert an.status == 0
    assert np.abs(an.voltage.magnitude - g["newtonRaphson_voltageMagnitude"]).max() <= 1e-8
    assert np.abs(an.voltage.angle - g["newtonRaphson_voltageAngle"]).max() <= 1e-8


def test_headline_configuration_at_its_own_size(jg, oracle):
    """The bench's configuration (BASELINE configs 3 / 5): case_ACTIVSg10k, 512 outage scenarios per batch, 3 batches in flight
    through ContingencyPipeline, lanes compacted while a batch iterates.  Per scenario the pipeline returns bitwise what ONE
    handle returns for the same batch; scenarios that finish in different iterations -- among them lanes the compaction moved --
    equal the oracle solving that outage alone (iteration count, V, theta to 1e-8)."""
    t = load_case("case_ACTIVSg10k")
    s = jg.powerSystem(t)
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    base.close()
    labels = [int(x) for x in jg.outageList(s, 3 * 512, seed=512)]
    jobs = [labels[i:i + 512] for i in range(0, len(labels), 512)]
    pipe = jg.ContingencyPipeline(s, 512, inflight=3, start=start)
    got = {}
    res = pipe.run(jobs, iteration=20, tolerance=1e-8,
                   on_done=lambda j, h: got.__setitem__(j, (h._pull_voltage(), h.voltage.magnitude.copy(), h.voltage.angle.copy())))
    one = jg.ContingencyPipeline(s, 512, inflight=1, start=start)
    for j, job in enumerate(jobs):
        ref = one.run([job], fetch=True)[0]
        assert np.array_equal(ref[0], res[j][0]) and np.array_equal(ref[1], res[j][1])
        assert np.array_equal(one.handles[0].voltage.magnitude, got[j][1]) and np.array_equal(one.handles[0].voltage.angle, got[j][2])
    one.close()
    it, st = res[0]
    assert (st == 0).sum() >= 508 and len(set(it[st == 0].tolist())) >= 2
    assert sum(jg.firstIterationCounts(h)[0] for h in pipe.handles) == len(jobs), "every batch started on the base case's shared factor"
    #  EVERY scenario of the first job against the oracle solving that outage alone -- among them the late finishers, which sit in lanes
    # the compaction moved
    osys = oracle.OracleSystem(t)
    worst = 0.0
    for sc in range(512):
        o = oracle.OracleNR(osys)
        ptr, dy = jg.outagePatch(s, jobs[0][sc])
        for p, d in zip(ptr, dy):
            o.add_ybus(p - 1, d)
        o.set_voltage(*start)
        stat = o.power_flow(iteration=20, tolerance=1e-8)
        assert stat == st[sc], (sc, jobs[0][sc])
        if stat == 0:
            vm, va = o.voltage()
            assert it[sc] == o.iteration, (sc, jobs[0][sc], it[sc], o.iteration)
            worst = max(worst, np.abs(got[0][1][sc] - vm).max(), np.abs(got[0][2][sc] - va).max())
    print(f"[headline 512 x 3] all 512 scenarios of job 0 against the oracle: max |dV|, |dtheta| {worst:.2e}; iterations {np.bincount(it).tolist()}")
    assert worst <= 1e-8
    pipe.close()


def test_one_full_batch_of_the_eight_gpu_shape_against_the_oracle(jg, oracle):
    """What a rank of the 8-GPU run solves at the driver's K = 20 (contingency.deviceBatching): 640 lanes = ten shares of 64 scenarios in ONE handle -- ten
    lane groups, the balanced launch mapping, the compensated first iteration -- EVERY lane against the oracle."""
    t = load_case("case_ACTIVSg10k")
    s = jg.powerSystem(t)
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    base.close()
    labels = [int(x) for x in jg.outageList(s, 2048, seed=512)[1400:1400 + 640]]
    pipe = jg.ContingencyPipeline(s, 640, inflight=1, start=start)
    it, st = pipe.run([labels], fetch=True)[0]
    an = pipe.handles[0]
    assert jg.firstIterationCounts(an) == (1, 0)
    osys = oracle.OracleSystem(t)
    worst = 0.0
    for sc in range(640):
        o = oracle.OracleNR(osys)
        ptr, dy = jg.outagePatch(s, labels[sc])
        for p, d in zip(ptr, dy):
            o.add_ybus(p - 1, d)
        o.set_voltage(*start)
        stat = o.power_flow(iteration=20, tolerance=1e-8)
        assert stat == st[sc], (sc, labels[sc], stat, st[sc])
        if stat == 0:
            vm, va = o.voltage()
            assert it[sc] == o.iteration, (sc, labels[sc])
            worst = max(worst, np.abs(an.voltage.magnitude[sc] - vm).max(), np.abs(an.voltage.angle[sc] - va).max())
    print(f"[640-lane batch] every lane against the oracle: max |dV|, |dtheta| {worst:.2e}; iterations {np.bincount(it).tolist()}, status {np.bincount(st).tolist()}")
    assert worst <= 1e-8 and (st == 0).sum() >= 630
    pipe.close()
