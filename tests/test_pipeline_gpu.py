"""ContingencyPipeline with a straggler pool (jg_nr_run_defer / jg_nr_move_lanes / jg_nr_resume / jg_nr_pack_rows_device):
every scenario's result -- iteration count, status, V, theta -- is BITWISE what the lockstep pipeline gives; scenarios without a
power flow and scenarios that never leave their batch are covered."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def _records(torch, count, lanes, n):
    ring = [torch.full((lanes, 2 * n + 2), -7.0, dtype=torch.float64, device="cuda") for _ in range(count)]
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    return ring


@pytest.mark.parametrize("name,batch,njobs,pool", [("case1354pegase", 192, 7, 128), ("case1354pegase", 128, 5, 64),
                                                  ("case_ACTIVSg10k", 512, 4, 256),
                                                  ("case1354pegase", 512, 3, 128),       # pool request on the other side of the 256-lane plan
                                                  ("case1354pegase", 320, 3, 64)])       # threshold than its batches: the pipeline moves it over
def test_pool_is_bitwise_the_lockstep_pipeline(jg, name, batch, njobs, pool):
    import torch
    s = jg.powerSystem(load_case(name))
    n = s.bus.number
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    labels = jg.outageList(s, batch * njobs, seed=3)
    jobs = [labels[i * batch:(i + 1) * batch] for i in range(njobs)]
    jobs[-1] = jobs[-1][:batch - 5]                                   # a ragged last job (padded with base-case lanes)
    out = {}
    for mode in ("lockstep", "pool"):
        pipe = jg.ContingencyPipeline(s, batch, inflight=3, start=start, pool=pool if mode == "pool" else 0)
        assert bool(pipe.pools) == (mode == "pool")
        ring = 4 if mode == "pool" else njobs
        rec = _records(torch, ring, batch, n)
        seen = []

        def on_done(j, an, rec=rec, ring=ring, seen=seen):
            seen.append((j, rec[j % ring].clone()))        # the library synchronised its own stream after writing the record
            torch.cuda.current_stream().synchronize()      # (no device-wide sync: another worker may be capturing its graphs)

        res = pipe.run(jobs, iteration=20, tolerance=1e-8, on_done=on_done, record=lambda j: rec[j % ring].data_ptr(), records=ring)
        assert [j for j, _ in seen] == list(range(njobs))
        out[mode] = (res, [r.cpu().numpy() for _, r in seen])
        pipe.close()
    moved = 0
    for j in range(njobs):
        (it_a, st_a), (it_b, st_b) = out["lockstep"][0][j], out["pool"][0][j]
        assert np.array_equal(it_a, it_b) and np.array_equal(st_a, st_b)
        assert not np.any(st_b == 4)                                  # nothing is left "deferred"
        ra, rb = out["lockstep"][1][j], out["pool"][1][j]
        assert np.array_equal(ra, rb)                                 # V | theta | iterations | status, bitwise
        assert np.array_equal(rb[:, 2 * n], it_b.astype(float)) and np.array_equal(rb[:, 2 * n + 1], st_b.astype(float))
        moved += int(np.sum(it_b > np.median(it_b)))
    assert moved > 0                                                  # some scenarios did need more iterations than their batch


def test_pool_of_a_mid_size_batch_keeps_its_plan_class(jg):
    """batches of 128 / 192 lanes with defer_at <= 32 and a small pool request used to get a 64-lane pool, which on a grid of
    4 000 buses and more runs the one-lane-group plan (another summation order): stragglers were no longer bitwise the lockstep scenarios, with
    no error.  The pool is now clamped into the class of its batches, and jg_nr_move_lanes refuses a hand-off between different plans."""
    import torch
    s = jg.powerSystem(load_case("case_ACTIVSg10k"))
    n = s.bus.number
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    base.close()
    labels = jg.outageList(s, 2 * 192, seed=11)
    jobs = [labels[:192], labels[192:]]
    out = {}
    for mode, pool in (("lockstep", 0), ("pool", 64)):
        pipe = jg.ContingencyPipeline(s, 192, inflight=2, start=start, pool=pool, defer_at=32)
        if pool:
            assert pipe.pools and 128 <= pipe.pools[0].handle.batch <= 192, "a pool beside 192-lane batches stays in their class"
        rec = _records(torch, 2, 192, n)
        res = pipe.run(jobs, iteration=20, tolerance=1e-8, on_done=lambda j, an: torch.cuda.current_stream().synchronize(),
                       record=lambda j: rec[j % 2].data_ptr(), records=2)
        out[mode] = (res, [r.cpu().numpy().copy() for r in rec])
        pipe.close()
    for j in range(2):
        assert np.array_equal(out["lockstep"][0][j][0], out["pool"][0][j][0]) and np.array_equal(out["lockstep"][1][j], out["pool"][1][j])
    # the refusal itself: a paused 192-lane batch and a 64-lane handle of the same grid
    big = jg.contingencyAnalysis(s, labels[:192])
    small = jg.newtonRaphson(s, batch=64, max_patch=4)
    left = jg._lib.C.c_int32(0)
    jg._lib.check(jg._lib.lib().jg_nr_run_defer(big._h, 20, 1e-8, 64, jg._lib.C.byref(left)))
    if left.value > 0:
        home = np.zeros(64, dtype=np.int32)
        cnt = jg._lib.C.c_int32(0)
        rc = jg._lib.lib().jg_nr_move_lanes(small._h, 0, big._h, home, jg._lib.C.byref(cnt))
        assert rc == 1 and b"another factorisation plan" in jg._lib.lib().jg_last_error()
    big.close(); small.close()


def test_pool_handles_scenarios_without_a_power_flow(jg):
    """An islanding outage (status 3) and a scenario that runs into the iteration limit end with the right status whether they
    finish in their batch or in the pool; screen() works with a pool (no records)."""
    s = jg.powerSystem(load_case("case1354pegase"))
    good = [int(x) for x in jg.outageList(s, 300, seed=5)]
    br = np.flatnonzero(jg.bridges(s) & (s.branch.layout.status == 1)) + 1
    labels = good[:100] + [int(br[0])] + good[100:200] + [int(br[1])] + good[200:]
    a = jg.ContingencyPipeline(s, 128, inflight=2)
    b = jg.ContingencyPipeline(s, 128, inflight=2, pool=128)
    it_a, st_a = a.screen(labels, iteration=6, tolerance=1e-8)
    it_b, st_b = b.screen(labels, iteration=6, tolerance=1e-8)
    a.close(); b.close()
    assert np.array_equal(it_a, it_b) and np.array_equal(st_a, st_b)
    assert set(np.unique(st_a)) <= {0, 1, 3} and st_a[100] != 0 and st_a[201] != 0


def test_bench_default_pipeline_ends_at_the_oracle(jg, oracle):
    """The configuration bench.py times by default -- case_ACTIVSg10k, 512 outage scenarios per batch, THREE batches in flight, a straggler
    pool of 256 lanes, results delivered as packed device records -- checked straight against the oracle (it was covered
    transitively only, pool == lockstep and lockstep == oracle).  Picked from the packed record of a job: every scenario that finished in
    the pool (it needed more iterations than its batch ran in lockstep), and a handful that never left their batch."""
    import torch
    t = load_case("case_ACTIVSg10k")
    s = jg.powerSystem(t)
    n = s.bus.number
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    base.close()
    labels = [int(x) for x in jg.outageList(s, 4 * 512, seed=512)]
    jobs = [labels[i:i + 512] for i in range(0, len(labels), 512)]
    pipe = jg.ContingencyPipeline(s, 512, inflight=3, start=start, pool=256)
    assert len(pipe.handles) == 3 and pipe.pools
    ring = 4
    rec = _records(torch, ring, 512, n)
    seen = {}

    def on_done(j, an):
        seen[j] = rec[j % ring].clone()
        torch.cuda.current_stream().synchronize()

    res = pipe.run(jobs, iteration=20, tolerance=1e-8, on_done=on_done, record=lambda j: rec[j % ring].data_ptr(), records=ring)
    pipe.close()
    osys = oracle.OracleSystem(t)
    checked_pool = 0
    for j in (0, 3):                                                  # the first job and the last one (its stragglers empty the pool at the end)
        r = seen[j].cpu().numpy()
        it, st = res[j]
        assert np.array_equal(r[:, 2 * n], it.astype(float)) and np.array_equal(r[:, 2 * n + 1], st.astype(float))
        ok = st == 0
        assert ok.sum() >= 500
        lock = int(np.median(it[ok]))                                 # what the batch ran in lockstep: later finishers went through the pool
        late = np.flatnonzero(ok & (it > lock))
        assert late.size > 0
        picks = sorted(set([int(x) for x in late[:6]] + [0, 77, 300, 511]))
        for sc in picks:
            o = oracle.OracleNR(osys)
            ptr, dy = jg.outagePatch(s, jobs[j][sc])
            for p, d in zip(ptr, dy):
                o.add_ybus(p - 1, d)
            o.set_voltage(*start)
            stat = o.power_flow(iteration=20, tolerance=1e-8)
            assert stat == st[sc]
            if stat == 0:
                vm, va = o.voltage()
                assert it[sc] == o.iteration
                assert np.abs(r[sc, :n] - vm).max() <= 1e-8 and np.abs(r[sc, n:2 * n] - va).max() <= 1e-8
                checked_pool += int(it[sc] > lock)
    assert checked_pool >= 4


@pytest.mark.parametrize("shared_first", [False, True])
def test_monte_carlo_injection_jobs_through_the_pipeline(jg, oracle, shared_first):
    """Jobs that carry per-scenario injections (load variations: the Monte-Carlo instances of the north star) instead of outage labels: three jobs on two handles
    with a straggler pool and a ring of device records; every job's record is bitwise the record of a plain batch with the same injections, samples agree with the oracle.
    shared_first (round 6): the pipeline's start is a base case and the first iteration of every job is a sweep pair on its shared factor (scenarios without Ybus edits:
    J_s = J_0 whatever their injections) -- the record then equals the refactorising batch's to rounding, iteration counts exactly."""
    import torch
    t = load_case("case1354pegase")
    s = jg.powerSystem(t)
    n, B = s.bus.number, 192
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    base.close()
    rng = np.random.default_rng(17)
    jobs = []
    for _ in range(3):
        scale = 1.0 + 0.05 * rng.standard_normal((B, 1))
        jobs.append({"active": s.bus.supply.active[None, :] - s.bus.demand.active[None, :] * scale,
                     "reactive": s.bus.supply.reactive[None, :] - s.bus.demand.reactive[None, :] * scale})
    pipe = jg.ContingencyPipeline(s, B, inflight=2, start=start, pool=128, shared_first=shared_first)
    ring = [torch.zeros((B, 2 * n + 2), dtype=torch.float64, device="cuda") for _ in range(3)]
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    seen = []

    def on_done(j, an):
        seen.append(ring[j % 3].clone())
        torch.cuda.current_stream().synchronize()

    res = pipe.run(jobs, on_done=on_done, record=lambda j: ring[j % 3].data_ptr(), records=3)
    started = [jg.firstIterationCounts(h) for h in pipe.handles]
    assert sum(c[0] for c in started) == (len(jobs) if shared_first else 0) and sum(c[1] for c in started) == (0 if shared_first else len(jobs))
    pipe.close()
    ref = jg.newtonRaphson(s, batch=B, max_patch=4)
    for j, job in enumerate(jobs):
        jg.setInjection_(ref, job["active"], job["reactive"])
        jg.powerflow._push_voltage(ref, *start)
        jg.powerFlow_(ref)
        rec = torch.zeros((B, 2 * n + 2), dtype=torch.float64, device="cuda")
        torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
        ref.pack_results_device(rec.data_ptr())
        if shared_first:
            assert torch.equal(rec[:, 2 * n:], seen[j][:, 2 * n:]) and float((rec[:, :2 * n] - seen[j][:, :2 * n]).abs().max()) <= 1e-10, j
        else:
            assert torch.equal(rec, seen[j]), j
        assert np.array_equal(res[j][0], ref.method.iteration) and np.all(res[j][1] == 0)
    osys = oracle.OracleSystem(t)
    o = oracle.OracleNR(osys)
    o.set_power(osys.ps, osys.qs, s.bus.supply.active - jobs[2]["active"][5], s.bus.supply.reactive - jobs[2]["reactive"][5])
    o.set_voltage(start[0], start[1])
    assert o.power_flow() == 0 and o.iteration == ref.method.iteration[5]
    vm, va = o.voltage()
    assert np.abs(ref.voltage.magnitude[5] - vm).max() < 1e-8 and np.abs(ref.voltage.angle[5] - va).max() < 1e-8
    ref.close()


def test_a_failing_on_done_surfaces_and_leaves_no_worker_waiting(jg):
    """The caller's own callback raises in the middle of a run: run() re-raises THAT error after its workers have ended (none of them may keep waiting
    for a delivery that never comes), and the pipeline is usable afterwards."""
    import threading
    s = jg.powerSystem(load_case("case118"))
    pipe = jg.ContingencyPipeline(s, 64, inflight=3)
    labels = [int(x) for x in jg.outageList(s, 64 * 6, seed=5)]
    jobs = [labels[i:i + 64] for i in range(0, len(labels), 64)]
    before = threading.active_count()

    def boom(j, h):
        if j == 2:
            raise RuntimeError("caller failed on job 2")
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("e", _raises(lambda: pipe.run(jobs, on_done=boom))), daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "run() hangs after a failing on_done"
    assert isinstance(box["e"], RuntimeError) and "job 2" in str(box["e"])
    assert threading.active_count() <= before + 1
    res = pipe.run(jobs[:2])
    assert all(int((r[1] == 0).sum()) >= 60 for r in res)
    pipe.close()


def _raises(f):
    try:
        f()
    except BaseException as e:
        return e
    return None

