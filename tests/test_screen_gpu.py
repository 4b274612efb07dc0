"""Contingency screen summary on the device (the next widening of the path; include/jgrid.h: jg_nr_screen): per scenario the worst branch
loading and its branch, the largest apparent power at a branch end and its branch, lowest / highest voltage and their buses, iterations, status -- against
the oracle's restatement of power! (src/postprocessing/acAnalysis.jl:898-904) evaluated on that scenario's own state with its branch switched off; against
the device's own power!; through a device record (the operand of the one gather of a sharded screen)."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


def _oracle_summary(oracle, t, label, vm, va, rating):
    t2 = {k: np.array(v) for k, v in t.items()}
    if label:
        t2["br_status"][label - 1] = 0
    ref = oracle.power_and_current(oracle.OracleSystem(t2), vm, va)
    s = np.maximum(np.hypot(ref["from_"][0], ref["from_"][1]), np.hypot(ref["to"][0], ref["to"][1]))
    on = np.asarray(t2["br_status"]) == 1
    s = np.where(on, s, 0.0)
    kf = int(np.argmax(s))
    load = np.where((rating > 0) & on, s / np.where(rating > 0, rating, 1.0), 0.0)
    kl = int(np.argmax(load))
    return (float(load[kl]), kl + 1 if load[kl] > 0 else 0, float(s[kf]), kf + 1, float(vm.min()), int(np.argmin(vm)) + 1, float(vm.max()), int(np.argmax(vm)) + 1)


@pytest.mark.parametrize("name,batch", [("case118", 6), ("case1354pegase", 70), ("case_ACTIVSg10k", 130)])
def test_summary_matches_the_oracle(jg, oracle, name, batch):
    t = load_case(name)
    s = jg.powerSystem(t)
    labels = [int(x) for x in jg.outageList(s, batch - 1, seed=7)] + [0]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an)
    nb = s.branch.number
    rng = np.random.Generator(np.random.PCG64(nb))
    base = jg.newtonRaphson(jg.powerSystem(t))
    jg.powerFlow_(base)
    jg.power_(base)
    flow0 = np.maximum(np.hypot(base.power.from_.active, base.power.from_.reactive), np.hypot(base.power.to.active, base.power.to.reactive))
    rating = np.where(rng.random(nb) < 0.8, (1.0 + rng.random(nb)) * np.maximum(flow0, 0.05), 0.0)     # every fifth branch has no limit
    base.close()
    out = jg.screenSummary_(an, rating=rating)
    assert np.array_equal(out.iteration, an.method.iteration) and np.array_equal(out.status, an.status)
    checked = 0
    for sc in list(range(0, batch, max(1, batch // 9))) + [batch - 1]:
        if an.status[sc] != 0:
            continue
        ref = _oracle_summary(oracle, t, labels[sc], an.voltage.magnitude[sc], an.voltage.angle[sc], rating)
        got = (out.loading[sc], out.loadingBranch[sc], out.flow[sc], out.flowBranch[sc], out.minMagnitude[sc], out.minBus[sc], out.maxMagnitude[sc], out.maxBus[sc])
        assert abs(got[0] - ref[0]) <= 1e-11 * max(1.0, ref[0]) and abs(got[2] - ref[2]) <= 1e-11 * max(1.0, ref[2]), (sc, got, ref)
        assert got[4] == ref[4] and got[6] == ref[6] and (got[5], got[7]) == (ref[5], ref[7]), (sc, got, ref)
        assert (got[1], got[3]) == (ref[1], ref[3]), (sc, got, ref)
        checked += 1
    assert checked >= 5
    # against the device's own power!: the summary is a reduction of exactly those numbers
    jg.power_(an)
    f = np.maximum(np.hypot(an.power.from_.active, an.power.from_.reactive), np.hypot(an.power.to.active, an.power.to.reactive))
    assert np.abs(out.flow - f.max(axis=1)).max() <= 1e-12 * max(1.0, f.max()) and np.array_equal(out.flowBranch, f.argmax(axis=1) + 1)
    # without ratings: no loading, everything else unchanged
    none = jg.screenSummary_(an)
    assert np.all(none.loading == 0.0) and np.all(none.loadingBranch == 0) and np.array_equal(none.flow, out.flow) and np.array_equal(none.minBus, out.minBus)
    an.close()


def test_summary_into_a_device_record(jg):
    """The record a sharded screen gathers: 10 doubles per scenario in a device buffer of the caller, bit-equal to the host record."""
    import torch
    s = jg.powerSystem(load_case("case1354pegase"))
    labels = [int(x) for x in jg.outageList(s, 200, seed=3)]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an)
    rating = np.full(s.branch.number, 2.5)
    host = jg.screenSummary_(an, rating=rating)
    rec = torch.full((200, 10), -1.0, dtype=torch.float64, device="cuda")
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    jg.screenSummary_(an, rating=rating, device_record=rec.data_ptr())
    r = rec.cpu().numpy()
    assert np.array_equal(r[:, 0], host.loading) and np.array_equal(r[:, 1], host.loadingBranch.astype(float)) and np.array_equal(r[:, 4], host.minMagnitude)
    assert np.array_equal(r[:, 8], an.method.iteration.astype(float)) and np.array_equal(r[:, 9], an.status.astype(float))
    assert (2 * s.bus.number + 2) / 10 > 250, "the gather shrinks by the ratio of the state record to the summary"
    an.close()


@pytest.mark.parametrize("name,batch,njobs,pool", [("case1354pegase", 192, 4, 128), ("case_ACTIVSg10k", 512, 3, 256)])
def test_pipeline_delivers_summary_records(jg, name, batch, njobs, pool):
    """ContingencyPipeline.run(summary=True): the record of a job is [batch, 10] screen summaries, the stragglers' rows written by the pool handle that
    finished them (jg_nr_screen_rows_device) -- bitwise what the lockstep pipeline and a plain batch + screenSummary_ give."""
    import torch
    t = load_case(name)
    s = jg.powerSystem(t)
    base = jg.newtonRaphson(s)
    jg.powerFlow_(base)
    start = (base.voltage.magnitude.copy(), base.voltage.angle.copy())
    jg.power_(base)
    flow0 = np.maximum(np.hypot(base.power.from_.active, base.power.from_.reactive), np.hypot(base.power.to.active, base.power.to.reactive))
    rating = np.maximum(1.1 * flow0, 0.05)
    base.close()
    labels = jg.outageList(s, batch * njobs, seed=5)
    jobs = [labels[i * batch:(i + 1) * batch] for i in range(njobs)]
    out = {}
    for mode in ("lockstep", "pool"):
        pipe = jg.ContingencyPipeline(s, batch, inflight=3, start=start, pool=pool if mode == "pool" else 0)
        pipe.setRating(rating)
        ring = 4 if mode == "pool" else njobs
        rec = [torch.zeros((batch, 10), dtype=torch.float64, device="cuda") for _ in range(ring)]
        torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
        seen = []

        def on_done(j, an, rec=rec, ring=ring, seen=seen):
            seen.append(rec[j % ring].clone())
            torch.cuda.current_stream().synchronize()

        res = pipe.run(jobs, iteration=20, tolerance=1e-8, on_done=on_done, record=lambda j: rec[j % ring].data_ptr(), records=ring, summary=True)
        out[mode] = (res, [r.cpu().numpy() for r in seen])
        pipe.close()
    moved = 0
    for j in range(njobs):
        ra, rb = out["lockstep"][1][j], out["pool"][1][j]
        assert np.array_equal(ra, rb)
        it, st = out["pool"][0][j]
        assert np.array_equal(rb[:, 8], it.astype(float)) and np.array_equal(rb[:, 9], st.astype(float)) and not np.any(st == 4)
        moved += int(np.sum(it > np.median(it)))
    assert moved > 0
    # job 0 as a plain batch from the case's own start point: the converged states agree to the solver's tolerance, so do the summaries
    an = jg.contingencyAnalysis(s, jobs[0])
    jg.powerFlow_(an)
    ref = jg.screenSummary_(an, rating=rating)
    r0 = out["pool"][1][0]
    ok = ref.status == 0
    assert np.array_equal(r0[:, 9], ref.status.astype(float))
    assert np.abs(r0[:, 0][ok] - ref.loading[ok]).max() <= 1e-6 * max(1.0, ref.loading[ok].max())
    assert np.abs(r0[:, 2][ok] - ref.flow[ok]).max() <= 1e-6 * max(1.0, ref.flow[ok].max())
    assert np.abs(r0[:, 4][ok] - ref.minMagnitude[ok]).max() <= 1e-6 and np.abs(r0[:, 6][ok] - ref.maxMagnitude[ok]).max() <= 1e-6
    an.close()


def test_summary_of_a_failed_scenario_is_not_the_safest_one(jg):
    """ A scenario that ends with status 3 has NaN voltages: every comparison of the reduction is false and the sentinels (loading 0, lowest voltage
    1e300) would read as 'nothing overloaded'.  Its summary is NaN with index 0 (include/jgrid.h), its neighbours in the batch are untouched."""
    from test_guard_gpu import _island_bridges
    s = jg.powerSystem(load_case("case1354pegase"))
    bad = _island_bridges(jg, s, 3)[:1]
    assert bad
    good = [int(x) for x in jg.outageList(s, 4, seed=11)]
    labels = good[:2] + bad + good[2:]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an)
    assert an.status[2] == 3 and all(an.status[i] == 0 for i in (0, 1, 3, 4))
    rating = np.full(s.branch.number, 5.0)
    out = jg.screenSummary_(an, rating=rating)
    assert np.isnan(out.loading[2]) and np.isnan(out.flow[2]) and np.isnan(out.minMagnitude[2]) and np.isnan(out.maxMagnitude[2])
    assert (out.loadingBranch[2], out.flowBranch[2], out.minBus[2], out.maxBus[2]) == (0, 0, 0, 0) and out.status[2] == 3
    keep = [0, 1, 3, 4]
    ref = jg.contingencyAnalysis(s, [labels[i] for i in keep])
    jg.powerFlow_(ref)
    r = jg.screenSummary_(ref, rating=rating)
    for name in ("loading", "loadingBranch", "flow", "flowBranch", "minMagnitude", "minBus", "maxMagnitude", "maxBus"):
        assert np.array_equal(getattr(out, name)[keep], getattr(r, name)), name
    an.close(); ref.close()


def test_summary_follows_a_branch_table_that_grew(jg):
    """ addBranch!(analysis) on an unchanged Ybus pattern re-uploads a LONGER branch table: the partial maxima of the screen are sized by the
    branch count (chunks of 256 branches) and an installed rating has the old length.  186 -> 257 branches crosses a chunk boundary: the summary must be the
    one of a fresh analysis of the grown system, and a rating of the old length must be gone, not read out of bounds."""
    s = jg.powerSystem(load_case("case118"))
    labels = [int(x) for x in jg.outageList(s, 5, seed=2)] + [0]
    an = jg.contingencyAnalysis(s, labels)
    jg.powerFlow_(an)
    nb0 = s.branch.number
    first = jg.screenSummary_(an, rating=np.full(nb0, 3.0))
    assert np.all(first.loading > 0)
    inv = {v: k for k, v in s.bus.label.items()}
    fr, to = s.branch.layout.from_, s.branch.layout.to
    rev = s.model.revision.acPattern
    k = 0
    while s.branch.number <= 256:                                # parallel circuits: the Ybus pattern stays, the branch table grows past a chunk of 256
        jg.addBranch_(an, from_=inv[int(fr[k])], to=inv[int(to[k])], resistance=0.02, reactance=0.4)
        k += 1
    assert s.model.revision.acPattern == rev and s.branch.number == 257
    jg.powerFlow_(an)
    assert np.all(an.status == 0)
    norating = jg.screenSummary_(an, rating=None)                # the old rating (186 values) went with the old table
    assert np.all(norating.loading == 0.0) and np.all(norating.loadingBranch == 0)
    rating = np.full(s.branch.number, 3.0)
    out = jg.screenSummary_(an, rating=rating)
    jg.power_(an)
    f = np.maximum(np.hypot(an.power.from_.active, an.power.from_.reactive), np.hypot(an.power.to.active, an.power.to.reactive))
    assert np.abs(out.flow - f.max(axis=1)).max() <= 1e-12 * max(1.0, f.max()) and np.array_equal(out.flowBranch, f.argmax(axis=1) + 1)
    assert np.abs(out.loading - f.max(axis=1) / 3.0).max() <= 1e-12 and np.array_equal(out.loadingBranch, out.flowBranch)
    an.close()
