"""Sharded Monte-Carlo state estimation (include/jgrid.h: jg_gn_get_objective, jg_gn_pack_results_device, jg_gn_allgather_results; montecarlo.py):
the result record of a Gauss-Newton batch -- magnitude | angle | iterations | status | objective per realisation -- is BITWISE what the getters
return, its objective is the oracle's se.objective (src/backend/equations.jl:689-698, correlated PMU pairs included), the 1-rank RCCL gather through the
C ABI reproduces the packed record, and a MonteCarloPipeline delivers per job exactly what a plain batch with the same seed computes."""
import numpy as np
import pytest

from conftest import load_case
from test_oracle_se import se_case14
from test_se_gpu import _all_families, _mirror, _system_like

pytestmark = pytest.mark.gpu


def _config4_like(jg, case="case1354pegase"):
    s = jg.powerSystem(load_case(case))
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, tolerance=1e-11)
    assert pf.status == 0
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf, variance=1e-4)
    jg.addWattmeter_(mon, pf, variance=1e-4)
    jg.addVarmeter_(mon, pf, variance=1e-4)
    jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
    pf.close()
    return s, mon


@pytest.mark.parametrize("pmu_kw", [dict(), dict(correlated=True)])
def test_objective_on_the_device_is_the_oracles(jg, oracle, pmu_kw):
    """se.objective after increment! (one scenario, every meter family, correlated rectangular PMUs: the cross term 2 r_a r_b W_ab of the pair's second row)."""
    t, osys, vm, va = se_case14(oracle)
    tab = _all_families(oracle, osys, vm, va, pmu_kw)
    an = jg.gaussNewton(_mirror(jg, _system_like(jg, t, osys), tab))
    gn = oracle.OracleGN(osys, tab)
    for _ in range(3):
        jg.incrementSE_(an); gn.increment()
        assert abs(an.objectiveDevice() - gn.objective) <= 1e-10 * max(1.0, gn.objective)
        assert abs(an.objectiveDevice() - an.objective) <= 1e-12 * max(1.0, an.objective)
        jg.solveSE_(an); gn.solve()
    an.close()


@pytest.mark.parametrize("batch", [70, 192])
def test_record_is_bitwise_the_getters(jg, batch):
    import torch
    s, mon = _config4_like(jg)
    n = s.bus.number
    an = jg.gaussNewton(mon, batch=batch)
    with pytest.raises(jg._lib.JGridError):                       # no run yet: iterations / status would be undefined
        an.pack_results_device(torch.zeros(1, dtype=torch.float64, device="cuda").data_ptr())
    jg.setNoise_(an, np.random.Generator(np.random.PCG64(4)), scale=1.0)
    an.setVoltage(np.ones(n), np.zeros(n))
    jg.stateEstimation_(an, iteration=40, tolerance=1e-8)
    assert np.all(an.status == 0) and an.method.iteration.min() >= 3
    rec = torch.full((batch, 2 * n + 3), np.nan, dtype=torch.float64, device="cuda")
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    an.pack_results_device(rec.data_ptr())
    r = rec.cpu().numpy()
    it, st, obj, vm, va = (x.cpu().numpy() for x in jg.unpackEstimates(rec))
    assert np.array_equal(vm, an.voltage.magnitude) and np.array_equal(va, an.voltage.angle)
    assert np.array_equal(it, an.method.iteration) and np.array_equal(st, an.status)
    assert np.array_equal(r[:, 2 * n + 2], an.objectiveDevice())
    host = an.objective                                           # r' W r from the pulled residuals, summed on the host
    assert np.abs(obj - host).max() <= 1e-12 * host.max()
    # noisy readings: the objective of a converged WLS estimate is chi-square with m - (2 n - 1) degrees of freedom (badData.jl:948-961 tests exactly that)
    dof = an.dims["m"] - (2 * n - 1)
    assert 0.8 * dof < obj.mean() < 1.2 * dof
    # bitwise run to run (fixed summation order)
    rec2 = torch.zeros_like(rec)
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    an.pack_results_device(rec2.data_ptr())
    assert torch.equal(rec, rec2)
    # the gather through the C ABI with a 1-rank RCCL communicator
    comm = jg._lib.Comm(0, 1, jg._lib.Comm.unique_id(), device=0)
    out = torch.full_like(rec, np.nan)
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    jg.gatherEstimatesDevice(an, comm, out.data_ptr())
    torch.cuda.current_stream().synchronize()
    assert torch.equal(out, rec)
    comm.close()
    an.close()


def test_pipeline_delivers_the_records_of_plain_batches(jg):
    """Four jobs (seeds) on two handles with a ring of two device records: every job's record is bitwise the record of a plain batch that drew the same
    realisations; on_done sees the jobs in order."""
    import torch
    s, mon = _config4_like(jg)
    n, B = s.bus.number, 128
    pipe = jg.MonteCarloPipeline(mon, B, inflight=2)
    assert pipe.record_width == 2 * n + 3
    ring = [torch.zeros((B, 2 * n + 3), dtype=torch.float64, device="cuda") for _ in range(2)]
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    seen, order = [], []

    def on_done(j, an):
        order.append(j)
        seen.append(ring[j % 2].clone())
        torch.cuda.current_stream().synchronize()

    seeds = [11, 12, 13, 14]
    res = pipe.run(seeds, iteration=40, tolerance=1e-8, on_done=on_done, record=lambda j: ring[j % 2].data_ptr(), records=2)
    assert order == [0, 1, 2, 3]
    pipe.close()
    ref = jg.gaussNewton(mon, batch=B)
    for j, seed in enumerate(seeds):
        jg.drawNoise_(ref, seed)                                  # the pipeline draws its realisations on the device (MonteCarloPipeline(host_noise=False))
        ref.setVoltage(np.ones(n), np.zeros(n))
        jg.stateEstimation_(ref, iteration=40, tolerance=1e-8)
        rec = torch.zeros((B, 2 * n + 3), dtype=torch.float64, device="cuda")
        torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
        ref.pack_results_device(rec.data_ptr())
        assert torch.equal(rec, seen[j]), j
        assert np.array_equal(res[j][0], ref.method.iteration) and np.array_equal(res[j][1], ref.status)
    assert not torch.equal(seen[0], seen[1])                      # different seeds, different realisations
    ref.close()


# ---- realisations drawn on the device (jg_gn_set_readings / jg_gn_draw_noise) against a numpy restatement of the generator + the host's value rules ----------
_GOLD, _LANE = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03


def _mix64(z):
    z = z ^ (z >> np.uint64(30)); z = z * np.uint64(0xBF58476D1CE4E5B9)
    z = z ^ (z >> np.uint64(27)); z = z * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _normals(seed, ndev, first, count):
    """(e1, e2) [count, ndev]: the generator of k_gn_noise (csrc/jg_gn.hip) in numpy uint64 arithmetic: splitmix64 finaliser on a (seed, device, realisation) counter, Box-Muller."""
    with np.errstate(over="ignore"):
        d = np.arange(ndev, dtype=np.uint64)[None, :]
        r = (np.uint64(first) + np.arange(count, dtype=np.uint64))[:, None]
        ctr = (np.uint64(seed) + np.uint64(_GOLD) * (np.uint64(2) * d)) ^ (r * np.uint64(_LANE))
        u1 = ((_mix64(ctr) >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * 2.0 ** -53
        u2 = ((_mix64(ctr + np.uint64(_GOLD)) >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * 2.0 ** -53
    rad = np.sqrt(-2.0 * np.log(u1))
    return rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)


def _every_value_rule(jg, oracle):
    t, osys, vm, va = se_case14(oracle)
    tab = oracle.MeterTable()
    for fam, kw in (("voltmeter", {}), ("ammeter", {}), ("ammeter", dict(square=True)), ("wattmeter", {}), ("varmeter", {}),
                    ("pmu", {}), ("pmu", dict(polar=True)), ("pmu", dict(polar=True, square=True)), ("pmu", dict(correlated=True))):
        oracle.add_from_power_flow(tab, osys, vm, va, fam, **kw)
    return _mirror(jg, _system_like(jg, t, osys), tab)


def test_noise_drawn_on_the_device_follows_the_generator_and_the_value_rules(jg, oracle):
    """Every kind of value rule (plain, squared current, polar PMU with plain / squared magnitude, rectangular PMU with and without its 2x2 precision block):
    the device's se.mean / se.precision of 70 realisations against the numpy restatement of its generator pushed through the HOST's rules (stateestimation._wls_values,
    the restatement of acWLS :135-236) -- 1e-12; realisation r of a seed is the same numbers at any lane of any batch (what lets the ranks of a sharded study draw their own
    shares); scale 0 is the noise-free set."""
    mon = _every_value_rule(jg, oracle)
    B = 70
    an = jg.gaussNewton(mon, batch=B)
    base = jg.measurementDevice(an)
    z1, v1, s1, z2, v2, s2 = an._z
    kinds = set()
    jg.drawNoise_(an, 20251001, scale=1.0, first=5)
    mean, wd, wo = jg.measurementDevice(an)
    e1, e2 = _normals(20251001, z1.size, 5, B)
    n1, n2 = z1[None, :] + np.sqrt(v1)[None, :] * e1, z2[None, :] + np.sqrt(v2)[None, :] * e2
    rm, rw, ro, _ = an._values(an.monitoring, an._devs, an._dev_row, an.dims["m"], n1, v1, s1, n2, v2, s2)
    assert ro.shape == wo.shape and wo.shape[1] > 0
    for got, ref in ((mean, rm), (wd, rw), (wo, ro)):
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), np.abs(got - ref).max()
    assert np.abs(mean - base[0]).max() > 1e-4                    # it IS noisy
    # the same realisations at other lanes of another batch
    other = jg.gaussNewton(mon, batch=16)
    jg.drawNoise_(other, 20251001, scale=1.0, first=5 + 30)
    m2, w2, o2 = jg.measurementDevice(other)
    assert np.array_equal(m2, mean[30:46]) and np.array_equal(w2, wd[30:46]) and np.array_equal(o2, wo[30:46])
    jg.drawNoise_(other, 20251002, scale=1.0, first=5 + 30)       # another seed: other numbers
    assert not np.array_equal(jg.measurementDevice(other)[0], m2)
    other.close()
    jg.drawNoise_(an, 1, scale=0.0)
    back = jg.measurementDevice(an)
    for got, ref in zip(back, base):                              # (the precision block of a correlated PMU is a difference of close numbers: sincos of the device and of numpy
        assert np.all(np.abs(got - ref) <= 1e-11 * np.maximum(1.0, np.abs(ref)))   # differ in the last bit and that comes out at 1e-13 relative)
    an.close()


def test_device_noise_is_standard_normal_and_the_estimates_are_consistent(jg):
    """512 realisations of the config-4-shaped set of case1354pegase drawn on the device: the standardised deviations (mean - z) / sigma of the plain rows have mean 0 and
    variance 1 (6.6e6 samples: 5 sigma of the sample moments), and the estimation they feed converges with a chi-square objective -- the statistics of a sound generator
    (badData.jl:948-961 tests se.objective against exactly that distribution)."""
    s, mon = _config4_like(jg)
    n, B = s.bus.number, 512
    an = jg.gaussNewton(mon, batch=B)
    base, wbase, _ = jg.measurementDevice(an)
    jg.drawNoise_(an, 7, scale=1.0)
    mean, wd, _ = jg.measurementDevice(an)
    plain = np.flatnonzero(np.isin(an.method._code, (1, 6, 7, 8, 9, 10, 11)) & (an.method.type != 0))
    dev = (mean[:, plain] - base[:, plain]) * np.sqrt(wbase[:, plain])
    N = dev.size
    assert abs(dev.mean()) < 5.0 / np.sqrt(N) and abs(dev.var() - 1.0) < 5.0 * np.sqrt(2.0 / N)
    assert abs(np.mean(dev ** 3)) < 5.0 * np.sqrt(6.0 / N) and abs(np.mean(dev ** 4) - 3.0) < 5.0 * np.sqrt(96.0 / N)
    c = np.corrcoef(dev[:, :200].T)                               # rows are independent of each other ...
    assert np.abs(c - np.eye(200)).max() < 6.0 / np.sqrt(B)
    c = np.corrcoef(dev[:200, :2000])                             # ... and realisations of each other
    assert np.abs(c - np.eye(200)).max() < 6.0 / np.sqrt(2000)
    an.setVoltage(np.ones(n), np.zeros(n))
    jg.stateEstimation_(an, iteration=40, tolerance=1e-8, fetch=False)
    assert np.all(an.status == 0)
    dof = an.dims["m"] - (2 * n - 1)
    obj = an.objectiveDevice()
    assert abs(obj.mean() - dof) < 5.0 * np.sqrt(2.0 * dof / B), (obj.mean(), dof)
    an.close()


def test_a_failing_on_done_surfaces_and_leaves_no_worker_waiting(jg):
    """The caller's callback raises in the middle of a Monte-Carlo run: run() re-raises that error once its workers have ended, and works afterwards."""
    import threading
    s, mon = _config4_like(jg, "case118")
    pipe = jg.MonteCarloPipeline(mon, 32, inflight=2)
    box = {}

    def boom(j, h):
        if j == 1:
            raise RuntimeError("caller failed on job 1")

    def go():
        try:
            pipe.run(list(range(6)), on_done=boom)
        except BaseException as e:
            box["e"] = e
    t = threading.Thread(target=go, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "run() hangs after a failing on_done"
    assert isinstance(box.get("e"), RuntimeError) and "job 1" in str(box["e"])
    res = pipe.run([7, 8])
    assert all(int((st == 0).sum()) == 32 for _, st in res)
    pipe.close()


def test_device_noise_follows_a_meter_updated_after_the_first_draw(jg):
    """updateVoltmeter!(analysis; ...) after realisations were drawn on the device: the next draw starts from the NEW reading, variance and status (the raw
    readings the device holds are refreshed with the analysis; scale 0 makes the draw the reading itself)."""
    s, mon = _config4_like(jg, "case118")
    an = jg.gaussNewton(mon, batch=8)
    jg.drawNoise_(an, 3, scale=1.0)
    jg.drawNoise_(an, 3, scale=0.0)
    base, wbase, _ = jg.measurementDevice(an)
    jg.updateVoltmeter_(an, label=5, magnitude=1.0321, variance=4e-4)
    jg.updateWattmeter_(an, label=2, status=0)
    jg.drawNoise_(an, 3, scale=0.0)
    mean, wd, _ = jg.measurementDevice(an)
    rv = int(an._dev_row[4])                                            # voltmeters come first: device 4 = voltmeter 5
    assert np.all(mean[:, rv] == 1.0321) and np.allclose(wd[:, rv], 1.0 / 4e-4, rtol=1e-15)
    nv = mon.voltmeter.number + mon.ammeter.number
    rw = int(an._dev_row[nv + 1])                                       # wattmeter 2
    assert np.all(mean[:, rw] == 0.0) and np.all(base[:, rw] != 0.0)
    keep = np.ones(mean.shape[1], dtype=bool); keep[[rv, rw]] = False
    assert np.array_equal(mean[:, keep], base[:, keep]) and np.array_equal(wd[:, keep], wbase[:, keep])
    an.close()


def test_host_mirrors_follow_realisations_drawn_on_the_device(jg, oracle):
    """ drawNoise_ rewrites se.mean and se.precision on the device only.  Whoever reads the host mirrors afterwards must see the lanes' own realisations:
    `objective` (host sum with the precision mirror: squared currents and rectangular PMUs carry z-dependent weights) equals the device's own reduction, a residual test
    keeps every lane's noisy readings apart from the rows it removes, and a later draw does not revive a removed row."""
    mon = _every_value_rule(jg, oracle)
    B = 6
    an = jg.gaussNewton(mon, batch=B)
    jg.drawNoise_(an, 99, scale=1.0)
    noisy = jg.measurementDevice(an)
    jg.incrementSE_(an)
    od, oh = an.objectiveDevice(), an.objective
    assert np.all(np.abs(od - oh) <= 1e-10 * np.maximum(1.0, np.abs(od))), (od, oh)
    assert np.abs(np.atleast_2d(an.precision) - np.diag(np.diag(np.atleast_2d(an.precision)))).max() > 0.0      # (correlated pairs present)
    out = jg.residualTest_(an, threshold=0.0)                     # every scenario loses the row of its largest normalised residual
    assert out.detect.all()
    mean, wd, wo = jg.measurementDevice(an)
    gone = wd == 0.0
    assert (gone.sum(axis=1) >= (noisy[1] == 0.0).sum(axis=1) + 1).all() and (gone.sum(axis=1) <= (noisy[1] == 0.0).sum(axis=1) + 2).all()
    keep = ~gone
    assert np.array_equal(mean[keep], noisy[0][keep]) and np.array_equal(wd[keep], noisy[1][keep]), "the other rows keep the lane's own realisation"
    assert not np.array_equal(mean[0][keep[0] & keep[1]], mean[1][keep[0] & keep[1]]), "lanes differ: nothing was reset to the shared base set"
    jg.drawNoise_(an, 100, scale=1.0)                             # new realisations: the removed rows stay out of the model
    m2, w2, _ = jg.measurementDevice(an)
    assert np.all(w2[gone] == 0.0) and np.all(m2[gone] == 0.0) and not np.array_equal(m2[keep], mean[keep])
    an.close()
