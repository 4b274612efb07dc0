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
    an.pack_results_device(rec2.data_ptr())
    assert torch.equal(rec, rec2)
    # the gather through the C ABI with a 1-rank RCCL communicator
    comm = jg._lib.Comm(0, 1, jg._lib.Comm.unique_id(), device=0)
    out = torch.full_like(rec, np.nan)
    jg.gatherEstimatesDevice(an, comm, out.data_ptr())
    torch.cuda.synchronize()
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
    seen, order = [], []

    def on_done(j, an):
        order.append(j)
        seen.append(ring[j % 2].clone())
        torch.cuda.synchronize()

    seeds = [11, 12, 13, 14]
    res = pipe.run(seeds, iteration=40, tolerance=1e-8, on_done=on_done, record=lambda j: ring[j % 2].data_ptr(), records=2)
    assert order == [0, 1, 2, 3]
    pipe.close()
    ref = jg.gaussNewton(mon, batch=B)
    for j, seed in enumerate(seeds):
        jg.setNoise_(ref, np.random.Generator(np.random.PCG64(seed)), scale=1.0)
        ref.setVoltage(np.ones(n), np.zeros(n))
        jg.stateEstimation_(ref, iteration=40, tolerance=1e-8)
        rec = torch.zeros((B, 2 * n + 3), dtype=torch.float64, device="cuda")
        ref.pack_results_device(rec.data_ptr())
        assert torch.equal(rec, seen[j]), j
        assert np.array_equal(res[j][0], ref.method.iteration) and np.array_equal(res[j][1], ref.status)
    assert not torch.equal(seen[0], seen[1])                      # different seeds, different realisations
    ref.close()
