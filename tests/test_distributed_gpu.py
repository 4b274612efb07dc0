"""The N > 1 machinery on the 1-GPU box: the RCCL gather behind the C ABI (a 1-rank communicator: init, pack, ncclAllGather, record
bit-equal to the packed one), and bench.py started the way the driver starts it -- plain `python bench.py --gpus N` -- with two ranks
sharing the GPU over gloo (control flow, sharding, strong-scaling accounting) and with a 1-rank RCCL group (device records through
all_gather_into_tensor)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_case

pytestmark = pytest.mark.gpu


def test_allgather_through_the_c_abi_equals_the_packed_record(jg):
    import torch
    s = jg.powerSystem(load_case("case118"))
    an = jg.contingencyAnalysis(s, jg.outageList(s, 70, seed=3))
    jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    n = s.bus.number
    packed = torch.zeros((70, 2 * n + 2), dtype=torch.float64, device="cuda")
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    jg._lib.check(jg._lib.lib().jg_nr_pack_results_device(an._h, jg._lib.VP(packed.data_ptr())))
    comm = jg._lib.Comm(0, 1, jg._lib.Comm.unique_id(), device=0)
    assert jg._lib.lib().jg_comm_rank(comm.h) == 0 and jg._lib.lib().jg_comm_world(comm.h) == 1
    out = torch.full((70, 2 * n + 2), np.nan, dtype=torch.float64, device="cuda")
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    jg.gatherResultsDevice(an, comm, out.data_ptr())
    torch.cuda.current_stream().synchronize()
    assert torch.equal(out, packed)
    it, st, vm, va = jg.unpackResults(out)
    assert np.array_equal(it.cpu().numpy(), np.asarray(an.method.iteration)) and np.array_equal(st.cpu().numpy(), np.asarray(an.status))
    assert np.array_equal(vm.cpu().numpy(), np.asarray(an.voltage.magnitude))
    # a record that is already packed (what a ContingencyPipeline delivers), out of place
    out2 = torch.empty_like(out)
    torch.cuda.current_stream().synchronize()                                    # the fill runs on torch's stream, the library writes on its own: finish it first
    comm.allgather_device(packed.data_ptr(), out2.data_ptr(), packed.numel())
    assert torch.equal(out2, packed)
    comm.close()
    an.close()


def _bench(args, **env):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line"
    return json.loads(lines[0])


def test_bench_starts_by_itself_with_two_ranks():
    """`python bench.py --gpus 2` without a launcher: it re-executes itself under torch.distributed.run; two ranks share the one GPU
    over gloo (JG_BENCH_BACKEND: the measured configuration is RCCL), 512 scenarios per step sharded 256 + 256."""
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-se"], JG_BENCH_BACKEND="gloo", JG_BENCH_MAX_REPEATS="3")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 4 and d["region_repeats"] == 3
    assert d["config"]["scenarios_per_step"] == 512 and d["config"]["batch_per_gpu"] == 256
    assert d["value"] > 0 and d["converged_fraction"] == 1.0


def test_bench_with_a_one_rank_rccl_group():
    d = _bench(["--steps", "4", "--warmup", "1", "--no-cpu", "--no-se"], JG_BENCH_FORCE_DIST="1", JG_BENCH_MAX_REPEATS="3")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["converged_fraction"] == 1.0
    d2 = _bench(["--steps", "4", "--warmup", "1", "--no-cpu", "--no-se"], JG_BENCH_FORCE_DIST="1", JG_BENCH_GATHER="abi", JG_BENCH_MAX_REPEATS="3")
    assert d2["config"].get("gather") == "abi" and d2["converged_fraction"] == 1.0


def test_bench_at_the_drivers_flags_with_four_ranks():
    """The driver's command shape for N = 4 (`--steps 20 --warmup 3`), four ranks sharing the one GPU over gloo: 512 scenarios sharded 4 x 128, a
    rank solves its shares of 4 steps as one 512-lane device batch; the K-step region is repeated and the line carries the median and the spread,
    and says whether a region reaches the steady state of the pipeline (5 device batches with 3 in flight: it does not)."""
    d = _bench(["--gpus", "4", "--steps", "20", "--warmup", "3", "--no-cpu", "--no-se"], JG_BENCH_BACKEND="gloo", JG_BENCH_MAX_REPEATS="4")
    assert d["n_gpus"] == 4 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "strong"
    assert d["config"]["batch_per_gpu"] == 128 and d["config"]["lanes_per_device_batch"] == 512 and d["config"]["steps_per_device_batch"] == 4
    assert 3 <= d["region_repeats"] <= 4 and d["region_ms_min"] <= d["region_ms_median"] <= d["region_ms_max"]
    assert abs(d["ms_per_step"] * d["steps"] - d["region_ms_median"]) < 1e-6 * d["region_ms_median"]
    assert d["config"]["device_batches_per_region"] == 5 and d["config"]["pipeline_steady_state"] is False
    assert d["value"] > 0 and d["converged_fraction"] == 1.0


def test_bench_region_statistics_at_one_gpu():
    """N = 1 at the driver's flags: 20 device batches per region with three in flight (steady state), the region repeated until ~1 s is timed."""
    d = _bench(["--steps", "20", "--warmup", "3", "--no-cpu", "--no-se"])
    assert d["n_gpus"] == 1 and d["region_repeats"] >= 3 and d["config"]["pipeline_steady_state"] is True
    assert d["region_ms_max"] < 1.5 * d["region_ms_min"], "regions of the same work on an otherwise idle GPU"
    assert d["region_repeats"] * d["region_ms_median"] > 500.0 or d["region_repeats"] == 50


def test_bench_gathers_screen_summaries_with_two_ranks():
    """`--record summary`: a device batch delivers the 10-double screen summary per scenario (reduced on the device, stragglers' rows by their pool handle)
    and that is what the ranks gather -- 80 bytes per scenario instead of 16 n + 16; two ranks over gloo, and one rank through the C ABI's RCCL gather."""
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-se", "--record", "summary"], JG_BENCH_BACKEND="gloo", JG_BENCH_MAX_REPEATS="3")
    assert d["n_gpus"] == 2 and d["config"]["record"].startswith("screen summary") and d["value"] > 0 and d["converged_fraction"] == 1.0
    d1 = _bench(["--steps", "4", "--warmup", "1", "--no-cpu", "--no-se", "--record", "summary"], JG_BENCH_FORCE_DIST="1", JG_BENCH_GATHER="abi", JG_BENCH_MAX_REPEATS="3")
    assert d1["config"]["record"].startswith("screen summary") and d1["config"].get("gather") == "abi" and d1["converged_fraction"] == 1.0


def test_bench_at_the_drivers_flags_with_eight_ranks():
    """ The driver's command shape for the 8-GPU run -- `--gpus 8 --steps 20 --warmup 5` -- as a dry run: eight ranks share the one GPU over gloo.
    512 scenarios shard 8 x 64; deviceBatching(64, 20) = 10 steps per device batch = 640 lanes, the rank's whole region as TWO device batches, both in flight: the K-step
    region cannot reach the pipeline's steady state, so the line ALSO carries value_steady (three rounds of the batches in flight); value keeps the caller's K."""
    d = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5", "--no-cpu", "--no-se"], JG_BENCH_BACKEND="gloo", JG_BENCH_MAX_REPEATS="3", JG_BENCH_MIN_SECONDS="0.2")
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "strong"
    c = d["config"]
    assert c["batch_per_gpu"] == 64 and c["steps_per_device_batch"] == 10 and c["lanes_per_device_batch"] == 640 and c["device_batches_per_region"] == 2
    assert c["pipeline_steady_state"] is False and c["gather"].startswith("torch.distributed")
    assert d["value"] > 0 and d["converged_fraction"] == 1.0
    assert d["value_steady"] > 0 and d["steady_steps"] == 3 * c["device_batches_in_flight_per_gpu"] * 10
    assert abs(d["ms_per_step"] * 20 - d["region_ms_median"]) < 1e-6 * d["region_ms_median"]


def test_bench_state_estimation_workload_sharded():
    """`--workload se`: BASELINE config 4 as a sharded Monte-Carlo run with the NR line's shape -- two ranks over gloo (128 + 128 realisations of the 9241-bus
    set's little brother would hide nothing: the real grid, a small batch), and one rank through the C ABI's RCCL gather of the [., 2 n + 3] record."""
    d = _bench(["--workload", "se", "--gpus", "2", "--batch", "256", "--steps", "2", "--warmup", "1", "--no-cpu"], JG_BENCH_BACKEND="gloo", JG_BENCH_MAX_REPEATS="3",
               JG_BENCH_MIN_SECONDS="0.1")
    assert d["n_gpus"] == 2 and d["unit"] == "GN iterations/s" and d["scaling"] == "strong" and d["dtype"] == "f64"
    c = d["config"]
    assert c["batch_per_gpu"] == 128 and c["scenarios_per_step"] == 256 and c["rows"] > 90000 and "config 4" in c["workload"]
    assert d["value"] > 0 and d["converged_fraction"] == 1.0 and 4.0 <= d["iterations_per_scenario"] <= 12.0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and set(d["kernels"]) == {"rows", "gain", "factor", "backward"}
    assert d["value_steady"] > 0                                       # one device batch per region: fill + drain
    d1 = _bench(["--workload", "se", "--batch", "256", "--steps", "4", "--warmup", "1", "--no-cpu"], JG_BENCH_FORCE_DIST="1", JG_BENCH_GATHER="abi",
                JG_BENCH_MAX_REPEATS="3", JG_BENCH_MIN_SECONDS="0.1")
    assert d1["n_gpus"] == 1 and d1["config"]["gather"] == "abi" and d1["converged_fraction"] == 1.0 and d1["config"]["record"].startswith("magnitude | angle")
