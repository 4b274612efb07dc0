"""N > 1 path on CPU: world_size-2 gloo run of the scenario sharding + final gather (the only collective)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT, load_case


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_is_a_contiguous_partition(jg):
    for count in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [jg.shard(count, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == count
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            assert max(hi - lo for lo, hi in blocks) - min(hi - lo for lo, hi in blocks if hi > lo or True) <= -(-count // world)


def test_device_batching_of_a_shrinking_share(jg):
    """bench.py --merge: steps per device batch under strong scaling (512 scenarios per step over N ranks, K steps)."""
    assert jg.deviceBatching(512, 48) == 1 and jg.deviceBatching(700, 48) == 1            # N = 1: a step already fills the lanes
    assert jg.deviceBatching(256, 48) == 2 and jg.deviceBatching(128, 48) == 4 and jg.deviceBatching(64, 48) == 8
    assert jg.deviceBatching(64, 20) == 10                                                 # round 5: the rank's whole run (1 280 scenarios) as TWO batches, both in flight
    assert jg.deviceBatching(64, 10) == 10 and jg.deviceBatching(64, 9) == 9               # ... as ONE batch of up to 640 lanes
    assert jg.deviceBatching(128, 10) == 5 and jg.deviceBatching(128, 20) == 4             # 2 x 640; beyond 1 280 scenarios: 512-lane batches
    assert jg.deviceBatching(64, 3) == 3 and jg.deviceBatching(64, 1) == 1
    for share in (64, 100, 171, 256):
        for steps in (1, 5, 7, 20, 24, 96):
            m = jg.deviceBatching(share, steps)
            if 512 < share * steps <= 1280:                                                # one batch of up to 640 lanes, or two of them
                assert m == (steps if share * steps <= 640 else -(-steps // 2)) and m * share <= 640 + share
                continue
            assert 1 <= m <= max(1, min(512 // share, steps)) and m * share <= max(512, share)
            spare = -(-steps // m) * m - steps
            assert all(spare <= -(-steps // q) * q - steps for q in range(max(1, (min(512 // share, steps) + 1) // 2), min(512 // share, steps) + 1))


def test_outage_list_is_seeded_and_avoids_bridges(jg):
    s = jg.powerSystem(load_case("case118"))
    a = jg.outageList(s, 40, seed=512)
    b = jg.outageList(s, 40, seed=512)
    assert np.array_equal(a, b) and a.min() >= 1
    br = jg.bridges(s)
    assert not br[a - 1].any()
    assert np.all(s.branch.layout.status[a - 1] == 1)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    import juliagrid.jl_amd as jg
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B, n = 6, 5
    total = B * world
    lo, hi = jg.shard(total, rank, world)
    ids = torch.arange(lo, hi)
    iters = (ids % 4 + 2).to(torch.int32)
    status = (ids % 3 == 0).to(torch.int32)
    vm = (ids[:, None] * 10 + torch.arange(n)[None, :]).to(torch.float64)
    va = -vm
    packed = torch.cat([vm, va, iters[:, None].double(), status[:, None].double()], dim=1)      # the record jg_nr_pack_results_device writes
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    g_it, g_st, g_vm, g_va = jg.gatherResults(dist, packed)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1                                   # ONE collective for the whole result (SURVEY 8e)
    all_ids = torch.arange(total)
    assert torch.equal(g_it, (all_ids % 4 + 2).long())
    assert torch.equal(g_st, (all_ids % 3 == 0).long())
    assert torch.equal(g_vm, (all_ids[:, None] * 10 + torch.arange(n)[None, :]).to(torch.float64))
    assert torch.equal(g_va, -g_vm)
    # whole-job accounting used by bench.py: sum of iterations, max of times
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([int(iters.sum())], dtype=torch.int64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    assert t.item() == world and c.item() == int((all_ids % 4 + 2).sum())
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %d ok" % rank + chr(10))          # ONE write: the two ranks share the pipe
    sys.stdout.flush()
""")


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    for attempt in range(3):                          # the rendezvous port is picked by probing: retry if another process grabbed it
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                             capture_output=True, text=True, timeout=300, env=env)
        if out.returncode == 0 or "AssertionError" in out.stderr:
            break
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


WORKER_SE = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    import juliagrid.jl_amd as jg
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B, n = 5, 4
    total = B * world
    lo, hi = jg.shard(total, rank, world)
    ids = torch.arange(lo, hi)
    vm = (ids[:, None] * 7 + torch.arange(n)[None, :]).to(torch.float64)
    va = -0.5 * vm
    iters = (ids % 3 + 4).double()
    status = (ids % 5 == 0).double()
    obj = (ids * 1000 + 0.25).double()
    packed = torch.cat([vm, va, iters[:, None], status[:, None], obj[:, None]], dim=1)      # the record jg_gn_pack_results_device writes: [B, 2 n + 3]
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    g_it, g_st, g_obj, g_vm, g_va = jg.gatherEstimates(dist, packed)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1                                   # ONE collective for the whole Monte-Carlo result
    all_ids = torch.arange(total)
    assert torch.equal(g_it, (all_ids % 3 + 4).long()) and torch.equal(g_st, (all_ids % 5 == 0).long())
    assert torch.equal(g_obj, (all_ids * 1000 + 0.25).double())
    assert torch.equal(g_vm, (all_ids[:, None] * 7 + torch.arange(n)[None, :]).to(torch.float64)) and torch.equal(g_va, -0.5 * g_vm)
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank %d ok" % rank + chr(10))
    sys.stdout.flush()
""")


def test_two_rank_gloo_gather_of_estimation_records(tmp_path):
    """The Monte-Carlo side of SURVEY 8(e): realisations shard contiguously, ONE all-gather of the [., 2 n + 3] record (magnitude | angle | iterations |
    status | objective) gives every rank the whole study in realisation order."""
    script = tmp_path / "worker_se.py"
    script.write_text(WORKER_SE.format(root=ROOT))
    for attempt in range(3):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                             capture_output=True, text=True, timeout=300, env=env)
        if out.returncode == 0 or "AssertionError" in out.stderr:
            break
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_predicted_value_of_an_n_gpu_run_comes_from_the_shard_emulation():
    """bench.py: the N > 1 line carries `predicted` = N x what ONE GPU reaches on a rank's share (profiles/bench_shards.json, the one-rank emulation of the
    strong-scaling shares merged to 512-lane device batches) -- the pool has one GPU per box, the 1 -> 8 curve itself is the driver's to measure."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    shards = json.load(open(os.path.join(ROOT, "profiles", "bench_shards.json")))
    assert b.predicted_from_shards("nr", 1, 512) is None                       # N = 1 is measured, not predicted
    for wl in ("nr", "se"):
        for world in (2, 4, 8):
            row = [r for r in shards[wl] if r["scenarios_per_step"] * world == 512][-1]
            p = b.predicted_from_shards(wl, world, 512)
            assert p["per_gpu_value"] == row["value"] and abs(p["value"] - world * row["value"]) < 1e-9 * p["value"]
            assert 0.8 * shards[wl][0]["value"] < row["value"] < 1.2 * shards[wl][0]["value"], "a rank's merged share runs at about the N = 1 rate"
    assert b.predicted_from_shards("nr", 3, 512) is None                       # no measured share for that N
