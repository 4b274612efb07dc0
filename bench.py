#!/usr/bin/env python3
"""bench.py -- NR iterations/s on the 10k-bus grid (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W            (N=1 direct; N>1 under torch.distributed.run)

Workload (config.workload): batched N-1 contingency AC power flow on the shipped 10 000-bus grid
case_ACTIVSg10k ("10k-bus grid" of the metric; case9241pegase is not shipped by the reference):
`--batch` (default 512) single-branch outage scenarios PER GPU, each started from the base-case
solution and iterated to 1e-8 with the reference's loop accounting.  One step = one pass of the hot
path over the batch: restore the start point inside HBM, run powerFlow! for every scenario
(fused mismatch+Jacobian assembly, block-LU refactorization, triangular solves, update, per
scenario convergence control), and for N > 1 gather the results over RCCL.  Scenarios shard
contiguously across ranks with no data-path collective (weak scaling: per-GPU batch fixed).
`--inflight` (default 3) steps are in flight per GPU at once (juliagrid.jl_amd ContingencyPipeline: one handle,
HIP stream and host thread each), because a single batch leaves most of the chip idle during the narrow
dependency levels of the sparse LU and during its last (straggler) iterations; every step is still a full,
independent solve of all its scenarios and all K steps complete inside the timed region.

value = total Newton-Raphson iterations (sum over all scenarios, all ranks, all K steps) / seconds.
Inputs are resident in HBM when the timed region starts.  The JSON line also carries `roofline`
(dominant kernel, algorithmic bytes / HIP-event time / 8 TB/s) and `cpu_baseline` (the C oracle on
one host core, bounded sample of the same scenarios).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(d, batch):
    """SURVEY.md 8(d) per-unit figures x units per launch (see DESIGN.md 'Algorithmic bytes').
    Assembly: per scenario J values + mismatch written, V/theta + injections read; shared: Ybus values,
    column index, row pointers, bus type.  Block layout: 4 doubles per Ybus block."""
    n, nnzY = d["n"], d["nnzY"]
    asm_per = 32 * nnzY + 16 * n + 16 * n + 16 * n          # J blocks + f(2) written; V,theta; P,Q read
    asm_shared = 20 * nnzY + 4 * (n + 1) + n                  # G,B (16) + col (4), rowptr, type
    lu_per = 32 * nnzY + 2 * 32 * d["lu_blocks"]              # read J once, write+read-back L/U/Dinv blocks
    solve_per = 32 * d["lu_blocks"] + 4 * 16 * n              # factor read once; rhs, work, increment
    return dict(assembly=batch * asm_per + asm_shared, lu=batch * lu_per, solve=batch * solve_per)


def cpu_baseline(case_tables, labels, start_vm, start_va, budget_s=12.0):
    """The oracle (restatement of the reference algorithm; KLU-style LU with refactor reuse) on ONE host
    core: the reference's own contingency loop over a bounded sample of the same scenarios."""
    from oracle import oracle as O
    import juliagrid.jl_amd as jg
    osys = O.OracleSystem(case_tables)
    cold = []
    for _ in range(3):                               # ONE power flow from the case's start point, symbolic analysis included
        o = O.OracleNR(osys)                         # (what a cold run of the reference pays: BASELINE configs 1-2)
        tc = time.perf_counter()
        o.power_flow()                               # base case: symbolic + first factorization
        cold.append(time.perf_counter() - tc)
    cold_iters = o.iteration
    s = jg.powerSystem(case_tables)
    jg.acModel_(s)
    iters = 0
    done = 0
    t0 = time.perf_counter()
    for lab in labels:
        ptr, dy = jg.outagePatch(s, int(lab))
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, dv)
        o.set_voltage(start_vm, start_va)
        o.power_flow(iteration=20, tolerance=1e-8)
        iters += o.iteration
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, -dv)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "NR iterations/s", "cores": 1, "kind": "port",
            "sample": f"{done} of the same N-1 scenarios, {iters} iterations in {dt:.2f} s, "
                      "oracle/jg_oracle.c: serial assembly + KLU-style refactor/solve, single thread",
            "ms_per_iteration": 1e3 * dt / max(iters, 1), "ms_per_solve": 1e3 * dt / max(done, 1),
            "single_cold": {"ms_per_solve": 1e3 * float(np.median(cold)), "iterations": int(cold_iters),
                            "what": "one power flow from the case's start point incl. symbolic analysis (compare single_instance)"},
            "_iters": iters, "_done": done}


def cpu_baseline_se(jg, s, case, pf, budget_s=15.0):
    """The C oracle (restatement of acWLS / normalEquation! / increment! / solve!, KLU-style LU with refactor reuse) on
    ONE host core: the same measurement configuration, noise-free readings, flat start, repeated until the budget."""
    from oracle import oracle as O
    tables = jg.case9241synth() if case == "case9241synth" else None
    if tables is None:
        with np.load(os.path.join(ROOT, "tests", "golden", "cases", case + ".npz")) as z:
            tables = {k: z[k] for k in z.files}
    osys = O.OracleSystem(tables)
    on = O.OracleNR(osys)
    assert on.power_flow(iteration=20, tolerance=1e-11) == 0
    vm, va = on.voltage()
    br, _ = O.exact_quantities(osys, vm, va)
    tab = O.MeterTable()
    O.add_from_power_flow(tab, osys, vm, va, "voltmeter")
    O.add_from_power_flow(tab, osys, vm, va, "wattmeter")
    O.add_from_power_flow(tab, osys, vm, va, "varmeter")
    sel = set(range(1, osys.n + 1, 10))
    for i in range(osys.n):
        if (i + 1) in sel:
            tab.add("pmu", 0, i + 1, vm[i], 1e-8, 1, va[i], 1e-8, 1)
    for k in np.flatnonzero(osys.status == 1):
        if int(tables["br_from"][k]) in sel and br[k, 4] >= 1e-6:
            tab.add("pmu", 1, k + 1, br[k, 4], 1e-8, 1, br[k, 5], 1e-8, 1)
    n = osys.n
    t0 = time.perf_counter()
    gn = O.OracleGN(osys, tab, np.ones(n), np.zeros(n))           # includes the symbolic analysis, like the first GPU solve does not
    t_setup = time.perf_counter() - t0
    iters = solves = 0
    t0 = time.perf_counter()
    while True:
        gn.set_voltage(np.ones(n), np.zeros(n))
        gn.state_estimation(iteration=40, tolerance=1e-8)
        iters += gn.iteration
        solves += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "GN iterations/s", "cores": 1, "kind": "port",
            "sample": f"{solves} solves of the same measurement configuration (noise-free), {iters} iterations in {dt:.2f} s, "
                      f"oracle/jg_oracle_se.c, single thread; model setup {t_setup:.2f} s not counted",
            "ms_per_iteration": 1e3 * dt / max(iters, 1), "ms_per_solve": 1e3 * dt / max(solves, 1)}


def cpu_baseline_all_cores(case, count, seed, total, cores, timeout_s=180):
    """The same oracle loop on every host core (one process per core, scenarios dealt out contiguously): the
    OpenMP-over-scenarios variant of SURVEY.md 8(d).  The reference itself is single-threaded.  Workers are plain
    subprocesses with a timeout (the parent holds a live HIP context); any failure just omits this leg."""
    import subprocess
    per = max(1, -(-count // cores))
    cmds = [[sys.executable, os.path.join(ROOT, "oracle", "baseline_worker.py"), case, str(f), str(min(per, count - f)), str(seed), str(total)]
            for f in range(0, count, per)]
    try:
        procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for c in cmds]
        out = []
        deadline = time.time() + timeout_s
        for p in procs:
            so, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            out.append(json.loads(so.strip().splitlines()[-1]))
    except Exception:
        for p in procs:
            if p.poll() is None:
                p.kill()
        return None
    iters, done, dt = sum(o["iters"] for o in out), sum(o["done"] for o in out), max(o["seconds"] for o in out)
    return {"value": iters / dt, "unit": "NR iterations/s", "cores": len(cmds), "kind": "port",
            "sample": f"{done} of the same N-1 scenarios over {len(cmds)} processes (one per host core), {iters} iterations, "
                      f"slowest process {dt:.2f} s (each process's base-case solve and symbolic analysis not counted, as in the "
                      "single-thread leg)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="scenarios per GPU")
    ap.add_argument("--case", default="case_ACTIVSg10k", help="case_ACTIVSg10k (the metric's 10k-bus grid) | case9241synth | any fixture")
    ap.add_argument("--inflight", type=int, default=3, help="batches (steps) in flight per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch
    import juliagrid.jl_amd as jg

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # JG_BENCH_BACKEND=gloo: dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share the
    # devices round-robin, collectives go through host memory).  The measured configuration is always nccl = RCCL.
    backend = os.environ.get("JG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    cdev = "cuda" if backend == "nccl" else "cpu"      # where the tensors of the collectives live
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    if args.case == "case9241synth":                 # the seeded PEGASE-shaped stand-in for case9241pegase
        tables = jg.case9241synth()
    else:
        with np.load(os.path.join(ROOT, "tests", "golden", "cases", args.case + ".npz")) as z:
            tables = {k: z[k] for k in z.files}

    # ---- setup (untimed): base case, scenario list, shard, upload ---------------------------
    B = args.batch
    system = jg.powerSystem(tables)
    base = jg.newtonRaphson(system, batch=1, device=local)
    jg.powerFlow_(base)
    assert base.status == 0
    base_iters = int(base.method.iteration)
    vm0, va0 = base.voltage.magnitude.copy(), base.voltage.angle.copy()
    # single-instance latency (BASELINE config 2/3 "ms/solve"), same handle, from the case's start point
    t_single = []
    for _ in range(5):
        jg.setInitialPoint_(base)
        t0 = time.perf_counter()
        jg.powerFlow_(base, fetch=False)
        t_single.append(time.perf_counter() - t0)
    base.close()

    # Scenario selection (untimed).  Rank r owns the contiguous block r of a seeded shuffle of the non-bridge branches
    # (2 B candidates per rank) and screens the first B of them THAT HAVE A POWER FLOW: a contingency without a solution
    # runs to the iteration limit (20 iterations for one lane while the other 511 of its batch have long finished), so one
    # of them inside a rank's list would measure that scenario, not the path.  They are counted and reported, not hidden:
    # the first 1024 candidates of case_ACTIVSg10k hold exactly one (branch 11127).  N = 1 keeps the first 512.
    cand = jg.outageList(system, 2 * B * world, seed=512)
    lo, hi = jg.shard(2 * B * world, rank, world)
    mine = cand[lo:hi]
    pipe = jg.ContingencyPipeline(system, B, inflight=args.inflight, device=local, start=(vm0, va0))
    it_pre, st_pre = pipe.screen(mine, iteration=20, tolerance=1e-8)
    solvable = np.flatnonzero(st_pre == 0)
    excluded_local = int(np.sum(st_pre[:solvable[B - 1] + 1] != 0)) if solvable.size >= B else int(np.sum(st_pre != 0))
    if solvable.size < B:
        raise SystemExit(f"rank {rank}: only {solvable.size} of {mine.size} candidate contingencies have a power flow")
    labels = mine[solvable[:B]]
    for h in pipe.handles:
        jg.setOutages_(h, labels)                     # the scenarios stay resident: a step re-solves them from the start point
    an = pipe.handles[0]
    n = system.bus.number
    out_vm = torch.empty((B, n), dtype=torch.float64, device="cuda")
    out_va = torch.empty((B, n), dtype=torch.float64, device="cuda")
    res = torch.empty((B, 2), dtype=torch.int32, device="cuda")

    def gather(job, h):                               # caller's thread, step order: the only collective (final gather of results)
        h.voltage_device(out_vm.data_ptr(), out_va.data_ptr())
        res.copy_(torch.from_numpy(np.stack([h.method.iteration, h.status], axis=1).astype(np.int32)))
        if cdev == "cuda":
            jg.gatherResults(dist, res[:, 0], res[:, 1], out_vm, out_va)
            torch.cuda.current_stream().synchronize()   # the staging buffers are rewritten by the next step's results
        else:
            jg.gatherResults(dist, res[:, 0].cpu(), res[:, 1].cpu(), out_vm.cpu(), out_va.cpu())

    def run(steps):
        out = pipe.run([None] * steps, iteration=20, tolerance=1e-8, on_done=gather if world > 1 else None)
        return int(sum(int(np.sum(it)) for it, _ in out)), out[-1][1]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    iters_local, last_status = run(args.steps)
    fence()
    dt = time.perf_counter() - t0

    conv_local = int(np.sum(last_status == 0))
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cnt = torch.tensor([iters_local, conv_local, excluded_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dt = float(tt.item())
        iters_total, conv_total, excluded_total = int(cnt[0].item()), int(cnt[1].item()), int(cnt[2].item())
    else:
        iters_total, conv_total, excluded_total = iters_local, conv_local, excluded_local

    if rank == 0:
        d = an.dims
        ab = algorithmic_bytes(d, an.batch)
        # live kernel timing with HIP events on the library's own stream
        t_asm = an.time_kernel(0, 20)
        t_lu = an.time_kernel(1, 10)
        t_sol = an.time_kernel(2, 10)
        kern = {
            "assembly": {"ms": t_asm, "bytes": ab["assembly"], "launches": 1},
            "lu": {"ms": t_lu, "bytes": ab["lu"], "launches": d["lu_launches"]},
            "solve": {"ms": t_sol, "bytes": ab["solve"], "launches": d["solve_launches"]},
        }
        for k in kern.values():
            k["GBps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            k["frac"] = k["GBps"] / HBM_PEAK_GBS
        dom = max(kern, key=lambda k: kern[k]["ms"])
        names = {"assembly": "k_assemble", "lu": "k_fact_level", "solve": "k_bwd_level"}
        # HBM bytes per logical launch from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE x2 +
        # WRITE_SIZE, calibrated on a kernel of known byte count); only valid for the grid and batch it was collected at
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                if int(pj.get("batch_ld", 0)) == an.batch and pj.get("grid") == args.case:
                    key = {"assembly": "k_assemble", "lu": "k_fact", "solve": "k_fwd+k_bwd"}[dom]
                    traffic = pj["traffic_per_logical_launch"].get(key)
                    for kk, nm in (("assembly", "k_assemble"), ("lu", "k_fact"), ("solve", "k_fwd+k_bwd")):
                        kern[kk]["hbm_traffic_bytes_pmc"] = pj["traffic_per_logical_launch"].get(nm)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic,
                    "per_launch_ms": kern[dom]["ms"] / kern[dom]["launches"],
                    "algorithmic_bytes": kern[dom]["bytes"]}
        line = {
            "metric": "NR iterations/sec (batched N-1 AC power flow, 10k-bus grid)",
            "value": iters_total / dt, "unit": "NR iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.case} batched N-1 contingency Newton-Raphson, {B} scenarios per GPU "
                                   f"({B * world} total), start = base-case solution, tol 1e-8, max 20 iterations",
                       "grid": args.case, "buses": n, "batch_per_gpu": B, "dimJ": d["dimJ"], "nnzJ": d["nnzJ"],
                       "lu_blocks_2x2": d["lu_blocks"], "lu_terms": d["lu_terms"],
                       "launches_per_iteration": 2 + d["lu_launches"] + d["solve_launches"],
                       "steps_in_flight_per_gpu": len(pipe.handles),
                       "parallelism": f"scenario-sharded x{world}, RCCL all-gather of results only",
                       "scenario_selection": f"per GPU the first {B} solvable contingencies of its block of a seeded shuffle of the "
                                             f"non-bridge branches; {excluded_total} candidate(s) without a power flow skipped"},
            "scenarios_per_s": B * world * args.steps / dt,
            "ms_per_solve_batched": 1e3 * dt / (B * world * args.steps),
            "iterations_per_scenario": iters_total / (B * world * args.steps),
            "converged_fraction": conv_total / (B * world),
            "single_instance": {"ms_per_solve": 1e3 * float(np.median(t_single)), "iterations": base_iters,
                                "ms_per_iteration": 1e3 * float(np.median(t_single)) / max(base_iters, 1)},
            "roofline": roofline,
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(tables, labels[: max(64, min(B, 512))], vm0, va0)
            cb.pop("_iters"), cb.pop("_done")
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if cores > 1:                               # the box's whole host: reported next to the single-thread reference path
                allc = cpu_baseline_all_cores(args.case, max(64, min(B, 512)), 512, B * world, min(cores, 64))
                if allc:
                    line["cpu_baseline_all_cores"] = allc
                    line["speedup_vs_cpu_all_cores"] = line["value"] / allc["value"]
        print(json.dumps(line))
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
