#!/usr/bin/env python3
"""bench.py -- NR iterations/s on the 10k-bus grid (BASELINE.json metric), one process per GPU.

  python bench.py --gpus N --steps K --warmup W            (N=1 direct; N>1 under torch.distributed.run)

Workload (config.workload): batched N-1 contingency AC power flow on the shipped 10 000-bus grid case_ACTIVSg10k ("10k-bus
grid" of the metric; case9241pegase is not shipped by the reference): `--batch` (default 512) single-branch outage scenarios,
each started from the base-case solution and iterated to 1e-8 with the reference's loop accounting.  One step = one pass of
the hot path over the batch: restore the start point inside HBM, run powerFlow! for every scenario (fused mismatch+Jacobian
assembly, block-LU refactorisation -- level launches + multifrontal top --, triangular solves, update, per-scenario
convergence control), pack the results (V | theta | iterations | status per scenario) into one device buffer and, for
N > 1, gather them with ONE RCCL all-gather.  Scenarios shard contiguously across ranks; no data-path collective.

  --scaling strong (default, BASELINE config 5): the SAME `--batch` scenarios are sharded over the N GPUs (512 / 8 = 64 per
      GPU at N = 8); at N = 1 this is the 512-scenario batch on one GPU.
  --scaling weak: `--batch` scenarios PER GPU.

`--merge` M: a rank solves its shares of M consecutive steps in ONE device batch of M x (scenarios per GPU) lanes (default
under strong scaling: as many as bring the shard back to 512 lanes -- M = 8 at N = 8, M = 1 at N = 1).  The scenarios of a
job are independent; how a rank groups its K x (512 / N) scenario solves into launches does not change the work, and the
collective then carries the records of M steps at once.

`--inflight` device batches are in flight per GPU at once (ContingencyPipeline: one handle, HIP stream and host thread each): a
single batch leaves most of the chip idle during the narrow dependency levels of the sparse LU and during its straggler
iterations.  Default: 3 at 512 lanes, more for smaller device batches (the scenarios in flight per GPU stay ~1 536, at
most 12 batches).  Every step is a full, independent solve and all K steps complete inside the timed region.

value = total Newton-Raphson iterations (sum over all scenarios, all ranks, all K steps) / seconds; inputs are resident in HBM
when the timed region starts.  The JSON line also carries `roofline` (the factorisation: algorithmic bytes / HIP-event time /
8 TB/s), `cpu_baseline` (the C oracle on one host core, bounded sample of the same scenarios) and, at N = 1, `config4_se`
(BASELINE config 4: Gauss-Newton WLS state estimation on the 9241-bus PEGASE-shaped grid).
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
    """SURVEY.md 8(d), per launch group of one iteration (DESIGN.md section 3 states each figure).
    assembly  8(d) verbatim: per scenario J values + mismatch written, V / theta and P / Q injections read; shared: Ybus values
              + column index, row pointers, bus type, position map.
    lu        the factor lives IN the storage the assembly wrote (no separate Jacobian array): every block of L + D + U is
              read once and written once = 2 x 32 B x lu_blocks per scenario (8(d)'s extra "read J" term is that same read).
    solve     the backward sweep reads U and D only (the forward elimination rides in the factorisation): 32 B x (upper +
              diagonal blocks); y read, x written twice (pivot order + bus order), V / theta read and written."""
    n, nnzY, nnzJ, dimJ, lu = d["n"], d["nnzY"], d["nnzJ"], d["dimJ"], d["lu_blocks"]
    asm_per = 8 * nnzJ + 8 * dimJ + 16 * n + 16 * n
    asm_shared = 20 * nnzY + 4 * (n + 1) + n + 4 * nnzJ
    lu_per = 64 * lu
    solve_per = 32 * ((lu + n) // 2) + 3 * 16 * n + 32 * n
    return dict(assembly=batch * asm_per + asm_shared, lu=batch * lu_per, solve=batch * solve_per)


def load_tables(jg, case):
    if case == "case9241synth":                  # the seeded PEGASE-shaped stand-in for case9241pegase
        return jg.case9241synth()
    if case == "synth25k":                       # stand-ins for the reference's 25 000 / 70 000-bus datasets (not shipped): tests/test_big_grids_gpu.py
        from juliagrid.jl_amd.synthetic import pegaseShaped
        return {k: np.array(v) for k, v in pegaseShaped(n=25000, nb=int(25000 * 16049 / 9241), ng=int(25000 * 1445 / 9241), seed=25000, load_scale=0.1).items()}
    if case in ("tiled70k", "tiled90k"):         # (tiled90k: nine instances tied to the first one -- for the 82 000-bus set)
        from juliagrid.jl_amd.synthetic import tiledGrid
        t = load_tables(jg, "case_ACTIVSg10k")
        one = jg.newtonRaphson(jg.powerSystem(t))
        jg.powerFlow_(one)
        jg.power_(one)
        slack = int(np.flatnonzero(np.asarray(one.system.bus.layout.type) == 3)[0])
        p_slack = float(np.asarray(one.power.supply.active).reshape(-1)[slack])
        one.close()
        return tiledGrid(t, 9, slack_active=p_slack, star=True) if case == "tiled90k" else tiledGrid(t, 7, slack_active=p_slack)
    with np.load(os.path.join(ROOT, "tests", "golden", "cases", case + ".npz")) as z:
        return {k: z[k] for k in z.files}


def single_instance_configs(jg, device, skip):
    """ BASELINE configs 2 and 3 are SINGLE-instance configs: one warm power flow from the case's start point on the device and on the oracle
    (one pinned host core, the analysis that already holds its symbolic factorisation), for the grids the headline does not run."""
    from oracle import oracle as O
    out = {}
    for case in ("case1354pegase", "case9241synth"):
        if case == skip:
            continue
        try:
            tables = load_tables(jg, case)
            an = jg.newtonRaphson(jg.powerSystem(tables), batch=1, device=device)
            jg.powerFlow_(an)
            gpu = []
            for _ in range(7):
                jg.setInitialPoint_(an)
                t0 = time.perf_counter()
                jg.powerFlow_(an, fetch=False)
                gpu.append(time.perf_counter() - t0)
            iters = int(an.method.iteration)
            an.close()
            osys = O.OracleSystem(tables)
            o = O.OracleNR(osys)
            svm, sva = o.vm.copy(), o.va.copy()
            o.power_flow()
            cpu = []
            for _ in range(5):
                o.set_voltage(svm, sva)
                t0 = time.perf_counter()
                o.power_flow()
                cpu.append(time.perf_counter() - t0)
            out[case] = {"ms_per_solve": 1e3 * float(np.median(gpu)), "iterations": iters, "cpu_warm_ms_per_solve": 1e3 * float(np.median(cpu)), "cpu_iterations": int(o.iteration),
                         "speedup_vs_cpu_warm": float(np.median(cpu) / np.median(gpu))}
        except Exception as e:                          # never break the line
            out[case] = {"error": repr(e)}
    return out


def cpu_baseline(case_tables, labels, start_vm, start_va, budget_s=12.0):
    """The oracle (restatement of the reference algorithm; KLU-style LU with refactor reuse) on ONE host core: the
    reference's own contingency loop over a bounded sample of the same scenarios, plus one power flow from the
    case's start point both ways a reference user can run it: cold (symbolic analysis included) and warm (lu! path)."""
    from oracle import oracle as O
    import juliagrid.jl_amd as jg
    O.lib()                                          # (JG_ORACLE_FAST=1 set in main(): the -O3 -march=native build of the same sources)
    aff = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if aff:                                          # the single-thread leg is pinned to one core
        try:
            os.sched_setaffinity(0, {sorted(aff)[len(aff) // 2]})
        except OSError:
            aff = None
    osys = O.OracleSystem(case_tables)
    cold = []
    for _ in range(3):                               # ONE power flow from the case's start point on a system whose AC model exists: newtonRaphson()
        tc = time.perf_counter()                     # (Jacobian pattern and maps) + powerFlow!() with the symbolic analysis of the LU -- what
        o = O.OracleNR(osys)                         # single_instance.setup_ms + ms_per_solve time on the device side
        o.power_flow()
        cold.append(time.perf_counter() - tc)
    cold_iters = o.iteration
    start = O.OracleNR(osys)
    svm, sva = start.vm.copy(), start.va.copy()      # the start newtonRaphson() builds
    warm = []
    for _ in range(5):                               # the same solve on the analysis that already holds its symbolic factorisation
        o.set_voltage(svm, sva)                      # (the reference's fast path: docs/src/manual/acPowerFlow.md:421)
        tc = time.perf_counter()
        o.power_flow()
        warm.append(time.perf_counter() - tc)
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
    if aff:
        os.sched_setaffinity(0, aff)
    return {"value": iters / dt, "unit": "NR iterations/s", "cores": 1, "kind": "port",
            "build": "gcc -O3 -march=native (oracle/_fast; the parity checks use the -O2 -ffp-contract=off build)" if O.FAST_BUILD else "gcc -O2 -ffp-contract=off",
            "pinned": bool(aff),
            "sample": f"{done} of the same N-1 scenarios, {iters} iterations in {dt:.2f} s, "
                      "oracle/jg_oracle.c: serial assembly + KLU-style refactor/solve, single thread",
            "ms_per_iteration": 1e3 * dt / max(iters, 1), "ms_per_solve": 1e3 * dt / max(done, 1),
            "single_cold": {"ms_per_solve": 1e3 * float(np.median(cold)), "iterations": int(cold_iters),
                            "what": "newtonRaphson() + one power flow from the case's start point incl. symbolic analysis, on a system whose AC model exists (compare single_instance.setup_ms + ms_per_solve)"},
            "single_warm": {"ms_per_solve": 1e3 * float(np.median(warm)), "iterations": int(o.iteration),
                            "what": "the same solve with the symbolic factorisation reused (compare single_instance.ms_per_solve)"}}


def cpu_splu_leg(J, reps=5):
    """an independent CPU solver on the same matrix -- scipy.sparse.linalg.splu (SuperLU, COLAMD) factorises the base-case
    Jacobian (the reference's CSC, read back from the handle) and solves one right-hand side; one core.  None without scipy."""
    try:
        import scipy.sparse as sp
        from scipy.sparse.linalg import splu
    except Exception:
        return None
    A = sp.csc_matrix((np.asarray(J.nzval, dtype=float), np.asarray(J.rowval) - 1, np.asarray(J.colptr) - 1))
    b = np.ones(A.shape[0])
    tf, ts = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); lu = splu(A); t1 = time.perf_counter(); lu.solve(b); t2 = time.perf_counter()
        tf.append(t1 - t0); ts.append(t2 - t1)
    return {"factor_ms": 1e3 * float(np.median(tf)), "solve_ms": 1e3 * float(np.median(ts)), "nnz_LU": int(lu.L.nnz + lu.U.nnz), "dim": int(A.shape[0]),
            "what": "scipy.sparse.linalg.splu (SuperLU, symbolic + numeric every call) + one solve of the base-case Jacobian, one core: the linear step of "
                    "ONE Newton iteration of ONE scenario"}


def hbm_measured_gbps(torch, nbytes=1 << 30, reps=20):
    """Measured device-to-device copy bandwidth (read + written bytes per second) printed beside the 8 TB/s the roofline is priced against
    (MI355X_MICROARCH.md quotes 6.29 TB/s for a copy)."""
    x = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del x, y
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def cpu_baseline_se(jg, s, case, pf, budget_s=15.0):
    """The C oracle (restatement of acWLS / normalEquation! / increment! / solve!, KLU-style LU with refactor reuse) on
    ONE host core: the same measurement configuration, noise-free readings, flat start, repeated until the budget."""
    from oracle import oracle as O
    tables = load_tables(jg, case)
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


def cpu_baseline_all_cores(case, count, seed, total, cores, timeout_s=240):
    """The same oracle loop on the host's cores, one PINNED process per core with a contiguous block of at least 64 scenarios
    (a worker's start-up -- python, symbolic analysis, base case -- is outside its clock; with fewer scenarios per worker
    the figure measured start-up skew, not the loop).  The reference itself is single-threaded.  Workers are plain
    subprocesses with a timeout (the parent holds a live HIP context); any failure just omits this leg."""
    import subprocess
    per = max(64, -(-count // cores))
    workers = max(1, min(cores, count // per))
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(workers))
    cmds = [[sys.executable, os.path.join(ROOT, "oracle", "baseline_worker.py"), case, str(w * per), str(per), str(seed), str(total), str(cpus[w % len(cpus)])]
            for w in range(workers)]
    procs = []
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
            "sample": f"{done} of the same N-1 scenarios over {len(cmds)} pinned processes ({per} scenarios each), {iters} iterations, "
                      f"slowest process {dt:.2f} s (each process's base-case solve and symbolic analysis not counted, as in the "
                      "single-thread leg)"}


def se_config4(jg, case="case9241synth", batch=512, steps=12, warmup=2, inflight=2, cpu=True, cpu_budget_s=10.0):
    """BASELINE config 4: Gauss-Newton WLS state estimation (PMU + legacy) on the 9241-bus PEGASE-shaped grid, 1 GPU.
    Measurement set: voltmeter at every bus, wattmeter + varmeter at every bus and both ends of every
    in-service branch (variance 1e-4), PMUs at every 10th bus (bus phasor + from-end current phasors, variance 1e-8),
    synthesised from the converged power flow; scenario b reads z + sigma * N(0,1) (seed 4).  One step = restore the flat
    start inside HBM and run stateEstimation! (tol 1e-8, max 40) for the whole batch."""
    import threading
    s = jg.powerSystem(case)
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, tolerance=1e-11)
    assert pf.status == 0
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf, variance=1e-4)
    jg.addWattmeter_(mon, pf, variance=1e-4)
    jg.addVarmeter_(mon, pf, variance=1e-4)
    jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
    n = s.bus.number
    handles = []
    for k in range(max(1, inflight)):                 # like ContingencyPipeline: the batches of different handles overlap on the GPU
        h = jg.gaussNewton(mon, batch=batch)
        jg.setNoise_(h, np.random.Generator(np.random.PCG64(4 + k)), scale=1.0)
        h.setVoltage(np.ones(n), np.zeros(n))
        h.snapshot_voltage()                          # the flat start stays resident in HBM
        handles.append(h)
    an = handles[0]

    def step(h):
        h.restore_voltage()
        jg.stateEstimation_(h, iteration=40, tolerance=1e-8, fetch=False)
        return int(np.sum(h.method.iteration))

    def run(nsteps):
        out = [0] * len(handles)

        def work(k):
            for _ in range(k, nsteps, len(handles)):
                out[k] += step(handles[k])
        ths = [threading.Thread(target=work, args=(k,)) for k in range(len(handles))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return sum(out)

    run(warmup * len(handles))
    t0 = time.perf_counter()
    iters = run(steps)
    dt = time.perf_counter() - t0
    d = an.dims
    B = batch
    kern = {}
    algo = {"rows": B * (16 * d["slots"] + 16 * d["m"] + 16 * n), "gain": B * (16 * d["slots"] + 8 * d["m"] + 32 * d["gain_blocks"] + 16 * n),
            "factor": B * (64 * ((d["lu_blocks"] + n) // 2)), "backward": B * (32 * ((d["lu_blocks"] + n) // 2) + 64 * n)}
    for k, name in enumerate(("rows", "gain", "factor", "backward")):
        # (groups long enough that the first launches on a chip that has just been idle do not set the figure -- as for the power flow kernels
        # below: 3 launches per group measured the row kernel at 0.89 ms, 12 at 0.80 on the same handle)
        ms = float(np.median([an.time_kernel(k, 6 if name == "factor" else 12) for _ in range(5)]))
        kern[name] = {"ms": ms, "bytes": algo[name], "GBps": algo[name] / ms / 1e6, "frac": algo[name] / ms / 1e6 / HBM_PEAK_GBS}
    line = {"metric": "GN iterations/sec (WLS state estimation, PMU + legacy, 9241-bus PEGASE-shaped grid)", "value": iters / dt,
            "unit": "GN iterations/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
            "ms_per_solve_batched": 1e3 * dt / (B * steps), "iterations_per_scenario": iters / (B * steps),
            "converged_fraction": float(np.mean(an.status == 0)), "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{case} Gauss-Newton WLS SE, {B} noisy realisations per batch, {len(handles)} batches in flight, "
                                   "flat start, tol 1e-8, max 40", "rows": d["m"],
                       "nnzH": d["nnzH"], "gain_blocks": d["gain_blocks"], "lu_blocks": d["lu_blocks"], "lu_terms": d["lu_terms"],
                       "factor_launches": d["factor_launches"], "backward_launches": d["backward_launches"]},
            "kernels": kern}
    if cpu:
        line["cpu_baseline"] = cpu_baseline_se(jg, s, case, pf, budget_s=cpu_budget_s)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    for h in handles:
        h.close()
    pf.close()
    return line


def make_gather(jg, torch, dist, rank, world, local, cdev, force_dist, lanes, width):
    """The ONE collective of a device batch.  Default for N > 1 on the GPU: the library's own C ABI (jg_comm_*: ncclAllGather of librccl,
    csrc/jg_comm.cpp) -- what a Julia host calls; rank 0 draws the communicator id and torch.distributed only ships its 128 bytes.  JG_BENCH_GATHER=torch
    (or a communicator that cannot be built: no librccl) falls back to torch.distributed.all_gather_into_tensor.  Returns (deliver(buf), label, close)."""
    if not (world > 1 or force_dist):
        return (lambda buf: None), "none (one rank)", (lambda: None)
    want = os.environ.get("JG_BENCH_GATHER", "abi" if world > 1 else "torch")
    comm = gathered = None
    note = ""
    if want == "abi" and cdev == "cuda":
        ok = torch.ones(1, dtype=torch.int32, device="cuda")
        try:
            uid = torch.from_numpy(jg._lib.Comm.unique_id() if rank == 0 else np.zeros(jg._lib.COMM_ID_BYTES, dtype=np.uint8)).cuda()
        except Exception as e:                        # librccl could not be bound on rank 0: every rank must take the same path
            uid = torch.zeros(jg._lib.COMM_ID_BYTES, dtype=torch.uint8, device="cuda"); ok[0] = 0; note = repr(e)
        dist.broadcast(uid, src=0)
        dist.broadcast(ok, src=0)
        if int(ok.item()):
            # RCCL through the C ABI has only ever met ONE rank on the build pool (one GPU per box): the communicator is built and tried -- one gather of the
            # real record size -- on a side thread with a deadline, so that a rendezvous that never completes costs the run its C-ABI gather, not the run
            import threading
            box = {}
            uid_host = uid.cpu().numpy()

            def build_and_try():
                try:
                    c = jg._lib.Comm(rank, world, uid_host, device=local)
                    torch.cuda.set_device(local)
                    probe = torch.zeros((lanes, width), dtype=torch.float64, device="cuda"); torch.cuda.current_stream().synchronize()   # (the fill runs on torch's stream, the gather on the communicator's)
                    out = torch.empty((world * lanes, width), dtype=torch.float64, device="cuda")
                    c.allgather_device(probe.data_ptr(), out.data_ptr(), probe.numel())
                    torch.cuda.synchronize()
                    box["comm"] = c
                except Exception as e:
                    box["error"] = repr(e)
            th = threading.Thread(target=build_and_try, daemon=True)
            th.start()
            th.join(float(os.environ.get("JG_BENCH_COMM_TIMEOUT", "120")))
            comm = box.get("comm")
            if comm is None:
                note = box.get("error", "communicator not ready after JG_BENCH_COMM_TIMEOUT seconds")
            good = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if not int(good.item()):
                if comm is not None:
                    comm.close()
                comm = None
        if comm is not None:
            gathered = torch.empty((world * lanes, width), dtype=torch.float64, device="cuda")

    def deliver(buf):
        if comm is not None:
            comm.allgather_device(buf.data_ptr(), gathered.data_ptr(), buf.numel())
        elif cdev == "cuda":
            g = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
            dist.all_gather_into_tensor(g, buf.contiguous())
            torch.cuda.current_stream().synchronize()
        else:
            b = buf.cpu()
            g = torch.empty((world * b.shape[0], b.shape[1]), dtype=b.dtype)
            dist.all_gather_into_tensor(g, b.contiguous())

    label = "abi" if comm is not None else ("torch.distributed" + (f" (C-ABI communicator unavailable: {note})" if want == "abi" and cdev == "cuda" and note else ""))
    return deliver, label, (lambda: comm.close() if comm is not None else None)


def device_state(index):
    """Clocks, power and temperature of the GPU as rocm-smi reports them (the same build measures 267k - 312k NR it/s from box to box; the line
    now records what the device was doing).  None when rocm-smi is missing or its output cannot be read; never raises."""
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "-d", str(index), "--showclocks", "--showpower", "--showtemp", "--showuse", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out[out.index("{"):])
        card = next(iter(j.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power", "temperature (sensor junction)", "temperature (sensor memory)", "gpu use")):
                keep[k] = v
        return keep
    except Exception as e:
        return {"error": repr(e)}


def timed_regions(torch, dist, world, force_dist, cdev, run, steps):
    """The K-step region (fenced on both sides, max over ranks), repeated until about a second has been timed; (regions, iterations of one region, last status)."""
    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def region():
        fence()
        t0 = time.perf_counter()
        it, st = run(steps)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, it, st

    dt0, iters_local, last_status = region()
    repeats = max(3, min(int(os.environ.get("JG_BENCH_MAX_REPEATS", "50")), int(np.ceil(float(os.environ.get("JG_BENCH_MIN_SECONDS", "1.0")) / max(dt0, 1e-6)))))
    regions = [dt0]
    for _ in range(repeats - 1):
        dtr, it_r, last_status = region()
        assert it_r == iters_local, "the same scenarios take the same iterations in every region"
        regions.append(dtr)
    return regions, iters_local, last_status


def live_pmc_traffic(case, batch, n):
    """roofline.traffic measured IN THIS RUN when rocprofv3 is on the box: two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE;
    kernel-trace only, as MI355X_MICROARCH.md prescribes) of tools/profile_kernels.py -- one handle of the bench's batch, two solves -- summarised by
    tools/pmc_summary.py (KiB units, FETCH_SIZE doubled on gfx950, calibrated on a device copy of known size).  None when rocprofv3 is missing, switched
    off (JG_BENCH_LIVE_PMC=0) or fails: the caller falls back to the committed profiles/pmc_traffic.json and says so."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("JG_BENCH_LIVE_PMC", "1") == "0" or not shutil.which("rocprofv3"):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_summary
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                r = subprocess.run(["rocprofv3", "--pmc", c, "--kernel-trace", "-d", os.path.join(td, c), "-o", "p", "--output-format", "csv", "--",
                                    sys.executable, os.path.join(ROOT, "tools", "profile_kernels.py"), str(batch), "2", case],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
                if r.returncode != 0:
                    return None

            def csv_of(c):
                for dp, _, fs in os.walk(os.path.join(td, c)):
                    for f in fs:
                        if f.endswith("counter_collection.csv"):
                            return os.path.join(dp, f)
                return None
            fc, wc = csv_of("FETCH_SIZE"), csv_of("WRITE_SIZE")
            if not fc or not wc:
                return None
            out = os.path.join(td, "pmc.json")
            real = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                pmc_summary.main(fc, wc, n, batch, 2, out, case)
            finally:
                sys.stdout.close(); sys.stdout = real
            return json.load(open(out))
    except Exception:
        return None


def live_pmc_traffic_se(batch):
    """The same for the state-estimation kernels: two `rocprofv3 --pmc` passes of tools/profile_se.py (config 4, ONE handle, three increments), summarised by
    tools/pmc_se_summary.py into bytes per Gauss-Newton increment: {"rows", "gain", "factor", "backward"} or None."""
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("JG_BENCH_LIVE_PMC", "1") == "0" or not shutil.which("rocprofv3"):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_se_summary
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            found = {}
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                r = subprocess.run(["rocprofv3", "--pmc", c, "--kernel-trace", "-d", os.path.join(td, c), "-o", "p", "--output-format", "csv", "--",
                                    sys.executable, os.path.join(ROOT, "tools", "profile_se.py"), str(batch), "2"],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
                if r.returncode != 0:
                    return None
                for dp, _, fs in os.walk(os.path.join(td, c)):
                    for f in fs:
                        if f.endswith("counter_collection.csv"):
                            found[c] = os.path.join(dp, f)
            if len(found) != 2:
                return None
            out = os.path.join(td, "pmc.json")
            real = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                pmc_se_summary.main(found["FETCH_SIZE"], found["WRITE_SIZE"], 2, out)
            finally:
                sys.stdout.close(); sys.stdout = real
            t = json.load(open(out))["traffic"]
            tot = lambda k: float(t[k]["fetch_bytes"] + t[k]["write_bytes"]) if k in t else None
            return {"rows": tot("k_gn_rows"), "gain": tot("k_gn_gain"), "factor": tot("factor"), "backward": tot("k_bwd_level")}
    except Exception:
        return None


def predicted_from_shards(workload, world, total):
    """What ONE rank's share of this N-GPU run does on one GPU (profiles/bench_shards.json, written by tools/run_evidence.sh on the last box that measured it):
    N x that rate is the strong-scaling prediction the line carries beside the measurement (the pool has one GPU per box: the 1 -> 8 curve itself
    is the driver's to measure)."""
    path = os.path.join(ROOT, "profiles", "bench_shards.json")
    if world < 2 or not os.path.exists(path):
        return None
    try:
        rows = [r for r in json.load(open(path)).get(workload, []) if r.get("merged") and r["scenarios_per_step"] * world == total]
        if not rows:
            return None
        r = rows[-1]
        out = {"value": world * r["value"], "per_gpu_value": r["value"], "from": "profiles/bench_shards.json",
               "what": f"{world} x the rate ONE GPU reaches on a rank's share ({r['scenarios_per_step']} scenarios per step, {r['steps_per_device_batch']} steps per device batch, "
                       f"{r['device_batches_in_flight']} in flight" + (f", K = {r['steps']} steps per region as the driver runs it" if r.get("steps") else "") +
                       f"), measured {r.get('measured', 'earlier')}; no gather, no host contention"}
        if r.get("value_steady"):
            out["value_steady"] = world * r["value_steady"]
        return out
    except Exception:
        return None


def workload_se(jg, torch, dist, args, rank, local, world, cdev, force_dist):
    """--workload se: BASELINE config 4 as a sharded Monte-Carlo run.  One step = `--batch` noisy realisations (strong scaling: in total) of the config-4
    measurement set, each estimated from the flat start by stateEstimation! (tol 1e-8, max 40); a rank estimates its contiguous share, a device batch packs
    its record (magnitude | angle | iterations | status | objective per realisation, jg_gn_pack_results_device) and ONE all-gather carries it."""
    case = args.case if args.case != "case_ACTIVSg10k" else "case9241synth"
    s = jg.powerSystem(load_tables(jg, case))
    pf = jg.newtonRaphson(s, device=local)
    jg.powerFlow_(pf, tolerance=1e-11)
    assert pf.status == 0
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf, variance=1e-4)
    jg.addWattmeter_(mon, pf, variance=1e-4)
    jg.addVarmeter_(mon, pf, variance=1e-4)
    jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
    n = s.bus.number
    total = args.batch if args.scaling == "strong" else args.batch * world
    lo, hi = jg.shard(total, rank, world)
    B = hi - lo
    if B < 1:
        raise SystemExit(f"rank {rank}: no realisations ({total} over {world} ranks)")
    b_max = -(-total // world)
    merge = args.merge if args.merge > 0 else (jg.deviceBatching(b_max, args.steps, 512) if args.scaling == "strong" else 1)
    merge = max(1, min(merge, args.steps))
    lanes = B * merge
    inflight = args.inflight if args.inflight > 0 else 2
    t0 = time.perf_counter()
    pipe = jg.MonteCarloPipeline(mon, lanes, inflight=inflight, device=local)
    t_pipe = time.perf_counter() - t0
    # every device batch estimates FRESH realisations, drawn on the device inside the timed region (jg_gn_draw_noise: seed 4, realisation ids unique over jobs and ranks:
    # what a Monte-Carlo study does; 0.8 GB of se.mean / se.precision rewritten per 512 realisations, ~1 % of a step); JG_BENCH_SE_RESIDENT=1: the realisations of
    # round 4's config4_se leg stay resident instead (drawn once, on the device)
    resident = bool(os.environ.get("JG_BENCH_SE_RESIDENT"))
    for k, h in enumerate(pipe.handles):
        jg.drawNoise_(h, 4, scale=1.0, first=(k * world + rank) * lanes)
    an = pipe.handles[0]
    width = pipe.record_width
    ring = len(pipe.handles)
    packed = [torch.empty((lanes, width), dtype=torch.float64, device="cuda") for _ in range(ring)]
    gather, gather_label, gather_close = make_gather(jg, torch, dist, rank, world, local, cdev, force_dist, lanes, width)

    def run(steps):
        jobs = -(-steps // merge)
        work = [None] * jobs if resident else [(4, (j * world + rank) * lanes) for j in range(jobs)]
        out = pipe.run(work, iteration=40, tolerance=1e-8, on_done=lambda j, h: gather(packed[j % ring]),
                       record=lambda j: packed[j % ring].data_ptr(), records=ring)
        real = [min(merge, steps - j * merge) * B for j in range(jobs)]
        return int(sum(int(np.sum(it[:r])) for (it, _), r in zip(out, real))), out[-1][1][:B]

    run(args.warmup)
    regions, iters_local, last_status = timed_regions(torch, dist, world, force_dist, cdev, run, args.steps)
    dt = float(np.median(regions))
    jobs_per_region = -(-args.steps // merge)
    steady = jobs_per_region >= 3 * len(pipe.handles)
    steady_extra = None
    if not steady:                                    # K as given measures fill + drain of the pipeline: ALSO a region long enough for its steady state
        ks = 3 * len(pipe.handles) * merge
        rs, it_s, _ = timed_regions(torch, dist, world, force_dist, cdev, run, ks)
        steady_extra = (ks, float(np.median(rs)), it_s)
    conv_local = int(np.sum(last_status == 0))
    counts = [iters_local, conv_local] + ([steady_extra[2]] if steady_extra else [])
    if world > 1:
        cnt = torch.tensor(counts, dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        counts = [int(x) for x in cnt.tolist()]
    line = None
    if rank == 0:
        d = an.dims
        L = an.batch
        algo = {"rows": L * (16 * d["slots"] + 16 * d["m"] + 16 * n), "gain": L * (16 * d["slots"] + 8 * d["m"] + 32 * d["gain_blocks"] + 16 * n),
                "factor": L * (64 * ((d["lu_blocks"] + n) // 2)), "backward": L * (32 * ((d["lu_blocks"] + n) // 2) + 64 * n)}
        kern = {}
        for k, name in enumerate(("rows", "gain", "factor", "backward")):
            ms = float(np.median([an.time_kernel(k, 6 if name == "factor" else 12) for _ in range(5)]))
            kern[name] = {"ms": ms, "bytes": algo[name], "GBps": algo[name] / ms / 1e6, "frac": algo[name] / ms / 1e6 / HBM_PEAK_GBS}
        dom = max(kern, key=lambda k: kern[k]["ms"])
        traffic = traffic_source = None
        live = live_pmc_traffic_se(L) if (world == 1 and not args.no_cpu and case == "case9241synth") else None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic_se.json")
        try:
            per = None
            if live is not None:
                per, traffic_source = live, "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_se.py inside this bench run"
            elif os.path.exists(pmc):
                pj = json.load(open(pmc))
                if int(pj.get("batch_ld", 0)) == L and pj.get("grid") == case:
                    per, traffic_source = pj["traffic_per_increment"], "profiles/pmc_traffic_se.json (committed: the same two passes on an earlier box)"
            if per:
                for kk in kern:
                    kern[kk]["hbm_traffic_bytes_pmc"] = per.get(kk)
                traffic = per.get(dom)
        except Exception:
            traffic = None
        nsc = total * args.steps
        line = {
            "metric": "GN iterations/sec (sharded Monte-Carlo WLS state estimation, PMU + legacy, 9241-bus PEGASE-shaped grid)",
            "value": counts[0] / dt, "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "region_repeats": len(regions), "region_ms_min": 1e3 * float(np.min(regions)), "region_ms_median": 1e3 * dt, "region_ms_max": 1e3 * float(np.max(regions)),
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{case} Gauss-Newton WLS state estimation (BASELINE config 4), {total} noisy realisations per step ({B} per GPU), flat start, tol 1e-8, max 40",
                       "grid": case, "buses": n, "rows": d["m"], "nnzH": d["nnzH"], "gain_blocks": d["gain_blocks"], "lu_blocks": d["lu_blocks"], "lu_terms": d["lu_terms"],
                       "factor_launches": d["factor_launches"], "backward_launches": d["backward_launches"],
                       "batch_per_gpu": B, "scenarios_per_step": total, "steps_per_device_batch": merge, "lanes_per_device_batch": lanes,
                       "device_batches_in_flight_per_gpu": len(pipe.handles), "device_batches_per_region": jobs_per_region, "pipeline_steady_state": bool(steady),
                       "gather": gather_label, "record": f"magnitude | angle | iterations | status | objective, 2 n + 3 = {width} doubles per realisation",
                       "realisations": "resident (drawn once on the device)" if resident else "fresh per device batch, drawn on the device inside the timed region (jg_gn_draw_noise)",
                       "parallelism": f"realisation-sharded x{world}, one all-gather of the packed record per device batch"},
            "scenarios_per_s": nsc / dt, "ms_per_solve_batched": 1e3 * dt / nsc, "iterations_per_scenario": counts[0] / nsc,
            "converged_fraction": counts[1] / total, "pipeline_construction_ms": 1e3 * t_pipe,
            "roofline": {"bound": "hbm", "kernel": {"rows": "k_gn_rows", "gain": "k_gn_gain", "factor": "k_fact_task + k_fact_top (symmetric plan)", "backward": "k_bwd_level"}[dom],
                         "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic,
                         "traffic_source": traffic_source if traffic is not None else None, "algorithmic_bytes": kern[dom]["bytes"]},
            "kernels": kern,
        }
        if steady_extra:
            ks, dts, _ = steady_extra
            line["value_steady"] = counts[2] / dts
            line["steady_steps"] = ks
            line["steady_ms_per_step"] = 1e3 * dts / ks
            line["steady_what"] = (f"the K = {args.steps}-step region is {jobs_per_region} device batch(es) with {len(pipe.handles)} in flight (fill + drain); value_steady is the same "
                                   f"measurement over {ks} steps (three rounds of the batches in flight).  value, ms_per_step: the K of the caller, unchanged")
        pred = predicted_from_shards("se", world, total)
        if pred:
            line["predicted"] = pred
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline_se(jg, s, case, pf, budget_s=10.0)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    gather_close()
    pipe.close()
    pf.close()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)       # 1.4 s of timed region at 512 scenarios per step
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="scenarios per step (strong scaling: in total; weak: per GPU)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--case", default="case_ACTIVSg10k", help="case_ACTIVSg10k (the metric's 10k-bus grid) | case9241synth | any fixture")
    ap.add_argument("--inflight", type=int, default=0, help="steps in flight per GPU (0: 3 at 512 scenarios per GPU, more for smaller shards)")
    ap.add_argument("--merge", type=int, default=0, help="steps solved together in one device batch per GPU (0: as many as bring a shard back to "
                    "512 lanes under strong scaling, i.e. 8 at 64 scenarios per GPU; 1: every step its own device batch)")
    ap.add_argument("--pool", type=int, default=256, help="lanes of the straggler pool (0: every batch finishes its own stragglers in lockstep)")
    ap.add_argument("--record", choices=("state", "summary"), default="state",
                    help="what a device batch delivers and a sharded run gathers per scenario: the state record V | theta | iterations | status "
                         "(2 n + 2 doubles: SURVEY 8(e), the default) or the contingency screen summary (10 doubles: worst loading, largest flow, "
                         "voltage extremes, iterations, status -- jg_nr_screen)")
    ap.add_argument("--workload", choices=("nr", "se"), default="nr",
                    help="nr: batched N-1 Newton-Raphson (the headline metric); se: BASELINE config 4 as a sharded Monte-Carlo run -- `--batch` noisy realisations "
                         "per step of the PMU + legacy measurement set on the 9241-bus grid, Gauss-Newton WLS, the same line shape (GN iterations/s)")
    ap.add_argument("--full-refactor", action="store_true", help="no base case: the first iteration of every batch refactorises like the others (the path of rounds 1-5)")
    ap.add_argument("--top-cap", type=int, default=0, help="pivots in the dense top of the shared-factor sweeps (0: the library's default, < 0: none)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-se", action="store_true", help="skip the config4_se object")
    args = ap.parse_args()

    import torch
    import juliagrid.jl_amd as jg
    os.environ.setdefault("JG_ORACLE_FAST", "1")      # the cpu_baseline legs (and only they) run the -O3 -march=native build of the oracle

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: run it the way the driver runs it -- one rank per GPU under torch.distributed.run on
        # this node -- and hand its output and exit code through (rank 0 prints the one JSON line)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # stdout carries the ONE JSON line and nothing else: whatever a library writes to file descriptor 1 while the bench runs (RCCL prints a version
    # banner through C stdio when a communicator is born) goes to stderr; the real stdout comes back for the line at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # JG_BENCH_BACKEND=gloo: dry run of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share the
    # devices round-robin, collectives go through host memory).  The measured configuration is always nccl = RCCL.
    backend = os.environ.get("JG_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    cdev = "cuda" if backend == "nccl" else "cpu"      # where the tensors of the collectives live
    torch.cuda.set_device(local)
    dist = None
    force_dist = world == 1 and bool(os.environ.get("JG_BENCH_FORCE_DIST"))   # probe: a 1-rank RCCL group, so that the collective path
    if force_dist:                                                             # (init, all_gather_into_tensor on device records) runs on a 1-GPU box
        os.environ.setdefault("MASTER_PORT", "29531"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    if args.workload == "se":
        line = workload_se(jg, torch, dist, args, rank, local, world, cdev, force_dist)
        if world > 1 or force_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            import ctypes
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            os.dup2(real_stdout, 1)
            print(json.dumps(line), flush=True)
        return

    tables = load_tables(jg, args.case)

    # ---- setup (untimed): base case, scenario list, shard, upload ---------------------------
    system = jg.powerSystem(tables)
    t0 = time.perf_counter()
    base = jg.newtonRaphson(system, batch=1, device=local)             # symbolic analysis of the block LU, tables, upload
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    jg.powerFlow_(base)                                                  # first solve: the hipGraphs are captured here
    t_first = time.perf_counter() - t0
    assert base.status == 0
    base_iters = int(base.method.iteration)
    vm0, va0 = base.voltage.magnitude.copy(), base.voltage.angle.copy()
    screen_rating = None
    if args.record == "summary":
        jg.power_(base)
        sf = np.hypot(base.power.from_.active, base.power.from_.reactive)
        st_ = np.hypot(base.power.to.active, base.power.to.reactive)
        screen_rating = np.maximum(1.2 * np.maximum(sf, st_), 0.05)
    # single-instance latency (BASELINE config 2/3 "ms/solve"): the SAME handle again from the case's start point (warm: what the
    # reference's reused analysis is); what a first solve pays on top is reported as setup_ms
    t_single = []
    for _ in range(7):
        jg.setInitialPoint_(base)
        t0 = time.perf_counter()
        jg.powerFlow_(base, fetch=False)
        t_single.append(time.perf_counter() - t0)
    base_jacobian = base.jacobian if (rank == 0 and not args.no_cpu) else None     # for the splu leg (CSC of the reference, base state)
    base.close()
    # the same solve through the level launches of a small batch (JG_SINGLE=0: what a single instance ran until round 6 -- a wave with one live lane per item)
    t_single_levels = []
    if rank == 0:
        os.environ["JG_SINGLE"] = "0"
        try:
            lv = jg.newtonRaphson(jg.powerSystem(tables), batch=1, device=local)
            jg.powerFlow_(lv)
            for _ in range(5):
                jg.setInitialPoint_(lv)
                t0 = time.perf_counter()
                jg.powerFlow_(lv, fetch=False)
                t_single_levels.append(time.perf_counter() - t0)
            lv.close()
        except Exception:
            t_single_levels = []
        finally:
            del os.environ["JG_SINGLE"]
    # The first handle of a process also pays the HIP context and the load of the library's code object.  What ONE MORE analysis costs in a
    # warm process (what a reference user pays per newtonRaphson() call) is measured twice: of the SAME grid while the library still holds
    # its plan (engines of one pattern share the symbolic analysis and the device tables), and after jg_plan_cache_clear (a full analysis):
    def prebuilt():                                  # a system whose AC model exists (acModel! is not part of what is compared: the CPU leg's
        ps = jg.powerSystem(tables)                  # OracleSystem holds its Ybus as well)
        jg.acModel_(ps)
        return ps

    s1 = prebuilt()
    t0 = time.perf_counter()
    again = jg.newtonRaphson(s1, batch=1, device=local)
    t_create_cached = time.perf_counter() - t0
    t0 = time.perf_counter()
    jg.powerFlow_(again)
    t_first_cached = time.perf_counter() - t0
    again.close()
    jg._lib.lib().jg_plan_cache_clear()
    s2 = prebuilt()
    t0 = time.perf_counter()
    again = jg.newtonRaphson(s2, batch=1, device=local)
    t_create2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    jg.powerFlow_(again)
    t_first2 = time.perf_counter() - t0
    again.close()

    # Scenario selection (untimed, identical on every rank): the first scenarios of a seeded shuffle of the non-bridge branches
    # THAT HAVE A POWER FLOW.  A contingency without a solution runs to the iteration limit (20 iterations for one lane while the
    # other lanes of its batch have long finished), so one of them in the list would measure that scenario, not the path.  They
    # are counted and reported, not hidden: the first 1024 candidates of case_ACTIVSg10k hold exactly one (branch 11127).
    total = args.batch if args.scaling == "strong" else args.batch * world
    lo, hi = jg.shard(total, rank, world)
    B = hi - lo                                                          # scenarios of this rank per step
    if B < 1:
        raise SystemExit(f"rank {rank}: no scenarios ({total} scenarios over {world} ranks)")
    # Device batch: under strong scaling a rank's share of a step shrinks with N (64 scenarios at N = 8), and the path is at its best
    # around 512 scenarios per launch.  The scenarios of a job are independent, so a rank solves its shares of M
    # consecutive steps together in ONE handle of M x B lanes (M = 512 // B): the same K x `total` scenarios are solved inside the
    # timed region, in wider launches, and the ONE collective then carries the records of M steps.  --merge 1 switches it off.
    b_max = -(-total // world)
    if args.merge > 0:
        merge = args.merge
    elif args.scaling == "strong":
        # device batches of the width the grid's size asks for (512 lanes on the 10 000-bus grid, up to 4 096 on small ones: contingency.recommendedLanes) with the
        # fewest spare lanes in the last one; a run of at most 1.25 (2.5) such batches per rank: one (two) wider batches
        merge = jg.deviceBatching(b_max, args.steps, jg.recommendedLanes(system.bus.number))
    else:
        merge = 1
    merge = max(1, min(merge, args.steps))
    lanes = B * merge
    inflight = args.inflight if args.inflight > 0 else max(3, min(12, 1536 // lanes))
    #  the timed region is a SCREEN: its K steps solve K x `total` DISTINCT outages -- the seeded shuffle of every non-bridge branch, taken in
    # order and cycled only when the grid has fewer solvable candidates than the region asks for -- and every device batch uploads its own outages inside the
    # region (jg_nr_patch_ybus_batch: what updateBranch!(...; status = 0) is per scenario of the reference's loop, branch.jl:453-459)
    cand = jg.outageList(system, system.branch.number, seed=512)
    cand = cand[np.sort(np.unique(cand, return_index=True)[1])]          # every candidate once, in shuffle order (outageList tiles a short list)
    t0 = time.perf_counter()
    pipe = jg.ContingencyPipeline(system, lanes, inflight=inflight, device=local, start=(vm0, va0), pool=args.pool,
                                  shared_first=not args.full_refactor, top_cap=args.top_cap)
    t_pipe = time.perf_counter() - t0                  # handles + pools: ONE symbolic analysis (shared plan), device storage, start point, the base case's factor
    it_pre, st_pre = pipe.screen(cand, iteration=20, tolerance=1e-8)
    solvable = np.flatnonzero(st_pre == 0)
    if os.environ.get("JG_BENCH_PROBE_UNIFORM"):        # probe only: scenarios that all need the same number of iterations (no stragglers)
        solvable = np.flatnonzero((st_pre == 0) & (it_pre == int(os.environ["JG_BENCH_PROBE_UNIFORM"])))
    if solvable.size < 1:
        raise SystemExit(f"none of the {cand.size} candidate contingencies has a power flow")
    excluded = int(cand.size - solvable.size)                  # (a step larger than the list -- small grids at large batches -- cycles it: scenario_selection says so)
    chosen = cand[solvable]                           # the screen's list: every solvable non-bridge outage, shuffle order

    def step_labels(k):                               # this rank's share of step k of a region
        idx = (k * total + np.arange(lo, hi)) % chosen.size
        return chosen[idx]

    def batch_labels(job):                            # lane m * B + s = scenario s of step job * merge + m
        return np.concatenate([step_labels(job * merge + m) for m in range(merge)])

    labels = batch_labels(0)
    for h in pipe.handles:
        jg.setOutages_(h, labels)
    an = pipe.handles[0]
    n = system.bus.number
    # result records: a ring the pipeline fills (the batch's own scenarios when its main phase ends, its stragglers when their pool
    # has finished them); a record is reused only after its job has been delivered
    ring = len(pipe.handles) + (12 if pipe.pools else 0)
    summary = args.record == "summary"
    width = 10 if summary else 2 * n + 2
    if summary:                                       # ratings of the screen: the case holds none -- 1.2 x the base-case flow of every branch, floor 0.05 pu
        pipe.setRating(screen_rating)
    packed = [torch.empty((lanes, width), dtype=torch.float64, device="cuda") for _ in range(ring)]

    # the ONE collective of a device batch (make_gather: the library's own C ABI by default for N > 1, torch.distributed as the fallback)
    gather, gather_label, gather_close = make_gather(jg, torch, dist, rank, world, local, cdev, force_dist, lanes, width)

    gather_times = []

    def deliver(job, h):                              # caller's thread, job order: the record is complete -> the ONE collective
        tg = time.perf_counter()
        gather(packed[job % ring])
        gather_times.append(time.perf_counter() - tg)

    def run(steps):
        jobs = -(-steps // merge)                     # device batches; the last one may hold fewer real steps: its spare lanes are
        out = pipe.run([batch_labels(j) for j in range(jobs)], iteration=20, tolerance=1e-8, on_done=deliver,      # solved (and timed) but not counted
                       record=lambda j: packed[j % ring].data_ptr(), records=ring, summary=summary)
        real = [min(merge, steps - j * merge) * B for j in range(jobs)]
        return int(sum(int(np.sum(it[:r])) for (it, _), r in zip(out, real))), out[-1][1][:B]

    run(args.warmup)
    # The timed region is K steps as given.  At the driver's K = 20 that is 0.12 s at N = 1 and ~15 ms at N = 8: one region is
    # mostly the fill and drain of the batches in flight plus whatever the box does in that instant.  So the SAME region -- K steps, fenced on
    # both sides, max over ranks -- is repeated until about a second has been timed (every rank takes the count from the max-reduced first
    # region: same number of collectives everywhere) and the line reports the MEDIAN region; min / max travel with it (timed_regions).
    dev_state0 = device_state(local) if rank == 0 else None
    regions, iters_local, last_status = timed_regions(torch, dist, world, force_dist, cdev, run, args.steps)
    dev_state1 = device_state(local) if rank == 0 else None
    dt = float(np.median(regions))
    #  when the K-step region cannot reach the pipeline's steady state (N = 8 at K = 20: 4 device batches, all in flight at once) the line
    # ALSO carries the same measurement over three rounds of the batches in flight -- value_steady; value stays the K of the caller
    jobs_per_region = -(-args.steps // merge)
    steady = jobs_per_region >= 3 * len(pipe.handles)
    steady_extra = None
    if not steady:
        ks = 3 * len(pipe.handles) * merge
        rs, it_s, _ = timed_regions(torch, dist, world, force_dist, cdev, run, ks)
        steady_extra = [ks, float(np.median(rs)), it_s]
        if world > 1:
            c2 = torch.tensor([it_s], dtype=torch.int64, device=cdev)
            dist.all_reduce(c2, op=dist.ReduceOp.SUM)
            steady_extra[2] = int(c2.item())

    #  the same regions with the first iteration refactorising like the others -- today's path against the path of rounds 1-5, same run, same box
    full_extra = None
    first_counts = [0, 0]
    if pipe.base is not None:
        for h in pipe.handles:
            c = jg.firstIterationCounts(h)
            first_counts[0] += c[0]; first_counts[1] += c[1]
        pipe.setFirstIteration(False)
        run(min(args.steps, 3 * len(pipe.handles) * merge))
        rf, it_f, _ = timed_regions(torch, dist, world, force_dist, cdev, run, args.steps)
        pipe.setFirstIteration(True)
        full_extra = [float(np.median(rf)), it_f]
        if world > 1:
            c3 = torch.tensor([it_f], dtype=torch.int64, device=cdev)
            dist.all_reduce(c3, op=dist.ReduceOp.SUM)
            full_extra[1] = int(c3.item())
    gather_ms = [1e3 * x for x in gather_times]

    conv_local = int(np.sum(last_status == 0))
    if world > 1:
        cnt = torch.tensor([iters_local, conv_local], dtype=torch.int64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        iters_total, conv_total = int(cnt[0].item()), int(cnt[1].item())
    else:
        iters_total, conv_total = iters_local, conv_local

    if rank == 0:
        d = an.dims
        ab = algorithmic_bytes(d, an.batch)
        # live kernel timing with HIP events on the library's own stream (one handle, nothing else in flight, all scenarios active)
        # (median of five event-bracketed groups of launches: one stalled group -- seen once in ~30 runs, an 11 ms gap inside a
        # group of 10 backward sweeps -- must not become the per-launch figure)
        def timed(kernel, reps):
            return float(np.median([an.time_kernel(kernel, reps) for _ in range(5)]))
        # (repetitions: the first launches of a group run on a chip that has just been idle -- 4 / 8 / 20 assemblies per group measure
        # 0.156 / 0.144 / 0.137 ms per launch on the same handle; the groups are long enough that this start-up does not set the figure)
        t_asm = timed(0, 24)
        t_lu = timed(1, 8)
        t_sol = timed(2, 12)
        kern = {
            "assembly": {"ms": t_asm, "bytes": ab["assembly"], "launches": 1},
            "lu": {"ms": t_lu, "bytes": ab["lu"], "launches": d["lu_launches"]},
            "solve": {"ms": t_sol, "bytes": ab["solve"], "launches": d["solve_launches"]},
        }
        first_kern = None
        if pipe.base is not None:
            binfo = pipe.base.info
            t_first_lin = timed(4, 12)
            t_mis = timed(5, 24)
            # algorithmic bytes of the shared-factor step: the right-hand side read once and the solution written once per scenario (the factor is shared:
            # 2.2 MB per batch), the state read and written by the fused update; of the mismatch pass: SURVEY 8(d)'s assembly figure without the Jacobian stream
            nn, bb = d["n"], an.batch
            first_kern = {"shared_factor_step": {"ms": t_first_lin, "bytes": bb * (16 * nn + 16 * nn + 32 * nn) + 32 * d["lu_blocks"],
                                                 "launches": binfo["forward_launches"] + binfo["backward_launches"] + 2},
                          "mismatch_pass": {"ms": t_mis, "bytes": bb * (16 * nn + 32 * nn), "launches": 1}}
            for k in first_kern.values():
                k["GBps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9
                k["frac"] = k["GBps"] / HBM_PEAK_GBS
        for k in kern.values():
            k["GBps"] = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            k["frac"] = k["GBps"] / HBM_PEAK_GBS
        dom = max(kern, key=lambda k: kern[k]["ms"])
        names = {"assembly": "k_assemble", "lu": "k_fact_task + k_fact_top (one factorisation: task launches of the bottom levels -- k_fact_level where a plan keeps wave records -- and the multifrontal top)", "solve": "k_bwd_level (one backward sweep)"}
        # HBM bytes per logical launch from the committed PMC passes (profiles/pmc_traffic.json: FETCH_SIZE x2 +
        # WRITE_SIZE, calibrated on a kernel of known byte count); only valid for the grid and batch it was collected at
        traffic = None
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        pj_live = live_pmc_traffic(args.case, an.batch, n) if (world == 1 and not args.no_cpu) else None
        if pj_live is not None or os.path.exists(pmc):
            try:
                pj = pj_live if pj_live is not None else json.load(open(pmc))
                traffic_source = ("two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/profile_kernels.py inside this bench run" if pj_live is not None
                                  else "profiles/pmc_traffic.json (committed: the same two passes on an earlier box; rocprofv3 not run in this bench run)")
                if int(pj.get("batch_ld", 0)) == an.batch and pj.get("grid") == args.case:
                    key = {"assembly": "k_assemble", "lu": "k_fact", "solve": "k_fwd+k_bwd"}[dom]
                    traffic = pj["traffic_per_logical_launch"].get(key)
                    for kk, nm in (("assembly", "k_assemble"), ("lu", "k_fact"), ("solve", "k_fwd+k_bwd")):
                        tr = pj["traffic_per_logical_launch"].get(nm)
                        kern[kk]["hbm_traffic_bytes_pmc"] = tr
                        if tr is not None:            # 8(d)'s sanity rule: the counters cannot be below the algorithmic bytes
                            kern[kk]["pmc_at_least_algorithmic"] = bool(tr >= 0.98 * kern[kk]["bytes"])
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
                    "per_launch_ms": kern[dom]["ms"] / kern[dom]["launches"], "launches": kern[dom]["launches"],
                    "algorithmic_bytes": kern[dom]["bytes"]}
        nsc = total * args.steps
        line = {
            "metric": "NR iterations/sec (batched N-1 AC power flow, 10k-bus grid)",
            "value": iters_total / dt, "unit": "NR iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "region_repeats": len(regions), "region_ms_min": 1e3 * float(np.min(regions)), "region_ms_median": 1e3 * dt, "region_ms_max": 1e3 * float(np.max(regions)),
            "region_what": f"the K = {args.steps}-step region (fenced on both sides, max over ranks) repeated {len(regions)} times; value and ms_per_step are its MEDIAN",
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.case} batched N-1 contingency Newton-Raphson, {total} scenarios per step "
                                   f"({B} per GPU), start = base-case solution, tol 1e-8, max 20 iterations",
                       "grid": args.case, "buses": n, "batch_per_gpu": B, "scenarios_per_step": total, "dimJ": d["dimJ"], "nnzJ": d["nnzJ"],
                       "lu_blocks_2x2": d["lu_blocks"], "lu_terms": d["lu_terms"],
                       "launches_per_iteration": 2 + d["lu_launches"] + d["solve_launches"],
                       "steps_per_device_batch": merge, "lanes_per_device_batch": lanes,
                       "device_batches_in_flight_per_gpu": len(pipe.handles),
                       "straggler_pool_lanes": pipe.pools[0].handle.batch if pipe.pools else 0,
                       "steps_in_flight_per_gpu": len(pipe.handles) * merge,
                       "device_batches_per_region": -(-args.steps // merge),
                       "pipeline_steady_state": bool(-(-args.steps // merge) >= 3 * len(pipe.handles)),
                       "pipeline_note": ("a region holds at least three rounds of the batches in flight" if -(-args.steps // merge) >= 3 * len(pipe.handles) else
                                         f"a region is {-(-args.steps // merge)} device batch(es) with {len(pipe.handles)} in flight: it measures the fill and drain of the pipeline, "
                                         "not its steady state -- more --steps per region (or --merge 1 for narrower batches) changes that, the K of the caller is kept as given"),
                       "gather": gather_label,
                       "record": ("screen summary, 10 doubles per scenario (jg_nr_screen: worst loading against 1.2 x the base-case flows, largest flow, voltage extremes, "
                                  "iterations, status), reduced on the device by the handle that finished the scenario") if summary else
                                 f"state record, 2 n + 2 = {2 * n + 2} doubles per scenario (V | theta | iterations | status)",
                       "parallelism": f"scenario-sharded x{world}, one RCCL all-gather of the packed results per device batch "
                                      f"({merge} step(s) of {B} scenarios per GPU)",
                       "scenario_selection": (f"a region's K x {total} = {args.steps * total} scenarios are taken in order from the seeded shuffle of ALL {chosen.size} solvable non-bridge "
                                              f"outages of the grid ({excluded} candidate(s) without a power flow skipped), uploaded per device batch INSIDE the region; "
                                              + ("every scenario of a region is a different outage" if args.steps * total <= chosen.size else
                                                 f"the list is cycled: {args.steps * total - chosen.size} scenario(s) of a region repeat an earlier outage"))},
            "scenarios_per_s": nsc / dt,
            "ms_per_solve_batched": 1e3 * dt / nsc,
            "iterations_per_scenario": iters_total / nsc,
            "converged_fraction": conv_total / total,
            "single_instance": {"ms_per_solve": 1e3 * float(np.median(t_single)), "iterations": base_iters,
                                "ms_per_iteration": 1e3 * float(np.median(t_single)) / max(base_iters, 1),
                                "what": "warm solves of ONE handle from the case's start point (setInitialPoint! between them, outside the clock): kernels whose lanes are items of the one "
                                        "scenario, the whole solve as one hipGraph from the second solve on (DESIGN.md 3.5)",
                                "ms_per_solve_level_launches": 1e3 * float(np.median(t_single_levels)) if t_single_levels else None,
                                "level_launches_what": "the same warm solve with JG_SINGLE=0: the level launches of a small batch (a wave per item, one live lane) instead of the "
                                                       "item-per-lane / row-per-lane kernels of ONE scenario (k_fact1_*, k_bwd1_*), same run",
                                "setup_ms": 1e3 * (t_create2 + t_first2) - 1e3 * float(np.median(t_single)),
                                "setup_what": f"a further analysis in a warm process with the library's plan cache emptied, on a system whose AC model exists: newtonRaphson() {1e3 * t_create2:.1f} ms (reference maps, "
                                              f"symbolic analysis of the block LU, replay tables, upload) + first powerFlow!() {1e3 * t_first2:.1f} ms (hipGraph capture) - one warm solve",
                                "setup_cached_ms": 1e3 * (t_create_cached + t_first_cached) - 1e3 * float(np.median(t_single)),
                                "setup_cached_what": f"the same while the plan of this grid is cached (second handle of a grid): newtonRaphson() {1e3 * t_create_cached:.1f} ms "
                                                     f"+ first powerFlow!() {1e3 * t_first_cached:.1f} ms - one warm solve",
                                "pipeline_construction_ms": 1e3 * t_pipe,
                                "setup_first_in_process_ms": 1e3 * (t_create + t_first) - 1e3 * float(np.median(t_single)),
                                "setup_first_what": f"the first analysis of the process (HIP context, code object load on top): newtonRaphson() {1e3 * t_create:.1f} ms "
                                                    f"+ first powerFlow!() {1e3 * t_first:.1f} ms - one warm solve"},
            "roofline": roofline,
            "kernels": kern,
        }
        # what a step costs by its kernels alone (each timed alone on one handle, all lanes active) against what the pipeline delivers: a batch of mean m iterations
        # runs m + 1 assemblies and m factorisations + sweeps when every iteration refactorises; with the shared first iteration a mismatch pass, the shared-factor
        # step, m assemblies and m - 1 factorisations + sweeps
        m_it = iters_total / nsc
        ksum_full = (m_it + 1) * t_asm + m_it * (t_lu + t_sol)
        if first_kern is not None:
            ksum = first_kern["mismatch_pass"]["ms"] + first_kern["shared_factor_step"]["ms"] + m_it * t_asm + (m_it - 1) * (t_lu + t_sol)
            line["kernels_first_iteration"] = first_kern
        else:
            ksum = ksum_full
        line["kernel_sum_ms"] = ksum / max(merge, 1)      # (the kernels were timed on a handle of B x merge lanes: `merge` steps of this rank)
        # the same weighting for the ALGORITHMIC bytes of a step (SURVEY 8(d) per kernel): what the whole path moves per step against what the step takes
        by_full = (m_it + 1) * ab["assembly"] + m_it * (ab["lu"] + ab["solve"])
        by = (first_kern["mismatch_pass"]["bytes"] + first_kern["shared_factor_step"]["bytes"] + m_it * ab["assembly"] + (m_it - 1) * (ab["lu"] + ab["solve"])) if first_kern is not None else by_full
        line["path_roofline"] = {"algorithmic_bytes_per_step": by / max(merge, 1) * world, "achieved_GBps": by / max(merge, 1) * world / (line["ms_per_step"] * 1e-3) / 1e9 / world,
                                 "frac": by / max(merge, 1) / (line["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "what": "algorithmic bytes of one step (per GPU: achieved_GBps, frac) -- the kernels' SURVEY 8(d) figures weighted by the mean iteration count -- over ms_per_step; "
                                         "the compensated first iteration lowers the bytes a step needs, so this fraction is not comparable across first-iteration modes: "
                                         "algorithmic_bytes_per_step_full_refactor is what the same step moves when every iteration refactorises",
                                 "algorithmic_bytes_per_step_full_refactor": by_full / max(merge, 1) * world}
        line["kernel_sum_full_refactor_ms"] = ksum_full / max(merge, 1)
        line["step_over_kernels"] = line["ms_per_step"] / line["kernel_sum_ms"]
        line["kernel_sum_what"] = ("per step and GPU: the kernels of one device batch timed ALONE (HIP events, one handle, every lane active, no compaction) weighted by the mean "
                                   "iteration count, divided by the steps a device batch holds; step_over_kernels = ms_per_step / kernel_sum_ms -- below 1 the batches in flight "
                                   "overlap and finished lanes drop out, above 1 the pipeline loses time to launches, verdicts and compaction")
        line["config"]["first_iteration"] = (
            (f"shared base factor (compensation): ONE factorisation of the base-case Jacobian per pipeline, per scenario a 4 x 4 correction and one sweep pair on it "
             f"({pipe.base.info['top_pivots']} pivots of the tree's top as a dense inverse, {pipe.base.info['forward_launches']} + {pipe.base.info['backward_launches']} level launches); "
             f"iterations >= 2 refactorise.  {first_counts[0]} of {first_counts[0] + first_counts[1]} runs of the batches started that way")
            if pipe.base is not None else "batched refactorisation (as every iteration)")
        if full_extra:
            line["value_full_refactor"] = full_extra[1] / full_extra[0]
            line["ms_per_step_full_refactor"] = 1e3 * full_extra[0] / args.steps
            line["full_refactor_what"] = "the same K-step regions in the same run with jg_nr_set_first_iteration(h, 0): every iteration refactorises (the path of rounds 1-5)"
        if world > 1 and gather_ms:
            line["gather_ms"] = float(np.mean(gather_ms))
            line["gather_ms_max"] = float(np.max(gather_ms))
            line["gather_exposed_ms"] = float(np.median(gather_ms[jobs_per_region - 1::jobs_per_region])) if len(gather_ms) >= jobs_per_region else float(gather_ms[-1])
            line["gather_what"] = ("wall time of the ONE collective per device batch on rank 0 (pack + all-gather + stream synchronisation); gather_exposed_ms: that of the LAST "
                                   "batch of a region, which no other batch in flight hides")
        line["device_state"] = {"region_start": dev_state0, "region_end": dev_state1}
        if steady_extra:
            ks, dts, its = steady_extra
            line["value_steady"] = its / dts
            line["steady_steps"] = ks
            line["steady_ms_per_step"] = 1e3 * dts / ks
            line["steady_what"] = (f"the K = {args.steps}-step region is {jobs_per_region} device batch(es) with {len(pipe.handles)} in flight (fill + drain of the pipeline); "
                                   f"value_steady is the same measurement over {ks} steps (three rounds of the batches in flight).  value, ms_per_step: the K of the caller, unchanged")
        pred = predicted_from_shards("nr", world, total)
        if pred:
            line["predicted"] = pred
        gather_close()
        pipe.close()
        try:
            hbm = hbm_measured_gbps(torch)
            line["roofline"]["hbm_measured_GBps"] = hbm
            line["roofline"]["hbm_measured_what"] = "device-to-device copy of 1 GiB (read + written bytes / time), torch copy kernel, this run"
            line["roofline"]["frac_of_measured"] = line["roofline"]["achieved"] / hbm
        except Exception as e:
            line["roofline"]["hbm_measured_GBps"] = None
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(tables, labels[: max(64, min(B, 512))], vm0, va0)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = line["value"] / cb["value"]
            si = line["single_instance"]
            si["speedup_vs_cpu_warm"] = cb["single_warm"]["ms_per_solve"] / si["ms_per_solve"]
            si["speedup_vs_cpu_cold"] = cb["single_cold"]["ms_per_solve"] / (si["setup_ms"] + si["ms_per_solve"])
            si["configs_2_3"] = single_instance_configs(jg, local, args.case)
            si["speedup_note"] = ("one scenario uses 1 of 64 lanes and is bound by dependent launches: the >= 10x of the north star holds in the "
                                  "batched regime (speedup_vs_cpu_baseline), not for a single instance")
            if base_jacobian is not None:
                sl = cpu_splu_leg(base_jacobian)
                if sl:
                    line["cpu_splu"] = sl
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if cores > 1:                               # the box's whole host: reported next to the single-thread reference path
                allc = cpu_baseline_all_cores(args.case, min(cores, 64) * 64, 512, max(2 * total, min(cores, 64) * 64), min(cores, 64))
                if allc:
                    line["cpu_baseline_all_cores"] = allc
                    line["speedup_vs_cpu_all_cores"] = line["value"] / allc["value"]
        if world == 1 and not args.no_se:
            try:
                se = se_config4(jg, cpu=not args.no_cpu)
                line["config4_se"] = {"metric": se["metric"], "value": se["value"], "unit": se["unit"], "ms_per_step": se["ms_per_step"],
                                      "iterations_per_scenario": se["iterations_per_scenario"], "converged_fraction": se["converged_fraction"],
                                      "workload": se["config"]["workload"], "rows": se["config"]["rows"],
                                      "kernels": {k: {"ms": v["ms"], "frac": v["frac"]} for k, v in se["kernels"].items()},
                                      "cpu_baseline": se.get("cpu_baseline"), "speedup_vs_cpu_baseline": se.get("speedup_vs_cpu_baseline")}
            except Exception as e:                      # the NR line is the contract; the SE object must never break it
                line["config4_se"] = {"error": repr(e)}
    else:
        gather_close()
        pipe.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (the banner sits in the C stdio buffer -- stdout is a pipe -- until it is flushed: before the descriptor is handed back)
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
