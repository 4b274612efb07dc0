#!/usr/bin/env python3
"""BASELINE config 4: Gauss-Newton WLS state estimation (PMU + legacy) on the 9241-bus PEGASE-shaped grid, 1 GPU.

  python tools/bench_se.py [--batch 256] [--steps 10] [--case case9241synth]

Measurement set: voltmeter at every bus, wattmeter + varmeter at every bus and both ends of every
in-service branch (variance 1e-4), PMUs at every 10th bus (bus phasor + from-end current phasors, variance 1e-8),
synthesised from the converged power flow; scenario b reads z + sigma * N(0,1) (seed 4).  One step = restore the flat
start inside HBM and run stateEstimation! (tol 1e-8, max 40) for the whole batch.  Prints one JSON line:
GN iterations/s, ms per solve, per-kernel times with algorithmic bytes, and the CPU oracle on one host core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


from bench import cpu_baseline_se as cpu_baseline      # noqa: E402  (the oracle is only ever touched from bench.py's CPU legs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--case", default="case9241synth")
    ap.add_argument("--inflight", type=int, default=2, help="batches in flight (own handle, stream and host thread each)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import juliagrid.jl_amd as jg

    s = jg.powerSystem(args.case)
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, tolerance=1e-11)
    assert pf.status == 0
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf, variance=1e-4)
    jg.addWattmeter_(mon, pf, variance=1e-4)
    jg.addVarmeter_(mon, pf, variance=1e-4)
    jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
    import threading
    n = s.bus.number
    handles = []
    for k in range(max(1, args.inflight)):            # like ContingencyPipeline: the batches of different handles overlap on the GPU
        h = jg.gaussNewton(mon, batch=args.batch)
        jg.setNoise_(h, np.random.Generator(np.random.PCG64(4 + k)), scale=1.0)
        h.setVoltage(np.ones(n), np.zeros(n))
        h.snapshot_voltage()                          # the flat start stays resident in HBM
        handles.append(h)
    an = handles[0]

    def step(h):
        h.restore_voltage()
        jg.stateEstimation_(h, iteration=40, tolerance=1e-8, fetch=False)
        return int(np.sum(h.method.iteration))

    def run(steps):
        out = [0] * len(handles)

        def work(k):
            for _ in range(k, steps, len(handles)):
                out[k] += step(handles[k])
        ths = [threading.Thread(target=work, args=(k,)) for k in range(len(handles))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        return sum(out)

    run(args.warmup * len(handles))
    t0 = time.perf_counter()
    iters = run(args.steps)
    dt = time.perf_counter() - t0
    d = an.dims
    B = args.batch
    kern = {}
    nnzH = d["nnzH"]
    algo = {"rows": B * (16 * d["slots"] + 16 * d["m"] + 16 * n), "gain": B * (16 * d["slots"] + 8 * d["m"] + 32 * d["gain_blocks"] + 16 * n),
            "factor": B * (32 * d["gain_blocks"] + 64 * d["lu_blocks"]), "backward": B * (32 * d["lu_blocks"] + 64 * n)}
    for k, name in enumerate(("rows", "gain", "factor", "backward")):
        ms = an.time_kernel(k, 5)
        kern[name] = {"ms": ms, "bytes": algo[name], "GBps": algo[name] / ms / 1e6, "frac": algo[name] / ms / 1e6 / HBM_PEAK_GBS}
    line = {"metric": "GN iterations/sec (WLS state estimation, PMU + legacy, 9241-bus PEGASE-shaped grid)", "value": iters / dt,
            "unit": "GN iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_solve_batched": 1e3 * dt / (B * args.steps), "iterations_per_scenario": iters / (B * args.steps),
            "converged_fraction": float(np.mean(an.status == 0)), "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.case} Gauss-Newton WLS SE, {B} noisy realisations per batch, {len(handles)} batches in flight, "
                                   "flat start, tol 1e-8, max 40", "rows": d["m"],
                       "nnzH": nnzH, "gain_blocks": d["gain_blocks"], "lu_blocks": d["lu_blocks"], "lu_terms": d["lu_terms"],
                       "factor_launches": d["factor_launches"], "backward_launches": d["backward_launches"]},
            "kernels": kern}
    if not args.no_cpu:
        line["cpu_baseline"] = cpu_baseline(jg, s, args.case, pf)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
