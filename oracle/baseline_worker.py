#!/usr/bin/env python3
"""One host process of bench.py's all-cores CPU baseline (TEST / MEASUREMENT INFRASTRUCTURE, not the product).

  python oracle/baseline_worker.py <case> <first> <count> <seed> <total> [cpu]

Runs the oracle's serial contingency loop (the reference's own loop) over scenarios
[first, first + count) of the seeded outage list and prints one JSON line {iters, done, seconds} (loop only: the
base-case solve and symbolic analysis are not counted, as in the single-thread leg)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(os.path.abspath(__file__))]
sys.path.insert(0, ROOT)


def main():
    case, first, count, seed, total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if len(sys.argv) > 6 and hasattr(os, "sched_setaffinity"):       # one worker per core, pinned
        try:
            os.sched_setaffinity(0, {int(sys.argv[6])})
        except OSError:
            pass
    from oracle import oracle as O
    import juliagrid.jl_amd as jg                     # host model only (outage list / Ybus patches); no GPU call is made
    if case == "case9241synth":
        tables = jg.case9241synth()
    else:
        with np.load(os.path.join(ROOT, "tests", "golden", "cases", case + ".npz")) as z:
            tables = {k: z[k] for k in z.files}
    s = jg.powerSystem(tables)
    jg.acModel_(s)
    labels = jg.outageList(s, total, seed=seed)[first:first + count]
    o = O.OracleNR(O.OracleSystem(tables))
    o.power_flow()
    vm, va = o.voltage()
    iters = 0
    t0 = time.perf_counter()
    for lab in labels:
        ptr, dy = jg.outagePatch(s, int(lab))
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, dv)
        o.set_voltage(vm, va)
        o.power_flow(iteration=20, tolerance=1e-8)
        iters += o.iteration
        for p, dv in zip(ptr, dy):
            o.add_ybus(p - 1, -dv)
    print(json.dumps({"iters": iters, "done": len(labels), "seconds": time.perf_counter() - t0}))


if __name__ == "__main__":
    main()
