#!/usr/bin/env python3
"""Round 5: which side of tests/test_pipeline_gpu.py::test_monte_carlo_injection_jobs_through_the_pipeline is not reproducible -- the pipeline's records or the plain batch's -- and in which lanes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import juliagrid.jl_amd as jg
from conftest import load_case
t = load_case("case1354pegase")
s = jg.powerSystem(t)
n, B = s.bus.number, 192
base = jg.newtonRaphson(s); jg.powerFlow_(base)
start = (base.voltage.magnitude.copy(), base.voltage.angle.copy()); base.close()
rng = np.random.default_rng(17)
jobs = []
for _ in range(3):
    scale = 1.0 + 0.05 * rng.standard_normal((B, 1))
    jobs.append({"active": s.bus.supply.active[None, :] - s.bus.demand.active[None, :] * scale, "reactive": s.bus.supply.reactive[None, :] - s.bus.demand.reactive[None, :] * scale})
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pool = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ref = jg.newtonRaphson(s, batch=B, max_patch=4)
plain = []
for r in range(reps):
    recs = []
    for job in jobs:
        jg.setInjection_(ref, job["active"], job["reactive"]); jg.powerflow._push_voltage(ref, *start); jg.powerFlow_(ref)
        rec = torch.zeros((B, 2 * n + 2), dtype=torch.float64, device="cuda"); torch.cuda.current_stream().synchronize(); ref.pack_results_device(rec.data_ptr()); recs.append(rec.cpu().numpy())
    plain.append(recs)
bad_plain = sum(int(not np.array_equal(plain[r][j], plain[0][j])) for r in range(reps) for j in range(3))
print("plain batch: records that differ from the first repetition:", bad_plain, "of", reps * 3)
for r in range(1, min(reps, 4)):
    for j in range(3):
        a, b = plain[r][j], plain[0][j]
        if not np.array_equal(a, b):
            rows = np.nonzero(np.any(a != b, axis=1))[0]
            print(f"  plain rep {r} job {j}: {rows.size} scenarios differ from rep 0; columns: V {int((a[:, :n] != b[:, :n]).sum())} theta {int((a[:, n:2*n] != b[:, n:2*n]).sum())} iterations {int((a[:, 2*n] != b[:, 2*n]).sum())} status {int((a[:, 2*n+1] != b[:, 2*n+1]).sum())}; max |dV| {np.abs(a[:, :n] - b[:, :n]).max():.3e}; rep {r} against rep {r-1 if r > 1 else 0}: {int(not np.array_equal(a, plain[r-1][j]))}")
pipe = jg.ContingencyPipeline(s, B, inflight=2, start=start, pool=pool)
ring = [torch.zeros((B, 2 * n + 2), dtype=torch.float64, device="cuda") for _ in range(3)]
torch.cuda.current_stream().synchronize()
nbad = 0
for r in range(reps):
    seen = []
    def on_done(j, an):
        seen.append(ring[j % 3].clone()); torch.cuda.current_stream().synchronize()
    res = pipe.run(jobs, on_done=on_done, record=lambda j: ring[j % 3].data_ptr(), records=3)
    for j in range(3):
        a = seen[j].cpu().numpy()
        if not np.array_equal(a, plain[0][j]):
            nbad += 1
            rows = np.nonzero(np.any(a != plain[0][j], axis=1))[0]
            it_p, it_r = a[rows, 2 * n], plain[0][j][rows, 2 * n]
            print(f"rep {r} job {j}: {rows.size} scenarios differ, lanes {rows[:12]}, iterations pipeline {it_p[:12]} plain {it_r[:12]}, max |dV| {np.abs(a[rows, :n] - plain[0][j][rows, :n]).max():.3e}")
print("pipeline (pool", pool, "): records that differ from the plain batch:", nbad, "of", reps * 3)
pipe.close(); ref.close()
