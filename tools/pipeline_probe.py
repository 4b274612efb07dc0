#!/usr/bin/env python3
"""Does a second batch in flight (own handle, own stream, own host thread) hide the straggler iterations of the first?
python tools/pipeline_probe.py [batch] [steps] [inflight]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 2
s = jg.powerSystem("case_ACTIVSg10k")
base = jg.newtonRaphson(s, batch=1)
jg.powerFlow_(base)
vm0, va0 = base.voltage.magnitude.copy(), base.voltage.angle.copy()
base.close()
labels = jg.outageList(s, batch, seed=512)
hs = []
for k in range(inflight):
    an = jg.contingencyAnalysis(s, labels)
    jg.powerflow._push_voltage(an, vm0, va0)
    an.snapshot_voltage()
    hs.append(an)

def work(an, n, out):
    it = 0
    for _ in range(n):
        an.restore_voltage()
        jg.powerFlow_(an, iteration=20, tolerance=1e-8, fetch=False)
        it += int(np.sum(an.method.iteration))
    out.append(it)

for an in hs:
    work(an, 2, [])
out = []
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(an, steps // inflight, out)) for an in hs]
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
print("inflight", inflight, "steps", steps, "ms/step", 1e3 * dt / steps, "NR it/s", sum(out) / dt)
