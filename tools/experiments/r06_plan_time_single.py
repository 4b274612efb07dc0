# Symbolic analysis of the 10k-bus grid under the policies of a handful of scenarios (round 5 / with 12 pivots per task / with the tables of a single instance): ms per analysis.
# GPU box: 14.2 / 14.2 / 15.5-16.5 ms (the single-instance tables are built on a thread of their own).
import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import juliagrid.jl_amd as jg
s = jg.powerSystem("case_ACTIVSg10k"); jg.acModel_(s)
Y = s.model.ac.nodalMatrix
base = 1 | 4 | (26 << 16 | 127 << 24) | 1 << 49
for name, pol in (("r05 tiny", base), ("mmin12", base | 12 << 54), ("single", base | 12 << 54 | 1 << 60)):
    ts=[]
    for _ in range(5):
        t0=time.perf_counter(); plan = jg._lib.Plan(Y.n, Y.colptr - 1, Y.rowval - 1, policy=pol); ts.append(time.perf_counter()-t0)
    print(name, "analysis ms", [round(1e3*t,1) for t in ts])
    t0=time.perf_counter(); plan.get(90); print("   single bwd tables ms", round(1e3*(time.perf_counter()-t0),1))
