import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np
import juliagrid.jl_amd as jg
s = jg.powerSystem("case9241synth")
pf = jg.newtonRaphson(s); jg.powerFlow_(pf, tolerance=1e-11)
mon = jg.measurement(s)
t=time.perf_counter()
jg.addVoltmeter_(mon, pf, variance=1e-4); jg.addWattmeter_(mon, pf, variance=1e-4); jg.addVarmeter_(mon, pf, variance=1e-4)
jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
print("meters %.1f ms" % ((time.perf_counter()-t)*1e3))
for k in range(3):
    t=time.perf_counter()
    if k==0:
        pr=cProfile.Profile(); pr.enable()
    h = jg.gaussNewton(mon, batch=512)
    if k==0:
        pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
    print("gaussNewton %d: %.1f ms" % (k, (time.perf_counter()-t)*1e3))
t=time.perf_counter()
p = jg.MonteCarloPipeline(mon, 512, inflight=2)
print("MonteCarloPipeline: %.1f ms" % ((time.perf_counter()-t)*1e3))
