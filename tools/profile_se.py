#!/usr/bin/env python3
"""Driver for the rocprofv3 passes of BASELINE config 4 (Gauss-Newton WLS SE on the 9241-bus grid): ONE handle, `iters`
Gauss-Newton iterations of a noisy batch from the flat start.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d ... -- python tools/profile_se.py [batch] [iters]   (WRITE_SIZE in its own pass)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s = jg.powerSystem("case9241synth")
pf = jg.newtonRaphson(s)
jg.powerFlow_(pf, tolerance=1e-11)
mon = jg.measurement(s)
jg.addVoltmeter_(mon, pf, variance=1e-4); jg.addWattmeter_(mon, pf, variance=1e-4); jg.addVarmeter_(mon, pf, variance=1e-4)
jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
an = jg.gaussNewton(mon, batch=batch)
jg.setNoise_(an, np.random.Generator(np.random.PCG64(4)), scale=1.0)
an.setVoltage(np.ones(s.bus.number), np.zeros(s.bus.number))
jg.stateEstimation_(an, iteration=iters, tolerance=1e-8, fetch=False)
print("dims", an.dims, "iterations", int(np.sum(an.method.iteration)))
