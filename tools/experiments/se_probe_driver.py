#!/usr/bin/env python3
"""Timing-probe driver of the gain factorisation (tools/r04_level_probe.sh): the base power flow and the measurement set are built with the real
tables; JG_PROBE_LOADS=<probe> is set only then, so that the TASK tables of the 512-lane gain plan (built at gaussNewton) carry the probe -- wrong
numbers by construction, only the factorisation is timed.  python tools/se_probe_driver.py <probe 0|1|2> [batch] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import juliagrid.jl_amd as jg  # noqa: E402
probe = sys.argv[1] if len(sys.argv) > 1 else "0"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
s = jg.powerSystem("case9241synth")
pf = jg.newtonRaphson(s)
jg.powerFlow_(pf, tolerance=1e-11)
mon = jg.measurement(s)
jg.addVoltmeter_(mon, pf); jg.addWattmeter_(mon, pf); jg.addVarmeter_(mon, pf)
jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
if probe != "0":
    os.environ["JG_PROBE_LOADS"] = probe
an = jg.gaussNewton(mon, batch=batch)
for _ in range(3):
    print("JG_PROBE_LOADS=%s gain factorisation %.4f ms" % (probe, an.time_kernel(2, reps)))
