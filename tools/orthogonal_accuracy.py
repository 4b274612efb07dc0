#!/usr/bin/env python3
"""Increment of the Orthogonal tag (corrected semi-normal equations on the device) against a dense QR solve of the same
least-squares problem, next to the plain normal equations, as the weight spread of the set grows: python tools/orthogonal_accuracy.py"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import juliagrid.jl_amd as jg
from oracle import oracle
from test_oracle_se import se_case14
from test_se_gpu import _mirror, _system_like
t, osys, vm, va = se_case14(oracle)
for v in (1e-8, 1e-12, 1e-14):
    tab = oracle.MeterTable()
    oracle.add_from_power_flow(tab, osys, vm, va, "wattmeter", variance=1.0)
    oracle.add_from_power_flow(tab, osys, vm, va, "varmeter", variance=1.0)
    oracle.add_from_power_flow(tab, osys, vm, va, "pmu", bus=True, frm=False, to=False, variance=v)
    s = _system_like(jg, t, osys)
    gn = oracle.OracleGN(osys, tab)
    ref = gn.increment_orthogonal()
    gn.increment(); cn = gn.vectors()["increment"]
    o = jg.gaussNewton(_mirror(jg, s, tab), jg.Orthogonal); jg.incrementSE_(o)
    p = jg.gaussNewton(_mirror(jg, s, tab)); jg.incrementSE_(p)
    sc = np.abs(ref).max()
    print("pmu variance", v, "rel err vs QR: device orthogonal %.2e, device normal %.2e, C oracle normal %.2e" % (np.abs(o.increment-ref).max()/sc, np.abs(p.increment-ref).max()/sc, np.abs(cn-ref).max()/sc))
