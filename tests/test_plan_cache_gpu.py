"""The process-wide plan cache (csrc/jg_engine.hip: acquire_plan): engines of one (pattern, policy, device) share ONE symbolic analysis and ONE
device copy of its tables.  Results must not depend on whether a plan came from the cache; clearing the cache while handles are live must leave them
working; creates from several host threads (same grid: one analyses, the others wait and hit; different grids: side by side) must all succeed."""
import hashlib
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, load_case

pytestmark = pytest.mark.gpu

_DIGEST = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import juliagrid.jl_amd as jg
from conftest import load_case
h = hashlib.sha256()
for name, batch in (("case118", 1), ("case1354pegase", 70), ("case1354pegase", 1)):
    s = jg.powerSystem(load_case(name))
    for rep in range(2):                                   # the second handle of a grid is the one that can hit
        an = jg.newtonRaphson(s, batch=batch)
        jg.powerFlow_(an)
        h.update(np.ascontiguousarray(an.voltage.magnitude).tobytes()); h.update(np.ascontiguousarray(an.voltage.angle).tobytes())
        h.update(np.ascontiguousarray(an.method.iteration).tobytes())
        an.close()
print("DIGEST", h.hexdigest())
"""


def _digest(**env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", _DIGEST % (ROOT, os.path.join(ROOT, "tests"))], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0]


def test_results_do_not_depend_on_the_cache():
    assert _digest() == _digest(JG_PLAN_CACHE="0")


def test_clearing_the_cache_under_live_handles(jg):
    s = jg.powerSystem(load_case("case1354pegase"))
    a = jg.newtonRaphson(s, batch=3)
    jg.powerFlow_(a)
    vm = a.voltage.magnitude.copy()
    jg._lib.lib().jg_plan_cache_clear()                      # a keeps its plan (shared_ptr); the cache forgets it
    b = jg.newtonRaphson(jg.powerSystem(load_case("case1354pegase")), batch=3)     # a full analysis again
    jg.powerFlow_(b)
    jg.setInitialPoint_(a)
    jg.powerFlow_(a)                                         # the old handle still runs on its own plan
    assert np.array_equal(a.voltage.magnitude, vm) and np.array_equal(b.voltage.magnitude, vm)
    a.close()
    jg._lib.lib().jg_plan_cache_clear()
    jg.setInitialPoint_(b)
    jg.powerFlow_(b)
    assert np.array_equal(b.voltage.magnitude, vm)
    b.close()


def test_creates_from_several_threads(jg):
    jg._lib.lib().jg_plan_cache_clear()
    cases = ["case1354pegase", "case1354pegase", "case1354pegase", "case118", "case300", "case118"]
    out, errs = [None] * len(cases), []

    def work(k):
        try:
            an = jg.newtonRaphson(jg.powerSystem(load_case(cases[k])))
            jg.powerFlow_(an)
            out[k] = (int(an.method.iteration), an.voltage.magnitude.copy())
            an.close()
        except Exception as e:                               # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(cases))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for k in range(len(cases)):
        first = cases.index(cases[k])
        assert out[k][0] == out[first][0] and np.array_equal(out[k][1], out[first][1])
