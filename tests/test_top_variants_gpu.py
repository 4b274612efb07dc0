"""The two shapes of the multifrontal top kernel (k_fact_top with a pivot wave / with the owner of the next diagonal block
factorising it, jg_engine.hip) and the two ways level 0 of a prefactor plan gets done (by the Jacobian assembly / by the plan's PRE
tables) must give the same BITS: which one runs depends on the size of a launch, and a scenario's result may not depend on the
batch it was solved in.  The variants are process-wide switches (environment), so each one runs in a process of its own."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
from conftest import load_case
import juliagrid.jl_amd as jg
s = jg.powerSystem(load_case({case!r}))
an = jg.contingencyAnalysis(s, jg.outageList(s, {batch}, seed=11))
jg.powerFlow_(an, iteration=20, tolerance=1e-8)
h = hashlib.sha256()
for a in (np.asarray(an.method.iteration), np.asarray(an.voltage.magnitude), np.asarray(an.voltage.angle)):
    h.update(np.ascontiguousarray(a).tobytes())
print("DIGEST", h.hexdigest(), int(np.sum(an.method.iteration)))
an.close()
"""


def _run(case, batch, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, case=case, batch=batch)], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    return line[1], int(line[2])


@pytest.mark.parametrize("case,batch", [("case1354pegase", 70), ("case_ACTIVSg10k", 130)])
def test_top_kernel_variants_and_level0_routes_give_the_same_bits(case, batch):
    ref, iters = _run(case, batch)                                   # the build's own choice per launch
    assert iters >= 3 * batch
    assert _run(case, batch, JG_TOP_PW=1)[0] == ref                  # every top launch with the pivot wave
    assert _run(case, batch, JG_TOP_PW=0)[0] == ref                  # every top launch without
    assert _run(case, batch, JG_NO_PREFACTOR=1)[0] == ref            # plain plan: the level kernel factorises the leaf blocks
    assert _run(case, batch, JG_TOP_FUSE=1)[0] == ref                # two pivots per barrier: every thread redoes what the owners of the
                                                                     # second pivot's row / column / block do, operation for operation


SCRIPT_STATE = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
from conftest import load_case
import juliagrid.jl_amd as jg
s = jg.powerSystem(load_case({case!r}))
an = jg.contingencyAnalysis(s, jg.outageList(s, {batch}, seed=11))
jg.powerFlow_(an, iteration=20, tolerance=1e-8)
np.savez({out!r}, it=np.asarray(an.method.iteration), st=np.asarray(an.status), vm=np.asarray(an.voltage.magnitude), va=np.asarray(an.voltage.angle))
an.close()
"""


def _state(tmp_path, tag, case, batch, env):
    import numpy as np
    out = str(tmp_path / f"{tag}.npz")
    ee = dict(os.environ)
    ee.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", SCRIPT_STATE.format(root=ROOT, case=case, batch=batch, out=out)], env=ee, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with np.load(out) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case,batch", [("case118", 70), ("case1354pegase", 300), ("case_ACTIVSg10k", 130), ("case_ACTIVSg10k", 512)])
def test_jordan_rows_and_small_chains_match_the_plain_backward_sweep(tmp_path, case, batch):
    """Round 3: the top tasks eliminate above the diagonal as well and leave Jordan rows (the backward sweep over the top is one plain level
    per task level instead of sequential chains: jg_symbolic.hpp), and the chains below the top run as 8-wave tasks.  Both change the order
    of a handful of additions, not the algorithm: against the plain sweep (JG_JORDAN=0) and the general chain tasks (JG_CHAIN_SMALL=0) the
    iteration counts and status are equal and V / theta agree to 1e-10."""
    import numpy as np
    ref = _state(tmp_path, "ref", case, batch, dict())
    ok = ref["st"] == 0
    assert ok.sum() >= 0.9 * batch
    for tag, env in (("plain", dict(JG_JORDAN=0)), ("general", dict(JG_CHAIN_SMALL=0)), ("both", dict(JG_JORDAN=0, JG_CHAIN_SMALL=0))):
        b = _state(tmp_path, tag, case, batch, env)
        assert np.array_equal(ref["it"], b["it"]) and np.array_equal(ref["st"], b["st"]), tag
        assert np.abs(ref["vm"] - b["vm"])[ok].max() < 1e-10 and np.abs(ref["va"] - b["va"])[ok].max() < 1e-10, tag


@pytest.mark.parametrize("case", ["case118", "case1354pegase", "case9241synth", "case_ACTIVSg10k"])
def test_single_instance_kernels_match_the_level_launches(tmp_path, case):
    """Round 6: a handle of ONE scenario factorises below the top with a quad of lanes per item (k_fact1_bottom / k_fact1_partial) and sweeps backward with rows as
    lanes (k_bwd1_top over compact Jordan rows, k_bwd1_bottom) instead of giving a wave with one live lane to every item (JG_SINGLE=0: the level launches).  Another
    summation order of long term lists, the same algorithm: equal iteration count and status, V / theta to 1e-10."""
    import numpy as np
    ref = _state(tmp_path, "ref", case, 1, dict())
    lvl = _state(tmp_path, "lvl", case, 1, dict(JG_SINGLE=0))
    assert int(np.atleast_1d(ref["st"])[0]) == 0
    assert np.array_equal(ref["it"], lvl["it"]) and np.array_equal(ref["st"], lvl["st"])
    assert np.abs(ref["vm"] - lvl["vm"]).max() < 1e-10 and np.abs(ref["va"] - lvl["va"]).max() < 1e-10
    quad = _state(tmp_path, "quad", case, 1, dict(JG_SINGLE=2))     # the top's sweep with a quad of lanes per row (k_bwd1_top) instead of the terms as lanes (k_bwd1_top2)
    assert np.array_equal(ref["it"], quad["it"]) and np.array_equal(ref["st"], quad["st"])
    assert np.abs(ref["vm"] - quad["vm"]).max() < 1e-10 and np.abs(ref["va"] - quad["va"]).max() < 1e-10


def test_refined_steps_switch_the_engine_back_to_plain_rows(jg):
    """Iterative refinement runs forward() + backsolve() on the factor of the step: the forward elimination of another right-hand side gives
    y, not the y' Jordan rows go with, so jg_nr_set_refine turns Engine::jordan off (and on again when refinement goes off).  A handle that
    refined and stopped refining must give the bits of one that never did."""
    import numpy as np
    from conftest import load_case
    s = jg.powerSystem(load_case("case1354pegase"))
    plain = jg.newtonRaphson(s)
    jg.powerFlow_(plain)
    once = jg.newtonRaphson(s, refine=True)
    jg.powerFlow_(once)
    assert np.array_equal(once.method.iteration, plain.method.iteration)
    assert np.abs(once.voltage.magnitude - plain.voltage.magnitude).max() < 1e-10 and np.abs(once.voltage.angle - plain.voltage.angle).max() < 1e-10
    jg.powerflow.setRefinement_(once, False)
    jg.setInitialPoint_(once)
    jg.powerFlow_(once)
    assert np.array_equal(once.voltage.magnitude, plain.voltage.magnitude) and np.array_equal(once.voltage.angle, plain.voltage.angle)
    plain.close(); once.close()


def test_shared_device_hint_keeps_the_bits(jg):
    """jg_nr_set_shared (a pipeline's handles: every top launch takes the 4-wave kernel variant) is a performance hint: same bits."""
    import numpy as np
    from conftest import load_case
    s = jg.powerSystem(load_case("case_ACTIVSg10k"))
    labels = jg.outageList(s, 130, seed=11)
    a = jg.contingencyAnalysis(s, labels)
    b = jg.contingencyAnalysis(s, labels)
    jg._lib.check(jg._lib.lib().jg_nr_set_shared(b._h, 1))
    for an in (a, b):
        jg.powerFlow_(an, iteration=20, tolerance=1e-8)
    assert np.array_equal(a.method.iteration, b.method.iteration)
    assert np.array_equal(a.voltage.magnitude, b.voltage.magnitude) and np.array_equal(a.voltage.angle, b.voltage.angle)
    a.close(); b.close()


SCRIPT_SE = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
import juliagrid.jl_amd as jg
s = jg.powerSystem({case!r})
pf = jg.newtonRaphson(s)
jg.powerFlow_(pf, tolerance=1e-11)
mon = jg.measurement(s)
jg.addVoltmeter_(mon, pf, variance=1e-4); jg.addWattmeter_(mon, pf, variance=1e-4); jg.addVarmeter_(mon, pf, variance=1e-4)
jg.addPmu_(mon, pf, buses=range(1, s.bus.number + 1, 10), statusTo=-1, minMagnitude=1e-6)
an = jg.gaussNewton(mon, batch={batch})
jg.drawNoise_(an, 4)
an.setVoltage(np.ones(s.bus.number), np.zeros(s.bus.number))
jg.stateEstimation_(an, iteration=40, tolerance=1e-8)
np.savez({out!r}, it=np.asarray(an.method.iteration), st=np.asarray(an.status), vm=np.asarray(an.voltage.magnitude), va=np.asarray(an.voltage.angle), obj=np.asarray(an.objectiveDevice()))
"""


@pytest.mark.parametrize("case,batch", [("case9241synth", 128), ("case9241synth", 1)])
def test_symmetric_top_kernel_against_the_mirrored_front(tmp_path, case, batch):
    """k_fact_top_sym (round 5: the top tasks of the Gauss-Newton gain keep and eliminate the upper triangle only) against k_fact_top on the mirrored front (JG_TOP_SYM=0):
    another summation order, not another algorithm -- equal iteration counts and status, estimates to 1e-10, objectives to 1e-9 relative on noisy realisations of config 4."""
    outs = []
    for mode in (1, 0):
        out = str(tmp_path / f"sym{mode}.npz")
        e = dict(os.environ, JG_TOP_SYM=str(mode))
        r = subprocess.run([sys.executable, "-c", SCRIPT_SE.format(root=ROOT, case=case, batch=batch, out=out)], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        import numpy as np
        outs.append(dict(np.load(out)))
    import numpy as np
    a, b = outs
    assert np.array_equal(a["it"], b["it"]) and np.array_equal(a["st"], b["st"]) and np.all(a["st"] == 0)
    assert np.abs(a["vm"] - b["vm"]).max() <= 1e-10 and np.abs(a["va"] - b["va"]).max() <= 1e-10
    assert np.abs(a["obj"] - b["obj"]).max() <= 1e-9 * np.abs(b["obj"]).max()


SCRIPT_COMPACT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
from conftest import load_case
import juliagrid.jl_amd as jg
s = jg.powerSystem(load_case({case!r}))
base = jg.newtonRaphson(s)
jg.powerFlow_(base, tolerance=1e-10)
an = jg.contingencyAnalysis(s, jg.outageList(s, {batch}, seed=512))
jg.contingency._push_voltage(an, base.voltage.magnitude.copy(), base.voltage.angle.copy())          # from the base case's solution: most outages need 3 iterations, some 2 or 4
jg.powerFlow_(an, iteration=20, tolerance=1e-8)
h = hashlib.sha256()
for a in (np.asarray(an.method.iteration), np.asarray(an.status), np.asarray(an.voltage.magnitude), np.asarray(an.voltage.angle), np.asarray(an.increment)):
    h.update(np.ascontiguousarray(a).tobytes())
print("DIGEST", h.hexdigest(), int(np.sum(an.method.iteration)), int(np.max(an.method.iteration)) - int(np.min(an.method.iteration)))
an.close()
"""


@pytest.mark.parametrize("case,batch", [("case_ACTIVSg10k", 200), ("case1354pegase", 512), ("case_ACTIVSg10k", 1100)])
def test_lane_moves_in_place_give_the_bits_of_the_two_pass_move(case, batch):
    """Compaction packs the still-active scenarios into the leading lanes.  Round 5 moves the lanes IN PLACE (k_lanes_permute: a workgroup holds every lane of a
    row; one launch) instead of through a staging area and back (JG_LANES_INPLACE=0): same bits in every per-scenario array that travels -- state, last
    increment, iteration count, status -- for batches of one, two and five workgroup-widths of lanes (1 100 scenarios: two lanes per thread)."""
    def run(**env):
        e = dict(os.environ)
        e.update({k: str(v) for k, v in env.items()})
        out = subprocess.run([sys.executable, "-c", SCRIPT_COMPACT.format(root=ROOT, case=case, batch=batch)], env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
        return line[1], int(line[2]), int(line[3])
    ref, iters, spread = run()
    assert iters >= 2 * batch and spread >= 1, "scenarios finish after different iteration counts: lanes move"
    assert run(JG_LANES_INPLACE=0)[0] == ref
