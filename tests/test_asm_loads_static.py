"""The hand-placed loads of the kernels (csrc/jg_engine.hpp: gload16 / gload8 -- inline-assembly `global_load` requests with ONE hand-written wait)
checked in the COMPILED ISA: between a request and the wait that covers it the compiler, which cannot see the request, must not touch its destination
registers.  tools/check_asm_loads.py walks the control-flow graph of every kernel of a `hipcc -S` listing (no GPU needed: hipcc cross-compiles).
It found a real defect while the technique was introduced (a register copied at a control-flow merge while its load was in flight, k_gn_gain)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_asm_loads  # noqa: E402

CSRC = os.path.join(ROOT, "juliagrid.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["jg_engine.hip", "jg_gn.hip", "jg_comp.hip"])
def test_no_compiler_instruction_touches_a_register_whose_asm_load_is_in_flight(tmp_path, src):
    out = tmp_path / (src + ".s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", "-Wno-unused-value", "-Wno-unused-result",
                           "-Wno-pass-failed", "-o", str(out), os.path.join(CSRC, src)], cwd=CSRC)
    lines = out.read_text().splitlines()
    with_asm = 0
    for name, body in check_asm_loads.kernels(lines):
        in_asm, hand = False, 0
        for l in body:
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm and "global_load" in l:
                hand += 1
        if not hand:
            continue
        with_asm += 1
        findings = check_asm_loads.check_kernel(name, body)
        assert not findings, (name, findings[:5])
    assert with_asm >= {"jg_engine.hip": 3, "jg_gn.hip": 1, "jg_comp.hip": 2}[src], "k_fact_level, k_sel_level, k_fact_task / k_gn_gain / k_csweep forward and backward carry hand-placed loads"


def test_the_checker_sees_a_planted_defect():
    """A listing with a copy of a requested register ahead of the wait must be reported (the checker is not a rubber stamp), a clean one not."""
    bad = """_Zplanted:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v1, s[0:1]
	;;#ASMEND
	v_mov_b32_e32 v9, v5
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f64 v[10:11], v[4:5], v[6:7]
	s_endpgm
.Lfunc_end0:
""".splitlines()
    (name, body), = check_asm_loads.kernels(bad)
    f = check_asm_loads.check_kernel(name, body)
    assert len(f) == 1 and "v_mov_b32_e32 v9, v5" in f[0][1]
    good = [l for l in bad if "v_mov_b32_e32 v9, v5" not in l]
    (name, body), = check_asm_loads.kernels(good)
    assert not check_asm_loads.check_kernel(name, body)
    # a counted wait that leaves the request in flight does not clear it; one behind a branch is followed along both edges
    counted = [l.replace("vmcnt(0)", "vmcnt(1)") for l in good]
    (name, body), = check_asm_loads.kernels(counted)
    assert check_asm_loads.check_kernel(name, body)


def test_the_checker_follows_numeric_local_labels_inside_an_asm_statement():
    """ k_gn_gain's multi-instruction statement branches to GNU local labels (`1f`, `2f`).  Unresolved, `s_branch 2f` had no successor and everything
    behind the first statement was unreachable -- the checker passed vacuously.  A register copied behind such a statement must be found; an asm load whose
    address register is the destination of a load in flight (an output that was not declared early-clobber) as well; and a block that is never reached is a
    finding of its own."""
    listing = """_Zlocal:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	;;#ASMSTART
	s_cmp_eq_u32 s8, 3
	s_cbranch_scc1 2f
	global_load_dwordx2 v[4:5], v1, s[0:1]
	s_cmp_eq_u32 s8, 1
	s_cbranch_scc1 1f
	global_load_dwordx4 v[10:13], v2, s[2:3]
	s_branch 2f
1:
	global_load_dwordx2 v[6:7], v1, s[2:3]
2:
	;;#ASMEND
	PLANT
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_f64 v[20:21], v[4:5], v[6:7]
	s_endpgm
.Lfunc_end0:
"""
    def run(plant):
        (name, body), = check_asm_loads.kernels(listing.replace("PLANT", plant).splitlines())
        return check_asm_loads.check_kernel(name, body)
    assert not run("s_nop 0")
    f = run("v_mov_b32_e32 v99, v11")                           # destination of the load behind `s_branch 2f`'s block
    assert len(f) == 1 and "v99, v11" in f[0][1]
    f = run("v_mov_b32_e32 v99, v6")                            # ... of the load at local label 1
    assert len(f) == 1 and "v99, v6" in f[0][1]
    # an output sharing a register with the lane offset a later load of the statement reads
    clash = listing.replace("global_load_dwordx2 v[4:5], v1, s[0:1]", "global_load_dwordx2 v[1:2], v1, s[0:1]").replace("PLANT", "s_nop 0")
    (name, body), = check_asm_loads.kernels(clash.splitlines())
    f = check_asm_loads.check_kernel(name, body)
    assert f and all("address register" in x[1] for x in f)
    # a branch to a label that does not exist: conservative fall-through, nothing is skipped
    lost = listing.replace("s_branch 2f", "s_branch 7f").replace("PLANT", "v_mov_b32_e32 v99, v6")
    (name, body), = check_asm_loads.kernels(lost.splitlines())
    assert any("v99, v6" in x[1] for x in check_asm_loads.check_kernel(name, body))
    # a block nobody reaches is reported, not passed over
    dead = listing.replace("PLANT", "s_branch .LBB9\n.Ldead:\n\tv_mov_b32_e32 v50, v51\n.LBB9:")
    (name, body), = check_asm_loads.kernels(dead.splitlines())
    assert any("unreachable" in x[1] for x in check_asm_loads.check_kernel(name, body))
