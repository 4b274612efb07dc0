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
@pytest.mark.parametrize("src", ["jg_engine.hip", "jg_gn.hip"])
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
    assert with_asm >= (3 if src == "jg_engine.hip" else 2), "k_fact_level, k_sel_level, k_fact_task / k_gn_gain, k_gn_gain_lds carry hand-placed loads"


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
