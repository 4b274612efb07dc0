"""The device-free host code of the library -- ordering, fill pattern, update terms, levels, replay tables, task tables, plan export
(csrc/jg_symbolic.cpp, jg_plan_api.cpp: ~1 000 lines of index arithmetic) -- under AddressSanitizer + UndefinedBehaviorSanitizer
(GPU sanitizers are not available, the CPU build is).  tools/asan_plan.sh builds a plan-only library with
-fsanitize=address,undefined and runs the plan tests against it (JG_PLAN_LIB); here: the graph-shape tests (degenerate graphs, random
graphs, every task geometry); `tools/asan_plan.sh` alone runs all of tests/test_plan_cpu.py."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan / libubsan")
def test_plan_code_is_clean_under_asan_and_ubsan():
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_plan.sh"), "-k", "degenerate or small_and_random"],
                       capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail, tail
