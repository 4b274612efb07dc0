"""Builds libjgrid_hip.so (hand-written HIP for gfx950 + host symbolic analysis) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["jg_symbolic.cpp", "jg_plan_api.cpp", "jg_comm.cpp", "jg_engine.hip", "jg_nr.hip", "jg_gn.hip"]
LIB = os.environ.get("JG_LIB_OUT") or os.path.join(HERE, "libjgrid_hip.so")     # JG_LIB_OUT: build a variant somewhere else (JG_EXTRA_HIPCC_FLAGS)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libjgrid_hip.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "jgrid.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"] + os.environ.get("JG_EXTRA_HIPCC_FLAGS", "").split() + ["-o", LIB + ".tmp"] + srcs + ["-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
