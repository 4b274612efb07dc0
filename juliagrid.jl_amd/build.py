"""Builds libjgrid_hip.so (hand-written HIP for gfx950 + host symbolic analysis) in-tree with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["jg_symbolic.cpp", "jg_plan_api.cpp", "jg_comm.cpp", "jg_engine.hip", "jg_comp.hip", "jg_nr.hip", "jg_gn.hip"]
LIB = os.environ.get("JG_LIB_OUT") or os.path.join(HERE, "libjgrid_hip.so")     # JG_LIB_OUT: build a variant somewhere else (JG_EXTRA_HIPCC_FLAGS)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libjgrid_hip.so)")


def source_hash():
    """SHA-256 over the sources the library is built from (csrc/*, include/jgrid.h), the compiler flags and this script's command shape."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "jgrid.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0")
            with open(f, "rb") as fh:
                h.update(fh.read())
            h.update(b"\0")
    h.update(" ".join(SOURCES).encode())
    h.update(os.environ.get("JG_EXTRA_HIPCC_FLAGS", "").encode())
    return h.hexdigest()


def stamp_path():
    return LIB + ".sha256"


def needs_build():
    """The decision rests on CONTENT, not on modification times (the git-ignored .so travels prebuilt to the GPU box and a push
    leaves whatever mtimes it leaves): the hash of the sources the library was built from is stored beside it."""
    if not os.path.exists(LIB) or not os.path.exists(stamp_path()):
        return True
    with open(stamp_path()) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    want = source_hash()
    if not force and not needs_build():
        if verbose or os.environ.get("JG_BUILD_VERBOSE"):
            print(f"[jgrid build] up to date: {LIB} was built from sources {want[:12]}")
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed"] + os.environ.get("JG_EXTRA_HIPCC_FLAGS", "").split() + ["-o", LIB + ".tmp"] + srcs + ["-ldl", "-pthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    with open(stamp_path(), "w") as fh:
        fh.write(want + "\n")
    if verbose or os.environ.get("JG_BUILD_VERBOSE"):
        print(f"[jgrid build] rebuilt {LIB} from sources {want[:12]}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
