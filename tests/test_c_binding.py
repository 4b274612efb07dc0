"""The C ABI driven from a plain C program (bindings/c/nr_from_file.c): compiled with gcc against include/jgrid.h and
libjgrid_hip.so, no Python and no torch on the calling side.  CPU: it compiles and links.  GPU: it reproduces the MATPOWER
goldens of the reference (test/data/results.h5: iteration counts, V, theta to 1e-8) and is bitwise identical across the
scenarios of a batch that holds the same problem 70 times."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_case, load_golden

SRC = os.path.join(ROOT, "bindings", "c", "nr_from_file.c")


def build(tmp_path):
    exe = str(tmp_path / "nr_from_file")
    libdir = os.path.join(ROOT, "juliagrid.jl_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, "-ljgrid_hip",
                    f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    return exe


def test_c_driver_compiles_and_links(jg, tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def write_model(jg, name, path, max_iter=20, tol=1e-8):
    s = jg.powerSystem(load_case(name))
    jg.acModel_(s)
    vm, va = jg.initializeACPowerFlow(s)
    Y, YT = s.model.ac.nodalMatrix, s.model.ac.nodalMatrixTranspose
    n = s.bus.number
    reim = lambda z: np.ascontiguousarray(np.stack([z.real, z.imag], axis=1), dtype=np.float64).tobytes()   # noqa: E731
    with open(path, "wb") as f:
        f.write(struct.pack("<qqqqd", n, Y.nnz, int(s.bus.layout.slack), max_iter, tol))
        f.write(np.ascontiguousarray(Y.colptr, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(Y.rowval, dtype=np.int64).tobytes())
        f.write(reim(Y.nzval)); f.write(reim(YT.nzval))
        f.write(np.ascontiguousarray(s.bus.layout.type, dtype=np.int8).tobytes())
        for a in (s.bus.supply.active - s.bus.demand.active, s.bus.supply.reactive - s.bus.demand.reactive, vm, va):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("name,iters,batch", [("case14test", 7, 1), ("case30test", 4, 70)])
def test_c_driver_hits_the_matpower_goldens(jg, tmp_path, name, iters, batch):
    exe = build(tmp_path)
    model, result = str(tmp_path / "model.bin"), str(tmp_path / "result.bin")
    n = write_model(jg, name, model)
    r = subprocess.run([exe, model, result, str(batch)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(result, "rb").read()
    rc, it, st = struct.unpack("<qqq", raw[:24])
    v = np.frombuffer(raw[24:], dtype=np.float64)
    g = load_golden(name)
    assert (rc, it, st) == (0, iters, 0)
    assert np.abs(v[:n] - g["newtonRaphson_voltageMagnitude"]).max() <= 1e-8
    assert np.abs(v[n:] - g["newtonRaphson_voltageAngle"]).max() <= 1e-8
