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


def build(tmp_path, name="nr_from_file"):
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "juliagrid.jl_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "c", name + ".c"),
                    "-L", libdir, "-ljgrid_hip", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("name", ["nr_from_file", "gn_from_file"])
def test_c_driver_compiles_and_links(jg, tmp_path, name):
    exe = build(tmp_path, name)
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


def write_se_model(jg, name, path, max_iter=200, tol=1e-12):
    """gaussNewton(monitoring) of the reference's known-answer test (test/utility/utility.jl:282-286): exact measurements of
    a converged power flow -- voltmeters, watt / varmeters and PMUs (rectangular, one of them correlated) -- from the system's
    stored start point; written as the flat arrays jg_gn_create / jg_gn_set_measurement take."""
    s = jg.powerSystem(load_case(name))
    pf = jg.newtonRaphson(s)
    jg.powerFlow_(pf, tolerance=1e-12)
    mon = jg.measurement(s)
    jg.addVoltmeter_(mon, pf, variance=1e-4)
    jg.addWattmeter_(mon, pf, variance=1e-4)
    jg.addVarmeter_(mon, pf, variance=1e-4)
    jg.addPmu_(mon, pf, correlated=True, minMagnitude=1e-6)
    an = jg.gaussNewton(mon)                                   # the Python mirror derives exactly what the Julia shim passes
    ac, br = s.model.ac, s.branch
    Y, YT = ac.nodalMatrix, ac.nodalMatrixTranspose
    n, nb, m = s.bus.number, br.number, an.dims["m"]
    me = an.method
    corr = np.asarray(me._corr, dtype=np.int64)
    reim = lambda z: np.ascontiguousarray(np.stack([z.real, z.imag], axis=1), dtype=np.float64).tobytes()   # noqa: E731
    par = br.parameter
    bp = np.ascontiguousarray(np.stack([ac.admittance.real, ac.admittance.imag, par.conductance, par.susceptance,
                                        par.turnsRatio, par.shiftAngle], axis=1), dtype=np.float64)
    status = (np.asarray(me.type) != 0).astype(np.int8)
    with open(path, "wb") as f:
        f.write(struct.pack("<qqqqqqqd", n, Y.nnz, nb, int(s.bus.layout.slack), m, corr.size, max_iter, tol))
        f.write(np.ascontiguousarray(Y.colptr, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(Y.rowval, dtype=np.int64).tobytes())
        f.write(reim(Y.nzval)); f.write(reim(YT.nzval))
        f.write(np.ascontiguousarray(br.layout.from_, dtype=np.int64).tobytes())
        f.write(np.ascontiguousarray(br.layout.to, dtype=np.int64).tobytes())
        f.write(bp.tobytes())
        f.write(np.ascontiguousarray(me._code, dtype=np.int8).tobytes())
        f.write(status.tobytes())
        f.write(np.ascontiguousarray(me.index, dtype=np.int64).tobytes())
        f.write((corr if corr.size else np.zeros(1, dtype=np.int64)).tobytes())
        woff = np.asarray(me._woff, dtype=np.float64).reshape(-1)
        for a in (np.asarray(me.mean).reshape(-1)[:m], np.asarray(me._wdiag).reshape(-1)[:m], woff if corr.size else np.zeros(1),
                  s.bus.voltage.magnitude, s.bus.voltage.angle):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    truth = (pf.voltage.magnitude.copy(), pf.voltage.angle.copy())
    an.close()
    pf.close()
    return n, truth


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("case14", 1), ("case_ieee30", 70)])
def test_c_driver_recovers_the_power_flow_state_from_exact_measurements(jg, tmp_path, name, batch):
    """The reference's acceptance rule for gaussNewton (test/stateEstimation/analysis.jl:27-171): exact measurements of a
    converged power flow => the estimate equals its state to 1e-10 -- through the C ABI from a plain C program, with the fused
    loop and with the caller's own increment! / solve! loop (equal iteration counts, equal states)."""
    exe = build(tmp_path, "gn_from_file")
    model, result = str(tmp_path / "model.bin"), str(tmp_path / "result.bin")
    n, (vm, va) = write_se_model(jg, name, model)
    r = subprocess.run([exe, model, result, str(batch)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(result, "rb").read()
    rc, it, st, own = struct.unpack("<qqqq", raw[:32])
    v = np.frombuffer(raw[32:], dtype=np.float64)
    assert (rc, st) == (0, 0) and it == own and 2 <= it < 30
    for k in (0, 2):
        assert np.abs(v[k * n:(k + 1) * n] - vm).max() <= 1e-10
        assert np.abs(v[(k + 1) * n:(k + 2) * n] - va).max() <= 1e-10
    assert np.abs(v[4 * n:6 * n]).max() < 1e-12                 # the last increment is what stopped the loop
