#!/usr/bin/env python3
"""One-off fixture converter (runs ONLY in the build container, where /root/reference exists).

Produces the committed DATA fixtures the GPU box needs (it has no /root/reference):

  tests/golden/cases/<name>.npz   power-system case tables (per-unit, radians, 1-based internal
                                  indices) = what the reference's loaders produce in memory
  tests/golden/results_<case>.npz MATPOWER golden vectors of test/data/results.h5

Loader conventions restated from the reference (behaviour only, no code copied):
  * HDF5 cases: src/powerSystem/load.jl:141-289, 1360-1367 -- a SCALAR dataset is a value
    broadcast to every element; from/to/generator-bus are already 1-based internal indices.
  * MATPOWER .m: src/powerSystem/load.jl:292-619 -- Pd,Qd,Gs,Bs,Pg,Qg are MULTIPLIED by
    1/baseMVA, angles by pi/180, ratio==0 -> 1, bus labels -> row order.

HDF5 is read through /opt/conda/bin/h5dump (h5py is not installed).
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
H5DUMP = "/opt/conda/bin/h5dump"
H5LS = "/opt/conda/bin/h5ls"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_CASES = os.path.join(ROOT, "tests", "golden", "cases")
OUT_GOLD = os.path.join(ROOT, "tests", "golden")

_DT = {"H5T_IEEE_F64LE": "<f8", "H5T_STD_I64LE": "<i8", "H5T_STD_I8LE": "i1", "H5T_STD_I32LE": "<i4"}


def h5_list(path):
    out = subprocess.check_output([H5LS, "-r", path], text=True)
    ds = []
    for line in out.splitlines():
        m = re.match(r"^(\S+)\s+Dataset\s+\{(.*)\}", line)
        if m:
            ds.append(m.group(1))
    return ds


def h5_read(path, dset):
    """Read one numeric dataset as a numpy array (None for strings)."""
    hdr = subprocess.check_output([H5DUMP, "-H", "-d", dset, path], text=True)
    m = re.search(r"DATATYPE\s+(\S+)", hdr)
    dt = _DT.get(m.group(1))
    if dt is None:
        return None
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.check_call([H5DUMP, "-d", dset, "-b", "LE", "-o", tmp.name, path],
                              stdout=subprocess.DEVNULL)
        arr = np.fromfile(tmp.name, dtype=dt)
    return arr


def bcast(a, n):
    a = np.asarray(a)
    return np.full(n, a.reshape(-1)[0], dtype=a.dtype) if a.size == 1 else a


def case_from_hdf5(path):
    g = lambda d: h5_read(path, d)
    btype = g("/bus/layout/type")
    n = btype.size
    frm = g("/branch/layout/from")
    nb = frm.size
    gbus = g("/generator/layout/bus")
    ng = gbus.size
    c = dict(
        base_power=np.float64(g("/base/power")[0]),
        bus_type=bcast(btype, n).astype(np.int8),
        bus_pd=bcast(g("/bus/demand/active"), n), bus_qd=bcast(g("/bus/demand/reactive"), n),
        bus_gs=bcast(g("/bus/shunt/conductance"), n), bus_bs=bcast(g("/bus/shunt/susceptance"), n),
        bus_vm=bcast(g("/bus/voltage/magnitude"), n), bus_va=bcast(g("/bus/voltage/angle"), n),
        br_from=frm.astype(np.int64), br_to=g("/branch/layout/to").astype(np.int64),
        br_status=bcast(g("/branch/layout/status"), nb).astype(np.int8),
        br_r=bcast(g("/branch/parameter/resistance"), nb), br_x=bcast(g("/branch/parameter/reactance"), nb),
        br_g=bcast(g("/branch/parameter/conductance"), nb), br_b=bcast(g("/branch/parameter/susceptance"), nb),
        br_tap=bcast(g("/branch/parameter/turnsRatio"), nb), br_shift=bcast(g("/branch/parameter/shiftAngle"), nb),
        gen_bus=gbus.astype(np.int64), gen_status=bcast(g("/generator/layout/status"), ng).astype(np.int8),
        gen_pg=bcast(g("/generator/output/active"), ng), gen_qg=bcast(g("/generator/output/reactive"), ng),
        gen_vg=bcast(g("/generator/voltage/magnitude"), ng),
        gen_qmin=bcast(g("/generator/capability/minReactive"), ng),
        gen_qmax=bcast(g("/generator/capability/maxReactive"), ng),
    )
    return {k: np.ascontiguousarray(v) for k, v in c.items()}


def _matrix(text, name):
    m = re.search(r"mpc\." + name + r"\s*=\s*\[(.*?)\];", text, re.S)
    rows = []
    for line in m.group(1).splitlines():
        line = line.split("%")[0].strip().rstrip(";").strip()
        if line:
            rows.append([float(t) for t in line.replace(",", " ").split()])
    return rows


def case_from_matpower(path):
    text = open(path).read()
    base = float(re.search(r"mpc\.baseMVA\s*=\s*([^;]+);", text).group(1))
    binv = 1.0 / base
    d2r = np.pi / 180
    bus = _matrix(text, "bus")
    gen = _matrix(text, "gen")
    br = _matrix(text, "branch")
    label = {int(r[0]): k + 1 for k, r in enumerate(bus)}
    n, nb, ng = len(bus), len(br), len(gen)
    f8 = lambda xs: np.array(xs, dtype=np.float64)
    tap = f8([r[8] for r in br])
    tap[tap == 0.0] = 1.0
    c = dict(
        base_power=np.float64(base * 1e6),
        bus_label=np.array([int(r[0]) for r in bus], dtype=np.int64),
        bus_type=np.array([int(r[1]) for r in bus], dtype=np.int8),
        bus_pd=f8([r[2] for r in bus]) * binv, bus_qd=f8([r[3] for r in bus]) * binv,
        bus_gs=f8([r[4] for r in bus]) * binv, bus_bs=f8([r[5] for r in bus]) * binv,
        bus_vm=f8([r[7] for r in bus]), bus_va=f8([r[8] for r in bus]) * d2r,
        br_from=np.array([label[int(r[0])] for r in br], dtype=np.int64),
        br_to=np.array([label[int(r[1])] for r in br], dtype=np.int64),
        br_status=np.array([int(r[10]) for r in br], dtype=np.int8),
        br_r=f8([r[2] for r in br]), br_x=f8([r[3] for r in br]),
        br_g=np.zeros(nb), br_b=f8([r[4] for r in br]),
        br_tap=tap, br_shift=f8([r[9] for r in br]) * d2r,
        gen_bus=np.array([label[int(r[0])] for r in gen], dtype=np.int64),
        gen_status=np.array([int(r[7]) for r in gen], dtype=np.int8),
        gen_pg=f8([r[1] for r in gen]) * binv, gen_qg=f8([r[2] for r in gen]) * binv,
        gen_vg=f8([r[5] for r in gen]),
        gen_qmax=f8([r[3] for r in gen]) * binv, gen_qmin=f8([r[4] for r in gen]) * binv,
    )
    assert (n, nb, ng) == (c["bus_type"].size, c["br_from"].size, c["gen_bus"].size)
    return c


def goldens(path, case):
    out = {}
    for d in h5_list(path):
        if (d.startswith("/" + case + "/newtonRaphson/") or d.startswith("/" + case + "/reactiveLimit/newtonRaphson/")
                or any(d == "/" + case + "/" + m + "/" + q for m in ("fastNewtonRaphsonBX", "fastNewtonRaphsonXB")
                       for q in ("iteration", "voltageMagnitude", "voltageAngle"))):
            a = h5_read(path, d)
            if a is not None:
                key = d[len(case) + 2:].replace("/", "_")
                out[key] = a
    return out


def main():
    os.makedirs(OUT_CASES, exist_ok=True)
    hdf5_cases = {
        "case14": f"{REF}/src/data/case14.h5",
        "case_ieee30": f"{REF}/docs/src/examples/cases/hdf5/case_ieee30.h5",
        "case118": f"{REF}/docs/src/examples/cases/hdf5/case118.h5",
        "case300": f"{REF}/docs/src/examples/cases/hdf5/case300.h5",
        "case1354pegase": f"{REF}/docs/src/examples/cases/hdf5/case1354pegase.h5",
        "case1951rte": f"{REF}/docs/src/examples/cases/hdf5/case1951rte.h5",
        "case_ACTIVSg10k": f"{REF}/docs/src/examples/cases/hdf5/case_ACTIVSg10k.h5",
    }
    for name, p in hdf5_cases.items():
        c = case_from_hdf5(p)
        np.savez_compressed(os.path.join(OUT_CASES, name + ".npz"), **c)
        print(name, "n=%d nb=%d ng=%d" % (c["bus_type"].size, c["br_from"].size, c["gen_bus"].size))
    for name in ("case14test", "case30test"):
        c = case_from_matpower(f"{REF}/test/data/{name}.m")
        np.savez_compressed(os.path.join(OUT_CASES, name + ".npz"), **c)
        g = goldens(f"{REF}/test/data/results.h5", name)
        np.savez_compressed(os.path.join(OUT_GOLD, f"results_{name}.npz"), **g)
        print(name, "n=%d" % c["bus_type"].size, "golden keys:", len(g), "iteration", g["newtonRaphson_iteration"])


if __name__ == "__main__":
    sys.exit(main())
