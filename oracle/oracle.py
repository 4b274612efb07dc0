"""ctypes binding of the CPU ORACLE (oracle/jg_oracle.c, oracle/jg_oracle_se.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (juliagrid.jl_amd) must never import this module.

Inputs are the raw case tables of tests/golden/cases/*.npz (per-unit, radians, 1-based indices);
everything downstream (Ybus, bus types, index maps, Jacobian, LU, NR loop) is computed by the C
restatement, independent of the product's host code.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
I8P = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
F64P = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libjg_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("jg_oracle.c", "jg_oracle_se.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.jgo_ac_model.restype = C.c_int64
        L.jgo_ac_model.argtypes = [C.c_int64, C.c_int64, I64P, I64P, I8P] + [F64P] * 8 + [I64P, I64P] + [F64P] * 5
        L.jgo_initialize_ac_power_flow.restype = C.c_int64
        L.jgo_initialize_ac_power_flow.argtypes = [C.c_int64, I8P, C.c_int64, I64P, F64P, F64P, F64P, F64P, F64P]
        L.jgo_nr_create.restype = C.c_void_p
        L.jgo_nr_create.argtypes = [C.c_int64, I64P, I64P, F64P, F64P, F64P, F64P, I8P, C.c_int64]
        L.jgo_nr_destroy.argtypes = [C.c_void_p]
        for f in ("jgo_nr_dim", "jgo_nr_nnz", "jgo_nr_iteration", "jgo_nr_lu_nnz"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.jgo_nr_set_power.argtypes = [C.c_void_p, F64P, F64P, F64P, F64P]
        L.jgo_nr_set_voltage.argtypes = [C.c_void_p, F64P, F64P]
        L.jgo_nr_get_voltage.argtypes = [C.c_void_p, F64P, F64P]
        L.jgo_nr_add_ybus.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double]
        L.jgo_nr_get_maps.argtypes = [C.c_void_p, I64P, I64P, I64P, I64P, I64P]
        L.jgo_nr_get_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.jgo_nr_mismatch.argtypes = [C.c_void_p, F64P]
        L.jgo_nr_solve.restype = C.c_int
        L.jgo_nr_solve.argtypes = [C.c_void_p]
        L.jgo_nr_power_flow.restype = C.c_int
        L.jgo_nr_power_flow.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleSystem:
    """Case tables + oracle Ybus (PF0)."""

    def __init__(self, tables):
        t = {k: np.array(tables[k]) for k in tables}
        self.t = t
        self.n = int(t["bus_type"].size)
        self.nb = int(t["br_from"].size)
        self.ng = int(t["gen_bus"].size)
        self.type = np.ascontiguousarray(t["bus_type"], dtype=np.int8).copy()
        s = np.flatnonzero(self.type == 3)
        self.slack = int(s[0]) + 1 if s.size else 1
        self.status = np.ascontiguousarray(t["br_status"], dtype=np.int8).copy()
        # bus.supply + first in-service generator per bus (load.jl:271-277)
        self.ps = np.zeros(self.n)
        self.qs = np.zeros(self.n)
        self.first_gen = np.zeros(self.n, dtype=np.int64)
        for k in range(self.ng):
            if t["gen_status"][k] == 1:
                i = int(t["gen_bus"][k]) - 1
                if self.first_gen[i] == 0:
                    self.first_gen[i] = k + 1
                self.ps[i] += t["gen_pg"][k]
                self.qs[i] += t["gen_qg"][k]
        self.pd = _f8(t["bus_pd"]).copy()
        self.qd = _f8(t["bus_qd"]).copy()
        self.ac_model()

    def ac_model(self):
        t, n, nb = self.t, self.n, self.nb
        cap = n + 2 * nb
        self.colptr = np.zeros(n + 1, dtype=np.int64)
        rowval = np.zeros(cap, dtype=np.int64)
        bufs = [np.zeros(cap) for _ in range(4)]
        self.twoport = np.zeros(nb * 10)
        nnz = lib().jgo_ac_model(
            n, nb, np.ascontiguousarray(t["br_from"], dtype=np.int64), np.ascontiguousarray(t["br_to"], dtype=np.int64),
            self.status, _f8(t["br_r"]), _f8(t["br_x"]), _f8(t["br_g"]), _f8(t["br_b"]), _f8(t["br_tap"]),
            _f8(t["br_shift"]), _f8(t["bus_gs"]), _f8(t["bus_bs"]), self.colptr, rowval, *bufs, self.twoport)
        self.nnz = int(nnz)
        self.rowval = rowval[:nnz].copy()
        self.yre, self.yim, self.ytre, self.ytim = (b[:nnz].copy() for b in bufs)

    @property
    def ybus(self):
        return self.yre + 1j * self.yim

    def ptr(self, row, col):
        """0-based pointer of stored entry (row, col), 1-based arguments."""
        lo, hi = self.colptr[col - 1] - 1, self.colptr[col] - 1
        p = lo + int(np.searchsorted(self.rowval[lo:hi], row))
        assert self.rowval[p] == row
        return p


class OracleNR:
    """newtonRaphson(system) + mismatch!/solve!/powerFlow! on the oracle."""

    def __init__(self, sys_: OracleSystem):
        L = lib()
        self.sys = sys_
        n = sys_.n
        self.type = sys_.type.copy()
        self.vm = np.zeros(n)
        self.va = np.zeros(n)
        slack = L.jgo_initialize_ac_power_flow(n, self.type, sys_.slack, sys_.first_gen, _f8(sys_.t["gen_vg"]),
                                               _f8(sys_.t["bus_vm"]), _f8(sys_.t["bus_va"]), self.vm, self.va)
        if slack == 0:
            raise RuntimeError("No generator buses with an in-service generator found in the power system.")
        self.slack = int(slack)
        self.h = L.jgo_nr_create(n, sys_.colptr, sys_.rowval, sys_.yre, sys_.yim, sys_.ytre, sys_.ytim, self.type, self.slack)
        self.dim = int(L.jgo_nr_dim(self.h))
        self.nnzJ = int(L.jgo_nr_nnz(self.h))
        L.jgo_nr_set_power(self.h, sys_.ps, sys_.qs, sys_.pd, sys_.qd)
        L.jgo_nr_set_voltage(self.h, self.vm, self.va)
        self.pq = np.zeros(n, dtype=np.int64)
        self.pvpq = np.zeros(n, dtype=np.int64)
        self.pcount = np.zeros(n, dtype=np.int64)
        self.jcolptr = np.zeros(self.dim + 1, dtype=np.int64)
        self.jrowval = np.zeros(self.nnzJ, dtype=np.int64)
        L.jgo_nr_get_maps(self.h, self.pq, self.pvpq, self.pcount, self.jcolptr, self.jrowval)

    def __del__(self):
        try:
            lib().jgo_nr_destroy(self.h)
        except Exception:
            pass

    def set_voltage(self, vm, va):
        lib().jgo_nr_set_voltage(self.h, _f8(vm), _f8(va))

    def set_power(self, ps, qs, pd, qd):
        lib().jgo_nr_set_power(self.h, _f8(ps), _f8(qs), _f8(pd), _f8(qd))

    def add_ybus(self, ptr, dy):
        lib().jgo_nr_add_ybus(self.h, int(ptr), float(np.real(dy)), float(np.imag(dy)))

    def voltage(self):
        vm, va = np.zeros(self.sys.n), np.zeros(self.sys.n)
        lib().jgo_nr_get_voltage(self.h, vm, va)
        return vm, va

    def mismatch(self):
        stop = np.zeros(2)
        lib().jgo_nr_mismatch(self.h, stop)
        return float(stop[0]), float(stop[1])

    def solve(self):
        rc = lib().jgo_nr_solve(self.h)
        if rc:
            raise RuntimeError(f"oracle LU failure {rc}")

    def vectors(self):
        j, m, i = np.zeros(self.nnzJ), np.zeros(self.dim), np.zeros(self.dim)
        lib().jgo_nr_get_vectors(self.h, j.ctypes.data, m.ctypes.data, i.ctypes.data)
        return j, m, i

    @property
    def iteration(self):
        return int(lib().jgo_nr_iteration(self.h))

    @property
    def lu_nnz(self):
        return int(lib().jgo_nr_lu_nnz(self.h))

    def power_flow(self, iteration=20, tolerance=1e-8):
        hist = np.zeros(2 * (iteration + 2))
        nh = C.c_int64(0)
        status = lib().jgo_nr_power_flow(self.h, iteration, tolerance, hist.ctypes.data, C.byref(nh))
        self.history = hist[: 2 * nh.value].reshape(-1, 2)
        return int(status)
