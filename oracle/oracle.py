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


def build_fast():
    """A second build of the same sources for the cpu_baseline legs of bench.py ONLY: `-O3 -march=native`, contraction allowed -- the
    parity checks keep the `-O2 -ffp-contract=off` build.  Compiled on the box it runs on (the flags are per CPU): the file name
    carries a hash of the CPU's model and flags, so a copy that travelled from another machine is never loaded."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            cpu = "".join(l for l in f if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        cpu = "unknown"
    tag = hashlib.sha1(cpu.encode()).hexdigest()[:10]
    out = os.path.join(_HERE, "_fast")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, f"libjg_oracle_fast_{tag}.so")
    srcs = [os.path.join(_HERE, f) for f in ("jg_oracle.c", "jg_oracle_se.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c99", "-shared", "-o", so + ".tmp"] + srcs + ["-lm"])
        os.replace(so + ".tmp", so)
    return so


FAST_BUILD = False          # True once lib() has loaded the -O3 -march=native build (JG_ORACLE_FAST=1: bench.py's baseline legs)


def lib():
    global _LIB, FAST_BUILD
    if _LIB is None:
        FAST_BUILD = os.environ.get("JG_ORACLE_FAST") == "1"
        L = C.CDLL(build_fast() if FAST_BUILD else build())
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


# ----------------------------------------------------------------------------------------------
# Gauss-Newton WLS state estimation (oracle/jg_oracle_se.c)
# ----------------------------------------------------------------------------------------------
def _se_lib():
    L = lib()
    if not hasattr(L, "_se_bound"):
        L.jgo_exact_quantities.argtypes = [C.c_int64, C.c_int64, I64P, I64P, I8P, F64P, I64P, I64P, F64P, F64P, F64P, F64P, F64P, F64P]
        L.jgo_gn_create.restype = C.c_void_p
        L.jgo_gn_create.argtypes = ([C.c_int64, C.c_int64, I64P, I64P, F64P, F64P, F64P, F64P, I64P, I64P, F64P, F64P, F64P, F64P, F64P,
                                     C.c_int64, C.c_int64, I8P, I8P, I64P, F64P, F64P, I8P, F64P, F64P, I8P, I8P, F64P, F64P])
        L.jgo_gn_destroy.argtypes = [C.c_void_p]
        for f in ("jgo_gn_rows", "jgo_gn_nnz", "jgo_gn_iteration"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.jgo_gn_objective.restype = C.c_double
        L.jgo_gn_objective.argtypes = [C.c_void_p]
        L.jgo_gn_get_model.argtypes = [C.c_void_p, I8P, I64P, I64P, F64P, F64P, F64P, I64P, I64P]
        L.jgo_gn_get_vectors.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.jgo_gn_set_voltage.argtypes = [C.c_void_p, F64P, F64P]
        L.jgo_gn_set_mean.argtypes = [C.c_void_p, F64P]
        L.jgo_gn_normal_equation.argtypes = [C.c_void_p]
        L.jgo_gn_increment.restype = C.c_int
        L.jgo_gn_increment.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.jgo_gn_solve.argtypes = [C.c_void_p]
        L.jgo_gn_state_estimation.restype = C.c_int
        L.jgo_gn_state_estimation.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p]
        L._se_bound = True
    return L


KIND = dict(voltmeter=1, ammeter=2, wattmeter=3, varmeter=4, pmu=5)


class MeterTable:
    """Flat device table in the reference's concatenation order (voltmeters, ammeters, wattmeters,
    varmeters, PMUs): what measurement(system) + add*!(monitoring, pf) hold."""

    FIELDS = ("kind", "loc", "index", "mean1", "var1", "status1", "mean2", "var2", "status2", "flags")

    def __init__(self):
        self.rows = []

    def add(self, kind, loc, index, mean1, var1, status1=1, mean2=0.0, var2=1.0, status2=1, square=False, polar=False,
            correlated=False):
        self.rows.append((KIND[kind], loc, int(index), float(mean1), float(var1), int(status1), float(mean2), float(var2),
                          int(status2), int(square) | (int(polar) << 1) | (int(correlated) << 2)))

    def arrays(self):
        rows = sorted(self.rows, key=lambda r: r[0])          # stable: family order, insertion order inside
        cols = list(zip(*rows)) if rows else [[] for _ in self.FIELDS]
        dt = (np.int8, np.int8, np.int64, np.float64, np.float64, np.int8, np.float64, np.float64, np.int8, np.int8)
        return {f: np.ascontiguousarray(c, dtype=d) for f, c, d in zip(self.FIELDS, cols, dt)}


def exact_quantities(sys_: OracleSystem, vm, va):
    L = _se_lib()
    br = np.zeros(sys_.nb * 8)
    bus = np.zeros(sys_.n * 2)
    t = sys_.t
    L.jgo_exact_quantities(sys_.n, sys_.nb, np.ascontiguousarray(t["br_from"], dtype=np.int64),
                           np.ascontiguousarray(t["br_to"], dtype=np.int64), sys_.status, sys_.twoport, sys_.colptr,
                           sys_.rowval, sys_.ytre, sys_.ytim, _f8(vm), _f8(va), br, bus)
    return br.reshape(-1, 8), bus.reshape(-1, 2)


def add_from_power_flow(tab: MeterTable, sys_: OracleSystem, vm, va, family, bus=True, frm=True, to=True, variance=None,
                        **flags):
    """add<Family>!(monitoring, pf): every bus in index order, then for each IN-SERVICE branch its from
    end then its to end (measurement/powermeter.jl:479-523, pmu.jl:327-400), exact noise-free values."""
    br, bq = exact_quantities(sys_, vm, va)
    on = np.flatnonzero(sys_.status == 1)
    var = variance if variance is not None else (1e-8 if family == "pmu" else 1e-4)
    if family == "voltmeter":
        for i in range(sys_.n):
            tab.add("voltmeter", 0, i + 1, vm[i], var)
        return
    if family == "ammeter":
        for k in on:
            if frm:
                tab.add("ammeter", 1, k + 1, br[k, 4], var, **flags)
            if to:
                tab.add("ammeter", 2, k + 1, br[k, 6], var, **flags)
        return
    if family in ("wattmeter", "varmeter"):
        c = 0 if family == "wattmeter" else 1
        if bus:
            for i in range(sys_.n):
                tab.add(family, 0, i + 1, bq[i, c], var)
        for k in on:
            if frm:
                tab.add(family, 1, k + 1, br[k, c], var)
            if to:
                tab.add(family, 2, k + 1, br[k, 2 + c], var)
        return
    if family == "pmu":
        if bus:
            bflags = dict(flags, square=False)                 # pmu.jl:335: layout.square = false for bus PMUs
            for i in range(sys_.n):
                tab.add("pmu", 0, i + 1, vm[i], var, 1, va[i], var, 1, **bflags)
        for k in on:
            if frm:
                tab.add("pmu", 1, k + 1, br[k, 4], var, 1, br[k, 5], var, 1, **flags)
            if to:
                tab.add("pmu", 2, k + 1, br[k, 6], var, 1, br[k, 7], var, 1, **flags)
        return
    raise ValueError(family)


class OracleGN:
    """gaussNewton(monitoring) + increment!/solve!/stateEstimation! on the oracle."""

    def __init__(self, sys_: OracleSystem, table, vm0=None, va0=None):
        L = _se_lib()
        self.sys = sys_
        t = sys_.t
        a = table.arrays() if isinstance(table, MeterTable) else table
        self.table = a
        vm0 = _f8(t["bus_vm"]) if vm0 is None else _f8(vm0)
        va0 = _f8(t["bus_va"]) if va0 is None else _f8(va0)
        self.h = L.jgo_gn_create(
            sys_.n, sys_.nb, sys_.colptr, sys_.rowval, sys_.yre, sys_.yim, sys_.ytre, sys_.ytim,
            np.ascontiguousarray(t["br_from"], dtype=np.int64), np.ascontiguousarray(t["br_to"], dtype=np.int64),
            sys_.twoport, _f8(t["br_g"]), _f8(t["br_b"]), _f8(t["br_tap"]), _f8(t["br_shift"]), sys_.slack,
            a["kind"].size, a["kind"], a["loc"], a["index"], a["mean1"], a["var1"], a["status1"], a["mean2"], a["var2"],
            a["status2"], a["flags"], vm0, va0)
        self.m = int(L.jgo_gn_rows(self.h))
        self.nnzH = int(L.jgo_gn_nnz(self.h))
        n = sys_.n
        self.type = np.zeros(self.m, dtype=np.int8)
        self.index = np.zeros(self.m, dtype=np.int64)
        self.range = np.zeros(6, dtype=np.int64)
        self.mean = np.zeros(self.m)
        self.wdiag = np.zeros(self.m)
        self.woff = np.zeros(self.m)
        self.hcolptr = np.zeros(2 * n + 1, dtype=np.int64)
        self.hrowval = np.zeros(self.nnzH, dtype=np.int64)
        L.jgo_gn_get_model(self.h, self.type, self.index, self.range, self.mean, self.wdiag, self.woff, self.hcolptr, self.hrowval)

    def __del__(self):
        try:
            _se_lib().jgo_gn_destroy(self.h)
        except Exception:
            pass

    def precision_dense(self):
        W = np.diag(self.wdiag)
        for r in np.flatnonzero(self.woff):
            W[r, r + 1] = W[r + 1, r] = self.woff[r]
        return W

    def vectors(self):
        n = self.sys.n
        hv, res, inc, vm, va = np.zeros(self.nnzH), np.zeros(self.m), np.zeros(2 * n), np.zeros(n), np.zeros(n)
        _se_lib().jgo_gn_get_vectors(self.h, hv.ctypes.data, res.ctypes.data, inc.ctypes.data, vm.ctypes.data, va.ctypes.data)
        return dict(jacobian=hv, residual=res, increment=inc, magnitude=vm, angle=va)

    def set_voltage(self, vm, va):
        _se_lib().jgo_gn_set_voltage(self.h, _f8(vm), _f8(va))

    def set_mean(self, mean):
        _se_lib().jgo_gn_set_mean(self.h, _f8(mean))

    def increment(self):
        mx = C.c_double(0.0)
        rc = _se_lib().jgo_gn_increment(self.h, C.byref(mx))
        if rc:
            raise RuntimeError(f"oracle gain factorisation failure {rc}")
        return mx.value

    def solve(self):
        _se_lib().jgo_gn_solve(self.h)

    @property
    def iteration(self):
        return int(_se_lib().jgo_gn_iteration(self.h))

    @property
    def objective(self):
        return float(_se_lib().jgo_gn_objective(self.h))

    # ---- increment!(::AcStateEstimation{GaussNewton{Orthogonal}}) / {PetersWilkinson} restated in dense numpy / scipy
    # (acStateEstimation.jl:906-932 and :934-971; small cases only).  Both start from normalEquation! (the C restatement,
    # through increment()), remove the slack angle column (:914 / :947) and scale by sqrt(W) (sqrtPrecision!,
    # dcStateEstimation.jl:488-492: diagonal precision only).
    def _scaled_system(self):
        import scipy.sparse as sp
        n = self.sys.n
        self.increment()                                              # normalEquation!: H and the residual at the current state
        v = self.vectors()
        assert not np.any(self.woff), "Orthogonal / PetersWilkinson need a diagonal precision matrix"
        H = sp.csc_matrix((v["jacobian"], self.hrowval - 1, self.hcolptr - 1), shape=(self.m, 2 * n)).toarray()
        sw = np.sqrt(self.wdiag)
        H = sw[:, None] * H
        H[:, self.sys.slack - 1] = 0.0                                # removeColumn(jacobian, slack)
        return H, sw * v["residual"], v

    def increment_orthogonal(self):
        """:906-932: qr(sqrt(W) H) \\ (sqrt(W) r), increment[slack] = 0."""
        H, z, _ = self._scaled_system()
        keep = np.flatnonzero(np.arange(H.shape[1]) != self.sys.slack - 1)
        Q, R = np.linalg.qr(H[:, keep])
        inc = np.zeros(H.shape[1])
        inc[keep] = np.linalg.solve(R, Q.T @ z)
        return inc

    def increment_peters_wilkinson(self):
        """:934-971: lu([sqrt(W) H; e_slack']) = L U (row permutation p), y = (L'L) \\ (L' z[p]), increment = U \\ y, increment[slack] = 0."""
        import scipy.linalg as sl
        H, z, _ = self._scaled_system()
        e = np.zeros((1, H.shape[1]))
        e[0, self.sys.slack - 1] = 1.0
        P, Lw, U = sl.lu(np.vstack([H, e]))                           # [H; e] = P L U, L (m+1) x 2n unit lower trapezoidal
        zp = P.T @ np.append(z, 0.0)
        y = np.linalg.solve(Lw.T @ Lw, Lw.T @ zp)
        inc = sl.solve_triangular(U, y)
        inc[self.sys.slack - 1] = 0.0
        return inc

    def state_estimation_with(self, increment, iteration=40, tolerance=1e-8):
        """stateEstimation! (:1286-1329) around one of the two increments above; returns (converged, iterations)."""
        n = self.sys.n
        it = 0
        for _ in range(iteration + 1):
            inc = increment()
            self.last_increment = inc
            if np.max(np.abs(inc)) < tolerance:
                return True, it
            if it == iteration:
                break
            v = self.vectors()
            self.set_voltage(v["magnitude"] + inc[n:], v["angle"] + inc[:n])    # solve! :1035-1047
            it += 1
        return False, it

    def state_estimation(self, iteration=40, tolerance=1e-8):
        hist = np.zeros(iteration + 2)
        nh = C.c_int64(0)
        st = _se_lib().jgo_gn_state_estimation(self.h, iteration, tolerance, hist.ctypes.data, C.byref(nh))
        self.history = hist[: nh.value]
        return int(st)


# ---- power!(analysis) / current!(analysis) restated (TEST ORACLE; src/postprocessing/acAnalysis.jl) ------------------
def power_and_current(sys_: OracleSystem, vm, va):
    """One voltage profile -> dict of the reference's power! (:30-169) and current! (:672-704) containers.
    Branch from/to powers and currents and the bus injections come from the C restatement (jgo_exact_quantities,
    :891-931); series (:906-908, :929-931), charging (:910-919), shunt (:884-889), supply (:53-61) and the generator
    allocation (:84-166) are restated here line by line in numpy / Python loops."""
    t = sys_.t
    n, nb, ng = sys_.n, sys_.nb, sys_.ng
    vm, va = _f8(vm), _f8(va)
    br, bq = exact_quantities(sys_, vm, va)
    on = sys_.status == 1
    V = vm * np.exp(1j * va)
    f, to = t["br_from"].astype(int) - 1, t["br_to"].astype(int) - 1
    tau_inv = 1.0 / _f8(t["br_tap"])
    tij = tau_inv * np.exp(-1j * _f8(t["br_shift"]))                                   # ViVjVij :846-851
    Vij = tij * V[f] - V[to]
    tp = sys_.twoport.reshape(nb, 10)
    y = tp[:, 0] + 1j * tp[:, 1]
    Is = np.where(on, y * Vij, 0)                                                      # IsPsis :929-931
    Sl = Vij * np.conj(Is)                                                             # PlQl :906-908
    Sc = np.where(on, 0.5 * np.conj(_f8(t["br_g"]) + 1j * _f8(t["br_b"])) * ((tau_inv * vm[f]) ** 2 + vm[to] ** 2), 0)   # PcQc :910-919
    Ss = vm ** 2 * np.conj(_f8(t["bus_gs"]) + 1j * _f8(t["bus_bs"]))                   # PsQs :884-889
    P, Q = bq[:, 0], bq[:, 1]
    slack = sys_.slack - 1
    sup_p = sys_.ps.copy()
    sup_q = np.where(sys_.type != 1, Q + sys_.qd, sys_.qs)                             # :53-59
    sup_p[slack] = P[slack] + sys_.pd[slack]                                           # :61
    S = P + 1j * Q
    Ii = np.conj(S / V)                                                                # Ii :867-882 (I = conj(S / V))
    # generators :84-166
    gbus = t["gen_bus"].astype(int) - 1
    gstat = t["gen_status"].astype(int)
    gpg = _f8(t["gen_pg"])
    qmin = _f8(t["gen_qmin"]) if "gen_qmin" in t else np.zeros(ng)
    qmax = _f8(t["gen_qmax"]) if "gen_qmax" in t else np.zeros(ng)
    at_bus = {}
    for k in range(ng):
        if gstat[k] == 1:
            at_bus.setdefault(int(gbus[k]), []).append(k)
    base_mva = float(np.asarray(t.get("base_power", 1e8)).reshape(-1)[0]) * 1e-6
    gp, gq = np.zeros(ng), np.zeros(ng)
    for i in range(ng):
        if gstat[i] != 1:
            continue
        ib = int(gbus[i])
        idx = at_bus[ib]
        if len(idx) == 1:
            gp[i] = gpg[i]
            gq[i] = Q[ib] + sys_.qd[ib]
            if ib == slack:
                gp[i] = P[ib] + sys_.pd[ib]
            continue
        qmins = sum(qmin[j] for j in idx if not np.isinf(qmin[j]))
        qmaxs = sum(qmax[j] for j in idx if not np.isinf(qmax[j]))
        qgen = Q[ib] + sys_.qd[ib]
        qmin_inf = qmax_inf = 0.0
        qmin_new, qmax_new = qmin[i], qmax[i]
        for j in idx:
            if np.isinf(qmin[j]):
                v = -abs(qgen) - abs(qmins) - abs(qmaxs)
                if qmin[j] == np.inf:
                    v = -v
                if i == j:
                    qmin_new = v
                qmin_inf += v
            if np.isinf(qmax[j]):
                v = abs(qgen) + abs(qmins) + abs(qmaxs)
                if qmax[j] == -np.inf:
                    v = -v
                if i == j:
                    qmax_new = v
                qmax_inf += v
        qmins += qmin_inf
        qmaxs += qmax_inf
        if base_mva * abs(qmins - qmaxs) > 10 * np.finfo(float).eps:
            gq[i] = qmin_new + ((qgen - qmins) / (qmaxs - qmins)) * (qmax_new - qmin_new)
        else:
            gq[i] = qmin_new + (qgen - qmins) / len(idx)
        if ib == slack and idx[0] == i:
            gp[i] = P[ib] + sys_.pd[ib] - sum(gpg[j] for j in idx[1:])
        else:
            gp[i] = gpg[i]
    return dict(injection=(P, Q), shunt=(Ss.real, Ss.imag), supply=(sup_p, sup_q), from_=(br[:, 0], br[:, 1]), to=(br[:, 2], br[:, 3]),
                series=(np.where(on, Sl.real, 0), np.where(on, Sl.imag, 0)), charging=(Sc.real, Sc.imag), generator=(gp, gq),
                i_injection=(np.abs(Ii), np.angle(Ii)), i_from=(br[:, 4], br[:, 5]), i_to=(br[:, 6], br[:, 7]),
                i_series=(np.abs(Is), np.where(on, np.angle(Is), 0)))


def reactive_limit(sys_: OracleSystem, nr_type, vm, va):
    """reactiveLimit!(analysis) restated (src/powerFlow/acPowerFlow.jl:1081-1155) on the oracle's system container:
    generator outputs from power! (:1093-1103), violated PV / slack buses become PQ with Q pinned at the limit
    (:1105-1130), a converted slack hands over to the first generator bus (:1131-1146).  Mutates sys_ (type, slack,
    supply, gen_pg / gen_qg) like the reference mutates `system`; returns the violate vector."""
    t = sys_.t
    saved_type = sys_.type
    sys_.type = np.ascontiguousarray(nr_type, dtype=np.int8).copy()      # the bus types newtonRaphson() settled on
    r = power_and_current(sys_, vm, va)
    gp, gq = r["generator"]
    ng = sys_.ng
    violate = np.zeros(ng, dtype=np.int64)
    sys_.ps[:] = 0.0
    sys_.qs[:] = 0.0
    out_q = np.zeros(ng)
    gstat, gbus = t["gen_status"].astype(int), t["gen_bus"].astype(int) - 1
    pg = _f8(t["gen_pg"]).copy()
    qg = _f8(t["gen_qg"]).copy()
    for k in range(ng):                                                   # label order = index order in our fixtures
        if gstat[k] == 1:
            pg[k] = gp[k]
            sys_.ps[gbus[k]] += gp[k]
            sys_.qs[gbus[k]] += gq[k]
            out_q[k] = gq[k]
    qmin, qmax = _f8(t["gen_qmin"]), _f8(t["gen_qmax"])
    for i in range(ng):
        if gstat[i] == 0 or not (qmin[i] < qmax[i]):
            continue
        j = int(gbus[i])
        lo, hi = out_q[i] < qmin[i], out_q[i] > qmax[i]
        if sys_.type[j] != 1 and (lo or hi):
            if lo:
                violate[i], new_q = -1, qmin[i]
            if hi:
                violate[i], new_q = 1, qmax[i]
            sys_.type[j] = 1
            sys_.qs[j] -= out_q[i]
            qg[i] = new_q
            sys_.qs[j] += new_q
            if j == sys_.slack - 1:
                for k in range(sys_.n):
                    if sys_.type[k] == 2:
                        sys_.slack = k + 1
                        sys_.type[k] = 3
                        break
    if sys_.type[sys_.slack - 1] != 3:
        sys_.type = saved_type
        raise RuntimeError("The slack bus is not defined.")
    t["gen_pg"], t["gen_qg"] = pg, qg
    t["bus_type"] = sys_.type.copy()
    return violate


class OracleFastNR:
    """fastNewtonRaphsonBX / XB restated (TEST ORACLE, numpy + scipy.sparse.linalg.splu):
      fastNewtonRaphsonModel, fastNewtonJacobian, fastNewtonJacobian!, jacobianCoefficient   acPowerFlow.jl:259-506
      mismatch!                                                                              acPowerFlow.jl:687-730
      solve!                                                                                 acPowerFlow.jl:913-983
      powerFlow!                                                                             acPowerFlow.jl:1389-1433
    Bus types / start voltages come from the same initializeACPowerFlow restatement the Newton-Raphson oracle uses."""

    def __init__(self, sys_: OracleSystem, bx: bool):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spl
        self.sys, self.bx = sys_, bool(bx)
        nr = OracleNR(sys_)                                    # type normalisation + start point (:1312-1358)
        self.type, self.slack = nr.type.copy(), nr.slack
        self.vm, self.va = nr.vm.copy(), nr.va.copy()
        t, n = sys_.t, sys_.n
        typ = self.type
        self.pq = np.zeros(n, dtype=np.int64)
        self.pvpq = np.zeros(n, dtype=np.int64)
        npq = npvpq = 0
        for i in range(n):                                     # fastNewtonJacobian :343-356
            if typ[i] == 1:
                npq += 1
                self.pq[i] = npq
            if typ[i] != 3:
                npvpq += 1
                self.pvpq[i] = npvpq
        self.npq = npq
        cp, rv = sys_.colptr, sys_.rowval
        # patterns with stored zeros, column by column of Ybus (:358-404)
        colP, rowP, colQ, rowQ = [], [], [], []
        for i in range(n):
            if i + 1 == self.slack:
                continue
            for p in range(cp[i] - 1, cp[i + 1] - 1):
                r = rv[p] - 1
                if typ[r] != 3:
                    colP.append(self.pvpq[i] - 1)
                    rowP.append(self.pvpq[r] - 1)
                if typ[i] == 1 and typ[r] == 1:
                    colQ.append(self.pq[i] - 1)
                    rowQ.append(self.pq[r] - 1)
        P = sp.csc_matrix((np.zeros(len(rowP)), (rowP, colP)), shape=(n - 1, n - 1)).tolil()
        Q = sp.csc_matrix((np.zeros(len(rowQ)), (rowQ, colQ)), shape=(npq, npq)).tolil()
        tp = sys_.twoport.reshape(sys_.nb, 10)
        for k in range(sys_.nb):                               # fastNewtonJacobian! over in-service branches (:322-326, 407-447)
            if sys_.status[k] != 1:
                continue
            i, j = int(t["br_from"][k]) - 1, int(t["br_to"][k]) - 1
            bsi = 0.5 * float(t["br_b"][k])
            tinv = 1.0 / float(t["br_tap"][k])
            sn, cs = np.sin(float(t["br_shift"][k])), np.cos(float(t["br_shift"][k]))
            yre, yim = tp[k, 0], tp[k, 1]
            if self.bx:                                        # jacobianCoefficient :449-474
                bmk, A, B = -1.0 / float(t["br_x"][k]), yre, yim
            else:
                bmk, A, B = yim, 0.0, -1.0 / float(t["br_x"][k])
            qA, qB, qC = -bmk * tinv, (bmk + bsi) * tinv ** 2, bmk + bsi
            den = cs * cs + sn * sn
            m, nn = self.pvpq[i] - 1, self.pvpq[j] - 1
            if i + 1 != self.slack and j + 1 != self.slack:
                P[m, nn] += (-A * sn - B * cs) / den           # Pij_theta_ij :476-481
                P[nn, m] += (A * sn - B * cs) / den
            if i + 1 != self.slack:
                P[m, m] += B / den
            if j + 1 != self.slack:
                P[nn, nn] += B
            ri, rj = self.pq[i] - 1, self.pq[j] - 1
            if ri >= 0 and rj >= 0:
                Q[ri, rj] += qA
                Q[rj, ri] += qA
            if typ[i] == 1:
                Q[ri, ri] += qB
            if typ[j] == 1:
                Q[rj, rj] += qC
        for i in range(n):                                     # shunt susceptance on the Q diagonal (:328-334)
            if typ[i] == 1 and t["bus_bs"][i] != 0:
                Q[self.pq[i] - 1, self.pq[i] - 1] += float(t["bus_bs"][i])
        self.P, self.Q = P.tocsc(), Q.tocsc()
        self.luP, self.luQ = spl.splu(self.P), spl.splu(self.Q)
        self.iteration = 0
        self.mismP = np.zeros(n - 1)
        self.mismQ = np.zeros(npq)

    def _row_sums(self, i, want_q):
        s = self.sys
        cp, rv = s.colptr, s.rowval
        cur_p = cur_q = 0.0
        for p in range(cp[i] - 1, cp[i + 1] - 1):
            r = rv[p] - 1
            g, b = s.ytre[p], s.ytim[p]                         # GijBijθij reads nodalMatrixTranspose.nzval (T1)
            d = self.va[i] - self.va[r]
            sn, cs = np.sin(d), np.cos(d)
            cur_p += self.vm[r] * (g * cs + b * sn)             # PiQiSumPlus
            if want_q:
                cur_q += self.vm[r] * (g * sn - b * cs)         # PiQiSumMinus
        return cur_p, cur_q

    def mismatch(self):
        s = self.sys
        stop_p = stop_q = 0.0
        for i in range(s.n):
            if i + 1 == self.slack:
                continue
            is_pq = self.type[i] == 1
            cp_, cq_ = self._row_sums(i, is_pq)
            vinv = 1.0 / self.vm[i]
            self.mismP[self.pvpq[i] - 1] = cp_ - (s.ps[i] - s.pd[i]) * vinv
            stop_p = max(stop_p, abs(self.mismP[self.pvpq[i] - 1]))
            if is_pq:
                self.mismQ[self.pq[i] - 1] = cq_ - (s.qs[i] - s.qd[i]) * vinv
                stop_q = max(stop_q, abs(self.mismQ[self.pq[i] - 1]))
        return stop_p, stop_q

    def solve(self):
        s = self.sys
        inc = self.luP.solve(self.mismP)
        for i in range(s.n):
            if i + 1 != self.slack:
                self.va[i] += inc[self.pvpq[i] - 1]
        for i in range(s.n):
            if self.type[i] == 1:
                _, cq_ = self._row_sums(i, True)
                self.mismQ[self.pq[i] - 1] = cq_ - (s.qs[i] - s.qd[i]) / self.vm[i]
        incq = self.luQ.solve(self.mismQ)
        for i in range(s.n):
            if self.type[i] == 1:
                self.vm[i] += incq[self.pq[i] - 1]
        self.iteration += 1

    def power_flow(self, iteration=20, tolerance=1e-8):
        self.iteration = 0
        for _ in range(iteration + 1):
            dp, dq = self.mismatch()
            if dp < tolerance and dq < tolerance:
                return 0
            if self.iteration == iteration:
                return 1
            self.solve()
        return 1


# ---- pmuStateEstimation(monitoring) + solve! restated (TEST ORACLE; src/stateEstimation/pmuStateEstimation.jl) -------
class OraclePmuWLS:
    """Linear WLS with PMUs only: coefficient / mean / precision as pmuEstimationWls builds them (:72-177), then
    solve!{Normal} (:369-399): gain = H' W H, b = H' W z, sparse LU, (Re, Im) -> polar.  Plain numpy / scipy.
    Pinned by the reference's own acceptance test (test/stateEstimation/analysis.jl:347-440, testPmuEstimation in
    test/utility/utility.jl:293-297): PMUs built from a solved power flow return its voltages to 1e-10."""

    def __init__(self, sys_: OracleSystem, table):
        import scipy.sparse as sp
        a = table.arrays() if isinstance(table, MeterTable) else table
        sel = np.flatnonzero(a["kind"] == KIND["pmu"])
        t, n = sys_.t, sys_.n
        y = 1.0 / (_f8(t["br_r"]) + 1j * _f8(t["br_x"]))               # ac.admittance (model.jl:55)
        g, b = y.real, y.imag
        gs, bs = 0.5 * _f8(t["br_g"]), 0.5 * _f8(t["br_b"])
        tinv, phi = 1.0 / _f8(t["br_tap"]), _f8(t["br_shift"])
        frm, to = np.asarray(t["br_from"], dtype=np.int64) - 1, np.asarray(t["br_to"], dtype=np.int64) - 1
        m = 2 * sel.size
        rows, cols, vals = [], [], []
        self.mean = np.zeros(m)
        W = sp.lil_matrix((m, m))
        for q, d in enumerate(sel):
            r = 2 * q
            zm, vm_, zs = a["mean1"][d], a["var1"][d], a["mean2"][d]
            va_ = a["var2"][d]
            s, c = np.sin(zs), np.cos(zs)
            vre = vm_ * c ** 2 + va_ * (zm * s) ** 2                     # variancePmu (equations.jl:576-588)
            vim = vm_ * s ** 2 + va_ * (zm * c) ** 2
            if a["flags"][d] & 4:                                        # correlated: inverse of the 2x2 covariance (:591-666)
                cov = s * c * (vm_ - va_ * zm ** 2)
                blk = np.linalg.inv(np.array([[vre, cov], [cov, vim]]))
                W[r, r], W[r, r + 1], W[r + 1, r], W[r + 1, r + 1] = blk[0, 0], blk[0, 1], blk[1, 0], blk[1, 1]
            else:
                W[r, r], W[r + 1, r + 1] = 1.0 / vre, 1.0 / vim
            if not (a["status1"][d] == 1 and a["status2"][d] == 1):
                continue                                                 # row kept, coefficients and mean zero (:120, :133)
            self.mean[r], self.mean[r + 1] = zm * c, zm * s
            k = int(a["index"][d]) - 1
            if a["loc"][d] == 0:
                rows += [r, r + 1]; cols += [k, k + n]; vals += [1.0, 1.0]
                continue
            cp, sn = np.cos(phi[k]), np.sin(phi[k])
            if a["loc"][d] == 1:                                         # ReImIijCoefficient (backend/expressions.jl:291-302)
                A = tinv[k] ** 2 * (g[k] + gs[k]); B = -tinv[k] ** 2 * (b[k] + bs[k])
                Cc = -tinv[k] * (g[k] * cp - b[k] * sn); D = tinv[k] * (b[k] * cp + g[k] * sn)
            else:                                                        # ReImIjiCoefficient (:338-349)
                A = -tinv[k] * (g[k] * cp + b[k] * sn); B = tinv[k] * (b[k] * cp - g[k] * sn)
                Cc = g[k] + gs[k]; D = -b[k] - bs[k]
            i, j = int(frm[k]), int(to[k])
            rows += [r, r + 1, r, r + 1, r, r + 1, r, r + 1]             # pmuIndices order (:158-161)
            cols += [i, i + n, j, j + n, i + n, i, j + n, j]
            vals += [A, A, Cc, Cc, B, -B, D, -D]
        self.n, self.m = n, m
        self.coefficient = sp.csc_matrix((vals, (rows, cols)), shape=(m, 2 * n))
        self.precision = W.tocsc()

    def solve(self):
        from scipy.sparse.linalg import splu
        temp = self.coefficient.T @ self.precision
        gain = (temp @ self.coefficient).tocsc()
        x = splu(gain).solve(temp @ self.mean)
        v = x[: self.n] + 1j * x[self.n:]
        self.magnitude, self.angle = np.abs(v), np.angle(v)
        return self.magnitude, self.angle


# ---- residualTest! / chiTest restated (TEST ORACLE; src/stateEstimation/badData.jl) --------------------------------------
def normalized_residuals(H, W, residual, slack_col=None):
    """|r_i| / sqrt(|1 / W_ii - c_i|), c = diag(H G^-1 H'), G = H' W H (badData.jl:133-150, 200-217, rowProjection
    :289-311) with dense linear algebra; slack_col (0-based) = column removed from H with gain[slack, slack] = 1 (:200-203).
    Rows with a zero residual get 0 (they are skipped by the reference's argmax loop)."""
    H = np.array(H, dtype=float)
    if slack_col is not None:
        H[:, slack_col] = 0.0
    G = H.T @ W @ H
    if slack_col is not None:
        G[slack_col, slack_col] = 1.0
    c = np.einsum("ij,ji->i", H, np.linalg.solve(G, H.T))
    d = np.abs(1.0 / np.diag(W) - c)
    out = np.zeros(H.shape[0])
    nz = residual != 0.0
    out[nz] = np.abs(residual[nz]) / np.sqrt(d[nz])
    return out


def gn_normalized_residuals(gn: "OracleGN"):
    """residualTest!(analysis::AcStateEstimation{GaussNewton}) up to the argmax: uses se.jacobian / se.residual as the last
    increment! left them (call gn.increment() at the state to test)."""
    import scipy.sparse as sp
    v = gn.vectors()
    n = gn.sys.n
    H = sp.csc_matrix((v["jacobian"], gn.hrowval - 1, gn.hcolptr - 1), shape=(gn.m, 2 * n)).toarray()
    return normalized_residuals(H, gn.precision_dense(), v["residual"], slack_col=gn.sys.slack - 1)


def pmu_normalized_residuals(p: "OraclePmuWLS"):
    """residualTest!(analysis::PmuStateEstimation) (:119-179) up to the argmax, at the estimate p.solve() produced."""
    x = np.concatenate([p.magnitude * np.cos(p.angle), p.magnitude * np.sin(p.angle)])
    r = p.mean - p.coefficient @ x
    return normalized_residuals(p.coefficient.toarray(), p.precision.toarray(), r)


def chi_threshold(df, confidence=0.95):
    """quantile(Chisq(df), confidence) (:960, :994)."""
    from scipy.stats import chi2
    return float(chi2.ppf(confidence, df))
