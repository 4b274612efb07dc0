"""Newton-Raphson AC power flow: host-side mirror of the reference interface over the C ABI.

Reference surface (paths relative to /root/reference)      here
  newtonRaphson(system)        acPowerFlow.jl:39-87          newtonRaphson(system, batch=1)
  mismatch!(analysis)          acPowerFlow.jl:645-685        mismatch_(analysis)
  solve!(analysis)             acPowerFlow.jl:793-911        solve_(analysis)
  powerFlow!(analysis; ...)    acPowerFlow.jl:1389-1433      powerFlow_(analysis, iteration=20, tolerance=1e-8)
  setInitialPoint!(analysis)   acPowerFlow.jl:1226-1249      setInitialPoint_(analysis[, source])
  updateBranch!(analysis; ...) branch.jl:453-459             updateBranch_(analysis, label, status=...)
  analysis.voltage.{magnitude,angle}, analysis.method.{jacobian,mismatch,increment,pq,pvpq,iteration}

(`!` is spelled with a trailing underscore.)  The only addition is `batch`: B independent scenarios
of the same grid (N-1 outages through setOutage_, Monte-Carlo injections through setInjection_)
advance in lock-step on the device; with batch == 1 every container has the reference's shape.

All numerics run in libjgrid_hip.so (hand-written HIP); this file is plumbing and bookkeeping.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace as NS

import numpy as np

from . import _lib
from .system import (CscMatrix, PowerSystem, acModel_, updateBranch_ as _update_branch_system,
                     updateBus_ as _update_bus_system, updateGenerator_ as _update_generator_system)


def _reim(z):
    out = np.empty(2 * z.size, dtype=np.float64)
    out[0::2], out[1::2] = z.real, z.imag
    return out


def initializeACPowerFlow(system: PowerSystem):
    """Bus-type normalisation, start voltages, slack relocation (acPowerFlow.jl:1312-1358)."""
    bus, gen = system.bus, system.generator
    magnitude = bus.voltage.magnitude.copy()
    angle = bus.voltage.angle.copy()
    typ = bus.layout.type
    for i in range(bus.number):
        has = (i + 1) in bus.supply.generator
        if not has and typ[i] == 2:
            typ[i] = 1
            system.model.revision.type += 1
        if has and typ[i] != 1:
            magnitude[i] = gen.voltage.magnitude[bus.supply.generator[i + 1][0] - 1]
    slack = bus.layout.slack
    if slack not in bus.supply.generator:                      # changeSlackBus! (:1334-1358)
        typ[slack - 1] = 1
        system.model.revision.type += 1
        for i in range(bus.number):
            if typ[i] == 2 and (i + 1) in bus.supply.generator:
                typ[i] = 3
                bus.layout.slack = i + 1
                system.model.revision.type += 1
                system.model.revision.slack += 1
                break
        if typ[bus.layout.slack - 1] == 1:
            raise RuntimeError("No generator buses with an in-service generator found in the power system.")
    return magnitude, angle


class AcPowerFlow:
    """AcPowerFlow{NewtonRaphson{HIP}} (src/definition/analysis.jl:154-164, 252-258)."""

    def __init__(self, system: PowerSystem, batch: int, device: int, max_patch: int):
        self.system = system
        self.batch = int(batch)
        self._device, self._max_patch = int(device), int(max_patch)
        self._injection = None                                           # per-scenario injections set by the caller (kept for a rebuild)
        self._h = None
        self._create()
        self.voltage = NS(magnitude=None, angle=None)
        self.power = NS(injection=None, supply=None, shunt=None, from_=None, to=None, charging=None, series=None, generator=None)
        self.current = NS(injection=None, from_=None, to=None, series=None)
        self._outage_labels = np.zeros(self.batch, dtype=np.int64)     # branch out of service per scenario (0 = none)
        self._branches_on_device = False
        self.status = None

    def _create(self):
        """jg_nr_create from the system's CURRENT nodal matrix: newtonJacobian (maps, pattern), symbolic analysis, upload."""
        L = _lib.lib()
        system, device, max_patch = self.system, self._device, self._max_patch
        ac = system.model.ac
        Y, YT = ac.nodalMatrix, ac.nodalMatrixTranspose
        n = system.bus.number
        self._h = _lib.VP()
        _lib.check(L.jg_nr_create(C.byref(self._h), n, Y.colptr, Y.rowval, _reim(Y.nzval), _reim(YT.nzval),
                                  np.ascontiguousarray(system.bus.layout.type, dtype=np.int8),
                                  system.bus.layout.slack, self.batch, int(max_patch), int(device)))
        dims = np.zeros(8, dtype=np.int64)
        _lib.check(L.jg_nr_dims(self._h, dims))
        self.dims = dict(dimJ=int(dims[0]), nnzJ=int(dims[1]), lu_blocks=int(dims[2]), lu_terms=int(dims[3]),
                         lu_launches=int(dims[4]), solve_launches=int(dims[5]), nnzY=Y.nnz, n=n)
        pq, pvpq, pcount = (np.zeros(n, dtype=np.int64) for _ in range(3))
        jcolptr = np.zeros(self.dims["dimJ"] + 1, dtype=np.int64)
        jrowval = np.zeros(self.dims["nnzJ"], dtype=np.int64)
        _lib.check(L.jg_nr_get_maps(self._h, pq, pvpq, pcount, jcolptr, jrowval))
        keep = getattr(self, "method", None)
        self.method = NS(pq=pq, pvpq=pvpq, pcount=pcount, iteration=0 if keep is None else keep.iteration, _jcolptr=jcolptr, _jrowval=jrowval,
                         signature=NS(topology=system.model.revision.topology, type=system.model.revision.type,
                                      acPattern=system.model.revision.acPattern))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().jg_nr_destroy(self._h)
            self._h = None

    __del__ = close

    # ---- containers with the reference's names --------------------------------------------
    def _shape(self, a):
        return a[0] if self.batch == 1 else a

    def _pull_voltage(self):
        n = self.system.bus.number
        vm = np.zeros((self.batch, n))
        va = np.zeros((self.batch, n))
        _lib.check(_lib.lib().jg_nr_get_voltage(self._h, vm, va))
        self.voltage.magnitude, self.voltage.angle = self._shape(vm), self._shape(va)

    @property
    def mismatch(self):
        m = np.zeros((self.batch, self.dims["dimJ"]))
        _lib.check(_lib.lib().jg_nr_get_mismatch(self._h, m))
        return self._shape(m)

    @property
    def increment(self):
        m = np.zeros((self.batch, self.dims["dimJ"]))
        fn = _lib.lib().jg_nr_fast_get_increment if getattr(self.method, "fast", False) else _lib.lib().jg_nr_get_increment
        _lib.check(fn(self._h, m))
        return self._shape(m)

    @property
    def jacobian(self):
        """analysis.method.jacobian: CSC with the reference's pattern; nzval [nnzJ] (or [batch, nnzJ])."""
        if getattr(self.method, "fast", False):
            raise RuntimeError("a fast Newton-Raphson analysis keeps the factorised B', B'' on the device, not the full Jacobian")
        v = np.zeros((self.batch, self.dims["nnzJ"]))
        _lib.check(_lib.lib().jg_nr_get_jacobian(self._h, v))
        return CscMatrix(self.dims["dimJ"], self.method._jcolptr, self.method._jrowval, self._shape(v))

    @property
    def iterations(self):
        it = np.zeros(self.batch, dtype=np.int32)
        _lib.check(_lib.lib().jg_nr_get_iteration(self._h, it))
        return it

    def snapshot_voltage(self):
        _lib.check(_lib.lib().jg_nr_snapshot_voltage(self._h))

    def restore_voltage(self):
        _lib.check(_lib.lib().jg_nr_restore_voltage(self._h))

    def voltage_device(self, vm_ptr: int, va_ptr: int):
        """Write [batch, n] voltages into caller-owned DEVICE buffers (raw pointers)."""
        _lib.check(_lib.lib().jg_nr_get_voltage_device(self._h, C.c_void_p(vm_ptr), C.c_void_p(va_ptr)))

    def pack_results_device(self, dst_ptr: int):
        """[batch, 2 n + 2] result records (V | theta | iterations | status) into a caller-owned DEVICE buffer."""
        _lib.check(_lib.lib().jg_nr_pack_results_device(self._h, C.c_void_p(dst_ptr)))

    # ---- straggler hand-off (jgrid.h: jg_nr_run_defer ...): used by ContingencyPipeline, not part of the reference's surface
    def run_defer(self, iteration: int = 20, tolerance: float = 1e-8, defer_at: int = 64) -> int:
        """powerFlow! that pauses once at most `defer_at` scenarios are active; returns how many are left (0: done).
        Follow with pool.take_lanes(self, ...) (if any are left) and self.finish()."""
        _check_signature(self)
        left = C.c_int32(0)
        _lib.check(_lib.lib().jg_nr_run_defer(self._h, int(iteration), float(tolerance), int(defer_at), C.byref(left)))
        return int(left.value)

    def finish(self):
        """Ends a paused run: method.iteration / status of every scenario (4 = handed to a pool)."""
        it = np.zeros(self.batch, dtype=np.int32)
        st = np.zeros(self.batch, dtype=np.int32)
        _lib.check(_lib.lib().jg_nr_finish(self._h, it, st))
        self.method.iteration = int(it[0]) if self.batch == 1 else it
        self.status = int(st[0]) if self.batch == 1 else st

    def take_lanes(self, src: "AcPowerFlow", lane0: int) -> np.ndarray:
        """The still-active scenarios of the paused analysis `src` continue in lanes lane0.. of this one; returns their
        scenario numbers in `src` (0-based), in lane order."""
        home = np.zeros(64, dtype=np.int32)
        count = C.c_int32(0)
        _lib.check(_lib.lib().jg_nr_move_lanes(self._h, int(lane0), src._h, home, C.byref(count)))
        return home[:count.value].copy()

    def resume(self, lanes: int, iteration: int = 20, tolerance: float = 1e-8):
        """Runs the scenarios in lanes [0, lanes) to the end; returns (iterations, status) of those lanes."""
        it = np.zeros(max(lanes, 1), dtype=np.int32)
        st = np.zeros(max(lanes, 1), dtype=np.int32)
        _lib.check(_lib.lib().jg_nr_resume(self._h, int(lanes), int(iteration), float(tolerance), it, st))
        return it[:lanes], st[:lanes]

    def pack_rows_device(self, dst_ptr: int, lane0: int, rows):
        """Result records of lanes lane0 .. lane0 + len(rows) - 1 into rows `rows` of a [., 2 n + 2] DEVICE record."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        _lib.check(_lib.lib().jg_nr_pack_rows_device(self._h, C.c_void_p(dst_ptr), int(lane0), int(rows.size), rows))

    def screen_device(self, dst_ptr: int):
        """[batch, 10] screen summaries (screenSummary_; ratings as last set) into a caller-owned DEVICE buffer."""
        L = _lib.lib()
        if not self._branches_on_device:
            _upload_branches(self)
        _lib.check(L.jg_nr_set_outage_labels(self._h, np.ascontiguousarray(self._outage_labels, dtype=np.int64)))
        _lib.check(L.jg_nr_screen_device(self._h, C.c_void_p(dst_ptr)))

    def screen_rows_device(self, dst_ptr: int, lane0: int, rows):
        """Screen summaries (screenSummary_) of lanes lane0 .. lane0 + len(rows) - 1 into rows `rows` of a [., 10] DEVICE record."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        L = _lib.lib()
        if not self._branches_on_device:
            _upload_branches(self)
        _lib.check(L.jg_nr_set_outage_labels(self._h, np.ascontiguousarray(self._outage_labels, dtype=np.int64)))
        _lib.check(L.jg_nr_screen_rows_device(self._h, C.c_void_p(dst_ptr), int(lane0), int(rows.size), rows))

    def time_kernel(self, kernel: int, reps: int = 10) -> float:
        ms = C.c_double(0.0)
        _lib.check(_lib.lib().jg_nr_time_kernel(self._h, int(kernel), int(reps), C.byref(ms)))
        return ms.value


class BaseCase:
    """The base case of a screen (jgrid.h: jg_nr_base_*): ONE factorisation of the Jacobian at the state every scenario starts from, shared by the
    batched analyses attached to it -- their first Newton iteration is then a sweep pair on that factor plus a 4 x 4 correction per scenario instead
    of a batched refactorisation (the compensation method; the reference refactorises per scenario, branch.jl:453-459 + acPowerFlow.jl:890-897).

    `single`: a batch-1 analysis whose current voltages / injections / nodal matrix ARE the base case (typically after powerFlow_)."""

    def __init__(self, single: "AcPowerFlow", top_cap: int = 0):
        if single.batch != 1:
            raise ValueError("BaseCase: pass a single-instance analysis")
        self._h = _lib.VP()
        _lib.check(_lib.lib().jg_nr_base_create(C.byref(self._h), single._h, int(top_cap)))
        self.n = single.system.bus.number
        self.nnz = single.dims["nnzY"]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().jg_nr_base_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def info(self):
        v = np.zeros(8, dtype=np.int64)
        _lib.check(_lib.lib().jg_nr_base_info(self._h, v))
        return dict(top_pivots=int(v[0]), split_level=int(v[1]), forward_launches=int(v[2]), backward_launches=int(v[3]),
                    forward_launches_no_top=int(v[4]), backward_launches_no_top=int(v[5]), create_ms=v[6] / 1000.0, attached=int(v[7]))

    def get(self, which: int, count: int):
        out = np.zeros(int(count))
        _lib.check(_lib.lib().jg_nr_base_get(self._h, int(which), out, out.size))
        return out

    def attach(self, an: "AcPowerFlow"):
        """`an`'s scenarios may start from this base (startFromBase_)."""
        _lib.check(_lib.lib().jg_nr_attach_base(an._h, self._h))
        an._base = self


def startFromBase_(an: AcPowerFlow):
    """Every scenario of `an` starts from the attached base case's state (what setInitialPoint!(analysis, base) is per scenario in the reference's
    loop, acPowerFlow.jl:1271-1295): the next powerFlow_ takes its first iteration on the base's shared factor when it can (jgrid.h)."""
    _lib.check(_lib.lib().jg_nr_start_from_base(an._h))


def setFirstIteration_(an: AcPowerFlow, shared: bool = True):
    """shared=False: the first iteration refactorises like every other (A/B switch)."""
    _lib.check(_lib.lib().jg_nr_set_first_iteration(an._h, 1 if shared else 0))


def firstIterationCounts(an: AcPowerFlow):
    """(runs that started on the shared factor, runs that refactorised)"""
    a, b = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().jg_nr_first_iteration_counts(an._h, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


def _push_voltage(an: AcPowerFlow, vm, va):
    vm = np.ascontiguousarray(vm, dtype=np.float64)
    va = np.ascontiguousarray(va, dtype=np.float64)
    n = an.system.bus.number
    stride = 0 if vm.ndim == 1 else n
    if vm.ndim == 2 and vm.shape[0] != an.batch:
        raise ValueError("voltage batch dimension mismatch")
    _lib.check(_lib.lib().jg_nr_set_voltage(an._h, vm.reshape(-1), va.reshape(-1), stride))
    # the host mirror is what was just sent (no read-back: 82 MB and a host-side transposition for 512 scenarios of a 10 000-bus grid)
    if vm.ndim == 1 and an.batch > 1:
        an.voltage.magnitude, an.voltage.angle = np.broadcast_to(vm, (an.batch, n)).copy(), np.broadcast_to(va, (an.batch, n)).copy()
    else:
        an.voltage.magnitude, an.voltage.angle = an._shape(vm.reshape(an.batch, n).copy()), an._shape(va.reshape(an.batch, n).copy())


def setInjection_(an: AcPowerFlow, active=None, reactive=None):
    """Net injections supply - demand per scenario ([n] broadcast or [batch, n]);
    default = the system's own bus.supply - bus.demand (acPowerFlow.jl:676-680)."""
    bus = an.system.bus
    p = bus.supply.active - bus.demand.active if active is None else active
    q = bus.supply.reactive - bus.demand.reactive if reactive is None else reactive
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    stride = 0 if p.ndim == 1 else bus.number
    _lib.check(_lib.lib().jg_nr_set_injection(an._h, p.reshape(-1), q.reshape(-1), stride))
    an._injection = None if active is None and reactive is None else (p.copy(), q.copy())


def setRefinement_(an: AcPowerFlow, on: bool = True):
    """One step of iterative refinement behind every Newton step (what UMFPACK's solve does for the reference's default LU tag,
    utility.jl:576-586); off by default."""
    _lib.check(_lib.lib().jg_nr_set_refine(an._h, 1 if on else 0))
    an._refine = bool(on)


def newtonRaphson(system: PowerSystem, batch: int = 1, device: int = 0, max_patch: int | None = None, refine: bool = False) -> AcPowerFlow:
    """newtonRaphson(system) (acPowerFlow.jl:39-87). Mutates bus types / slack like the reference."""
    if system.bus.layout.slack == 0:
        raise RuntimeError("The slack bus is missing.")
    if system.model.ac.nodalMatrix is None:
        acModel_(system)                                          # model!(system, ac) :43
    vm, va = initializeACPowerFlow(system)
    if max_patch is None:
        max_patch = 0 if batch == 1 else 4
    an = AcPowerFlow(system, batch, device, max_patch)
    setInjection_(an)
    _push_voltage(an, vm, va)
    if refine:
        setRefinement_(an, True)
    return an


def _fast_model(system: PowerSystem, bx: bool):
    """fastNewtonJacobian + fastNewtonJacobian! + jacobianCoefficient (acPowerFlow.jl:338-474) and the shunt terms
    (:328-334): the two constant matrices in the reference's CSC layout (stored zeros kept) and, per stored Ybus entry,
    the pair (B'[r,c], B''[r,c]) the device factorises as one block matrix (identity padding where a bus is not in a
    reduced matrix)."""
    bus, br, ac = system.bus, system.branch, system.model.ac
    n, typ, slack = bus.number, bus.layout.type, bus.layout.slack
    Y = ac.nodalMatrix
    pq = np.zeros(n, dtype=np.int64)
    pvpq = np.zeros(n, dtype=np.int64)
    pq[typ == 1] = np.arange(1, int(np.sum(typ == 1)) + 1)
    pvpq[typ != 3] = np.arange(1, n)
    col_of = np.repeat(np.arange(n), np.diff(Y.colptr))
    row_of = Y.rowval - 1
    inP = (typ[row_of] != 3) & (col_of != slack - 1)
    inQ = (typ[row_of] == 1) & (typ[col_of] == 1)
    posP = np.cumsum(inP) - 1                                   # Ybus pointer -> pointer in the P / Q matrix (column major, same order)
    posQ = np.cumsum(inQ) - 1
    valP, valQ = np.zeros(int(inP.sum())), np.zeros(int(inQ.sum()))
    par = br.parameter

    def add(mat_in, pos, val, r, c, v):                         # addStored!: the entry exists whenever both buses are in the matrix
        p = Y.position(r + 1, c + 1)
        if mat_in[p]:
            val[pos[p]] += v

    for k in np.flatnonzero(br.layout.status == 1):
        i, j = int(br.layout.from_[k]) - 1, int(br.layout.to[k]) - 1
        bsi, tinv = 0.5 * par.susceptance[k], 1.0 / par.turnsRatio[k]
        sn, cs = np.sin(par.shiftAngle[k]), np.cos(par.shiftAngle[k])
        y = ac.admittance[k]
        if bx:
            bmk, A, B = -1.0 / par.reactance[k], y.real, y.imag
        else:
            bmk, A, B = y.imag, 0.0, -1.0 / par.reactance[k]
        den = cs * cs + sn * sn
        if i != slack - 1 and j != slack - 1:
            add(inP, posP, valP, i, j, (-A * sn - B * cs) / den)
            add(inP, posP, valP, j, i, (A * sn - B * cs) / den)
        if i != slack - 1:
            add(inP, posP, valP, i, i, B / den)
        if j != slack - 1:
            add(inP, posP, valP, j, j, B)
        qA, qB, qC = -bmk * tinv, (bmk + bsi) * tinv ** 2, bmk + bsi
        if typ[i] == 1 and typ[j] == 1:
            add(inQ, posQ, valQ, i, j, qA)
            add(inQ, posQ, valQ, j, i, qA)
        if typ[i] == 1:
            add(inQ, posQ, valQ, i, i, qB)
        if typ[j] == 1:
            add(inQ, posQ, valQ, j, j, qC)
    for i in np.flatnonzero((typ == 1) & (bus.shunt.susceptance != 0)):
        add(inQ, posQ, valQ, int(i), int(i), bus.shunt.susceptance[i])
    cols = np.flatnonzero(np.arange(n) != slack - 1)
    colptrP = np.concatenate([[1], 1 + np.cumsum([inP[Y.colptr[c] - 1:Y.colptr[c + 1] - 1].sum() for c in cols])]).astype(np.int64)
    colsQ = np.flatnonzero(typ == 1)
    colptrQ = np.concatenate([[1], 1 + np.cumsum([inQ[Y.colptr[c] - 1:Y.colptr[c + 1] - 1].sum() for c in colsQ])]).astype(np.int64)
    P = CscMatrix(n - 1, colptrP, pvpq[row_of[inP]], valP)
    Q = CscMatrix(colsQ.size, colptrQ, pq[row_of[inQ]], valQ)
    diag = row_of == col_of
    bp = np.where(inP, 0.0, diag.astype(float))
    bq = np.where(inQ, 0.0, diag.astype(float))
    bp[inP] = valP
    bq[inQ] = valQ
    return P, Q, pq, pvpq, bp, bq


def fastOutagePatch(system: PowerSystem, label: int, bx: bool):
    """What `updateBranch!(analysis::AcPowerFlow{<:FastNewtonRaphson}; label, status = 0)` takes out of the two constant matrices
    (_updateBranch!, branch.jl:477; Pijtheta*, QijV*, acPowerFlow.jl:476-537 -- the terms fastNewtonJacobian! put in, :416-447) as
    (1-based pointers into the stored Ybus pattern in outagePatch's order (i,i), (j,j), (i,j), (j,i); deltas of B'; deltas of B''), zero where an
    entry is not in the reduced matrix (slack row / column of B', PV and slack buses in B'')."""
    k = int(label) - 1
    bus, br, ac = system.bus, system.branch, system.model.ac
    if int(br.layout.status[k]) != 1:                            # already out of service: fastNewtonJacobian! put nothing in for it (acPowerFlow.jl:416-447), as Ybus holds none of it
        return np.zeros(4, dtype=np.int64), np.zeros(4), np.zeros(4)
    typ, slack, par, Y = bus.layout.type, bus.layout.slack, br.parameter, ac.nodalMatrix
    i, j = int(br.layout.from_[k]) - 1, int(br.layout.to[k]) - 1
    bsi, tinv = 0.5 * par.susceptance[k], 1.0 / par.turnsRatio[k]
    sn, cs = np.sin(par.shiftAngle[k]), np.cos(par.shiftAngle[k])
    y = ac.admittance[k]
    if bx:
        bmk, A, B = -1.0 / par.reactance[k], y.real, y.imag
    else:
        bmk, A, B = y.imag, 0.0, -1.0 / par.reactance[k]
    den = cs * cs + sn * sn
    ni, nj = i != slack - 1, j != slack - 1
    qi, qj = typ[i] == 1, typ[j] == 1
    qA, qB, qC = -bmk * tinv, (bmk + bsi) * tinv ** 2, bmk + bsi
    ptr = np.array([Y.position(i + 1, i + 1), Y.position(j + 1, j + 1), Y.position(i + 1, j + 1), Y.position(j + 1, i + 1)], dtype=np.int64) + 1
    dbp = -np.array([B / den if ni else 0.0, B if nj else 0.0, (-A * sn - B * cs) / den if ni and nj else 0.0, (A * sn - B * cs) / den if ni and nj else 0.0])
    dbq = -np.array([qB if qi else 0.0, qC if qj else 0.0, qA if qi and qj else 0.0, qA if qi and qj else 0.0])
    return ptr, dbp, dbq


def _fast_newton_raphson(system: PowerSystem, bx: bool, batch: int, device: int, max_patch: int = 0) -> AcPowerFlow:
    if system.bus.layout.slack == 0:
        raise RuntimeError("The slack bus is missing.")
    if system.model.ac.nodalMatrix is None:
        acModel_(system)
    vm, va = initializeACPowerFlow(system)
    an = AcPowerFlow(system, batch, device, max_patch)
    P, Q, pq, pvpq, bp, bq = _fast_model(system, bx)
    _lib.check(_lib.lib().jg_nr_fast_setup(an._h, np.ascontiguousarray(bp), np.ascontiguousarray(bq)))
    an.method.fast, an.method.bx = True, bool(bx)
    an.method.active = NS(jacobian=P)
    an.method.reactive = NS(jacobian=Q)
    an.method.pq, an.method.pvpq = pq, pvpq                      # fast numbering: pq = 1..npq (acPowerFlow.jl:343-356)
    setInjection_(an)
    _push_voltage(an, vm, va)
    return an


def fastNewtonRaphsonBX(system: PowerSystem, batch: int = 1, device: int = 0, max_patch: int = 0) -> AcPowerFlow:
    """fastNewtonRaphsonBX(system) (acPowerFlow.jl:215-217, 259-336): both constant matrices factorised once on the device.
    max_patch = 4: a batch whose scenarios carry branch outages (setOutages_: per-scenario edits of Ybus AND of B', B'')."""
    return _fast_newton_raphson(system, True, batch, device, max_patch)


def fastNewtonRaphsonXB(system: PowerSystem, batch: int = 1, device: int = 0, max_patch: int = 0) -> AcPowerFlow:
    """fastNewtonRaphsonXB(system) (acPowerFlow.jl:255-257)."""
    return _fast_newton_raphson(system, False, batch, device, max_patch)


def mismatch_(an: AcPowerFlow):
    """mismatch!(analysis) -> (max|f_P|, max|f_Q|); arrays of length batch when batch > 1."""
    if an.system.model.revision.acPattern != an.method.signature.acPattern:
        _rebuild(an)                                              # the device copy of Ybus has the old pattern
    p = np.zeros(an.batch)
    q = np.zeros(an.batch)
    fn = _lib.lib().jg_nr_fast_mismatch if getattr(an.method, "fast", False) else _lib.lib().jg_nr_mismatch
    _lib.check(fn(an._h, p, q))
    return (float(p[0]), float(q[0])) if an.batch == 1 else (p, q)


def _rebuild(an: AcPowerFlow):
    """The Ybus pattern changed under a live analysis (addBranch! between buses that had no entry, dropZeros!): the reference
    rebuilds the Jacobian and factorises from scratch on the next solve! (acPowerFlow.jl:806-811, 890-897).  Here the handle
    is rebuilt -- new maps, new symbolic analysis -- and the state of the analysis moves over: voltages, injections, outages."""
    vm, va = np.zeros((an.batch, an.system.bus.number)), np.zeros((an.batch, an.system.bus.number))
    _lib.check(_lib.lib().jg_nr_get_voltage(an._h, vm, va))
    old = an.method
    fast = getattr(old, "fast", False)
    _lib.lib().jg_nr_destroy(an._h)
    an._h = None
    an._create()
    if fast:                                                     # the fast method's own containers (B', B'', its numbering) stay
        for k, v in vars(old).items():
            if not hasattr(an.method, k):
                setattr(an.method, k, v)
    inj = an._injection
    if getattr(an, "_refine", False):
        setRefinement_(an, True)
    if inj is not None:
        setInjection_(an, *inj)
    else:
        setInjection_(an)
    _push_voltage(an, vm, va)
    an._branches_on_device = False
    if np.any(an._outage_labels):
        setOutages_(an, [int(x) for x in an._outage_labels])
    if fast:
        _refresh_fast(an)


def _check_signature(an: AcPowerFlow):
    rev, sig = an.system.model.revision, an.method.signature
    if rev.topology != sig.topology or rev.type != sig.type:     # acPowerFlow.jl:802-804
        raise RuntimeError("The power flow model cannot be reused due to required bus type conversion.")
    if rev.acPattern != sig.acPattern:                            # acPowerFlow.jl:806-811
        _rebuild(an)


def solve_(an: AcPowerFlow):
    """solve!(analysis): Jacobian fill + refactorization + solve + state update on the device."""
    _check_signature(an)
    fn = _lib.lib().jg_nr_fast_solve if getattr(an.method, "fast", False) else _lib.lib().jg_nr_solve
    _lib.check(fn(an._h))
    an.method.iteration += 1
    an._pull_voltage()


def powerFlow_(an: AcPowerFlow, iteration: int = 20, tolerance: float = 1e-8, fetch: bool = True):
    """powerFlow!(analysis; iteration, tolerance). Sets analysis.method.iteration (array if batch > 1)
    and analysis.status (0 converged, 1 iteration limit, 3 numeric failure).  fetch=False leaves the
    voltages in HBM (batched loops read them later through analysis._pull_voltage / voltage_device)."""
    _check_signature(an)
    it = np.zeros(an.batch, dtype=np.int32)
    st = np.zeros(an.batch, dtype=np.int32)
    fn = _lib.lib().jg_nr_fast_run if getattr(an.method, "fast", False) else _lib.lib().jg_nr_run
    _lib.check(fn(an._h, int(iteration), float(tolerance), it, st))
    an.method.iteration = int(it[0]) if an.batch == 1 else it
    an.status = int(st[0]) if an.batch == 1 else st
    if fetch:
        an._pull_voltage()


def setInitialPoint_(an: AcPowerFlow, source=None):
    """setInitialPoint!(analysis) / setInitialPoint!(target, source) (acPowerFlow.jl:1226-1295)."""
    bus, gen = an.system.bus, an.system.generator
    if source is not None:
        _push_voltage(an, source.voltage.magnitude, source.voltage.angle)
        return
    vm = bus.voltage.magnitude.copy()
    for i in range(bus.number):
        if (i + 1) in bus.supply.generator and bus.layout.type[i] != 1:
            vm[i] = gen.voltage.magnitude[bus.supply.generator[i + 1][0] - 1]
    _push_voltage(an, vm, bus.voltage.angle.copy())


def _upload_ybus(an: AcPowerFlow):
    if an.system.model.revision.acPattern != an.method.signature.acPattern:
        return                                                    # new pattern: the next solve rebuilds the device model (_rebuild)
    ac = an.system.model.ac
    _lib.check(_lib.lib().jg_nr_set_ybus(an._h, _reim(ac.nodalMatrix.nzval), _reim(ac.nodalMatrixTranspose.nzval)))


def updateBranch_(an: AcPowerFlow, label: int, status: int | None = None, **parameters):
    """updateBranch!(analysis; label, status, resistance, reactance, conductance, susceptance, turnsRatio, shiftAngle)
    (branch.jl:453-463): edits the system (stored zeros keep the pattern, so the symbolic analysis is reused exactly like
    the reference's `lu!` path) and syncs the device copy of the nodal matrix."""
    _update_branch_system(an.system, label, status=status, **parameters)
    _upload_ybus(an)
    an._branches_on_device = False                                       # power!/current! re-upload the branch table
    an.method.signature.topology = an.system.model.revision.topology     # syncTopology! branch.jl:461-463
    _refresh_fast(an)


def addBranch_(an: AcPowerFlow, **kwargs) -> int:
    """addBranch!(analysis; from, to, ...) (branch.jl:182-188): the branch joins the system; if its buses had no Ybus entry yet the
    pattern revision moves and the next mismatch! / solve! / powerFlow! rebuilds the device model (acPowerFlow.jl:806-811)."""
    rev, sig = an.system.model.revision, an.method.signature
    if rev.type != sig.type:                                      # errorTypeConversion (branch.jl:190-192)
        raise RuntimeError("The power flow model cannot be reused due to required bus type conversion.")
    from .system import addBranch_ as _add
    label = _add(an.system, **kwargs)
    if rev.acPattern == sig.acPattern:
        _upload_ybus(an)                                          # same pattern: new values only
        _refresh_fast(an)
    an._branches_on_device = False
    sig.topology = rev.topology                                   # syncTopology!
    return label


def dropZeros_(an: AcPowerFlow) -> None:
    """dropZeros!(system, system.model.ac) under a live analysis (model.jl:342-352)."""
    from .system import dropZeros_ as _drop
    _drop(an.system)


def _refresh_fast(an: AcPowerFlow):
    """fast Newton-Raphson keeps two CONSTANT matrices: after a change of the grid they are rebuilt and refactorised
    (the reference patches them entry by entry and calls lu! again, acPowerFlow.jl:476-537)."""
    if not getattr(an.method, "fast", False):
        return
    if an.system.model.revision.acPattern != an.method.signature.acPattern:
        return                                                    # _rebuild refreshes them on the new pattern
    P, Q, _, _, bp, bq = _fast_model(an.system, an.method.bx)
    _lib.check(_lib.lib().jg_nr_fast_setup(an._h, np.ascontiguousarray(bp), np.ascontiguousarray(bq)))
    an.method.active.jacobian, an.method.reactive.jacobian = P, Q
    if np.any(an._outage_labels):                                # new shared matrices drop the scenarios' edits (jg_nr_fast_setup): put them back
        setOutages_(an, [int(x) for x in an._outage_labels])


def updateBus_(an: AcPowerFlow, label: int, **kwargs):
    """updateBus!(analysis; label, active, reactive, conductance, susceptance, magnitude, angle) (bus.jl:343-420):
    demand -> injections, shunt -> nodal matrix diagonal; magnitude / angle are start values (setInitialPoint_)."""
    _update_bus_system(an.system, label, **kwargs)
    if "conductance" in kwargs or "susceptance" in kwargs:
        _upload_ybus(an)
        _refresh_fast(an)
    if "active" in kwargs or "reactive" in kwargs:
        setInjection_(an)


def updateGenerator_(an: AcPowerFlow, label: int, **kwargs):
    """updateGenerator!(analysis; label, status, active, reactive, magnitude) (generator.jl:382-408): supply ->
    injections; a generator bus that would lose its last unit needs a new analysis (errorTypeConversion)."""
    sysm = an.system
    k = int(label) - 1
    if 0 <= k < sysm.generator.number and kwargs.get("status") == 0 and sysm.generator.layout.status[k] == 1:
        i = int(sysm.generator.layout.bus[k])
        if sysm.bus.layout.type[i - 1] in (2, 3) and sysm.bus.supply.generator.get(i, []) == [k + 1]:
            raise RuntimeError("The power flow model cannot be reused due to required bus type conversion.")
    _update_generator_system(sysm, label, **kwargs)
    setInjection_(an)


def outagePatch(system: PowerSystem, label: int):
    """The 4 Ybus edits of `updateBranch!(...; label, status = 0)` (branch.jl:344-350, model.jl:93-101)
    as (1-based pointers into nodalMatrix.nzval, complex deltas)."""
    k = int(label) - 1
    ac, Y = system.model.ac, system.model.ac.nodalMatrix
    i, j = int(system.branch.layout.from_[k]), int(system.branch.layout.to[k])
    ptr = np.array([Y.position(i, i), Y.position(j, j), Y.position(i, j), Y.position(j, i)], dtype=np.int64) + 1
    dy = -np.array([ac.nodalFromFrom[k], ac.nodalToTo[k], ac.nodalFromTo[k], ac.nodalToFrom[k]], dtype=np.complex128)
    return ptr, dy


def outagePatchTable(system: PowerSystem):
    """outagePatch of EVERY branch at once -- (pointers [nb, 4], deltas [nb, 4]) -- kept on the system until its AC model moves: a screen uploads the
    outages of a whole device batch per job, and 2 048 pattern look-ups in Python per batch would cost more than the batch's power flows."""
    rev = system.model.revision
    key = (rev.acModel, rev.acPattern, rev.topology, system.branch.number)
    tab = getattr(system, "_outage_table", None)
    if tab is not None and tab[0] == key:
        return tab[1], tab[2]
    ac, Y = system.model.ac, system.model.ac.nodalMatrix
    i = np.asarray(system.branch.layout.from_, dtype=np.int64)
    j = np.asarray(system.branch.layout.to, dtype=np.int64)
    n = system.bus.number
    rows = np.asarray(Y.rowval, dtype=np.int64)
    cols = np.repeat(np.arange(1, n + 1, dtype=np.int64), np.diff(np.asarray(Y.colptr, dtype=np.int64)))
    keys = cols * (n + 1) + rows                                      # CSC order: ascending in (col, row)

    def pos(r, c):
        q = c * (n + 1) + r
        p = np.searchsorted(keys, q)
        if np.any(p >= keys.size) or np.any(keys[np.minimum(p, keys.size - 1)] != q):
            raise KeyError("The sparse matrix pattern does not contain the requested entry.")
        return p

    ptr = np.stack([pos(i, i), pos(j, j), pos(i, j), pos(j, i)], axis=1).astype(np.int64) + 1
    dy = -np.stack([ac.nodalFromFrom, ac.nodalToTo, ac.nodalFromTo, ac.nodalToFrom], axis=1).astype(np.complex128)
    system._outage_table = (key, ptr, dy)
    return ptr, dy


def setOutage_(an: AcPowerFlow, scenario: int, label: int | None):
    """Scenario `scenario` of a batched analysis = base grid with branch `label` out of service
    (None restores the base grid)."""
    if getattr(an.method, "fast", False):                        # Ybus AND the two constant matrices: one path (setOutages_)
        return setOutages_(an, [int(label) if label else 0], int(scenario))
    if label is None:
        _lib.check(_lib.lib().jg_nr_patch_ybus(an._h, int(scenario), 0, np.zeros(1, dtype=np.int64), np.zeros(2)))
        an._outage_labels[int(scenario)] = 0
        return
    ptr, dy = outagePatch(an.system, label)
    _lib.check(_lib.lib().jg_nr_patch_ybus(an._h, int(scenario), 4, ptr, _reim(dy)))
    an._outage_labels[int(scenario)] = int(label)


def setOutages_(an: AcPowerFlow, labels, scenario0: int = 0):
    """setOutage_ for consecutive scenarios in ONE upload: scenario scenario0 + s = base grid with branch labels[s] out
    of service (0 / None = base grid)."""
    lab = np.array([int(x) if x else 0 for x in labels], dtype=np.int64)
    labels = [int(x) for x in lab]
    tptr, tdy = outagePatchTable(an.system)
    if lab.size and (lab.min() < 0 or lab.max() > an.system.branch.number):
        raise IndexError("setOutages_: branch label out of range")
    on = lab > 0
    ptr = np.where(on[:, None], tptr[np.maximum(lab, 1) - 1], 0)
    dy = np.where(on[:, None], tdy[np.maximum(lab, 1) - 1], 0.0)
    _lib.check(_lib.lib().jg_nr_patch_ybus_batch(an._h, int(scenario0), len(labels), 4, np.ascontiguousarray(ptr.reshape(-1)), _reim(np.ascontiguousarray(dy.reshape(-1)))))
    an._outage_labels[scenario0:scenario0 + len(labels)] = lab
    if getattr(an.method, "fast", False):
        # fast Newton-Raphson: the outage also leaves the two constant matrices (branch.jl:477); every scenario keeps the shared B', B'' plus its own
        # (at most) 4 + 4 edits, and the batch is factorised ONCE -- the iterations stay forward / backward sweeps
        fptr = np.zeros((len(labels), 4), dtype=np.int64)
        dbp, dbq = np.zeros((len(labels), 4)), np.zeros((len(labels), 4))
        for s, lab in enumerate(labels):
            if lab:
                fptr[s], dbp[s], dbq[s] = fastOutagePatch(an.system, int(lab), an.method.bx)
        _lib.check(_lib.lib().jg_nr_fast_patch_batch(an._h, int(scenario0), len(labels), 4, fptr.reshape(-1), dbp.reshape(-1), dbq.reshape(-1)))


# ---- post-processing (N2): power!(analysis) / current!(analysis) for every scenario of the batch ------------------
def _upload_branches(an: AcPowerFlow):
    system = an.system
    ac, par, lay = system.model.ac, system.branch.parameter, system.branch.layout
    nb = system.branch.number
    tij = (1.0 / par.turnsRatio) * np.exp(-1j * par.shiftAngle)                     # acAnalysis.jl:846-851
    tab = np.zeros((nb, 16))
    for c, z in enumerate((ac.nodalFromFrom, ac.nodalFromTo, ac.nodalToFrom, ac.nodalToTo, ac.admittance, tij)):
        tab[:, 2 * c], tab[:, 2 * c + 1] = np.real(z), np.imag(z)
    tab[:, 12], tab[:, 13], tab[:, 14] = par.conductance, par.susceptance, 1.0 / par.turnsRatio
    L = _lib.lib()
    _lib.check(L.jg_nr_set_branches(an._h, nb, np.ascontiguousarray(lay.from_, dtype=np.int64), np.ascontiguousarray(lay.to, dtype=np.int64),
                                    np.ascontiguousarray(lay.status, dtype=np.int8), np.ascontiguousarray(tab.reshape(-1))))
    an._branches_on_device = True
    #  a rating belongs to the branch table it was given for: the library drops it when the branch count changes, so it is pushed again
    # with every new table -- or forgotten here too when it no longer has one value per branch (addBranch_ on an unchanged pattern)
    r = getattr(an, "_screen_rating", None)
    if r is not None:
        if r.shape == (nb,):
            _lib.check(L.jg_nr_set_screen(an._h, r.ctypes.data))
        else:
            an._screen_rating = None


def _pairs(an, fn, rows, *slots):
    """Call a C-ABI post-processing entry with [batch][rows][2] host buffers for the requested slots."""
    bufs = [np.zeros((an.batch, rows, 2)) if want else None for want in slots]
    _lib.check(fn(an._h, *[b.ctypes.data if b is not None else None for b in bufs]))
    return bufs


def _ns2(an, buf, names=("active", "reactive")):
    return NS(**{names[0]: an._shape(buf[:, :, 0]), names[1]: an._shape(buf[:, :, 1])})


def power_(an: AcPowerFlow):
    """power!(analysis::AcPowerFlow) (src/postprocessing/acAnalysis.jl:30-169) at the current state of EVERY scenario:
    the Ybus row walk (injections) and the branch formulas run on the device, the O(n) bus / generator bookkeeping
    (shunt :884-889, supply :53-61, generators :84-166) on the host.  Arrays are [batch, ...] (1-D for batch 1)."""
    system, bus, gen = an.system, an.system.bus, an.system.generator
    L = _lib.lib()
    if not an._branches_on_device:
        _upload_branches(an)
    _lib.check(L.jg_nr_set_outage_labels(an._h, np.ascontiguousarray(an._outage_labels, dtype=np.int64)))
    inj = np.zeros((an.batch, bus.number, 2))
    _lib.check(L.jg_nr_bus_injection(an._h, inj.reshape(-1)))
    fr, to, se, ch = _pairs(an, lambda h, a, b, c, d: L.jg_nr_branch_quantities(h, a, b, c, d, None, None, None),
                            system.branch.number, True, True, True, True)
    an._pull_voltage()
    vm = np.atleast_2d(an.voltage.magnitude)
    P, Q = inj[:, :, 0], inj[:, :, 1]
    v2 = vm * vm
    shunt = np.stack([v2 * bus.shunt.conductance[None, :], -v2 * bus.shunt.susceptance[None, :]], axis=2)   # V^2 conj(g + jb)
    typ, slack = bus.layout.type, bus.layout.slack - 1
    sup_p = np.broadcast_to(bus.supply.active[None, :], P.shape).copy()
    sup_q = np.where(typ[None, :] != 1, Q + bus.demand.reactive[None, :], bus.supply.reactive[None, :])        # :55-59
    sup_p[:, slack] = P[:, slack] + bus.demand.active[slack]                                                 # :61
    # generators (:84-166)
    ng = gen.number
    gp, gq = np.zeros((an.batch, ng)), np.zeros((an.batch, ng))
    base_mva = system.base.power * 1e-6
    qmin_all, qmax_all = np.asarray(gen.capability.minReactive, dtype=float), np.asarray(gen.capability.maxReactive, dtype=float)
    for i in range(ng):
        if gen.layout.status[i] != 1:
            continue
        ib = int(gen.layout.bus[i]) - 1
        idx = [g - 1 for g in bus.supply.generator[ib + 1]]
        Pi, Qi = P[:, ib], Q[:, ib]
        if len(idx) == 1:
            gp[:, i] = gen.output.active[i]
            gq[:, i] = Qi + bus.demand.reactive[ib]
            if ib == slack:
                gp[:, i] = Pi + bus.demand.active[ib]
            continue
        qgen = Qi + bus.demand.reactive[ib]
        qmins = sum(qmin_all[j] for j in idx if not np.isinf(qmin_all[j]))
        qmaxs = sum(qmax_all[j] for j in idx if not np.isinf(qmax_all[j]))
        big = np.abs(qgen) + abs(qmins) + abs(qmaxs)
        qmin_new = np.full(an.batch, qmin_all[i])
        qmax_new = np.full(an.batch, qmax_all[i])
        qmin_inf = np.zeros(an.batch)
        qmax_inf = np.zeros(an.batch)
        for j in idx:
            if np.isinf(qmin_all[j]):
                qm = -big if qmin_all[j] != np.inf else big
                if j == i:
                    qmin_new = qm
                qmin_inf = qmin_inf + qm
            if np.isinf(qmax_all[j]):
                qm = big if qmax_all[j] != -np.inf else -big
                if j == i:
                    qmax_new = qm
                qmax_inf = qmax_inf + qm
        qmin_sum, qmax_sum = qmins + qmin_inf, qmaxs + qmax_inf
        prop = base_mva * np.abs(qmin_sum - qmax_sum) > 10 * np.finfo(float).eps
        with np.errstate(divide="ignore", invalid="ignore"):
            gq[:, i] = np.where(prop, qmin_new + ((qgen - qmin_sum) / (qmax_sum - qmin_sum)) * (qmax_new - qmin_new),
                                qmin_new + (qgen - qmin_sum) / len(idx))
        if ib == slack and idx[0] == i:
            gp[:, i] = Pi + bus.demand.active[ib] - sum(gen.output.active[j] for j in idx[1:])
        else:
            gp[:, i] = gen.output.active[i]
    pw = an.power
    pw.injection, pw.shunt = _ns2(an, inj), _ns2(an, shunt)
    pw.supply = NS(active=an._shape(sup_p), reactive=an._shape(sup_q))
    pw.from_, pw.to, pw.series, pw.charging = _ns2(an, fr), _ns2(an, to), _ns2(an, se), _ns2(an, ch)
    pw.generator = NS(active=an._shape(gp), reactive=an._shape(gq))


def current_(an: AcPowerFlow):
    """current!(analysis) (acAnalysis.jl:672-704): injection, from, to and series currents (magnitude, angle)."""
    system = an.system
    L = _lib.lib()
    if not an._branches_on_device:
        _upload_branches(an)
    _lib.check(L.jg_nr_set_outage_labels(an._h, np.ascontiguousarray(an._outage_labels, dtype=np.int64)))
    inj = np.zeros((an.batch, system.bus.number, 2))
    _lib.check(L.jg_nr_bus_injection(an._h, inj.reshape(-1)))
    fi, ti, si = _pairs(an, lambda h, a, b, c: L.jg_nr_branch_quantities(h, None, None, None, None, a, b, c),
                        system.branch.number, True, True, True)
    an._pull_voltage()
    vm, va = np.atleast_2d(an.voltage.magnitude), np.atleast_2d(an.voltage.angle)
    # I_i = conj(S_i / V_i): |I_i| = |S_i| / V_i, arg I_i = theta_i - arg S_i   (Ii, :867-882)
    S = inj[:, :, 0] + 1j * inj[:, :, 1]
    I = np.conj(S / (vm * np.exp(1j * va)))
    cur = an.current
    cur.injection = NS(magnitude=an._shape(np.abs(I)), angle=an._shape(np.angle(I)))
    names = ("magnitude", "angle")
    cur.from_, cur.to, cur.series = _ns2(an, fi, names), _ns2(an, ti, names), _ns2(an, si, names)


def screenSummary_(an: AcPowerFlow, rating=None, device_record: int | None = None):
    """Contingency screen summary on the device (include/jgrid.h: jg_nr_screen): per scenario the worst branch loading against `rating`
    (pu of apparent power per branch, 0 / None = no limit) and its branch, the largest apparent power at a branch end and its branch, the lowest and the
    highest voltage magnitude and their buses, iterations and status -- what a user of the reference reads off power!(analysis) after every powerFlow! of
    the outage loop (branch.jl:453-459), reduced where the states are.  device_record: DEVICE pointer of a [batch, 10] float64 buffer to fill instead
    (the operand of the one gather of a sharded screen); else a namespace of arrays ([batch]; scalars for batch 1) comes back."""
    L = _lib.lib()
    if not an._branches_on_device:
        _upload_branches(an)
    _lib.check(L.jg_nr_set_outage_labels(an._h, np.ascontiguousarray(an._outage_labels, dtype=np.int64)))
    if rating is not None:
        r = np.ascontiguousarray(np.asarray(rating, dtype=np.float64))
        if r.shape != (an.system.branch.number,):
            raise ValueError("rating: one value per branch")
        _lib.check(L.jg_nr_set_screen(an._h, r.ctypes.data))
        an._screen_rating = r
    elif getattr(an, "_screen_rating", None) is not None:
        _lib.check(L.jg_nr_set_screen(an._h, None))
        an._screen_rating = None
    if device_record is not None:
        _lib.check(L.jg_nr_screen_device(an._h, _lib.VP(int(device_record))))
        return None
    rec = np.zeros((an.batch, 10))
    _lib.check(L.jg_nr_screen(an._h, rec))
    one = an.batch == 1
    f = (lambda x: x[0]) if one else (lambda x: x)
    return NS(loading=f(rec[:, 0]), loadingBranch=f(rec[:, 1].astype(np.int64)), flow=f(rec[:, 2]), flowBranch=f(rec[:, 3].astype(np.int64)),
              minMagnitude=f(rec[:, 4]), minBus=f(rec[:, 5].astype(np.int64)), maxMagnitude=f(rec[:, 6]), maxBus=f(rec[:, 7].astype(np.int64)),
              iteration=f(rec[:, 8].astype(np.int32)), status=f(rec[:, 9].astype(np.int32)))


def reactiveLimit_(an: AcPowerFlow):
    """reactiveLimit!(analysis) (src/powerFlow/acPowerFlow.jl:1081-1155), single-scenario analyses: generator outputs
    from power! on the device, then the reference's bookkeeping on the PowerSystem container -- a generator whose
    reactive output violates its limits pins Q at the limit and turns its bus into a demand bus; a converted slack
    hands over to the first generator bus.  Mutates `an.system` like the reference; returns the violate vector.
    The caller then builds a new analysis (`newtonRaphson(system)`) and solves again."""
    if an.batch != 1:
        raise ValueError("reactiveLimit_ works on a single-scenario analysis")
    system, bus, gen = an.system, an.system.bus, an.system.generator
    power_(an)
    gp, gq = an.power.generator.active, an.power.generator.reactive
    violate = np.zeros(gen.number, dtype=np.int64)
    bus.supply.active[:] = 0.0
    bus.supply.reactive[:] = 0.0
    out_q = np.zeros(gen.number)
    for k in range(gen.number):                                            # :1093-1103
        if gen.layout.status[k] == 1:
            i = int(gen.layout.bus[k]) - 1
            gen.output.active[k] = gp[k]
            bus.supply.active[i] += gp[k]
            bus.supply.reactive[i] += gq[k]
            out_q[k] = gq[k]
    qmin, qmax = gen.capability.minReactive, gen.capability.maxReactive
    typ = bus.layout.type
    for i in range(gen.number):                                            # :1105-1148
        if gen.layout.status[i] == 0 or not (qmin[i] < qmax[i]):
            continue
        j = int(gen.layout.bus[i]) - 1
        lo, hi = out_q[i] < qmin[i], out_q[i] > qmax[i]
        if typ[j] != 1 and (lo or hi):
            if lo:
                violate[i], new_q = -1, qmin[i]
            if hi:
                violate[i], new_q = 1, qmax[i]
            typ[j] = 1
            system.model.revision.type += 1
            bus.supply.reactive[j] -= out_q[i]
            gen.output.reactive[i] = new_q
            bus.supply.reactive[j] += new_q
            if j == bus.layout.slack - 1:
                for k in range(bus.number):
                    if typ[k] == 2:
                        bus.layout.slack = k + 1
                        system.model.revision.slack += 1
                        typ[k] = 3
                        system.model.revision.type += 1
                        break
    if typ[bus.layout.slack - 1] != 3:
        raise RuntimeError("The slack bus is not defined.")                # errorSlackDefinition (:1151-1153)
    return violate


def adjustAngle_(an: AcPowerFlow, slack: int):
    """adjustAngle!(analysis; slack) (acPowerFlow.jl:1196-1206): shift every angle so that bus `slack` (1-based index)
    carries the angle the PowerSystem container holds for it."""
    shift = an.system.bus.voltage.angle[int(slack) - 1] - np.atleast_2d(an.voltage.angle)[:, int(slack) - 1]
    an.voltage.angle = an._shape(np.atleast_2d(an.voltage.angle) + shift[:, None])
