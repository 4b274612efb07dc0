"""Host-side power-system container and AC model (the INPUT PRODUCER of the hot path).

Mirrors, by name and meaning, the reference types the NR / Gauss-Newton path reads:

  PowerSystem{bus, branch, generator, model}      src/definition/system.jl:213-271
  powerSystem(file)                               src/powerSystem/load.jl:36-67 (npz fixtures and
                                                  MATPOWER .m and the reference's HDF5 case files here: hdf5.py)
  acModel!(system)      -> acModel_(system)       src/powerSystem/model.jl:23-78
  updateBranch!(system; label, status)            src/powerSystem/branch.jl:313-431 (status toggles
  -> updateBranch_(system, label, status=...)     only: :344-350 outage, :381-386 re-close)

All containers keep the reference's conventions: per-unit, radians, 1-based Int64 indices, CSC with
sorted rows and duplicates summed in insertion order (src/backend/sparse.jl:43-95), explicit stored
zeros for out-of-service branches (model.jl:70-71).

This module is product host code (numpy). It never imports anything from oracle/.
"""
from __future__ import annotations

import os
import re
from types import SimpleNamespace as NS

import numpy as np

_CASE_DIRS = [
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "data"),
    os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cases"),
]


class PowerSystem:
    """Container with the reference's field names (bus/branch/generator/model)."""

    def __init__(self, tables: dict):
        t = {k: np.array(v) for k, v in tables.items()}
        n = t["bus_type"].size
        nb = t["br_from"].size
        ng = t["gen_bus"].size
        label = t["bus_label"] if "bus_label" in t else np.arange(1, n + 1, dtype=np.int64)
        self.base = NS(power=float(np.asarray(t.get("base_power", 1e8)).reshape(-1)[0]))
        self.bus = NS(
            number=n,
            label={int(l): i + 1 for i, l in enumerate(label)},
            layout=NS(type=t["bus_type"].astype(np.int8), slack=0),
            demand=NS(active=t["bus_pd"].astype(np.float64), reactive=t["bus_qd"].astype(np.float64)),
            supply=NS(active=np.zeros(n), reactive=np.zeros(n), generator={}),
            shunt=NS(conductance=t["bus_gs"].astype(np.float64), susceptance=t["bus_bs"].astype(np.float64)),
            voltage=NS(magnitude=t["bus_vm"].astype(np.float64), angle=t["bus_va"].astype(np.float64)),
        )
        # slack: the HDF5 loader stops at the FIRST type-3 bus (load.jl:155-160), the MATPOWER loader keeps the LAST one
        # (load.jl:429-431); none -> bus 1 (load.jl:434-437)
        s = np.flatnonzero(self.bus.layout.type == 3)
        pick = -1 if t.get("slack_rule") is not None and str(np.asarray(t["slack_rule"]).reshape(-1)[0]) == "last" else 0
        self.bus.layout.slack = int(s[pick]) + 1 if s.size else 1
        self.branch = NS(
            number=nb,
            layout=NS(from_=t["br_from"].astype(np.int64), to=t["br_to"].astype(np.int64),
                      status=t["br_status"].astype(np.int8)),
            parameter=NS(resistance=t["br_r"].astype(np.float64), reactance=t["br_x"].astype(np.float64),
                         conductance=t["br_g"].astype(np.float64), susceptance=t["br_b"].astype(np.float64),
                         turnsRatio=t["br_tap"].astype(np.float64), shiftAngle=t["br_shift"].astype(np.float64)),
        )
        self.generator = NS(
            number=ng,
            layout=NS(bus=t["gen_bus"].astype(np.int64), status=t["gen_status"].astype(np.int8)),
            output=NS(active=t["gen_pg"].astype(np.float64), reactive=t["gen_qg"].astype(np.float64)),
            voltage=NS(magnitude=t["gen_vg"].astype(np.float64)),
            capability=NS(minReactive=t.get("gen_qmin", np.zeros(ng)), maxReactive=t.get("gen_qmax", np.zeros(ng))),
        )
        # bus.supply = sum of in-service generator outputs, generator lists in index order
        # (load.jl:271-277, 589-596)
        for k in range(ng):
            if self.generator.layout.status[k] == 1:
                i = int(self.generator.layout.bus[k])
                self.bus.supply.generator.setdefault(i, []).append(k + 1)
                self.bus.supply.active[i - 1] += self.generator.output.active[k]
                self.bus.supply.reactive[i - 1] += self.generator.output.reactive[k]
        self.model = NS(
            ac=NS(nodalMatrix=None, nodalMatrixTranspose=None, nodalFromFrom=None, nodalFromTo=None,
                  nodalToTo=None, nodalToFrom=None, admittance=None),
            revision=NS(topology=0, type=0, slack=0, acModel=0, acPattern=0),
        )

    def copy(self) -> "PowerSystem":
        import copy as _copy
        return _copy.deepcopy(self)


class CscMatrix:
    """Minimal CSC holder with the reference's field names (1-based colptr/rowval)."""

    def __init__(self, n, colptr, rowval, nzval):
        self.n = int(n)
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        self.rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        self.nzval = np.ascontiguousarray(nzval)

    @property
    def nnz(self):
        return int(self.rowval.size)

    def position(self, row, col):
        """storedPosition (src/backend/sparse.jl:104-121), 0-based pointer into nzval."""
        lo, hi = self.colptr[col - 1] - 1, self.colptr[col] - 1
        p = lo + int(np.searchsorted(self.rowval[lo:hi], row))
        if p >= hi or self.rowval[p] != row:
            raise KeyError("The sparse matrix pattern does not contain the requested entry.")
        return p

    def has(self, row, col):
        lo, hi = self.colptr[col - 1] - 1, self.colptr[col] - 1
        p = lo + int(np.searchsorted(self.rowval[lo:hi], row))
        return p < hi and self.rowval[p] == row

    def insert(self, row, col):
        """A[row, col] += x on an absent entry of a SparseMatrixCSC: a structural entry (value 0) at its sorted place."""
        lo, hi = self.colptr[col - 1] - 1, self.colptr[col] - 1
        p = lo + int(np.searchsorted(self.rowval[lo:hi], row))
        self.rowval = np.insert(self.rowval, p, row)
        self.nzval = np.insert(self.nzval, p, 0)
        self.colptr = self.colptr.copy()
        self.colptr[col:] += 1

    def toscipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.nzval, self.rowval - 1, self.colptr - 1), shape=(self.n, self.n))


def _matpower_tables(path: str) -> dict:
    """MATPOWER .m reader with the unit conventions of src/powerSystem/load.jl:341-619."""
    text = open(path).read()

    def matrix(name):
        m = re.search(r"mpc\." + name + r"\s*=\s*\[(.*?)\];", text, re.S)
        rows = []
        for line in m.group(1).splitlines():
            line = line.split("%")[0].strip().rstrip(";").strip()
            if line:
                rows.append([float(x) for x in line.replace(",", " ").split()])
        return rows

    base = float(re.search(r"mpc\.baseMVA\s*=\s*([^;]+);", text).group(1))
    binv, d2r = 1.0 / base, np.pi / 180
    bus, gen, br = matrix("bus"), matrix("gen"), matrix("branch")
    lab = {int(r[0]): k + 1 for k, r in enumerate(bus)}
    col = lambda rows, j: np.array([r[j] for r in rows], dtype=np.float64)
    tap = col(br, 8)
    tap[tap == 0.0] = 1.0
    return dict(
        base_power=base * 1e6, bus_label=np.array([int(r[0]) for r in bus]),
        bus_type=col(bus, 1).astype(np.int8), bus_pd=col(bus, 2) * binv, bus_qd=col(bus, 3) * binv,
        bus_gs=col(bus, 4) * binv, bus_bs=col(bus, 5) * binv, bus_vm=col(bus, 7), bus_va=col(bus, 8) * d2r,
        br_from=np.array([lab[int(r[0])] for r in br]), br_to=np.array([lab[int(r[1])] for r in br]),
        br_status=col(br, 10).astype(np.int8), br_r=col(br, 2), br_x=col(br, 3), br_g=np.zeros(len(br)),
        br_b=col(br, 4), br_tap=tap, br_shift=col(br, 9) * d2r,
        gen_bus=np.array([lab[int(r[0])] for r in gen]), gen_status=col(gen, 7).astype(np.int8),
        gen_pg=col(gen, 1) * binv, gen_qg=col(gen, 2) * binv, gen_vg=col(gen, 5),
        gen_qmax=col(gen, 3) * binv, gen_qmin=col(gen, 4) * binv,
        slack_rule=np.array(["last"]),
    )


def powerSystem(source) -> PowerSystem:
    """powerSystem("case14") / powerSystem("x.npz") / powerSystem("x.m") / powerSystem("x.h5") / powerSystem(dict);
    "case9241synth" = the seeded PEGASE-shaped grid of synthetic.py (stands in for case9241pegase)."""
    if isinstance(source, dict):
        return PowerSystem(source)
    path = str(source)
    if path == "case9241synth":
        from .synthetic import case9241synth
        return PowerSystem(case9241synth())
    if path.endswith(".m"):
        return PowerSystem(_matpower_tables(path))
    if path.endswith(".h5"):                         # the reference's own case format (load.jl:36-67, 141-289)
        from .hdf5 import case_tables
        return PowerSystem(case_tables(path))
    if not os.path.exists(path):
        for d in _CASE_DIRS:
            for cand in (os.path.join(d, path), os.path.join(d, path + ".npz")):
                if os.path.exists(cand):
                    path = cand
                    break
    if not os.path.exists(path):
        raise FileNotFoundError(f"power system case {source!r} not found")
    with np.load(path) as z:
        return PowerSystem({k: z[k] for k in z.files})


def _branch_two_port(r, x, g, b, tap, shift):
    """Unified branch model Y-parameters (model.jl:54-64), explicit real arithmetic (Smith 1/z)."""
    r, x = np.asarray(r, dtype=np.float64), np.asarray(x, dtype=np.float64)
    big = np.abs(r) >= np.abs(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        q1 = np.where(big, x / r, r / x)
        d = np.where(big, r + x * q1, r * q1 + x)
        yre = np.where(big, 1.0 / d, q1 / d)
        yim = np.where(big, -q1 / d, -1.0 / d)
    y = yre + 1j * yim
    tinv = 1.0 / tap
    tr = tinv * np.cos(-shift) + 1j * (tinv * np.sin(-shift))   # turnsRatioInv * cis(-shift)
    ytt = y + 0.5 * (g + 1j * b)
    yff = (tinv * tinv) * ytt
    yft = -np.conj(tr) * y
    ytf = -tr * y
    return y, yff, yft, ytt, ytf


def acModel_(system: PowerSystem) -> None:
    """acModel!(system): Ybus, its transpose copy and per-branch two-port parameters.

    Reference: src/powerSystem/model.jl:23-78 with the CSC builder of src/backend/sparse.jl:2-101.
    Entries of a column are inserted as [diagonal, then branch stamps in branch order], stably
    sorted by row, duplicates summed in that order; out-of-service branches insert zeros.
    """
    n, nb = system.bus.number, system.branch.number
    lay, par = system.branch.layout, system.branch.parameter
    f = lay.from_ - 1
    t = lay.to - 1
    on = lay.status == 1
    y, yff, yft, ytt, ytf = _branch_two_port(par.resistance, par.reactance, par.conductance,
                                             par.susceptance, par.turnsRatio, par.shiftAngle)
    zero = np.zeros(nb, dtype=np.complex128)
    y, yff, yft, ytt, ytf = (np.where(on, a, zero) for a in (y, yff, yft, ytt, ytf))

    # diagonal: shunt, then += yff[from], += ytt[to] sequentially in branch order (model.jl:66-67)
    diag = system.bus.shunt.conductance + 1j * system.bus.shunt.susceptance
    diag = diag.astype(np.complex128)
    idx = np.empty(2 * nb, dtype=np.int64)
    val = np.empty(2 * nb, dtype=np.complex128)
    idx[0::2], idx[1::2] = f, t
    val[0::2], val[1::2] = yff, ytt
    sel = np.repeat(on, 2)
    np.add.at(diag, idx[sel], val[sel])

    # off-diagonals in insertion order: (from,to)->column `to`, then (to,from)->column `from`
    rows = np.empty(2 * nb, dtype=np.int64)
    cols = np.empty(2 * nb, dtype=np.int64)
    vals = np.empty(2 * nb, dtype=np.complex128)
    rows[0::2], cols[0::2], vals[0::2] = f, t, yft
    rows[1::2], cols[1::2], vals[1::2] = t, f, ytf
    rows = np.concatenate([np.arange(n), rows])
    cols = np.concatenate([np.arange(n), cols])
    vals = np.concatenate([diag, vals])
    order = np.lexsort((np.arange(rows.size), rows, cols))          # stable by (col, row, insertion)
    rows, cols, vals = rows[order], cols[order], vals[order]
    first = np.ones(rows.size, dtype=bool)
    first[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    group = np.cumsum(first) - 1
    nzval = np.zeros(int(group[-1]) + 1, dtype=np.complex128)
    np.add.at(nzval, group, vals)                                    # sequential -> insertion order
    rowval = rows[first] + 1
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, cols[first] + 1, 1)
    colptr = np.cumsum(colptr) + 1

    ac = system.model.ac
    ac.nodalMatrix = CscMatrix(n, colptr, rowval, nzval)
    # transpose copy (model.jl:75): pattern is symmetric, so the transpose shares colptr/rowval
    # and value p of the transpose is Y[col, row] of pointer p.
    ac.nodalMatrixTranspose = CscMatrix(n, colptr.copy(), rowval.copy(), _transpose_values(ac.nodalMatrix))
    ac.admittance, ac.nodalFromFrom, ac.nodalFromTo, ac.nodalToTo, ac.nodalToFrom = y, yff, yft, ytt, ytf


def _transpose_perm(A: CscMatrix) -> np.ndarray:
    """perm[p] = pointer of entry (col, row) for pointer p = (row, col); needs a symmetric pattern."""
    n = A.n
    col_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.colptr))
    row_of = A.rowval - 1
    key = row_of * n + col_of            # key of the transposed entry in column-major order
    perm = np.argsort(key, kind="stable")
    # entry p=(r,c) has column-major key c*n+r; transposed entry (c,r) has key r*n+c
    own = col_of * n + row_of
    pos = np.searchsorted(own, key)      # own is sorted (CSC order)
    if not np.array_equal(own[pos], key):
        raise ValueError("Ybus pattern is not structurally symmetric")
    del perm
    return pos


def _transpose_values(A: CscMatrix) -> np.ndarray:
    return A.nzval[_transpose_perm(A)]


def _ac_nodal_update(system: PowerSystem, k: int, sign: float) -> None:
    """acNodalUpdate! (model.jl:81-110): add sign*two-port stamps of branch k (0-based) in place."""
    ac = system.model.ac
    i, j = int(system.branch.layout.from_[k]), int(system.branch.layout.to[k])
    Y, YT = ac.nodalMatrix, ac.nodalMatrixTranspose
    ff, tt = sign * ac.nodalFromFrom[k], sign * ac.nodalToTo[k]
    ft, tf = sign * ac.nodalFromTo[k], sign * ac.nodalToFrom[k]
    Y.nzval[Y.position(i, i)] += ff
    Y.nzval[Y.position(j, j)] += tt
    YT.nzval[YT.position(i, i)] += ff
    YT.nzval[YT.position(j, j)] += tt
    Y.nzval[Y.position(i, j)] += ft
    Y.nzval[Y.position(j, i)] += tf
    YT.nzval[YT.position(j, i)] += ft
    YT.nzval[YT.position(i, j)] += tf
    system.model.revision.acModel += 1


def _ensure_pair(system: PowerSystem, i: int, j: int) -> None:
    """Structural entries (i,j) and (j,i) of Ybus and of its transpose copy; a new one changes the pattern (model.jl:103-107)."""
    ac = system.model.ac
    grew = False
    for A in (ac.nodalMatrix, ac.nodalMatrixTranspose):
        for r, c in ((i, j), (j, i)):
            if not A.has(r, c):
                A.insert(r, c)
                grew = True
    if grew:
        system.model.revision.acPattern += 1


def addBranch_(system: PowerSystem, from_: int, to: int, resistance: float = 0.0, reactance: float = 0.0, conductance: float = 0.0,
               susceptance: float = 0.0, turnsRatio: float = 1.0, shiftAngle: float = 0.0, status: int = 1) -> int:
    """addBranch!(system; from, to, ...) (branch.jl:79-167), per-unit / radian inputs, bus LABELS for from / to.
    Returns the new branch label (its 1-based index).  On a built AC model an in-service branch enters the nodal matrix like
    acNodalUpdate! does it (model.jl:81-110): a pair of buses that had no entry yet grows the pattern (acPattern revision)."""
    if from_ == to:
        raise ValueError("Invalid value for from or to keywords.")
    if resistance == 0.0 and reactance == 0.0:
        raise ValueError("At least one of resistance or reactance is required.")
    if status not in (0, 1):
        raise ValueError("The status must be 0 or 1.")
    try:
        i, j = system.bus.label[int(from_)], system.bus.label[int(to)]
    except KeyError as e:
        raise KeyError(f"The bus label {e.args[0]} that has been specified does not exist.") from None
    br = system.branch
    lay, par = br.layout, br.parameter
    lay.from_ = np.append(lay.from_, i)
    lay.to = np.append(lay.to, j)
    lay.status = np.append(lay.status, np.int8(status))
    for name, v in (("resistance", resistance), ("reactance", reactance), ("conductance", conductance), ("susceptance", susceptance),
                    ("turnsRatio", turnsRatio), ("shiftAngle", shiftAngle)):
        setattr(par, name, np.append(getattr(par, name), float(v)))
    br.number += 1
    k = br.number - 1
    ac = system.model.ac
    if ac.nodalMatrix is not None:
        for name in ("admittance", "nodalFromFrom", "nodalFromTo", "nodalToTo", "nodalToFrom"):          # acPushZeros!
            setattr(ac, name, np.append(getattr(ac, name), 0j))
        if status == 1:
            y, yff, yft, ytt, ytf = _branch_two_port(*(np.array([getattr(par, nm)[k]]) for nm in
                                                       ("resistance", "reactance", "conductance", "susceptance", "turnsRatio", "shiftAngle")))
            ac.admittance[k], ac.nodalFromFrom[k], ac.nodalFromTo[k], ac.nodalToTo[k], ac.nodalToFrom[k] = y[0], yff[0], yft[0], ytt[0], ytf[0]
            _ensure_pair(system, i, j)
            _ac_nodal_update(system, k, +1.0)
    system.model.revision.topology += 1                                                                     # topologyChanged!
    return br.number


def dropZeros_(system: PowerSystem) -> None:
    """dropZeros!(system, system.model.ac) (model.jl:342-352): stored zeros of Ybus (out-of-service branches) leave the pattern;
    if any did, the pattern revision moves and every analysis rebuilds its Jacobian on its next solve."""
    ac = system.model.ac
    before = ac.nodalMatrix.nnz
    for A in (ac.nodalMatrix, ac.nodalMatrixTranspose):
        keep = A.nzval != 0
        col_of = np.repeat(np.arange(A.n), np.diff(A.colptr))
        colptr = np.zeros(A.n + 1, dtype=np.int64)
        np.add.at(colptr, col_of[keep] + 1, 1)
        A.colptr, A.rowval, A.nzval = np.cumsum(colptr) + 1, A.rowval[keep], A.nzval[keep]
    if ac.nodalMatrix.nnz != before:
        system.model.revision.acPattern += 1


def updateBranch_(system: PowerSystem, label: int, status: int | None = None, resistance=None, reactance=None,
                  conductance=None, susceptance=None, turnsRatio=None, shiftAngle=None) -> None:
    """updateBranch!(system; label, status, resistance, reactance, conductance, susceptance, turnsRatio, shiftAngle)
    (branch.jl:313-431).

    A branch in service first leaves the nodal matrix with its OLD two-port stamps (:344-350; the pattern keeps stored
    zeros), the parameters change, and a branch that ends up in service re-enters with the new stamps (:381-386)."""
    k = int(label) - 1
    if not 0 <= k < system.branch.number:
        raise KeyError(f"The branch label {label} that has been specified does not exist.")
    old = int(system.branch.layout.status[k])
    new = old if status is None else int(status)
    if new not in (0, 1):
        raise ValueError("The status 0 or 1 is required.")
    par = system.branch.parameter
    edits = dict(resistance=resistance, reactance=reactance, conductance=conductance, susceptance=susceptance,
                 turnsRatio=turnsRatio, shiftAngle=shiftAngle)
    changed = any(v is not None for v in edits.values())
    ac = system.model.ac
    has = ac.nodalMatrix is not None
    if has and old == 1 and (new == 0 or changed):
        _ac_nodal_update(system, k, -1.0)
        for a in (ac.nodalFromFrom, ac.nodalFromTo, ac.nodalToTo, ac.nodalToFrom, ac.admittance):
            a[k] = 0.0
    for name, v in edits.items():
        if v is not None:
            getattr(par, name)[k] = float(v)
    if has and new == 1 and (old == 0 or changed):
        y, yff, yft, ytt, ytf = _branch_two_port(par.resistance[k:k + 1], par.reactance[k:k + 1],
                                                 par.conductance[k:k + 1], par.susceptance[k:k + 1],
                                                 par.turnsRatio[k:k + 1], par.shiftAngle[k:k + 1])
        ac.admittance[k], ac.nodalFromFrom[k], ac.nodalFromTo[k] = y[0], yff[0], yft[0]
        ac.nodalToTo[k], ac.nodalToFrom[k] = ytt[0], ytf[0]
        _ensure_pair(system, int(system.branch.layout.from_[k]), int(system.branch.layout.to[k]))      # after dropZeros!: the entry comes back
        _ac_nodal_update(system, k, +1.0)
    system.branch.layout.status[k] = new
    if new != old:
        system.model.revision.topology += 1


def updateBus_(system: PowerSystem, label: int, type=None, active=None, reactive=None, conductance=None, susceptance=None,
               magnitude=None, angle=None) -> None:
    """updateBus!(system; label, type, active, reactive, conductance, susceptance, magnitude, angle) (bus.jl:170-255): bus
    type and slack (:179-206; a live analysis goes stale, acPowerFlow.jl:802-804 -- build a new one, as in the reference),
    demand, shunt (the nodal matrix diagonal follows) and the initial voltage."""
    if int(label) not in system.bus.label:
        raise KeyError(f"The bus label {label} that has been specified does not exist.")
    i = system.bus.label[int(label)] - 1
    bus = system.bus
    if type is not None:
        if int(type) not in (1, 2, 3):
            raise ValueError("bus type must be 1 (demand), 2 (generator) or 3 (slack)")
        type_old, slack_old = int(bus.layout.type[i]), int(bus.layout.slack)
        if int(type) in (1, 2):
            if bus.layout.slack == i + 1:
                bus.layout.slack = 0
            bus.layout.type[i] = int(type)
        else:
            if bus.layout.slack not in (0, i + 1):
                raise RuntimeError(f"To set bus with label {label} as the slack bus, reassign the current slack bus to either a "
                                   "generator or demand bus.")
            bus.layout.type[i] = 3
            bus.layout.slack = i + 1
        if int(bus.layout.type[i]) != type_old:
            system.model.revision.type += 1
        if int(bus.layout.slack) != slack_old:
            system.model.revision.slack += 1
    if active is not None:
        bus.demand.active[i] = float(active)
    if reactive is not None:
        bus.demand.reactive[i] = float(reactive)
    if conductance is not None or susceptance is not None:
        ac = system.model.ac
        g = bus.shunt.conductance[i] if conductance is None else float(conductance)
        b = bus.shunt.susceptance[i] if susceptance is None else float(susceptance)
        if ac.nodalMatrix is not None:
            d = (g - bus.shunt.conductance[i]) + 1j * (b - bus.shunt.susceptance[i])
            ac.nodalMatrix.nzval[ac.nodalMatrix.position(i + 1, i + 1)] += d
            ac.nodalMatrixTranspose.nzval[ac.nodalMatrixTranspose.position(i + 1, i + 1)] += d
            system.model.revision.acModel += 1
        bus.shunt.conductance[i], bus.shunt.susceptance[i] = g, b
    if magnitude is not None:
        bus.voltage.magnitude[i] = float(magnitude)
    if angle is not None:
        bus.voltage.angle[i] = float(angle)


def updateGenerator_(system: PowerSystem, label: int, status: int | None = None, active=None, reactive=None,
                     magnitude=None) -> None:
    """updateGenerator!(system; label, status, active, reactive, magnitude) (generator.jl:262-360): the bus supply and
    the per-bus generator lists follow the status and the outputs."""
    k = int(label) - 1
    gen, bus = system.generator, system.bus
    if not 0 <= k < gen.number:
        raise KeyError(f"The generator label {label} that has been specified does not exist.")
    old = int(gen.layout.status[k])
    new = old if status is None else int(status)
    if new not in (0, 1):
        raise ValueError("The status 0 or 1 is required.")
    i = int(gen.layout.bus[k])
    output = active is not None or reactive is not None
    if old == 1:
        if new == 0 or output:
            bus.supply.active[i - 1] -= gen.output.active[k]
            bus.supply.reactive[i - 1] -= gen.output.reactive[k]
        if new == 0:
            bus.supply.generator[i].remove(k + 1)
            if not bus.supply.generator[i]:
                del bus.supply.generator[i]
    if active is not None:
        gen.output.active[k] = float(active)
    if reactive is not None:
        gen.output.reactive[k] = float(reactive)
    if new == 1:
        if old == 0 or output:
            bus.supply.active[i - 1] += gen.output.active[k]
            bus.supply.reactive[i - 1] += gen.output.reactive[k]
        if old == 0:
            lst = bus.supply.generator.setdefault(i, [])
            lst.append(k + 1)
            lst.sort()
    gen.layout.status[k] = new
    if magnitude is not None:
        gen.voltage.magnitude[k] = float(magnitude)
