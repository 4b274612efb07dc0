"""Gauss-Newton WLS state estimation: host-side mirror of the reference interface over the C ABI.

Reference surface (paths relative to /root/reference)                 here
  gaussNewton(monitoring)      stateEstimation/acStateEstimation.jl:43-75     gaussNewton(monitoring, batch=1)
  increment!(analysis)         acStateEstimation.jl:878-904                   increment_(analysis)
  solve!(analysis)             acStateEstimation.jl:1035-1047                 solve_(analysis)
  stateEstimation!(analysis)   acStateEstimation.jl:1286-1329                 stateEstimation_(analysis, iteration=40, tolerance=1e-8)
  analysis.voltage, analysis.method.{jacobian,precision,mean,residual,increment,type,index,range,objective,iteration}

The host does the value bookkeeping of acWLS (:77-259): type = status * code, se.mean (squared
for types 4/5, rectangular z cos / z sin for 16-21), se.precision (1/variance, 4 z^2 sigma^2 for
squared currents, variancePmu / covariancePmu blocks for rectangular PMUs; equations.jl:576-677,
measurement/utility.jl:115-129).  Everything numeric per iteration runs in libjgrid_hip.so.
`batch` > 1 = Monte-Carlo noise realisations of one measurement configuration (setNoise_).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace as NS

import numpy as np

from . import _lib
from .measurement import Measurement
from .powerflow import _reim
from .system import CscMatrix, acModel_


def _wls_layout(mon: Measurement):
    """Row layout of acWLS: per row (code, status source, index) + per device raw-reading slots."""
    sysm = mon.system
    code, index, dev_row = [], [], []       # dev_row: first row of each device, in concatenation order
    fam_range = [1]
    devs = []                               # (family, device number)
    for i, k in enumerate(mon.voltmeter.layout.index):
        devs.append(("v", i)); dev_row.append(len(code)); code.append(1); index.append(k)
    fam_range.append(len(code) + 1)
    a = mon.ammeter
    for i, k in enumerate(a.layout.index):
        devs.append(("a", i)); dev_row.append(len(code))
        code.append((4 if a.layout.from_[i] else 5) if a.layout.square[i] else (2 if a.layout.from_[i] else 3)); index.append(k)
    fam_range.append(len(code) + 1)
    for fam, meter, cb, cf, ct in (("w", mon.wattmeter, 6, 7, 8), ("r", mon.varmeter, 9, 10, 11)):
        for i, k in enumerate(meter.layout.index):
            devs.append((fam, i)); dev_row.append(len(code))
            code.append(cb if meter.layout.bus[i] else (cf if meter.layout.from_[i] else ct)); index.append(k)
        fam_range.append(len(code) + 1)
    p = mon.pmu
    corr_rows = []
    for i, k in enumerate(p.layout.index):
        devs.append(("p", i)); dev_row.append(len(code))
        frm = p.layout.from_[i]
        if p.layout.polar[i]:
            if p.layout.bus[i]:
                code += [12, 13]
            else:
                code += [((4 if frm else 5) if p.layout.square[i] else (2 if frm else 3)), (14 if frm else 15)]
        else:
            if p.layout.correlated[i]:
                corr_rows.append(len(code) + 1)
            code += [16, 17] if p.layout.bus[i] else ([18, 20] if frm else [19, 21])
        index += [k, k]
    fam_range.append(len(code) + 1)
    del sysm
    return (np.array(code, dtype=np.int8), np.array(index, dtype=np.int64), np.array(fam_range, dtype=np.int64),
            np.array(corr_rows, dtype=np.int64), devs, np.array(dev_row, dtype=np.int64))


def _readings(mon: Measurement):
    """Raw meter readings in device order: z1/var1/st1 (magnitude or the single quantity), z2/var2/st2 (PMU angle)."""
    z1, v1, s1, z2, v2, s2 = [], [], [], [], [], []
    for g in (mon.voltmeter.magnitude, mon.ammeter.magnitude, mon.wattmeter.active, mon.varmeter.reactive):
        z1 += g.mean; v1 += g.variance; s1 += g.status
        z2 += [0.0] * len(g.mean); v2 += [1.0] * len(g.mean); s2 += [1] * len(g.mean)
    z1 += mon.pmu.magnitude.mean; v1 += mon.pmu.magnitude.variance; s1 += mon.pmu.magnitude.status
    z2 += mon.pmu.angle.mean; v2 += mon.pmu.angle.variance; s2 += mon.pmu.angle.status
    f8 = lambda x: np.array(x, dtype=np.float64)
    return f8(z1), f8(v1), np.array(s1, dtype=np.int8), f8(z2), f8(v2), np.array(s2, dtype=np.int8)


def _rectangular_pmu(zz, var, za, vara, correlated):
    """Rectangular form of one polar PMU reading: (z cos, z sin), the diagonal of its precision block and the
    off-diagonal (None when uncorrelated): variancePmu / covariancePmu + precision! (equations.jl:576-666)."""
    s, c = np.sin(za), np.cos(za)
    vre = var * c ** 2 + vara * (zz * s) ** 2
    vim = var * s ** 2 + vara * (zz * c) ** 2
    if not correlated:
        return zz * c, zz * s, 1.0 / vre, 1.0 / vim, None
    l1i = 1.0 / np.sqrt(vre)
    l2 = s * c * (var - vara * zz ** 2) * l1i
    l3i2 = 1.0 / (vim - l2 ** 2)
    off = (-l2 * l1i) * l3i2
    return zz * c, zz * s, (l1i - l2 * off) * l1i, l3i2, off


def _wls_values(mon: Measurement, devs, dev_row, m, z1, v1, s1, z2, v2, s2):
    """se.mean, diag(se.precision), correlated off-diagonals and row status for readings of shape [..., ndev]."""
    shape = z1.shape[:-1]
    mean = np.zeros(shape + (m,))
    wdiag = np.ones(shape + (m,))
    status = np.zeros(m, dtype=np.int8)
    woff = []
    a, p = mon.ammeter, mon.pmu
    # the legacy meters (one row each) as whole arrays -- 90 000 of the 92 000 devices of config 4; the formulas of the per-device branch below, element by element
    fam_of = np.array([f for f, _ in devs])
    legacy = np.nonzero(fam_of != "p")[0]
    if legacy.size:
        rows = np.asarray(dev_row, dtype=np.int64)[legacy]
        sq = np.zeros(legacy.size, dtype=bool)
        amm = fam_of[legacy] == "a"
        if amm.any():
            sq[amm] = np.asarray(a.layout.square, dtype=bool)[np.array([devs[d][1] for d in legacy[amm]], dtype=np.int64)]
        zz, var, st = z1[..., legacy], v1[legacy], s1[legacy]
        status[rows] = st
        mean[..., rows] = st * np.where(sq, zz ** 2, zz)                       # :138, :149, :163, :177
        with np.errstate(divide="ignore"):
            wdiag[..., rows] = 1.0 / np.where(sq, 4.0 * zz ** 2 * var, var)    # varianceSquare
    for d, (fam, i) in enumerate(devs):
        if fam != "p":
            continue
        r = int(dev_row[d])
        zz, var, st = z1[..., d], v1[d], int(s1[d])
        za, vara, sta = z2[..., d], v2[d], int(s2[d])
        if p.layout.polar[i]:
            sq = p.layout.square[i] and not p.layout.bus[i]
            status[r], status[r + 1] = st, sta
            mean[..., r] = st * (zz ** 2 if sq else zz)                        # :195-199
            wdiag[..., r] = 1.0 / (4.0 * zz ** 2 * var if sq else var)
            mean[..., r + 1] = sta * za
            wdiag[..., r + 1] = 1.0 / vara
        else:
            stt = st * sta
            status[r] = status[r + 1] = stt
            re, im, wre, wim, off = _rectangular_pmu(zz, var, za, vara, p.layout.correlated[i])
            mean[..., r] = stt * re                                            # :216-217
            mean[..., r + 1] = stt * im
            wdiag[..., r], wdiag[..., r + 1] = wre, wim
            if off is not None:
                woff.append(off)
    woff = np.stack(woff, axis=-1) if woff else np.zeros(shape + (0,))
    if not (np.all(np.isfinite(wdiag)) and np.all(wdiag > 0) and np.all(np.isfinite(woff))):
        raise ValueError("The variance of a measurement is zero or negative (errorVariance): "
                         "check zero-magnitude squared-current / rectangular PMU readings.")
    return mean, wdiag, woff, status


class AcStateEstimation:
    """AcStateEstimation{GaussNewton{HIP}} (src/definition/analysis.jl:532-545, 643-650)."""

    def __init__(self, monitoring: Measurement, batch: int, device: int):
        L = _lib.lib()
        self.monitoring = monitoring
        self.system = sysm = monitoring.system
        self.batch = int(batch)
        if self._needs_slack and sysm.bus.layout.slack == 0:
            raise RuntimeError("The slack bus is missing.")
        if sysm.model.ac.nodalMatrix is None:
            acModel_(sysm)                                                     # model!(system, ac) :88
        ac = sysm.model.ac
        Y, YT = ac.nodalMatrix, ac.nodalMatrixTranspose
        n, nb = sysm.bus.number, sysm.branch.number
        code, index, rng, corr, devs, dev_row = self._layout(monitoring)
        self._devs, self._dev_row = devs, dev_row
        m = code.size
        if m == 0:
            raise ValueError("the measurement set is empty")
        z = self._raw(monitoring)
        self._z = z
        mean, wdiag, woff, status = self._values(monitoring, devs, dev_row, m, *z)
        par = sysm.branch.parameter
        bp = np.ascontiguousarray(np.stack([ac.admittance.real, ac.admittance.imag, par.conductance, par.susceptance,
                                            par.turnsRatio, par.shiftAngle], axis=1), dtype=np.float64)
        self._h = _lib.VP()
        _lib.check(L.jg_gn_create(C.byref(self._h), n, Y.colptr, Y.rowval, _reim(Y.nzval), _reim(YT.nzval), nb,
                                  np.ascontiguousarray(sysm.branch.layout.from_, dtype=np.int64),
                                  np.ascontiguousarray(sysm.branch.layout.to, dtype=np.int64), bp.reshape(-1),
                                  sysm.bus.layout.slack if self._needs_slack else 0, m, code, status, index, corr.size,
                                  corr if corr.size else np.zeros(1, dtype=np.int64), self.batch, int(device)))
        dims = np.zeros(8, dtype=np.int64)
        _lib.check(L.jg_gn_dims(self._h, dims))
        self.dims = dict(m=int(dims[0]), nnzH=int(dims[1]), gain_blocks=int(dims[2]), lu_blocks=int(dims[3]), lu_terms=int(dims[4]),
                         factor_launches=int(dims[5]), backward_launches=int(dims[6]), slots=int(dims[7]), n=n, nnzY=Y.nnz)
        typ = np.zeros(m, dtype=np.int8)
        hcolptr = np.zeros(2 * n + 1, dtype=np.int64)
        hrowval = np.zeros(self.dims["nnzH"], dtype=np.int64)
        _lib.check(L.jg_gn_get_maps(self._h, typ, hcolptr, hrowval))
        self.method = NS(type=typ, index=index, range=rng, mean=mean, iteration=0, _hcolptr=hcolptr, _hrowval=hrowval,
                         _wdiag=wdiag, _woff=woff, _corr=corr, _code=code)
        self._upload_measurement(mean, wdiag, woff)
        self.voltage = NS(magnitude=None, angle=None)
        self._start(sysm)
        self.status = None

    _needs_slack = True
    _layout = staticmethod(_wls_layout)
    _raw = staticmethod(_readings)
    _values = staticmethod(_wls_values)

    def _start(self, sysm):
        self.setVoltage(sysm.bus.voltage.magnitude, sysm.bus.voltage.angle)    # acStateEstimation.jl:52-55

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().jg_gn_destroy(self._h)
            self._h = None

    __del__ = close

    def _shape(self, a):
        return a[0] if self.batch == 1 else a

    def _upload_measurement(self, mean, wdiag, woff):
        m = self.dims["m"]
        mean, wdiag, woff = (np.ascontiguousarray(x, dtype=np.float64) for x in (mean, wdiag, woff))
        sm = 0 if mean.ndim == 1 else m
        sc = 0 if woff.ndim == 1 else woff.shape[-1]
        if woff.size == 0:
            woff = np.zeros(1)
        _lib.check(_lib.lib().jg_gn_set_measurement(self._h, mean.reshape(-1), wdiag.reshape(-1), woff.reshape(-1), sm, sc))
        self.method.mean, self.method._wdiag, self.method._woff = mean, wdiag, woff
        self._mirror_stale = False

    def _sync_mirrors(self):
        """ drawNoise_ rewrites se.mean / se.precision on the DEVICE only; whoever reads the host mirrors afterwards -- residualTest_'s bookkeeping,
        objective, precision, chiTest -- pulls the lanes' own realisations first (their z-dependent weights with them: squared currents, rectangular PMUs)."""
        if not getattr(self, "_mirror_stale", False):
            return
        mean, wd, wo = measurementDevice(self)
        one = self.batch == 1
        self.method.mean, self.method._wdiag, self.method._woff = (mean[0], wd[0], wo[0]) if one else (mean, wd, wo)
        self._mirror_stale = False

    def setVoltage(self, magnitude, angle):
        vm = np.ascontiguousarray(magnitude, dtype=np.float64)
        va = np.ascontiguousarray(angle, dtype=np.float64)
        stride = 0 if vm.ndim == 1 else self.system.bus.number
        _lib.check(_lib.lib().jg_gn_set_voltage(self._h, vm.reshape(-1), va.reshape(-1), stride))
        self._pull_voltage()

    def snapshot_voltage(self):
        """Keep the current state in HBM as the start point of later restarts (restore_voltage)."""
        _lib.check(_lib.lib().jg_gn_snapshot_voltage(self._h))

    def restore_voltage(self):
        _lib.check(_lib.lib().jg_gn_restore_voltage(self._h))

    def _pull_voltage(self):
        n = self.system.bus.number
        vm, va = np.zeros((self.batch, n)), np.zeros((self.batch, n))
        _lib.check(_lib.lib().jg_gn_get_voltage(self._h, vm, va))
        self.voltage.magnitude, self.voltage.angle = self._shape(vm), self._shape(va)

    @property
    def jacobian(self):
        v = np.zeros((self.batch, self.dims["nnzH"]))
        _lib.check(_lib.lib().jg_gn_get_jacobian(self._h, v))
        return CscMatrix(2 * self.system.bus.number, self.method._hcolptr, self.method._hrowval, self._shape(v))

    @property
    def residual(self):
        r = np.zeros((self.batch, self.dims["m"]))
        _lib.check(_lib.lib().jg_gn_get_residual(self._h, r))
        return self._shape(r)

    @property
    def increment(self):
        r = np.zeros((self.batch, 2 * self.system.bus.number))
        _lib.check(_lib.lib().jg_gn_get_increment(self._h, r))
        return self._shape(r)

    @property
    def precision(self):
        """se.precision as a dense [m, m] matrix (first scenario): diagonal + 2x2 PMU blocks."""
        self._sync_mirrors()
        wd = np.atleast_2d(self.method._wdiag)[0]
        W = np.diag(wd)
        wo = np.atleast_2d(self.method._woff)[0] if self.method._corr.size else []
        for q, r in enumerate(self.method._corr):
            W[r - 1, r] = W[r, r - 1] = wo[q]
        return W

    @property
    def objective(self):
        """se.objective = r' W r (equations.jl:689-698), evaluated on demand from the device residuals."""
        self._sync_mirrors()
        res = np.atleast_2d(self.residual)
        wd = np.broadcast_to(np.atleast_2d(self.method._wdiag), res.shape)
        obj = np.sum(res * res * wd, axis=1)
        if self.method._corr.size:
            wo = np.broadcast_to(np.atleast_2d(self.method._woff), (res.shape[0], self.method._corr.size))
            r0 = self.method._corr - 1
            obj = obj + 2.0 * np.sum(res[:, r0] * res[:, r0 + 1] * wo, axis=1)
        return float(obj[0]) if self.batch == 1 else obj

    def objectiveDevice(self):
        """se.objective per scenario, reduced on the device in a fixed order (jg_gn_get_objective): what the result record carries.  `objective`
        above sums the pulled residuals on the host -- [batch, m] doubles over PCIe; both follow equations.jl:689-698 and agree to rounding."""
        o = np.zeros(self.batch)
        _lib.check(_lib.lib().jg_gn_get_objective(self._h, o))
        return float(o[0]) if self.batch == 1 else o

    def pack_results_device(self, ptr: int):
        """The result record of the last stateEstimation_ -- magnitude | angle | iterations | status | objective per scenario, [batch, 2 n + 3] float64 --
        into DEVICE memory of the caller (the operand of the one gather of a sharded Monte-Carlo run)."""
        _lib.check(_lib.lib().jg_gn_pack_results_device(self._h, _lib.VP(int(ptr))))

    def time_kernel(self, kernel: int, reps: int = 10) -> float:
        ms = C.c_double(0.0)
        _lib.check(_lib.lib().jg_gn_time_kernel(self._h, int(kernel), int(reps), C.byref(ms)))
        return ms.value


def _pmu_layout(mon: Measurement):
    """Rows of pmuEstimationWls (pmuStateEstimation.jl:72-177): two LINEAR rows (Re, Im) per PMU in device order."""
    p = mon.pmu
    code, index, corr_rows, devs, dev_row = [], [], [], [], []
    for i, k in enumerate(p.layout.index):
        devs.append(("p", i)); dev_row.append(len(code))
        if p.layout.correlated[i]:
            corr_rows.append(len(code) + 1)
        code += [22, 23] if p.layout.bus[i] else ([24, 25] if p.layout.from_[i] else [26, 27])
        index += [k, k]
    return (np.array(code, dtype=np.int8), np.array(index, dtype=np.int64), np.array([1, len(code) + 1], dtype=np.int64),
            np.array(corr_rows, dtype=np.int64), devs, np.array(dev_row, dtype=np.int64))


def _pmu_readings(mon: Measurement):
    f8 = lambda x: np.array(x, dtype=np.float64)
    p = mon.pmu
    return (f8(p.magnitude.mean), f8(p.magnitude.variance), np.array(p.magnitude.status, dtype=np.int8),
            f8(p.angle.mean), f8(p.angle.variance), np.array(p.angle.status, dtype=np.int8))


def _pmu_values(mon: Measurement, devs, dev_row, m, z1, v1, s1, z2, v2, s2):
    """se.mean and se.precision of the linear model (:98-117): every PMU in rectangular form, whatever its `polar` flag."""
    shape = z1.shape[:-1]
    mean, wdiag = np.zeros(shape + (m,)), np.ones(shape + (m,))
    status = np.zeros(m, dtype=np.int8)
    woff = []
    for d, (_, i) in enumerate(devs):
        r = int(dev_row[d])
        stt = int(s1[d]) * int(s2[d])                                          # both channels in service (:120, :133)
        status[r] = status[r + 1] = stt
        re, im, wre, wim, off = _rectangular_pmu(z1[..., d], v1[d], z2[..., d], v2[d], mon.pmu.layout.correlated[i])
        mean[..., r], mean[..., r + 1] = stt * re, stt * im
        wdiag[..., r], wdiag[..., r + 1] = wre, wim
        if off is not None:
            woff.append(off)
    woff = np.stack(woff, axis=-1) if woff else np.zeros(shape + (0,))
    if not (np.all(np.isfinite(wdiag)) and np.all(wdiag > 0) and np.all(np.isfinite(woff))):
        raise ValueError("The variance of a measurement is zero or negative (errorVariance).")
    return mean, wdiag, woff, status


class PmuStateEstimation(AcStateEstimation):
    """PmuStateEstimation{WLS{HIP}} (src/definition/analysis.jl, src/stateEstimation/pmuStateEstimation.jl:43-70): the
    linear model z = H [Re V; Im V] + u.  The device runs it as ONE Gauss-Newton step of linear rows from the zero
    state (residual = z): per-scenario gain H' W H (W depends on the readings through variancePmu), block LU, solve.
    `method.coefficient` is H (constant), `method.precision`/`mean` as in the reference."""
    _needs_slack = False
    _layout = staticmethod(_pmu_layout)
    _raw = staticmethod(_pmu_readings)
    _values = staticmethod(_pmu_values)

    def _start(self, sysm):
        n = sysm.bus.number
        self._zero = np.zeros(n)
        self._rect = None
        self.voltage.magnitude = self._shape(np.tile(np.asarray(sysm.bus.voltage.magnitude, dtype=np.float64), (self.batch, 1)))
        self.voltage.angle = self._shape(np.tile(np.asarray(sysm.bus.voltage.angle, dtype=np.float64), (self.batch, 1)))

    @property
    def coefficient(self):
        """se.coefficient: the constant H [2 * pmu.number, 2 * bus.number] (CSC, the reference's pattern)."""
        L = _lib.lib()
        _lib.check(L.jg_gn_set_voltage(self._h, self._zero, self._zero, 0))
        mx = np.zeros(self.batch)
        _lib.check(L.jg_gn_increment(self._h, mx))
        return self.jacobian


class WlsMethod:
    """Factorisation tags of the reference (src/definition/analysis.jl:36-99).  The Normal tags all run the gain matrix through
    the block engine (the L/U factors are parity-unpinned); Orthogonal and PetersWilkinson -- the reference's two
    ways around the squared condition number of the gain matrix -- run the corrected semi-normal equations (include/jgrid.h,
    jg_gn_set_method)."""
    code = 0


class Normal(WlsMethod): pass          # noqa: E701
class LU(Normal): pass                 # noqa: E701
class KLU(Normal): pass                # noqa: E701
class QR(Normal): pass                 # noqa: E701
class LDLt(Normal): pass               # noqa: E701
class LL(Normal): pass                 # noqa: E701
class Orthogonal(WlsMethod): code = 1          # noqa: E701
class PetersWilkinson(WlsMethod): code = 1     # noqa: E701


def _tagged(an, method, who):
    if not (isinstance(method, type) and issubclass(method, WlsMethod)):
        an.close()
        raise TypeError(who + "(monitoring, T): T must be one of LU, KLU, QR, LDLt, LL, Orthogonal, PetersWilkinson")
    an.factorization = method
    if method.code:
        try:
            _lib.check(_lib.lib().jg_gn_set_method(an._h, method.code))
        except Exception:
            an.close()
            raise
    return an


def pmuStateEstimation(monitoring: Measurement, method=LU, batch: int = 1, device: int = 0) -> PmuStateEstimation:
    """pmuStateEstimation(monitoring[, T]) (pmuStateEstimation.jl:43-70): linear WLS model with PMUs only."""
    if monitoring.pmu.number == 0:
        raise ValueError("the measurement set holds no PMU")
    return _tagged(PmuStateEstimation(monitoring, batch, device), method, "pmuStateEstimation")


def _solve_pmu(an: PmuStateEstimation, fetch: bool = True):
    """solve!(analysis::PmuStateEstimation{WLS{Normal}}) (:369-399): gain = H' W H, b = H' W z, factorise, solve,
    then (Re, Im) -> (magnitude, angle)."""
    L = _lib.lib()
    _lib.check(L.jg_gn_set_voltage(an._h, an._zero, an._zero, 0))
    mx = np.zeros(an.batch)
    _lib.check(L.jg_gn_increment(an._h, mx))
    _lib.check(L.jg_gn_solve(an._h))
    an.status = 0 if an.batch == 1 else np.zeros(an.batch, dtype=np.int32)   # a singular gain raised above
    if fetch:
        n = an.system.bus.number
        im, re = np.zeros((an.batch, n)), np.zeros((an.batch, n))
        _lib.check(L.jg_gn_get_voltage(an._h, im, re))                         # (vm, va) hold (Im, Re)
        an.voltage.magnitude, an.voltage.angle = an._shape(np.hypot(re, im)), an._shape(np.arctan2(im, re))


def gaussNewton(monitoring: Measurement, method=LU, batch: int = 1, device: int = 0) -> AcStateEstimation:
    """gaussNewton(monitoring[, T]) (acStateEstimation.jl:43-75): WLS model + symbolic analysis + upload; start = system.bus.voltage.
    T: LU (default) | KLU | QR | LDLt | LL | Orthogonal | PetersWilkinson."""
    return _tagged(AcStateEstimation(monitoring, batch, device), method, "gaussNewton")


def _upload_readings(an: AcStateEstimation):
    """The raw readings per device, once, for jg_gn_draw_noise (include/jgrid.h: jg_gn_set_readings): first row, kind of value rule, z / variance / status."""
    mon, devs = an.monitoring, an._devs
    z1, v1, s1, z2, v2, s2 = an._z
    kind = np.zeros(len(devs), dtype=np.int8)
    a, p = mon.ammeter, mon.pmu
    for d, (fam, i) in enumerate(devs):
        if fam != "p":
            kind[d] = 1 if (fam == "a" and a.layout.square[i]) else 0
        elif p.layout.polar[i]:
            kind[d] = 3 if (p.layout.square[i] and not p.layout.bus[i]) else 2
        else:
            kind[d] = 5 if p.layout.correlated[i] else 4
    _lib.check(_lib.lib().jg_gn_set_readings(an._h, len(devs), np.ascontiguousarray(an._dev_row + 1, dtype=np.int64), kind,
                                             np.ascontiguousarray(z1), np.ascontiguousarray(v1), np.ascontiguousarray(s1, dtype=np.int8),
                                             np.ascontiguousarray(z2), np.ascontiguousarray(v2), np.ascontiguousarray(s2, dtype=np.int8)))
    an._readings_on_device = True


def drawNoise_(an: AcStateEstimation, seed: int, scale: float = 1.0, first: int = 0):
    """Monte-Carlo realisations drawn ON the device (jg_gn_draw_noise): lane b becomes realisation `first + b` of `seed` -- z + scale * sigma * N(0,1) on every raw
    reading (measurement/utility.jl:70-73) from a counter-based generator, then the acWLS value rules -- without a byte over PCIe; the same (seed, realisation)
    gives the same numbers on any rank, in any batch, at any lane.  The host mirrors (method.mean, the precision) are marked stale and pulled from the device by
    whoever reads them next (residualTest_, objective, precision, chiTest); rows a residual test removed stay removed in the new realisations."""
    if isinstance(an, PmuStateEstimation):
        raise TypeError("drawNoise_: Gauss-Newton analyses (the linear PMU model keeps its own rows)")
    if not getattr(an, "_readings_on_device", False):
        _upload_readings(an)
    _lib.check(_lib.lib().jg_gn_draw_noise(an._h, int(seed), float(scale), int(first)))
    an._mirror_stale = True
    gone = getattr(an, "_removed", None)
    if gone is not None and np.any(gone):                         # the draw rewrote every row: what bad-data processing took out of a scenario's model goes out again
        an._sync_mirrors()
        mean = np.array(np.atleast_2d(an.method.mean))
        wd = np.array(np.atleast_2d(an.method._wdiag))
        wo = np.array(np.atleast_2d(an.method._woff))
        mean[gone] = 0.0
        wd[gone] = 0.0
        for q, r in enumerate(an.method._corr):
            wo[gone[:, r - 1] | gone[:, r], q] = 0.0
        one = an.batch == 1
        an._upload_measurement(mean[0] if one else mean, wd[0] if one else wd, (wo[0] if one else wo) if an.method._corr.size else np.zeros(0))


def measurementDevice(an: AcStateEstimation):
    """(se.mean, diag(se.precision), pair terms) of every scenario as the device holds them: [batch, m], [batch, m], [batch, n_corr]."""
    m, nc = an.dims["m"], int(an.method._corr.size)
    mean, wd, wo = np.zeros((an.batch, m)), np.zeros((an.batch, m)), np.zeros((an.batch, max(nc, 1)))
    _lib.check(_lib.lib().jg_gn_get_measurement(an._h, mean, wd, wo))
    return mean, wd, wo[:, :nc]


def setNoise_(an: AcStateEstimation, rng, scale: float = 1.0):
    """Monte-Carlo realisations: scenario b reads z + scale * sigma * N(0,1) on every raw meter quantity
    (what `noise = true` does in add*!, measurement/utility.jl:70-73), then the acWLS value rules are
    re-applied per scenario (squared currents and rectangular PMUs make mean AND precision depend on z).
    Host path (numpy generator, [batch, m] arrays over PCIe); drawNoise_ does the same on the device from an integer seed."""
    z1, v1, s1, z2, v2, s2 = an._z
    B = an.batch
    n1 = z1[None, :] + scale * np.sqrt(v1)[None, :] * rng.standard_normal((B, z1.size))
    n2 = z2[None, :] + scale * np.sqrt(v2)[None, :] * rng.standard_normal((B, z2.size))
    mean, wdiag, woff, _ = an._values(an.monitoring, an._devs, an._dev_row, an.dims["m"], n1, v1, s1, n2, v2, s2)
    an._upload_measurement(mean if B > 1 else mean[0], wdiag if B > 1 else wdiag[0], woff if B > 1 else woff[0])


def increment_(an: AcStateEstimation):
    """increment!(analysis) -> maximum(abs, increment) (array for batch > 1)."""
    mx = np.zeros(an.batch)
    _lib.check(_lib.lib().jg_gn_increment(an._h, mx))
    return float(mx[0]) if an.batch == 1 else mx


def solve_(an: AcStateEstimation):
    """solve!(analysis): theta += increment[1:n], V += increment[n+1:2n]; iteration += 1.
    For a PmuStateEstimation: the whole linear WLS solve."""
    if isinstance(an, PmuStateEstimation):
        return _solve_pmu(an)
    _lib.check(_lib.lib().jg_gn_solve(an._h))
    an.method.iteration += 1
    an._pull_voltage()


def stateEstimation_(an: AcStateEstimation, iteration: int = 40, tolerance: float = 1e-8, fetch: bool = True):
    """stateEstimation!(analysis; iteration, tolerance)."""
    if isinstance(an, PmuStateEstimation):
        return _solve_pmu(an, fetch)
    it = np.zeros(an.batch, dtype=np.int32)
    st = np.zeros(an.batch, dtype=np.int32)
    _lib.check(_lib.lib().jg_gn_run(an._h, int(iteration), float(tolerance), it, st))
    an.method.iteration = int(it[0]) if an.batch == 1 else it
    an.status = int(st[0]) if an.batch == 1 else st
    if fetch:
        an._pull_voltage()


# ---- bad data processing (src/stateEstimation/badData.jl) ----------------------------------------------------------------
_FAMILY = {"v": ("Voltmeter", "voltmeter", "magnitude"), "a": ("Ammeter", "ammeter", "magnitude"),
           "w": ("Wattmeter", "wattmeter", "active"), "r": ("Varmeter", "varmeter", "reactive"), "p": ("PMU", "pmu", None)}


def _device_of_row(an, row0):
    """(device position d, first row of the device) for a 0-based measurement row."""
    d = int(np.searchsorted(an._dev_row, row0, side="right") - 1)
    return d, int(an._dev_row[d])


def residualTest_(an: AcStateEstimation, threshold: float = 3.0):
    """residualTest!(analysis; threshold) (badData.jl:119-311), for every scenario of the batch.

    The device recomputes residual, Jacobian, gain and its factor at the current state, forms the selected inverse of
    the gain on its factor pattern and the normalised residuals |r_i| / sqrt(|1 / W_ii - c_i|) (jg_gn_residual_test).
    Here: the reference's bookkeeping.  When the largest normalised residual of a scenario exceeds `threshold`, that
    measurement leaves the scenario's model (its weight becomes 0 = removeRow + mean = residual = 0 + type = 0; both
    rows of a rectangular PMU, :280-290 / :155-166) and the iteration counter restarts.  batch == 1 additionally sets the
    device's status in the Measurement container, like the reference.
    Returns a namespace (detect, maxNormalizedResidual, label, index), fields are arrays for batch > 1; index is the
    1-based row of se.mean (0: every residual is zero)."""
    mx = np.zeros(an.batch)
    idx = np.zeros(an.batch, dtype=np.int32)
    _lib.check(_lib.lib().jg_gn_residual_test(an._h, mx, idx))
    detect = mx > threshold
    mon, m = an.monitoring, an.dims["m"]
    an._sync_mirrors()                                            # the lanes' own realisations, not the set the analysis was created with
    linear = isinstance(an, PmuStateEstimation)
    labels = []
    wd = np.array(np.broadcast_to(np.atleast_2d(an.method._wdiag), (an.batch, m)))
    nc = an.method._corr.size
    wo = np.array(np.broadcast_to(np.atleast_2d(an.method._woff), (an.batch, max(nc, 1)))) if nc else np.zeros((an.batch, 0))
    mean = np.array(np.broadcast_to(np.atleast_2d(an.method.mean), (an.batch, m)))
    changed = False
    for b in range(an.batch):
        if idx[b] == 0:
            labels.append("")
            continue
        r = int(idx[b]) - 1
        d, r0 = _device_of_row(an, r)
        fam, i = an._devs[d]
        name, field, chan = _FAMILY[fam]
        labels.append(f"{name} {i + 1}")
        if not detect[b]:
            continue
        rows = [r]
        both = fam == "p" and (linear or not mon.pmu.layout.polar[i])
        if both:
            rows = [r0, r0 + 1]
        for q in rows:
            wd[b, q] = 0.0
            mean[b, q] = 0.0
        if both and nc:
            hit = np.flatnonzero(an.method._corr - 1 == r0)
            wo[b, hit] = 0.0
        changed = True
        if an.batch == 1:                                   # the Measurement container follows, like the reference
            if fam != "p":
                getattr(getattr(mon, field), chan).status[i] = 0
            elif both:
                mon.pmu.magnitude.status[i] = 0
                mon.pmu.angle.status[i] = 0
            elif an.method._code[r] in (2, 3, 4, 5, 12):
                mon.pmu.magnitude.status[i] = 0
            else:
                mon.pmu.angle.status[i] = 0
            for q in rows:
                an.method.type[q] = 0
    if changed:
        one = an.batch == 1
        an._upload_measurement(mean[0] if one else mean, wd[0] if one else wd, (wo[0] if one else wo) if nc else np.zeros(0))
        prev = getattr(an, "_removed", None)
        an._removed = (np.zeros((an.batch, m), dtype=bool) if prev is None else prev) | (wd == 0.0)
        an.method.iteration = 0
    one = an.batch == 1
    return NS(detect=bool(detect[0]) if one else detect, maxNormalizedResidual=float(mx[0]) if one else mx,
              label=labels[0] if one else labels, index=int(idx[0]) if one else idx)


def normalizedResidual(an: AcStateEstimation):
    """All normalised residuals of the last residualTest_ call [batch, m] (the reference keeps only the largest)."""
    r = np.zeros((an.batch, an.dims["m"]))
    _lib.check(_lib.lib().jg_gn_get_normalized_residual(an._h, r))
    return an._shape(r)


def chiTest(an: AcStateEstimation, confidence: float = 0.95):
    """chiTest(analysis; confidence) (badData.jl:948-995): objective r' W r at the current state against the
    chi-square quantile with df = rows in service - state variables.  Returns (detect, threshold, objective)."""
    from scipy.stats import chi2
    _lib.check(_lib.lib().jg_gn_evaluate(an._h))
    n = an.system.bus.number
    dead = np.count_nonzero(an.method.type == 0)
    gone = getattr(an, "_removed", None)
    extra = (gone & (an.method.type != 0)[None, :]).sum(axis=1) if gone is not None and an.batch > 1 else 0
    if isinstance(an, PmuStateEstimation):
        df = an.method.type.size - dead - extra - 2 * n                       # se.inservice - 2 * bus.number (:992)
    else:
        df = an.method.type.size - dead - extra - 2 * n + 1                   # :958
    thr = chi2.ppf(confidence, df)
    obj = an.objective
    if an.batch == 1:
        return NS(detect=bool(obj >= thr), threshold=float(thr), objective=float(obj))
    return NS(detect=obj >= thr, threshold=np.broadcast_to(thr, obj.shape).copy(), objective=obj)


# ---- update<Meter>!(monitoring | analysis; label, ...) (src/measurement/*.jl) -------------------------------------------
def _refresh(an: AcStateEstimation):
    """The analysis follows its Measurement container (powermeter.jl:640-677, voltmeter.jl:258-290, ammeter.jl:367-420,
    pmu.jl:559-700): se.type = status * code, se.mean, se.precision; the Jacobian pattern stays.  For batch > 1 every
    scenario restarts from the container's readings (noise realisations of setNoise_ are dropped)."""
    z = an._raw(an.monitoring)
    an._z = z
    an._readings_on_device = False                                 # drawNoise_ uploads the new readings / variances / statuses before its next draw
    code = an._layout(an.monitoring)[0]
    mean, wdiag, woff, status = an._values(an.monitoring, an._devs, an._dev_row, an.dims["m"], *z)
    _lib.check(_lib.lib().jg_gn_set_status(an._h, np.ascontiguousarray(status, dtype=np.int8), code))
    an.method._code = code
    an.method.type = (status * code).astype(np.int8)
    an._upload_measurement(mean, wdiag, woff)
    an._removed = None


def _update(target, family, label, channels, layout=None):
    """channels: {gauss field: (mean, variance, status)}; label = 1-based device number inside its family;
    layout: {layout flag: new value or None}."""
    mon = target.monitoring if isinstance(target, AcStateEstimation) else target
    meter = getattr(mon, family)
    i = int(label) - 1
    if not 0 <= i < meter.number:
        raise KeyError(f"The {family} labelled {label} does not exist.")
    for flag, value in (layout or {}).items():
        if value is not None:
            getattr(meter.layout, flag)[i] = bool(value)
    for field, (mean, variance, status) in channels.items():
        g = getattr(meter, field)
        if mean is not None:
            g.mean[i] = float(mean)
        if variance is not None:
            g.variance[i] = float(variance)
        if status is not None:
            if status not in (0, 1):
                raise ValueError("status must be 0 or 1")
            g.status[i] = int(status)
    if isinstance(target, AcStateEstimation):
        _refresh(target)


def updateVoltmeter_(target, label, magnitude=None, variance=None, status=None):
    """updateVoltmeter!(monitoring | analysis; label, magnitude, variance, status) (voltmeter.jl:194-290)."""
    _update(target, "voltmeter", label, {"magnitude": (magnitude, variance, status)})


def updateAmmeter_(target, label, magnitude=None, variance=None, status=None, square=None):
    """updateAmmeter!(...) (ammeter.jl:293-420); `square` switches between the magnitude and its square."""
    _update(target, "ammeter", label, {"magnitude": (magnitude, variance, status)}, {"square": square})


def updateWattmeter_(target, label, active=None, variance=None, status=None):
    """updateWattmeter!(...) (powermeter.jl:561-677)."""
    _update(target, "wattmeter", label, {"active": (active, variance, status)})


def updateVarmeter_(target, label, reactive=None, variance=None, status=None):
    """updateVarmeter!(...) (powermeter.jl:822-940)."""
    _update(target, "varmeter", label, {"reactive": (reactive, variance, status)})


def updatePmu_(target, label, magnitude=None, angle=None, varianceMagnitude=None, varianceAngle=None,
               statusMagnitude=None, statusAngle=None, status=None):
    """updatePmu!(...) (pmu.jl:435-700); `status` sets both channels."""
    sm = status if statusMagnitude is None else statusMagnitude
    sa = status if statusAngle is None else statusAngle
    _update(target, "pmu", label, {"magnitude": (magnitude, varianceMagnitude, sm), "angle": (angle, varianceAngle, sa)})
