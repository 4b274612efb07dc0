"""Measurement container and exact-measurement synthesis (host side of the state-estimation path).

Mirrors the parts of the reference's measurement layer the Gauss-Newton path consumes
(paths relative to /root/reference):

  measurement(system)                         src/measurement/load.jl (empty container)
  addVoltmeter!(monitoring, analysis; ...)    src/measurement/voltmeter.jl:123
  addAmmeter!(monitoring, analysis; ...)      src/measurement/ammeter.jl:169
  addWattmeter!/addVarmeter!(monitoring, a.)  src/measurement/powermeter.jl:412-529
  addPmu!(monitoring, analysis; ...)          src/measurement/pmu.jl:253-405
  add*!(monitoring; bus|from|to = ..., ...)   single devices
  power!/current! per element                 src/postprocessing/acAnalysis.jl:30-78, 672-704, 838-931

Row ordering: inside a family built from an analysis, all buses in index order, then
for every IN-SERVICE branch its from end, then its to end.  Values are exact (noise-free) unless
`noise=True`, in which case `rng` (numpy Generator) supplies the Gaussian draws -- the reference uses
Julia's global RNG, which cannot be reproduced, so parity tests use noise-free sets.
Default variances 1e-4 (legacy) / 1e-8 (PMU), default statuses 1 (src/definition/internal.jl:173-232).
status -1 = do not create that group (powermeter.jl:440-452).

Product host code (numpy); never imports oracle/.
"""
from __future__ import annotations

from types import SimpleNamespace as NS

import numpy as np

from .system import PowerSystem


def _gauss():
    return NS(mean=[], variance=[], status=[])


class Measurement:
    def __init__(self, system: PowerSystem):
        self.system = system
        self.voltmeter = NS(number=0, layout=NS(index=[]), magnitude=_gauss())
        self.ammeter = NS(number=0, layout=NS(index=[], from_=[], to=[], square=[]), magnitude=_gauss())
        self.wattmeter = NS(number=0, layout=NS(index=[], bus=[], from_=[], to=[]), active=_gauss())
        self.varmeter = NS(number=0, layout=NS(index=[], bus=[], from_=[], to=[]), reactive=_gauss())
        self.pmu = NS(number=0, layout=NS(index=[], bus=[], from_=[], to=[], correlated=[], polar=[], square=[]),
                      magnitude=_gauss(), angle=_gauss())


def measurement(system: PowerSystem, source: str | None = None) -> Measurement:
    """measurement(system) / measurement(system, "monitoring.h5") (src/measurement/load.jl: the reference's HDF5 layout,
    one group per meter family, layout / mean / variance / status datasets, one-element datasets broadcast)."""
    mon = Measurement(system)
    if source is None:
        return mon
    from .hdf5 import H5File
    f = H5File(str(source))

    def col(name, count, cast):
        if name not in f:
            return None
        a = f.read(name)
        a = np.full(count, a[0], dtype=a.dtype) if a.size == 1 and count != 1 else a
        if a.size != count:
            raise ValueError(f"{source}: {name} has {a.size} elements, expected {count} or 1")
        return [cast(x) for x in a]

    families = (("voltmeter", ("magnitude",), ()), ("ammeter", ("magnitude",), ("from", "to", "square")),
                ("wattmeter", ("active",), ("bus", "from", "to")), ("varmeter", ("reactive",), ("bus", "from", "to")),
                ("pmu", ("magnitude", "angle"), ("bus", "from", "to", "correlated", "polar", "square")))
    for fam, channels, flags in families:
        if f"/{fam}/layout/index" not in f:
            continue
        meter = getattr(mon, fam)
        index = [int(x) for x in f.read(f"/{fam}/layout/index")]
        k = len(index)
        meter.layout.index = index
        for flag in flags:
            setattr(meter.layout, "from_" if flag == "from" else flag, col(f"/{fam}/layout/{flag}", k, bool) or [False] * k)
        for ch in channels:
            g = getattr(meter, ch)
            g.mean, g.variance, g.status = (col(f"/{fam}/{ch}/mean", k, float), col(f"/{fam}/{ch}/variance", k, float),
                                            col(f"/{fam}/{ch}/status", k, int))
        meter.number = k
    return mon


def ems(case: str, monitoring: str):
    """ems("case14.h5", "monitoring.h5") (src/JuliaGrid.jl export; load.jl): the power system and its measurements."""
    from .system import powerSystem
    system = powerSystem(case)
    return system, measurement(system, monitoring)


# ---- exact quantities from a solved state (postprocessing/acAnalysis.jl:838-931) -------------------
def exactQuantities(system: PowerSystem, magnitude, angle):
    """Bus injections (P_i, Q_i), branch flows (P_ij, Q_ij, P_ji, Q_ji) and branch currents
    (|I_ij|, arg I_ij, |I_ji|, arg I_ji) for one voltage profile; zeros on out-of-service branches."""
    ac = system.model.ac
    V = np.asarray(magnitude) * np.exp(1j * np.asarray(angle))
    Y = ac.nodalMatrix
    n = system.bus.number
    col_of = np.repeat(np.arange(n), np.diff(Y.colptr))
    I = np.zeros(n, dtype=np.complex128)
    np.add.at(I, Y.rowval - 1, Y.nzval * V[col_of])                 # I = Y V  (Ii, :867-883)
    S = V * np.conj(I)                                              # PiQi (:885-891)
    f = system.branch.layout.from_ - 1
    t = system.branch.layout.to - 1
    on = system.branch.layout.status == 1
    Iij = np.where(on, V[f] * ac.nodalFromFrom + V[t] * ac.nodalFromTo, 0)     # :915-917
    Iji = np.where(on, V[f] * ac.nodalToFrom + V[t] * ac.nodalToTo, 0)         # :919-921
    Sij = V[f] * np.conj(Iij)                                       # PijQij (:893-895)
    Sji = V[t] * np.conj(Iji)
    return NS(injectionActive=S.real, injectionReactive=S.imag, fromActive=Sij.real, fromReactive=Sij.imag,
              toActive=Sji.real, toReactive=Sji.imag, fromMagnitude=np.abs(Iij), fromAngle=np.angle(Iij),
              toMagnitude=np.abs(Iji), toAngle=np.angle(Iji))


def _analysis_state(analysis):
    vm, va = np.asarray(analysis.voltage.magnitude), np.asarray(analysis.voltage.angle)
    if vm.ndim != 1:
        raise ValueError("measurement synthesis needs a single-scenario analysis (batch == 1)")
    return vm, va


def _put(meter, mean, variance, status, noise, rng):
    if noise:
        mean = mean + np.sqrt(variance) * (rng if rng is not None else np.random.default_rng()).standard_normal()
    meter.mean.append(float(mean))
    meter.variance.append(float(variance))
    meter.status.append(int(status))


def addVoltmeter_(monitoring: Measurement, analysis=None, *, bus=None, magnitude=None, variance=1e-4, status=1,
                  noise=False, rng=None):
    v = monitoring.voltmeter
    if analysis is None:
        v.layout.index.append(int(bus))
        _put(v.magnitude, magnitude, variance, status, noise, rng)
        v.number += 1
        return
    if status == -1:
        return
    vm, _ = _analysis_state(analysis)
    for i in range(monitoring.system.bus.number):
        v.layout.index.append(i + 1)
        _put(v.magnitude, vm[i], variance, status, noise, rng)
        v.number += 1


def addAmmeter_(monitoring: Measurement, analysis=None, *, from_=None, to=None, magnitude=None, variance=None,
                varianceFrom=1e-4, varianceTo=1e-4, status=1, statusFrom=1, statusTo=1, square=False, noise=False, rng=None,
                minMagnitude=0.0):
    """minMagnitude (extension, analysis form): skip branch ends whose exact current is below it -- a zero
    current makes the squared-magnitude variance 4 z^2 sigma^2 vanish (the reference raises there)."""
    a = monitoring.ammeter
    if analysis is None:
        k, is_from = (int(from_), True) if from_ is not None else (int(to), False)
        a.layout.index.append(k); a.layout.from_.append(is_from); a.layout.to.append(not is_from); a.layout.square.append(bool(square))
        _put(a.magnitude, magnitude, variance if variance is not None else (varianceFrom if is_from else varianceTo), status, noise, rng)
        a.number += 1
        return
    q = exactQuantities(monitoring.system, *_analysis_state(analysis))
    for k in np.flatnonzero(monitoring.system.branch.layout.status == 1):
        for is_from, st, var, val in ((True, statusFrom, varianceFrom, q.fromMagnitude[k]), (False, statusTo, varianceTo, q.toMagnitude[k])):
            if st == -1 or val < minMagnitude:
                continue
            a.layout.index.append(int(k) + 1); a.layout.from_.append(is_from); a.layout.to.append(not is_from); a.layout.square.append(bool(square))
            _put(a.magnitude, val, var, st, noise, rng)
            a.number += 1


def _add_powermeter(meter, gauss, system, analysis, bus, from_, to, value, variance, varianceBus, varianceFrom, varianceTo,
                    status, statusBus, statusFrom, statusTo, noise, rng, busval, fromval, toval):
    lay = meter.layout
    if analysis is None:
        loc = 0 if bus is not None else (1 if from_ is not None else 2)
        k = int(bus if loc == 0 else (from_ if loc == 1 else to))
        lay.index.append(k); lay.bus.append(loc == 0); lay.from_.append(loc == 1); lay.to.append(loc == 2)
        _put(gauss, value, variance if variance is not None else (varianceBus, varianceFrom, varianceTo)[loc], status, noise, rng)
        meter.number += 1
        return
    if statusBus != -1:
        for i in range(system.bus.number):
            lay.index.append(i + 1); lay.bus.append(True); lay.from_.append(False); lay.to.append(False)
            _put(gauss, busval[i], varianceBus, statusBus, noise, rng)
            meter.number += 1
    for k in np.flatnonzero(system.branch.layout.status == 1):
        for loc, st, var, val in ((1, statusFrom, varianceFrom, fromval[k]), (2, statusTo, varianceTo, toval[k])):
            if st == -1:
                continue
            lay.index.append(int(k) + 1); lay.bus.append(False); lay.from_.append(loc == 1); lay.to.append(loc == 2)
            _put(gauss, val, var, st, noise, rng)
            meter.number += 1


def addWattmeter_(monitoring: Measurement, analysis=None, *, bus=None, from_=None, to=None, active=None, variance=None,
                  varianceBus=1e-4, varianceFrom=1e-4, varianceTo=1e-4, status=1, statusBus=1, statusFrom=1, statusTo=1,
                  noise=False, rng=None):
    q = exactQuantities(monitoring.system, *_analysis_state(analysis)) if analysis is not None else None
    _add_powermeter(monitoring.wattmeter, monitoring.wattmeter.active, monitoring.system, analysis, bus, from_, to, active,
                    variance, varianceBus, varianceFrom, varianceTo, status, statusBus, statusFrom, statusTo, noise, rng,
                    q and q.injectionActive, q and q.fromActive, q and q.toActive)


def addVarmeter_(monitoring: Measurement, analysis=None, *, bus=None, from_=None, to=None, reactive=None, variance=None,
                 varianceBus=1e-4, varianceFrom=1e-4, varianceTo=1e-4, status=1, statusBus=1, statusFrom=1, statusTo=1,
                 noise=False, rng=None):
    q = exactQuantities(monitoring.system, *_analysis_state(analysis)) if analysis is not None else None
    _add_powermeter(monitoring.varmeter, monitoring.varmeter.reactive, monitoring.system, analysis, bus, from_, to, reactive,
                    variance, varianceBus, varianceFrom, varianceTo, status, statusBus, statusFrom, statusTo, noise, rng,
                    q and q.injectionReactive, q and q.fromReactive, q and q.toReactive)


def addPmu_(monitoring: Measurement, analysis=None, *, bus=None, from_=None, to=None, magnitude=None, angle=None,
            varianceMagnitude=None, varianceAngle=None, varianceMagnitudeBus=1e-8, varianceAngleBus=1e-8,
            varianceMagnitudeFrom=1e-8, varianceAngleFrom=1e-8, varianceMagnitudeTo=1e-8, varianceAngleTo=1e-8,
            statusMagnitude=1, statusAngle=1, statusBus=1, statusFrom=1, statusTo=1, correlated=False, polar=False,
            square=False, noise=False, rng=None, buses=None, minMagnitude=0.0):
    """`buses` (optional, analysis form): restrict bus PMUs to these 1-based bus indices and branch PMUs to
    branches leaving them (PMU placement), keeping the reference's ordering inside the selection.
    `minMagnitude` (extension): skip branch PMUs whose exact current magnitude is below it (a zero phasor
    has no angle and a vanishing rectangular variance; the reference raises errorVariance there)."""
    p = monitoring.pmu
    lay = p.layout

    def put(k, loc, zm, za, vm_, va_, sm, sa, sq):
        lay.index.append(int(k)); lay.bus.append(loc == 0); lay.from_.append(loc == 1); lay.to.append(loc == 2)
        lay.correlated.append(bool(correlated)); lay.polar.append(bool(polar)); lay.square.append(bool(sq))
        _put(p.magnitude, zm, vm_, sm, noise, rng)
        _put(p.angle, za, va_, sa, noise, rng)
        p.number += 1

    if analysis is None:
        loc = 0 if bus is not None else (1 if from_ is not None else 2)
        k = bus if loc == 0 else (from_ if loc == 1 else to)
        vm_ = varianceMagnitude if varianceMagnitude is not None else (varianceMagnitudeBus, varianceMagnitudeFrom, varianceMagnitudeTo)[loc]
        va_ = varianceAngle if varianceAngle is not None else (varianceAngleBus, varianceAngleFrom, varianceAngleTo)[loc]
        put(k, loc, magnitude, angle, vm_, va_, statusMagnitude, statusAngle, square and loc != 0)
        return
    system = monitoring.system
    vmag, vang = _analysis_state(analysis)
    q = exactQuantities(system, vmag, vang)
    sel = None if buses is None else set(int(b) for b in buses)
    if statusBus != -1:
        for i in range(system.bus.number):
            if sel is None or (i + 1) in sel:
                put(i + 1, 0, vmag[i], vang[i], varianceMagnitudeBus, varianceAngleBus, statusBus, statusBus, False)   # pmu.jl:335 square=false
    for k in np.flatnonzero(system.branch.layout.status == 1):
        f, t = int(system.branch.layout.from_[k]), int(system.branch.layout.to[k])
        if statusFrom != -1 and (sel is None or f in sel) and q.fromMagnitude[k] >= minMagnitude:
            put(k + 1, 1, q.fromMagnitude[k], q.fromAngle[k], varianceMagnitudeFrom, varianceAngleFrom, statusFrom, statusFrom, square)
        if statusTo != -1 and (sel is None or t in sel) and q.toMagnitude[k] >= minMagnitude:
            put(k + 1, 2, q.toMagnitude[k], q.toAngle[k], varianceMagnitudeTo, varianceAngleTo, statusTo, statusTo, square)
