# JuliaGridHIP.jl -- ccall shim that plugs libjgrid_hip.so (include/jgrid.h) behind JuliaGrid's own
# newtonRaphson() / mismatch!() / solve!() / powerFlow!() and gaussNewton() / increment!() / solve!() / stateEstimation!()
# surface.  Written against JuliaGrid v0.6.2; it has NOT been executed (no Julia toolchain in the build containers) --
# see INTEGRATION.md for the one-token patch of the reference it needs and for how every row below maps to the C ABI.
#
# The analyses are the REFERENCE'S OWN structs, parametrised by a new solver tag:
#     AcPowerFlow{NewtonRaphson{HIP}}          (src/definition/analysis.jl:154-164, 252-258)
#     AcStateEstimation{GaussNewton{HIP}}      (src/definition/analysis.jl:532-545, 643-650)
# so the reference's generic loops powerFlow! (src/powerFlow/acPowerFlow.jl:1389-1433) and stateEstimation!
# (src/stateEstimation/acStateEstimation.jl:1286-1329), its printing, power!/current! and every reader of
# analysis.voltage.{magnitude,angle} / analysis.method.{jacobian,mismatch,increment,residual,iteration} work unchanged;
# what dispatches here are the methods that do numerics.
#
#   reference method (file:line)                                        -> C ABI
#   newtonRaphson(system, T)             acPowerFlow.jl:39-87           -> jg_nr_create, jg_nr_set_injection, jg_nr_set_voltage
#   mismatch!(analysis)                  acPowerFlow.jl:645-685         -> jg_nr_mismatch, jg_nr_get_mismatch
#   solve!(analysis)                     acPowerFlow.jl:793-911         -> jg_nr_solve, jg_nr_get_increment, jg_nr_get_voltage
#   powerFlow!(analysis)  (fused)        acPowerFlow.jl:1389-1433       -> jg_nr_run, jg_nr_get_voltage, jg_nr_get_iteration
#   setInitialPoint!(analysis)           acPowerFlow.jl:1226-1249       -> jg_nr_set_voltage
#   updateBus!/Branch!/Generator!        bus.jl:286, branch.jl:453, generator.jl:382 -> jg_nr_set_injection, jg_nr_set_ybus
#   gaussNewton(monitoring, T)           acStateEstimation.jl:43-75     -> jg_gn_create, jg_gn_set_measurement, jg_gn_set_voltage
#   increment!(analysis)                 acStateEstimation.jl:878-904   -> jg_gn_increment, jg_gn_get_increment, jg_gn_get_residual
#   solve!(analysis)                     acStateEstimation.jl:1035-1047 -> jg_gn_solve, jg_gn_get_voltage
#   stateEstimation!(analysis) (fused)   acStateEstimation.jl:1286-1329 -> jg_gn_run, jg_gn_get_voltage
#   setInitialPoint!(analysis)           acStateEstimation.jl:1071      -> jg_gn_set_voltage
#   update<Meter>!(analysis; ...)        measurement/*.jl               -> jg_gn_set_status, jg_gn_set_measurement
#   residualTest!(analysis)  (numerics)  badData.jl:181-311             -> jg_gn_residual_test, jg_gn_get_normalized_residual
#   fastNewtonRaphsonBX/XB(system, T)    acPowerFlow.jl:215-537         -> jg_nr_create, jg_nr_fast_setup
#   mismatch!/solve!/powerFlow! (fast)   acPowerFlow.jl:687-730, 913-983 -> jg_nr_fast_mismatch, jg_nr_fast_solve, jg_nr_fast_run, jg_nr_fast_get_increment
#   power!(analysis), current!(analysis) postprocessing/acAnalysis.jl:30-169, 672-704 -> jg_nr_set_branches, jg_nr_bus_injection, jg_nr_branch_quantities
#   gaussNewton(monitoring, HIPOrthogonal)  (Orthogonal / PetersWilkinson rows) acStateEstimation.jl:906-971 -> jg_gn_set_method
#   pmuStateEstimation(monitoring, T), solve!  pmuStateEstimation.jl:43-177, 369-399 -> jg_gn_create (codes 22-27), jg_gn_increment, jg_gn_solve
#   chiTest(analysis)                    badData.jl:948-995             -> jg_gn_evaluate, jg_gn_get_residual (se.objective)
#   Monte-Carlo batch (user loop over add<Meter>!(noise = true) + stateEstimation!, measurement/utility.jl:70-73) -> jg_gn_run on a batch,
#                                        jg_gn_get_objective, jg_gn_pack_results_device, jg_gn_allgather_results
module JuliaGridHIP

using JuliaGrid
using SparseArrays
import JuliaGrid: newtonRaphson, fastNewtonRaphsonBX, fastNewtonRaphsonXB, gaussNewton, pmuStateEstimation, mismatch!, solve!, increment!,
                  powerFlow!, stateEstimation!, setInitialPoint!, power!, current!, chiTest,
                  updateBus!, updateBranch!, updateGenerator!,
                  updateVoltmeter!, updateAmmeter!, updateWattmeter!, updateVarmeter!, updatePmu!,
                  AcPowerFlow, AcStateEstimation, PmuStateEstimation, NewtonRaphson, FastNewtonRaphson, FastNewtonRaphsonModel, GaussNewton, WLS,
                  PowerSystem, Measurement, LU

const lib = get(ENV, "JGRID_HIP_LIB", "libjgrid_hip.so")

"""
    HIP <: Normal

Solver tag: `newtonRaphson(system, HIP)`, `gaussNewton(monitoring, HIP)`.  `GaussNewton{T <: WlsMethod}` accepts it as it
is; `NewtonRaphson{T <: Union{LU, KLU, QR}}` needs its bound widened to `T <: Normal` (INTEGRATION.md, one token).
"""
struct HIP <: JuliaGrid.Normal end

"""
    HIPOrthogonal <: WlsMethod

`gaussNewton(monitoring, HIPOrthogonal)` / `pmuStateEstimation(monitoring, HIPOrthogonal)`: what the `Orthogonal` and `PetersWilkinson` tags of
the reference are for (acStateEstimation.jl:906-971: an increment whose error does not carry the squared condition number of the gain matrix), on the
device: the corrected semi-normal equations on the factor the engine holds (`jg_gn_set_method(h, 1)`, include/jgrid.h).  Diagonal precision only, as
in the reference (`sqrtPrecision!`).
"""
struct HIPOrthogonal <: JuliaGrid.WlsMethod end

# The structs type their `factorization` field with the FactorSparse union; the device factor lives behind the C handle, so the
# field keeps an (unused) LU placeholder and the handle is found through the method object.
JuliaGrid.selectFactorization(::Type{HIP}) = JuliaGrid.selectFactorization(LU)
JuliaGrid.selectFactorization(::Type{HIPOrthogonal}) = JuliaGrid.selectFactorization(LU)

mutable struct Handle
    ptr::Ptr{Cvoid}
    kind::Symbol                                  # :nr | :gn
    function Handle(p, kind)
        h = new(p, kind)
        finalizer(h) do x
            x.ptr == C_NULL && return
            x.kind === :nr ? ccall((:jg_nr_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ptr) :
                             ccall((:jg_gn_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ptr)
            x.ptr = C_NULL
        end
        return h
    end
end
const HANDLES = WeakKeyDict{Any, Handle}()        # analysis.method (a mutable struct) -> device handle
handle(analysis) = HANDLES[analysis.method].ptr

check(rc) = rc == 0 ? nothing :
    throw(ErrorException(unsafe_string(ccall((:jg_last_error, lib), Cstring, ()))))

reim(z::AbstractVector{ComplexF64}) = collect(reinterpret(Float64, z))

const HipTag = Union{HIP, HIPOrthogonal}
const HipPowerFlow = AcPowerFlow{NewtonRaphson{HIP}}
const HipFastPowerFlow = AcPowerFlow{FastNewtonRaphson{HIP}}          # needs `FastNewtonRaphson{T <: Normal}` / `FastNewtonRaphsonModel{T <: Normal}` (INTEGRATION.md)
const HipAnyPowerFlow = Union{HipPowerFlow, HipFastPowerFlow}
const HipStateEstimation = Union{AcStateEstimation{GaussNewton{HIP}}, AcStateEstimation{GaussNewton{HIPOrthogonal}}}
const HipPmuStateEstimation = Union{PmuStateEstimation{WLS{HIP}}, PmuStateEstimation{WLS{HIPOrthogonal}}}

# ------------------------------------------------------------------------------------------------------------------
# Newton-Raphson AC power flow
# ------------------------------------------------------------------------------------------------------------------
function pushInjection!(analysis::HipAnyPowerFlow)
    bus = analysis.system.bus
    p = bus.supply.active .- bus.demand.active
    q = bus.supply.reactive .- bus.demand.reactive
    check(ccall((:jg_nr_set_injection, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), handle(analysis), p, q, 0))
end

pushVoltage!(analysis::HipAnyPowerFlow) =
    check(ccall((:jg_nr_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
        handle(analysis), analysis.voltage.magnitude, analysis.voltage.angle, 0))

pullVoltage!(analysis::HipAnyPowerFlow) =
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        handle(analysis), analysis.voltage.magnitude, analysis.voltage.angle))

function pushYbus!(analysis::HipAnyPowerFlow)
    ac = analysis.system.model.ac
    check(ccall((:jg_nr_set_ybus, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        handle(analysis), reim(ac.nodalMatrix.nzval), reim(ac.nodalMatrixTranspose.nzval)))
end

"""
    newtonRaphson(system, HIP; device = 0)

Same set-up as `newtonRaphson(system, LU)` (bus-type normalisation, start voltages, index maps: done by the reference itself),
with the Jacobian pattern, the symbolic analysis and all per-iteration numerics on the GPU.  One scenario, like every
analysis of the reference; batches go through `NewtonRaphsonBatch`.
"""
function newtonRaphson(system::PowerSystem, ::Type{HIP}; device::Int = 0)
    base = newtonRaphson(system, LU)                          # host bookkeeping of the reference: maps, containers, start point
    m = base.method
    method = NewtonRaphson{HIP}(m.jacobian, m.mismatch, m.increment, JuliaGrid.selectFactorization(HIP), m.pq, m.pvpq, m.pcount,
        m.signature, 0)
    analysis = AcPowerFlow(base.voltage, base.power, base.current, method, system)
    ac = system.model.ac
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_nr_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}, Int64, Int64, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval,
        reim(ac.nodalMatrix.nzval), reim(ac.nodalMatrixTranspose.nzval),
        system.bus.layout.type, system.bus.layout.slack, 1, 0, device))
    HANDLES[method] = Handle(h[], :nr)
    pushInjection!(analysis)
    pushVoltage!(analysis)
    return analysis
end

function staleCheck(analysis::HipAnyPowerFlow)                  # acPowerFlow.jl:802-811
    rev, sig = analysis.system.model.revision, analysis.method.signature
    (rev.topology != sig.topology || rev.type != sig.type) && JuliaGrid.errorTypeConversion()
    if sig.acPattern != -1 && rev.acPattern != sig.acPattern  # addBranch!/dropZeros! changed the Ybus pattern: new handle
        throw(ErrorException("The Ybus pattern changed; build a new analysis with newtonRaphson(system, HIP)."))
    end
    sig.acPattern = rev.acPattern
end

function mismatch!(analysis::HipPowerFlow)                   # acPowerFlow.jl:645-685
    maxp = Ref(0.0); maxq = Ref(0.0)
    check(ccall((:jg_nr_mismatch, lib), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), handle(analysis), maxp, maxq))
    check(ccall((:jg_nr_get_mismatch, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.mismatch))
    return maxp[], maxq[]
end

function solve!(analysis::HipPowerFlow)                      # acPowerFlow.jl:793-911
    staleCheck(analysis)
    check(ccall((:jg_nr_solve, lib), Cint, (Ptr{Cvoid},), handle(analysis)))     # code 3 = singular Jacobian -> ErrorException
    check(ccall((:jg_nr_get_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.increment))
    pullVoltage!(analysis)
    analysis.method.iteration += 1
    return nothing
end

"method.jacobian at the current state, in the reference's CSC order (only when somebody looks: it stays on the device)"
function jacobian!(analysis::HipPowerFlow)
    check(ccall((:jg_nr_get_jacobian, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.jacobian.nzval))
    return analysis.method.jacobian
end

"""
    powerFlow!(analysis::AcPowerFlow{NewtonRaphson{HIP}}; iteration, tolerance, power, current)

The reference's loop (acPowerFlow.jl:1389-1433) with the same accounting, run on the device without a host round trip per
iteration.  The generic `powerFlow!` of the reference also works on this analysis (it only calls `mismatch!` / `solve!`).
"""
function powerFlow!(analysis::HipPowerFlow; iteration::Int64 = 20, tolerance::Float64 = 1e-8, power::Bool = false,
                    current::Bool = false, verbose::Int64 = 0)
    staleCheck(analysis)
    iters = Vector{Int32}(undef, 1); status = Vector{Int32}(undef, 1)
    check(ccall((:jg_nr_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}),
        handle(analysis), iteration, tolerance, iters, status))
    pullVoltage!(analysis)
    check(ccall((:jg_nr_get_mismatch, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.mismatch))
    check(ccall((:jg_nr_get_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.increment))
    analysis.method.iteration = iters[1]
    status[1] == 3 && throw(ErrorException("The Jacobian is singular."))        # SingularException of the reference's lu!
    power && power!(analysis)
    current && current!(analysis)
    return nothing
end

# ---- fast Newton-Raphson BX / XB (acPowerFlow.jl:215-537): the two constant matrices go to the device as values per stored Ybus entry and
# ---- are factorised there once; an iteration is two forward / backward sweeps on the device
function fastModel(system::PowerSystem, bx::Bool, device::Int)
    base = bx ? fastNewtonRaphsonBX(system, LU) : fastNewtonRaphsonXB(system, LU)     # B', B'', numbering, start point: the reference's own set-up
    m = base.method
    act = FastNewtonRaphsonModel{HIP}(m.active.jacobian, m.active.mismatch, m.active.increment, JuliaGrid.selectFactorization(HIP))
    rea = FastNewtonRaphsonModel{HIP}(m.reactive.jacobian, m.reactive.mismatch, m.reactive.increment, JuliaGrid.selectFactorization(HIP))
    method = FastNewtonRaphson{HIP}(act, rea, m.pq, m.pvpq, m.signature, bx, 0)
    analysis = AcPowerFlow(base.voltage, base.power, base.current, method, system)
    ac, bus = system.model.ac, system.bus
    Y = ac.nodalMatrix
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_nr_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}, Int64, Int64, Int64, Cint),
        h, bus.number, Y.colptr, Y.rowval, reim(Y.nzval), reim(ac.nodalMatrixTranspose.nzval), bus.layout.type, bus.layout.slack, 1, 0, device))
    HANDLES[method] = Handle(h[], :nr)
    pushFast!(analysis)
    pushInjection!(analysis)
    pushVoltage!(analysis)
    return analysis
end

"B'[pvpq r, pvpq c] and B''[pq r, pq c] at every stored Ybus entry (r, c); identity on the diagonal / zero elsewhere outside the reduced matrices"
function pushFast!(analysis::HipFastPowerFlow)
    system, m = analysis.system, analysis.method
    Y, typ, slack = system.model.ac.nodalMatrix, system.bus.layout.type, system.bus.layout.slack
    bp = zeros(nnz(Y)); bq = zeros(nnz(Y))
    for c = 1:system.bus.number, p = Y.colptr[c]:(Y.colptr[c + 1] - 1)
        r = Y.rowval[p]
        bp[p] = (typ[r] != 3 && c != slack) ? m.active.jacobian[m.pvpq[r], m.pvpq[c]] : (r == c ? 1.0 : 0.0)
        bq[p] = (typ[r] == 1 && typ[c] == 1) ? m.reactive.jacobian[m.pq[r], m.pq[c]] : (r == c ? 1.0 : 0.0)
    end
    check(ccall((:jg_nr_fast_setup, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), handle(analysis), bp, bq))
end

fastNewtonRaphsonBX(system::PowerSystem, ::Type{HIP}; device::Int = 0) = fastModel(system, true, device)
fastNewtonRaphsonXB(system::PowerSystem, ::Type{HIP}; device::Int = 0) = fastModel(system, false, device)

function mismatch!(analysis::HipFastPowerFlow)               # acPowerFlow.jl:687-730
    m = analysis.method
    maxp = Ref(0.0); maxq = Ref(0.0)
    check(ccall((:jg_nr_fast_mismatch, lib), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), handle(analysis), maxp, maxq))
    both = Vector{Float64}(undef, length(m.active.mismatch) + length(m.reactive.mismatch))       # [active.mismatch | reactive.mismatch]
    check(ccall((:jg_nr_get_mismatch, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), both))
    m.active.mismatch .= view(both, 1:length(m.active.mismatch))
    m.reactive.mismatch .= view(both, (length(m.active.mismatch) + 1):length(both))
    return maxp[], maxq[]
end

function pullFastIncrement!(analysis::HipFastPowerFlow)
    m = analysis.method
    both = Vector{Float64}(undef, length(m.active.increment) + length(m.reactive.increment))     # [active.increment | reactive.increment]
    check(ccall((:jg_nr_fast_get_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), both))
    m.active.increment .= view(both, 1:length(m.active.increment))
    m.reactive.increment .= view(both, (length(m.active.increment) + 1):length(both))
end

function solve!(analysis::HipFastPowerFlow)                  # acPowerFlow.jl:913-983
    staleCheck(analysis)
    check(ccall((:jg_nr_fast_solve, lib), Cint, (Ptr{Cvoid},), handle(analysis)))
    pullFastIncrement!(analysis)
    pullVoltage!(analysis)
    analysis.method.iteration += 1
    return nothing
end

function powerFlow!(analysis::HipFastPowerFlow; iteration::Int64 = 20, tolerance::Float64 = 1e-8, power::Bool = false,
                    current::Bool = false, verbose::Int64 = 0)
    staleCheck(analysis)
    iters = Vector{Int32}(undef, 1); status = Vector{Int32}(undef, 1)
    check(ccall((:jg_nr_fast_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}),
        handle(analysis), iteration, tolerance, iters, status))
    pullVoltage!(analysis)
    pullFastIncrement!(analysis)
    analysis.method.iteration = iters[1]
    status[1] == 3 && throw(ErrorException("The Jacobian is singular."))
    power && power!(analysis)
    current && current!(analysis)
    return nothing
end

# ---- power!(analysis) / current!(analysis) (postprocessing/acAnalysis.jl:30-169, 672-704): the Ybus row walk (injections) and the branch formulas run on
# ---- the device at the state the handle holds; the O(n) bus / generator bookkeeping is the reference's own (its power! runs first, the device
# ---- quantities then replace what it computed for injections and branches)
const BRANCHES = WeakKeyDict{Any, Bool}()                    # analysis.method -> the branch table is on the device

function pushBranches!(analysis::HipAnyPowerFlow)
    get(BRANCHES, analysis.method, false) && return
    system = analysis.system
    ac, br = system.model.ac, system.branch
    nb = br.number
    param = zeros(16, nb)                                     # [nb][16] row-major (jgrid.h: jg_nr_set_branches)
    for k = 1:nb
        tij = (1.0 / br.parameter.turnsRatio[k]) * cis(-br.parameter.shiftAngle[k])      # acAnalysis.jl:846-851
        for (c, z) in enumerate((ac.nodalFromFrom[k], ac.nodalFromTo[k], ac.nodalToFrom[k], ac.nodalToTo[k], ac.admittance[k], tij))
            param[2c - 1, k] = real(z); param[2c, k] = imag(z)
        end
        param[13, k] = br.parameter.conductance[k]; param[14, k] = br.parameter.susceptance[k]; param[15, k] = 1.0 / br.parameter.turnsRatio[k]
    end
    check(ccall((:jg_nr_set_branches, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int8}, Ptr{Float64}),
        handle(analysis), nb, br.layout.from, br.layout.to, Int8.(br.layout.status), param))
    BRANCHES[analysis.method] = true
end

function power!(analysis::HipAnyPowerFlow)
    invoke(power!, Tuple{AcPowerFlow}, analysis)              # containers, shunt, supply, generator allocation: the reference's bookkeeping
    pushBranches!(analysis)
    n, nb, pw = analysis.system.bus.number, analysis.system.branch.number, analysis.power
    inj = Matrix{Float64}(undef, 2, n)                        # [n][2] row-major
    check(ccall((:jg_nr_bus_injection, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), inj))
    pw.injection.active .= view(inj, 1, :); pw.injection.reactive .= view(inj, 2, :)
    from = Matrix{Float64}(undef, 2, nb); to = similar(from); series = similar(from); charging = similar(from)
    check(ccall((:jg_nr_branch_quantities, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        handle(analysis), from, to, series, charging, C_NULL, C_NULL, C_NULL))
    for (dst, src) in ((pw.from, from), (pw.to, to), (pw.series, series), (pw.charging, charging))
        dst.active .= view(src, 1, :); dst.reactive .= view(src, 2, :)
    end
    return nothing
end

function current!(analysis::HipAnyPowerFlow)
    invoke(current!, Tuple{JuliaGrid.AC}, analysis)           # containers + injection currents (O(n) from the injections)
    pushBranches!(analysis)
    nb, cu = analysis.system.branch.number, analysis.current
    from = Matrix{Float64}(undef, 2, nb); to = similar(from); series = similar(from)
    check(ccall((:jg_nr_branch_quantities, lib), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        handle(analysis), C_NULL, C_NULL, C_NULL, C_NULL, from, to, series))
    for (dst, src) in ((cu.from, from), (cu.to, to), (cu.series, series))
        dst.magnitude .= view(src, 1, :); dst.angle .= view(src, 2, :)
    end
    return nothing
end

function setInitialPoint!(analysis::HipPowerFlow)            # acPowerFlow.jl:1226-1249
    invoke(setInitialPoint!, Tuple{AcPowerFlow}, analysis)
    pushVoltage!(analysis)
end

function setInitialPoint!(target::HipPowerFlow, source::JuliaGrid.AC)   # acPowerFlow.jl:1281-1295
    invoke(setInitialPoint!, Tuple{AcPowerFlow, JuliaGrid.AC}, target, source)
    pushVoltage!(target)
end

function updateBus!(analysis::HipPowerFlow; label, kwargs...)           # bus.jl:286-298
    invoke(updateBus!, Tuple{AcPowerFlow}, analysis; label, kwargs...)
    pushInjection!(analysis); pushYbus!(analysis)                       # demand / shunt edits
    pushVoltage!(analysis)                                              # magnitude / angle keywords
end

function updateBranch!(analysis::HipPowerFlow; label, kwargs...)        # branch.jl:453-459 (pattern kept: stored zeros, model.jl:70-71)
    invoke(updateBranch!, Tuple{JuliaGrid.PowerFlow}, analysis; label, kwargs...)
    pushYbus!(analysis)
    delete!(BRANCHES, analysis.method)                                  # the branch table of power! / current! goes up again on the next call
end

function updateGenerator!(analysis::HipPowerFlow; label, kwargs...)     # generator.jl:382-388
    invoke(updateGenerator!, Tuple{JuliaGrid.PowerFlow}, analysis; label, kwargs...)
    pushInjection!(analysis)
    pushVoltage!(analysis)
end

# ---- batched scenarios (no counterpart in the reference, which loops updateBranch!/powerFlow! per outage, SURVEY.md 3.5) ----
"""
    NewtonRaphsonBatch(system, batch; device = 0, maxPatch = 4)

`batch` scenarios of one grid that share the Ybus pattern: per-scenario injections, start points and up to `maxPatch` Ybus
edits (an outage = the 4 edits acNodalUpdate! makes, model.jl:93-101).  Results are [n, batch] matrices / [batch] vectors.
"""
mutable struct NewtonRaphsonBatch
    system::PowerSystem
    batch::Int
    handle::Handle
    magnitude::Matrix{Float64}
    angle::Matrix{Float64}
    iteration::Vector{Int32}
    status::Vector{Int32}                          # 0 converged, 1 iteration limit, 3 numeric failure (singular Jacobian / NaN)
    outage::Vector{Int64}                          # branch out of service per scenario (0: none), for branchQuantities
end

function NewtonRaphsonBatch(system::PowerSystem, batch::Int; device::Int = 0, maxPatch::Int = 4)
    base = newtonRaphson(system, LU)
    ac = system.model.ac
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_nr_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}, Int64, Int64, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval, reim(ac.nodalMatrix.nzval),
        reim(ac.nodalMatrixTranspose.nzval), system.bus.layout.type, system.bus.layout.slack, batch, maxPatch, device))
    b = NewtonRaphsonBatch(system, batch, Handle(h[], :nr), zeros(system.bus.number, batch), zeros(system.bus.number, batch),
        zeros(Int32, batch), zeros(Int32, batch), zeros(Int64, batch))
    bus = system.bus
    check(ccall((:jg_nr_set_injection, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), b.handle.ptr,
        bus.supply.active .- bus.demand.active, bus.supply.reactive .- bus.demand.reactive, 0))
    check(ccall((:jg_nr_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), b.handle.ptr,
        base.voltage.magnitude, base.voltage.angle, 0))
    return b
end

"""
    shareDevice!(batch, on = true)

Hint for a pipeline of several batches in flight on one GPU (`jg_nr_set_shared`): the top of the factorisation runs the kernel variant that
leaves room on a compute unit for the other batches' workgroups.  Results are bitwise unchanged.
"""
shareDevice!(b::NewtonRaphsonBatch, on::Bool = true) =
    check(ccall((:jg_nr_set_shared, lib), Cint, (Ptr{Cvoid}, Cint), b.handle.ptr, on ? 1 : 0))

"scenario s (1-based) = base grid with branch labels[s] out of service (0: base case); one upload for the whole batch"
function setOutages!(b::NewtonRaphsonBatch, labels::Vector{Int64})
    length(labels) == b.batch || throw(DimensionMismatch("one label per scenario"))
    ac, Y, lay = b.system.model.ac, b.system.model.ac.nodalMatrix, b.system.branch.layout
    ptr = zeros(Int64, 4, b.batch); dy = zeros(Float64, 2, 4, b.batch)
    position(r, c) = Y.colptr[c] + searchsortedfirst(view(Y.rowval, Y.colptr[c]:(Y.colptr[c + 1] - 1)), r) - 1
    for (s, k) in enumerate(labels)
        k == 0 && continue
        i, j = lay.from[k], lay.to[k]
        ptr[:, s] = [position(i, i), position(j, j), position(i, j), position(j, i)]
        for (m, v) in enumerate((ac.nodalFromFrom[k], ac.nodalToTo[k], ac.nodalFromTo[k], ac.nodalToFrom[k]))
            dy[1, m, s], dy[2, m, s] = -real(v), -imag(v)
        end
    end
    check(ccall((:jg_nr_patch_ybus_batch, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
        b.handle.ptr, 0, b.batch, 4, ptr, dy))
    b.outage .= labels
    return nothing
end

function powerFlow!(b::NewtonRaphsonBatch; iteration::Int64 = 20, tolerance::Float64 = 1e-8)
    check(ccall((:jg_nr_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}),
        b.handle.ptr, iteration, tolerance, b.iteration, b.status))              # [batch] outputs
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        b.handle.ptr, b.magnitude, b.angle))                                     # [batch][n] row-major = [n, batch] column-major
    return nothing
end

"""
    branchQuantities(batch; currents = false) -> (from, to, series, charging) | (from, to, series)

`power!` / `current!` of the branches for EVERY scenario of the batch at its current state (acAnalysis.jl:898-931), each `[2, nb, batch]`
(active | reactive resp. magnitude | angle); the branch that is out of service in a scenario (`setOutages!`) reads zero there.
"""
function branchQuantities(b::NewtonRaphsonBatch; currents::Bool = false)
    system = b.system
    ac, br = system.model.ac, system.branch
    nb = br.number
    param = zeros(16, nb)
    for k = 1:nb
        tij = (1.0 / br.parameter.turnsRatio[k]) * cis(-br.parameter.shiftAngle[k])
        for (c, z) in enumerate((ac.nodalFromFrom[k], ac.nodalFromTo[k], ac.nodalToFrom[k], ac.nodalToTo[k], ac.admittance[k], tij))
            param[2c - 1, k] = real(z); param[2c, k] = imag(z)
        end
        param[13, k] = br.parameter.conductance[k]; param[14, k] = br.parameter.susceptance[k]; param[15, k] = 1.0 / br.parameter.turnsRatio[k]
    end
    check(ccall((:jg_nr_set_branches, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int8}, Ptr{Float64}),
        b.handle.ptr, nb, br.layout.from, br.layout.to, Int8.(br.layout.status), param))
    check(ccall((:jg_nr_set_outage_labels, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), b.handle.ptr, b.outage))
    out = [Array{Float64}(undef, 2, nb, b.batch) for _ = 1:(currents ? 3 : 4)]
    if currents
        check(ccall((:jg_nr_branch_quantities, lib), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            b.handle.ptr, C_NULL, C_NULL, C_NULL, C_NULL, out[1], out[2], out[3]))
    else
        check(ccall((:jg_nr_branch_quantities, lib), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            b.handle.ptr, out[1], out[2], out[3], out[4], C_NULL, C_NULL, C_NULL))
    end
    return Tuple(out)
end

"""
    screenSummary(batch; rating = nothing, record = nothing) -> Matrix [10, batch]

Contingency screen summary on the device (jgrid.h: jg_nr_screen): per scenario the worst branch loading against `rating` (pu of apparent power per branch,
0 = no limit) and its branch, the largest apparent power at a branch end and its branch, the lowest / highest voltage magnitude and their buses,
iterations, status -- what the user loop of branch.jl:453-459 reads off power!(analysis) after every powerFlow!, reduced where the states are.
`record`: a DEVICE pointer for `10 * batch` doubles instead (the operand of `allgatherDevice` in a sharded screen).
"""
function screenSummary(b::NewtonRaphsonBatch; rating::Union{Nothing, Vector{Float64}} = nothing, record::Union{Nothing, Ptr{Float64}} = nothing)
    branchQuantities(b; currents = true)                      # uploads the branch table and the outage labels (and is cheap: three outputs)
    if rating === nothing
        check(ccall((:jg_nr_set_screen, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, C_NULL))
    else
        check(ccall((:jg_nr_set_screen, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, rating))
    end
    if record !== nothing
        check(ccall((:jg_nr_screen_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, record))
        return nothing
    end
    rec = Matrix{Float64}(undef, 10, b.batch)
    check(ccall((:jg_nr_screen, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, rec))
    return rec
end

# ---- straggler hand-off between batches (jgrid.h: jg_nr_run_defer ...): a pipeline of batches stops a batch once <= deferAt
# scenarios are active, moves them into a POOL batch that collects the stragglers of several batches, and finishes them together
# (what ContingencyPipeline(pool = ...) does on the Python side; a Julia driver runs batches from Threads.@spawn tasks).
"runs until at most `deferAt` (<= 64) scenarios are still active; returns their number (0: the batch is done).  Follow with
moveLanes!(pool, lane0, b) if any are left, then finish!(b)."
function powerFlowDefer!(b::NewtonRaphsonBatch; iteration::Int64 = 20, tolerance::Float64 = 1e-8, deferAt::Int64 = 64)
    left = Ref{Int32}(0)
    check(ccall((:jg_nr_run_defer, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Int64, Ref{Int32}),
        b.handle.ptr, iteration, tolerance, deferAt, left))
    return Int(left[])
end

"the still-active scenarios of the paused batch `src` continue in lanes lane0 + 1, ... of `pool`; returns their (1-based)
scenario numbers in `src`, in lane order; in `src` they end with status 4 (deferred)"
function moveLanes!(pool::NewtonRaphsonBatch, lane0::Int, src::NewtonRaphsonBatch)
    home = zeros(Int32, 64); count = Ref{Int32}(0)
    check(ccall((:jg_nr_move_lanes, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cvoid}, Ptr{Int32}, Ref{Int32}),
        pool.handle.ptr, lane0, src.handle.ptr, home, count))
    return Int.(home[1:count[]]) .+ 1
end

"ends a paused run: iteration / status of every scenario (4 = handed to a pool), voltages of the others"
function finish!(b::NewtonRaphsonBatch)
    check(ccall((:jg_nr_finish, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), b.handle.ptr, b.iteration, b.status))
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), b.handle.ptr, b.magnitude, b.angle))
    return nothing
end

"runs the scenarios in lanes 1:lanes of a pool to the end, each with the iteration count it arrived with"
function resume!(pool::NewtonRaphsonBatch, lanes::Int; iteration::Int64 = 20, tolerance::Float64 = 1e-8)
    check(ccall((:jg_nr_resume, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Float64, Ptr{Int32}, Ptr{Int32}),
        pool.handle.ptr, lanes, iteration, tolerance, pool.iteration, pool.status))    # first `lanes` entries are written
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), pool.handle.ptr, pool.magnitude, pool.angle))
    return nothing
end

# ---- the first iteration of a screen on ONE shared factor (jgrid.h: jg_nr_base_*): the reference refactorises per scenario of its loop
# (branch.jl:453-459 + acPowerFlow.jl:890-897); batches whose scenarios all start from one base case pay one factorisation per BASE CASE
"""
    BaseCase(system; device = 0, topCap = 0, iteration = 20, tolerance = 1e-8)

Solves the power flow of `system` (one instance on the device), then factorises the Jacobian at that solution ONCE and keeps what the batches
attached to it need for a compensated first iteration (J0^-1 on the Ybus pattern, J0^-1 f0, the dense inverse of the top of the elimination tree:
`topCap` pivots at most, 0 = default).  The state it was built at is the start of every attached batch (`startFromBase!`).
"""
mutable struct BaseCase
    system::PowerSystem
    ptr::Ptr{Cvoid}
    function BaseCase(system::PowerSystem; device::Int = 0, topCap::Int = 0, iteration::Int64 = 20, tolerance::Float64 = 1e-8)
        single = NewtonRaphsonBatch(system, 1; device = device, maxPatch = 0)
        powerFlow!(single; iteration = iteration, tolerance = tolerance)
        single.status[1] == 0 || throw(ErrorException("BaseCase: the power flow of the base case did not converge"))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:jg_nr_base_create, lib), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Int64), h, single.handle.ptr, topCap))
        b = new(system, h[])
        finalizer(x -> (x.ptr == C_NULL || ccall((:jg_nr_base_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), b)
        return b
    end
end

"(pivots in the dense top, split level, forward / backward level launches of a sweep pair, the same two without a top, creation time in microseconds, attached batches)"
function baseInfo(base::BaseCase)
    info = zeros(Int64, 8)
    check(ccall((:jg_nr_base_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), base.ptr, info))
    return info
end

"J0^-1 on the stored Ybus pattern, `[4, nnz]` (a 2 x 2 block (theta_i, V_i) x (P_j, Q_j) per stored entry, row-major, in row-CSR order of the pattern) -- test access"
function baseInverseOnPattern(base::BaseCase)
    nnz = length(base.system.model.ac.nodalMatrix.nzval)
    out = zeros(Float64, 4, nnz)
    check(ccall((:jg_nr_base_get, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Int64), base.ptr, 0, out, length(out)))
    return out
end

"the scenarios of `b` may start from `base` (same grid and device); `nothing` detaches"
function attach!(b::NewtonRaphsonBatch, base::Union{BaseCase, Nothing})
    check(ccall((:jg_nr_attach_base, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), b.handle.ptr, base === nothing ? C_NULL : base.ptr))
    return nothing
end

"every scenario of `b` starts from the attached base case's state (what setInitialPoint!(analysis, base) is per scenario of the reference's loop,
acPowerFlow.jl:1271-1295); the next powerFlow! / powerFlowDefer! takes its first iteration on the base's shared factor when it can (jgrid.h lists the conditions)"
function startFromBase!(b::NewtonRaphsonBatch)
    check(ccall((:jg_nr_start_from_base, lib), Cint, (Ptr{Cvoid},), b.handle.ptr))
    return nothing
end

"shared = false: the first iteration refactorises like every other (A/B switch)"
firstIteration!(b::NewtonRaphsonBatch, shared::Bool = true) =
    check(ccall((:jg_nr_set_first_iteration, lib), Cint, (Ptr{Cvoid}, Cint), b.handle.ptr, shared ? 1 : 0))

"(runs of `b` that started on the shared factor, runs that refactorised)"
function firstIterationCounts(b::NewtonRaphsonBatch)
    a = Ref{Int64}(0); r = Ref{Int64}(0)
    check(ccall((:jg_nr_first_iteration_counts, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}), b.handle.ptr, a, r))
    return Int(a[]), Int(r[])
end

# ------------------------------------------------------------------------------------------------------------------
# Gauss-Newton WLS state estimation
# ------------------------------------------------------------------------------------------------------------------
# Type code of every Jacobian row BEFORE status masking, in acWLS's row order (acStateEstimation.jl:135-235): the library
# rebuilds the H pattern from (code, index) and needs the code of a masked row as well (se.type holds status * code).
function typeCodes(monitoring::Measurement)
    volt, amp, watt, var, pmu = monitoring.voltmeter, monitoring.ammeter, monitoring.wattmeter, monitoring.varmeter, monitoring.pmu
    code = Int8[]; status = Int8[]; corr = Int64[]
    for i = 1:volt.number
        push!(code, 1); push!(status, volt.magnitude.status[i])
    end
    for i = 1:amp.number
        sq, from = amp.layout.square[i], amp.layout.from[i]
        push!(code, sq ? (from ? 4 : 5) : (from ? 2 : 3)); push!(status, amp.magnitude.status[i])
    end
    for i = 1:watt.number
        push!(code, watt.layout.bus[i] ? 6 : (watt.layout.from[i] ? 7 : 8)); push!(status, watt.active.status[i])
    end
    for i = 1:var.number
        push!(code, var.layout.bus[i] ? 9 : (var.layout.from[i] ? 10 : 11)); push!(status, var.reactive.status[i])
    end
    for i = 1:pmu.number
        sm, sa, from = pmu.magnitude.status[i], pmu.angle.status[i], pmu.layout.from[i]
        if pmu.layout.polar[i]
            if pmu.layout.bus[i]
                append!(code, (12, 13))
            else
                append!(code, (pmu.layout.square[i] ? (from ? 4 : 5) : (from ? 2 : 3), from ? 14 : 15))
            end
            append!(status, (sm, sa))
        else
            pmu.layout.correlated[i] && push!(corr, length(code) + 1)
            append!(code, pmu.layout.bus[i] ? (16, 17) : (from ? (18, 20) : (19, 21)))
            append!(status, (sm * sa, sm * sa))
        end
    end
    return code, status, corr
end

function pushMeasurement!(analysis::HipStateEstimation)       # se.mean, se.precision (diagonal + W[r, r+1] of correlated pairs)
    se = analysis.method
    _, _, corr = typeCodes(analysis.monitoring)
    W = se.precision
    wdiag = [W[r, r] for r = 1:length(se.mean)]
    woff = Float64[W[r, r + 1] for r in corr]
    check(ccall((:jg_gn_set_measurement, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64),
        handle(analysis), se.mean, wdiag, isempty(woff) ? [0.0] : woff, 0, 0))
end

function pushStatus!(analysis::HipStateEstimation)            # update<Meter>!(...; status, square): se.type = status * code
    code, status, _ = typeCodes(analysis.monitoring)
    check(ccall((:jg_gn_set_status, lib), Cint, (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int8}), handle(analysis), status, code))
end

pushVoltage!(analysis::HipStateEstimation) =
    check(ccall((:jg_gn_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
        handle(analysis), analysis.voltage.magnitude, analysis.voltage.angle, 0))

pullVoltage!(analysis::HipStateEstimation) =
    check(ccall((:jg_gn_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        handle(analysis), analysis.voltage.magnitude, analysis.voltage.angle))

"""
    gaussNewton(monitoring, HIP; device = 0)

`gaussNewton(monitoring, LU)` of the reference builds the model (acWLS: type, index, range, mean, precision, Jacobian pattern);
the device gets what acWLS derived, row by row, and rebuilds pattern, gain pattern and symbolic analysis from it.
"""
function gaussNewton(monitoring::Measurement, ::Type{T}; device::Int = 0) where {T <: HipTag}
    base = gaussNewton(monitoring, LU)
    m = base.method
    method = GaussNewton{T}(m.jacobian, m.precision, m.mean, m.residual, m.increment, JuliaGrid.selectFactorization(T),
        m.type, m.index, m.range, m.signature, 0.0, 0)
    system = monitoring.system
    analysis = AcStateEstimation(base.voltage, base.power, base.current, method, system, monitoring)
    ac, br = system.model.ac, system.branch
    code, status, corr = typeCodes(monitoring)
    param = Matrix{Float64}(undef, 6, br.number)                          # [nb][6] row-major
    for k = 1:br.number
        param[:, k] .= (real(ac.admittance[k]), imag(ac.admittance[k]), br.parameter.conductance[k],
                        br.parameter.susceptance[k], br.parameter.turnsRatio[k], br.parameter.shiftAngle[k])
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_gn_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
         Int64, Int64, Ptr{Int8}, Ptr{Int8}, Ptr{Int64}, Int64, Ptr{Int64}, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval, reim(ac.nodalMatrix.nzval),
        reim(ac.nodalMatrixTranspose.nzval), br.number, br.layout.from, br.layout.to, param,
        system.bus.layout.slack, length(code), code, status, m.index, length(corr), isempty(corr) ? Int64[0] : corr, 1, device))
    HANDLES[method] = Handle(h[], :gn)
    if T === HIPOrthogonal                                                # the Orthogonal / PetersWilkinson rows: corrected semi-normal equations;
        check(ccall((:jg_gn_set_method, lib), Cint, (Ptr{Cvoid}, Cint), h[], 1))   # returns 1 (-> ErrorException) on correlated PMUs, like the reference
    end
    pushMeasurement!(analysis)
    pushVoltage!(analysis)
    return analysis
end

"se.residual from the device and se.objective = r' W r (equations.jl:689-698: what chiTest reads, badData.jl:948-961)"
function pullResidual!(analysis::HipStateEstimation)
    se = analysis.method
    check(ccall((:jg_gn_get_residual, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), se.residual))
    se.objective = transpose(se.residual) * (se.precision * se.residual)
    return nothing
end

function increment!(analysis::HipStateEstimation)             # acStateEstimation.jl:878-904 (HIPOrthogonal: :906-971)
    se = analysis.method
    maxInc = Vector{Float64}(undef, 1)
    check(ccall((:jg_gn_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), maxInc))   # code 3: singular gain
    check(ccall((:jg_gn_get_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), se.increment))
    pullResidual!(analysis)
    se.signature[:pattern] = 0
    return maxInc[1]
end

function solve!(analysis::HipStateEstimation)                 # acStateEstimation.jl:1035-1047
    check(ccall((:jg_gn_solve, lib), Cint, (Ptr{Cvoid},), handle(analysis)))
    pullVoltage!(analysis)
    analysis.method.iteration += 1
    return nothing
end

"method.jacobian (H, m x 2n) at the current state in the reference's CSC order, on request"
function jacobian!(analysis::HipStateEstimation)
    check(ccall((:jg_gn_get_jacobian, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), analysis.method.jacobian.nzval))
    return analysis.method.jacobian
end

"""
    stateEstimation!(analysis::AcStateEstimation{GaussNewton{HIP}}; iteration, tolerance, power, current)

The reference's loop (acStateEstimation.jl:1286-1329) fused on the device; the generic `stateEstimation!` works as well.
"""
function stateEstimation!(analysis::HipStateEstimation; iteration::Int64 = 40, tolerance::Float64 = 1e-8, power::Bool = false,
                          current::Bool = false, verbose::Int64 = 0)
    se = analysis.method
    iters = Vector{Int32}(undef, 1); status = Vector{Int32}(undef, 1)
    check(ccall((:jg_gn_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}),
        handle(analysis), iteration, tolerance, iters, status))
    pullVoltage!(analysis)
    check(ccall((:jg_gn_get_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), se.increment))
    pullResidual!(analysis)
    se.iteration = iters[1]
    se.signature[:pattern] = 0
    status[1] == 3 && throw(ErrorException("The gain matrix is singular."))
    power && JuliaGrid.power!(analysis)
    current && JuliaGrid.current!(analysis)
    return nothing
end

"""
    chiTest(analysis::AcStateEstimation{GaussNewton{HIP}}; confidence = 0.95)

badData.jl:948-961 reads `se.objective`; here the residual (and with it the objective) is first re-evaluated on the device AT THE CURRENT STATE
(`jg_gn_evaluate`: after the last `solve!` the device holds the residual of the state before that step), then the reference's own test runs.
"""
function chiTest(analysis::HipStateEstimation; confidence::Float64 = 0.95)
    check(ccall((:jg_gn_evaluate, lib), Cint, (Ptr{Cvoid},), handle(analysis)))
    pullResidual!(analysis)
    return invoke(chiTest, Tuple{AcStateEstimation{<:GaussNewton}}, analysis; confidence)
end

"all normalised residuals of the last largestNormalizedResidual / residualTest! call (badData.jl:289-311), one per Jacobian row"
function normalizedResiduals(analysis::HipStateEstimation)
    out = Vector{Float64}(undef, length(analysis.method.mean))
    check(ccall((:jg_gn_get_normalized_residual, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), out))
    return out
end

# ---- pmuStateEstimation(monitoring, HIP) (pmuStateEstimation.jl:43-177): the LINEAR model z = H [Re V; Im V] + u.  The device runs it as one
# ---- Gauss-Newton step of linear rows (type codes 22-27, no slack) from the zero state: gain H' W H, block LU, solve (:369-399)
function pmuStateEstimation(monitoring::Measurement, ::Type{T}; device::Int = 0) where {T <: HipTag}
    base = pmuStateEstimation(monitoring, LU)                  # coefficient, precision, mean, inservice: the reference's pmuEstimationWls
    m = base.method
    method = WLS{T}(m.coefficient, m.precision, m.mean, JuliaGrid.selectFactorization(T), m.index, m.number, m.inservice, m.signature)
    system, pmu = monitoring.system, monitoring.pmu
    analysis = PmuStateEstimation(base.voltage, base.power, base.current, method, system, monitoring)
    ac, br = system.model.ac, system.branch
    code = Int8[]; status = Int8[]; index = Int64[]; corr = Int64[]
    for i = 1:pmu.number                                       # two rows (Re, Im) per device, in device order
        pmu.layout.correlated[i] && push!(corr, length(code) + 1)
        append!(code, pmu.layout.bus[i] ? (22, 23) : (pmu.layout.from[i] ? (24, 25) : (26, 27)))
        on = Int8(pmu.magnitude.status[i] * pmu.angle.status[i])
        append!(status, (on, on)); append!(index, (pmu.layout.index[i], pmu.layout.index[i]))
    end
    T === HIPOrthogonal && !isempty(corr) && throw(ErrorException("A non-diagonal precision matrix prevents the use of the select method."))
    param = Matrix{Float64}(undef, 6, br.number)
    for k = 1:br.number
        param[:, k] .= (real(ac.admittance[k]), imag(ac.admittance[k]), br.parameter.conductance[k],
                        br.parameter.susceptance[k], br.parameter.turnsRatio[k], br.parameter.shiftAngle[k])
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_gn_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
         Int64, Int64, Ptr{Int8}, Ptr{Int8}, Ptr{Int64}, Int64, Ptr{Int64}, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval, reim(ac.nodalMatrix.nzval),
        reim(ac.nodalMatrixTranspose.nzval), br.number, br.layout.from, br.layout.to, param,
        0, length(code), code, status, index, length(corr), isempty(corr) ? Int64[0] : corr, 1, device))     # slack = 0: every variable is estimated
    HANDLES[method] = Handle(h[], :gn)
    T === HIPOrthogonal && check(ccall((:jg_gn_set_method, lib), Cint, (Ptr{Cvoid}, Cint), h[], 1))
    W = method.precision
    wdiag = [W[r, r] for r = 1:length(method.mean)]
    woff = Float64[W[r, r + 1] for r in corr]
    check(ccall((:jg_gn_set_measurement, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64),
        h[], method.mean, wdiag, isempty(woff) ? [0.0] : woff, 0, 0))
    return analysis
end

function solve!(analysis::HipPmuStateEstimation)              # pmuStateEstimation.jl:369-399
    n = analysis.system.bus.number
    zero = zeros(n)
    check(ccall((:jg_gn_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), handle(analysis), zero, zero, 0))
    maxInc = Vector{Float64}(undef, 1)
    check(ccall((:jg_gn_increment, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), handle(analysis), maxInc))    # residual = z at the zero state: H'WH, H'Wz
    check(ccall((:jg_gn_solve, lib), Cint, (Ptr{Cvoid},), handle(analysis)))
    im = Vector{Float64}(undef, n); re = Vector{Float64}(undef, n)
    check(ccall((:jg_gn_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), handle(analysis), im, re))   # the state arrays hold (Im V, Re V)
    resize!(analysis.voltage.magnitude, n); resize!(analysis.voltage.angle, n)
    analysis.voltage.magnitude .= hypot.(re, im)
    analysis.voltage.angle .= atan.(im, re)
    return nothing
end

function stateEstimation!(analysis::HipPmuStateEstimation; power::Bool = false, current::Bool = false, verbose::Int64 = 0)
    solve!(analysis)
    power && JuliaGrid.power!(analysis)
    current && JuliaGrid.current!(analysis)
    return nothing
end

function setInitialPoint!(analysis::HipStateEstimation)       # acStateEstimation.jl:1071-1077
    invoke(setInitialPoint!, Tuple{AcStateEstimation}, analysis)
    pushVoltage!(analysis)
end

# update<Meter>!(analysis; label, ...): the reference updates the monitoring container AND se.mean / se.precision / se.type of
# the analysis (measurement/*.jl: _update<Meter>!); the new values and masks follow to the device.  The Jacobian pattern stays.
for (fn, T) in ((:updateVoltmeter!, :AcStateEstimation), (:updateAmmeter!, :AcStateEstimation), (:updateVarmeter!, :AcStateEstimation),
                (:updateWattmeter!, :(Union{AcStateEstimation, JuliaGrid.DcStateEstimation})), (:updatePmu!, :(JuliaGrid.StateEstimation)))
    @eval function $fn(analysis::HipStateEstimation; label, kwargs...)
        invoke($fn, Tuple{$T}, analysis; label, kwargs...)
        pushStatus!(analysis)
        pushMeasurement!(analysis)
    end
end

"""
    largestNormalizedResidual(analysis) -> (value, row)

Numeric part of residualTest! (badData.jl:181-311) on the device: gain and its factor at the current state, selected inverse on
the factor pattern, normalised residuals, arg-max.  The label / status bookkeeping follows from `row` exactly as in
badData.jl:225-300.
"""
function largestNormalizedResidual(analysis::HipStateEstimation)
    mx = Vector{Float64}(undef, 1); idx = Vector{Int32}(undef, 1)
    check(ccall((:jg_gn_residual_test, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}), handle(analysis), mx, idx))
    return mx[1], Int64(idx[1])
end

# ---- sharded contingency screen: the ONE collective behind the C ABI (jgrid.h: jg_comm_*, jg_nr_allgather_results) --------------
# One Julia process per GPU.  Rank 0 draws the communicator id, the HOST ships its 128 bytes to the other ranks (MPI.bcast, a file,
# a socket -- the library does not care), every rank creates its communicator (a collective), solves its contiguous block of the
# scenario list and receives the record of the whole screen with ONE ncclAllGather of RCCL over xGMI.
const COMM_ID_BYTES = 128

"id of a new communicator (rank 0 calls this and ships the bytes to every rank)"
function commUniqueId()
    id = Vector{UInt8}(undef, COMM_ID_BYTES)
    check(ccall((:jg_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    return id
end

mutable struct Comm
    ptr::Ptr{Cvoid}
    rank::Int
    world::Int
    device::Int
    function Comm(rank::Int, world::Int, id::Vector{UInt8}; device::Int = 0)
        length(id) == COMM_ID_BYTES || throw(DimensionMismatch("communicator id: $COMM_ID_BYTES bytes"))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:jg_comm_create, lib), Cint, (Ptr{Ptr{Cvoid}}, Int64, Int64, Ptr{UInt8}, Cint), h, rank, world, id, device))
        c = new(h[], rank, world, device)
        finalizer(x -> (x.ptr == C_NULL || ccall((:jg_comm_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ptr); x.ptr = C_NULL), c)
        return c
    end
end

"contiguous block of scenarios owned by `rank` of `world` (0-based rank; 1-based inclusive range): SURVEY 8(e)"
function shard(count::Int, rank::Int, world::Int)
    per, extra = divrem(count, world)
    lo = rank * per + min(rank, extra)
    return (lo + 1):(lo + per + (rank < extra ? 1 : 0))
end

"""
    contingencyAnalysis(system, labels, comm; iteration = 20, tolerance = 1e-8, record)

N-1 screen of `labels` (branch labels, 0 = base case) sharded over the ranks of `comm`: this rank solves `labels[shard(...)]` as one
batch on its GPU (every rank must hold the same number of scenarios: pad with 0), then ONE all-gather hands every rank the record of
the whole screen.  `record` is DEVICE memory of the caller for `world * batch` rows of `2n + 2` doubles (V | theta | iterations |
status per scenario, scenario order), e.g. an AMDGPU.jl `ROCArray{Float64}`; pass its pointer.  Replaces the reference's serial
user-level loop (src/powerSystem/branch.jl:453-459: updateBranch!(...; status = 0) -> powerFlow! -> updateBranch!(...; status = 1)).
"""
function contingencyAnalysis(system::PowerSystem, labels::Vector{Int64}, comm::Comm; iteration::Int64 = 20, tolerance::Float64 = 1e-8,
                             record::Ptr{Float64})
    mine = labels[shard(length(labels), comm.rank, comm.world)]
    b = NewtonRaphsonBatch(system, length(mine); device = comm.device)
    setOutages!(b, mine)
    check(ccall((:jg_nr_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}), b.handle.ptr, iteration, tolerance, b.iteration, b.status))
    check(ccall((:jg_nr_allgather_results, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, comm.ptr, record))
    return b
end

# ---- sharded Monte-Carlo state estimation (jgrid.h: jg_gn_pack_results_device, jg_gn_allgather_results) ---------------------------------
"""
    GaussNewtonBatch(monitoring, batch; device = 0)

`batch` noisy realisations of ONE measurement set (the reference draws a realisation inside `add<Meter>!(...; noise = true)`,
src/measurement/utility.jl:70-73, and estimates them one after the other, acStateEstimation.jl:1286-1329): same rows, type codes and Jacobian
pattern, per-realisation means and precisions.  Results are [n, batch] matrices / [batch] vectors.
"""
mutable struct GaussNewtonBatch
    monitoring::Measurement
    batch::Int
    handle::Handle
    rows::Int
    magnitude::Matrix{Float64}
    angle::Matrix{Float64}
    iteration::Vector{Int32}
    status::Vector{Int32}                          # 0 converged, 1 iteration limit, 3 singular gain matrix
    objective::Vector{Float64}                     # se.objective = r' W r per realisation (equations.jl:689-698)
    pairs::Int                                     # correlated rectangular PMUs: rows of `woff` in setRealisations!
end

function GaussNewtonBatch(monitoring::Measurement, batch::Int; device::Int = 0)
    base = gaussNewton(monitoring, LU)
    m = base.method
    system = monitoring.system
    ac, br = system.model.ac, system.branch
    code, status, corr = typeCodes(monitoring)
    param = Matrix{Float64}(undef, 6, br.number)
    for k = 1:br.number
        param[:, k] .= (real(ac.admittance[k]), imag(ac.admittance[k]), br.parameter.conductance[k],
                        br.parameter.susceptance[k], br.parameter.turnsRatio[k], br.parameter.shiftAngle[k])
    end
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_gn_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
         Int64, Int64, Ptr{Int8}, Ptr{Int8}, Ptr{Int64}, Int64, Ptr{Int64}, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval, reim(ac.nodalMatrix.nzval),
        reim(ac.nodalMatrixTranspose.nzval), br.number, br.layout.from, br.layout.to, param,
        system.bus.layout.slack, length(code), code, status, m.index, length(corr), isempty(corr) ? Int64[0] : corr, batch, device))
    b = GaussNewtonBatch(monitoring, batch, Handle(h[], :gn), length(code), zeros(system.bus.number, batch), zeros(system.bus.number, batch),
        zeros(Int32, batch), zeros(Int32, batch), zeros(batch), length(corr))
    W = m.precision
    wdiag = [W[r, r] for r = 1:length(m.mean)]
    woff = Float64[W[r, r + 1] for r in corr]
    check(ccall((:jg_gn_set_measurement, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64),
        b.handle.ptr, m.mean, wdiag, isempty(woff) ? [0.0] : woff, 0, 0))
    check(ccall((:jg_gn_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), b.handle.ptr,
        base.voltage.magnitude, base.voltage.angle, 0))
    return b
end

"se.mean and the diagonal (+ pair terms `woff` [pairs, batch]) of se.precision per realisation: [rows, batch] each (acStateEstimation.jl:135-236)"
function setRealisations!(b::GaussNewtonBatch, mean::Matrix{Float64}, wdiag::Matrix{Float64}, woff::Matrix{Float64} = zeros(max(b.pairs, 1), b.batch))
    size(mean) == (b.rows, b.batch) && size(wdiag) == (b.rows, b.batch) || throw(DimensionMismatch("[rows, batch] means and precisions"))
    # (ADVICE r05) the pair terms of correlated rectangular PMUs travel per realisation: a short `woff` would be read past its end, a default of zeros would silently
    # drop the cross terms of the precision matrix
    b.pairs == 0 || size(woff) == (b.pairs, b.batch) || throw(DimensionMismatch("[correlated pairs, batch] off-diagonal precision terms: $(b.pairs) pair(s) in this model"))
    check(ccall((:jg_gn_set_measurement, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64),
        b.handle.ptr, mean, wdiag, woff, b.rows, size(woff, 1)))
end

"""
    setReadings!(b, row, kind, z1, v1, s1, z2, v2, s2);  drawNoise!(b, seed; scale = 1.0, first = 0)

Realisations drawn ON the device (jg_gn_set_readings / jg_gn_draw_noise): the raw readings per device once (first row 1-based, kind of acWLS value rule 0..5 --
include/jgrid.h --, mean / variance / status of the magnitude-type and of the angle-type quantity), then lane b of the batch becomes realisation `first + b` of `seed`:
what `add<Meter>!(...; noise = true)` draws (measurement/utility.jl:70-73) and acWLS makes of it (acStateEstimation.jl:135-236), without a byte over PCIe.
"""
setReadings!(b::GaussNewtonBatch, row::Vector{Int64}, kind::Vector{Int8}, z1::Vector{Float64}, v1::Vector{Float64}, s1::Vector{Int8},
             z2::Vector{Float64}, v2::Vector{Float64}, s2::Vector{Int8}) =
    check(ccall((:jg_gn_set_readings, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int8}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}),
        b.handle.ptr, length(row), row, kind, z1, v1, s1, z2, v2, s2))
drawNoise!(b::GaussNewtonBatch, seed::UInt64; scale::Float64 = 1.0, first::Int64 = 0) =
    check(ccall((:jg_gn_draw_noise, lib), Cint, (Ptr{Cvoid}, UInt64, Float64, Int64), b.handle.ptr, seed, scale, first))
"se.mean and diag(se.precision) of every realisation as the device holds them: [rows, batch] each (+ the pair terms [pairs, batch])"
function measurementDevice(b::GaussNewtonBatch, pairs::Int = 0)
    mean = Matrix{Float64}(undef, b.rows, b.batch); wdiag = similar(mean); woff = Matrix{Float64}(undef, max(pairs, 1), b.batch)
    check(ccall((:jg_gn_get_measurement, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.handle.ptr, mean, wdiag, woff))
    return mean, wdiag, woff
end

"stateEstimation!(analysis; iteration, tolerance) for every realisation of the batch (acStateEstimation.jl:1286-1329), then voltages and objectives"
function stateEstimation!(b::GaussNewtonBatch; iteration::Int64 = 40, tolerance::Float64 = 1e-8, fetch::Bool = true)
    check(ccall((:jg_gn_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ptr{Int32}, Ptr{Int32}), b.handle.ptr, iteration, tolerance, b.iteration, b.status))
    if fetch
        check(ccall((:jg_gn_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), b.handle.ptr, b.magnitude, b.angle))
        check(ccall((:jg_gn_get_objective, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, b.objective))
    end
    return nothing
end

"the batch's result record (magnitude | angle | iterations | status | objective per realisation, [2n + 3, batch]) into a DEVICE buffer of the caller"
packResults!(b::GaussNewtonBatch, record::Ptr{Float64}) =
    check(ccall((:jg_gn_pack_results_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, record))

"""
    monteCarloEstimation(monitoring, mean, wdiag, comm; iteration = 40, tolerance = 1e-8, record)

Noisy realisations (columns of `mean` / `wdiag`, [rows, realisations]) sharded over the ranks of `comm`: this rank estimates its contiguous block as one batch
(every rank the same count), then ONE all-gather hands every rank the record of all realisations: `record` = DEVICE memory for `world * batch` rows
of `2n + 3` doubles (magnitude | angle | iterations | status | objective).
"""
function monteCarloEstimation(monitoring::Measurement, mean::Matrix{Float64}, wdiag::Matrix{Float64}, comm::Comm; iteration::Int64 = 40,
                              tolerance::Float64 = 1e-8, record::Ptr{Float64})
    mine = shard(size(mean, 2), comm.rank, comm.world)
    b = GaussNewtonBatch(monitoring, length(mine); device = comm.device)
    setRealisations!(b, mean[:, mine], wdiag[:, mine])
    stateEstimation!(b; iteration, tolerance, fetch = false)
    check(ccall((:jg_gn_allgather_results, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, comm.ptr, record))
    return b
end

"drops the library's cached symbolic analyses (live analyses keep theirs): the next newtonRaphson(system, HIP) pays a full analysis again"
clearPlanCache() = ccall((:jg_plan_cache_clear, lib), Cvoid, ())

# ---- the rest of the ABI: sizes, maps, device-resident start points and records, the optional guard (include/jgrid.h) -------------------------
"number of HIP devices the library sees"
deviceCount() = Int(ccall((:jg_device_count, lib), Cint, ()))

"(dimJ, nnz(J), blocks of L + D + U, update terms, launches per factorisation, launches per backward sweep)"
function dims(analysis::HipAnyPowerFlow)
    d = Vector{Int64}(undef, 6)
    check(ccall((:jg_nr_dims, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), handle(analysis), d))
    return d
end
"(m, nnz(H), gain blocks, blocks of L + D + U, update terms, factor launches, backward launches, H slots)"
function dims(analysis::Union{HipStateEstimation, HipPmuStateEstimation})
    d = Vector{Int64}(undef, 8)
    check(ccall((:jg_gn_dims, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), handle(analysis), d))
    return d
end

"one step of iterative refinement behind every Newton step (what UMFPACK's solve does behind ldiv!, utility.jl:576-586); off by default"
setRefinement!(analysis::HipPowerFlow, on::Bool = true) =
    check(ccall((:jg_nr_set_refine, lib), Cint, (Ptr{Cvoid}, Cint), handle(analysis), on ? 1 : 0))

"pq, pvpq, pcount and the CSC pattern of the Jacobian as the DEVICE built them (bit-exact copies of newtonJacobian's, acPowerFlow.jl:89-175)"
function deviceMaps(analysis::HipPowerFlow)
    n, d = analysis.system.bus.number, dims(analysis)
    pq = Vector{Int64}(undef, n); pvpq = Vector{Int64}(undef, n); pcount = Vector{Int64}(undef, n)
    colptr = Vector{Int64}(undef, d[1] + 1); rowval = Vector{Int64}(undef, d[2])
    check(ccall((:jg_nr_get_maps, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
        handle(analysis), pq, pvpq, pcount, colptr, rowval))
    return pq, pvpq, pcount, colptr, rowval
end
"se.type and the CSC pattern of H as the device built them (bit-exact copies of acWLS's, acStateEstimation.jl:77-259)"
function deviceMaps(analysis::HipStateEstimation)
    d = dims(analysis)
    type = Vector{Int8}(undef, d[1]); colptr = Vector{Int64}(undef, 2 * analysis.system.bus.number + 1); rowval = Vector{Int64}(undef, d[2])
    check(ccall((:jg_gn_get_maps, lib), Cint, (Ptr{Cvoid}, Ptr{Int8}, Ptr{Int64}, Ptr{Int64}), handle(analysis), type, colptr, rowval))
    return type, colptr, rowval
end

"scenario s (1-based) of a batch: branch `label` out of service (0: back to the base grid) -- the four edits acNodalUpdate! makes, model.jl:93-101"
function setOutage!(b::NewtonRaphsonBatch, s::Int, label::Int64)
    ac, Y, lay = b.system.model.ac, b.system.model.ac.nodalMatrix, b.system.branch.layout
    position(r, c) = Y.colptr[c] + searchsortedfirst(view(Y.rowval, Y.colptr[c]:(Y.colptr[c + 1] - 1)), r) - 1
    ptr = zeros(Int64, 4); dy = zeros(Float64, 2, 4)
    if label != 0
        i, j = lay.from[label], lay.to[label]
        ptr .= (position(i, i), position(j, j), position(i, j), position(j, i))
        for (m, v) in enumerate((ac.nodalFromFrom[label], ac.nodalToTo[label], ac.nodalFromTo[label], ac.nodalToFrom[label]))
            dy[1, m], dy[2, m] = -real(v), -imag(v)
        end
    end
    check(ccall((:jg_nr_patch_ybus, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Float64}), b.handle.ptr, s - 1, label == 0 ? 0 : 4, ptr, dy))
    b.outage[s] = label
    return nothing
end

"""
    fastPatch!(batch, scenario0, ptr, dbp, dbq)

Fast Newton-Raphson under batched outages (jg_nr_fast_patch_batch): scenarios `scenario0 + 1 : scenario0 + size(ptr, 2)` keep the shared B', B'' of
`jg_nr_fast_setup` plus the edits `dbp`, `dbq` ([k, count], k <= 4) at the stored Ybus pointers `ptr` ([k, count], 1-based, 0 = unused); the batch is
factorised once.  What `_updateBranch!(::AcPowerFlow{<:FastNewtonRaphson})` (branch.jl:477) does entry by entry for one scenario at a time.
"""
fastPatch!(b::NewtonRaphsonBatch, scenario0::Int, ptr::Matrix{Int64}, dbp::Matrix{Float64}, dbq::Matrix{Float64}) =
    check(ccall((:jg_nr_fast_patch_batch, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
        b.handle.ptr, scenario0, size(ptr, 2), size(ptr, 1), ptr, dbp, dbq))

"keeps the current voltages of every scenario inside HBM / brings them back (the start point of a Monte-Carlo or benchmark loop without a PCIe round trip)"
snapshotVoltage!(b::NewtonRaphsonBatch) = check(ccall((:jg_nr_snapshot_voltage, lib), Cint, (Ptr{Cvoid},), b.handle.ptr))
restoreVoltage!(b::NewtonRaphsonBatch) = check(ccall((:jg_nr_restore_voltage, lib), Cint, (Ptr{Cvoid},), b.handle.ptr))
snapshotVoltage!(analysis::HipStateEstimation) = check(ccall((:jg_gn_snapshot_voltage, lib), Cint, (Ptr{Cvoid},), handle(analysis)))
restoreVoltage!(analysis::HipStateEstimation) = check(ccall((:jg_gn_restore_voltage, lib), Cint, (Ptr{Cvoid},), handle(analysis)))

"method.iteration of every scenario, straight from the device"
function iterations(b::NewtonRaphsonBatch)
    check(ccall((:jg_nr_get_iteration, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), b.handle.ptr, b.iteration))
    return b.iteration
end
function iterations(analysis::HipStateEstimation)
    it = Vector{Int32}(undef, 1)
    check(ccall((:jg_gn_get_iteration, lib), Cint, (Ptr{Cvoid}, Ptr{Int32}), handle(analysis), it))
    return Int(it[1])
end

"V and theta of every scenario into DEVICE buffers of the caller ([n, batch] each, e.g. AMDGPU.jl arrays: pass their pointers)"
voltageDevice!(b::NewtonRaphsonBatch, magnitude::Ptr{Float64}, angle::Ptr{Float64}) =
    check(ccall((:jg_nr_get_voltage_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), b.handle.ptr, magnitude, angle))
"the result record of the batch (V | theta | iterations | status per scenario, [2n + 2, batch]) into a DEVICE buffer of the caller: the operand of the one gather"
packResults!(b::NewtonRaphsonBatch, record::Ptr{Float64}) =
    check(ccall((:jg_nr_pack_results_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.handle.ptr, record))
"rows `rows` (0-based) of a record another batch owns <- lanes lane0 + 1 : lane0 + length(rows) of this pool (jg_nr_pack_rows_device)"
packRows!(pool::NewtonRaphsonBatch, record::Ptr{Float64}, lane0::Int, rows::Vector{Int32}) =
    check(ccall((:jg_nr_pack_rows_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Int32}), pool.handle.ptr, record, lane0, length(rows), rows))
"rows `rows` (0-based) of a screen-summary record ([10, .] doubles, screenSummary) another batch owns <- the summaries of lanes lane0 + 1 : lane0 + length(rows)
of this pool (jg_nr_screen_rows_device); the pool's branch table, ratings and outage labels must be set (screenSummary / setOutage! on the pool)"
screenRows!(pool::NewtonRaphsonBatch, record::Ptr{Float64}, lane0::Int, rows::Vector{Int32}) =
    check(ccall((:jg_nr_screen_rows_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Ptr{Int32}), pool.handle.ptr, record, lane0, length(rows), rows))

"the collective for a record that is already packed: `count` doubles per rank from device pointer `send` into `recv` [count, world]"
allgatherDevice(comm::Comm, send::Ptr{Float64}, recv::Ptr{Float64}, count::Int) =
    check(ccall((:jg_comm_allgather_device, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), comm.ptr, send, recv, count))
commRank(comm::Comm) = Int(ccall((:jg_comm_rank, lib), Cint, (Ptr{Cvoid},), comm.ptr))
commWorld(comm::Comm) = Int(ccall((:jg_comm_world, lib), Cint, (Ptr{Cvoid},), comm.ptr))

"mean milliseconds of `reps` executions of one kernel group on the handle's stream (HIP events): 0 assembly, 1 factorisation, 2 backward sweep, 3 branch post-processing"
function timeKernel(analysis::HipAnyPowerFlow, kernel::Int, reps::Int = 10)
    ms = Ref(0.0)
    check(ccall((:jg_nr_time_kernel, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Float64}), handle(analysis), kernel, reps, ms))
    return ms[]
end
"... of a state estimation: 0 measurement rows, 1 gain + rhs gather, 2 factorisation, 3 backward sweep, 4 selected inverse"
function timeKernel(analysis::HipStateEstimation, kernel::Int, reps::Int = 10)
    ms = Ref(0.0)
    check(ccall((:jg_gn_time_kernel, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ref{Float64}), handle(analysis), kernel, reps, ms))
    return ms[]
end

export HIP, HIPOrthogonal, NewtonRaphsonBatch, BaseCase, baseInfo, baseInverseOnPattern, attach!, startFromBase!, firstIteration!, firstIterationCounts, setOutages!, shareDevice!, branchQuantities, screenSummary, powerFlowDefer!, moveLanes!, finish!, resume!, jacobian!,
       largestNormalizedResidual, normalizedResiduals, commUniqueId, Comm, shard, contingencyAnalysis, clearPlanCache,
       deviceCount, dims, setRefinement!, deviceMaps, setOutage!, snapshotVoltage!, restoreVoltage!, iterations, voltageDevice!, packResults!, packRows!,
       allgatherDevice, commRank, commWorld, timeKernel, GaussNewtonBatch, setRealisations!, monteCarloEstimation, fastPatch!, setReadings!, drawNoise!, measurementDevice

end # module
