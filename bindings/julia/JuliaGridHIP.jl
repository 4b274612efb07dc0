# JuliaGridHIP.jl -- thin ccall shim that plugs libjgrid_hip.so (include/jgrid.h) behind JuliaGrid's own
# newtonRaphson()/mismatch!()/solve!()/powerFlow!() surface.  Written against JuliaGrid v0.6.2; it has
# NOT been executed (no Julia toolchain in the build containers) -- see INTEGRATION.md.
#
# Reference seams (paths relative to the JuliaGrid checkout):
#   tag types            src/definition/analysis.jl:36-99      (LU/KLU/QR <: Normal <: WlsMethod)
#   NewtonRaphson{T}     src/definition/analysis.jl:154-164    (field `factorization::FactorSparse`)
#   newtonRaphson        src/powerFlow/acPowerFlow.jl:39-87
#   mismatch!/solve!     src/powerFlow/acPowerFlow.jl:645-685, 793-911
#   powerFlow!           src/powerFlow/acPowerFlow.jl:1389-1433   (unchanged: it only calls mismatch!/solve!)
module JuliaGridHIP

using JuliaGrid
using SparseArrays

const lib = get(ENV, "JGRID_HIP_LIB", "libjgrid_hip.so")

struct HIP <: JuliaGrid.Normal end          # the new factorization tag

mutable struct HipHandle                    # stands in for `factorization` (widen FactorSparse, INTEGRATION.md)
    ptr::Ptr{Cvoid}
    function HipHandle(p)
        h = new(p)
        finalizer(x -> ccall((:jg_nr_destroy, lib), Cvoid, (Ptr{Cvoid},), x.ptr), h)
        return h
    end
end

check(rc) = rc == 0 ? nothing :
    throw(ErrorException(unsafe_string(ccall((:jg_last_error, lib), Cstring, ()))))

reim_interleaved(z::Vector{ComplexF64}) = collect(reinterpret(Float64, z))

"""
    newtonRaphson(system, HIP; batch = 1, device = 0)

Same set-up as `newtonRaphson(system, LU)` (bus-type normalisation, start voltages, index maps), with
the Jacobian pattern, symbolic analysis and all per-iteration numerics living on the GPU.
"""
function JuliaGrid.newtonRaphson(system::PowerSystem, ::Type{HIP}; batch::Int = 1, device::Int = 0)
    analysis = newtonRaphson(system, LU)                      # host bookkeeping, maps, containers
    ac = system.model.ac
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:jg_nr_create, lib), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int8}, Int64, Int64, Int64, Cint),
        h, system.bus.number, ac.nodalMatrix.colptr, ac.nodalMatrix.rowval,
        reim_interleaved(ac.nodalMatrix.nzval), reim_interleaved(ac.nodalMatrixTranspose.nzval),
        system.bus.layout.type, system.bus.layout.slack, batch, batch == 1 ? 0 : 4, device))
    handle = HipHandle(h[])
    p = system.bus.supply.active .- system.bus.demand.active
    q = system.bus.supply.reactive .- system.bus.demand.reactive
    check(ccall((:jg_nr_set_injection, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64), handle.ptr, p, q, 0))
    check(ccall((:jg_nr_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
        handle.ptr, analysis.voltage.magnitude, analysis.voltage.angle, 0))
    return HipPowerFlow(analysis, handle)
end

struct HipPowerFlow                          # AcPowerFlow{NewtonRaphson{HIP}} once the unions are widened
    base::AcPowerFlow
    handle::HipHandle
end

function JuliaGrid.mismatch!(a::HipPowerFlow)                # acPowerFlow.jl:645-685
    maxp = Ref(0.0); maxq = Ref(0.0)
    check(ccall((:jg_nr_mismatch, lib), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), a.handle.ptr, maxp, maxq))
    return maxp[], maxq[]
end

function JuliaGrid.solve!(a::HipPowerFlow)                   # acPowerFlow.jl:793-911
    rev, sig = a.base.system.model.revision, a.base.method.signature
    (rev.topology != sig.topology || rev.type != sig.type) && JuliaGrid.errorTypeConversion()
    check(ccall((:jg_nr_solve, lib), Cint, (Ptr{Cvoid},), a.handle.ptr))
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        a.handle.ptr, a.base.voltage.magnitude, a.base.voltage.angle))
    a.base.method.iteration += 1
    return nothing
end

function JuliaGrid.powerFlow!(a::HipPowerFlow; iteration::Int64 = 20, tolerance::Float64 = 1e-8)
    iters = Ref{Int32}(0); status = Ref{Int32}(0)            # acPowerFlow.jl:1389-1433, fused on the device
    check(ccall((:jg_nr_run, lib), Cint, (Ptr{Cvoid}, Int64, Float64, Ref{Int32}, Ref{Int32}),
        a.handle.ptr, iteration, tolerance, iters, status))
    check(ccall((:jg_nr_get_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        a.handle.ptr, a.base.voltage.magnitude, a.base.voltage.angle))
    a.base.method.iteration = iters[]
    return nothing
end

function JuliaGrid.setInitialPoint!(a::HipPowerFlow)         # acPowerFlow.jl:1226-1249
    setInitialPoint!(a.base)
    check(ccall((:jg_nr_set_voltage, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
        a.handle.ptr, a.base.voltage.magnitude, a.base.voltage.angle, 0))
end

function JuliaGrid.updateBranch!(a::HipPowerFlow; label, kwargs...)   # branch.jl:453-459
    updateBranch!(a.base; label, kwargs...)
    ac = a.base.system.model.ac
    check(ccall((:jg_nr_set_ybus, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
        a.handle.ptr, reim_interleaved(ac.nodalMatrix.nzval), reim_interleaved(ac.nodalMatrixTranspose.nzval)))
end

# Batched N-1 screening (no counterpart in the reference, which loops updateBranch!/powerFlow! per outage,
# branch.jl:453-459): scenario s of a batched analysis = base grid with branch labels[s] out of service, expressed as the
# 4 Ybus edits acNodalUpdate! would make (model.jl:93-101); one upload for the whole batch.
function setOutages!(a::HipPowerFlow, labels::Vector{Int64})
    ac, Y = a.base.system.model.ac, a.base.system.model.ac.nodalMatrix
    lay = a.base.system.branch.layout
    ptr = zeros(Int64, 4, length(labels)); dy = zeros(Float64, 2, 4, length(labels))
    position(r, c) = (p = searchsortedfirst(view(Y.rowval, Y.colptr[c]:(Y.colptr[c + 1] - 1)), r); Y.colptr[c] + p - 1)
    for (s, k) in enumerate(labels)
        k == 0 && continue
        i, j = lay.from[k], lay.to[k]
        ptr[:, s] = [position(i, i), position(j, j), position(i, j), position(j, i)]
        for (m, v) in enumerate((ac.nodalFromFrom[k], ac.nodalToTo[k], ac.nodalFromTo[k], ac.nodalToFrom[k]))
            dy[1, m, s], dy[2, m, s] = -real(v), -imag(v)
        end
    end
    check(ccall((:jg_nr_patch_ybus_batch, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Float64}),
        a.handle.ptr, 0, length(labels), 4, ptr, dy))
end

# residualTest!(analysis; threshold) -- badData.jl:181-311: the numeric part on the device (gain, its factor, the selected
# inverse on the factor pattern, normalised residuals, arg-max); returns (maxNormalizedResidual, index) and leaves the
# label / status bookkeeping of the reference untouched (it follows from `index` exactly as in badData.jl:225-300).
function largestNormalizedResidual(handle::HipHandle)
    mx = Ref(0.0); idx = Ref{Int32}(0)
    check(ccall((:jg_gn_residual_test, lib), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Int32}), handle.ptr, mx, idx))
    return mx[], Int64(idx[])
end

end # module
