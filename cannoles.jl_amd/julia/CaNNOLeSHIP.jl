# CaNNOLeSHIP.jl — Julia glue for the MI355X backend (shipped, NOT executed in this repository:
# there is no julia binary in the build image; the same C ABI is exercised by the Python harness).
#
# It adds a fourth `LinearSolverStruct` backend next to MA57Struct / LDLFactStruct
# (CaNNOLeS.jl src/solver_types.jl:17-98) over include/cannoles_hip.h.  The four things a backend
# must provide (src/solver_types.jl:1-15,67) are: a constructor, get_vals, try_to_factorize and
# solve_ldl! dispatching on the type of the `factor` field.  See INTEGRATION.md for the two-line
# change in CaNNOLeS.jl (src/CaNNOLeS.jl:322-332) that selects it with `linsolve = :hipldl`.
module CaNNOLeSHIP

using CaNNOLeS
import CaNNOLeS: LinearSolverStruct, try_to_factorize, solve_ldl!, get_vals

const libcnl = get(ENV, "CANNOLES_HIP_LIB", "libcannoles_hip.so")

struct CnlError <: Exception
  code::Cint
  msg::String
end

function check(rc::Cint)
  rc == 0 && return
  msg = unsafe_string(ccall((:cnl_last_error, libcnl), Cstring, ()))
  throw(CnlError(rc, msg))
end

"Native handle; `solve_ldl!` dispatches on this type (src/solver_types.jl:10-15)."
mutable struct HIPFactor
  handle::Ptr{Cvoid}
  N::Int
  function HIPFactor(handle, N)
    f = new(handle, N)
    finalizer(f -> (f.handle != C_NULL && ccall((:cnl_destroy, libcnl), Cint, (Ptr{Cvoid},), f.handle); f.handle = C_NULL), f)
    return f
  end
end

mutable struct HIPLDLStruct{Ti <: Integer} <: LinearSolverStruct
  rows::Vector{Ti}
  cols::Vector{Ti}
  vals::Vector{Float64}      # aliased by the solver: the driver mutates the rho slots (src/CaNNOLeS.jl:1027)
  factor::HIPFactor
  nvar::Int
  nequ::Int
  ncon::Int
  # scalar out-parameters, preallocated so that the hot path does not allocate (test/runtests.jl:28-36)
  success::Base.RefValue{Int32}
  npos::Base.RefValue{Int64}
  nzero::Base.RefValue{Int64}
end

"""
    HIPLDLStruct(N, rows, cols, vals, nvar, nequ, ncon; device = 0)

Replaces `LDLFactStruct(N, rows, cols, vals)` (src/solver_types.jl:61-65): symbolic analysis on the
host, plan upload to the device.  Float64 only; other element types must keep using LDLFactStruct.
"""
function HIPLDLStruct(N, rows::Vector{Ti}, cols::Vector{Ti}, vals::Vector{Float64}, nvar, nequ, ncon; device = 0) where {Ti}
  h = Ref{Ptr{Cvoid}}(C_NULL)
  r64, c64 = Vector{Int64}(rows), Vector{Int64}(cols)
  check(ccall((:cnl_create, libcnl), Cint,
    (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Int64, Int64, Cint),
    h, N, length(rows), r64, c64, nvar, nequ, ncon, 1, device))
  HIPLDLStruct{Ti}(rows, cols, vals, HIPFactor(h[], N), nvar, nequ, ncon, Ref(Int32(0)), Ref(Int64(0)), Ref(Int64(0)))
end

get_vals(LDLT::HIPLDLStruct) = LDLT.vals

# src/solver_types.jl:79-98
function try_to_factorize(LDLT::HIPLDLStruct, vals::AbstractVector, nvar::Integer, nequ::Integer, ncon::Integer, eig_tol::Real)
  check(ccall((:cnl_factorize, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Float64, Ref{Int32}, Ref{Int64}, Ref{Int64}),
    LDLT.factor.handle, vals, eig_tol, LDLT.success, LDLT.npos, LDLT.nzero))
  return LDLT.success[] != 0
end

# src/solver_types.jl:69-77: d = -(K^-1 rhs); rhs untouched; returns true
function solve_ldl!(rhs::AbstractVector, factor::HIPFactor, d::AbstractVector)
  check(ccall((:cnl_solve, libcnl), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), factor.handle, rhs, d))
  return true
end

"""
    newton_system!(d, nvar, nequ, ncon, rhs, vals, LDLT::HIPLDLStruct, rho_old, params)

Optional fused override of src/CaNNOLeS.jl:1008-1052: the factorise / rho-ladder / solve sequence runs
in one device launch (`cnl_newton_system`), so `vals` crosses PCIe once instead of once per retry.
"""
function CaNNOLeS.newton_system!(d::AbstractVector{Float64}, nvar::Integer, nequ::Integer, ncon::Integer,
                                 rhs::AbstractVector{Float64}, vals::AbstractVector{Float64}, LDLT::HIPLDLStruct,
                                 ρold::Float64, params::CaNNOLeS.ParamCaNNOLeS{Float64})
  p = (params.eig_tol, params.δmin, params.κdec, params.κinc, params.κlargeinc, params.ρ0, params.ρmax, params.ρmin, params.γA)
  pv = Ref(p)
  ρ, ρout, nfact, ok = Ref(0.0), Ref(0.0), Ref(Int32(0)), Ref(Int32(0))
  ρin = Ref(ρold)
  check(ccall((:cnl_newton_system, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{NTuple{9, Float64}}, Ref{Float64}, Ref{Float64}, Ref{Int32}, Ref{Int32}),
    LDLT.factor.handle, get_vals(LDLT), rhs, d, ρin, pv, ρ, ρout, nfact, ok))
  return d, ok[] != 0, ρ[], ρout[], Int(nfact[])
end

# ---- optional device-resident helpers (SURVEY rows a4/f1/f2/f4).  They take DEVICE pointers (e.g. `pointer(::ROCArray)` from
# AMDGPU.jl) and a hipStream_t; batched layouts are problem-major.  Unexecuted here, like the rest of this file. ----------

"rhs = [Jx'r - Jc'λ; F - r; c] and (‖dual‖∞, ‖primal‖∞) per problem — src/CaNNOLeS.jl:507-508,519-524,528-529,631-632"
residual_vectors_dev!(LDLT::HIPLDLStruct, vals, r, λ, Fx, cx, rhs, norms; stream = C_NULL) =
  check(ccall((:cnl_residual_vectors_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    LDLT.factor.handle, vals, r, λ, Fx, cx, rhs, norms, stream))

"xt = x + dx, rt = r + dr, dλ = -d[n+m+1:N] capped at Mdλ, λt = λ + dλ — src/CaNNOLeS.jl:654,661-668"
trial_point_dev!(LDLT::HIPLDLStruct, x, r, λ, d, Mdλ, xt, rt, λt, dλ; stream = C_NULL) =
  check(ccall((:cnl_trial_point_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    LDLT.factor.handle, x, r, λ, d, Mdλ, xt, rt, λt, dλ, stream))

"prepare_newton_system! with model values that already live on the device — src/CaNNOLeS.jl:947-981"
prepare_newton_system_dev!(LDLT::HIPLDLStruct, nnzhF, nnzhc, nnzjF, nnzjc, hF, hc, Jx, Jcx, δ, vals; stream = C_NULL) =
  check(ccall((:cnl_prepare_newton_system_dev, libcnl), Cint,
    (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    LDLT.factor.handle, nnzhF, nnzhc, nnzjF, nnzjc, hF, hc, Jx, Jcx, δ, vals, stream))

"λ = argmin ‖Jc'λ − Jx'r‖ by CGLS (Krylov.jl defaults) — src/CaNNOLeS.jl:507-518"
cgls_multipliers_dev!(LDLT::HIPLDLStruct, vals, r, λ; Jxtr = C_NULL, atol = √eps(Float64), rtol = √eps(Float64), itmax = 0,
                      ones_if_zero = true, iters = C_NULL, stream = C_NULL) =
  check(ccall((:cnl_cgls_multipliers_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int64, Cint, Ptr{Int32}, Ptr{Cvoid}),
    LDLT.factor.handle, vals, r, λ, Jxtr, atol, rtol, itmax, ones_if_zero ? 1 : 0, iters, stream))

end # module
