# CaNNOLeSHIP.jl — thin `ccall` layer over include/cannoles_hip.h (libcannoles_hip.so), the MI355X backend of the
# Newton-system path of CaNNOLeS.jl.  Shipped, NOT executed in this repository: the build image has no julia binary; the
# same C ABI is exercised by the Python/ctypes mirror (cannoles.jl_amd/hipldl.py) that the tests use.
#
# This package does NOT depend on CaNNOLeS (no circular dependency): it only wraps the C ABI.  The methods that plug it into
# CaNNOLeS' `LinearSolverStruct` surface live in the package extension `CaNNOLeSHIPExt` of CaNNOLeS itself
# (cannoles.jl_amd/julia/ext/CaNNOLeSHIPExt.jl, INTEGRATION.md §3), which Julia loads when both packages are present.
module CaNNOLeSHIP

const libcnl = get(ENV, "CANNOLES_HIP_LIB", "libcannoles_hip.so")

struct CnlError <: Exception
  code::Cint
  msg::String
end

@noinline function throw_error(rc::Cint)
  msg = unsafe_string(ccall((:cnl_last_error, libcnl), Cstring, ()))
  throw(CnlError(rc, msg))
end
@inline check(rc::Cint) = rc == 0 ? nothing : throw_error(rc)

"""
Native handle of one solver object (`cnl_handle*`) together with every out-parameter of the hot calls, preallocated, so that
`factorize!`, `solve!` and `newton_system!` allocate nothing (the reference asserts <= 96 B per `solve!`,
/root/reference/test/runtests.jl:28-36).  `solve_ldl!` of the extension dispatches on this type
(/root/reference/src/solver_types.jl:10-15).
"""
mutable struct HIPFactor
  handle::Ptr{Cvoid}
  N::Int
  success::Base.RefValue{Int32}
  npos::Base.RefValue{Int64}
  nzero::Base.RefValue{Int64}
  rho_in::Base.RefValue{Float64}
  rho::Base.RefValue{Float64}
  rho_out::Base.RefValue{Float64}
  nfact::Base.RefValue{Int32}
  params::Base.RefValue{NTuple{9, Float64}}
end

function destroy!(f::HIPFactor)
  if f.handle != C_NULL
    ccall((:cnl_destroy, libcnl), Cint, (Ptr{Cvoid},), f.handle)
    f.handle = C_NULL
  end
  return nothing
end

"""
    HIPFactor(N, rows, cols, nvar, nequ, ncon; device = 0)

`cnl_create` with batch = 1: the reference's one-solver-one-problem case (replaces `ldl_analyze`,
/root/reference/src/solver_types.jl:61-65).  `rows`/`cols`: 1-based lower-triangular COO of the KKT pattern in the
reference's 7-segment order (/root/reference/src/CaNNOLeS.jl:256-315).
"""
function HIPFactor(N::Integer, rows::Vector{Int64}, cols::Vector{Int64}, nvar::Integer, nequ::Integer, ncon::Integer; device::Integer = 0)
  # an EXPERIMENT build of the library (timing probes / diagnostic stamps compiled in) reports a negative version
  ccall((:cnl_version, libcnl), Int32, ()) >= 200 || error("libcannoles_hip: ABI version 0.2.0 or later required (experiment builds are refused)")
  h = Ref{Ptr{Cvoid}}(C_NULL)
  check(ccall((:cnl_create, libcnl), Cint,
    (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Int64, Int64, Cint),
    h, N, length(rows), rows, cols, nvar, nequ, ncon, 1, device))
  f = HIPFactor(h[], N, Ref(Int32(0)), Ref(Int64(0)), Ref(Int64(0)), Ref(0.0), Ref(0.0), Ref(0.0), Ref(Int32(0)),
                Ref(ntuple(_ -> 0.0, 9)))
  finalizer(destroy!, f)
  return f
end

"try_to_factorize — /root/reference/src/solver_types.jl:79-98"
function factorize!(f::HIPFactor, vals::Vector{Float64}, eig_tol::Float64)
  check(ccall((:cnl_factorize, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Float64, Ref{Int32}, Ref{Int64}, Ref{Int64}),
    f.handle, vals, eig_tol, f.success, f.npos, f.nzero))
  return f.success[] != 0
end

"""
solve_ldl! — /root/reference/src/solver_types.jl:69-77: d = -(K^-1 rhs), rhs untouched.  The reference calls it only after a
successful `try_to_factorize` (/root/reference/src/CaNNOLeS.jl:1049); after a failed one `cnl_solve` returns CNL_ERR_STATE and
this throws `CnlError` (nothing is written to `d`).
"""
function solve!(f::HIPFactor, rhs::Vector{Float64}, d::Vector{Float64})
  check(ccall((:cnl_solve, libcnl), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), f.handle, rhs, d))
  return true
end

"""
newton_system! — /root/reference/src/CaNNOLeS.jl:1008-1052, fused on the device (`cnl_newton_system`): `vals` crosses PCIe
once, the rho ladder runs on the GPU, the rho slots of `vals` are written back as the reference leaves them.
`params` = (eig_tol, δmin, κdec, κinc, κlargeinc, ρ0, ρmax, ρmin, γA).  Returns (success, ρ, ρold, nfact); allocation-free:
the tuple is isbits, the out-parameters are the preallocated Refs of `f`.
"""
function newton_system!(f::HIPFactor, vals::Vector{Float64}, rhs::Vector{Float64}, d::Vector{Float64}, ρold::Float64,
                        params::NTuple{9, Float64})
  f.rho_in[] = ρold
  f.params[] = params
  check(ccall((:cnl_newton_system, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}, Ref{NTuple{9, Float64}}, Ref{Float64}, Ref{Float64}, Ref{Int32}, Ref{Int32}),
    f.handle, vals, rhs, d, f.rho_in, f.params, f.rho, f.rho_out, f.nfact, f.success))
  return f.success[] != 0, f.rho[], f.rho_out[], Int(f.nfact[])
end

# ---- optional device-resident helpers (SURVEY rows a4/f1/f2/f4).  They take DEVICE pointers (e.g. `pointer(::ROCArray)` from
# AMDGPU.jl) and a hipStream_t; batched layouts are problem-major. ---------------------------------------------------------

"rhs = [Jx'r - Jc'λ; F - r; c] and (‖dual‖∞, ‖primal‖∞) per problem — /root/reference/src/CaNNOLeS.jl:507-508,519-524,528-529,631-632"
residual_vectors_dev!(f::HIPFactor, vals, r, λ, Fx, cx, rhs, norms; stream = C_NULL) =
  check(ccall((:cnl_residual_vectors_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    f.handle, vals, r, λ, Fx, cx, rhs, norms, stream))

"xt = x + dx, rt = r + dr, dλ = -d[n+m+1:N] capped at Mdλ, λt = λ + dλ — /root/reference/src/CaNNOLeS.jl:654,661-668"
trial_point_dev!(f::HIPFactor, x, r, λ, d, Mdλ, xt, rt, λt, dλ; stream = C_NULL) =
  check(ccall((:cnl_trial_point_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    f.handle, x, r, λ, d, Mdλ, xt, rt, λt, dλ, stream))

"prepare_newton_system! with model values that already live on the device — /root/reference/src/CaNNOLeS.jl:947-981"
prepare_newton_system_dev!(f::HIPFactor, nnzhF, nnzhc, nnzjF, nnzjc, hF, hc, Jx, Jcx, δ, vals; stream = C_NULL) =
  check(ccall((:cnl_prepare_newton_system_dev, libcnl), Cint,
    (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    f.handle, nnzhF, nnzhc, nnzjF, nnzjc, hF, hc, Jx, Jcx, δ, vals, stream))

"λ = argmin ‖Jc'λ − Jx'r‖ by CGLS (Krylov.jl defaults) — /root/reference/src/CaNNOLeS.jl:507-518"
cgls_multipliers_dev!(f::HIPFactor, vals, r, λ; Jxtr = C_NULL, atol = √eps(Float64), rtol = √eps(Float64), itmax = 0,
                      ones_if_zero = true, iters = C_NULL, stream = C_NULL) =
  check(ccall((:cnl_cgls_multipliers_dev, libcnl), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int64, Cint, Ptr{Int32}, Ptr{Cvoid}),
    f.handle, vals, r, λ, Jxtr, atol, rtol, itmax, ones_if_zero ? 1 : 0, iters, stream))

"the same as `residual_vectors_dev!` with the Jacobian values read from the model's arrays Jx [batch][nnzjF], Jcx [batch][nnzjc] (no prepare pass in front; any `vals` layout)"
residual_vectors_jac_dev!(f::HIPFactor, nnzjF, nnzjc, Jx, Jcx, r, λ, Fx, cx, rhs, norms; stream = C_NULL) =
  check(ccall((:cnl_residual_vectors_jac_dev, libcnl), Cint,
    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
    f.handle, nnzjF, nnzjc, Jx, Jcx, r, λ, Fx, cx, rhs, norms, stream))

"the same as `cgls_multipliers_dev!` with the Jacobian values read from the model's arrays"
cgls_multipliers_jac_dev!(f::HIPFactor, nnzjF, nnzjc, Jx, Jcx, r, λ; Jxtr = C_NULL, atol = √eps(Float64), rtol = √eps(Float64), itmax = 0,
                          ones_if_zero = true, iters = C_NULL, stream = C_NULL) =
  check(ccall((:cnl_cgls_multipliers_jac_dev, libcnl), Cint,
    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int64, Cint, Ptr{Int32}, Ptr{Cvoid}),
    f.handle, nnzjF, nnzjc, Jx, Jcx, r, λ, Jxtr, atol, rtol, itmax, ones_if_zero ? 1 : 0, iters, stream))

# ---- `vals` interleaved over groups of 32 problems (cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED, include/cannoles_hip.h): batched
# handles created through `cnl_create_ex` with that option take / write this layout at the device-pointer entry points --------------

"doubles of the interleaved array of the handle's batch; which = 0: vals, 1: an N-vector per problem"
function layout_len(f::HIPFactor, which::Integer)
  n = Ref{Int64}(0)
  check(ccall((:cnl_layout_len, libcnl), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}), f.handle, which, n))
  return n[]
end

"problem-major -> interleaved on the device, out of place"
interleave_dev!(f::HIPFactor, which::Integer, src, dst; stream = C_NULL) =
  check(ccall((:cnl_interleave_dev, libcnl), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), f.handle, which, src, dst, stream))

"interleaved -> problem-major on the device, out of place"
deinterleave_dev!(f::HIPFactor, which::Integer, src, dst; stream = C_NULL) =
  check(ccall((:cnl_deinterleave_dev, libcnl), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), f.handle, which, src, dst, stream))

end # module
