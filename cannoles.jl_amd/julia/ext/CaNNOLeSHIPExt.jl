# CaNNOLeSHIPExt.jl — package EXTENSION of CaNNOLeS (file to be placed at CaNNOLeS.jl/ext/CaNNOLeSHIPExt.jl, with the
# [weakdeps] / [extensions] entries of INTEGRATION.md §3 in CaNNOLeS' Project.toml).  Julia loads it when CaNNOLeS and
# CaNNOLeSHIP are both present, so neither package imports the other at top level (no circular dependency).
# Shipped, NOT executed in this repository (no julia binary in the build image).
#
# It adds the fourth `LinearSolverStruct` backend next to MA57Struct / LDLFactStruct
# (/root/reference/src/solver_types.jl:17-98).  The four things a backend must provide
# (/root/reference/src/solver_types.jl:1-15,67): a constructor, get_vals, try_to_factorize, and solve_ldl! dispatching on the
# type of the `factor` field — plus the fused override of newton_system! (/root/reference/src/CaNNOLeS.jl:1008-1052).
module CaNNOLeSHIPExt

using CaNNOLeS, CaNNOLeSHIP
import CaNNOLeS: LinearSolverStruct, try_to_factorize, solve_ldl!, get_vals, newton_system!, linear_solver_struct

mutable struct HIPLDLStruct{Ti <: Integer} <: LinearSolverStruct
  rows::Vector{Ti}
  cols::Vector{Ti}
  vals::Vector{Float64}          # aliased by the solver: the driver mutates the rho slots (/root/reference/src/CaNNOLeS.jl:1027)
  factor::CaNNOLeSHIP.HIPFactor  # solve_ldl! dispatches on the type of this field
end

# Below this order of the KKT system the reference's own CPU backend is kept: one `newton_system!` through the device costs a fixed
# ~0.1 ms of launches and ~0.1 ms of transfers whatever the size (INTEGRATION.md §6), and the one-time symbolic analysis 0.1 ... 0.3 s at
# N = 2e4; LDLFactorizations needs microseconds for the reference's small test problems.  A `Ref` so that a caller can move it.
const MIN_ORDER = Ref(2000)

# the hook the if-chain of /root/reference/src/CaNNOLeS.jl:322-332 calls for `linsolve = :hipldl` (INTEGRATION.md §3)
function linear_solver_struct(::Val{:hipldl}, N, rows::Vector{Ti}, cols::Vector{Ti}, vals::Vector{Float64}, nvar, nequ, ncon) where {Ti}
  N < MIN_ORDER[] && return CaNNOLeS.LDLFactStruct(N, rows, cols, vals)   # small systems stay on the CPU backend (src/solver_types.jl:61-65)
  r64 = Ti === Int64 ? rows : Vector{Int64}(rows)
  c64 = Ti === Int64 ? cols : Vector{Int64}(cols)
  return HIPLDLStruct{Ti}(rows, cols, vals, CaNNOLeSHIP.HIPFactor(N, r64, c64, nvar, nequ, ncon))
end

get_vals(LDLT::HIPLDLStruct) = LDLT.vals

# /root/reference/src/solver_types.jl:79-98
try_to_factorize(LDLT::HIPLDLStruct, vals::Vector{Float64}, nvar::Integer, nequ::Integer, ncon::Integer, eig_tol::Real) =
  CaNNOLeSHIP.factorize!(LDLT.factor, vals, Float64(eig_tol))

# /root/reference/src/solver_types.jl:69-77
solve_ldl!(rhs::Vector{Float64}, factor::CaNNOLeSHIP.HIPFactor, d::Vector{Float64}) = CaNNOLeSHIP.solve!(factor, rhs, d)

# /root/reference/src/CaNNOLeS.jl:1008-1052, one device call; allocation-free: the parameter tuple is isbits, every
# out-parameter is a preallocated Ref of the factor object
function newton_system!(d::Vector{Float64}, nvar::Integer, nequ::Integer, ncon::Integer, rhs::Vector{Float64},
                        vals::Vector{Float64}, LDLT::HIPLDLStruct, ρold::Float64, params::CaNNOLeS.ParamCaNNOLeS{Float64})
  p = (params.eig_tol, params.δmin, params.κdec, params.κinc, params.κlargeinc, params.ρ0, params.ρmax, params.ρmin, params.γA)
  ok, ρ, ρout, nfact = CaNNOLeSHIP.newton_system!(LDLT.factor, get_vals(LDLT), rhs, d, ρold, p)
  return d, ok, ρ, ρout, nfact
end

end # module
