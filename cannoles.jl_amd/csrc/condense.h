// condense.h — static condensation of the residual block of the KKT system.
//
// The Newton matrix of /root/reference/src/CaNNOLeS.jl:282 is
//     [ H + rho I   Jx'   Jc' ]
//     [ Jx          -I    0   ]
//     [ Jc          0   -dI   ]
// Every residual node r (a diagonal entry of the -I block whose row holds only
// Jacobian entries) is a leaf of the elimination tree: eliminating it first adds
// -J_ra J_rb / d_r to the (a, b) entries of the x block and -J_ra rhs_r / d_r to
// rhs_a.  These contributions are independent of each other, so instead of
// spending a sequential pivot step on each of them inside the multifrontal
// kernel, a thread-parallel pre-pass forms the condensed system
//     K2 = [ H + rho I + sum_r (-J_r' J_r / d_r)   Jc' ; Jc  -dI ]   (plus any residual row kept as a node)
// and a thread-parallel post-pass recovers the r components of the solution.
// Mathematically this IS the LDL^T of the permuted K with the r nodes first
// (pivots d_r, L rows J_r / d_r): inertia and solution are those of the
// reference; only the summation order of the Schur contributions differs.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cnl {

struct Cond {
  bool active = false;
  int64_t N = 0, nnz = 0, nvar = 0, nequ = 0, ncon = 0;  // outer (reference) dimensions
  int64_t N2 = 0, nequ2 = 0;                              // condensed system: N2 = nvar + nequ2 + ncon
  int64_t ncs = 0;                                        // unique lower-triangular slots of K2 (without the rho entries)
  int64_t cstride = 0;                                    // doubles per problem of the condensed buffer: [ncs slots | nvar rho | N2 rhs]
  // slot s in [0, ncs + nvar + N2) = sum over contributions c in [c_ptr[s], c_ptr[s+1]):
  //   c_b < 0 : x(c_a)                     plain
  //   else    : -x(c_a) * x(c_b) / x(c_d)  product
  // where x(i) = vals[i] for i < nnz and rhs[i - nnz] otherwise.
  std::vector<int32_t> c_ptr, c_a, c_b, c_d;
  // processing order of the slots (identity: the natural column-major order coalesces best)
  std::vector<int32_t> c_order;
  // Tiling for the LDS-staged condense kernel: consecutive slots are grouped in chunks; the sources a chunk
  // reads form a few contiguous ranges of [vals | rhs], which the kernel stages in LDS with coalesced loads.
  // chunk k: slots [ch_slot[k], ch_slot[k+1]), ranges [ch_rng[k], ch_rng[k+1]) of (rng_start, rng_len) in the
  // unified source space (index >= nnz = rhs), tile doubles per problem ch_tile[k]; c_la/c_lb/c_ld are the
  // contribution sources as offsets into that tile.  ch_region[r] = first chunk of region r
  // (0: matrix slots, 1: rho slots, 2: right-hand-side slots, 3: end).
  std::vector<int32_t> ch_slot, ch_rng, ch_tile, rng_start, rng_len, c_la, c_lb, c_ld;
  std::vector<int32_t> ch_tptr, tile_src;  // flattened staging list: tile position -> source index (chunk k: [ch_tptr[k], ch_tptr[k+1]))
  int32_t ch_region[4] = {0, 0, 0, 0};
  int32_t tile_max = 0;
  // condensed residual nodes: diag source, original index, Jacobian row (sources / reduced x indices)
  std::vector<int32_t> r_orig, r_dsrc, r_ptr, r_jsrc, r_jx;
  // original index -> reduced index (-1 for condensed nodes) and back
  std::vector<int32_t> red_of, orig_of;
  // pattern of the condensed system handed to build_plan (1-based, lower triangle, rho entries last)
  std::vector<int64_t> rows2, cols2;
};

// Decides which residual nodes are condensed and builds the contribution lists.
// Returns 0 or an error code; when nothing can be condensed `active` stays false.
int build_condensation(Cond& C, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar,
                       int64_t nequ, int64_t ncon, std::string& msg);

}  // namespace cnl
