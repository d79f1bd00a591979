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

constexpr int CONDENSE_TILED_PROBLEMS = 4;   // problems per workgroup of condense_tiled_kernel (kernels_aux.hip: TPB)


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
  // Tiling for the LDS-staged condense kernel: chunk k is a column range of the condensed system and owns three
  // contiguous slot ranges, ch_slot[6k..6k+5] = (start, len) of its matrix, rho and right-hand-side slots.  The
  // sources it reads ([vals | rhs], index >= nnz = rhs) are staged in LDS: tile position t holds source
  // tile_src[ch_tptr[k] + t], ch_tile[k] positions per problem; c_la/c_lb/c_ld are the contribution sources as
  // tile offsets.  ch_region[3] = number of chunks.
  std::vector<int32_t> ch_slot, ch_rng, ch_tile, rng_start, rng_len, c_la, c_lb, c_ld;
  std::vector<int32_t> ch_tptr, tile_src;  // flattened staging list: tile position -> source index (chunk k: [ch_tptr[k], ch_tptr[k+1]))
  int32_t ch_region[4] = {0, 0, 0, 0};
  int32_t tile_max = 0;
  int32_t chunk_ncon_max = 0, chunk_nslot_max = 0;  // largest contribution / slot count of a chunk
  bool tiled_ok = false;                            // the tile fits the LDS budget of the tiled kernel
  std::vector<uint64_t> c_pack;                     // per contribution: la | (lb+1) << 16 | (ld+1) << 32  (tile offsets)
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
                       int64_t nequ, int64_t ncon, std::string& msg, bool enable = true);

}  // namespace cnl
