// dense.hip — dense backend for gfx950 (see dense.h): hand-written fp64 MFMA kernels, no vendor BLAS.
//
// The condensed system S = H + rho I - J' diag(1/d_r) J (bordered by the constraint rows) of order ns is kept in
// 64 x 64 tiles.  One Newton step is a fixed, host-independent sequence of launches on the caller's stream:
//
//   dn_gather    J and w = -1/d_r out of the caller's COO values (+ inertia of the residual pivots)
//   dn_syrk      J' W J by v_mfma_f64_16x16x4_f64, lower tiles only, K split into slabs (fixed summation order)
//   dn_assemble  S0 = slabs + H entries (COO order), S = S0 + rho slots
//   per tile column k:  dn_panel (one wavefront factorises the diagonal tile, the others turn the tiles of the column
//                       into L, as soon as the pivot rows are published in LDS)  +  dn_update (MFMA rank-64 update)
//   dn_decide    inertia rule of /root/reference/src/solver_types.jl:90-97
//   dn_ladder    rho ladder of /root/reference/src/CaNNOLeS.jl:1023-1047 for the problems that failed (one workgroup
//                per problem walks the same device code; exits at once in the common case) — decided on the device
//   solve        d = -K^-1 rhs by products with G = L^-T D^-1 (see below) and one step of iterative refinement
//
// The factorisation carries an identity block below S: the same panel / update steps turn it into G = L^-T D^-1
// (upper triangular, tiles (I, J >= I)), so S^-1 = G D G' and both triangular solves become tile GEMVs that run on all
// CUs instead of two latency-bound substitution sweeps.  L itself is only needed inside the step that produces it.
// Nothing here synchronises with the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <numeric>

#include "dense.h"

namespace cnl {

namespace {

constexpr int TS = 64;       // tile order
constexpr int TT = TS * TS;  // doubles per tile
typedef double d4 __attribute__((ext_vector_type(4)));

#define DCHK(x)                                                                               \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 4; } \
  } while (0)

struct DnState {  // per problem, device
  double rho, rho_old, wrote;
  int nfact, success, done, pad;
};

struct DnDev {
  int ns, nv, T, nsp;        // order of the dense system, variables (they carry rho), tiles per side, 64 T
  int n, m, p, mp, npj, Tn;  // residual-block form: variables, residual rows (0: general condensed system), constraints, padded sizes
  int nnz;                   // COO entries per problem of the caller's vals
  int ks, ntl, nlt;          // K splits of J'WJ, lower tiles of the variable block, lower tiles of S
  int Tm, nml;               // 128 x 128 macro tiles per side of the variable block, lower macro tiles (dn_syrk2)
  int npos_ok;               // success <=> #positive pivots == npos_ok and no zero pivot
  const int *jslot, *dslot;  // gather lists
  const int *lI, *lJ;        // lower tiles of S: tile -> (I, J)
  const int *ht_ptr, *hu_loc, *hu_ptr, *hslot;  // H entries per lower tile: unique positions, their COO entries in COO order
  const int* gpos;           // general path: position i + ns j of every slot
  int nslots;
  double *Jw;                // diag(w) J: the A operand of J'WJ (dn_gather writes it next to J)
  double *Jd, *w, *slab, *S0, *S, *G, *Wn, *dv, *rsh, *y, *x, *jxp;  // Wn: [batch][T] tiles -L(J, k) d_k of the current column
  double *part1, *part2, *partA, *partB;  // [batch][T * T][64] partial tile products of the solve (summed in a fixed order)
  // J'WJ work partition: workgroup -> pieces, piece = (problem, lower tile, rows [ch0, ch1) in chunks of KT); pieces of a tile
  const int *sy_wg, *sy_b, *sy_tl, *sy_ch0, *sy_ch1, *sy_tp;
  int* cnt;                  // [batch][4]: dense pivots pos/zero, residual pivots pos/zero
  DnState* st;
};

__device__ __forceinline__ size_t tile_off(int T, int I, int J) { return ((size_t)J * T + I) * TT; }

__device__ __forceinline__ double recip(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  return fma(r, e, r);
}

// ---------------------------------------------------------------------------------------------------------------------
// J(i, j) = vals[jslot], zero-padded to mp x npj; w_i = -1 / d_r(i); inertia of the residual pivots (solver_types.jl:90-95)
__global__ void __launch_bounds__(256) dn_gather(DnDev D, const double* __restrict__ vals, double eig_tol) {
  const int b = blockIdx.y;
  const double* v = vals + (size_t)b * D.nnz;
  double* Jd = D.Jd + (size_t)b * D.mp * D.npj;
  double* Jw = D.Jw + (size_t)b * D.mp * D.npj;
  double* w = D.w + (size_t)b * D.mp;
  const long long tot = (long long)D.m * D.n;
  int np = 0, nz = 0;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < tot; t += (long long)gridDim.x * 256) {
    const int i = (int)(t % D.m), j = (int)(t / D.m);
    const double jv = v[D.jslot[t]], dvv = v[D.dslot[i]], wi = -1.0 / dvv;
    Jd[(size_t)i + (size_t)D.mp * j] = jv;
    Jw[(size_t)i + (size_t)D.mp * j] = jv * wi;
    if (j == 0) {
      w[i] = wi;
      np += dvv > eig_tol;
      nz += fabs(dvv) <= eig_tol;
    }
  }
  if (np) atomicAdd(&D.cnt[b * 4 + 2], np);
  if (nz) atomicAdd(&D.cnt[b * 4 + 3], nz);
}

// ---------------------------------------------------------------------------------------------------------------------
// slab[piece] = sum over the rows of the piece of  J(:, I)' diag(w) J(:, J)   (lower tiles I >= J).
// C' = C^T is what the matrix core produces here (D'[c][r]): the 16 lanes that share a register then hold 16 consecutive
// rows r of one column c, so the tile is written in 128-byte runs.  A operand: rows m of the J columns scaled by w,
// B operand: the I columns; both staged through LDS as [column][row] with a stride of LDK doubles (conflict-free for the
// 16 x 4 operand fetch).  Rounds 1-2 used one 64 x 64 tile per workgroup (4 waves, 32 x 32 each, 2 x 2 accumulators):
// a 64 x 64 tile per workgroup streams 32 KB of J per 32-row chunk for
// 131 k FMAs (8 flop per byte: at the matrix cores' rate that is ~6 TB/s out of L2 / Infinity Cache — the kernel ran at 27 % of
// the MFMA rate); a 128 x 128 tile halves the operand traffic per flop and every LDS operand read feeds two matrix
// instructions instead of one.  4 waves, a 64 x 64 quadrant each (4 x 4 accumulators = 16 independent chains), operands of the
// next chunk prefetched into registers while this one is multiplied.  A piece of work = (problem, lower macro tile, row
// chunks); its partial product is stored as the four 64 x 64 tiles dn_assemble sums.
// Chunks of 16 rows: the prefetch registers of a 32-row chunk (96) next to the 128 accumulator registers pushed the kernel to
// 360 VGPRs = one wavefront per SIMD, nothing to hide the barriers behind (670 us for 8 problems against 641 us before).
constexpr int MT = 128, KT = 16, LDK = 18;
__global__ void __launch_bounds__(256, 2) dn_syrk2(DnDev D) {
  extern __shared__ double sm_syrk[];
  double* tA = sm_syrk;
  double* tB = sm_syrk + MT * LDK;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, li = lane & 15, lg = lane >> 4;
  const int col = t >> 1, part = t & 1;          // staging: thread = (column of the macro tile, 8-row half of the chunk)
  const int wc = (wave >> 1) * 64, wr = (wave & 1) * 64;  // this wave's quadrant: A-side columns wc.., B-side columns wr..
  for (int pc = D.sy_wg[blockIdx.x]; pc < D.sy_wg[blockIdx.x + 1]; pc++) {
    const int b = D.sy_b[pc], ml = D.sy_tl[pc], ch0 = D.sy_ch0[pc], ch1 = D.sy_ch1[pc];
    int MJ = 0, rem = ml;  // lower macro tile, column by column
    while (rem >= D.Tm - MJ) { rem -= D.Tm - MJ; MJ++; }
    const int MI = MJ + rem;
    const double* Jb = D.Jd + (size_t)b * D.mp * D.npj;
    const double* Jwb = D.Jw + (size_t)b * D.mp * D.npj;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const double* pa = Jwb + (size_t)D.mp * min(MT * MJ + col, D.npj - 1) + 8 * part;
    const double* pb = Jb + (size_t)D.mp * min(MT * MI + col, D.npj - 1) + 8 * part;
    // named scalars: as arrays these stay allocas that the backend refuses to promote under the two-waves register budget
    double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define DN_GLOAD2(CH)                                                                 \
  {                                                                                   \
    const double2* qa_ = reinterpret_cast<const double2*>(pa + (CH) * KT);            \
    const double2* qb_ = reinterpret_cast<const double2*>(pb + (CH) * KT);            \
    ra0 = qa_[0]; ra1 = qa_[1]; ra2 = qa_[2]; ra3 = qa_[3];                           \
    rb0 = qb_[0]; rb1 = qb_[1]; rb2 = qb_[2]; rb3 = qb_[3];                           \
  }
    DN_GLOAD2(ch0)
    for (int ch = ch0; ch < ch1; ch++) {
      __syncthreads();
      {
        double2* wa_ = reinterpret_cast<double2*>(&tA[col * LDK + 8 * part]);
        double2* wb_ = reinterpret_cast<double2*>(&tB[col * LDK + 8 * part]);
        wa_[0] = ra0; wa_[1] = ra1; wa_[2] = ra2; wa_[3] = ra3;
        wb_[0] = rb0; wb_[1] = rb1; wb_[2] = rb2; wb_[3] = rb3;
      }
      __syncthreads();
      if (ch + 1 < ch1) DN_GLOAD2(ch + 1)
#pragma unroll 2
      for (int s = 0; s < KT / 4; s++) {
        double a[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          a[i] = tA[(wc + 16 * i + li) * LDK + 4 * s + lg];
          bb[i] = tB[(wr + 16 * i + li) * LDK + 4 * s + lg];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
      }
    }
#undef DN_GLOAD2
    // quadrant (wr, wc) of the macro tile = 64 x 64 tile (2 MI + (wave & 1), 2 MJ + (wave >> 1)); the upper one of a diagonal
    // macro tile is not needed
    if (!(MI == MJ && (wave & 1) == 0 && (wave >> 1) == 1)) {
      double* out = D.slab + ((size_t)pc * 4 + (size_t)((wave & 1) * 2 + (wave >> 1))) * TT;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int reg = 0; reg < 4; reg++) out[(16 * j + li) + 64 * (16 * i + lg + 4 * reg)] = acc[i][j][reg];
    }
  }
}

// S = S0 + shift on the diagonal of one lower tile (pad rows: unit diagonal)
__device__ __forceinline__ void shift_tile(const DnDev& D, const double* S0t, double* St, int I, int J, const double* rsh, int tid, int nthr) {
  for (int e = tid; e < TT; e += nthr) {
    double v = S0t[e];
    if (I == J) {
      const int r = e & 63, c = e >> 6;
      if (r == c) { const int gi = 64 * I + r; v = gi < D.ns ? v + rsh[gi] : 1.0; }
    }
    St[e] = v;
  }
}

// S0 = sum of the pieces of J'WJ (fixed order) + H entries in COO order; S = S0 + rho slots as given (first attempt)
__global__ void __launch_bounds__(256) dn_assemble(DnDev D, const double* __restrict__ vals) {
  __shared__ double tile[TT], sh64[64];
  const int b = blockIdx.y, lt = blockIdx.x, t = threadIdx.x;
  const int I = D.lI[lt], J = D.lJ[lt];
  const double* v = vals + (size_t)b * D.nnz;
  double* rsh = D.rsh + (size_t)b * D.nsp;
  const bool has_slab = D.m > 0 && I < D.Tn;
  const int MI = I >> 1, MJ = J >> 1, sub = (I & 1) * 2 + (J & 1);   // pieces come per 128 x 128 macro tile, four tiles each
  const int ml = MJ * D.Tm - (MJ * (MJ - 1)) / 2 + (MI - MJ);
  int p0 = 0, p1 = 0;
  if (has_slab) { p0 = D.sy_tp[b * D.nml + ml]; p1 = D.sy_tp[b * D.nml + ml + 1]; }
  double2 s2[8];
#pragma unroll
  for (int q = 0; q < 8; q++) s2[q] = make_double2(0.0, 0.0);
  for (int pc = p0; pc < p1; pc++) {
    const double2* sp = reinterpret_cast<const double2*>(D.slab + ((size_t)pc * 4 + sub) * TT) + t;
    double2 g[8];
#pragma unroll
    for (int q = 0; q < 8; q++) g[q] = sp[256 * q];
#pragma unroll
    for (int q = 0; q < 8; q++) { s2[q].x += g[q].x; s2[q].y += g[q].y; }
  }
#pragma unroll
  for (int q = 0; q < 8; q++) reinterpret_cast<double2*>(tile)[t + 256 * q] = s2[q];
  if (I == J && t < 64) {
    const int gi = 64 * I + t;
    const double sv = gi < D.nv ? v[D.nnz - D.nv + gi] : 0.0;
    rsh[gi] = sv;
    sh64[t] = sv;
  }
  __syncthreads();
  for (int e = D.ht_ptr[lt] + t; e < D.ht_ptr[lt + 1]; e += 256) {
    double s = tile[D.hu_loc[e]];
    for (int q = D.hu_ptr[e]; q < D.hu_ptr[e + 1]; q++) s += v[D.hslot[q]];
    tile[D.hu_loc[e]] = s;
  }
  __syncthreads();
  const size_t off = (size_t)b * D.T * D.T * TT + tile_off(D.T, I, J);
#pragma unroll
  for (int q = 0; q < 8; q++) reinterpret_cast<double2*>(D.S0 + off)[t + 256 * q] = reinterpret_cast<const double2*>(tile)[t + 256 * q];
  shift_tile(D, tile, D.S + off, I, J, sh64 - 64 * I, t, 256);
}

// general condensed system: S0 from the slots of the condensed buffer (unique positions), rho slots behind them
__global__ void __launch_bounds__(256) dn_scatter_general(DnDev D, const double* __restrict__ cbuf, long long cstride) {
  const int b = blockIdx.y;
  const double* cb = cbuf + (size_t)b * cstride;
  double* S0 = D.S0 + (size_t)b * D.T * D.T * TT;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < D.nslots; e += gridDim.x * 256) {
    const int pos = D.gpos[e], i = pos % D.ns, j = pos / D.ns;
    S0[tile_off(D.T, i >> 6, j >> 6) + (i & 63) + 64 * (j & 63)] = cb[e];
  }
  double* rsh = D.rsh + (size_t)b * D.nsp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < D.nsp; i += gridDim.x * 256) rsh[i] = i < D.nv ? cb[D.nslots + i] : 0.0;
}
__global__ void __launch_bounds__(256) dn_shift(DnDev D) {
  const int b = blockIdx.y, lt = blockIdx.x;
  const int I = D.lI[lt], J = D.lJ[lt];
  const size_t off = (size_t)b * D.T * D.T * TT + tile_off(D.T, I, J);
  shift_tile(D, D.S0 + off, D.S + off, I, J, D.rsh + (size_t)b * D.nsp, threadIdx.x, 256);
}

// ---------------------------------------------------------------------------------------------------------------------
// Panel step.  Row-per-lane: lane r holds row r of a 64 x 64 tile in registers.  The wave that owns the diagonal tile
// publishes, per pivot j, the column below the pivot BEFORE it is divided (w_r = a_r[j], one value per lane) in LDS row j
// and the reciprocal pivot; every row x of the tile column then becomes  l = x[j] / d_j,  x[c] -= l w_c (c > j).
// The next pivot's column is updated and published first, the rest of the rank-1 update runs while that LDS round
// trip is in flight.  The other waves consume the published rows eight pivots at a time.
struct PanelLds {
  double W[TS * TS];
  double inv[TS + 8];  // [TS ..]: dump slots, so that "lane 0 stores" needs no branch
  int pub[12];         // [0]: pivots published; [1 ..]: dump slots
};
#define DN_PIN(v) asm volatile("" : "+v"(v))  // the value is computed here, not sunk to its next use

__device__ __forceinline__ void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// The rank-1 update x[c] -= l w_c needs a wave-uniform w_c per instruction.  Fetching it from LDS costs one LDS read per
// two updates and the LDS latency limits the wave to ~28 cycles per update.  Instead the published column is read ONCE
// per pivot as four registers (lane i of every 16-lane row holds w[16 k + i]) and each update takes its operand through
// the DPP row broadcast of the fp64 FMA:  x[c] += W_{c / 16}[row_newbcast: c % 16] * (-l)  — one 8-cycle instruction, no
// LDS traffic.  (hipcc does not pad hazards inside asm: the DPP source comes from an LDS read, not from a VALU write.)
#define DN_FMAC_BCAST(X, WK, NL, I) \
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #I " row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(WK), "v"(NL))
template <int C>
__device__ __forceinline__ void fmac_bcast(double& x, const double (&W)[4], double nl) {
  constexpr int K = C >> 4, I = C & 15;
  if constexpr (I == 0) DN_FMAC_BCAST(x, W[K], nl, 0);
  else if constexpr (I == 1) DN_FMAC_BCAST(x, W[K], nl, 1);
  else if constexpr (I == 2) DN_FMAC_BCAST(x, W[K], nl, 2);
  else if constexpr (I == 3) DN_FMAC_BCAST(x, W[K], nl, 3);
  else if constexpr (I == 4) DN_FMAC_BCAST(x, W[K], nl, 4);
  else if constexpr (I == 5) DN_FMAC_BCAST(x, W[K], nl, 5);
  else if constexpr (I == 6) DN_FMAC_BCAST(x, W[K], nl, 6);
  else if constexpr (I == 7) DN_FMAC_BCAST(x, W[K], nl, 7);
  else if constexpr (I == 8) DN_FMAC_BCAST(x, W[K], nl, 8);
  else if constexpr (I == 9) DN_FMAC_BCAST(x, W[K], nl, 9);
  else if constexpr (I == 10) DN_FMAC_BCAST(x, W[K], nl, 10);
  else if constexpr (I == 11) DN_FMAC_BCAST(x, W[K], nl, 11);
  else if constexpr (I == 12) DN_FMAC_BCAST(x, W[K], nl, 12);
  else if constexpr (I == 13) DN_FMAC_BCAST(x, W[K], nl, 13);
  else if constexpr (I == 14) DN_FMAC_BCAST(x, W[K], nl, 14);
  else DN_FMAC_BCAST(x, W[K], nl, 15);
}
// x[c] += W[c] * nl for c = C0 .. 63
template <int C0>
__device__ __forceinline__ void bulk_from(double (&x)[TS], const double (&W)[4], double nl) {
  if constexpr (C0 < TS) {
    fmac_bcast<C0>(x[C0], W, nl);
    bulk_from<C0 + 1>(x, W, nl);
  }
}
// registers k >= K0 of row j of the published columns: lane i of every row of 16 lanes takes w[16 k + i]
template <int K0>
__device__ __forceinline__ void load_wrow(const PanelLds& P, int j, int li, double (&W)[4]) {
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (k >= K0) W[k] = P.W[j * 64 + 16 * k + li];
}

// returns the pivot counts through np / nz (same value in every lane); the pivots stay in P.W[j][j].
// Software pipeline: the region of pivot J holds the CHAIN of pivot J + 1 (publish its column, read the pivot back,
// reciprocal, multiplier) next to the BULK of pivot J (the other 62 - J updates), so the 32-cycle dependent-issue gaps
// of the fp64 chain are filled with independent updates.
template <int J>
__device__ __forceinline__ void diag_step(PanelLds& P, double (&a)[TS], double& l, double& rinv, double& d, double (&W)[4], int lane,
                                          int nreal, double eig_tol, int& np, int& nz) {
  // here: l = multiplier of pivot J (a[J] / d_J), rinv ~ 1 / d_J, d = d_J, W = row J of the published columns
  if (J < nreal) { np += d > eig_tol; nz += fabs(d) <= eig_tol; }
  const double nl = -l;
  double ln = 0.0, rn = 0.0, dn = 0.0;
  double Wn[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (J + 1 < TS) {
    fmac_bcast<J + 1>(a[J + 1], W, nl);
    P.W[(J + 1) * 64 + lane] = a[J + 1];
  }
  // the reciprocal the other waves use is refined off the critical chain
  {
    const double e = fma(-d, rinv, 1.0);
    const double r2 = fma(rinv, e, rinv);
    P.inv[lane == 0 ? J : TS + (lane & 7)] = r2;  // branch-free "lane 0 stores": no control flow inside the pipeline
  }
  // LDS executes one wave's operations in order: the counter becomes visible after the column and the reciprocal
  // without waiting for them here (a release store would stall the chain for an LDS round trip)
  asm volatile("" ::: "memory");
  __hip_atomic_store(&P.pub[lane == 0 ? 0 : 1 + (lane & 7)], J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  if constexpr (J + 1 < TS) {
    dn = P.W[(J + 1) * 64 + J + 1];
    load_wrow<((J + 2) >> 4)>(P, J + 1, lane & 15, Wn);
    // multiplier through the raw reciprocal (2^-25 on gfx950) and one residual correction: error ~ 2^-50, four dependent
    // operations on the chain instead of six
    const double r0 = __builtin_amdgcn_rcp(dn);
    const double l0 = a[J + 1] * r0;
    const double res = fma(-dn, l0, a[J + 1]);
    ln = fma(res, r0, l0);
    const double e0 = fma(-dn, r0, 1.0);
    rn = fma(r0, e0, r0);
  }
  bulk_from<J + 2>(a, W, nl);
  DN_PIN(ln);
  __builtin_amdgcn_sched_barrier(0);
  l = ln; rinv = rn; d = dn;
#pragma unroll
  for (int k = 0; k < 4; k++) W[k] = Wn[k];
  if constexpr (J + 1 < TS) diag_step<J + 1>(P, a, l, rinv, d, W, lane, nreal, eig_tol, np, nz);
}
__device__ void panel_diag(PanelLds& P, const double* __restrict__ tile, int nreal, double eig_tol, int& np_out, int& nz_out) {
  const int lane = threadIdx.x & 63;
  double a[TS];
#pragma unroll
  for (int c = 0; c < TS; c++) a[c] = tile[lane + 64 * c];
  P.W[lane] = a[0];
  int np = 0, nz = 0;
  double W[4];
  double d = P.W[0];
  load_wrow<0>(P, 0, lane & 15, W);
  double rinv = recip(d);
  double l = a[0] * rinv;
  diag_step<0>(P, a, l, rinv, d, W, lane, nreal, eig_tol, np, nz);
  np_out = np; nz_out = nz;
}

// one 64-row tile of the column: in = tile (or the identity when `ident`), out = L form
template <int J>
__device__ __forceinline__ void rows_step(PanelLds& P, double (&x)[TS], double (&W)[4], int li, double* __restrict__ wout, int lane) {
  if constexpr (J % 8 == 0) {  // eight pivots per wait on the publisher
    while (__hip_atomic_load(&P.pub[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < J + 8) __builtin_amdgcn_s_sleep(1);
    load_wrow<((J + 1) >> 4)>(P, J, li, W);
  }
  if (wout) wout[lane + 64 * J] = -x[J];  // -l d: the B operand of this column's trailing update (wave-uniform branch)
  const double l = x[J] * P.inv[J];
  x[J] = l;
  const double nl = -l;
  double Wn[4] = {0.0, 0.0, 0.0, 0.0};
  if constexpr (J % 8 != 7 && J + 1 < TS) load_wrow<((J + 2) >> 4)>(P, J + 1, li, Wn);
  bulk_from<J + 1>(x, W, nl);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (J % 8 != 7) {
#pragma unroll
    for (int k = 0; k < 4; k++) W[k] = Wn[k];
  }
  if constexpr (J + 1 < TS) rows_step<J + 1>(P, x, W, li, wout, lane);
}
__device__ void panel_rows(PanelLds& P, const double* __restrict__ in, double* __restrict__ out, double* __restrict__ wout, bool ident) {
  const int lane = threadIdx.x & 63;
  double x[TS];
  if (ident) {
#pragma unroll
    for (int c = 0; c < TS; c++) x[c] = c == lane ? 1.0 : 0.0;
  } else {
#pragma unroll
    for (int c = 0; c < TS; c++) x[c] = in[lane + 64 * c];
  }
  double W[4] = {0.0, 0.0, 0.0, 0.0};
  rows_step<0>(P, x, W, lane & 15, wout, lane);
#pragma unroll
  for (int c = 0; c < TS; c++) out[lane + 64 * c] = x[c];
}

// row tile rt of tile column k: the T - 1 - k tiles below the diagonal of S first, then the k + 1 tiles of the G block
__device__ __forceinline__ void row_tile_ptrs(const DnDev& D, int b, int k, int rt, const double*& in, double*& out, double*& wout, bool& ident) {
  const size_t pb = (size_t)b * D.T * D.T * TT;
  const int ntop = D.T - 1 - k;
  if (rt < ntop) {
    double* p = D.S + pb + tile_off(D.T, k + 1 + rt, k);
    in = p; out = p; ident = false;
    wout = D.Wn + ((size_t)b * D.T + (k + 1 + rt)) * TT;
  } else {
    const int I = rt - ntop;
    double* p = D.G + pb + tile_off(D.T, I, k);
    in = p; out = p; ident = I == k;
    wout = nullptr;
  }
}

__global__ void __launch_bounds__(256) dn_panel(DnDev D, int k, double eig_tol, int b0) {
  __shared__ PanelLds P;
  const int b = b0 + blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef DN_STAMPS
  const long long st0 = __builtin_amdgcn_s_memtime();
#endif
  if (threadIdx.x == 0) P.pub[0] = 0;
  __syncthreads();
  if (wave == 0) {
    int np, nz;
    panel_diag(P, D.S + (size_t)b * D.T * D.T * TT + tile_off(D.T, k, k), min(64, D.ns - 64 * k), eig_tol, np, nz);
#ifdef DN_STAMPS
    if (lane == 0 && blockIdx.x == 0) { long long* sp = reinterpret_cast<long long*>(D.jxp); sp[k * 4 + 0] = __builtin_amdgcn_s_memtime() - st0; }
#endif
    if (blockIdx.x == 0) {
      lds_fence();
      D.dv[(size_t)b * D.nsp + 64 * k + lane] = P.W[lane * 64 + lane];
      if (lane == 0) {
        if (np) atomicAdd(&D.cnt[b * 4 + 0], np);
        if (nz) atomicAdd(&D.cnt[b * 4 + 1], nz);
      }
    }
  } else {
    const int rt = blockIdx.x * 3 + wave - 1;
    if (rt < D.T) {
      const double* in; double* out; double* wout; bool ident;
      row_tile_ptrs(D, b, k, rt, in, out, wout, ident);
      panel_rows(P, in, out, wout, ident);
#ifdef DN_STAMPS
      if (lane == 0 && blockIdx.x == 0) { long long* sp = reinterpret_cast<long long*>(D.jxp); sp[k * 4 + wave] = __builtin_amdgcn_s_memtime() - st0; }
#endif
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Panel step, round 3 (dn_panel2): both the diagonal tile and every row tile are cut into four blocks of 16 COLUMNS, one
// wavefront each (lane = row, 16 registers).  The old kernel above gives a whole tile to one wavefront: 2016 rank-1 updates of
// 14 cycles each = 28 k cycles of issue per tile, for the diagonal tile (which paces everything) as for the row tiles — the
// panel step took 14.2 us.  With column blocks a wavefront applies the pivots of the blocks to its left as they are published
// (16 updates per pivot), then runs its own 16 pivots; the pivot chain of the diagonal tile walks from block to block and is
// free of LDS round trips: the pivot and the operands of the updates inside the block come from v_readlane (scalar
// operands of a plain v_fmac_f64), only the blocks to the right read the published column back.
// Workgroup = row tile rt of the tile column: waves 0 .. 3 the diagonal tile's blocks (recomputed by every workgroup, as
// before), waves 4 .. 7 the row tile's.  Multipliers of the row tile go from block to block through LDS (Lr).
#ifndef DN_P2_SINGLES   // pivots of the block to the left that are applied one at a time (finer hand-over): measured 0.428 ms per system
#define DN_P2_SINGLES 0  // with 0, 0.431 with 4, 0.435 with 8 — the coarser wait keeps more loads in flight
#endif
#ifndef DN_P2_ABL   // timing experiments (results wrong): 1 no row tiles, 2 no earlier-block updates in the diagonal tile, 4 no in-block
#define DN_P2_ABL 0 // bulk in the diagonal tile, 8 no earlier-block updates in the row tiles
#endif
struct Panel2Lds {
  double W[TS * TS];   // [J][r]: column J of the diagonal tile at the moment pivot J is taken (undivided)
  double Lr[TS * TS];  // [J][r]: multiplier of pivot J for row r of the row tile
  double inv[TS + 8];  // [TS ..]: dump slots ("lane 0 stores" without a branch)
  int pub[12];         // [0]: pivots of the diagonal tile published so far, [1]: pivots whose row-tile multipliers are in Lr; dump slots
};
__device__ __forceinline__ void lds_wait_ge(const int* c, int v) {
  while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// x[c] += W[row_newbcast: c] * nl for the 16 columns of a block (c >= C0)
template <int C0>
__device__ __forceinline__ void blk_update(double (&x)[16], double w, double nl) {
#define DN_B(I) if constexpr (C0 <= I) DN_FMAC_BCAST(x[I], w, nl, I);
  DN_B(0) DN_B(1) DN_B(2) DN_B(3) DN_B(4) DN_B(5) DN_B(6) DN_B(7) DN_B(8) DN_B(9) DN_B(10) DN_B(11) DN_B(12) DN_B(13) DN_B(14) DN_B(15)
#undef DN_B
}

// own pivot JJ (compile-time position inside the block) of a diagonal block.  Chain of a pivot: the column is final -> pivot and
// the operand of the NEXT column's update through v_readlane (scalar operands) -> reciprocal, multiplier -> next column.  The
// other columns of the block take their operands from the published column (one LDS read, DPP broadcast): they are needed a
// pivot later at the earliest.  No branch inside (lane-0 stores go to dump slots otherwise).
template <int JJ>
__device__ __forceinline__ void diag_own(Panel2Lds& P, double (&a)[16], int q, int lane, int li, int nreal, double eig_tol, int& np, int& nz,
                                         double rr, double wprev, double nlprev) {
  // here: a[JJ] is final; rr = v_rcp_f64 of an APPROXIMATION of its pivot (good to 2^-25, like the instruction itself), started
  // a pivot ago; wprev / nlprev: operand register and multiplier of the PREVIOUS pivot, whose updates of the columns >= JJ + 2
  // are still due.  Two chains of ~150 cycles run side by side (a dependent fp64 operation is 32 cycles on this part):
  //   pivot (v_readlane) -> Newton step on rr against the exact pivot -> multiplier -> next column            (vector path)
  //   next pivot's approximation a11 - w1 (w1 rr) -> v_rcp_f64                                               (scalar-valued path)
  const int J = 16 * q + JJ;
  const double col = a[JJ];
  P.W[J * 64 + lane] = col;
  const double d = readlane_f64(col, J);
  np += (J < nreal) & (d > eig_tol); nz += (J < nreal) & (fabs(d) <= eig_tol);
  double rrn = 1.0;
  double w1 = 0.0;
  if constexpr (JJ + 1 < 16) {
    w1 = readlane_f64(col, J + 1);
    const double a11 = readlane_f64(a[JJ + 1], J + 1);
    const double dapp = fma(-w1, w1 * rr, a11);
    rrn = __builtin_amdgcn_rcp(dapp);
    DN_PIN(rrn);  // issued here, ahead of the deferred updates below
  }
  const double e = fma(-d, rr, 1.0);
  double r1 = fma(rr, e, rr);
  // The approximation a11 - w1 (w1 rr) loses what the subtraction cancels: a pivot 10^-6 times its two terms is known to 2^-25 x
  // 10^6 = 3 % only, and one Newton step leaves the multipliers with an error of 10^-3 — enough to turn an inertia count (round 4,
  // tools/fuzz_parity.py case 40828: a rank-deficient Gauss-Newton block regularised with rho = 1.7e-3, 92 positive pivots
  // counted instead of 94 by try_to_factorize, while the ladder's own sequential refactorisation counted 94).  When the residual
  // of the speculative reciprocal is not small, the reciprocal is formed from the exact pivot (wave-uniform branch, rare).
  // (tested on the exponent field: an integer compare on the chain instead of an fp64 one; NaN and Inf take the exact path too)
  if ((__double2hiint(e) & 0x7fffffff) > 0x3e700000) r1 = recip(d);
  const double l = col * r1;
  const double nl = -l;
  if constexpr (JJ + 1 < 16) a[JJ + 1] = fma(w1, nl, a[JJ + 1]);
  // the column after that feeds the next region's a11: scalar operand as well, no LDS round trip on the chain
  if constexpr (JJ + 2 < 16) a[JJ + 2] = fma(readlane_f64(col, J + 2), nl, a[JJ + 2]);
  {  // the reciprocal the other wavefronts use: one more Newton step, off the chain
    const double e1 = fma(-d, r1, 1.0);
    P.inv[lane == 0 ? J : TS + (lane & 7)] = fma(r1, e1, r1);
  }
  asm volatile("" ::: "memory");
  __hip_atomic_store(&P.pub[lane == 0 ? 0 : 2 + (lane & 7)], J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // LDS keeps a wave's operations in order
  asm volatile("" ::: "memory");
  // operand register of THIS pivot's remaining updates (columns >= JJ + 3): read back now, used a pivot later, behind the next
  // chain's first instructions — its LDS round trip never stalls the in-order issue
  double w = 0.0;
  if constexpr (JJ + 3 < 16 && !(DN_P2_ABL & 4)) w = P.W[J * 64 + 16 * q + li];
  if constexpr (JJ >= 1 && JJ + 2 < 16 && !(DN_P2_ABL & 4)) blk_update<JJ + 2>(a, wprev, nlprev);
  if constexpr (JJ + 1 < 16) diag_own<JJ + 1>(P, a, q, lane, li, nreal, eig_tol, np, nz, rrn, w, nl);
}

__device__ __forceinline__ void panel2_diag_block(Panel2Lds& P, const double* __restrict__ tile, int q, int lane, int nreal, double eig_tol,
                                                  int& np, int& nz) {
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; c++) a[c] = tile[lane + 64 * (16 * q + c)];
  const int li = lane & 15;
  const int nbulk = (DN_P2_ABL & 2) ? 0 : 16 * q - DN_P2_SINGLES;
  for (int J0 = 0; J0 < nbulk; J0 += 4) {  // four pivots per wait: their operands are in flight together
    lds_wait_ge(&P.pub[0], J0 + 4);
    double l[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { l[u] = P.W[(J0 + u) * 64 + lane] * P.inv[J0 + u]; w[u] = P.W[(J0 + u) * 64 + 16 * q + li]; }
#pragma unroll
    for (int u = 0; u < 4; u++) blk_update<0>(a, w[u], -l[u]);
  }
  // the last pivots of the block to the left one at a time: this wavefront's own chain starts right behind its last pivot
  for (int J = nbulk < 0 ? 0 : nbulk; J < 16 * q && !(DN_P2_ABL & 2); J++) {
    lds_wait_ge(&P.pub[0], J + 1);
    const double l = P.W[J * 64 + lane] * P.inv[J];
    const double w = P.W[J * 64 + 16 * q + li];
    blk_update<0>(a, w, -l);
  }
  np = 0; nz = 0;
  diag_own<0>(P, a, q, lane, li, nreal, eig_tol, np, nz, __builtin_amdgcn_rcp(readlane_f64(a[0], 16 * q)), 0.0, 0.0);
}

// own pivots JJ .. JJ + 3 of a row block: the four reciprocals and operand registers are read from LDS TOGETHER behind one wait
// on the publisher, so that the chain from pivot to pivot is one update and one multiplication (in-kernel stamps, round 3: with an
// LDS round trip for the reciprocal and another for the operands per pivot a row block needed 350 ticks per pivot against the
// diagonal tile's 300 — the row tiles, not the pivot chain, set the length of the panel step)
template <int JJ>
__device__ __forceinline__ void rows_own(Panel2Lds& P, double (&x)[16], int q, int lane, int li, double* __restrict__ wout) {
  const int J0 = 16 * q + JJ;
  lds_wait_ge(&P.pub[0], J0 + 4);
  double iv[4], w[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { iv[u] = P.inv[J0 + u]; w[u] = P.W[(J0 + u) * 64 + 16 * q + li]; }
#define DN_ROWS_PIVOT(U)                                                                       \
  {                                                                                            \
    constexpr int C = JJ + U;                                                                  \
    if (wout) wout[lane + 64 * (J0 + U)] = -x[C];  /* -l d: the B operand of this column's trailing update */ \
    const double l = x[C] * iv[U];                                                             \
    P.Lr[(J0 + U) * 64 + lane] = l;                                                            \
    x[C] = l;                                                                                  \
    if constexpr (C + 1 < 16) blk_update<C + 1>(x, w[U], -l);                                  \
  }
  DN_ROWS_PIVOT(0) DN_ROWS_PIVOT(1) DN_ROWS_PIVOT(2) DN_ROWS_PIVOT(3)
#undef DN_ROWS_PIVOT
  asm volatile("" ::: "memory");
  __hip_atomic_store(&P.pub[lane == 0 ? 1 : 2 + (lane & 7)], J0 + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  if constexpr (JJ + 4 < 16) rows_own<JJ + 4>(P, x, q, lane, li, wout);
}

__device__ __forceinline__ void panel2_rows_block(Panel2Lds& P, const double* __restrict__ in, double* __restrict__ out, double* __restrict__ wout,
                                                  bool ident, int q, int lane) {
  double x[16];
#pragma unroll
  for (int c = 0; c < 16; c++) x[c] = ident ? ((16 * q + c) == lane ? 1.0 : 0.0) : in[lane + 64 * (16 * q + c)];
  const int li = lane & 15;
  for (int J0 = 0; J0 < 16 * q && !(DN_P2_ABL & 8); J0 += 4) {
    lds_wait_ge(&P.pub[1], J0 + 4);
    double l[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { l[u] = P.Lr[(J0 + u) * 64 + lane]; w[u] = P.W[(J0 + u) * 64 + 16 * q + li]; }
#pragma unroll
    for (int u = 0; u < 4; u++) blk_update<0>(x, w[u], -l[u]);
  }
  rows_own<0>(P, x, q, lane, li, wout);
#pragma unroll
  for (int c = 0; c < 16; c++) out[lane + 64 * (16 * q + c)] = x[c];
}

__global__ void __launch_bounds__(512) dn_panel2(DnDev D, int k, double eig_tol) {
  extern __shared__ double dn_p2_lds[];
  Panel2Lds& P = *reinterpret_cast<Panel2Lds*>(dn_p2_lds);
  const int b = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { P.pub[0] = 0; P.pub[1] = 0; }
#ifdef DN_STAMPS
  const long long st0 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
  if (wave < 4) {
    int np, nz;
    panel2_diag_block(P, D.S + (size_t)b * D.T * D.T * TT + tile_off(D.T, k, k), wave, lane, min(64, D.ns - 64 * k), eig_tol, np, nz);
#ifdef DN_STAMPS
    if (lane == 0 && blockIdx.x == 0) { long long* sp = reinterpret_cast<long long*>(D.jxp); sp[D.T * 4 + k * 8 + wave] = __builtin_amdgcn_s_memtime() - st0; }
#endif
    if (blockIdx.x == 0) {
      if (lane >= 16 * wave && lane < 16 * wave + 16) D.dv[(size_t)b * D.nsp + 64 * k + lane] = P.W[lane * 64 + lane];  // own column: written by this wave
      if (lane == 0) {
        if (np) atomicAdd(&D.cnt[b * 4 + 0], np);
        if (nz) atomicAdd(&D.cnt[b * 4 + 1], nz);
      }
    }
  } else if (!(DN_P2_ABL & 1)) {
    const int rt = blockIdx.x;
    const double* in; double* out; double* wout; bool ident;
    row_tile_ptrs(D, b, k, rt, in, out, wout, ident);
    panel2_rows_block(P, in, out, wout, ident, wave - 4, lane);
#ifdef DN_STAMPS
    if (lane == 0 && blockIdx.x == 0) { long long* sp = reinterpret_cast<long long*>(D.jxp); sp[D.T * 4 + k * 8 + wave] = __builtin_amdgcn_s_memtime() - st0; }
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// C(I, J) -= L(I, k) diag(d) L(J, k)'  (tiles of S, or of the G block with G(I, k) in place of L(I, k)): rank-64 update of
// one tile on the fp64 matrix cores.  The core produces C' (see dn_syrk); its A operand is the tile -L(J, k) d that the
// panel step left in Wn, its B operand the L form of the row tile.  One wave per 16 x 16 block: it loads its 16 + 16
// operand fragments (one fp64 per lane each) straight from L2 in the operand layout, all loads in flight at once, then
// runs the 16 dependent matrix instructions — no LDS staging, no barriers (the kernel is latency-bound: a whole step has
// fewer tiles than the chip has CUs).
__device__ __forceinline__ void update_block(const double* __restrict__ At, const double* __restrict__ Wt, double* __restrict__ Ct, bool czero,
                                             int blk, int lane) {
  const int li = lane & 15, lg = lane >> 4, bc = blk >> 2, br = blk & 3;
  double aop[16], bop[16];
#pragma unroll
  for (int s = 0; s < 16; s++) {
    aop[s] = Wt[(16 * bc + li) + 64 * (4 * s + lg)];
    bop[s] = At[(16 * br + li) + 64 * (4 * s + lg)];
  }
  // two accumulators (even / odd k-steps): a dependent fp64 MFMA waits 196 cycles, two independent chains issue every ~150
  d4 acc, acc2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int reg = 0; reg < 4; reg++) acc[reg] = czero ? 0.0 : Ct[(16 * br + li) + 64 * (16 * bc + lg + 4 * reg)];
#pragma unroll
  for (int s = 0; s < 16; s += 2) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s], bop[s], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s + 1], bop[s + 1], acc2, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; reg++) Ct[(16 * br + li) + 64 * (16 * bc + lg + 4 * reg)] = acc[reg] + acc2[reg];
}

// trailing tile idx of step k: lower tiles (I >= J > k) of S first, then the tiles (I <= k, J > k) of the G block
__device__ __forceinline__ void trailing_tile(const DnDev& D, int b, int k, int idx, const double*& At, const double*& Bt, double*& Ct, bool& czero) {
  const size_t pb = (size_t)b * D.T * D.T * TT;
  const int nt = D.T - 1 - k, ntop = nt * (nt + 1) / 2;
  int I, J;
  if (idx < ntop) {
    int c = 0, rem = idx;
    while (rem >= nt - c) { rem -= nt - c; c++; }
    J = k + 1 + c; I = J + rem;
    At = D.S + pb + tile_off(D.T, I, k);
    Ct = D.S + pb + tile_off(D.T, I, J);
    czero = false;
  } else {
    const int q = idx - ntop;
    I = q / nt; J = k + 1 + q % nt;
    At = D.G + pb + tile_off(D.T, I, k);
    Ct = D.G + pb + tile_off(D.T, I, J);
    czero = I == k;  // first touch of this tile of the G block
  }
  Bt = D.Wn + ((size_t)b * D.T + J) * TT;
}

// one workgroup = one 32 x 32 quadrant of a trailing tile (4 waves, a 16 x 16 block each)
__global__ void __launch_bounds__(256) dn_update(DnDev D, int k, int b0) {
  const int b = b0 + blockIdx.y;
  const double *At, *Bt; double* Ct; bool czero;
  trailing_tile(D, b, k, blockIdx.x >> 2, At, Bt, Ct, czero);
  const int q = blockIdx.x & 3, wave = threadIdx.x >> 6;
  const int blk = ((q >> 1) * 2 + (wave >> 1)) * 4 + (q & 1) * 2 + (wave & 1);
  update_block(At, Bt, Ct, czero, blk, threadIdx.x & 63);
}

// (Round 3, not kept: trailing update of column k and panel of column k + 1 in ONE launch — the panel wavefronts wait on per-tile
//  counters for the quadrants of their tiles, the tiles of column k + 1 are updated first.  Producer and consumer may sit on
//  different XCDs, whose L2s are not coherent with each other: the hand-over needs either an agent-scope release per quadrant (an
//  L2 write-back: 0.46 -> 0.55 ms at one problem, 1.10 -> 2.05 ms at eight) or device-scope write-through stores and loads
//  (0.63 / 1.30 ms; 32 problems 3.27 -> 5.10 ms).  A kernel boundary is the cheaper hand-over on this part.)
// (Round 3: a workgroup-per-tile variant with both operand tiles staged in LDS once — a quarter of the operand traffic out of L2 —
//  was built for batches: 18.7 us per step at 8 problems against 18.1 us for the kernel above.  One or two workgroups of 80 KB per
//  CU serialise their load and multiply phases; the many small independent wavefronts of this kernel overlap them.  Not kept.)

// ---------------------------------------------------------------------------------------------------------------------
// inertia rule + first rung of the ladder state.  mode 0: newton, 1: factorize
__global__ void __launch_bounds__(256) dn_decide(DnDev D, int batch, int mode, const double* __restrict__ rho_old, int32_t* success,
                                                 int64_t* npos, int64_t* nzero) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  const int tp = D.cnt[b * 4 + 0] + D.cnt[b * 4 + 2], tz = D.cnt[b * 4 + 1] + D.cnt[b * 4 + 3];
  const bool ok = tp == D.npos_ok && tz == 0;
  D.cnt[b * 4 + 0] = 0; D.cnt[b * 4 + 1] = 0; D.cnt[b * 4 + 2] = 0; D.cnt[b * 4 + 3] = 0;  // clean for the next call
  DnState s;
  s.rho = 0.0; s.wrote = 0.0; s.rho_old = rho_old ? rho_old[b] : 0.0;
  s.nfact = 1; s.success = ok ? 1 : 0; s.done = (ok || mode == 1) ? 1 : 0; s.pad = 0;
  D.st[b] = s;
  if (mode == 1) {
    success[b] = ok ? 1 : 0;
    if (npos) npos[b] = tp;
    if (nzero) nzero[b] = tz;
  }
}

// rho ladder for the problems whose first factorisation failed: one workgroup per problem repeats the factorisation with
// S = S0 + rho I (src/CaNNOLeS.jl:1029-1047).  Same device code as the per-step kernels, walked tile by tile.
constexpr int LNW = 8;
__global__ void __launch_bounds__(LNW * 64) dn_ladder(DnDev D, double* vals, const double* __restrict__ rho_old_in, double eig_tol, double kdec,
                                                      double kinc, double klarge, double rho0, double rhomax, double rhomin) {
  __shared__ PanelLds P;
  __shared__ int scnt[2];
  __shared__ int s_first[3];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // inertia rule on the first factorisation (src/solver_types.jl:90-97); the counters are left clean for the next call
  if (tid == 0) {
    const int c0 = D.cnt[b * 4 + 0], c1 = D.cnt[b * 4 + 1], c2 = D.cnt[b * 4 + 2], c3 = D.cnt[b * 4 + 3];
    s_first[0] = (c0 + c2 == D.npos_ok && c1 + c3 == 0) ? 1 : 0; s_first[1] = c2; s_first[2] = c3;
    D.cnt[b * 4 + 0] = 0; D.cnt[b * 4 + 1] = 0; D.cnt[b * 4 + 2] = 0; D.cnt[b * 4 + 3] = 0;
  }
  __syncthreads();
  DnState s;
  s.rho = 0.0; s.wrote = 0.0; s.rho_old = rho_old_in[b]; s.nfact = 1; s.success = 1; s.done = 1; s.pad = 0;
  if (s_first[0]) {
    if (tid == 0) D.st[b] = s;
    return;
  }
  const size_t pb = (size_t)b * D.T * D.T * TT;
  double* rsh = D.rsh + (size_t)b * D.nsp;
  const int rp = s_first[1], rz = s_first[2];
  double rho = s.rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * s.rho_old);
  double wrote = rho;
  int nfact = 1;
  bool success = false;
  while (true) {
    // ---- one attempt at `rho`
    for (int i = tid; i < D.nsp; i += LNW * 64) rsh[i] = i < D.nv ? rho : 0.0;
    if (tid == 0) { scnt[0] = 0; scnt[1] = 0; }
    __threadfence();
    __syncthreads();
    for (int lt = 0; lt < D.nlt; lt++) {
      const int I = D.lI[lt], J = D.lJ[lt];
      const size_t off = pb + tile_off(D.T, I, J);
      shift_tile(D, D.S0 + off, D.S + off, I, J, rsh, tid, LNW * 64);
    }
    __threadfence();
    __syncthreads();
    for (int k = 0; k < D.T; k++) {
      if (tid == 0) P.pub[0] = 0;
      __syncthreads();
      for (int pass = 0; pass * (LNW - 1) < D.T; pass++) {
        if (wave == 0) {
          if (pass == 0) {
            int np, nz;
            panel_diag(P, D.S + pb + tile_off(D.T, k, k), min(64, D.ns - 64 * k), eig_tol, np, nz);
            lds_fence();
            D.dv[(size_t)b * D.nsp + 64 * k + lane] = P.W[lane * 64 + lane];
            if (lane == 0) { scnt[0] += np; scnt[1] += nz; }
          }
        } else {
          const int rt = pass * (LNW - 1) + wave - 1;
          if (rt < D.T) {
            const double* in; double* out; double* wout; bool ident;
            row_tile_ptrs(D, b, k, rt, in, out, wout, ident);
            panel_rows(P, in, out, wout, ident);
          }
        }
      }
      __threadfence();
      __syncthreads();
      const int nt = D.T - 1 - k, ntr = nt * (nt + 1) / 2 + (k + 1) * nt;
      for (int idx = 0; idx < ntr; idx++) {
        const double *At, *Bt; double* Ct; bool czero;
        trailing_tile(D, b, k, idx, At, Bt, Ct, czero);
        update_block(At, Bt, Ct, czero, 2 * wave, lane);
        update_block(At, Bt, Ct, czero, 2 * wave + 1, lane);
      }
      __threadfence();
      __syncthreads();
    }
    nfact++;
    success = scnt[0] + rp == D.npos_ok && scnt[1] + rz == 0;
    __syncthreads();
    if (success || rho > rhomax) break;
    rho = s.rho_old == 0.0 ? klarge * rho : kinc * rho;
    if (rho > rhomax) break;
    wrote = rho;
  }
  if (rho <= rhomax) s.rho_old = rho;
  s.rho = rho; s.wrote = wrote; s.nfact = nfact; s.success = success ? 1 : 0; s.done = 1;
  if (tid == 0) D.st[b] = s;
  double* vt = vals + (size_t)b * D.nnz + (D.nnz - D.nv);
  for (int i = tid; i < D.nv; i += LNW * 64) vt[i] = wrote;
}

// ---------------------------------------------------------------------------------------------------------------------
// solve: y = [rhs_x - J' (rhs_r / d_r); rhs_c]  ->  x = G D G' y  (+ one refinement step with S)  ->  d
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per variable column: y_c = rhs_x[c] + sum_i J(i, c) w_i rhs_r[i]; the constraint part and the pad behind it.
// t = w .* rhs_r is staged in LDS once per workgroup (2048 rows at a time), the column of J streams through 16-byte loads.
__global__ void __launch_bounds__(256) dn_yrhs(DnDev D, const double* __restrict__ rhs, int gate) {
  __shared__ double ts[2048];
  const int b = blockIdx.y;
  if (gate && !D.st[b].success) return;
  const int N = D.n + D.m + D.p;
  const double* rb = rhs + (size_t)b * N;
  double* y = D.y + (size_t)b * D.nsp;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const double* w = D.w + (size_t)b * D.mp;
  const double* Jc = D.Jd + (size_t)b * D.mp * D.npj + (size_t)D.mp * min(c, D.npj - 1);
  double s = 0.0;
  for (int i0 = 0; i0 < D.mp; i0 += 2048) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256) { const int gi = i0 + i; ts[i] = gi < D.m ? w[gi] * rb[D.n + gi] : 0.0; }
    __syncthreads();
    const int lim = min(2048, D.mp - i0);  // mp is a multiple of 64
#pragma unroll 4
    for (int i = 2 * lane; i < lim; i += 128) {
      const double2 j2 = *reinterpret_cast<const double2*>(Jc + i0 + i);
      s = fma(j2.x, ts[i], s);
      s = fma(j2.y, ts[i + 1], s);
    }
  }
  s = wave_sum(s);
  if (lane == 0) {
    if (c < D.n) y[c] = rb[c] + s;
    else if (c < D.nsp) y[c] = c < D.ns ? rb[D.n + D.m + (c - D.n)] : 0.0;
  }
}
__global__ void __launch_bounds__(256) dn_yrhs_general(DnDev D, const double* __restrict__ cbuf, long long cstride, int gate) {
  const int b = blockIdx.y;
  if (gate && !D.st[b].success) return;
  const double* crhs = cbuf + (size_t)b * cstride + D.nslots + D.nv;
  double* y = D.y + (size_t)b * D.nsp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < D.nsp; i += gridDim.x * 256) y[i] = i < D.ns ? crhs[i] : 0.0;
}

// ---- tile GEMVs of the solve.  Every tile of G (upper, J >= I) resp. S (lower) is one workgroup; a tile product is 64
// partial sums, and the vector a later kernel needs is summed from the partials in a fixed order by the workgroup that
// needs it (a few KB), so the results are deterministic and no reduction launches are needed.
__device__ __forceinline__ size_t part_off(const DnDev& D, int b, int I, int J) { return (((size_t)b * D.T + J) * D.T + I) * 64; }

// sum_{q = q0}^{q1 - 1} base[q * stride] in index order; the loads of eight terms are in flight together (a plain loop waits
// an L2 round trip per term: these reductions were most of the time of the small solve kernels)
__device__ __forceinline__ double sum_strided(const double* __restrict__ base, size_t stride, int q0, int q1) {
  double s = 0.0;
  for (int c0 = q0; c0 < q1; c0 += 8) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; q++) v[q] = c0 + q < q1 ? base[(size_t)(c0 + q) * stride] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) s += v[q];
  }
  return s;
}

// row sums out[r] = sum_c tile[r + 64 c] v[c] over the columns with mask(r, c); 256 threads, v in LDS, red = [4][64] LDS
template <class M>
__device__ __forceinline__ void tile_rowsum(const double* __restrict__ tile, const double* v, double (*red)[64], double* __restrict__ out, M mask) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double tv[16];
#pragma unroll
  for (int q = 0; q < 16; q++) tv[q] = tile[lane + 64 * (16 * wave + q)];
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q++) if (mask(lane, 16 * wave + q)) acc = fma(tv[q], v[16 * wave + q], acc);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) out[lane] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  __syncthreads();
}
// column sums out[c] = sum_r tile[r + 64 c] v[r] over the rows with mask(r, c)
template <class M>
__device__ __forceinline__ void tile_colsum(const double* __restrict__ tile, const double* v, double* __restrict__ out, M mask) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double tv[16];
#pragma unroll
  for (int q = 0; q < 16; q++) tv[q] = tile[lane + 64 * (16 * wave + q)];
  const double vr = v[lane];
  double mine = 0.0;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const double s = wave_sum(mask(lane, 16 * wave + q) ? tv[q] * vr : 0.0);
    if (lane == q) mine = s;
  }
  if (lane < 16) out[16 * wave + lane] = mine;
}
struct MaskAll { __device__ bool operator()(int, int) const { return true; } };
struct MaskLowerEq { __device__ bool operator()(int r, int c) const { return c <= r; } };
struct MaskStrictLower { __device__ bool operator()(int r, int c) const { return r > c; } };

// residual of the first solve, block I, entry i:  y - shift x - (S x)  from the partial products of dn_resid_part
__device__ __forceinline__ double resid_entry(const DnDev& D, int b, int I, int i) {
  const int gi = 64 * I + i;
  const size_t vb = (size_t)b * D.nsp;
  const double sh = gi < D.ns ? D.rsh[vb + gi] : 1.0;
  double s = sh * D.x[vb + gi];
  s += sum_strided(D.partA + part_off(D, b, I, 0) + i, (size_t)D.T * 64, 0, I + 1);   // J = 0 .. I: part_off advances by T * 64 per J
  s += sum_strided(D.partB + part_off(D, b, 0, I) + i, 64, I, D.T);                     // K = I .. T - 1: by 64 per K
  return D.y[vb + gi] - s;
}

// part1(I, J) = G(I, J)' v_I for the upper tiles; v = y (pass 0) or the residual of the first solve (pass 1)
__global__ void __launch_bounds__(256) dn_gt_part(DnDev D, int pass, int gate) {
  __shared__ double vs[64];
  const int b = blockIdx.y, I = D.lJ[blockIdx.x], J = D.lI[blockIdx.x];  // lower-tile list read transposed: J >= I
  if (gate && !D.st[b].success) return;
  if (threadIdx.x < 64) vs[threadIdx.x] = pass == 0 ? D.y[(size_t)b * D.nsp + 64 * I + threadIdx.x] : resid_entry(D, b, I, threadIdx.x);
  __syncthreads();
  tile_colsum(D.G + (size_t)b * D.T * D.T * TT + tile_off(D.T, I, J), vs, D.part1 + part_off(D, b, I, J), MaskAll());
}

// part2(I, J) = G(I, J) u_J with u_J = dv_J .* sum_{I' <= J} part1(I', J)
__global__ void __launch_bounds__(256) dn_g_part(DnDev D, int gate) {
  __shared__ double us[64], red[4][64];
  const int b = blockIdx.y, I = D.lJ[blockIdx.x], J = D.lI[blockIdx.x];
  if (gate && !D.st[b].success) return;
  if (threadIdx.x < 64) {
    const double s = sum_strided(D.part1 + part_off(D, b, 0, J) + threadIdx.x, 64, 0, J + 1);
    us[threadIdx.x] = D.dv[(size_t)b * D.nsp + 64 * J + threadIdx.x] * s;
  }
  __syncthreads();
  tile_rowsum(D.G + (size_t)b * D.T * D.T * TT + tile_off(D.T, I, J), us, red, D.part2 + part_off(D, b, I, J), MaskAll());
}

// x = sum of part2 (first solve); partA(I, J) = S(I, J) x_J, partB(I, J) = S(I, J)' x_I over the lower tiles of S0
// (the diagonal tiles hold the lower triangle: A takes c <= r, B the strictly lower part transposed)
__global__ void __launch_bounds__(256) dn_resid_part(DnDev D, int gate) {
  __shared__ double xJ[64], xI[64], red[4][64];
  const int b = blockIdx.y, I = D.lI[blockIdx.x], J = D.lJ[blockIdx.x];
  if (gate && !D.st[b].success) return;
  if (threadIdx.x < 128) {
    const int B0 = threadIdx.x < 64 ? J : I, i = threadIdx.x & 63;
    const double s = sum_strided(D.part2 + part_off(D, b, B0, 0) + i, (size_t)D.T * 64, B0, D.T);
    (threadIdx.x < 64 ? xJ : xI)[i] = s;
    if (I == J && threadIdx.x < 64) D.x[(size_t)b * D.nsp + 64 * I + i] = s;
  }
  __syncthreads();
  const double* tile = D.S0 + (size_t)b * D.T * D.T * TT + tile_off(D.T, I, J);
  if (I == J) {
    tile_rowsum(tile, xJ, red, D.partA + part_off(D, b, I, J), MaskLowerEq());
    tile_colsum(tile, xI, D.partB + part_off(D, b, I, J), MaskStrictLower());
  } else {
    tile_rowsum(tile, xJ, red, D.partA + part_off(D, b, I, J), MaskAll());
    tile_colsum(tile, xI, D.partB + part_off(D, b, I, J), MaskAll());
  }
}

// x += correction of the refinement step (sum of its part2)
__global__ void __launch_bounds__(64) dn_xfinal(DnDev D, int gate) {
  const int b = blockIdx.y, I = blockIdx.x, i = threadIdx.x;
  if (gate && !D.st[b].success) return;
  const double s = D.x[(size_t)b * D.nsp + 64 * I + i] + sum_strided(D.part2 + part_off(D, b, I, 0) + i, (size_t)D.T * 64, I, D.T);
  D.x[(size_t)b * D.nsp + 64 * I + i] = s;
}

// partial products of J x: rows in blocks of 256, columns in chunks of 32
__global__ void __launch_bounds__(256) dn_jx(DnDev D, int gate) {
  const int b = blockIdx.z;
  if (gate && !D.st[b].success) return;
  const int i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
  const double* Jb = D.Jd + (size_t)b * D.mp * D.npj;
  const double* x = D.x + (size_t)b * D.nsp;
  double acc = 0.0;
  if (i < D.mp) {
#pragma unroll 8
    for (int c = 32 * ch; c < 32 * ch + 32; c++) acc = fma(Jb[(size_t)i + (size_t)D.mp * c], x[c], acc);
    D.jxp[((size_t)b * (D.npj / 32) + ch) * D.mp + i] = acc;
  }
}

// d = [-x; -(rhs_r - J x) / d_r; -x_c] and the scalars of newton_system!'s return tuple
__global__ void __launch_bounds__(256) dn_out(DnDev D, const double* __restrict__ rhs, double* __restrict__ d, int mode, double* rho_old,
                                              double* rho, int32_t* nfact, int32_t* success) {
  const int b = blockIdx.y;
  const DnState s = D.st[b];
  if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    rho[b] = s.rho; rho_old[b] = s.rho_old; nfact[b] = s.nfact; success[b] = s.success;
  }
  if (mode == 0 && !s.success) return;
  const int N = D.n + D.m + D.p;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double* x = D.x + (size_t)b * D.nsp;
  double* db = d + (size_t)b * N;
  if (i < D.n) db[i] = -x[i];
  else if (i < D.n + D.m) {
    const int q = i - D.n;
    const double jx = sum_strided(D.jxp + (size_t)b * (D.npj / 32) * D.mp + q, (size_t)D.mp, 0, D.npj / 32);
    db[i] = D.w[(size_t)b * D.mp + q] * (rhs[(size_t)b * N + i] - jx);
  } else db[i] = -x[D.n + (i - D.n - D.m)];
}
__global__ void __launch_bounds__(256) dn_out_general(DnDev D, double* __restrict__ d2, int mode, double* rho_old, double* rho, int32_t* nfact,
                                                      int32_t* success) {
  const int b = blockIdx.y;
  const DnState s = D.st[b];
  if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    rho[b] = s.rho; rho_old[b] = s.rho_old; nfact[b] = s.nfact; success[b] = s.success;
  }
  if (mode == 0 && !s.success) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < D.ns) d2[(size_t)b * D.ns + i] = -D.x[(size_t)b * D.nsp + i];
}
// residual pivots counted outside (general path)
__global__ void __launch_bounds__(256) dn_set_counts(DnDev D, int batch, const int* __restrict__ xpos, const int* __restrict__ xzer) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  D.cnt[b * 4 + 0] = 0; D.cnt[b * 4 + 1] = 0;
  D.cnt[b * 4 + 2] = xpos ? xpos[b] : 0; D.cnt[b * 4 + 3] = xzer ? xzer[b] : 0;
}

inline int blocks(long long n) { return (int)((n + 255) / 256); }

}  // namespace

// A call is a fixed sequence of ~50 short dependent launches; replayed as a hipGraph the boundary between two of them costs
// ~1.7 us instead of ~2.7 us (tools/microbench.hip).  Graphs are cached per argument set (the pointers are baked in).
struct GraphKey {
  int mode = -1;
  const void* p[10] = {};
  double params[9] = {};
  long long extra = 0;
  bool operator==(const GraphKey& o) const {
    if (mode != o.mode || extra != o.extra) return false;
    for (int i = 0; i < 10; i++) if (p[i] != o.p[i]) return false;
    for (int i = 0; i < 9; i++) if (params[i] != o.params[i]) return false;
    return true;
  }
};
struct GraphEntry { GraphKey key; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };

struct DenseState {
  DnDev d{};
  int64_t batch = 1;
  bool general = false;
  bool factored = false;
  std::vector<void*> allocs;
  std::vector<GraphEntry> graphs;
  hipStream_t cap = nullptr;
  bool use_graph = true;
  bool panel2 = true;  // column-block panel kernel (dn_panel2)
  bool panel2_auto = true;  // ... chosen by batch x tiles (setup_common)
  int misses = 0;  // consecutive calls whose argument set was not cached (a caller that hands over fresh buffers every step)
};

bool detect_dense(DensePlan& D, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                  int64_t ncon) {
  D.active = false;
  if (nequ <= 0 || nvar < 32 || N != nvar + nequ + ncon) return false;
  if ((long long)nvar * nequ > (1ll << 28) || nnz > (1ll << 30) || (nvar + ncon) > 4096) return false;  // the solve kernels keep a vector of the dense order in LDS
  D.n = (int32_t)nvar; D.m = (int32_t)nequ; D.p = (int32_t)ncon; D.nnz = (int32_t)nnz;
  const int64_t ns = nvar + ncon;  // order of the dense system
  D.jslot.assign((size_t)nvar * nequ, -1);
  D.dslot.assign(nequ, -1);
  D.hslot.clear(); D.hpos.clear();
  const int64_t rho_begin = nnz - nvar;
  for (int64_t e = 0; e < nnz; e++) {
    const int64_t i = rows1[e] - 1, j = cols1[e] - 1;
    if (i < nvar) {  // H_F / H_c entry or rho slot (lower triangle, column j <= row i)
      if (e >= rho_begin) continue;  // rho slots are applied as a shift of the diagonal
      D.hslot.push_back((int32_t)e); D.hpos.push_back((int32_t)(i + ns * j));
    } else if (i < nvar + nequ) {
      if (j < nvar) {
        int32_t& s = D.jslot[(size_t)(i - nvar) + (size_t)nequ * j];
        if (s >= 0) return false;  // duplicate Jacobian entry: not handled here
        s = (int32_t)e;
      } else if (i == j) {
        if (D.dslot[i - nvar] >= 0) return false;
        D.dslot[i - nvar] = (int32_t)e;
      } else return false;
    } else {  // constraint row k: Jacobian entries and the -delta diagonal go straight into the dense system
      const int64_t k = i - nvar - nequ;
      if (j < nvar) { D.hslot.push_back((int32_t)e); D.hpos.push_back((int32_t)((nvar + k) + ns * j)); }
      else if (i == j) { D.hslot.push_back((int32_t)e); D.hpos.push_back((int32_t)((nvar + k) + ns * (nvar + k))); }
      else return false;
    }
  }
  for (int32_t s : D.jslot) if (s < 0) return false;  // a residual row that does not hold every variable
  for (int32_t s : D.dslot) if (s < 0) return false;
  // rho slots must be the diagonal (i, i), in order
  for (int64_t k = 0; k < nvar; k++) if (rows1[rho_begin + k] - 1 != k || cols1[rho_begin + k] - 1 != k) return false;
  D.active = true;
  return true;
}

namespace {

template <class T>
int dalloc_(DenseState* st, T** p, size_t count, std::string& err, bool zero = false) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess) { err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e); return 4; }
  st->allocs.push_back(q);
  if (zero && hipMemset(q, 0, bytes) != hipSuccess) { err = "hipMemset failed"; return 4; }
  *p = (T*)q;
  return 0;
}
template <class T>
int upload_(DenseState* st, const T** p, const std::vector<T>& v, std::string& err) {
  T* q = nullptr;
  int rc = dalloc_(st, &q, v.size(), err);
  if (rc) return rc;
  if (!v.empty() && hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy failed"; return 4; }
  *p = q;
  return 0;
}

// sizes, tile lists and buffers common to both forms
int setup_common(DenseState* st, int ns, int nv, int64_t batch, std::string& err) {
  DnDev& d = st->d;
#ifdef DN_STAMPS
  st->use_graph = false;
#endif
  d.ns = ns; d.nv = nv; d.T = (ns + TS - 1) / TS; d.nsp = d.T * TS;
  d.nlt = d.T * (d.T + 1) / 2;
  // The column-block panel kernel shortens the latency of a step at the price of T workgroups per problem that each refactorise
  // the diagonal tile with four wavefronts (the one-wavefront kernel: T / 3 workgroups).  It pays while the step is latency-bound:
  // measured crossovers at 96 .. 640 problems of order 256 (T = 4) and at 32 problems of order 1050 (T = 17).
  if (st->panel2_auto) st->panel2 = (long long)batch * d.T <= 512;
  if (st->panel2 && hipFuncSetAttribute(reinterpret_cast<const void*>(dn_panel2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Panel2Lds)) != hipSuccess) {
    (void)hipGetLastError();
    st->panel2 = false;  // the one-wavefront-per-tile panel kernel needs no opt-in
  }
  std::vector<int> lI, lJ;
  for (int J = 0; J < d.T; J++) for (int I = J; I < d.T; I++) { lI.push_back(I); lJ.push_back(J); }
  int rc;
  if ((rc = upload_(st, &d.lI, lI, err))) return rc;
  if ((rc = upload_(st, &d.lJ, lJ, err))) return rc;
  const size_t B = (size_t)batch, mat = (size_t)d.T * d.T * TT;
  if ((rc = dalloc_(st, &d.S0, B * mat, err, true))) return rc;
  if ((rc = dalloc_(st, &d.S, B * mat, err, true))) return rc;
  if ((rc = dalloc_(st, &d.G, B * mat, err, true))) return rc;
  if ((rc = dalloc_(st, &d.Wn, B * d.T * TT, err, true))) return rc;
  for (double** v : {&d.dv, &d.rsh, &d.y, &d.x})
    if ((rc = dalloc_(st, v, B * d.nsp, err, true))) return rc;
  for (double** v : {&d.part1, &d.part2, &d.partA, &d.partB})
    if ((rc = dalloc_(st, v, B * d.T * d.T * 64, err, true))) return rc;
  if ((rc = dalloc_(st, &d.cnt, B * 4, err, true))) return rc;
  if ((rc = dalloc_(st, &d.st, B, err, true))) return rc;
  return 0;
}

}  // namespace

int dense_create(DenseState** out, const DensePlan& P, int64_t batch, std::string& err, bool use_graph, int syrk_wgs, int panel_blocks) {
  DenseState* st = new DenseState();
  st->batch = batch;
  st->use_graph = use_graph;
  st->panel2 = panel_blocks != 0; st->panel2_auto = panel_blocks == 1;
  *out = st;
  DnDev& d = st->d;
  d.n = P.n; d.m = P.m; d.p = P.p; d.nnz = P.nnz;
  d.mp = ((P.m + 63) / 64) * 64;
  d.Tn = (P.n + TS - 1) / TS; d.npj = d.Tn * TS;
  d.ntl = d.Tn * (d.Tn + 1) / 2;
  d.npos_ok = P.n;
  int rc;
  if ((rc = setup_common(st, P.n + P.p, P.n, batch, err))) return rc;
  // J'WJ work partition: the (problem, tile, row chunk) space in equal shares, two workgroups per CU
  std::vector<int> sy_tp;
  int npieces = 0;
  {
    d.Tm = (d.Tn + 1) / 2; d.nml = d.Tm * (d.Tm + 1) / 2;
    const long long nch = d.mp / KT, units = (long long)batch * d.nml * nch;
    int nwg = 512;
    if (syrk_wgs > 0) nwg = syrk_wgs;
    nwg = (int)std::min<long long>(nwg, units);
    std::vector<int> wg(nwg + 1, 0), pb, ptl, pc0, pc1;
    for (int w = 0; w < nwg; w++) {
      long long u = units * w / nwg;
      const long long u1 = units * (w + 1) / nwg;
      while (u < u1) {
        const long long tile = u / nch, c0 = u % nch, c1 = std::min(nch, c0 + (u1 - u));
        pb.push_back((int)(tile / d.nml)); ptl.push_back((int)(tile % d.nml)); pc0.push_back((int)c0); pc1.push_back((int)c1);
        u += c1 - c0;
      }
      wg[w + 1] = (int)pb.size();
    }
    npieces = (int)pb.size();
    sy_tp.assign((size_t)batch * d.nml + 1, 0);
    for (int q = 0; q < npieces; q++) sy_tp[(size_t)pb[q] * d.nml + ptl[q] + 1]++;
    for (size_t q = 0; q + 1 < sy_tp.size(); q++) sy_tp[q + 1] += sy_tp[q];  // pieces are generated in (problem, tile, chunk) order
    d.ks = nwg;
    if ((rc = upload_(st, &d.sy_wg, wg, err))) return rc;
    if ((rc = upload_(st, &d.sy_b, pb, err))) return rc;
    if ((rc = upload_(st, &d.sy_tl, ptl, err))) return rc;
    if ((rc = upload_(st, &d.sy_ch0, pc0, err))) return rc;
    if ((rc = upload_(st, &d.sy_ch1, pc1, err))) return rc;
    if ((rc = upload_(st, &d.sy_tp, sy_tp, err))) return rc;
  }
  // H entries grouped by lower tile and unique position, COO order inside a position
  {
    const int64_t ns = d.ns;
    std::vector<int> order(P.hslot.size());
    std::iota(order.begin(), order.end(), 0);
    auto tkey = [&](int e) {
      const int64_t pos = P.hpos[e], i = pos % ns, j = pos / ns;
      const int64_t I = i >> 6, J = j >> 6;
      const int64_t lt = J * d.T - (J * (J - 1)) / 2 + (I - J);
      return (lt << 12) | ((i & 63) + 64 * (j & 63));
    };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tkey(a) < tkey(b); });
    std::vector<int> ht_ptr(d.nlt + 1, 0), hu_loc, hu_ptr, hslot;
    int64_t prev = -1;
    for (int e : order) {
      const int64_t key = tkey(e);
      if (key != prev) {
        hu_loc.push_back((int)(key & 4095)); hu_ptr.push_back((int)hslot.size());
        ht_ptr[(key >> 12) + 1]++;
        prev = key;
      }
      hslot.push_back(P.hslot[e]);
    }
    hu_ptr.push_back((int)hslot.size());
    for (int t = 0; t < d.nlt; t++) ht_ptr[t + 1] += ht_ptr[t];
    if ((rc = upload_(st, &d.ht_ptr, ht_ptr, err))) return rc;
    if ((rc = upload_(st, &d.hu_loc, hu_loc, err))) return rc;
    if ((rc = upload_(st, &d.hu_ptr, hu_ptr, err))) return rc;
    if ((rc = upload_(st, &d.hslot, hslot, err))) return rc;
  }
  if ((rc = upload_(st, &d.jslot, P.jslot, err))) return rc;
  if ((rc = upload_(st, &d.dslot, P.dslot, err))) return rc;
  const size_t B = (size_t)batch;
  if ((rc = dalloc_(st, &d.Jd, B * d.mp * d.npj, err, true))) return rc;
  if ((rc = dalloc_(st, &d.Jw, B * d.mp * d.npj, err, true))) return rc;
  if ((rc = dalloc_(st, &d.w, B * d.mp, err, true))) return rc;
  if ((rc = dalloc_(st, &d.slab, (size_t)npieces * 4 * TT, err))) return rc;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(dn_syrk2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * MT * LDK * sizeof(double))) != hipSuccess) {
    err = "hipFuncSetAttribute(dn_syrk2) failed";
    return 4;
  }
  if ((rc = dalloc_(st, &d.jxp, B * (d.npj / 32) * d.mp, err))) return rc;
  return 0;
}

int dense_create_general(DenseState** out, int32_t ns, int32_t nv, int32_t nslots, const int32_t* d_pos, int64_t batch, std::string& err,
                         bool use_graph, int panel_blocks) {
  DenseState* st = new DenseState();
  st->batch = batch;
  st->use_graph = use_graph;
  st->panel2 = panel_blocks != 0; st->panel2_auto = panel_blocks == 1;
  st->general = true;
  *out = st;
  DnDev& d = st->d;
  d.n = ns; d.m = 0; d.p = 0; d.nnz = 0; d.mp = 0; d.npj = 0; d.Tn = 0; d.ntl = 0; d.ks = 1;
  d.npos_ok = nv;
  d.nslots = nslots; d.gpos = d_pos;
  int rc;
  if ((rc = setup_common(st, ns, nv, batch, err))) return rc;
  return 0;
}

void dense_destroy(DenseState* st) {
  if (!st) return;
  for (auto& g : st->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
  if (st->cap) (void)hipStreamDestroy(st->cap);
  for (void* p : st->allocs) (void)hipFree(p);
  delete st;
}

namespace {

// the per-column steps of one factorisation of S (all problems), then the decision
int enqueue_factor(DenseState* st, double eig_tol, hipStream_t stream, std::string& err) {
  const DnDev& d = st->d;
  const int B = (int)st->batch;
  // (Round 3: the factorisations of different problems as parallel branches — streams forked and joined with events, parallel
  //  branches of the graph — so that one problem's latency-bound panel overlaps the others' updates: measured at 8 problems
  //  1.109 ms in lockstep, 1.090 with two branches, 1.175 with four, 1.45 with eight.  Not kept.)
  for (int k = 0; k < d.T; k++) {
    if (st->panel2) hipLaunchKernelGGL(dn_panel2, dim3(d.T, B), dim3(512), sizeof(Panel2Lds), stream, d, k, eig_tol);
    else hipLaunchKernelGGL(dn_panel, dim3((d.T + 2) / 3, B), dim3(256), 0, stream, d, k, eig_tol, 0);
    const int nt = d.T - 1 - k, ntr = nt * (nt + 1) / 2 + (k + 1) * nt;
    if (ntr > 0) hipLaunchKernelGGL(dn_update, dim3(4 * ntr, B), dim3(256), 0, stream, d, k, 0);
  }
  DCHK(hipGetLastError());
  return 0;
}

int enqueue_solve_core(DenseState* st, int gate, hipStream_t stream, std::string& err) {
  const DnDev& d = st->d;
  const int B = (int)st->batch;
  hipLaunchKernelGGL(dn_gt_part, dim3(d.nlt, B), dim3(256), 0, stream, d, 0, gate);
  hipLaunchKernelGGL(dn_g_part, dim3(d.nlt, B), dim3(256), 0, stream, d, gate);
  hipLaunchKernelGGL(dn_resid_part, dim3(d.nlt, B), dim3(256), 0, stream, d, gate);
  hipLaunchKernelGGL(dn_gt_part, dim3(d.nlt, B), dim3(256), 0, stream, d, 1, gate);
  hipLaunchKernelGGL(dn_g_part, dim3(d.nlt, B), dim3(256), 0, stream, d, gate);
  hipLaunchKernelGGL(dn_xfinal, dim3(d.T, B), dim3(64), 0, stream, d, gate);
  DCHK(hipGetLastError());
  return 0;
}

}  // namespace

namespace {
int dense_enqueue(DenseState* st, int mode, double* vals, const double* rhs, double* dout, double* rho_old, double* rho, int32_t* nfact,
                  int32_t* success, int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err);
int dense_enqueue_general(DenseState* st, const GeneralOps& G, int mode, const double* cbuf, const int* xpos, const int* xzer, double* d2,
                          double* vals_rho0, int64_t vals_stride, double* rho_old, double* rho, int32_t* nfact, int32_t* success,
                          int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err);

// replay the cached graph of this argument set, or capture it first (fallback: plain launches)
template <class F>
int run_cached(DenseState* st, const GraphKey& key, hipStream_t stream, std::string& err, F enqueue) {
  if (!st->use_graph) return enqueue(stream);
  for (auto& g : st->graphs)
    if (g.key == key) {
      st->misses = 0;
      DCHK(hipGraphLaunch(g.exec, stream));
      return 0;
    }
  // Capture + instantiation cost far more than the launches they replace: a caller whose pointers change with every call
  // (fresh device buffers per step) would pay them each time.  After a few misses in a row such a handle stays on plain
  // launches (a hit resets the count, so alternating between a few fixed argument sets keeps its graphs).
  if (++st->misses > 4) return enqueue(stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return enqueue(stream);  // the caller captures
  if (!st->cap && hipStreamCreateWithFlags(&st->cap, hipStreamNonBlocking) != hipSuccess) { st->use_graph = false; return enqueue(stream); }
  if (hipStreamBeginCapture(st->cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    if (getenv("CNL_VERBOSE")) fprintf(stderr, "[cnl] dense: hipStreamBeginCapture failed, plain launches\n");
    (void)hipGetLastError(); st->use_graph = false; return enqueue(stream);
  }
  const int rc = enqueue(st->cap);
  GraphEntry ge;
  ge.key = key;
  const hipError_t e1 = hipStreamEndCapture(st->cap, &ge.graph);
  if (rc) { if (ge.graph) (void)hipGraphDestroy(ge.graph); return rc; }
  hipError_t e2 = hipSuccess;
  if (e1 != hipSuccess || (e2 = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0)) != hipSuccess) {
    if (getenv("CNL_VERBOSE")) fprintf(stderr, "[cnl] dense: graph capture failed (%s / %s), plain launches\n", hipGetErrorString(e1), hipGetErrorString(e2));
    (void)hipGetLastError();
    if (ge.graph) (void)hipGraphDestroy(ge.graph);
    st->use_graph = false;
    return enqueue(stream);
  }
  if (st->graphs.size() >= 8) {
    // the evicted graph may still be running on whatever stream its last caller used (the _dev calls are asynchronous):
    // drain the device first (rare: the ninth distinct argument set of a handle)
    (void)hipDeviceSynchronize();
    (void)hipGraphExecDestroy(st->graphs.front().exec);
    (void)hipGraphDestroy(st->graphs.front().graph);
    st->graphs.erase(st->graphs.begin());
  }
  st->graphs.push_back(ge);
  if (getenv("CNL_VERBOSE")) fprintf(stderr, "[cnl] dense: call sequence captured as a graph (%zu cached)\n", st->graphs.size());
  DCHK(hipGraphLaunch(ge.exec, stream));
  return 0;
}
}  // namespace

int dense_run(DenseState* st, const DensePlan& P, int mode, double* vals, const double* rhs, double* dout, double* rho_old, double* rho,
              int32_t* nfact, int32_t* success, int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream,
              std::string& err) {
  (void)P;
  if (mode == 2 && !st->factored) { err = "cnl_solve before cnl_factorize"; return 5; }
  GraphKey key;
  key.mode = mode;
  const void* ps[10] = {vals, rhs, dout, rho_old, rho, nfact, success, npos, nzero, nullptr};
  for (int i = 0; i < 10; i++) key.p[i] = ps[i];
  for (int i = 0; i < 9; i++) key.params[i] = params[i];
  const int rc = run_cached(st, key, stream, err, [&](hipStream_t s2) {
    return dense_enqueue(st, mode, vals, rhs, dout, rho_old, rho, nfact, success, npos, nzero, params, s2, err);
  });
  if (!rc && mode != 2) st->factored = true;
  return rc;
}

int dense_run_general(DenseState* st, const GeneralOps& G, int mode, const double* cbuf, const int* xpos, const int* xzer, double* d2,
                      double* vals_rho0, int64_t vals_stride, double* rho_old, double* rho, int32_t* nfact, int32_t* success,
                      int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err) {
  if (mode == 2 && !st->factored) { err = "cnl_solve before cnl_factorize"; return 5; }
  GraphKey key;
  key.mode = 16 + mode;
  const void* ps[10] = {cbuf, xpos, xzer, d2, vals_rho0, rho_old, rho, nfact, success, npos};
  for (int i = 0; i < 10; i++) key.p[i] = ps[i];
  for (int i = 0; i < 9; i++) key.params[i] = params[i];
  key.extra = (long long)vals_stride ^ ((long long)(uintptr_t)nzero << 1);
  const int rc = run_cached(st, key, stream, err, [&](hipStream_t s2) {
    return dense_enqueue_general(st, G, mode, cbuf, xpos, xzer, d2, vals_rho0, vals_stride, rho_old, rho, nfact, success, npos, nzero, params, s2, err);
  });
  if (!rc && mode != 2) st->factored = true;
  return rc;
}

namespace {
int dense_enqueue(DenseState* st, int mode, double* vals, const double* rhs, double* dout, double* rho_old, double* rho, int32_t* nfact,
                  int32_t* success, int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err) {
  const DnDev& d = st->d;
  const int B = (int)st->batch;
  const int N = d.n + d.m + d.p;
  const double eig_tol = params[0];
  int rc;
  if (mode != 2) {
    hipLaunchKernelGGL(dn_gather, dim3(std::min(2048, blocks((long long)d.m * d.n)), B), dim3(256), 0, stream, d, vals, eig_tol);
    hipLaunchKernelGGL(dn_syrk2, dim3(d.ks), dim3(256), 2 * MT * LDK * sizeof(double), stream, d);
    hipLaunchKernelGGL(dn_assemble, dim3(d.nlt, B), dim3(256), 0, stream, d, vals);
    if ((rc = enqueue_factor(st, eig_tol, stream, err))) return rc;
    if (mode == 0)
      hipLaunchKernelGGL(dn_ladder, dim3(B), dim3(LNW * 64), 0, stream, d, vals, rho_old, eig_tol, params[2], params[3], params[4], params[5],
                         params[6], params[7]);
    else
      hipLaunchKernelGGL(dn_decide, dim3(blocks(B)), dim3(256), 0, stream, d, B, mode, nullptr, success, npos, nzero);
    DCHK(hipGetLastError());
#ifdef DN_STAMPS
    if (mode == 1) {
      (void)hipStreamSynchronize(stream);
      std::vector<long long> hs(d.T * 4);
      (void)hipMemcpy(hs.data(), d.jxp, hs.size() * sizeof(long long), hipMemcpyDeviceToHost);
      for (int k2 = 0; k2 < d.T; k2++) fprintf(stderr, "[dn stamps] k=%d diag %lld rows %lld %lld %lld\n", k2, hs[k2 * 4], hs[k2 * 4 + 1], hs[k2 * 4 + 2], hs[k2 * 4 + 3]);
      std::vector<long long> h2(d.T * 8);
      (void)hipMemcpy(h2.data(), reinterpret_cast<long long*>(d.jxp) + d.T * 4, h2.size() * sizeof(long long), hipMemcpyDeviceToHost);
      for (int k2 = 0; k2 < d.T; k2++) fprintf(stderr, "[dn stamps2] k=%d diag blocks %lld %lld %lld %lld rows %lld %lld %lld %lld\n", k2, h2[k2 * 8], h2[k2 * 8 + 1], h2[k2 * 8 + 2], h2[k2 * 8 + 3], h2[k2 * 8 + 4], h2[k2 * 8 + 5], h2[k2 * 8 + 6], h2[k2 * 8 + 7]);
    }
#endif
    if (mode == 1) return 0;
  }
  const int gate = mode == 0 ? 1 : 0;
  hipLaunchKernelGGL(dn_yrhs, dim3((d.nsp + 3) / 4, B), dim3(256), 0, stream, d, rhs, gate);
  if ((rc = enqueue_solve_core(st, gate, stream, err))) return rc;
  hipLaunchKernelGGL(dn_jx, dim3(d.mp / 256 + (d.mp % 256 ? 1 : 0), d.npj / 32, B), dim3(256), 0, stream, d, gate);
  hipLaunchKernelGGL(dn_out, dim3(blocks(N), B), dim3(256), 0, stream, d, rhs, dout, mode, rho_old, rho, nfact, success);
  DCHK(hipGetLastError());
  return 0;
}

int dense_enqueue_general(DenseState* st, const GeneralOps& G, int mode, const double* cbuf, const int* xpos, const int* xzer, double* d2,
                          double* vals_rho0, int64_t vals_stride, double* rho_old, double* rho, int32_t* nfact, int32_t* success,
                          int64_t* npos, int64_t* nzero, const double params[9], hipStream_t stream, std::string& err) {
  DnDev d = st->d;
  const int B = (int)st->batch;
  const double eig_tol = params[0];
  int rc;
  if (mode != 2) {
    hipLaunchKernelGGL(dn_set_counts, dim3(blocks(B)), dim3(256), 0, stream, d, B, xpos, xzer);
    DCHK(hipMemsetAsync(d.S0, 0, (size_t)B * d.T * d.T * TT * sizeof(double), stream));
    hipLaunchKernelGGL(dn_scatter_general, dim3(std::min(1024, blocks(std::max(d.nslots, d.nsp))), B), dim3(256), 0, stream, d, cbuf, (long long)G.cstride);
    hipLaunchKernelGGL(dn_shift, dim3(d.nlt, B), dim3(256), 0, stream, d);
    if ((rc = enqueue_factor(st, eig_tol, stream, err))) return rc;
    if (mode == 0) {
      // the ladder writes the last rho tried into the caller's rho slots: vals_rho0 + b * vals_stride, nv entries
      DnDev dl = d;
      dl.nnz = (int)vals_stride;  // dn_ladder addresses vals + b * nnz + (nnz - nv)
      hipLaunchKernelGGL(dn_ladder, dim3(B), dim3(LNW * 64), 0, stream, dl, vals_rho0 - (vals_stride - d.nv), rho_old, eig_tol, params[2], params[3],
                         params[4], params[5], params[6], params[7]);
    } else
      hipLaunchKernelGGL(dn_decide, dim3(blocks(B)), dim3(256), 0, stream, d, B, mode, nullptr, success, npos, nzero);
    DCHK(hipGetLastError());
    if (mode == 1) return 0;
  }
  const int gate = mode == 0 ? 1 : 0;
  hipLaunchKernelGGL(dn_yrhs_general, dim3(std::min(256, blocks(d.nsp)), B), dim3(256), 0, stream, d, cbuf, (long long)G.cstride, gate);
  if ((rc = enqueue_solve_core(st, gate, stream, err))) return rc;
  hipLaunchKernelGGL(dn_out_general, dim3(blocks(d.ns), B), dim3(256), 0, stream, d, d2, mode, rho_old, rho, nfact, success);
  DCHK(hipGetLastError());
  return 0;
}

}  // namespace

}  // namespace cnl
