// band.hip — newton_system! (/root/reference/src/CaNNOLeS.jl:1008-1052) of a batch of band-structured problems as a
// sliding-window elimination with ONE LANE per (problem, part): executes the band program of band.h / band.cpp.
//
// Mapping.  A workgroup serves NL problems with one wavefront per part of the chain (two parts: the first eliminates upwards
// from variable 0, the second downwards from variable n-1, the four variables between them are the junction).  A wavefront
// has two roles that alternate:
//  * mover — all 64 lanes: lane (lq, le) = (lane / 8, lane % 8) moves element le of the 64-byte pieces of problems lq, lq + 8, ...
//    between HBM and the problems' LDS blocks: the operand pieces of the NEXT epoch are loaded into registers while the current
//    epoch computes (the registers are the look-ahead buffer), written to LDS at the next epoch's start, and the factor records /
//    solution components an epoch produced go out the same way.  Every vector-memory instruction moves eight 64-byte runs.
//  * compute — lanes 0 .. NL-1, lane = problem: the window (5 band slots x 5, one border row, the right-hand side: 27 doubles) lives
//    in registers, every operand is one ds_read_b64 at an offset the generator fixed, every update a plain v_fma_f64.  No
//    cross-lane operation, no LDS atomics, no index decode: the step blocks are wave-uniform and come through the scalar cache.
// Summation order (documented deviation, within the fp64 bar of DESIGN section 5): entering position = plain entries in COO
// order (src/solver_types.jl:53-59: duplicates summed in COO order), then the condensed rows' products in row order, then the
// pivots' updates as they happen.
#include <hip/hip_runtime.h>

#include "band.h"
#include "kernels.h"

// timing probes (results wrong or missing): 1 no backward sweep, 2 no arithmetic in the forward steps, 4 none in the backward steps,
// 8 no operand loads after the first epoch, 16 no factor / solution stores
#ifndef BAND_DBG
#define BAND_DBG 0
#endif
// probe: every operand piece / factor flush moved down to a multiple of eight doubles (WRONG results; what perfectly aligned streams
// would be worth — build with -DBAND_DBG=1 so that garbage triggers no ladder)
// probe: only every other operand load instruction is issued (WRONG results: what half the vector-memory LOAD instructions would be worth)
#ifndef BAND_PROBE_HALF_LOADS
#define BAND_PROBE_HALF_LOADS 0
#endif
#ifdef BAND_PROBE_ALIGNED
constexpr int BAND_ALIGN_MASK = ~7;
#else
constexpr int BAND_ALIGN_MASK = ~0;
#endif
#if (BAND_DBG || defined(BAND_STAMPS) || defined(BAND_PROBE_ALIGNED) || BAND_PROBE_HALF_LOADS) && !defined(CNL_EXPERIMENT)
#error "BAND_DBG needs -DCNL_EXPERIMENT=1"
#endif

#ifdef BAND_STAMPS   // diagnostic: time per phase of an epoch (s_memtime ticks, summed), written over d[0 .. 15] of the workgroup's first problem
#define BSTAMP_DECL unsigned long long bst_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bst_t = __builtin_amdgcn_s_memtime();
#define BSTAMP(K) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); bst_[K] += t_ - bst_t; bst_t = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define BSTAMP_DECL
#define BSTAMP(K)
#endif

namespace cnl {

namespace {

template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

typedef const __attribute__((address_space(4))) int* cptr;   // program blocks: wave-uniform, read through the scalar cache
__device__ __forceinline__ cptr as_const(const int* p) { return (cptr)(const __attribute__((address_space(1))) int*)p; }

constexpr int NPC = BAND_NPIECE;
constexpr int LANE_D = BAND_LANE_DOUBLES;
constexpr int EXCH_OFF = BAND_IN_OFF;   // junction exchange (46 doubles) in a lane block: over the operand pieces, idle between the sweeps
static_assert(48 <= BAND_NPIECE * 8, "exchange area inside the operand pieces");

__device__ __forceinline__ constexpr int sidx(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// refined reciprocal (v_rcp_f64 is good to 2^-25 on gfx950: one Newton step) and a quotient with a residual correction: the same
// division the register-front kernel uses (kernels2.hip, fast_div), the reciprocal shared by the multipliers of one pivot
__device__ __forceinline__ double rrcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ double rdiv(double w, double d, double r) {
  const double q = w * r;
  const double res = fma(-d, q, w);
  return fma(res, r, q);
}

// The window: BAND_NS slots, of which the five that hold the variables entered last are live (the allocator sees that: every
// index below is a compile-time constant, so the dead slots' registers are free).
constexpr int NS = BAND_NS;
struct Win {
  double S[NS * (NS + 1) / 2];   // band slots, packed lower triangle
  double X[NS];                  // border row
  double S55;
  double c[NS];                  // right-hand side
  double c5;
};
// live slot k of step phase PH: 0 = the step's pivot .. BAND_HW = the entering variable
__device__ __forceinline__ constexpr int lslot(int PH, int k) { return (PH - BAND_HW + k + NS) % NS; }

#define LDSD(off) (*reinterpret_cast<const double*>(myb + (off)))
#define LDSW(off) (*reinterpret_cast<double*>(myb + (off)))

// Step and row blocks of the current epoch sit in the wavefront's LDS record buffer (copied there by the mover with the operand
// pieces): every compute lane reads the same address (one broadcast read), the operand offsets stay in VGPRs — they are only ever
// added to the lane's LDS base — and the flags word goes to an SGPR for the wave-uniform branches.
struct Rec { int v[BAND_SW]; };
struct RowRec { int v[BAND_RW]; };
__device__ __forceinline__ void load_rec(Rec& R, const char* recb, int o) {
  const int4* q = reinterpret_cast<const int4*>(recb + 4 * o);
#pragma unroll
  for (int k = 0; k < BAND_SW / 4; k++) { const int4 t = q[k]; R.v[4 * k] = t.x; R.v[4 * k + 1] = t.y; R.v[4 * k + 2] = t.z; R.v[4 * k + 3] = t.w; }
}
__device__ __forceinline__ void load_row(RowRec& R, const char* recb, int o) {
  const int4* q = reinterpret_cast<const int4*>(recb + 4 * o);
#pragma unroll
  for (int k = 0; k < BAND_RW / 4; k++) { const int4 t = q[k]; R.v[4 * k] = t.x; R.v[4 * k + 1] = t.y; R.v[4 * k + 2] = t.z; R.v[4 * k + 3] = t.w; }
}

// operands of one forward step: the entering variable's fifteen entries and the first residual row
struct FOps { double eo[15]; double rj[BAND_NB]; double rdr, rrr; };
__device__ __forceinline__ void fload(FOps& P, const Rec& st, const RowRec& row0, const int fl, char* myb) {
#pragma unroll
  for (int q = 0; q < 15; q++) P.eo[q] = LDSD(st.v[BS_DG0 + q]);
  static_assert(BS_RX == BS_DG0 + 14, "the fifteen operands of an entering variable are consecutive words of the step block");
  P.rdr = 1.0; P.rrr = 0.0;
#pragma unroll
  for (int k = 0; k < BAND_NB; k++) P.rj[k] = 0.0;
  if ((fl >> 8) & 255) {
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) P.rj[k] = LDSD(row0.v[BR_J0 + k]);
    P.rdr = LDSD(row0.v[BR_DI]);
    P.rrr = LDSD(row0.v[BR_RR]);
  }
}

// offset of element e of a problem from the problem's first element: e for the ABI's problem-major arrays (stride 0); for arrays
// interleaved over the NL problems of a workgroup in blocks of eight doubles (stride = NL * 8) block e / 8 is NL * 8 doubles further
__device__ __forceinline__ long long band_il_offset(int e, int stride) { return stride ? (long long)(e >> 3) * stride + (e & 7) : (long long)e; }

// one forward step of phase PH = step number % 8: enter slot PH, pivot slot PH - 4; fl = flags word (wave-uniform), o = int
// offset of the step block in the record buffer.  The operands were read (fload) while the previous step computed.
template <int PH>
__device__ __forceinline__ void fstep(Win& W, const FOps& OP, const Rec& st, const int fl, const char* recb, const int o, char* myb,
                                      const double* __restrict__ gvals, const double* __restrict__ grhs, cptr borders, long long pv, long long pr,
                                      bool has_rhs, double rho, bool ovr, double tol, int& npos, int& nzer, const int vstride = 0, const int rstride = 0) {
  constexpr int es = PH, ps = lslot(PH, 0);
  const int nrows = (fl >> 8) & 255;
  const double (&eo)[15] = OP.eo;
  const double (&rj)[BAND_NB] = OP.rj;
  const double rdr = OP.rdr, rrr = OP.rrr;
  // ---- enter ----
  {
    const double dg = (eo[0] + eo[1]) + eo[2];
    double rv = eo[BS_RHO - BS_DG0];
    if (!(fl & (1 << 16))) rv = ovr ? rho : rv;
    W.S[sidx(es, es)] = dg + rv;
#pragma unroll
    for (int k = 1; k <= BAND_HW; k++) W.S[sidx(es, lslot(PH, BAND_HW - k))] = eo[BS_OD - BS_DG0 + 2 * (k - 1)] + eo[BS_OD - BS_DG0 + 2 * (k - 1) + 1];
    W.X[es] = eo[BS_BC0 - BS_DG0] + eo[BS_BC1 - BS_DG0];
    W.c[es] = eo[BS_RX - BS_DG0];
  }
  // ---- residual rows completed by the entering variable: products -J_a J_b / d_r, counted in the inertia ----
  for (int i = 0; i < nrows; i++) {
    double J[BAND_NB], dr, rr;
    if (i == 0) {
#pragma unroll
      for (int k = 0; k < BAND_NB; k++) J[k] = rj[k];
      dr = rdr; rr = rrr;
    } else {
      RowRec rb;
      load_row(rb, recb, o + BAND_SW + BAND_RW * i);
#pragma unroll
      for (int k = 0; k < BAND_NB; k++) J[k] = LDSD(rb.v[BR_J0 + k]);
      dr = LDSD(rb.v[BR_DI]);
      rr = LDSD(rb.v[BR_RR]);
    }
    npos += dr > tol;
    nzer += fabs(dr) <= tol;
    const double r = rrcp(dr);
    const double w = rdiv(-1.0, dr, r);
    const double tr = rr * w;
#pragma unroll
    for (int ka = 0; ka < BAND_NB; ka++) {
      const double ta = J[ka] * w;
#pragma unroll
      for (int kb = 0; kb <= ka; kb++) W.S[sidx(lslot(PH, ka), lslot(PH, kb))] = fma(ta, J[kb], W.S[sidx(lslot(PH, ka), lslot(PH, kb))]);
      W.c[lslot(PH, ka)] = fma(tr, J[ka], W.c[lslot(PH, ka)]);
    }
  }
  // ---- border pivot ----
  if (fl & BF_PIVOT_B) {
    cptr bt = borders + BAND_BW * __builtin_amdgcn_readfirstlane(st.v[BS_BORDER]);
    // (vstride / rstride != 0: the array is interleaved over the workgroup's problems in blocks of eight doubles, see band_il_offset)
    W.S55 += gvals[pv + band_il_offset(bt[BB_DSRC], vstride)];
    W.c5 += has_rhs ? grhs[pr + band_il_offset(bt[BB_RHS], rstride)] : 0.0;
    const double d = W.S55;
    npos += d > tol;
    nzer += fabs(d) <= tol;
    const double r = rrcp(d);
    double l[BAND_NB], w[BAND_NB];
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) { w[k] = W.X[lslot(PH, k)]; l[k] = rdiv(w[k], d, r); }
    const double z = rdiv(W.c5, d, r);
#pragma unroll
    for (int ka = 0; ka < BAND_NB; ka++) {
#pragma unroll
      for (int kb = 0; kb <= ka; kb++) W.S[sidx(lslot(PH, ka), lslot(PH, kb))] = fma(w[ka], -l[kb], W.S[sidx(lslot(PH, ka), lslot(PH, kb))]);
      W.c[lslot(PH, ka)] = fma(w[ka], -z, W.c[lslot(PH, ka)]);
    }
    double* lo = reinterpret_cast<double*>(myb + st.v[BS_LB]);
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) lo[k] = l[k];
    lo[BAND_NB] = z;
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) W.X[lslot(PH, k)] = 0.0;
    W.S55 = 0.0; W.c5 = 0.0;
  }
  // ---- band pivot ----
  if (fl & BF_PIVOT_X) {
    const double d = W.S[sidx(ps, ps)];
    npos += d > tol;
    nzer += fabs(d) <= tol;
    const double r = rrcp(d);
    double l[BAND_NB], w[BAND_NB];
#pragma unroll
    for (int k = 1; k < BAND_NB; k++) { w[k] = W.S[sidx(lslot(PH, k), ps)]; l[k] = rdiv(w[k], d, r); }
    const double w5 = W.X[ps], l5 = rdiv(w5, d, r), z = rdiv(W.c[ps], d, r);
#pragma unroll
    for (int ka = 1; ka < BAND_NB; ka++) {
#pragma unroll
      for (int kb = 1; kb <= ka; kb++) W.S[sidx(lslot(PH, ka), lslot(PH, kb))] = fma(w[ka], -l[kb], W.S[sidx(lslot(PH, ka), lslot(PH, kb))]);
      W.X[lslot(PH, ka)] = fma(w5, -l[ka], W.X[lslot(PH, ka)]);
      W.c[lslot(PH, ka)] = fma(w[ka], -z, W.c[lslot(PH, ka)]);
    }
    W.S55 = fma(w5, -l5, W.S55);
    W.c5 = fma(w5, -z, W.c5);
    double* lo = reinterpret_cast<double*>(myb + st.v[BS_LX]);
#pragma unroll
    for (int k = 1; k < BAND_NB; k++) lo[k - 1] = l[k];
    lo[4] = l5;
    lo[5] = z;
  }
}

// one backward step of phase PH: x of the band pivot, then of the border pivot, then the residual components
struct BOps { double lx[BAND_LREC], lb[BAND_LREC], rj[BAND_NB]; double rdr, rrr; };
__device__ __forceinline__ void bload(BOps& P, const Rec& st, const RowRec& row0, const int fl, char* myb) {
#pragma unroll
  for (int q = 0; q < BAND_LREC; q++) { P.lx[q] = 0.0; P.lb[q] = 0.0; }
#pragma unroll
  for (int k = 0; k < BAND_NB; k++) P.rj[k] = 0.0;
  P.rdr = 1.0; P.rrr = 0.0;
  if (fl & BF_PIVOT_X) {
    const double* lo = reinterpret_cast<const double*>(myb + st.v[BS_LX]);
#pragma unroll
    for (int q = 0; q < BAND_LREC; q++) P.lx[q] = lo[q];
  }
  if (fl & BF_PIVOT_B) {
    const double* lo = reinterpret_cast<const double*>(myb + st.v[BS_LB]);
#pragma unroll
    for (int q = 0; q < BAND_LREC; q++) P.lb[q] = lo[q];
  }
  if ((fl >> 8) & 255) {
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) P.rj[k] = LDSD(row0.v[BR_J0 + k]);
    P.rdr = LDSD(row0.v[BR_DI]);
    P.rrr = LDSD(row0.v[BR_RR]);
  }
}
template <int PH>
__device__ __forceinline__ void bstep(double (&xs)[NS + 1], const BOps& OP, const Rec& st, const RowRec& row0, const int fl, const char* recb,
                                      const int o, char* myb, cptr borders, double* __restrict__ gd, long long pd, bool okme) {
  constexpr int ps = lslot(PH, 0);
  const int nrows = (fl >> 8) & 255;
  const double (&lx)[BAND_LREC] = OP.lx;
  const double (&lb)[BAND_LREC] = OP.lb;
  const double (&rj)[BAND_NB] = OP.rj;
  const double rdr = OP.rdr, rrr = OP.rrr;
  if (fl & BF_PIVOT_X) {
    double x = lx[5];
#pragma unroll
    for (int k = 1; k < BAND_NB; k++) x = fma(-lx[k - 1], xs[lslot(PH, k)], x);
    x = fma(-lx[4], xs[NS], x);
    xs[ps] = x;
    LDSW(st.v[BS_DX]) = -x;
  }
  if (fl & BF_PIVOT_B) {
    double x = lb[BAND_NB];
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) x = fma(-lb[k], xs[lslot(PH, k)], x);
    xs[NS] = x;
    if (okme) gd[pd + borders[BAND_BW * __builtin_amdgcn_readfirstlane(st.v[BS_BORDER]) + BB_DOUT]] = -x;
  }
  for (int i = 0; i < nrows; i++) {
    double J[BAND_NB], dr, rr;
    int dro;
    if (i == 0) {
#pragma unroll
      for (int k = 0; k < BAND_NB; k++) J[k] = rj[k];
      dr = rdr; rr = rrr; dro = row0.v[BR_DR];
    } else {
      RowRec rb;
      load_row(rb, recb, o + BAND_SW + BAND_RW * i);
#pragma unroll
      for (int k = 0; k < BAND_NB; k++) J[k] = LDSD(rb.v[BR_J0 + k]);
      dr = LDSD(rb.v[BR_DI]); rr = LDSD(rb.v[BR_RR]); dro = rb.v[BR_DR];
    }
    double acc = -rr;
#pragma unroll
    for (int k = 0; k < BAND_NB; k++) acc = fma(J[k], xs[lslot(PH, k)], acc);
    LDSW(dro) = rdiv(acc, dr, rrcp(dr));
  }
  if (fl & BF_ENTER_B) xs[NS] = 0.0;
}

}  // namespace

// LDS of a workgroup: [part][NL] lane blocks | [part] record buffers | control block: per problem [rho | flags], one word "all done"
template <int NL>
__global__ void __launch_bounds__(128, (NL <= 8 ? 2 : 1)) band_newton_kernel(const BandDev P, const LaunchArgs Ain) {
  constexpr int NI = NL / 8;
  extern __shared__ double lds[];
  const int mode = Ain.mode, batch = Ain.batch;
  double* const gvals = as_global(Ain.vals);
  const double* const grhs = as_global(Ain.rhs);
  double* const gd = as_global(Ain.d);
  double* const gL = as_global(Ain.L);
  const int lane = threadIdx.x & 63;
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lq = lane >> 3, le = lane & 7;
  const int prob0 = blockIdx.x * NL;
  // MODE_SOLVE = solve_ldl! behind a factorisation of this handle (src/solver_types.jl:69-77): the band kernels keep no factor a later
  // right-hand side could be run through (their six-double records hold z = c / d of the ONE right-hand side they were computed with),
  // so the solve factorises the same values again — the rho slots hold what the ladder left — and sweeps the new right-hand side
  // in the same launch: the arithmetic of the first attempt of newton_system!, no ladder, no outputs but d.
  const bool has_rhs = mode != MODE_FACTOR && grhs != nullptr;
  char* wblk = reinterpret_cast<char*>(lds + (size_t)part * NL * LANE_D);
  char* recb = reinterpret_cast<char*>(lds + (size_t)P.nparts * NL * LANE_D) + (size_t)part * BAND_REC_MAX * 4;
  double* ctrl = reinterpret_cast<double*>(reinterpret_cast<char*>(lds + (size_t)P.nparts * NL * LANE_D) + (size_t)P.nparts * BAND_REC_MAX * 4);
  const int* fops_g = as_global(P.fops[part]);
  const int* bops_g = as_global(P.bops[part]);
  cptr epochs = as_const(P.epochs[part]);
  cptr borders = as_const(P.borders[part]);
  const int nepochs = P.nepochs[part];
  const int nnz = P.nnz, N = P.N;
  const long long lsize = P.lsize;
  // mover: wave-uniform base pointers + 32-bit per-lane byte offsets (NL problems span < 4 GB)
  // cnl_options.batch_layout = 1: `vals` is given INTERLEAVED over groups of 32 problems in blocks of eight doubles, the layout of the
  // factor records below — element e of problem p of group g at
  // ((g * band_il_blocks(nnz) + e / 8) * 32 + p) * 8 + e % 8 — so that the eight 64-byte runs of a mover load are 512 contiguous bytes
  // and every 128-byte line that is fetched is used whole (16 384 problems: 12.5 -> 11.0 ms with vals and rhs interleaved, bit-equal).
  // Bit 1 of the layout word: the same for `rhs`.  d is problem-major always.
  // (the groups of the LAYOUT are 32 problems whatever the workgroup holds: a workgroup of 16 problems works on half a group)
  constexpr int G8 = BAND_IL_GROUP * 8;
  static_assert(BAND_IL_GROUP % NL == 0, "a workgroup's problems lie in one group of the interleaved layout");
  const bool vil = (Ain.layout & 1) != 0, ril = (Ain.layout & 2) != 0;
  const int vstride = vil ? G8 : 0, rstride = ril ? G8 : 0;
  const long long ilg = prob0 / BAND_IL_GROUP;        // group of the workgroup's problems
  const int ilp = (prob0 % BAND_IL_GROUP) * 8;        // ... and the offset of its first problem inside a block row
  const double* vbase = gvals + (vil ? ilg * band_il_blocks(nnz) * G8 + ilp : (long long)prob0 * nnz);
  const double* rbase = !has_rhs ? gvals : grhs + (ril ? ilg * band_il_blocks(N) * G8 + ilp : (long long)prob0 * N);
  // The factor records are private to the launch (written by the forward sweep, read by the backward sweep of the SAME workgroup), so
  // their layout is the kernel's choice: INTERLEAVED over the NL problems of the workgroup in blocks of eight doubles — element e of
  // problem p of the workgroup lives at ((e >> 3) * NL + p) * 8 + (e & 7) of the workgroup's region — so that the eight 64-byte runs one
  // store / load instruction touches are 512 contiguous bytes (tools/seg_bench.hip: this memory system gives scattered 64-byte
  // segments 3.4 ... 4.5 TB/s, runs of 256 bytes and more 6.0).
  // Measured on one box (tools/ab_lib.py): 16 384 problems, 32 per workgroup: 12.41 -> 12.22 ms; 8 192 problems, 16 per workgroup:
  // 7.40 -> 7.75 ms (the address arithmetic costs the latency-bound case more than the layout gives) — so only the 32-problem
  // instantiation interleaves.
  constexpr bool LINT = NL >= 32;
  double* lbase_g = gL + (long long)prob0 * lsize + (LINT ? 0 : P.loff[part]);
  const int loff8 = LINT ? (int)P.loff[part] : 0;   // a multiple of 8
  double* dbase = gd ? gd + (long long)prob0 * N : nullptr;
  unsigned movp[NI], ldsb[NI];   // problem of the lane inside the workgroup (clamped to the batch), LDS byte offset of its element
  bool movok[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    int pl = i * 8 + lq;
    movok[i] = prob0 + pl < batch;
    if (!movok[i]) pl = batch - 1 - prob0;
    movp[i] = (unsigned)pl;
    ldsb[i] = ((unsigned)(i * 8 + lq) * (unsigned)LANE_D + (unsigned)le) << 3;
  }
  // compute lanes
  const bool clane = lane < NL;
  const int cprob = prob0 + (clane ? lane : 0);
  const bool valid = clane && cprob < batch;
  const int cpl = valid ? cprob - prob0 : batch - 1 - prob0;   // problem whose data this lane's block holds
  char* myb = wblk + (size_t)(clane ? lane : 0) * LANE_D * 8;
  const long long pv = vil ? ilg * band_il_blocks(nnz) * G8 + ilp + cpl * 8 : (long long)(prob0 + cpl) * nnz;
  const long long pr = ril ? ilg * band_il_blocks(N) * G8 + ilp + cpl * 8 : (long long)(prob0 + cpl) * N;
  for (int t = lane; t < NL; t += 64) *reinterpret_cast<double*>(wblk + ((size_t)t * LANE_D + BAND_ZERO_OFF) * 8) = 0.0;   // every block's zero cell

  const double tol = Ain.params[0], kdec = Ain.params[2], kinc = Ain.params[3], klarge = Ain.params[4], rho0 = Ain.params[5], rhomax = Ain.params[6],
               rhomin = Ain.params[7];
  double rho = 0.0, wrote = 0.0;
  double rho_old = (mode == MODE_NEWTON && valid) ? as_global(Ain.rho_old)[cprob] : 0.0;
  int nfact = 0;
  bool done = !valid, success = false, ovr = false;

  static_assert(NI <= 4, "at most four problem groups per mover lane");
  constexpr bool BAND_ISSUE_ALWAYS = false;
  double stg[NPC][NI];   // operand pieces in flight
  int4 rstg0;            // step blocks in flight (one 16-byte word per lane: 256 ints)
  int pcs[NPC];          // piece descriptors of the epoch being loaded
  static_assert(BAND_REC_MAX <= 256, "record buffer: one dwordx4 per lane");
  // The mover is written for back-to-back issue: the fifteen piece descriptors of an epoch come with one scalar load, unused
  // pieces are skipped with a wave-uniform branch (a load that hits in cache costs the CU's memory pipeline what any other costs),
  // every staged piece is written to LDS whether used or not.  One load per piece and problem group: base pointer and stride of the
  // piece's array are selected with scalar instructions, the lane's offset is problem * stride + element.  (Three guarded loads
  // made the compiler form all three 64-bit addresses of every piece up front — 96 NI VGPRs; selecting among per-array offset
  // arrays made it index them in scratch memory; lambdas instead of macros put every captured variable into scratch.)
#if BAND_PROBE_HALF_LOADS == 2   /* one 16-byte load per lane instead of two 8-byte ones: the same lines, half the instructions (interleaved vals only) */
#define BAND_ISSUE1(K, I)                                                                                                     \
  if constexpr (I < NI && (I & 1) == 0) {                                                                                     \
    const double2 v2_ = *reinterpret_cast<const double2*>(pb + ((movp[I] * strd + tl + (unsigned)lane) << 3));                \
    stg[K][I] = v2_.x;                                                                                                        \
    if constexpr (I + 1 < NI) stg[K][I + 1] = v2_.y;                                                                          \
  }
#else
#define BAND_ISSUE1(K, I) if constexpr (I < NI && (!BAND_PROBE_HALF_LOADS || (I & 1) == 0)) stg[K][I] = *reinterpret_cast<const double*>(pb + ((movp[I] * strd + tl) << 3));
#endif
#define BAND_COMMIT1(K, I) if constexpr (I < NI) *reinterpret_cast<double*>(wblk + ldsb[I] + (BAND_IN_OFF + 8 * K) * 8) = stg[K][I];
#define BAND_ISSUE(K)                                                                                                         \
  if (BAND_ISSUE_ALWAYS || pcs[K] >= 0) {   /* (wave-uniform: an unused piece costs the memory pipeline what a used one does) */ \
    const int pc = pcs[K] >= 0 ? pcs[K] : 0;   /* (ALWAYS: an unused piece loads element 0 of vals — a static number of loads per epoch) */ \
    const int arr = pc >> 28;                                                                                                 \
    const int el_ = ((pc & ((1 << 28) - 1)) + (arr == 2 ? loff8 : 0)) & BAND_ALIGN_MASK;   /* (probe builds: -DBAND_PROBE_ALIGNED) */ \
    /* lane offset (doubles) = problem * strd + tl, tl = t + (t >> 3) * gap with t = m + element of the lane: the caller's arrays    \
       are problem-major (m = 0, t < 8: gap = 0), the factor is interleaved in blocks of eight (see lbase_g) */                \
    const bool il_ = arr == 0 ? vil : arr == 1 ? ril : LINT;   /* (wave-uniform) */                                           \
    const int ilw_ = arr == 2 ? NL * 8 : BAND_IL_GROUP * 8;    /* doubles per block row: the factor's own layout / the ABI's */  \
    const int m_ = il_ ? (el_ & 7) : 0;                                                                                       \
    const unsigned gap_ = il_ ? (unsigned)(ilw_ - 8) : 0u;                                                                    \
    const char* pb = (arr == 0 ? reinterpret_cast<const char*>(vbase) : arr == 1 ? reinterpret_cast<const char*>(rbase)       \
                                                                                 : reinterpret_cast<const char*>(lbase_g)) +   \
                     ((il_ ? (long long)(el_ >> 3) * ilw_ : (long long)el_) << 3);                                            \
    const unsigned strd = il_ ? 8u : arr == 0 ? (unsigned)nnz : arr == 1 ? (unsigned)N : (unsigned)lsize;                     \
    const unsigned t_ = (unsigned)m_ + (unsigned)le;                                                                          \
    const unsigned tl = t_ + (t_ >> 3) * gap_;                                                                                \
    BAND_ISSUE1(K, 0) BAND_ISSUE1(K, 1) BAND_ISSUE1(K, 2) BAND_ISSUE1(K, 3)                                                   \
  }
#define BAND_COMMIT(K) { BAND_COMMIT1(K, 0) BAND_COMMIT1(K, 1) BAND_COMMIT1(K, 2) BAND_COMMIT1(K, 3) }
  // (macros, not lambdas: a closure made the compiler keep every captured variable — the staging registers included — in scratch memory)
#define BAND_ISSUE_DESC(EP, OFS) { cptr E_ = (EP) + (OFS); _Pragma("unroll") for (int k_ = 0; k_ < NPC; k_++) pcs[k_] = E_[k_]; }
  /* the epoch's step / row blocks (the streams are padded: reading past the epoch's blocks is harmless) */
#define BAND_ISSUE_REC(OPS, OPOFF) { rstg0 = (reinterpret_cast<const int4*>((OPS) + (OPOFF)) + lane)[0]; }
#define BAND_ISSUE_ALL(EP, OFS, OPS, OPOFF)                                                                                   \
  {                                                                                                                           \
    BAND_ISSUE_DESC(EP, OFS)                                                                                                  \
    BAND_ISSUE(0) BAND_ISSUE(1) BAND_ISSUE(2) BAND_ISSUE(3) BAND_ISSUE(4) BAND_ISSUE(5) BAND_ISSUE(6) BAND_ISSUE(7)           \
    BAND_ISSUE(8) BAND_ISSUE(9) BAND_ISSUE(10) BAND_ISSUE(11) BAND_ISSUE(12) BAND_ISSUE(13) BAND_ISSUE(14)                    \
    BAND_ISSUE_REC(OPS, OPOFF)                                                                                                \
  }
#define BAND_COMMIT_ALL()                                                                                                     \
  {                                                                                                                           \
    BAND_COMMIT(0) BAND_COMMIT(1) BAND_COMMIT(2) BAND_COMMIT(3) BAND_COMMIT(4) BAND_COMMIT(5) BAND_COMMIT(6) BAND_COMMIT(7)   \
    BAND_COMMIT(8) BAND_COMMIT(9) BAND_COMMIT(10) BAND_COMMIT(11) BAND_COMMIT(12) BAND_COMMIT(13) BAND_COMMIT(14)             \
    reinterpret_cast<int4*>(recb)[lane] = rstg0;                                                                              \
  }
  static_assert(NPC == 15, "fifteen operand pieces");

  BSTAMP_DECL
  Win W;
  int npos = 0, nzer = 0;
  double lj[6], zj[4];   // junction factor (first wavefront)
  for (int q = 0; q < 6; q++) lj[q] = 0.0;
  for (int q = 0; q < 4; q++) zj[q] = 0.0;
  while (true) {
    // ================= forward: assembly, elimination, forward substitution =================
#pragma unroll
    for (int q = 0; q < NS * (NS + 1) / 2; q++) W.S[q] = 0.0;
#pragma unroll
    for (int q = 0; q < NS; q++) { W.X[q] = 0.0; W.c[q] = 0.0; }
    W.S55 = 0.0; W.c5 = 0.0;
    npos = 0; nzer = 0;
    BAND_ISSUE_ALL(epochs, BE_FP, fops_g, 0)
    for (int e = 0; e < nepochs; e++) {
      cptr E = epochs + e * BAND_EW;
      BSTAMP(0)
      BAND_COMMIT_ALL()
      BSTAMP(1)
      // the next epoch's loads are issued in four groups behind the first four steps (a burst of 34 loads stalled the wavefront on
      // the CU's memory pipeline for ~2 500 cycles per epoch, in-kernel stamps), still four steps ahead of their use
      const bool more_ = e + 1 < nepochs && !(BAND_DBG & 8);
      if (more_) BAND_ISSUE_DESC(epochs + (e + 1) * BAND_EW, BE_FP)
      // The steps of the epoch: step t works on the slots of phase t (every epoch but the last has BAND_EPOCH steps), so the eight
      // instantiations follow each other in straight-line code and the window keeps its registers from step to step.
      const int nst = E[BE_NSTEP];
      int o = 0;
      BSTAMP(2)
      // the blocks of step t + 1 (step block + first row block) are read from the record buffer while step t computes; the step
      // reads all its operands at its top (one LDS round trip).  (A third stage — operands a step ahead — was measured: no gain,
      // 170 more registers.)
      Rec stC, stN;
      RowRec rwC, rwN;
      load_rec(stC, recb, 0);
      load_row(rwC, recb, BAND_SW);
#define BAND_FSTEP(PHV)                                                                                                     \
      if (PHV < nst) {                                                                                                      \
        const int fl = __builtin_amdgcn_readfirstlane(stC.v[BS_FLAGS]);                                                     \
        const int onext = o + BAND_SW + BAND_RW * ((fl >> 8) & 255);                                                        \
        if (PHV + 1 < nst) { load_rec(stN, recb, onext); load_row(rwN, recb, onext + BAND_SW); }                            \
        if (clane && !(BAND_DBG & 2)) {                                                                                     \
          FOps op_;                                                                                                         \
          fload(op_, stC, rwC, fl, myb);                                                                                    \
          fstep<PHV>(W, op_, stC, fl, recb, o, myb, gvals, grhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer, vstride, rstride); \
        }                                                                                                                   \
        o = onext; stC = stN; rwC = rwN;                                                                                    \
      }
      // factor records: the out ring holds those of half an epoch; whole 64-byte pieces are read from it (all reads first), lanes
      // past the records do not store
      // (try_to_factorize keeps no records: a later solve_ldl! factorises again, see MODE_SOLVE above)
#define BAND_LFLUSH(LB, LC)                                                                                                 \
      if (mode != MODE_FACTOR) {                                                                                            \
        const int lc_ = (LC);                                                                                               \
        const int lb_ = ((LB) + loff8) & BAND_ALIGN_MASK;                                                                   \
        char* lout = reinterpret_cast<char*>(lbase_g) + ((LINT ? (long long)(lb_ >> 3) * (NL * 8) : (long long)lb_) << 3);  \
        const unsigned t_ = (LINT ? (unsigned)(lb_ & 7) : 0u) + (unsigned)le;                                               \
        const unsigned tl = t_ + (LINT ? (t_ >> 3) * (unsigned)(NL * 8 - 8) : 0u);                                          \
        const unsigned lstr_ = LINT ? 8u : (unsigned)lsize;                                                                 \
        constexpr int lcp_ = LINT ? NL * 64 : 64;                                                                           \
        double lx_[BAND_LOUT_MAX / 8][NI];                                                                                  \
        _Pragma("unroll") for (int cpc = 0; cpc < BAND_LOUT_MAX / 8; cpc++)                                                 \
          _Pragma("unroll") for (int i = 0; i < NI; i++) lx_[cpc][i] = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_LOUT_OFF + 8 * cpc) * 8); \
        _Pragma("unroll") for (int cpc = 0; cpc < BAND_LOUT_MAX / 8; cpc++)                                                 \
          _Pragma("unroll") for (int i = 0; i < NI; i++)                                                                    \
            if (movok[i] && cpc * 8 + le < lc_ && !(BAND_DBG & 16))                                                         \
              *reinterpret_cast<double*>(lout + (((movp[i] * lstr_ + tl) << 3) + lcp_ * cpc)) = lx_[cpc][i];                 \
      }
      BAND_FSTEP(0)
      if (more_) { BAND_ISSUE(0) BAND_ISSUE(1) BAND_ISSUE(2) BAND_ISSUE(3) }
      BAND_FSTEP(1)
      if (more_) { BAND_ISSUE(4) BAND_ISSUE(5) BAND_ISSUE(6) BAND_ISSUE(7) }
      BAND_FSTEP(2)
      if (more_) { BAND_ISSUE(8) BAND_ISSUE(9) BAND_ISSUE(10) BAND_ISSUE(11) }
      BAND_FSTEP(3)
      if (more_) { BAND_ISSUE(12) BAND_ISSUE(13) BAND_ISSUE(14) BAND_ISSUE_REC(fops_g, epochs[(e + 1) * BAND_EW + BE_FOFF]) }
      BAND_LFLUSH(E[BE_LBASE], E[BE_LCNT])
      BAND_FSTEP(4) BAND_FSTEP(5) BAND_FSTEP(6) BAND_FSTEP(7)
      BSTAMP(3)
      static_assert(BAND_EPOCH == 8, "eight step instantiations per epoch");
      if (nst == BAND_EPOCH) {
        // behind a full epoch the slots 0 .. 3 are dead (pivoted in phases 4 .. 7): give them a constant, so that only the ten
        // entries among the live slots, their border row and right-hand side are carried from epoch to epoch (the junction
        // behind the loop reads every entry, which would otherwise keep all 52 in registers through the whole sweep)
#pragma unroll
        for (int a = 0; a < NS; a++)
#pragma unroll
          for (int b = 0; b <= a; b++)
            if (b < 4) W.S[sidx(a, b)] = 0.0;
#pragma unroll
        for (int a = 0; a < 4; a++) { W.X[a] = 0.0; W.c[a] = 0.0; }
      }
      BAND_LFLUSH(E[BE_LBASE2], E[BE_LCNT2])
    }
    BSTAMP(4)
    // ================= junction + inertia rule + rho ladder (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047) ==========
    int tpos = npos, tzer = nzer;
    if (P.nparts == 2) {
      // windows to LDS (slot order): the junction reads both with run-time slot numbers
      if (clane) {
        double* ex = reinterpret_cast<double*>(myb + EXCH_OFF * 8);
#pragma unroll
        for (int q = 0; q < NS * (NS + 1) / 2; q++) ex[q] = W.S[q];
#pragma unroll
        for (int q = 0; q < NS; q++) ex[36 + q] = W.c[q];
        ex[44] = (double)npos; ex[45] = (double)nzer;
      }
      __syncthreads();
      if (part == 0 && clane) {
        const double* exL = reinterpret_cast<const double*>(myb + EXCH_OFF * 8);
        const double* exR = reinterpret_cast<const double*>(myb + (size_t)NL * LANE_D * 8 + EXCH_OFF * 8);
        const int tL = P.m0 % NS, tR = (P.n - 1 - P.m0) % NS;
        double SJ[10], cJ[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int aL = (tL + i) % NS, aR = (tR - i + NS) % NS;
#pragma unroll
          for (int j = 0; j <= i; j++) {
            const int bL = (tL + j) % NS, bR = (tR - j + NS) % NS;
            const int iL = aL >= bL ? aL * (aL + 1) / 2 + bL : bL * (bL + 1) / 2 + aL;
            const int iR = aR >= bR ? aR * (aR + 1) / 2 + bR : bR * (bR + 1) / 2 + aR;
            SJ[i * (i + 1) / 2 + j] = exL[iL] + exR[iR];
          }
          cJ[i] = exL[36 + aL] + exR[36 + aR];
        }
        tpos += (int)exR[44]; tzer += (int)exR[45];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double d = SJ[sidx(i, i)];
          tpos += d > tol;
          tzer += fabs(d) <= tol;
          const double r = rrcp(d);
          zj[i] = rdiv(cJ[i], d, r);
          double w[4];
#pragma unroll
          for (int a = i + 1; a < 4; a++) { w[a] = SJ[sidx(a, i)]; lj[sidx(a - 1, i)] = rdiv(w[a], d, r); }
#pragma unroll
          for (int a = i + 1; a < 4; a++) {
#pragma unroll
            for (int b = i + 1; b <= a; b++) SJ[sidx(a, b)] = fma(w[a], -lj[sidx(b - 1, i)], SJ[sidx(a, b)]);
            cJ[a] = fma(w[a], -zj[i], cJ[a]);
          }
        }
      }
    }
    bool alldone = true;
    if (part == 0) {
      const bool ok = (BAND_DBG != 0) || (tpos == P.nvar && tzer == 0);   // (timing probes compute garbage: no ladder behind them)
      if (mode == MODE_FACTOR) {
        if (valid) {
          as_global(Ain.success)[cprob] = ok ? 1 : 0;
          if (Ain.npos) as_global(Ain.npos)[cprob] = tpos;
          if (Ain.nzero) as_global(Ain.nzero)[cprob] = tzer;
        }
        done = true;
      } else if (mode == MODE_SOLVE) {
        success = ok;   // (a problem whose factorisation fails the inertia rule has no factor: its rows of d stay untouched)
        done = true;
      } else if (!done) {
        nfact++;
        if (ok) { done = true; success = true; }
        else if (nfact == 1) {
          rho = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old);
          ovr = true; wrote = rho;
        } else if (rho <= rhomax) {
          rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
          if (rho <= rhomax) wrote = rho; else done = true;
        } else done = true;
      }
      alldone = __all(done || !clane);
      if (clane) { ctrl[lane] = rho; ctrl[NL + lane] = (ovr ? 1.0 : 0.0) + (success ? 2.0 : 0.0); }
      if (lane == 0) ctrl[2 * NL] = alldone ? 1.0 : 0.0;
    }
    if (P.nparts == 2) {
      __syncthreads();
      if (part == 1) {
        if (clane) { rho = ctrl[lane]; const int f = (int)ctrl[NL + lane]; ovr = f & 1; success = f & 2; }
        alldone = ctrl[2 * NL] != 0.0;
      }
      __syncthreads();   // the control block is rewritten by the next rung
    }
    if (alldone) break;
  }
  if (mode == MODE_FACTOR || (BAND_DBG & 1)) {
    if (mode == MODE_NEWTON && part == 0 && valid) as_global(Ain.success)[cprob] = 1;
    return;
  }
  // ================= backward: d = -K^-1 rhs where the factorisation succeeded =================
  {
    double xs[NS + 1];
#pragma unroll
    for (int q = 0; q < NS + 1; q++) xs[q] = 0.0;
    if (P.nparts == 2) {
      if (part == 0 && clane) {
        double xj[4];
        xj[3] = zj[3];
        xj[2] = fma(-lj[sidx(2, 2)], xj[3], zj[2]);
        xj[1] = fma(-lj[sidx(2, 1)], xj[3], fma(-lj[sidx(1, 1)], xj[2], zj[1]));
        xj[0] = fma(-lj[sidx(2, 0)], xj[3], fma(-lj[sidx(1, 0)], xj[2], fma(-lj[sidx(0, 0)], xj[1], zj[0])));
        double* ex = reinterpret_cast<double*>(myb + EXCH_OFF * 8);
        double* exR = reinterpret_cast<double*>(myb + (size_t)NL * LANE_D * 8 + EXCH_OFF * 8);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ex[i] = xj[i]; exR[i] = xj[i];
          if (valid && success) gd[(long long)cprob * N + P.m0 + i] = -xj[i];
        }
      }
      __syncthreads();
      if (clane) {
        const double* ex = reinterpret_cast<const double*>(myb + EXCH_OFF * 8);
        const int t0 = part == 0 ? P.m0 % NS : (P.n - 1 - P.m0) % NS;
#pragma unroll
        for (int s = 0; s < NS; s++) {
          // junction variable i sits in slot (t0 + i) % 8 (first part) / (t0 - i) % 8 (second part)
          const int i = part == 0 ? (s - t0 + NS) % NS : (t0 - s + NS) % NS;
          xs[s] = i < 4 ? ex[i] : 0.0;
        }
      }
    }
    // which problems store: flags of the workgroup in the control block
    if (part == 0 && clane) ctrl[NL + lane] = (valid && success) ? 2.0 : 0.0;
    __syncthreads();
    bool movst[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) movst[i] = movok[i] && ctrl[NL + i * 8 + lq] != 0.0;
    const bool okme = valid && ctrl[NL + lane % NL] != 0.0 && clane;
    const long long pd = (long long)(prob0 + cpl) * N;
    BAND_ISSUE_ALL(epochs + (nepochs - 1) * BAND_EW, BE_BP, bops_g, 0)
    for (int e = nepochs - 1; e >= 0; e--) {
      cptr E = epochs + e * BAND_EW;
      BSTAMP(5)
      BAND_COMMIT_ALL()
      BSTAMP(6)
      const bool more_ = e > 0 && !(BAND_DBG & 8);
      if (more_) BAND_ISSUE_DESC(epochs + (e - 1) * BAND_EW, BE_BP)
      const int nst = E[BE_NSTEP];
      int o = 0;
      BSTAMP(7)
      Rec stC, stN;
      RowRec rwC, rwN;
      load_rec(stC, recb, 0);
      load_row(rwC, recb, BAND_SW);
#define BAND_BSTEP(PHV)                                                                                                     \
      if (PHV < nst) {                                                                                                      \
        const int fl = __builtin_amdgcn_readfirstlane(stC.v[BS_FLAGS]);                                                     \
        const int onext = o + BAND_SW + BAND_RW * ((fl >> 8) & 255);                                                        \
        if (PHV > 0) { load_rec(stN, recb, onext); load_row(rwN, recb, onext + BAND_SW); }                                  \
        if (clane && !(BAND_DBG & 4)) {                                                                                     \
          BOps op_;                                                                                                         \
          bload(op_, stC, rwC, fl, myb);                                                                                    \
          bstep<PHV>(xs, op_, stC, rwC, fl, recb, o, myb, borders, gd, pd, okme);                                           \
        }                                                                                                                   \
        o = onext; stC = stN; rwC = rwN;                                                                                    \
      }
      BAND_BSTEP(7)
      if (more_) { BAND_ISSUE(0) BAND_ISSUE(1) BAND_ISSUE(2) BAND_ISSUE(3) }
      BAND_BSTEP(6)
      if (more_) { BAND_ISSUE(4) BAND_ISSUE(5) BAND_ISSUE(6) BAND_ISSUE(7) }
      BAND_BSTEP(5)
      if (more_) { BAND_ISSUE(8) BAND_ISSUE(9) BAND_ISSUE(10) BAND_ISSUE(11) }
      BAND_BSTEP(4)
      if (more_) { BAND_ISSUE(12) BAND_ISSUE(13) BAND_ISSUE(14) BAND_ISSUE_REC(bops_g, epochs[(e - 1) * BAND_EW + BE_BOFF]) }
      BAND_BSTEP(3) BAND_BSTEP(2) BAND_BSTEP(1) BAND_BSTEP(0)
      BSTAMP(8)
      // solution components of the epoch
      const int xlo = E[BE_DXLO], xc = E[BE_DXCNT], rlo = E[BE_DRLO], rc = E[BE_DRCNT];
      char* dxo = reinterpret_cast<char*>(dbase) + ((long long)xlo << 3);
      char* dro = reinterpret_cast<char*>(dbase) + ((long long)rlo << 3);
      {
        double dx_[NI], dr_[BAND_DR_MAX / 8][NI];
#pragma unroll
        for (int i = 0; i < NI; i++) dx_[i] = *reinterpret_cast<const double*>(wblk + ldsb[i] + BAND_DX_OFF * 8);
#pragma unroll
        for (int cpc = 0; cpc < BAND_DR_MAX / 8; cpc++)
#pragma unroll
          for (int i = 0; i < NI; i++) dr_[cpc][i] = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_DR_OFF + 8 * cpc) * 8);
#pragma unroll
        for (int i = 0; i < NI; i++)
          if (movst[i] && le < xc && !(BAND_DBG & 16)) *reinterpret_cast<double*>(dxo + ((movp[i] * (unsigned)N + (unsigned)le) << 3)) = dx_[i];
#pragma unroll
        for (int cpc = 0; cpc < BAND_DR_MAX / 8; cpc++)
#pragma unroll
          for (int i = 0; i < NI; i++)
            if (movst[i] && cpc * 8 + le < rc && !(BAND_DBG & 16)) *reinterpret_cast<double*>(dro + (((movp[i] * (unsigned)N + (unsigned)le) << 3) + 64 * cpc)) = dr_[cpc][i];
      }
    }
  }
#ifdef BAND_STAMPS
  BSTAMP(9)
  if (blockIdx.x == 0 && lane == 0) for (int k = 0; k < 12; k++) gd[(long long)prob0 * N + part * 12 + k] = (double)bst_[k];
#endif
  // ================= outputs of newton_system! =================
  if (part == 0 && mode == MODE_NEWTON) {
    if (nfact > 1 && rho <= rhomax) rho_old = rho;
    if (valid) {
      as_global(Ain.rho)[cprob] = rho;
      as_global(Ain.rho_old)[cprob] = rho_old;
      as_global(Ain.nfact)[cprob] = nfact;
      as_global(Ain.success)[cprob] = success ? 1 : 0;
    }
    // rho slots of the problems that climbed (src/CaNNOLeS.jl:1031,1038,1044-1046): the last nvar entries of vals
    for (int q = 0; q < NL; q++) {
      const int nf = __builtin_amdgcn_readlane(nfact, q);
      const int vq = __builtin_amdgcn_readlane((int)valid, q);
      if (nf > 1 && vq) {
        const double wq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wrote), q), __builtin_amdgcn_readlane(__double2loint(wrote), q));
        if (vil) {
          double* vg = gvals + ilg * band_il_blocks(nnz) * G8 + ilp + q * 8;
          for (int i = lane; i < P.nvar; i += 64) vg[band_il_offset(nnz - P.nvar + i, vstride)] = wq;
        } else {
          double* vt = gvals + (long long)(prob0 + q) * nnz + (nnz - P.nvar);
          for (int i = lane; i < P.nvar; i += 64) vt[i] = wq;
        }
      }
    }
  }
}

#if defined(BAND_MW) && !defined(CNL_EXPERIMENT)
#error "BAND_MW needs -DCNL_EXPERIMENT=1"
#endif
#ifdef BAND_MW
// ======================================================================================================================================
// EXPERIMENT (round 6; not in the product build: -DCNL_EXPERIMENT=1 -DBAND_MW, tuning key band_movers=1..3): the same program with LOADER
// wavefronts.  A workgroup of EIGHT wavefronts serves two groups of NL problems; per (group, part) one wavefront computes AND streams
// its own factor records / solution components out (as in band_newton_kernel), and one wavefront only LOADS into staging registers and
// commits them to LDS around the workgroup barriers of an epoch.  Only a wavefront without stores can wait for a set of loads alone: on
// gfx950 loads and stores share one counter (vmcnt) and return out of order with respect to each other (profiles/r06_band_movers.jsonl:
// the first two look-ahead experiments waited with vmcnt(0) for loads issued a few steps earlier).  Roles: waves 0,1 compute / 2,3 load
// for group 0, waves 4,5 load / 6,7 compute for group 1 — wavefronts w and w + 4 share a SIMD (tools/simd_map.hip), so every SIMD holds
// one of each.  Measured (same file): +6 ... 10 % at 8 192 problems, nothing at 16 384 — the look-ahead question is closed.
#define MW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// v5: every (problem, part) owns TWO standard lane blocks (pieces | half ring | zero cell): the loader commits the operands of epoch
// e + 1 into block (e + 1) % 2 WHILE epoch e computes out of block e % 2 — the compute wavefronts never wait for a commit.  One staging
// set in registers: the loads of epoch e + 2 are issued behind the commit of e + 1 (a single generation in flight: the waits at the
// next commit are counted, nothing younger to wait for).  The step blocks have ONE buffer, written between the two barriers of an epoch
// boundary.  307 doubles per lane (odd): 4 x 16 lanes of a workgroup = 157 KB of the 160 KB of a CU.
// v6 (DB = false): ONE block per lane and the staging registers as the second buffer — the loads of epoch e + 1 are issued behind
// barrier X of epoch e and committed behind its barrier Y; 32 problems per group (4 x 32 lanes = the LDS of a CU, as band_newton_kernel<32>
// at 16 384 problems).
constexpr int MW_BLK = BAND_LANE_DOUBLES * 8;   // byte offset of the second block
constexpr int mw_lane_d(bool db) { return db ? 2 * BAND_LANE_DOUBLES + 1 : BAND_LANE_DOUBLES; }

template <int NL, bool DB>
__global__ void __launch_bounds__(512, 2) band_newton_mw_kernel(const BandDev P, const LaunchArgs Ain) {
  constexpr int NI = NL / 8;
  constexpr int LANE_D = mw_lane_d(DB);
  constexpr bool LINT = true;   // factor records interleaved over the group's problems (the address arithmetic is the mover's, off the chain)
  extern __shared__ double lds[];
  const int mode = Ain.mode, batch = Ain.batch;
  double* const gvals = as_global(Ain.vals);
  const double* const grhs = as_global(Ain.rhs);
  double* const gd = as_global(Ain.d);
  double* const gL = as_global(Ain.L);
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wv >> 2, part = wv & 1;
  const bool computes = ((wv >> 1) & 1) == grp;
  const int lq = lane >> 3, le = lane & 7;
  const int prob0 = (blockIdx.x * 2 + grp) * NL;
  const bool has_rhs = mode != MODE_FACTOR && grhs != nullptr;
  char* wblk = reinterpret_cast<char*>(lds + (size_t)(grp * 2 + part) * NL * LANE_D);
  char* recb = reinterpret_cast<char*>(lds + (size_t)4 * NL * LANE_D) + (size_t)(grp * 2 + part) * BAND_REC_MAX * 4;
  double* ctrl = reinterpret_cast<double*>(reinterpret_cast<char*>(lds + (size_t)4 * NL * LANE_D) + (size_t)4 * BAND_REC_MAX * 4) + (size_t)grp * (2 * NL + 8);
  cptr epochs = as_const(P.epochs[part]);
  const int nepochs = P.nepochs[part];
  const int nepmax = P.nepochs[0] > P.nepochs[1] ? P.nepochs[0] : P.nepochs[1];
  const int nnz = P.nnz, N = P.N;
  const bool live = prob0 < batch;   // (the last workgroup's second group may be empty: it only joins the barriers)
  constexpr int G8 = BAND_IL_GROUP * 8;
  const bool vil = (Ain.layout & 1) != 0, ril = (Ain.layout & 2) != 0;   // interleaved vals / rhs (band_newton_kernel)
  const int vstride = vil ? G8 : 0, rstride = ril ? G8 : 0;
  const long long ilg = prob0 / BAND_IL_GROUP;
  const int ilp = (prob0 % BAND_IL_GROUP) * 8;
  const double tol = Ain.params[0];

  if (!computes) {
    // ================================================ loader ================================================
    const int* fops_g = as_global(P.fops[part]);
    const int* bops_g = as_global(P.bops[part]);
    const long long lsize = P.lsize;
    const double* vbase = gvals + (vil ? ilg * band_il_blocks(nnz) * G8 + ilp : (long long)prob0 * nnz);
    const double* rbase = !has_rhs ? gvals : grhs + (ril ? ilg * band_il_blocks(N) * G8 + ilp : (long long)prob0 * N);
    double* lbase_g = gL + (long long)prob0 * lsize;
    const int loff8 = (int)P.loff[part];
    unsigned movp[NI], ldsb0[NI];
    bool movok[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
      int pl = i * 8 + lq;
      movok[i] = prob0 + pl < batch;
      if (!movok[i]) pl = live ? batch - 1 - prob0 : 0;
      movp[i] = (unsigned)pl;
      ldsb0[i] = ((unsigned)(i * 8 + lq) * (unsigned)LANE_D + (unsigned)le) << 3;
    }
    double stg[NPC][NI];   // ONE staging set
    int4 rstg0, rkeep;     // step blocks: in flight / waiting for their epoch boundary
    int pcs[NPC];
    constexpr bool BAND_ISSUE_ALWAYS = true;   // (a static number of loads per generation: counted waits)
    // pieces of the staged epoch into block PAR of every lane; the step blocks are kept for the epoch boundary
#define MW_COMMIT_PIECES(PAR)                                                                                                 \
    {                                                                                                                         \
      unsigned ldsb[NI];                                                                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) ldsb[i_] = ldsb0[i_] + (unsigned)(DB ? (PAR) * MW_BLK : 0);                     \
      BAND_COMMIT(0) BAND_COMMIT(1) BAND_COMMIT(2) BAND_COMMIT(3) BAND_COMMIT(4) BAND_COMMIT(5) BAND_COMMIT(6) BAND_COMMIT(7) \
      BAND_COMMIT(8) BAND_COMMIT(9) BAND_COMMIT(10) BAND_COMMIT(11) BAND_COMMIT(12) BAND_COMMIT(13) BAND_COMMIT(14)           \
      rkeep = rstg0;                                                                                                          \
    }
#define MW_COMMIT_REC() { reinterpret_cast<int4*>(recb)[lane] = rkeep; }
#define MW_ISSUE_ALL(EP, OFS, OPS, OPOFF)                                                                                     \
    {                                                                                                                         \
      BAND_ISSUE_DESC(EP, OFS)                                                                                                \
      BAND_ISSUE_REC(OPS, OPOFF)                                                                                              \
      BAND_ISSUE(0) BAND_ISSUE(1) BAND_ISSUE(2) BAND_ISSUE(3) BAND_ISSUE(4) BAND_ISSUE(5) BAND_ISSUE(6) BAND_ISSUE(7)         \
      BAND_ISSUE(8) BAND_ISSUE(9) BAND_ISSUE(10) BAND_ISSUE(11) BAND_ISSUE(12) BAND_ISSUE(13) BAND_ISSUE(14)                  \
    }
    while (true) {
      // ---- forward: epoch e computes out of block e % 2 ----
      if (live && nepochs > 0) {
        MW_ISSUE_ALL(epochs, BE_FP, fops_g, 0)
        MW_COMMIT_PIECES(0)
        MW_COMMIT_REC()
        if (DB && nepochs > 1) MW_ISSUE_ALL(epochs + BAND_EW, BE_FP, fops_g, epochs[BAND_EW + BE_FOFF])
      }
      for (int e = 0; e < nepmax; e++) {
        MW_BARRIER();   // X: pieces and step blocks of epoch e are in LDS
        if (live) {
          if constexpr (DB) {
            if (e + 1 < nepochs) MW_COMMIT_PIECES((e + 1) & 1)   // (its block is free: the compute wavefront is past epoch e - 1)
            if (e + 2 < nepochs) MW_ISSUE_ALL(epochs + (e + 2) * BAND_EW, BE_FP, fops_g, epochs[(e + 2) * BAND_EW + BE_FOFF])
          } else {
            if (e + 1 < nepochs) MW_ISSUE_ALL(epochs + (e + 1) * BAND_EW, BE_FP, fops_g, epochs[(e + 1) * BAND_EW + BE_FOFF])
          }
        }
        MW_BARRIER();   // Y: the compute wavefront is through epoch e
        if (live && e + 1 < nepochs) {
          if constexpr (!DB) MW_COMMIT_PIECES(0)
          MW_COMMIT_REC()
        }
      }
      MW_BARRIER();   // J1
      MW_BARRIER();   // J2: the decision is in the control block
      const bool alldone = ctrl[2 * NL] != 0.0;
      MW_BARRIER();   // J3
      if (alldone) break;
    }
    if (mode == MODE_FACTOR) return;
    // ---- backward: trip t handles epoch e = nepmax - 1 - t out of block t % 2 ----
    MW_BARRIER();   // K1
    MW_BARRIER();   // K2 (behind it the junction exchange in block 0 is read)
    {
      // the first trip with an epoch of this part, and the one behind it
      const int t0 = nepmax - nepochs;   // trips 0 .. t0 - 1 have no epoch here
      if (live && nepochs > 0) {
        MW_ISSUE_ALL(epochs + (nepochs - 1) * BAND_EW, BE_BP, bops_g, epochs[(nepochs - 1) * BAND_EW + BE_BOFF])
        MW_COMMIT_PIECES(t0 & 1)
        MW_COMMIT_REC()
        if (DB && nepochs > 1) MW_ISSUE_ALL(epochs + (nepochs - 2) * BAND_EW, BE_BP, bops_g, epochs[(nepochs - 2) * BAND_EW + BE_BOFF])
      }
      for (int t = 0; t < nepmax; t++) {
        const int e = nepmax - 1 - t;   // epoch of this trip (>= nepochs: none)
        MW_BARRIER();   // X
        if (live && e < nepochs) {
          if constexpr (DB) {
            if (e - 1 >= 0) MW_COMMIT_PIECES((t + 1) & 1)
            if (e - 2 >= 0) MW_ISSUE_ALL(epochs + (e - 2) * BAND_EW, BE_BP, bops_g, epochs[(e - 2) * BAND_EW + BE_BOFF])
          } else {
            if (e - 1 >= 0) MW_ISSUE_ALL(epochs + (e - 1) * BAND_EW, BE_BP, bops_g, epochs[(e - 1) * BAND_EW + BE_BOFF])
          }
        }
        MW_BARRIER();   // Y
        if (live && e < nepochs && e - 1 >= 0) {
          if constexpr (!DB) MW_COMMIT_PIECES(0)
          MW_COMMIT_REC()
        }
      }
    }
    return;
  }

  // ================================================ compute ================================================
  cptr borders = as_const(P.borders[part]);
  const bool clane = lane < NL;
  const int cprob = prob0 + (clane ? lane : 0);
  const bool valid = clane && cprob < batch;
  const int cpl = valid ? cprob - prob0 : (live ? batch - 1 - prob0 : 0);
  char* myb = wblk + (size_t)(clane ? lane : 0) * LANE_D * 8;
  const long long pv = !live ? 0 : vil ? ilg * band_il_blocks(nnz) * G8 + ilp + cpl * 8 : (long long)(prob0 + cpl) * nnz;
  const long long pr = !live ? 0 : ril ? ilg * band_il_blocks(N) * G8 + ilp + cpl * 8 : (long long)(prob0 + cpl) * N;
  // the wavefront streams its own factor records / solution components out (all 64 lanes: lane (lq, le) = element le of problems lq, lq + 8)
  const long long lsize = P.lsize;
  double* lbase_g = gL + (long long)prob0 * lsize;
  const int loff8 = (int)P.loff[part];
  double* dbase = gd ? gd + (long long)prob0 * N : nullptr;
  unsigned movp[NI], ldsb[NI];
  bool movok[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    int pl = i * 8 + lq;
    movok[i] = prob0 + pl < batch;
    if (!movok[i]) pl = live ? batch - 1 - prob0 : 0;
    movp[i] = (unsigned)pl;
    ldsb[i] = ((unsigned)(i * 8 + lq) * (unsigned)LANE_D + (unsigned)le) << 3;
  }
  for (int t = lane; t < NL; t += 64) {   // the zero cells
    *reinterpret_cast<double*>(wblk + ((size_t)t * LANE_D + BAND_ZERO_OFF) * 8) = 0.0;
    if constexpr (DB) *reinterpret_cast<double*>(wblk + ((size_t)t * LANE_D + BAND_ZERO_OFF) * 8 + MW_BLK) = 0.0;
  }
  char* const wblk0 = wblk;
  char* const myb0 = myb;
  const double kdec = Ain.params[2], kinc = Ain.params[3], klarge = Ain.params[4], rho0 = Ain.params[5], rhomax = Ain.params[6], rhomin = Ain.params[7];
  double rho = 0.0, wrote = 0.0;
  double rho_old = (mode == MODE_NEWTON && valid) ? as_global(Ain.rho_old)[cprob] : 0.0;
  int nfact = 0;
  bool done = !valid, success = false, ovr = false;
  Win W;
  int npos = 0, nzer = 0;
  double lj[6], zj[4];
  for (int q = 0; q < 6; q++) lj[q] = 0.0;
  for (int q = 0; q < 4; q++) zj[q] = 0.0;
  while (true) {
#pragma unroll
    for (int q = 0; q < NS * (NS + 1) / 2; q++) W.S[q] = 0.0;
#pragma unroll
    for (int q = 0; q < NS; q++) { W.X[q] = 0.0; W.c[q] = 0.0; }
    W.S55 = 0.0; W.c5 = 0.0;
    npos = 0; nzer = 0;
    for (int e = 0; e < nepmax; e++) {
      MW_BARRIER();   // X
      if (live && e < nepochs) {
        cptr E = epochs + e * BAND_EW;
        const int nst = E[BE_NSTEP];
        char* const wblk = wblk0 + (DB ? (e & 1) * MW_BLK : 0);   // the block of the epoch: operands, out ring, zero cell
        char* const myb = myb0 + (DB ? (e & 1) * MW_BLK : 0);
        int o = 0;
        Rec stC, stN;
        RowRec rwC, rwN;
        load_rec(stC, recb, 0);
        load_row(rwC, recb, BAND_SW);
#define BAND_FSTEP_MW(PHV)                                                                                                  \
        if (PHV < nst) {                                                                                                    \
          const int fl = __builtin_amdgcn_readfirstlane(stC.v[BS_FLAGS]);                                                   \
          const int onext = o + BAND_SW + BAND_RW * ((fl >> 8) & 255);                                                      \
          if (PHV + 1 < nst) { load_rec(stN, recb, onext); load_row(rwN, recb, onext + BAND_SW); }                          \
          if (clane) {                                                                                                      \
            FOps op_;                                                                                                       \
            fload(op_, stC, rwC, fl, myb);                                                                                  \
            fstep<PHV>(W, op_, stC, fl, recb, o, myb, gvals, grhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer, vstride, rstride); \
          }                                                                                                                 \
          o = onext; stC = stN; rwC = rwN;                                                                                  \
        }
        BAND_FSTEP_MW(0) BAND_FSTEP_MW(1) BAND_FSTEP_MW(2) BAND_FSTEP_MW(3)
        BAND_LFLUSH(E[BE_LBASE], E[BE_LCNT])
        BAND_FSTEP_MW(4) BAND_FSTEP_MW(5) BAND_FSTEP_MW(6) BAND_FSTEP_MW(7)
        if (nst == BAND_EPOCH) {
#pragma unroll
          for (int a = 0; a < NS; a++)
#pragma unroll
            for (int b = 0; b <= a; b++)
              if (b < 4) W.S[sidx(a, b)] = 0.0;
#pragma unroll
          for (int a = 0; a < 4; a++) { W.X[a] = 0.0; W.c[a] = 0.0; }
        }
        BAND_LFLUSH(E[BE_LBASE2], E[BE_LCNT2])
      }
      MW_BARRIER();   // Y
    }
    // ---- junction + inertia rule + rho ladder: as in band_newton_kernel (two parts) ----
    int tpos = npos, tzer = nzer;
    if (clane) {
      double* ex = reinterpret_cast<double*>(myb + EXCH_OFF * 8);
#pragma unroll
      for (int q = 0; q < NS * (NS + 1) / 2; q++) ex[q] = W.S[q];
#pragma unroll
      for (int q = 0; q < NS; q++) ex[36 + q] = W.c[q];
      ex[44] = (double)npos; ex[45] = (double)nzer;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the factor records are in L2 before the loader wavefront reads them back
    MW_BARRIER();   // J1
    if (part == 0 && clane) {
      const double* exL = reinterpret_cast<const double*>(myb + EXCH_OFF * 8);
      const double* exR = reinterpret_cast<const double*>(myb + (size_t)NL * LANE_D * 8 + EXCH_OFF * 8);
      const int tL = P.m0 % NS, tR = (P.n - 1 - P.m0) % NS;
      double SJ[10], cJ[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int aL = (tL + i) % NS, aR = (tR - i + NS) % NS;
#pragma unroll
        for (int j = 0; j <= i; j++) {
          const int bL = (tL + j) % NS, bR = (tR - j + NS) % NS;
          const int iL = aL >= bL ? aL * (aL + 1) / 2 + bL : bL * (bL + 1) / 2 + aL;
          const int iR = aR >= bR ? aR * (aR + 1) / 2 + bR : bR * (bR + 1) / 2 + aR;
          SJ[i * (i + 1) / 2 + j] = exL[iL] + exR[iR];
        }
        cJ[i] = exL[36 + aL] + exR[36 + aR];
      }
      tpos += (int)exR[44]; tzer += (int)exR[45];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const double d = SJ[sidx(i, i)];
        tpos += d > tol;
        tzer += fabs(d) <= tol;
        const double r = rrcp(d);
        zj[i] = rdiv(cJ[i], d, r);
        double w[4];
#pragma unroll
        for (int a = i + 1; a < 4; a++) { w[a] = SJ[sidx(a, i)]; lj[sidx(a - 1, i)] = rdiv(w[a], d, r); }
#pragma unroll
        for (int a = i + 1; a < 4; a++) {
#pragma unroll
          for (int b = i + 1; b <= a; b++) SJ[sidx(a, b)] = fma(w[a], -lj[sidx(b - 1, i)], SJ[sidx(a, b)]);
          cJ[a] = fma(w[a], -zj[i], cJ[a]);
        }
      }
    }
    bool alldone = true;
    if (part == 0) {
      const bool ok = tpos == P.nvar && tzer == 0;
      if (mode == MODE_FACTOR) {
        if (valid) {
          as_global(Ain.success)[cprob] = ok ? 1 : 0;
          if (Ain.npos) as_global(Ain.npos)[cprob] = tpos;
          if (Ain.nzero) as_global(Ain.nzero)[cprob] = tzer;
        }
        done = true;
      } else if (mode == MODE_SOLVE) {
        success = ok;
        done = true;
      } else if (!done) {
        nfact++;
        if (ok) { done = true; success = true; }
        else if (nfact == 1) {
          rho = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old);
          ovr = true; wrote = rho;
        } else if (rho <= rhomax) {
          rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
          if (rho <= rhomax) wrote = rho; else done = true;
        } else done = true;
      }
      alldone = __all(done || !clane);
      if (clane) { ctrl[lane] = rho; ctrl[NL + lane] = (ovr ? 1.0 : 0.0) + (success ? 2.0 : 0.0); }
      if (lane == 0) ctrl[2 * NL] = alldone ? 1.0 : 0.0;
    }
    MW_BARRIER();   // J2
    if (part == 1) {
      if (clane) { rho = ctrl[lane]; const int f = (int)ctrl[NL + lane]; ovr = f & 1; success = f & 2; }
      alldone = ctrl[2 * NL] != 0.0;
    }
    MW_BARRIER();   // J3
    if (alldone) break;
  }
  if (mode == MODE_FACTOR) return;
  // ---- backward ----
  {
    double xs[NS + 1];
#pragma unroll
    for (int q = 0; q < NS + 1; q++) xs[q] = 0.0;
    if (part == 0 && clane) {
      double xj[4];
      xj[3] = zj[3];
      xj[2] = fma(-lj[sidx(2, 2)], xj[3], zj[2]);
      xj[1] = fma(-lj[sidx(2, 1)], xj[3], fma(-lj[sidx(1, 1)], xj[2], zj[1]));
      xj[0] = fma(-lj[sidx(2, 0)], xj[3], fma(-lj[sidx(1, 0)], xj[2], fma(-lj[sidx(0, 0)], xj[1], zj[0])));
      double* ex = reinterpret_cast<double*>(myb + EXCH_OFF * 8);
      double* exR = reinterpret_cast<double*>(myb + (size_t)NL * LANE_D * 8 + EXCH_OFF * 8);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        ex[i] = xj[i]; exR[i] = xj[i];
        if (valid && success) gd[(long long)cprob * N + P.m0 + i] = -xj[i];
      }
    }
    MW_BARRIER();   // K1
    if (clane) {
      const double* ex = reinterpret_cast<const double*>(myb + EXCH_OFF * 8);
      const int t0 = part == 0 ? P.m0 % NS : (P.n - 1 - P.m0) % NS;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const int i = part == 0 ? (s - t0 + NS) % NS : (t0 - s + NS) % NS;
        xs[s] = i < 4 ? ex[i] : 0.0;
      }
    }
    if (part == 0 && clane) ctrl[NL + lane] = (valid && success) ? 2.0 : 0.0;
    MW_BARRIER();   // K2
    const bool okme = valid && ctrl[NL + lane % NL] != 0.0 && clane;
    bool movst[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) movst[i] = movok[i] && ctrl[NL + i * 8 + lq] != 0.0;
    const long long pd = live ? (long long)(prob0 + cpl) * N : 0;
    for (int e = nepmax - 1; e >= 0; e--) {
      MW_BARRIER();   // X
      if (live && e < nepochs) {
        cptr E = epochs + e * BAND_EW;
        const int nst = E[BE_NSTEP];
        char* const wblk = wblk0 + (DB ? ((nepmax - 1 - e) & 1) * MW_BLK : 0);   // the block of the trip
        char* const myb = myb0 + (DB ? ((nepmax - 1 - e) & 1) * MW_BLK : 0);
        int o = 0;
        Rec stC, stN;
        RowRec rwC, rwN;
        load_rec(stC, recb, 0);
        load_row(rwC, recb, BAND_SW);
#define BAND_BSTEP_MW(PHV)                                                                                                  \
        if (PHV < nst) {                                                                                                    \
          const int fl = __builtin_amdgcn_readfirstlane(stC.v[BS_FLAGS]);                                                   \
          const int onext = o + BAND_SW + BAND_RW * ((fl >> 8) & 255);                                                      \
          if (PHV > 0) { load_rec(stN, recb, onext); load_row(rwN, recb, onext + BAND_SW); }                                \
          if (clane) {                                                                                                      \
            BOps op_;                                                                                                       \
            bload(op_, stC, rwC, fl, myb);                                                                                  \
            bstep<PHV>(xs, op_, stC, rwC, fl, recb, o, myb, borders, gd, pd, okme);                                         \
          }                                                                                                                 \
          o = onext; stC = stN; rwC = rwN;                                                                                  \
        }
        BAND_BSTEP_MW(7) BAND_BSTEP_MW(6) BAND_BSTEP_MW(5) BAND_BSTEP_MW(4) BAND_BSTEP_MW(3) BAND_BSTEP_MW(2) BAND_BSTEP_MW(1) BAND_BSTEP_MW(0)
        {   // solution components of the epoch
          const int xlo = E[BE_DXLO], xc = E[BE_DXCNT], rlo = E[BE_DRLO], rc = E[BE_DRCNT];
          char* dxo = reinterpret_cast<char*>(dbase) + ((long long)xlo << 3);
          char* dro = reinterpret_cast<char*>(dbase) + ((long long)rlo << 3);
          double dx_[NI], dr_[BAND_DR_MAX / 8][NI];
#pragma unroll
          for (int i = 0; i < NI; i++) dx_[i] = *reinterpret_cast<const double*>(wblk + ldsb[i] + BAND_DX_OFF * 8);
#pragma unroll
          for (int cpc = 0; cpc < BAND_DR_MAX / 8; cpc++)
#pragma unroll
            for (int i = 0; i < NI; i++) dr_[cpc][i] = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_DR_OFF + 8 * cpc) * 8);
#pragma unroll
          for (int i = 0; i < NI; i++)
            if (movst[i] && le < xc) *reinterpret_cast<double*>(dxo + ((movp[i] * (unsigned)N + (unsigned)le) << 3)) = dx_[i];
#pragma unroll
          for (int cpc = 0; cpc < BAND_DR_MAX / 8; cpc++)
#pragma unroll
            for (int i = 0; i < NI; i++)
              if (movst[i] && cpc * 8 + le < rc) *reinterpret_cast<double*>(dro + (((movp[i] * (unsigned)N + (unsigned)le) << 3) + 64 * cpc)) = dr_[cpc][i];
        }
      }
      MW_BARRIER();   // Y
    }
  }
  // ---- outputs of newton_system! ----
  if (part == 0 && mode == MODE_NEWTON) {
    if (nfact > 1 && rho <= rhomax) rho_old = rho;
    if (valid) {
      as_global(Ain.rho)[cprob] = rho;
      as_global(Ain.rho_old)[cprob] = rho_old;
      as_global(Ain.nfact)[cprob] = nfact;
      as_global(Ain.success)[cprob] = success ? 1 : 0;
    }
    for (int q = 0; q < NL; q++) {
      const int nf = __builtin_amdgcn_readlane(nfact, q);
      const int vq = __builtin_amdgcn_readlane((int)valid, q);
      if (nf > 1 && vq) {
        const double wq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wrote), q), __builtin_amdgcn_readlane(__double2loint(wrote), q));
        if (vil) {
          double* vg = gvals + ilg * band_il_blocks(nnz) * G8 + ilp + q * 8;
          for (int i = lane; i < P.nvar; i += 64) vg[band_il_offset(nnz - P.nvar + i, vstride)] = wq;
        } else {
          double* vt = gvals + (long long)(prob0 + q) * nnz + (nnz - P.nvar);
          for (int i = lane; i < P.nvar; i += 64) vt[i] = wq;
        }
      }
    }
  }
}

// variants (tuning key band_movers): 1 = 16 problems per group, two blocks per lane (v5); 2 = 32 per group, one block (v6); 3 = 16, one block
static int mw_nl(int variant) { return variant == 2 ? 32 : 16; }
int band_mw_group(int variant) { return variant >= 1 && variant <= 3 ? mw_nl(variant) : 0; }
size_t band_mw_lds_bytes(int variant) {
  const int nl = mw_nl(variant);
  return ((size_t)4 * nl * mw_lane_d(variant == 1) + 2 * (2 * nl + 8)) * sizeof(double) + (size_t)4 * BAND_REC_MAX * 4;
}

hipError_t launch_band_mw(const BandDev& P, int variant, const LaunchArgs& a, hipStream_t stream) {
  if (P.nparts != 2 || variant < 1 || variant > 3) return hipErrorInvalidConfiguration;
  const size_t ldsb = band_mw_lds_bytes(variant);
  const int nl = mw_nl(variant);
  const int grid = (a.batch + 2 * nl - 1) / (2 * nl);
  auto go = [&](auto kern) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)ldsb));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsb, stream, P, a);
    return hipGetLastError();
  };
  if (variant == 1) return go(band_newton_mw_kernel<16, true>);
  if (variant == 2) return go(band_newton_mw_kernel<32, false>);
  return go(band_newton_mw_kernel<16, false>);
}

#else
int band_mw_group(int) { return 0; }
size_t band_mw_lds_bytes(int) { return (size_t)-1; }   // (not compiled in: see the experiment above)
hipError_t launch_band_mw(const BandDev&, int, const LaunchArgs&, hipStream_t) { return hipErrorNotSupported; }
#endif
size_t band_lds_bytes(int nparts, int nl) { return ((size_t)nparts * nl * LANE_D + 2 * nl + 8) * sizeof(double) + (size_t)nparts * BAND_REC_MAX * 4; }

hipError_t launch_band(const BandDev& P, int nl, const LaunchArgs& a, hipStream_t stream) {
  const size_t ldsb = band_lds_bytes(P.nparts, nl);
  const int grid = (a.batch + nl - 1) / nl;
  auto go = [&](auto kern) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)ldsb));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * P.nparts), ldsb, stream, P, a);
    return hipGetLastError();
  };
  if (nl == 32) return go(band_newton_kernel<32>);
  if (nl == 16) return go(band_newton_kernel<16>);
  if (nl == 8) return go(band_newton_kernel<8>);
  return hipErrorInvalidConfiguration;
}

}  // namespace cnl
