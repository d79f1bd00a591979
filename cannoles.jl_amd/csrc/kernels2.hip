// kernels2.hip — "register front" kernel for gfx950: the batched hot path.
//
// Same contract as kernels.hip (one launch = newton_system! of
// /root/reference/src/CaNNOLeS.jl:1008-1052 for every problem of the batch), but
// organised for small fronts (order <= 64), which is what a fill-reducing
// ordering of the sparse KKT systems produces:
//
//  * one wavefront owns FOUR problems.  Data movement (assembly gather,
//    extend-add, update-matrix store) uses 16 lanes per problem; index data is
//    shared by the four problems, so it is read once per wave.
//  * a front is eliminated in REGISTERS: lane b holds column b of the packed
//    lower triangle (register k = row top-k, so the pivot row is always r0 and the
//    update shifts the triangle up as a side effect); per pivot the undivided pivot
//    row is published in LDS once and read back by all lanes in one batch, every lane
//    forms its own multiplier, and the rank-1 update is one FMA per row: one LDS round
//    trip and one division per pivot, no cross-lane shuffle, no index decode.
//    Fronts of order <= 16 run four problems at once, <= 32 two, <= 64 one.
//  * the plan is a self-describing record stream (analysis.cpp): the record of front
//    s+2 is prefetched into registers and copied into the wave's LDS buffer when front
//    s+1's lists are no longer needed; the values of front s+1 are gathered while front
//    s is being eliminated, so no global-memory latency sits on the per-front critical
//    path.  With "direct" records the lists address the caller's arrays and carry the
//    products of the condensed residual block: the kernel condenses on the fly.
//  * update matrices wait for their parent on a per-problem LDS stack whose
//    offsets were fixed on the host; the few large ones (and the staging of
//    fronts of order > 32) use a per-problem global scratch instead.
//  * L rows are stored to HBM straight from registers at their pivot step.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

// timing experiments only (tests/support/ablate.py): -DCNL_ABL=<bits> removes pieces of the hot path; results are wrong
#ifndef CNL_ABL
#define CNL_ABL 0
#endif
// Every switch that makes this file compute something else than the product (timing probes with wrong or unguaranteed results,
// diagnostic stamps, the shortened division) compiles only in an EXPERIMENT build: -DCNL_EXPERIMENT=1, which also turns
// cnl_version() negative (capi.cpp) so that such a library cannot pass for the product.
#if (CNL_ABL != 0) || defined(CNL_DBG_NOCONF) || defined(CNL_DBG_VSTRIDE0) || defined(CNL_DBG_LSTRIDE0) || defined(CNL_DF_NOFENCE) || \
    defined(CNL_STAMPS) || (defined(CNL_QUICK_DIV) && CNL_QUICK_DIV) || defined(CNL_DBG_COALJ)
#ifndef CNL_EXPERIMENT
#error "timing probes / diagnostic builds need -DCNL_EXPERIMENT=1 (cnl_version() then reports an experimental library)"
#endif
#endif

namespace cnl {

namespace {

constexpr int RN = 3;    // record prefetch: RN x dwordx4 per lane = RN*256 words
#ifndef CNL_PVR
#define CNL_PVR 6
#endif
#ifndef CNL_PVN
#define CNL_PVN 8
#endif
constexpr int PVR = CNL_PVR;   // raw-value prefetch of the on-the-fly condensation: PVR*16 matrix values per problem (+ 2 x 16 rhs values)
constexpr int PVN = CNL_PVN;   // value prefetch: PVN doubles per lane = PVN*16 entries per problem
constexpr int KB = 10;   // panel rows prefetched per front in the solve sweeps (chain-like orders: up to 10 pivots per front)

__device__ __forceinline__ int tri2(int i) { return (i * (i + 1)) >> 1; }
// row a of the packed position a(a+1)/2 + b of a fast front's image (pos < FAST_IMG_DOUBLES)
__device__ __forceinline__ int tri_row(int pos) {
  int a = (int)((__builtin_sqrtf(8.0f * (float)pos + 1.0f) - 1.0f) * 0.5f);
  a += tri2(a + 1) <= pos;
  a -= tri2(a) > pos;
  return a;
}

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// also drains global stores/loads of the wave (global scratch hand-offs between lanes)
__device__ __forceinline__ void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// word k of a record header that lane (k & 15) read into `hv` with ONE LDS read per wave
#define HDRW(hv, k) __builtin_amdgcn_readlane(hv, k)

// Pointers that arrive inside by-value structs are generic ("flat") to the compiler.  Flat accesses
// count on lgkmcnt as well as vmcnt, so every LDS wait would also wait for outstanding global
// prefetches.  Round-tripping through address space 1 tells the compiler they are global.
template <class T>
__device__ __forceinline__ T* as_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}

// broadcast lane (group base + a) of a TE-lane group
template <int TE>
__device__ __forceinline__ double bcast(double v, int a, int grp4) {
  int lo = __builtin_amdgcn_ds_bpermute(grp4 + a * 4, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(grp4 + a * 4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// w / d through a refined reciprocal: shorter dependent chain than the IEEE expansion (no scaling /
// fix-up steps; a zero or non-finite pivot still yields inf/NaN, which fails the inertia test anyway)
// v_rcp_f64 is good to ~2^-25 (measured on gfx950); one Newton step brings the reciprocal to ~10 ulp, and the
// residual correction of the quotient squares that error away: q' = q + (w - d q) r = (w/d)(1 - eps^2).
#ifndef CNL_QUICK_DIV
#define CNL_QUICK_DIV 0
#endif
__device__ __forceinline__ double fast_div(double w, double d) {
  double r = __builtin_amdgcn_rcp(d);
#if !CNL_QUICK_DIV
  const double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
#endif
  const double q = w * r;
  const double res = fma(-d, q, w);
  return fma(res, r, q);
}

#ifndef CNL_DPP_BC
#define CNL_DPP_BC true
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // bound_ctrl: every lane of a row rotation reads a valid lane, so the destination needs no initial value (the compiler
  // emitted a v_mov 0 in front of every DPP move otherwise: 8 of the 24 instructions of a 16-lane sum)
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, CNL_DPP_BC);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, CNL_DPP_BC);
  return __hiloint2double(hi, lo);
}

// sum over the TE lanes of a group, result in every lane
template <int TE>
__device__ __forceinline__ double gsum(double v) {
  v += dpp_mov<0x128>(v);  // row_ror:8
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x122>(v);  // row_ror:2
  v += dpp_mov<0x121>(v);  // row_ror:1
  if (TE >= 32) v += __shfl_xor(v, 16, 64);
  if (TE >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// optional in-kernel stamps (diagnostic build only: -DCNL_STAMPS); sums of s_memtime deltas per phase
#ifdef CNL_STAMPS
#define STAMP_DECL unsigned long long st_t0 = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_BEGIN { __builtin_amdgcn_sched_barrier(0); st_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#if CNL_STAMPS == 2   // the backward sweep in detail (BSTAMP 0..5); everything else in slot 7
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[7] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); }
#define BSTAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(k) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t0; st_t0 = t_; __builtin_amdgcn_sched_barrier(0); }
#endif
#else
#define STAMP_DECL
#define STAMP_BEGIN
#define STAMP(k)
#endif
#ifndef BSTAMP
#define BSTAMP(k)
#endif

struct Ctx2 {
  const double* vals;   // batch base
  const double* rhs;
  double* L;
  double* gs;           // global scratch base
  double* dout;
  int batch;
};

// ------------------------------------------------------------------------------------------
// Register elimination of one front for the problems of pass `pass` (class TE lanes per problem).
// Lane b holds column b.  Register r<k> holds row (top - k) where `top` is the highest live row,
// so the pivot row is always r0; the rank-1 update writes row (top-k) into r<k-1>, which shifts
// the triangle up by one as a side effect (no moves, no dynamic register index):
//     r<k-1> = fma(-w[top-k], l, r<k>),   w = pivot row read back from LDS, l = this lane's multiplier.
// The rows are spelled as individual scalars through the X-macro lists of elim_lists.inc.
#include "elim_lists.inc"

#define CNL_DECL(k) double r##k;
// lanes b > row read past the row: harmless garbage in the unused upper triangle (staging is padded)
#define CNL_LOAD(k) { const int a_ = top - k > 0 ? top - k : 0; r##k = Fs[tri2(a_) + b]; }  /* rows below 0: unused copies of row 0 */
// The pivot row w (one entry per lane, the pivot d in lane i) is published once per pivot in LDS, undivided
// (lane b -> lb[i - b], so lb[0] = d).  Every lane reads d, forms ITS OWN multiplier lv = w_b / d, and updates
// its column with  F(a,b) -= w_a * lv  where the w_a come back two rows at a time with one broadcast 16-byte
// read.  One LDS round trip per pivot sits on the dependent chain (publish -> read), no cross-lane shuffles.
#define CNL_STEPA(km1, k) r##km1 = fma(-lb[1], lv, r##k);
#define CNL_STEPP(km1, k, kp1)                                                     \
  {                                                                                \
    const double2 l2_ = *reinterpret_cast<const double2*>(lb + k);                 \
    r##km1 = fma(-l2_.x, lv, r##k);                                                \
    r##k = fma(-l2_.y, lv, r##kp1);                                                \
  }
#define CNL_CHK(k) if (k >= i) goto rows_done;  /* rows i-k >= 1 only (row 0 is the unused rhs-row diagonal) */
// Update-matrix rows are stored in ASCENDING row order with all lanes active: the lanes b > a of row a
// land on entries of later rows (or just past the matrix, in free stack space) and are overwritten by
// the stores that follow in program order, so no per-row lane predicate is needed.
#define CNL_USTG(k) { const int a_ = nupd - k; if (a_ >= 0) Ug[tri2(a_) + b] = r##k; }
#define CNL_USTL(k) { const int a_ = nupd - k; if (a_ >= 0) Ul[tri2(a_) + b] = r##k; }

// fast fronts (order <= 16, LDS staging): the image is the packed triangle (plan.h, FAST_IMG_*), row a at a(a+1)/2, so the
// 16 row loads are one base address plus immediate offsets; lanes past the diagonal read entries of later rows (never used)
#define CNL_LOADS(k) r##k = Fss[((15 - k) * (16 - k)) / 2];
#ifdef CNL_STAMPS
#define ESTAMP0 unsigned long long et0_ = 0; if (st_) { __builtin_amdgcn_sched_barrier(0); et0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define ESTAMP(k) if (st_) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_[k] += t_ - et0_; et0_ = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define ESTAMP0
#define ESTAMP(k)
#endif
#define CNL_DEFINE_ELIM(NAME, INL, TEV, GFS, ALL, REV, STEPS, LOADM)                                                   \
  __device__ INL void NAME(int P_prob_doubles, int P_u2_peak, long long P_gs_doubles, long long P_lsize, double* cL_,   \
                           double* cgs_, int cbatch, int lane, int prob0, int pass,                                   \
                           int f, int nupd, long long lptr, int uoff, int fsoff, bool uglob, double* pbase0,          \
                           int* cnt, double eig_tol, unsigned long long* st_ = nullptr) {                              \
    Ctx2 c;                                                                                                            \
    ESTAMP0                                                                                                            \
    c.L = as_global(cL_); c.gs = as_global(cgs_); c.batch = cbatch;                                                    \
    constexpr int TE_ = TEV;                                                                                           \
    constexpr int PPW = 64 / TE_;                                                                                      \
    const int gp = pass * PPW + (TE_ == 64 ? 0 : lane / TE_);                                                          \
    const int b = lane % TE_;                                                                                          \
    const int grp4 = (lane - b) * 4;                                                                                   \
    const int prob = prob0 + gp;                                                                                       \
    const bool valid = prob < c.batch;                                                                                 \
    int pc32_ = valid ? prob : prob0;                                                                                  \
    /* opaque: the per-lane factor base would otherwise be hoisted out of the fronts loop, live across the   */       \
    /* out-of-line calls, spilled, and every reload from scratch waits for ALL outstanding prefetches (vmcnt) */       \
    asm volatile("" : "+v"(pc32_));                                                                                    \
    const long long pclamp = pc32_;                                                                                    \
    double* pb = pbase0 + gp * P_prob_doubles;                                                                         \
    const double* Fs = GFS ? (c.gs + pclamp * P_gs_doubles + fsoff) : (pb + P_u2_peak);                                \
    double* Lp = c.L + pclamp * P_lsize + lptr;                                                                        \
    const int tu = tri2(1 + nupd);                                                                                     \
    const int top = f - 1;                                                                                             \
    const double* Fss = Fs + (top - 15) * 16 + b;                                                                      \
    double* lb = pb + P_u2_peak; /* the LDS staging area is dead once the rows are in registers */                     \
    (void)Fss;                                                                                                         \
    ALL(CNL_DECL)                                                                                                      \
    ALL(LOADM)                                                                                                         \
    ESTAMP(7)                                                                                                          \
    int npos = 0, nzer = 0;                                                                                            \
    (void)grp4;                                                                                                        \
    for (int i = top; i > nupd; i--) {                                                                                 \
      const double w = r0;                                                                                             \
      {                                                                                                                \
        int li_ = i - b;                                                                                               \
        li_ = li_ >= 0 ? li_ : TE_ + 1; /* lanes b > i park their value in an unused slot */                           \
        lb[li_] = w;                                                                                                   \
      }                                                                                                                \
      const double dpiv = lb[0];                                                                                       \
      const double lv = fast_div(w, dpiv);                                                                             \
      npos += dpiv > eig_tol;                                                                                          \
      nzer += fabs(dpiv) <= eig_tol;                                                                                   \
      if (valid && b <= i && !(CNL_ABL & 64)) Lp[tri2(i) - tu + b] = (b == i) ? dpiv : lv;                                                \
      if (1 >= i || (CNL_ABL & 32)) goto rows_done;                                                                    \
      STEPS(CNL_STEPP, CNL_CHK, CNL_STEPA)                                                                             \
    rows_done:;                                                                                                        \
    }                                                                                                                  \
    ESTAMP(5)                                                                                                          \
    if (b == 0) { cnt[gp * 2] += npos; cnt[gp * 2 + 1] += nzer; }                                                      \
    if (uglob) {                                                                                                       \
      if (valid) {                                                                                                     \
        double* Ug = c.gs + pclamp * P_gs_doubles + uoff;                                                              \
        REV(CNL_USTG)                                                                                                  \
      }                                                                                                                \
    } else if (!(CNL_ABL & 128)) {                                                                                     \
      double* Ul = pb + uoff;                                                                                          \
      REV(CNL_USTL)                                                                                                    \
    }                                                                                                                  \
  }

// The rare large classes are real calls (CNL_DEFINE_ELIM, LDS form) so that their register needs do not leak into the hot path.
// The hot class (order <= 16) without an LDS round trip per pivot.  Rows are absolute
// (R<a> = row a of the front, lane b = column b).  A problem is 16 lanes = one DPP row, so "the value of lane a for every
// lane of the problem" is the DPP row broadcast of the fp64 FMA:
//     pivot i:  d = w[lane i],  l_b = w_b / d,   R<a>[b] += w[lane a] * (-l_b)   for a < i
//               as  v_fmac_f64_dpp R<a>, w row_newbcast:a, nl      (w = R<i>)
// One instruction per row update, no publish / read-back through LDS (the LDS pipe is the busiest unit of this kernel)
// and a shorter dependent chain per pivot (broadcast -> division -> update).  Lane numbers are immediates, hence one
// block per pivot position (elim_dpp.inc), entered by wave-uniform branches.
// (Round-2 experiments on this kernel that were measured and NOT kept — a row-per-lane backward sweep through the DPP broadcast,
// a backward sweep fed by global_load_lds_dwordx4 with headers through the scalar cache, a four-operation division chain — are
// recorded in profiles/HISTORY.md section 4; their code lives in the git history, not here.)
#define CNL_DPP_DIV fast_div
#ifndef CNL_PIV_BITMASK
#define CNL_PIV_BITMASK 1
#endif
#if CNL_PIV_BITMASK
#define CNL_PIV_GUARD(i) (pm_ & (1u << (i)))
#else
#define CNL_PIV_GUARD(i) ((i) <= top && (i) > nupd)
#endif
#ifndef CNL_LANE_OPAQUE
#define CNL_LANE_OPAQUE 1
#endif
#if CNL_LANE_OPAQUE
#define CNL_LANE_FENCE asm volatile("" : "+v"(bm_));
#else
#define CNL_LANE_FENCE
#endif
#include "elim_dpp.inc"
// L rows one front late.  Stores and loads return out of order with respect to each other, so a wait for ANY load is a
// vmcnt(0) while stores are in flight: with the ten row stores of a front issued during its elimination, the first use of the
// next front's prefetched values waited for those stores' round trips (ablation, round 3: no L stores = -0.7 ms of 8.0).  The
// rows of front s are therefore kept in registers and stored in step (4) of front s + 1, BEFORE that step's gathers are
// issued: by the time anything waits on those gathers, the stores are a whole front old.
// That pays where two wavefronts share a SIMD (large batches: 7.79 -> 7.38 ms at 8192 problems).  A wavefront that runs
// alone (the staged execution of small batches) loses by it — one system 0.1205 -> 0.133 ms, 256 problems 462 k -> 436 k
// systems/s: its waits are short anyway and the burst of ten stores sits in front of the next gathers — so the kernel is
// compiled both ways (template parameter LATE) and the launcher chooses by the wavefronts in flight.
#define CNL_LPEND_LIST(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
struct LPend {
#define CNL_LP_MEMBER(i) double v##i;
  CNL_LPEND_LIST(CNL_LP_MEMBER)
#undef CNL_LP_MEMBER
  unsigned pm;     // pivot positions whose rows are pending (0: nothing)
  unsigned lofs;   // per-lane byte offset into the factor of the wave's first problem
  long long lptr;  // factor offset (doubles) of the front
};
#define CNL_LROW_OUT(i)                                                                                                              \
  if constexpr (LATE) LP.v##i = (bm_ == i) ? dpiv : lv;                                                                              \
  else if (bm_ <= i && !(CNL_ABL & 64)) *reinterpret_cast<double*>(L_wb + (lofs + ((unsigned)tri2(i) << 3))) = (bm_ == i) ? dpiv : lv;
__device__ __forceinline__ void flush_lrows(LPend& LP, char* L_wb0, int bm_) {
  if (LP.pm == 0) return;
  char* L_wb = L_wb0 + (LP.lptr << 3);
  const unsigned lofs = LP.lofs;
#define CNL_LP_FLUSH(i)                                                                                                   \
  if (LP.pm & (1u << i)) {                                                                                                \
    asm volatile("" : "+v"(bm_));                                                                                         \
    if (bm_ <= i && !(CNL_ABL & 64)) *reinterpret_cast<double*>(L_wb + (lofs + ((unsigned)tri2(i) << 3))) = LP.v##i;      \
  }
  CNL_LPEND_LIST(CNL_LP_FLUSH)
#undef CNL_LP_FLUSH
  LP.pm = 0;
}
#define CNL_DPPF(X, W, NL, A) \
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #A " row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(W), "v"(NL));
// first DPP read of a row the previous pivot's updates wrote: hipcc pads no hazards inside asm (VALU write -> DPP read: 2 wait states)
#define CNL_DPPF_NOP(X, W, NL, A) \
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #A " row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(W), "v"(NL));
#define CNL_DPP_DECL(a) double R##a = Fs[(a * (a + 1)) / 2 + b];
#define CNL_DPP_PRE(i)                                                                                    \
  const double lv = CNL_DPP_DIV(w_, dpiv);                                                                \
  npos += dpiv > eig_tol;                                                                                 \
  nzer += fabs(dpiv) <= eig_tol;                                                                          \
  /* the lane number is made opaque per pivot: the sixteen (b <= i) and sixteen (b == i) lane masks would otherwise be hoisted  */ \
  /* out of the fronts loop into 64 SGPRs, spilled to VGPR lanes and fetched back with two v_readlane each (round 2 ISA)       */ \
  CNL_LANE_FENCE                                                                                          \
  CNL_LROW_OUT(i)                                                                                         \
  const double nl_ = (CNL_ABL & 32) ? 0.0 : -lv;
#define CNL_DPP_POST(i)
#define CNL_DPP_USTG(a) if (a <= nupd) Ug[tri2(a) + b] = R##a;
#define CNL_DPP_USTL(a) if (a <= nupd) Ul[tri2(a) + b] = R##a;
// BNF (round 4): band form of the front.  0: every pivot updates every row below it.  2 / 3: the analysis has shown (structurally,
// analysis.cpp: band word of the record header) that pivot I's row holds non-zeros only in the BNF lowest columns (right-hand
// side, multipliers) and in the CNL_BAND_HW columns right below the pivot — what the fronts of a band problem look like — so the
// other row updates (w_a = 0 exactly) are not compiled in: 6 instead of 10.5 v_fmac_f64_dpp per pivot on cfg3's chain fronts,
// and that instruction (14 cycles) is what the elimination phase is made of.
constexpr int CNL_BAND_HW = 4;
#define CNL_DPPU(I, A, RA) if constexpr (BNF == 0 || (A) < BNF || (A) >= (I) - CNL_BAND_HW) { CNL_DPPF(RA, w_, nl_, A) }
template <bool LATE, int BNF>
__device__ __forceinline__ void eliminate16_dpp(int P_prob_doubles, int P_u2_peak, long long P_gs_doubles, long long P_lsize, double* cL_,
                                                double* cgs_, int cbatch, int lane, int prob0, int f, int nupd, long long lptr, int uoff,
                                                bool uglob, double* pbase0, int* cnt, double eig_tol, LPend& LP) {
  double* Lg = as_global(cL_);
  double* gsg = as_global(cgs_);
  const int gp = lane >> 4;
  const int b = lane & 15;
  const int prob = prob0 + gp;
  const bool valid = prob < cbatch;
  int pc32_ = valid ? prob : prob0;
  asm volatile("" : "+v"(pc32_));  // opaque: the per-lane factor base must not be hoisted out of the fronts loop and kept alive (or spilled)
  const long long pclamp = pc32_;
  double* pb = pbase0 + gp * P_prob_doubles;
  const double* Fs = pb + P_u2_peak + b;
  const int prob0u = __builtin_amdgcn_readfirstlane(prob0);
  char* L_wb = reinterpret_cast<char*>(Lg + (long long)prob0u * P_lsize + lptr);
  const int tu = tri2(1 + nupd);
  const unsigned lofs = ((valid ? (unsigned)gp : 0u) * (unsigned)P_lsize + (unsigned)b - (unsigned)tu) * 8u;
  const int top = f - 1;
  { const double* Fs_ = Fs; (void)Fs_; }
#undef CNL_DPP_DECL
#define CNL_DPP_DECL(a) double R##a = Fs[(a * (a + 1)) / 2];
  CNL_DPP_ROWS(CNL_DPP_DECL)
  int npos = 0, nzer = 0;
  const double one_ = 1.0;
  // bit I set <=> I is a pivot position (nupd < I <= top), wave-uniform
  int bm_ = valid ? b : 64;  // column of this lane; lanes of problems past the batch never store
  const unsigned pm_ = __builtin_amdgcn_readfirstlane(((2u << top) - 1u) & ~((2u << nupd) - 1u));
  CNL_DPP_PIVOTS(CNL_DPP_PRE, CNL_DPP_POST)
  if constexpr (LATE) { LP.pm = pm_; LP.lofs = lofs; LP.lptr = lptr; }
  (void)L_wb;
  if (b == 0) { cnt[gp * 2] += npos; cnt[gp * 2 + 1] += nzer; }
  // update matrix: rows 0 .. nupd in ascending order with all lanes active (see CNL_USTG)
  if (uglob) {
    if (valid) {
      double* Ug = gsg + pclamp * P_gs_doubles + uoff;
      CNL_DPP_ROWS(CNL_DPP_USTG)
    }
  } else if (!(CNL_ABL & 128)) {
    double* Ul = pb + uoff;
    CNL_DPP_ROWS(CNL_DPP_USTL)
  }
  // The image is dead: zero it for the NEXT front here (LDS executes a wavefront's operations in order), so that the next front
  // starts with its extend-add instead of five stores and a barrier.  Behind the update-matrix store: its rows are written
  // with all sixteen lanes and may run up to 15 doubles past the top of the stack, into the image.
  {
    double2* z2 = reinterpret_cast<double2*>(pb + P_u2_peak) + b;
#pragma unroll
    for (int j = 0; j < 4; j++) z2[16 * j] = make_double2(0.0, 0.0);
    if (b < FAST_IMG_DOUBLES / 2 - 64) z2[64] = make_double2(0.0, 0.0);
  }
}
CNL_DEFINE_ELIM(eliminate16g, __attribute__((noinline)), 16, true, CNL_ALL16, CNL_REV16, CNL_STEPS16, CNL_LOAD)
CNL_DEFINE_ELIM(eliminate32, __attribute__((noinline)), 32, false, CNL_ALL32, CNL_REV32, CNL_STEPS32, CNL_LOAD)
CNL_DEFINE_ELIM(eliminate32g, __attribute__((noinline)), 32, true, CNL_ALL32, CNL_REV32, CNL_STEPS32, CNL_LOAD)
CNL_DEFINE_ELIM(eliminate64, __attribute__((noinline)), 64, true, CNL_ALL64, CNL_REV64, CNL_STEPS64, CNL_LOAD)

// backward substitution of one front for the problems of a pass
template <int TE>
__device__ __attribute__((noinline)) void back_front_call(int P_prob_doubles, long long P_lsize, long long P_dstride, double* cL_, double* cdout_,
                                                     int cbatch, int lane, int prob0, int pass, const int* rec, int f, int nupd, int npiv,
                                                     long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag);

template <int TE>
__device__ __forceinline__ void back_front(int P_prob_doubles, long long P_lsize, long long P_dstride, double* cL_, double* cdout_,
                                           int cbatch, int lane, int prob0, int pass, const int* rec, int f, int nupd, int npiv,
                                           long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag) {
  Ctx2 c;
  c.L = as_global(cL_); c.dout = as_global(cdout_); c.batch = cbatch;
  constexpr int PPW = 64 / TE;
  const int gp = pass * PPW + (TE == 64 ? 0 : lane / TE);
  const int b = lane % TE;
  const int prob = prob0 + gp;
  const bool valid = prob < c.batch && okflag[gp] != 0;
  const long long pclamp = prob < c.batch ? prob : prob0;
  double* xs = pbase0 + gp * P_prob_doubles;
  const double* Lp = c.L + pclamp * P_lsize + lptr;
  const int tu = tri2(1 + nupd);
  double xb = 0.0;
  if (pxoff >= 0 && b >= 1 && b <= nupd) xb = xs[pxoff + rec[B_HDR + b]];
  if (pxoff == B_PX_GLOBAL && b >= 1 && b <= nupd) xb = -__hip_atomic_load(c.dout + pclamp * P_dstride + rec[B_HDR + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  wsync();
  // pivots in blocks of KB: the panel rows of a block are loaded together (row i: entries 0..i; lane b takes entry b)
  constexpr int KB = TE - 1 < 8 ? TE - 1 : 8;
  for (int k0 = 0; k0 < npiv; k0 += KB) {
    double lrow[KB];
#pragma unroll
    for (int k = 0; k < KB; k++) {
      double v0 = 0.0;
      if (k0 + k < npiv) {
        const int i = nupd + 1 + k0 + k;
        if (b < i) v0 = Lp[tri2(i) - tu + b];
      }
      lrow[k] = v0;
    }
#pragma unroll
    for (int k = 0; k < KB; k++) {
      if (k0 + k < npiv) {
        const int i = nupd + 1 + k0 + k;
        const double t = (b >= 1 && b < i) ? lrow[k] * xb : 0.0;
        const double s = gsum<TE>(t);
        const double z = bcast<TE>(lrow[k], 0, (lane - b) * 4);
        const double xi = z - s;
        if (b == i) {
          xb = xi;
          if (valid) c.dout[pclamp * P_dstride + rec[B_HDR + 1 + nupd + k0 + k]] = -xi;
        }
      }
    }
  }
  if (b >= 1 && b < f) xs[xoff + b] = xb;
}

template <int TE>
__device__ __attribute__((noinline)) void back_front_call(int P_prob_doubles, long long P_lsize, long long P_dstride, double* cL_, double* cdout_,
                                                     int cbatch, int lane, int prob0, int pass, const int* rec, int f, int nupd, int npiv,
                                                     long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag) {
  back_front<TE>(P_prob_doubles, P_lsize, P_dstride, cL_, cdout_, cbatch, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, okflag);
}

// Out-of-line handling of the rare fronts (order > 16, or staged in the global scratch): staging,
// extend-add and elimination without the register prefetch of the hot path.  Lists are read from
// the record stream in global memory.
struct SlowArgs {  // the few plan scalars the out-of-line path needs (passed by value: no plan struct in scratch)
  int prob_doubles, u2_peak, nnz, rho_begin;
  long long gs_doubles, lsize, vstride, rstride;
  const int* rec;
};
// (round 4: the eight plan scalars come through the wavefront's LDS block — SLOW_ARGS_AT, written once per kernel — so that every
//  argument of the call travels in registers: with 25 parameters seven dwords went over the stack)
constexpr int SLOW_ARGS_AT = 16;   // ints behind `cnt`: eight 64-bit slots
__device__ __attribute__((noinline)) void slow_front(const int* prec_,
                                                     const double* vals_, const double* rhs_, double* L_,
                                                     double* gs_, int batch, int lane, int prob0, const int* rec, int roff,
                                                     double* pbase0, int* cnt, double eig_tol, double rho, bool ovr, bool count_d) {
  SlowArgs P;
  {
    const long long* ps = reinterpret_cast<const long long*>(cnt + SLOW_ARGS_AT);
    P.prob_doubles = (int)rfl((int)ps[0]); P.u2_peak = (int)rfl((int)ps[1]); P.nnz = (int)rfl((int)ps[2]); P.rho_begin = (int)rfl((int)ps[3]);
    P.gs_doubles = ps[4]; P.lsize = ps[5]; P.vstride = ps[6]; P.rstride = ps[7]; P.rec = prec_;
  }
  const double* vals = as_global(vals_);
  const double* rhsb = as_global(rhs_);
  double* Lb = as_global(L_);
  double* gsb = as_global(gs_);
  const int* grec = as_global(P.rec) + roff;
  const int g = lane >> 4, l = lane & 15;
  const int prob = prob0 + g;
  const bool valid = prob < batch;
  const long long pclamp = valid ? prob : prob0;
  const double* myvals = vals + pclamp * P.vstride;
  const double* myrhs = rhs_ ? rhsb + pclamp * P.rstride : nullptr;
  double* mygs = gsb + pclamp * P.gs_doubles;
  double* myU = pbase0 + g * P.prob_doubles;
  double* myFs = myU + P.u2_peak;
  const int npiv = rfl(rec[R_NPIV]), nupd = rfl(rec[R_NUPD]), nasm = rfl(rec[R_NASM]);
  const int nchild = rfl(rec[R_NCHILD]), uoff = rfl(rec[R_UOFF]), flags = rfl(rec[R_FLAGS]), fsoff = rfl(rec[R_FSOFF]);
  const int cls = flags >> 8, aoff = rfl(rec[R_ASM_OFF]), coff = rfl(rec[R_CHILD_OFF]);
  const long long lptr = (long long)rfl(rec[R_LPTR_LO]) | ((long long)rfl(rec[R_LPTR_HI]) << 31);
  const int f = 1 + nupd + npiv;
  const int tf = tri2(f);
  const bool gfs = flags & RF_FS_GLOBAL;
  const bool uglob = flags & RF_U_GLOBAL;
  if (!gfs) {
    for (int t = l; t < tf; t += 16) myFs[t] = 0.0;
    wsync();
    for (int e = l; e < nasm; e += 16) {
      const int src = grec[aoff + e], pos = grec[aoff + nasm + e];
      double v = 0.0;
      if (src >= P.nnz) v = myrhs ? myrhs[src - P.nnz] : 0.0;
      else v = (ovr && src >= P.rho_begin) ? rho : myvals[src];
      __hip_atomic_fetch_add(&myFs[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  } else {
    double* Fg = mygs + fsoff;
    for (int t = l; t < tf; t += 16) Fg[t] = 0.0;
    gsync();
    for (int e0 = 0; e0 < nasm; e0 += 16) {  // one round of 16 entries at a time: duplicates of a slot sit in different rounds
      const int e = e0 + l;
      const int src = grec[aoff + e], pos = grec[aoff + nasm + e];
      double v = 0.0;
      if (src >= P.nnz) v = myrhs ? myrhs[src - P.nnz] : 0.0;
      else v = (ovr && src >= P.rho_begin) ? rho : myvals[src];
      if (valid) Fg[pos] += v;
      gsync();
    }
  }
  // on-the-fly condensation, out-of-line form: two words per product (pos, ia | ib<<10 | id<<20), raw values gathered directly
  const int nprodw = rfl(rec[R_NPROD]), nraw = rfl(rec[R_NRAW]);
  const int nprod = nprodw & 0xffff, nrd_own = count_d ? nprodw >> 16 : 0;
  if (nrd_own > 0) {
    // residual pivots owned by this front: counted in the inertia (src/solver_types.jl:90-95)
    const int raw_off = aoff + 2 * nasm;
    int np_ = 0, nz_ = 0;
    for (int t = l; t < nrd_own; t += 16) {
      const double dv = myvals[grec[raw_off + t]];
      np_ += dv > eig_tol;
      nz_ += fabs(dv) <= eig_tol;
    }
    if (valid && np_) __hip_atomic_fetch_add(&cnt[g * 2], np_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (valid && nz_) __hip_atomic_fetch_add(&cnt[g * 2 + 1], nz_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  if (nprod > 0) {
    const int raw_off = aoff + 2 * nasm, prod_off = raw_off + nraw;
    if (gfs) gsync(); else wsync();
    for (int e = l; e < nprod; e += 16) {
      const int pos = grec[prod_off + 2 * e], w = grec[prod_off + 2 * e + 1];
      double x[3];
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int src = grec[raw_off + ((w >> (10 * q)) & 1023)];
        x[q] = src >= P.nnz ? (myrhs ? myrhs[src - P.nnz] : 0.0) : myvals[src];
      }
      const double v = fast_div(-(x[0] * x[1]), x[2]);
      if (!gfs) __hip_atomic_fetch_add(&myFs[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      else if (valid) __hip_atomic_fetch_add(&mygs[fsoff + pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (gfs) gsync();
  }
  int co = coff;
  for (int ci = 0; ci < nchild; ci++) {
    const int cu = rfl(grec[co + C_UOFF]), tuc = rfl(grec[co + C_TUC]), cfl = rfl(grec[co + C_FLAGS]);
    const int* dest = grec + co + C_HDR;
    const double* Ul = myU + cu;
    const double* Ug = mygs + cu;
    if (!gfs) {
      for (int t = l; t < tuc; t += 16) {
        const double u = cfl ? Ug[t] : Ul[t];
        __hip_atomic_fetch_add(&myFs[dest[t]], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    } else {
      double* Fg = mygs + fsoff;
      for (int t = l; t < tuc; t += 16) {
        const double u = cfl ? Ug[t] : Ul[t];
        if (valid) Fg[dest[t]] += u;
      }
      gsync();
    }
    co += C_HDR + ((tuc + 3) & ~3);
  }
  if (gfs) gsync(); else wsync();
  if (cls == 16) {
    eliminate16g(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, Lb, gsb, batch, lane, prob0, 0, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
  } else if (cls == 32) {
    for (int pass = 0; pass < 2; pass++) {
      if (prob0 + pass * 2 >= batch) break;
      if (gfs) eliminate32g(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, Lb, gsb, batch, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
      else eliminate32(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, Lb, gsb, batch, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
    }
  } else {
    for (int pass = 0; pass < 4; pass++) {
      if (prob0 + pass >= batch) break;
      eliminate64(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, Lb, gsb, batch, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
    }
  }
}

}  // namespace

// Gathers of the first PVN*16 assembly values of a front (lists are padded to multiples of 16, so
// every lane of an issued round has a valid entry).  Loads sit in wave-uniform branches and their
// results are first used one front later: nothing waits between consecutive loads.
// First KB rows of a front's L panel, lane l takes entry l of each row (row k of the panel starts
// k(k+1)/2 + k*(nupd+1) doubles after the panel start).  No clamps: lanes past the end of a row and
// rows past the last pivot read finite data that is never used (the factor storage is zero-padded).
// Same addressing as the forward gathers (wave-uniform base + 32-bit byte offset); rows past the last pivot are
// not loaded.
#ifndef CNL_DBG_LSTRIDE0   // timing probe: every problem reads problem 0's factor in the solve sweeps (cache hits; wrong results)
#define CNL_DBG_LSTRIDE0 0
#endif
#define PREFETCH_ROWS(DST, LPTR, NUPD, NPIV)                                           \
  {                                                                                    \
    const char* rb_ = CNL_DBG_LSTRIDE0 ? reinterpret_cast<const char*>(A.L) + ((long long)(LPTR) << 3) : L_wb + ((long long)(LPTR) << 3); \
    unsigned ro_ = CNL_DBG_LSTRIDE0 ? (unsigned)l * 8u : gofs_l;                       \
    _Pragma("unroll") for (int k = 0; k < KB; k++) {                                   \
      if (k < (NPIV) && !(CNL_ABL & 16384)) DST[k] = *reinterpret_cast<const double*>(rb_ + ro_); else DST[k] = 0.0; \
      ro_ += (unsigned)((NUPD) + 2 + k) << 3;                                          \
    }                                                                                  \
  }

// Value prefetch of the NEXT front.  The lists of a record hold the entries that read the matrix values first and
// the entries that read the right-hand side last, so every round of 16 gathers from one array: the address is a
// wave-uniform base (SGPR pair) plus a 32-bit per-lane byte offset.  All source indices are read from the LDS
// record in one batch; rounds past the end of a list are skipped (the texture addresser is the busiest shared
// unit of this kernel) and their register is set to zero: a plain "keep the old value" made the compiler merge the
// guarded values through copies, with a vmcnt(0) wait right behind the first gather.
#define GATHER_V(SRC) (*reinterpret_cast<const double*>(vals_wb + (((unsigned)(SRC) << 3) + gofs_v)))
#define GATHER_R(SRC) (*reinterpret_cast<const double*>(rhs_wb + (((unsigned)(SRC) << 3) + gofs_r)))
#define PREFETCH_VALUES(RECP, AOFF, NASMV, NASM)                                       \
  {                                                                                    \
    const int nv_ = (NASMV);                                                           \
    const int* sp_ = (RECP) + (AOFF) + l;                                              \
    int src_[PVN + 1];                                                                 \
    _Pragma("unroll") for (int j = 0; j < PVN; j++) src_[j] = sp_[j * 16];             \
    src_[PVN] = sp_[nv_];                                                              \
    _Pragma("unroll") for (int j = 0; j < PVN; j++) {                                  \
      if (j * 16 < nv_) pv[j] = GATHER_V(src_[j]); else pv[j] = 0.0;                   \
    }                                                                                  \
    if (nv_ < (NASM)) prh = GATHER_R(src_[PVN]); else prh = 0.0;                       \
  }
// MODE_SOLVE needs only the right-hand-side round of the plain entries
#define PREFETCH_RHS_ROUND(RECP, AOFF, NASMV, NASM)                                    \
  {                                                                                    \
    const int nv_ = (NASMV);                                                           \
    const int src_ = ((RECP) + (AOFF) + l)[nv_];                                       \
    if (nv_ < (NASM)) prh = GATHER_R(src_); else prh = 0.0;                            \
  }
// Raw values (Jacobian entries, residual pivots, residual right-hand sides) of the products a front's
// condensed slots are made of: same scheme.
#define PREFETCH_RAW(RECP, ROFF, NRAWV, NRAW)                                          \
  {                                                                                    \
    const int nv_ = (NRAWV), nr_ = (NRAW);                                             \
    const int* sp_ = (RECP) + (ROFF) + l;                                              \
    int src_[PVR + 2];                                                                 \
    _Pragma("unroll") for (int j = 0; j < PVR; j++) src_[j] = sp_[j * 16];             \
    src_[PVR] = sp_[nv_];                                                              \
    src_[PVR + 1] = sp_[nv_ + 16];                                                     \
    _Pragma("unroll") for (int j = 0; j < PVR; j++) {                                  \
      if (j * 16 < nv_) pvr[j] = GATHER_V(src_[j]); else pvr[j] = 0.0;                 \
    }                                                                                  \
    if (nv_ < nr_) prr[0] = GATHER_R(src_[PVR]); else prr[0] = 0.0;                    \
    if (nv_ + 16 < nr_) prr[1] = GATHER_R(src_[PVR + 1]); else prr[1] = 0.0;           \
  }

// timing probe (results wrong): the Jacobian operands of the row form read 16 CONSECUTIVE doubles per problem and instruction (from
// row 0's first operand on: the same 5 x 16 doubles a front of 16 band rows owns) instead of one operand of each row (lane stride
// = row length) — what do the vector-memory pipeline's extra line look-ups of the strided form cost?  1: forward, 2: backward, 3: both
// (round 4: nothing — 7.97 against 8.06 ms — although a strided gather costs a CU 58 cycles against 17, tools/vmem_bench.hip)
#ifdef CNL_DBG_COALJ
#define ROWJ_SRC(BIT, RECP, ROFF, J, SRCJ) ((CNL_DBG_COALJ & (BIT)) ? (((RECP) + (ROFF))[16] + 16 * (J) + l) : (SRCJ))
#else
#define ROWJ_SRC(BIT, RECP, ROFF, J, SRCJ) (SRCJ)
#endif
// Row form of the condensation products (plan.h, RF_ROWS): lane l takes residual row l of the front.  Seven gathers without a
// guard: the row's pivot (kept in pvr[ROWS_KM]), its ROWS_KM Jacobian operands (pvr[0 ..]) and its right-hand-side entry.
static_assert(PVR == ROWS_KM + 1, "the raw-value prefetch registers double as the row operands");
#define PREFETCH_ROWFORM(RECP, ROFF)                                                   \
  {                                                                                    \
    const int* sp_ = (RECP) + (ROFF) + l;                                              \
    int src_[ROWS_KM + 2];                                                             \
    _Pragma("unroll") for (int j = 0; j < ROWS_KM + 2; j++) src_[j] = sp_[j * 16];     \
    pvr[ROWS_KM] = GATHER_V(src_[0]);                                                  \
    _Pragma("unroll") for (int j = 0; j < ROWS_KM; j++) pvr[j] = GATHER_V(ROWJ_SRC(1, RECP, ROFF, j, src_[1 + j])); \
    prr[0] = GATHER_R(src_[ROWS_KM + 1]);                                              \
    prr[1] = 0.0;                                                                      \
  }
// Backward rows (plan.h, B_ROWS_FLAG): lane l gathers the operands of residual row l of the front whose record is RECP — pivot,
// ROWS_KM Jacobian entries, right-hand-side entry — one front ahead, like the panel rows.  ON == 0 (records without the
// sections): every source reads entry 0 and the index word names no row.  Operands a row does
// not have carry local index 1 (a finite x) and get the coefficient 0.
#define PREFETCH_BACKROWS(DST, IXW, RSRC, RECP, ROFF, ON)                              \
  {                                                                                    \
    const int* sp_ = (RECP) + (ROFF) + l;                                              \
    int src_[ROWS_KM + 3];                                                             \
    _Pragma("unroll") for (int j = 0; j < ROWS_KM + 3; j++) src_[j] = (ON) ? sp_[j * 16] : 0; \
    if (!(ON)) { src_[ROWS_KM + 1] = P.nnz; src_[ROWS_KM + 2] = 0x11111; }             \
    DST[0] = GATHER_V(src_[0]);                                                        \
    _Pragma("unroll") for (int j = 1; j < ROWS_KM + 1; j++) DST[j] = GATHER_V(ROWJ_SRC(2, RECP, ROFF, j - 1, src_[j])); \
    DST[ROWS_KM + 1] = GATHER_R(src_[ROWS_KM + 1]);                                    \
    IXW = src_[ROWS_KM + 2];                                                           \
    RSRC = src_[ROWS_KM + 1] - P.nnz;                                                  \
  }
// image position of pair K (compile-time) out of the lane's position words
#ifdef CNL_DBG_NOCONF   // timing probe (results wrong): the row products' atomics land on sixteen consecutive doubles per problem — what do their bank conflicts cost?
#define ROW_POS(PW, K) ((unsigned)(l + 16 * ((K) & 7)))
#else
#define ROW_POS(PW, K) (((unsigned)(PW)[(K) >> 2] >> (8 * ((K) & 3))) & 255u)
#endif

// ==========================================================================================
// dataflow execution of a staged plan: wait until *p >= target, then make the producer's global stores visible; signal = all
// stores of this wavefront first, then the counter.  The wait is bounded (a broken dependency — workgroups not dispatched in
// index order, a time-sliced device — must not hang the device): a wait that gives up is COUNTED in status_total / status_call,
// and the classic launch that follows every staged attempt then redoes the whole batch sequentially, so a timeout costs time,
// never a wrong result.  Once one wait of the call has given up the others stop waiting at once.
#ifdef CNL_DBG_TRACE   // experiment builds: records of four ints in a device buffer, printed by the launcher behind the ladder launch
__device__ int cnl_trace_buf[1 << 16];
__device__ int cnl_trace_n;
#define CNL_TRACE(TAG, A_, B_, C_) do { if ((threadIdx.x & 63) == 0) { const int i_ = atomicAdd(&cnl_trace_n, 4); if (i_ < (1 << 16) - 4) { cnl_trace_buf[i_] = (TAG) | ((int)blockIdx.x << 8); cnl_trace_buf[i_ + 1] = (A_); cnl_trace_buf[i_ + 2] = (B_); cnl_trace_buf[i_ + 3] = (C_); } } } while (0)
#else
#define CNL_TRACE(TAG, A_, B_, C_) do { } while (0)
#endif
// Every branch of the wait is WAVE-UNIFORM by construction (the polled values go through readfirstlane, the give-up bookkeeping is
// done by all lanes, 63 of which add zero).  Round 5, the root of the staged-execution fault of round 4 on plans with out-of-line
// front classes: with the natural form — `if (!ok && lane == 0) { atomics }` — the region is lane-divergent for the compiler, and
// in the instantiations that CALL slow_front right behind a wait it placed the register spills that belong in front of the call
// into the tail of that region, i.e. on a path the regular case never takes: the reloads behind the call then returned whatever
// earlier kernels had left in scratch memory (pointers among them).  Garbage-filled scratch made it deterministic, zeroed scratch
// hid it; ending the wavefront in the give-up branch (a tracing build) made it disappear.  (profiles/HISTORY.md 4c.)
__device__ __forceinline__ void spin_until(const int* p, int target, int limit, int* status_total, int* status_call, [[maybe_unused]] int site = 0) {
  int ok = 0;
  for (int it = 0; it < limit; it++) {
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= target) { ok = 1; break; }
    if ((it & 255) == 255 && __builtin_amdgcn_readfirstlane(__hip_atomic_load(status_call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) break;
    __builtin_amdgcn_s_sleep(4);
  }
#ifdef CNL_DBG_TRACE
  if (!ok) { CNL_TRACE(9, target, __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), site); __builtin_amdgcn_endpgm(); }
#endif
  if (!ok) {   // scalar branch
    const int one = (threadIdx.x & 63) == 0 ? 1 : 0;
    __hip_atomic_fetch_add(status_total, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(status_call, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#ifndef CNL_DF_NOFENCE   // (timing probe: -DCNL_DF_NOFENCE drops both fences; results are then not guaranteed)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
// Commit of the in-kernel ladder for problem g of a group (see the only_if_status prologue of the kernel): out of line, so that
// the hot instantiations pay nothing for it.
__device__ __noinline__ void ladder_commit(const int* lq, int g, int l, double* rho_old, double* slots, int nvar) {
  const double wr = reinterpret_cast<const double*>(lq + 8)[g], ro = reinterpret_cast<const double*>(lq + 16)[g];
  if (wr == 0.0) return;   // the problem never climbed
  if (l == 0) *rho_old = ro;
  for (int i = l; i < nvar; i += 16) slots[i] = wr;
}
__device__ __forceinline__ void task_done(int* counter, int lane, bool release = true) {
  if (!counter) return;
#ifndef CNL_DF_NOFENCE
  if (release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
  if (release) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two wavefronts per SIMD (256 VGPRs).  A wavefront alone on its SIMD walks a front in 11.2 k cycles and issues instructions
// 68 % of that time; two share a SIMD at 16.3 k cycles per front.  A 168-register variant with THREE per SIMD (the triangle
// image makes twelve wavefronts fit the LDS of a CU) was built and measured: 12 288 problems in 11.6 ms against 8 192 in
// 7.24 ms — 6 % less throughput; the CU-wide units (LDS, scalar/branch issue) are what the third wavefront queues for.
#ifndef CNL_WAVES_PER_SIMD
#define CNL_WAVES_PER_SIMD 2
#endif
// LEAN (round 3): plans whose fronts are all of the fast class with row-form (or no) condensation products — what the chain-like
// orders of band problems give — run an instantiation WITHOUT the cold paths: the forward substitution of MODE_SOLVE, the
// out-of-line front classes, raw-value staging and product lists.  Those paths are never executed for such plans, but they sit
// in the middle of the hot loop's code, cost registers (256 with spills against 187) and instruction-cache footprint: the
// lean instantiation is 10 % faster on the headline (same box: 965 k -> 1 058 k systems/s).
// SOLVE: the lean instantiation of solve_ldl! (MODE_SOLVE only): forward substitution + backward sweep with the residual components
// recovered by the backward records — nothing of the factorisation is compiled in, no post-pass behind the launch.
// FUSED (round 4): the staged instantiation that runs the in-kernel rho ladder (LaunchArgs.phase == 2) — a separate instantiation,
// because the loop over the rungs makes the forward sweep's state loop-carried (the plain staged kernel leaves after one sweep:
// 162 VGPRs, three wavefronts per SIMD; with the rung loop 213).  FUSED && !STAGED: the sequential launch BEHIND the fused ones
// (only_if_status): it commits the ladder's rho_old / rho slots, or redoes the call when a wait gave up — the commit is compiled
// into this instantiation only (in the hot lean kernel the call cost four more spilled SGPRs).
template <bool STAGED, bool LATE, bool LEAN, bool SOLVE = false, bool FUSED = false>
__global__ void __launch_bounds__(256, CNL_WAVES_PER_SIMD) newton2_kernel_t(const DevPlan2 Pin, const LaunchArgs Ain) {
  DevPlan2 P = Pin;
  P.rec = as_global(Pin.rec); P.brec = as_global(Pin.brec);
  LaunchArgs A = Ain;
  A.vals = as_global(Ain.vals); A.rhs = as_global(Ain.rhs); A.d = as_global(Ain.d); A.L = as_global(Ain.L);
  A.scratch = as_global(Ain.scratch); A.rho_old = as_global(Ain.rho_old); A.rho = as_global(Ain.rho);
  A.nfact = as_global(Ain.nfact); A.success = as_global(Ain.success); A.npos = as_global(Ain.npos); A.nzero = as_global(Ain.nzero);
  A.extra_pos = as_global(Ain.extra_pos); A.extra_zer = as_global(Ain.extra_zer);
  const int WPB = blockDim.x >> 6;
  extern __shared__ double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, l = lane & 15;
  int widx = blockIdx.x * WPB + wave;
  // STAGED: the wave runs ONE task (a subtree of the elimination tree, or a front above the cut) of one group of problems
  int t_rec = 0, t_nfr = P.nsuper, t_brec = 0, t_root = 1;
  int* dep_signal = nullptr;  // dataflow execution: the counter this wavefront bumps when its task is done
  // ... and the counter it waits for.  The wait is DEFERRED behind the prologue of the task (record load, value / panel
  // prefetch: three to four dependent round trips that need nothing from other tasks): a single system is ~17 task levels
  // deep, and the start-up latency of a task, not its fronts, is most of its critical path.
  const int* dep_wait = nullptr;
  int dep_target = 0;
  // A wave-uniform test BY CONSTRUCTION (readfirstlane): the pointer is the same in all lanes, but the compiler cannot know (it derives
  // from the wavefront's index), and a lane-divergent `if (dep_wait)` in front of a call made it place the register spills of the call
  // in front of the EXEC restore of the join block — reached with EXEC = 0 whenever there was nothing to wait for, so the spills stored
  // nothing (tools/check_spill_exec.py finds the pattern in the ISA; profiles/HISTORY.md 4c)
#define DEP_WAITING (__builtin_amdgcn_readfirstlane(dep_wait != nullptr ? 1 : 0) != 0)
  int t_nchild_l = 0, t_parent_l = -1, tix_l = 0;   // fused ladder launch (phase 2): the task's links, kept for every rung
  [[maybe_unused]] int* lad = nullptr;              // ... and the control block of its group of problems
  if constexpr (STAGED) {
    const bool fused = FUSED && A.phase == 2;
    const int per = fused ? A.lad_slots : A.nquads;   // groups of problems this launch covers
    int task = widx / per;
    if (task >= A.ntasks) return;
    widx -= task * per;
    if (fused) {
      widx += A.quad0;
      if (!A.lad_first) {  // behind a staged first attempt: only groups with a problem that failed it have work (the common case: none)
        const int pq = widx * 4 + (lane >> 4);
        if (!__any(pq < A.batch && as_global(A.success)[pq < A.batch ? pq : 0] == 0)) return;
      }
    }
    // Dataflow execution (A.dep): ONE launch per phase covers every task.  Workgroups are dispatched in index order, tasks
    // are sorted by stage, so a task's children (forward) have smaller indices; the backward launch runs the indices in
    // reverse, parents first.  A wavefront waits on a device counter for the tasks it depends on: the lowest unfinished
    // index is always resident and free to run, whatever the size of the grid.
    if (A.dep && A.phase == 1) task = A.ntasks - 1 - task;
    const int tix = A.task0 + task;
    const int32_t* tk = as_global(A.tasks) + 6 * tix;
    t_rec = rfl(tk[0]); t_nfr = rfl(tk[1]); t_brec = rfl(tk[2]); t_root = rfl(tk[3]);
    if (fused) {
      // all tasks of the group are resident at once: forward counters are monotone over the rungs (target = rung * children)
      t_parent_l = rfl(tk[4]); t_nchild_l = rfl(tk[5]); tix_l = tix;
      int* depf = as_global(A.ldep);
      if (t_nchild_l > 0) { dep_wait = depf + tix * A.nquads + widx; dep_target = t_nchild_l; }
      if (t_parent_l >= 0) dep_signal = depf + t_parent_l * A.nquads + widx;
      lad = as_global(A.lad) + (size_t)widx * LAD_WORDS;
    } else if (A.dep) {
      const int t_parent = rfl(tk[4]), t_nchild = rfl(tk[5]);
      int* depf = as_global(A.dep);
      int* depb = depf + A.ntasks_all * A.nquads;
      if (A.phase == 0) {
        if (t_nchild > 0 && A.df_live) { dep_wait = depf + tix * A.nquads + widx; dep_target = t_nchild; }
        if (t_parent >= 0) dep_signal = depf + t_parent * A.nquads + widx;
      } else {
        if (t_parent >= 0 && A.df_live) { dep_wait = depb + t_parent * A.nquads + widx; dep_target = 1; }
        dep_signal = depb + tix * A.nquads + widx;
      }
    }
  }
  const int nfr = t_nfr;
  const int prob0 = widx * 4;
  if (prob0 >= A.batch) return;
  const int prob = prob0 + g;
  bool valid_ = prob < A.batch;
  if (!STAGED && (A.skip_done || A.only_if_status)) {
    // behind a staged attempt: only the problems that failed it are processed — all of them when a dataflow wait of the
    // attempt gave up (its results are then not to be trusted)
    const bool redo_all = A.status_call && __hip_atomic_load(as_global(A.status_call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (A.only_if_status && !redo_all) {
      // Behind the in-kernel ladder (phase 2 launches): no wait gave up, so its results stand — COMMIT what the ladder must not
      // touch while a sequential redo may still need the caller's inputs: rho_old (in/out) and the rho slots of vals
      // (src/CaNNOLeS.jl:1031,1038,1044-1046).  One wavefront per group of four problems; nothing to do unless a problem climbed.
      if constexpr (FUSED) if (A.lad && A.mode == MODE_NEWTON && valid_) ladder_commit(as_global(A.lad) + (size_t)widx * LAD_WORDS, g, l, A.rho_old + prob, A.vals + (long long)prob * P.vstride + P.rho_begin, P.nvar);
      return;
    }
    if (A.skip_done && !redo_all && valid_ && as_global(A.success)[prob] == 1) valid_ = false;
    if (!__any(valid_)) return;
  }
  const bool valid = valid_;
  // A problem the launch skips (skip_done: it succeeded in the staged attempt) still walks the stream with its wave-mates: it
  // must read ITS OWN data, so that what it recomputes — and stores — is its own factor again.  (Until round 4 it was clamped to
  // the first problem of the wave like a problem past the end of the batch, and the elimination, which only knows the batch
  // size, overwrote its factor with the first problem's: a later solve_ldl! of that problem was wrong.)
  const bool inb = prob < A.batch;
  const long long pclamp = inb ? prob : prob0;

  const int wave_doubles = (P.recwords >> 1) + 4 * P.prob_doubles + 16;
  double* wbase = smem + wave * wave_doubles;
  int* recbuf = reinterpret_cast<int*>(wbase);  // forward: one record of up to reccap words; backward: two of breccap
  double* pbase0 = wbase + (P.recwords >> 1);
  int* cnt = reinterpret_cast<int*>(pbase0 + 4 * P.prob_doubles);
  double* myU = pbase0 + g * P.prob_doubles;
  double* myFs = myU + P.u2_peak;
  double* jraw = myU + P.jraw_off;  // raw values of the current front's products (reciprocal pivots first)
  if constexpr (!LEAN) {   // plan scalars of the out-of-line path (slow_front)
    if (lane == 0) {
      long long* ps = reinterpret_cast<long long*>(cnt + SLOW_ARGS_AT);
      ps[0] = P.prob_doubles; ps[1] = P.u2_peak; ps[2] = P.nnz; ps[3] = P.rho_begin;
      ps[4] = P.gs_doubles; ps[5] = P.lsize; ps[6] = P.vstride; ps[7] = P.rstride;
    }
    wsync();
  }

  Ctx2 c;
  c.vals = A.vals; c.rhs = A.rhs; c.L = A.L; c.gs = A.scratch; c.dout = A.d; c.batch = A.batch;
  const double* myvals = A.vals + pclamp * P.vstride;
  const bool has_rhs = !(A.mode == MODE_FACTOR || !A.rhs);  // wave-uniform
  const double* myrhs = has_rhs ? A.rhs + pclamp * P.rstride : nullptr;
  double* mygs = A.scratch + pclamp * P.gs_doubles;
  const double eig_tol = A.params[0];
  const int xpos = A.extra_pos ? A.extra_pos[pclamp] : 0, xzer = A.extra_zer ? A.extra_zer[pclamp] : 0;
  // gathers: wave-uniform bases of the first problem of the wave + 32-bit byte offsets (4 problems span < 4 GB)
  const int prob0u = __builtin_amdgcn_readfirstlane(prob0);
#ifdef CNL_DBG_VSTRIDE0   // timing probe (results wrong): every problem gathers problem 0's values — what do the gathers cost when they hit in cache?
  const char* vals_wb = reinterpret_cast<const char*>(A.vals);
#else
  const char* vals_wb = reinterpret_cast<const char*>(A.vals + (long long)prob0u * P.vstride);
#endif
  const char* rhs_wb = has_rhs ? reinterpret_cast<const char*>(A.rhs + (long long)prob0u * P.rstride) : vals_wb;
  const unsigned gsel = inb ? (unsigned)g : 0u;
#ifdef CNL_DBG_VSTRIDE0
  const unsigned gofs_v = 0u;
#else
  const unsigned gofs_v = gsel * (unsigned)P.vstride * 8u;
#endif
  const unsigned gofs_r = (gsel * (unsigned)(has_rhs ? P.rstride : P.vstride) - (unsigned)P.nnz) * 8u;  // rhs sources are nnz + index
  const char* L_wb = reinterpret_cast<const char*>(A.L + (long long)prob0u * P.lsize);
  const unsigned gofs_l = (gsel * (unsigned)P.lsize + (unsigned)l) * 8u;

  // per-problem ladder state, replicated over the 16 lanes of the group
  double rho = 0.0, wrote = 0.0;
  double rho_old = (A.mode == MODE_NEWTON) ? A.rho_old[pclamp] : 0.0;
  int nfact = 0;
  bool done = !valid, success = false, ovr = false;
  const double kdec = A.params[2], kinc = A.params[3], klarge = A.params[4], rho0 = A.params[5], rhomax = A.params[6],
               rhomin = A.params[7];
  [[maybe_unused]] int rung = 0;  // fused ladder: factorisations this launch has made of the group
  if constexpr (STAGED) {
    // behind a staged first attempt the problems that failed it enter the ladder at its first rung (src/CaNNOLeS.jl:1030)
    if (FUSED && A.phase == 2 && !A.lad_first && valid && A.success[prob] == 0) { rho = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old); ovr = true; }
  }

  // ---------------- MODE_SOLVE: forward substitution with the stored factor (solve_ldl!, src/solver_types.jl:69-77) ------
  // Only the right-hand-side column of every front is assembled (plain rhs entries, the products -J_ra rhs_r / d_r of the
  // condensed rows, the children's update vectors), then  c_a -= l_ia c_i  over the pivots from the top and z_i = c_i / d_i
  // goes into column 0 of the stored panel, where the backward sweep below expects it.  The host sends a plan here only
  // when every front is of the fast class (order <= 16, LDS staging).
  constexpr bool CNL_LEAN = LEAN;
  static_assert(!SOLVE || LEAN, "the solve-only instantiation is a lean one");
  static_assert(!FUSED || !SOLVE, "the fused ladder is an execution of newton_system");
  if ((SOLVE || (!CNL_LEAN && A.mode == MODE_SOLVE)) && (!STAGED || A.phase == 0)) {
    const int4* rstream = reinterpret_cast<const int4*>(P.rec);
    int4 R0, R1, R2;
    double pvr[PVR], prr[2], prh = 0.0;
    double lr[KB], lrn[KB];
    int roff = t_rec, nxt_off = 0;
    int* recw = recbuf;
    {
      int len = P.rec[t_rec + R_RECLEN];
      nxt_off = t_rec + len;
      if (len > P.reccap) len = P.reccap;
      for (int w4 = lane; w4 * 4 < len; w4 += 64) reinterpret_cast<int4*>(recw)[w4] = rstream[(t_rec >> 2) + w4];
      wsync();
      R0 = rstream[(nxt_off >> 2) + lane];
      R1 = rstream[(nxt_off >> 2) + lane + 64];
      R2 = rstream[(nxt_off >> 2) + lane + 128];
      const int hv0 = recw[lane & 15];
      const int nasm0 = HDRW(hv0, R_NASM), aoff0 = HDRW(hv0, R_ASM_OFF);
      PREFETCH_RHS_ROUND(recw, aoff0, HDRW(hv0, R_NASMV), nasm0)
      if (HDRW(hv0, R_FLAGS) & RF_ROWS) PREFETCH_ROWFORM(recw, aoff0 + 2 * nasm0)
      else PREFETCH_RAW(recw, aoff0 + 2 * nasm0, HDRW(hv0, R_NRD) >> 16, HDRW(hv0, R_NRAW))
      const long long lp0 = (long long)HDRW(hv0, R_LPTR_LO) | ((long long)HDRW(hv0, R_LPTR_HI) << 31);
      PREFETCH_ROWS(lr, lp0, HDRW(hv0, R_NUPD), HDRW(hv0, R_NPIV))
    }
    for (int s = 0; s < nfr; s++) {
      if constexpr (STAGED) if (DEP_WAITING) { spin_until(dep_wait, dep_target, A.spin_limit, as_global(A.status_total), as_global(A.status_call)); dep_wait = nullptr; }  // children's update vectors are read below
      const int* rec = recw;
      const int hv = rec[lane & 15];
      const int npiv = HDRW(hv, R_NPIV), nupd = HDRW(hv, R_NUPD), nasm = HDRW(hv, R_NASM), nasmv = HDRW(hv, R_NASMV);
      const int nchild = HDRW(hv, R_NCHILD), uoff = HDRW(hv, R_UOFF), flags = HDRW(hv, R_FLAGS) & 0xff;
      const int aoff = HDRW(hv, R_ASM_OFF), coff = HDRW(hv, R_CHILD_OFF);
      const int nprod = HDRW(hv, R_NPROD) & 0xffff, nraw = HDRW(hv, R_NRAW), nrdw = HDRW(hv, R_NRD);
      const int nrd = nrdw & 0xffff, nrawv = nrdw >> 16;
      const long long lptr = (long long)HDRW(hv, R_LPTR_LO) | ((long long)HDRW(hv, R_LPTR_HI) << 31);
      const bool uglob = flags & RF_U_GLOBAL;
      double* cvec = myFs;  // c_a = entry (a, 0) of the front, a = 0 .. f-1 (a = 0 unused)
      cvec[l] = 0.0;
      const int raw_off = aoff + 2 * nasm;
      const bool rowform = flags & RF_ROWS;
      if (!CNL_LEAN && nraw > 0 && !rowform) {
#pragma unroll
        for (int j = 0; j < PVR; j++)
          if (j * 16 < nrawv) {
            double v = pvr[j];
            if (j * 16 < nrd) {
              const double r = fast_div(-1.0, v);
              v = j * 16 + l < nrd ? r : v;
            }
            jraw[j * 16 + l] = v;
          }
        for (int e = PVR * 16 + l; e < nrawv; e += 16) {
          const double v = myvals[rec[raw_off + e]];
          jraw[e] = e < nrd ? fast_div(-1.0, v) : v;
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
          if (nrawv + q * 16 < nraw) jraw[nrawv + q * 16 + l] = prr[q];
      }
      wsync();
      // plain right-hand-side entries (positions (row, 0); padding entries sit in other columns and are skipped)
      if (nasmv < nasm) {
        const int pos = rec[aoff + nasm + nasmv + l], prow = tri_row(pos);
        if (pos == tri2(prow) && pos < FAST_IMG_TRI) __hip_atomic_fetch_add(&cvec[prow], prh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      for (int e = nasmv + 16 + l; e < nasm; e += 16) {
        const int src = rec[aoff + e], pos = rec[aoff + nasm + e], prow = tri_row(pos);
        if (pos == tri2(prow) && pos < FAST_IMG_TRI) __hip_atomic_fetch_add(&cvec[prow], myrhs[src - P.nnz], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      // products that land in column 0: in the row form the pairs (right-hand side, Jacobian operand q)
      if (rowform) {
        const int* rw = rec + raw_off + l;
        int pw_[ROWS_PW];
#pragma unroll
        for (int g = 0; g < ROWS_PW; g++) pw_[g] = rw[(2 + ROWS_KM + g) * 16];
        const double tr = prr[0] * fast_div(-1.0, pvr[ROWS_KM]);
#pragma unroll
        for (int q = 0; q < ROWS_KM; q++) {
          const int pos = (int)ROW_POS(pw_, ROWS_KM * (ROWS_KM + 1) / 2 + q), prow = tri_row(pos);
          if (pos == tri2(prow) && pos < FAST_IMG_TRI) __hip_atomic_fetch_add(&cvec[prow], tr * pvr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      } else if (!CNL_LEAN) {
        const int* pw = rec + raw_off + nraw + l;
        for (int e = 0; e < nprod; e += 16) {
          const int w = pw[e], pos = w & 255, prow = tri_row(pos);
          if (pos == tri2(prow) && pos < FAST_IMG_TRI) {
            const double v = jraw[(w >> 8) & 127] * jraw[(w >> 15) & 127] * jraw[(w >> 22) & 127];
            __hip_atomic_fetch_add(&cvec[prow], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
        }
      }
      // children's update vectors: entry a of a child goes to the row its (a, 0) entry maps to
      {
        int co = coff;
        for (int ci = 0; ci < nchild; ci++) {
          const int cv4 = rec[co + (lane & 3)];
          const int cu = HDRW(cv4, C_UOFF), tuc = HDRW(cv4, C_TUC), cfl = HDRW(cv4, C_FLAGS);
          if (tri2(l) < tuc) {
            const int prow = tri_row(rec[co + C_HDR + tri2(l)]);
            const double u = cfl ? mygs[cu + l] : myU[cu + l];
            __hip_atomic_fetch_add(&cvec[prow], u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          co += C_HDR + ((tuc + 3) & ~3);
        }
      }
      wsync();
      double cv = cvec[l];
      wsync();
      // next record over the current one, prefetches for the next front
      int nroff = nxt_off;
      if (s + 1 < nfr) {
        int* nrec = recbuf;
        const int nlen = __builtin_amdgcn_readlane(R0.z, 0);
        const int nasm1 = __builtin_amdgcn_readlane(R0.w, 0);
        const int nasmv1 = __builtin_amdgcn_readlane(R0.z, 2);
        const int aoff1 = __builtin_amdgcn_readlane(R0.w, 2);
        const int nraw1 = __builtin_amdgcn_readlane(R0.z, 3);
        const int nrawv1 = __builtin_amdgcn_readlane(R0.w, 3) >> 16;
        const int nflags1 = __builtin_amdgcn_readlane(R0.z, 1);
        const int npiv1 = __builtin_amdgcn_readlane(R0.x, 0), nupd1 = __builtin_amdgcn_readlane(R0.y, 0);
        const long long lp1 = (long long)__builtin_amdgcn_readlane(R0.x, 2) | ((long long)__builtin_amdgcn_readlane(R0.y, 2) << 31);
        static_assert(R_NPIV == 0 && R_NUPD == 1 && R_LPTR_LO == 8 && R_LPTR_HI == 9, "record header layout");
        const int clen = nlen < P.reccap ? nlen : P.reccap;
        if (lane * 4 < clen) reinterpret_cast<int4*>(nrec)[lane] = R0;
        if ((lane + 64) * 4 < clen) reinterpret_cast<int4*>(nrec)[lane + 64] = R1;
        if ((lane + 128) * 4 < clen) reinterpret_cast<int4*>(nrec)[lane + 128] = R2;
        wsync();
        for (int w4 = RN * 64 + lane; w4 * 4 < clen; w4 += 64) reinterpret_cast<int4*>(nrec)[w4] = rstream[(nxt_off >> 2) + w4];
        wsync();
        const int nn_off = nxt_off + nlen;
        R0 = rstream[(nn_off >> 2) + lane];
        R1 = rstream[(nn_off >> 2) + lane + 64];
        R2 = rstream[(nn_off >> 2) + lane + 128];
        PREFETCH_RHS_ROUND(nrec, aoff1, nasmv1, nasm1)
        if (nflags1 & RF_ROWS) PREFETCH_ROWFORM(nrec, aoff1 + 2 * nasm1)
        else PREFETCH_RAW(nrec, aoff1 + 2 * nasm1, nrawv1, nraw1)
        PREFETCH_ROWS(lrn, lp1, nupd1, npiv1)
        nxt_off = nn_off;
      } else {
#pragma unroll
        for (int k = 0; k < KB; k++) lrn[k] = lr[k];
      }
      // substitution over the pivots from the top: lane a holds c_a, row i of the panel is (l_i1 .. l_i,i-1, d_i) in lanes 1..i
      const int tu = tri2(1 + nupd);
      double* Lcol0 = A.L + pclamp * P.lsize + lptr - tu;  // entry (i, 0) of the panel at tri(i)
      for (int k0 = npiv - 1; k0 >= KB; k0--) {  // fronts with more than KB pivots: their top rows are loaded on demand
        const int i = nupd + 1 + k0;
        const double lv = A.L[pclamp * P.lsize + lptr + tri2(i) - tu + l];
        const double ci = bcast<16>(cv, i, (lane - l) * 4);
        if (l == i && valid) Lcol0[tri2(i)] = fast_div(cv, lv);
        if (l >= 1 && l < i) cv = fma(-lv, ci, cv);
      }
#pragma unroll
      for (int k = KB - 1; k >= 0; k--) {
        if (k < npiv) {
          const int i = nupd + 1 + k;
          const double ci = bcast<16>(cv, i, (lane - l) * 4);
          if (l == i && valid) Lcol0[tri2(i)] = fast_div(cv, lr[k]);
          if (l >= 1 && l < i) cv = fma(-lr[k], ci, cv);
        }
      }
      // update vector for the parent (entries 0 .. nupd; entry 0 is the unused corner)
      if (l <= nupd) {
        if (uglob) { if (valid) mygs[uoff + l] = cv; }
        else myU[uoff + l] = cv;
      }
      if (uglob) gsync(); else wsync();
#pragma unroll
      for (int k = 0; k < KB; k++) lr[k] = lrn[k];
      roff = nroff;
    }
    (void)roff;
    success = valid;
    gsync();  // the z column is read back through global memory by the backward sweep
    if constexpr (STAGED) { task_done(dep_signal, lane, A.df_live != 0); return; }  // the backward sweep of the tasks comes in a launch of its own
  }

  STAMP_DECL
  const bool do_fwd = !SOLVE && (STAGED ? ((A.phase == 0 || (FUSED && A.phase == 2)) && A.mode != MODE_SOLVE) : A.mode != MODE_SOLVE);
  while (do_fwd) {
    STAMP_BEGIN
    // ---------------- forward pass over the record stream ----------------
    if (l == 0) { cnt[g * 2] = STAGED ? 0 : xpos; cnt[g * 2 + 1] = STAGED ? 0 : xzer; }
    int rpos = 0, rzer = 0;  // per-lane tallies of the condensed residual pivots staged by this lane
    const int4* rstream = reinterpret_cast<const int4*>(P.rec);
    const bool needs_fix = __any(ovr) || !has_rhs;  // wave-uniform: some value must be replaced at assembly time
    int4 R0, R1, R2;  // record prefetch registers (named values: an array would be kept in scratch)
    double pv[PVN], pvr[PVR], prr[2], prh = 0.0;
#pragma unroll
    for (int j = 0; j < PVN; j++) pv[j] = 0.0;
#pragma unroll
    for (int j = 0; j < PVR; j++) pvr[j] = 0.0;
    prr[0] = prr[1] = 0.0;
    int roff = t_rec;  // word offset of the current record
    int nxt_off = 0;   // word offset of the next record
    int s = 0;
    LPend LP;          // L rows of the front just eliminated, stored one front late (flush_lrows)
    LP.pm = 0; LP.lofs = 0; LP.lptr = 0;
    const int lp_bm = valid ? l : 64;
    // Outer loop: (re)start the pipeline at front s, then either hand a rare large front to the out-of-line path
    // or run the inner loop over a stretch of fast fronts.  The inner loop contains NO call: values that live
    // across a call are spilled, and every reload from scratch (a VMEM access) waits for all outstanding
    // prefetches, which used to stall every front.
    while (s < nfr) {
      int* recw = recbuf;
      {
        // record s synchronously, then prefetch record s+1 and the values of s
        int len = P.rec[roff + R_RECLEN];
        nxt_off = roff + len;
        if (len > P.reccap) len = P.reccap;
        for (int w4 = lane; w4 * 4 < len; w4 += 64) reinterpret_cast<int4*>(recw)[w4] = rstream[(roff >> 2) + w4];
        wsync();
      }
      {
        const int hv0 = recw[lane & 15];
        const int fw0 = HDRW(hv0, R_FLAGS);  // flags | class << 8
        const bool fast0 = (fw0 >> 8) == 16 && !(fw0 & RF_FS_GLOBAL);
        if (!CNL_LEAN && !fast0) {
          // rare: large or globally staged front, handled out of line
          if constexpr (STAGED) if (DEP_WAITING) { spin_until(dep_wait, dep_target, A.spin_limit, as_global(A.status_total), as_global(A.status_call), 1); dep_wait = nullptr; }
          if (!(CNL_ABL & 2048)) slow_front(P.rec, A.vals, has_rhs ? A.rhs : nullptr, A.L, A.scratch, A.batch, lane, prob0, recw, roff, pbase0, cnt, eig_tol, rho, ovr, P.count_d != 0);
          gsync();
          roff = nxt_off;
          s++;
          continue;
        }
        R0 = rstream[(nxt_off >> 2) + lane];  // stream is padded: over-read is safe
        R1 = rstream[(nxt_off >> 2) + lane + 64];
        R2 = rstream[(nxt_off >> 2) + lane + 128];
        const int nasm0 = HDRW(hv0, R_NASM), aoff0 = HDRW(hv0, R_ASM_OFF);
        PREFETCH_VALUES(recw, aoff0, HDRW(hv0, R_NASMV), nasm0)
        if (fw0 & RF_ROWS) PREFETCH_ROWFORM(recw, aoff0 + 2 * nasm0)
        else PREFETCH_RAW(recw, aoff0 + 2 * nasm0, HDRW(hv0, R_NRD) >> 16, HDRW(hv0, R_NRAW))
      }
      bool more = true;
      bool img_clean = false;  // the staging image is known to be all zeros
      while (more) {
      if constexpr (STAGED) if (DEP_WAITING) { spin_until(dep_wait, dep_target, A.spin_limit, as_global(A.status_total), as_global(A.status_call), 2); dep_wait = nullptr; }  // children's update matrices are read below
      const int* rec = recw;
      const int hv = rec[lane & 15];
      const int npiv = HDRW(hv, R_NPIV), nupd = HDRW(hv, R_NUPD), nasm = HDRW(hv, R_NASM);
      const int nchild = HDRW(hv, R_NCHILD), uoff = HDRW(hv, R_UOFF), flags = HDRW(hv, R_FLAGS);
      const int nasmv = HDRW(hv, R_NASMV), aoff = HDRW(hv, R_ASM_OFF), coff = HDRW(hv, R_CHILD_OFF);
      const int nprodw = HDRW(hv, R_NPROD), nraw = HDRW(hv, R_NRAW), nrdw = HDRW(hv, R_NRD);
      const int nprod = nprodw & 0xffff, nrd = nrdw & 0xffff, nrawv = nrdw >> 16;
      const int nrd_own = P.count_d ? nprodw >> 16 : 0;  // residual pivots this front counts in the inertia
      const long long lptr = (long long)HDRW(hv, R_LPTR_LO) | ((long long)HDRW(hv, R_LPTR_HI) << 31);
      const int f = 1 + nupd + npiv;
      const bool uglob = flags & RF_U_GLOBAL;
      // (1) zero the staging image (packed triangle + padding slots = 152 doubles = 76 pairs: no loop; the fifth round covers 12)
      //     — only at the start of a stretch of fast fronts: inside it the previous front's elimination left the image zeroed
      if (!img_clean) {
        double2* z2 = reinterpret_cast<double2*>(myFs) + l;
#pragma unroll
        for (int j = 0; j < 4; j++) z2[16 * j] = make_double2(0.0, 0.0);
        if (l < FAST_IMG_DOUBLES / 2 - 64) z2[64] = make_double2(0.0, 0.0);
        wsync();
      }
      img_clean = !(CNL_ABL & 1024);
      // (2) extend-add the children's update matrices: needs nothing from global memory, so the stores of the
      //     previous front (L rows) retire behind it before the prefetched values are waited for
      {
        int co = coff;
        for (int ci = 0; ci < ((CNL_ABL & 16) ? 0 : nchild); ci++) {
          const int cv = rec[co + (lane & 3)];
          const int cu = HDRW(cv, C_UOFF), tuc = HDRW(cv, C_TUC), cfl = HDRW(cv, C_FLAGS);
          const int* dest = rec + co + C_HDR;
          if (!cfl) {
            // four rounds in flight; reads past the end of the list / matrix are not used
            const double* U = myU + cu + l;
            const int* dl = dest + l;
            for (int t = 0; t < tuc; t += 64) {
              int dp[4];
              double uv[4];
#pragma unroll
              for (int q = 0; q < 4; q++) { dp[q] = dl[t + 16 * q]; uv[q] = U[t + 16 * q]; }
#pragma unroll
              for (int q = 0; q < 4; q++)
                if (t + 16 * q + l < tuc) __hip_atomic_fetch_add(&myFs[dp[q]], uv[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          } else {
            const double* Ug = mygs + cu;
            for (int t = l; t < tuc; t += 16)
              __hip_atomic_fetch_add(&myFs[dest[t]], Ug[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          co += C_HDR + ((tuc + 3) & ~3);
        }
      }
      // (3) assemble the prefetched values (then any overflow)
      const int raw_off = aoff + 2 * nasm;
      const bool rowform = flags & RF_ROWS;
      if (!CNL_LEAN && nraw > 0 && !rowform && !(CNL_ABL & 4)) {
        // on-the-fly condensation: raw values to LDS, matrix values first (the first nrd are the residual pivots
        // d_r: keep -1/d_r), then the right-hand-side operands (a missing right-hand side reads as zero)
#pragma unroll
        for (int j = 0; j < PVR; j++)
          if (j * 16 < nrawv) {
            double v = pvr[j];
            if (j * 16 < nrd) {
              // the pivots d_r this front owns are counted here (src/solver_types.jl:90-95): per-lane tallies, summed
              // over the 16 lanes once per factorisation
              const bool own = j * 16 + l < nrd_own;
              rpos += own && v > eig_tol;
              rzer += own && fabs(v) <= eig_tol;
              const double r = fast_div(-1.0, v);
              v = j * 16 + l < nrd ? r : v;
            }
            jraw[j * 16 + l] = v;
          }
        for (int e = PVR * 16 + l; e < nrawv; e += 16) {
          const double v = myvals[rec[raw_off + e]];
          rpos += e < nrd_own && v > eig_tol;
          rzer += e < nrd_own && fabs(v) <= eig_tol;
          jraw[e] = e < nrd ? fast_div(-1.0, v) : v;
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
          if (nrawv + q * 16 < nraw) jraw[nrawv + q * 16 + l] = has_rhs ? prr[q] : 0.0;
      }
      wsync();
      {
        // positions read in one batch ahead of the guards (see PREFETCH_VALUES); the prefetched values go in as they
        // are unless some problem of the wave overrides rho
        int pos[PVN + 1];
#pragma unroll
        for (int j = 0; j < PVN; j++) pos[j] = rec[aoff + nasm + j * 16 + l];
        pos[PVN] = rec[aoff + nasm + nasmv + l];
        if (!needs_fix) {
#pragma unroll
          for (int j = 0; j < PVN; j++)
            if (j * 16 < nasmv && !(CNL_ABL & 2)) __hip_atomic_fetch_add(&myFs[pos[j]], pv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
#pragma unroll
          for (int j = 0; j < PVN; j++)
            if (j * 16 < nasmv) {
              const int src = rec[aoff + j * 16 + l];
              const double v = (ovr && src >= P.rho_begin) ? rho : pv[j];
              __hip_atomic_fetch_add(&myFs[pos[j]], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        if (nasmv < nasm) __hip_atomic_fetch_add(&myFs[pos[PVN]], has_rhs ? prh : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      for (int e = PVN * 16 + l; e < nasmv; e += 16) {  // matrix entries beyond the prefetched rounds
        const int src = rec[aoff + e], pos = rec[aoff + nasm + e];
        const double v = (ovr && src >= P.rho_begin) ? rho : myvals[src];
        __hip_atomic_fetch_add(&myFs[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      for (int e = nasmv + 16 + l; e < nasm; e += 16) {  // right-hand-side entries beyond the prefetched round
        const int src = rec[aoff + e], pos = rec[aoff + nasm + e];
        __hip_atomic_fetch_add(&myFs[pos], myrhs ? myrhs[src - P.nnz] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      if (rowform && !(CNL_ABL & 1)) {
        // row form: lane l = residual row l.  w = -1/d_r once, then (J_p w) J_q to the position byte of pair (p, q); lanes
        // without a row multiply dummy operands into the padding slots.  The pivots this front owns are counted here
        // (src/solver_types.jl:90-95).
        const int* rw = rec + raw_off + l;
        int pw_[ROWS_PW];
#pragma unroll
        for (int g = 0; g < ROWS_PW; g++) pw_[g] = rw[(2 + ROWS_KM + g) * 16];
        const double dv = pvr[ROWS_KM];
        const bool own = l < nrd_own;
        rpos += own && dv > eig_tol;
        rzer += own && fabs(dv) <= eig_tol;
        const double w = fast_div(-1.0, dv);
        double t_[ROWS_KM];
#pragma unroll
        for (int q = 0; q < ROWS_KM; q++) t_[q] = pvr[q] * w;
        const double tr = (has_rhs ? prr[0] : 0.0) * w;
#pragma unroll
        for (int p_ = 0; p_ < ROWS_KM; p_++)
#pragma unroll
          for (int q = 0; q <= p_; q++)
            __hip_atomic_fetch_add(&myFs[ROW_POS(pw_, p_ * (p_ + 1) / 2 + q)], t_[p_] * pvr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
        for (int q = 0; q < ROWS_KM; q++)
          __hip_atomic_fetch_add(&myFs[ROW_POS(pw_, ROWS_KM * (ROWS_KM + 1) / 2 + q)], tr * pvr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      if (!CNL_LEAN && nprod > 0 && !(CNL_ABL & 1)) {
        // products -J_ra J_rb / d_r of the condensed residual rows: one packed word each, pos | ia<<8 | ib<<15 | id<<22
        // (PB rounds in flight: the LDS round trips of a round are dependent, those of different rounds are not)
        wsync();
#ifndef CNL_PB
#define CNL_PB 8
#endif
        constexpr int PB = CNL_PB;
        const int* pw = rec + raw_off + nraw + l;
        for (int e = 0; e < nprod; e += 16 * PB) {
          int w[PB];
          double v[PB];
#pragma unroll
          for (int q = 0; q < PB; q++) w[q] = pw[e + 16 * q];  // reads past the list stay inside the wave's LDS and are not used
#pragma unroll
          for (int q = 0; q < PB; q++) v[q] = jraw[(w[q] >> 8) & 127] * jraw[(w[q] >> 15) & 127] * jraw[(w[q] >> 22) & 127];
#pragma unroll
          for (int q = 0; q < PB; q++)
            if (e + 16 * q < nprod) __hip_atomic_fetch_add(&myFs[w[q] & 255], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      }
      STAMP(0)
      wsync();
      STAMP(2)
      // (4) next record over the current one (nothing below reads the lists); prefetch the one after and the next front's values
      int nroff = nxt_off;
      more = false;
      if (s + 1 < nfr) {
        int* nrec = recbuf;
        // header words of the next record straight from the prefetch registers: lane q holds words 4q .. 4q+3
        const int nlen = __builtin_amdgcn_readlane(R0.z, 0);          // R_RECLEN = 2
        const int nasm1 = __builtin_amdgcn_readlane(R0.w, 0);         // R_NASM = 3
        const int nflags1 = __builtin_amdgcn_readlane(R0.z, 1);       // R_FLAGS = 6 (flags | class << 8)
        const int nasmv1 = __builtin_amdgcn_readlane(R0.z, 2);        // R_NASMV = 10
        const int aoff1 = __builtin_amdgcn_readlane(R0.w, 2);         // R_ASM_OFF = 11
        const int nraw1 = __builtin_amdgcn_readlane(R0.z, 3);         // R_NRAW = 14
        const int nrawv1 = __builtin_amdgcn_readlane(R0.w, 3) >> 16;  // R_NRD = 15 (nrd | nrawv << 16)
        static_assert(R_RECLEN == 2 && R_NASM == 3 && R_FLAGS == 6 && R_NASMV == 10 && R_ASM_OFF == 11 && R_NRAW == 14 && R_NRD == 15, "record header layout");
        const int clen = nlen < P.reccap ? nlen : P.reccap;   // globally staged fronts keep only their head in LDS
        if (lane * 4 < clen) reinterpret_cast<int4*>(nrec)[lane] = R0;
        if ((lane + 64) * 4 < clen) reinterpret_cast<int4*>(nrec)[lane + 64] = R1;
        if ((lane + 128) * 4 < clen) reinterpret_cast<int4*>(nrec)[lane + 128] = R2;
        wsync();
        for (int w4 = RN * 64 + lane; w4 * 4 < clen; w4 += 64) reinterpret_cast<int4*>(nrec)[w4] = rstream[(nxt_off >> 2) + w4];
        wsync();
        // the previous front's L rows: behind the last use of a prefetched register (a wait for it would wait for these
        // stores), ahead of this step's loads (whoever waits for those finds the stores a whole elimination old)
        flush_lrows(LP, const_cast<char*>(L_wb), lp_bm);
        const int nn_off = nxt_off + nlen;
        R0 = rstream[(nn_off >> 2) + lane];
        R1 = rstream[(nn_off >> 2) + lane + 64];
        R2 = rstream[(nn_off >> 2) + lane + 128];
        const bool nfast = (nflags1 >> 8) == 16 && !(nflags1 & RF_FS_GLOBAL);
        if (!(CNL_ABL & 8)) {
        PREFETCH_VALUES(nrec, aoff1, nfast ? nasmv1 : 0, nfast ? nasm1 : 0)
        if (nfast && (nflags1 & RF_ROWS)) PREFETCH_ROWFORM(nrec, aoff1 + 2 * nasm1)
        else PREFETCH_RAW(nrec, aoff1 + 2 * nasm1, nfast ? nrawv1 : 0, nfast ? nraw1 : 0)
        }
        nxt_off = nn_off;
        more = nfast;  // a large front ends the stretch: the outer loop takes over
      }
      else flush_lrows(LP, const_cast<char*>(L_wb), lp_bm);
      STAMP(1)
      // (5) eliminate in registers, store L rows and the update matrix
      if (!(CNL_ABL & 1024)) {
        const int bandw = HDRW(hv, R_FSOFF);   // fast fronts: the band form (0: none), see eliminate16_dpp
        if (bandw == (2 | (CNL_BAND_HW << 8))) eliminate16_dpp<LATE, 2>(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, c.L, c.gs, c.batch, lane, prob0, f, nupd, lptr, uoff, uglob, pbase0, cnt, eig_tol, LP);
        else if (bandw == (3 | (CNL_BAND_HW << 8))) eliminate16_dpp<LATE, 3>(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, c.L, c.gs, c.batch, lane, prob0, f, nupd, lptr, uoff, uglob, pbase0, cnt, eig_tol, LP);
        else eliminate16_dpp<LATE, 0>(P.prob_doubles, P.u2_peak, P.gs_doubles, P.lsize, c.L, c.gs, c.batch, lane, prob0, f, nupd, lptr, uoff, uglob, pbase0, cnt, eig_tol, LP);
      }
      STAMP(3)
      if (uglob) gsync(); else wsync();
      roff = nroff;
      s++;
      STAMP(4)
      }
      flush_lrows(LP, const_cast<char*>(L_wb), lp_bm);  // end of a stretch of fast fronts: nothing stays pending
    }
    // ---------------- inertia test and rho ladder (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047) ----
    wsync();
    for (int o = 8; o > 0; o >>= 1) { rpos += __shfl_xor(rpos, o, 16); rzer += __shfl_xor(rzer, o, 16); }
    const int tpos = cnt[g * 2] + rpos, tzer = cnt[g * 2 + 1] + rzer;
    if constexpr (STAGED) {  // the counts of all tasks meet in global memory; the backward launches read them
      int* gcw = as_global((FUSED && A.phase == 2) ? A.lgcnt : A.gcnt);
      if (valid && l == 0) {
        if (tpos) atomicAdd(gcw + prob * 2, tpos);
        if (tzer) atomicAdd(gcw + prob * 2 + 1, tzer);
      }
      if (!FUSED || A.phase != 2) {
        task_done(dep_signal, lane, A.df_live != 0);
        return;
      }
      if constexpr (FUSED) {
      // ---- fused ladder: the last task of the group to finish this rung decides (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047)
      rung++;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_wave_barrier();
      int fin = 0;
      if (lane == 0) {
        if (dep_signal) __hip_atomic_fetch_add(dep_signal, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin = __hip_atomic_fetch_add(lad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fin = rfl(fin);
      int* st_w = lad + 4;
      double* wr_w = reinterpret_cast<double*>(lad + 8);
      double* ro_w = reinterpret_cast<double*>(lad + 16);
      int epoch;
      if (fin == rung * A.ntasks_all - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // ladder state of problem g (replicated over its 16 lanes): from the control block, or — first rung — as the launch
        // found it (after a staged first attempt: rho = 0, one factorisation, success flag)
        int st = 0, nf = 0;
        double rh = 0.0, wr = 0.0;
        if (rung > 1) {
          st = __hip_atomic_load(st_w + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          nf = __hip_atomic_load(A.nfact + pclamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          rh = __hip_atomic_load(A.rho + pclamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          wr = __hip_atomic_load(wr_w + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!A.lad_first) {
          st = A.success[pclamp] == 1 ? 1 : 0;
          nf = 1;
          if (!st) { rh = rho; wr = rho; }   // the first rung's rho was formed in the prologue of every wavefront
        }
        if (!valid) st = 1;
        const int tp = __hip_atomic_load(gcw + pclamp * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int tz = __hip_atomic_load(gcw + pclamp * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == 0) {
          nf++;
          if (tp == P.nvar && tz == 0) st = 1;
          else if (nf == 1) { rh = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old); wr = rh; }
          else if (rh <= rhomax) {
            rh = rho_old == 0.0 ? klarge * rh : kinc * rh;
            if (rh <= rhomax) wr = rh; else st = 2;
          } else st = 2;
        }
        const bool fin_all = __all(st != 0);
        if (valid && l == 0) {
          __hip_atomic_store(st_w + g, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(wr_w + g, wr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(A.rho + prob, rh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(A.nfact + prob, nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gcw + prob * 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(gcw + prob * 2 + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (fin_all) {
            A.success[prob] = st == 1 ? 1 : 0;
            ro_w[g] = (nf > 1 && rh <= rhomax) ? rh : rho_old;   // committed by the launch behind this one (only_if_status)
          }
        }
        epoch = 2 * rung + (fin_all ? 1 : 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(lad + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        spin_until(lad + 1, 2 * rung, A.spin_limit, as_global(A.status_total), as_global(A.status_call), 3);
        epoch = rfl(__hip_atomic_load(lad + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      // a wait that gave up anywhere: the sequential launch behind this one redoes the call; leave
      if (rfl(__hip_atomic_load(as_global(A.status_call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
      if (epoch & 1) {
        success = valid && __hip_atomic_load(st_w + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
        break;
      }
      // next rung: every problem of the group is factorised again with its current rho (a problem that is done repeats its
      // last factorisation: same factor)
      rho = __hip_atomic_load(A.rho + pclamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ovr = valid && rho != 0.0;
      if (t_nchild_l > 0) { dep_wait = as_global(A.ldep) + tix_l * A.nquads + widx; dep_target = (rung + 1) * t_nchild_l; }
      continue;
      }
    }
    const bool ok = (CNL_ABL != 0) || (tpos == P.nvar && tzer == 0);
    if (A.mode == MODE_FACTOR) {
      if (valid && l == 0) {
        A.success[prob] = ok ? 1 : 0;
        if (A.npos) A.npos[prob] = tpos;
        if (A.nzero) A.nzero[prob] = tzer;
      }
      return;
    }
    if (!done) {
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
    wsync();
    if (__all(done)) break;
  }
  if (nfact > 1) {
    if (rho <= rhomax) rho_old = rho;
    if (valid) {
      double* vt = A.vals + pclamp * P.vstride + P.rho_begin;
      for (int i = l; i < P.nvar; i += 16) vt[i] = wrote;
    }
  }
  if constexpr (STAGED) {
    if (FUSED && A.phase == 2) {
      // fused ladder: `success` is the group's decision; the backward sweeps of the tasks run in dataflow fashion, parents first
      int* depb = as_global(A.ldep) + A.ntasks_all * A.nquads;
      dep_wait = nullptr;
      if (t_parent_l >= 0) { dep_wait = depb + t_parent_l * A.nquads + widx; dep_target = 1; }
      dep_signal = depb + tix_l * A.nquads + widx;
    } else {  // inertia rule on the sums of the forward launches (src/solver_types.jl:90-97)
      const int* gc = as_global(A.gcnt) + pclamp * 2;
      success = valid && (A.mode == MODE_SOLVE || (gc[0] == P.nvar && gc[1] == 0));  // solve_ldl! follows a successful factorisation
    }
    nfact = 1;
  }
  if (l == 0) cnt[8 + g] = (success && valid) ? 1 : 0;
  gsync();
  STAMP(5)
  // ---------------- backward pass (d = -K^-1 rhs), only where the factorisation succeeded -----------
  // (problems that failed still walk the stream with the wave; their output is not stored)
  if (__any(success) && !(CNL_ABL & 256)) {
    const int4* bstream = reinterpret_cast<const int4*>(P.brec);
    const int* okflag = cnt + 8;
    const bool okme = valid && okflag[g] != 0;
    const double* myL = A.L + pclamp * P.lsize;
    double* mydout = A.d + pclamp * P.dstride;
    double* xs = myU;  // the x stack reuses the per-problem LDS area
    int boff = t_brec, nxt = 0;
    int4 Rb;
    double lr[KB];     // panel rows of the CURRENT front (first KB pivots), prefetched one front ahead
    bool primed = false;
    // the scattered store of a front's solution components is issued one front late, right before the NEXT front's prefetch
    // loads: a wait for any load is a vmcnt(0) while a store is in flight (see LPend), and a store issued at the end of front s
    // put its whole round trip on the critical path of front s + 1
    int ipend = -1;
    double dpend = 0.0;
    // (lean) operands of the residual rows the CURRENT front owns, prefetched one front ahead; their store is deferred too
    const bool brows = CNL_LEAN && A.back_rows != 0 && !(CNL_ABL & 32768);
    double bpv[ROWS_KM + 2];
    int bix = 0, brs = 0, ipend2 = -1;
    double dpend2 = 0.0;
    int s = 0;
    while (s < nfr) {
      int* recw = recbuf + (s & 1) * P.breccap;
      if (!primed) {
        const int len = P.brec[boff + B_RECLEN];
        nxt = boff + len;
        for (int w4 = lane; w4 * 4 < len; w4 += 64) reinterpret_cast<int4*>(recw)[w4] = bstream[(boff >> 2) + w4];
        wsync();
        Rb = bstream[(nxt >> 2) + lane];  // padded stream
        const int hb0 = recw[lane & 7];
        const int nupd0 = HDRW(hb0, B_NUPD), npiv0 = HDRW(hb0, B_NPIV);
        const long long lp0 = (long long)HDRW(hb0, B_LPTR_LO) | ((long long)HDRW(hb0, B_LPTR_HI) << 31);
        if constexpr (CNL_LEAN) PREFETCH_BACKROWS(bpv, bix, brs, recw, (B_HDR + 1 + nupd0 + npiv0 + 3) & ~3, brows)
        PREFETCH_ROWS(lr, lp0, nupd0, npiv0)
        primed = true;
      }
      if constexpr (STAGED) if (DEP_WAITING) { spin_until(dep_wait, dep_target, A.spin_limit, as_global(A.status_total), as_global(A.status_call), 4); dep_wait = nullptr; }  // the parent's x is read below
      const int* rec = recw;
      const int hb = rec[lane & 7];
      const int npiv = HDRW(hb, B_NPIV), nupd = HDRW(hb, B_NUPD), xoff = HDRW(hb, B_XOFF), pxoff = HDRW(hb, B_PXOFF);
      const int cls = HDRW(hb, B_CLS) & 255;
      const long long lptr = (long long)HDRW(hb, B_LPTR_LO) | ((long long)HDRW(hb, B_LPTR_HI) << 31);
      const int f = 1 + nupd + npiv;
      if (!CNL_LEAN && cls != 16) {
        // rare large front: out of line, then restart the pipeline
        if (ipend >= 0) { mydout[ipend] = dpend; ipend = -1; }
        if (CNL_ABL & 4096) {
        } else if (cls == 32) {
          for (int pass = 0; pass < 2; pass++) {
            if (prob0 + pass * 2 >= A.batch) break;
            back_front_call<32>(P.prob_doubles, P.lsize, P.dstride, A.L, A.d, A.batch, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, okflag);
          }
        } else {
          for (int pass = 0; pass < 4; pass++) {
            if (prob0 + pass >= A.batch) break;
            back_front_call<64>(P.prob_doubles, P.lsize, P.dstride, A.L, A.d, A.batch, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, okflag);
          }
        }
        wsync();
        boff = nxt;
        s++;
        primed = false;
        continue;
      }
      // next record into the other buffer, then prefetch the record after it and the next front's panel rows
      double lrn[KB];
      double bpvn[ROWS_KM + 2];
      int bixn = 0, brsn = 0;
      int nboff = nxt;
      if (s + 1 < nfr) {
        int* nrec = recbuf + ((s + 1) & 1) * P.breccap;
        // next header words straight from the prefetch registers (lane 0: words 0..3, lane 1: words 4..7)
        const int nlen = __builtin_amdgcn_readlane(Rb.z, 0);   // B_RECLEN = 2
        const int npiv1 = __builtin_amdgcn_readlane(Rb.x, 0), nupd1 = __builtin_amdgcn_readlane(Rb.y, 0);
        const long long lp1 = (long long)__builtin_amdgcn_readlane(Rb.y, 1) | ((long long)__builtin_amdgcn_readlane(Rb.z, 1) << 31);
        static_assert(B_NPIV == 0 && B_NUPD == 1 && B_RECLEN == 2 && B_LPTR_LO == 5 && B_LPTR_HI == 6, "backward header layout");
        if (lane * 4 < nlen) reinterpret_cast<int4*>(nrec)[lane] = Rb;
        wsync();
        for (int w4 = 64 + lane; w4 * 4 < nlen; w4 += 64) reinterpret_cast<int4*>(nrec)[w4] = bstream[(nxt >> 2) + w4];
        wsync();
        if (ipend >= 0) { mydout[ipend] = dpend; ipend = -1; }   // the previous front's solution components (see above)
        if constexpr (CNL_LEAN) if (ipend2 >= 0) { mydout[ipend2] = dpend2; ipend2 = -1; }
        const int nn = nxt + nlen;
        Rb = bstream[(nn >> 2) + lane];
        nxt = nn;
        if constexpr (CNL_LEAN) PREFETCH_BACKROWS(bpvn, bixn, brsn, nrec, (B_HDR + 1 + nupd1 + npiv1 + 3) & ~3, brows)
        PREFETCH_ROWS(lrn, lp1, nupd1, npiv1)
      } else {
#pragma unroll
        for (int k = 0; k < KB; k++) lrn[k] = lr[k];
        if constexpr (CNL_LEAN) {
#pragma unroll
          for (int k = 0; k < ROWS_KM + 2; k++) bpvn[k] = bpv[k];
        }
      }
      // x of the update rows from the parent's vector (in place when this front reuses the parent's slot).
      // Lane l keeps x of local row l; rows not known yet hold 0, so the dot product of a pivot row needs no lane
      // predicate: entries beyond the row multiply zeros.  Lane 0 (the right-hand-side column, where the panel
      // keeps z = D^-1 L^-1 b) holds -1: the lane sum is then (L x) - z = -x_pivot, no separate broadcast of z.
      const int tu = tri2(1 + nupd);
      double xb = l == 0 ? -1.0 : 0.0;
      if (pxoff >= 0 && l >= 1 && l <= nupd) xb = xs[pxoff + rec[B_HDR + l]];
      if (pxoff == B_PX_GLOBAL && l >= 1 && l <= nupd)  // the parent was solved by another task: x = -d of the named components
        xb = -__hip_atomic_load(mydout + rec[B_HDR + l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wsync();
#pragma unroll
      for (int k = 0; k < KB; k++) {
        if (k < npiv && !(CNL_ABL & 8192)) {
          const double sum = gsum<16>(-lr[k] * xb);
          if (l == nupd + 1 + k) xb = sum;
        }
      }
      for (int k0 = KB; k0 < npiv; k0++) {  // fronts with more than KB pivots: remaining rows loaded on demand
        const int i = nupd + 1 + k0;
        const double lv = myL[lptr + tri2(i) - tu + l];
        const double sum = gsum<16>(-lv * xb);
        if (l == i) xb = sum;
      }
      // d = -x of the pivots: one scattered store per front (rec holds the original index of every pivot)
      if (ipend >= 0) mydout[ipend] = dpend;   // (only behind the last prefetch: the front before this one had no successor to carry it)
      ipend = (okme && l > nupd && l < f && !(CNL_ABL & 65536)) ? rec[B_HDR + l] : -1;
      dpend = -xb;
      if (l >= 1 && l < f) xs[xoff + l] = xb;
      wsync();
      if constexpr (CNL_LEAN) {
        // residual components of the rows this front owns: d_r = (sum_p J_p x[l_p] - rhs_r) / d_r with x of the front's own
        // vector (every column of an owned row is a row of this front)
        const double* xf = xs + xoff;
        const int nm = (bix >> 20) & 7;
        double sacc = -bpv[ROWS_KM + 1];
#pragma unroll
        for (int p_ = 0; p_ < ROWS_KM; p_++) {
          const double cf = p_ < nm ? bpv[1 + p_] : 0.0;
          sacc = fma(cf, xf[(bix >> (4 * p_)) & 15], sacc);
        }
        if (ipend2 >= 0) mydout[ipend2] = dpend2;
        ipend2 = (okme && (bix & (1 << 23))) ? brs : -1;
        dpend2 = fast_div(sacc, bpv[0]);
#pragma unroll
        for (int k = 0; k < ROWS_KM + 2; k++) bpv[k] = bpvn[k];
        bix = bixn; brs = brsn;
      }
#pragma unroll
      for (int k = 0; k < KB; k++) lr[k] = lrn[k];
      boff = nboff;
      s++;
    }
    if (ipend >= 0) mydout[ipend] = dpend;
    if constexpr (CNL_LEAN) if (ipend2 >= 0) mydout[ipend2] = dpend2;
  }
#ifdef CNL_STAMPS
  STAMP(6)
  if (lane == 0 && A.npos) for (int k = 0; k < 8; k++) A.npos[(blockIdx.x * WPB + wave) * 8 + k] = (long long)st_acc[k];
#endif
  if constexpr (STAGED) {
    if (FUSED && A.phase == 2) { task_done(dep_signal, lane, true); return; }  // (outputs: written by the deciding wavefront; rho_old and the slots: committed behind)
    if (A.phase == 1) task_done(dep_signal, lane, A.df_live != 0);  // the children of this task may read its solution components now
    // first attempt only: rho = 0, rho_old untouched; problems that failed are handed to the launch that follows
    if (valid && l == 0 && t_root && A.mode == MODE_NEWTON) { A.rho[prob] = 0.0; A.nfact[prob] = 1; A.success[prob] = success ? 1 : 0; }
  } else if (valid && l == 0 && A.mode == MODE_NEWTON) {
    A.rho[prob] = rho;
    A.rho_old[prob] = rho_old;
    A.nfact[prob] = nfact;
    A.success[prob] = success ? 1 : 0;
  }
}

// The per-function limit of dynamic LDS is process-wide state: it is always set to the SAME value (what the device allows), so that two
// host threads that drive handles with different LDS needs cannot lower it under each other's launches.
int lds_attr_cap(int need) {
  const int cap = (int)std::min<size_t>(max_lds_bytes(), (size_t)160 * 1024);
  return cap > need ? cap : need;
}

hipError_t launch_newton2(const DevPlan2& P, int wpb, size_t lds_bytes, const LaunchArgs& a, hipStream_t stream) {
  if (wpb < 1 || wpb > 4) return hipErrorInvalidConfiguration;
  const int waves = (a.batch + 3) / 4;
  const int grid = (waves + wpb - 1) / wpb;
  // per device and cheap: set on every launch (a process may drive several devices)
  // L rows one front late (LPend) where wavefronts share their SIMDs: from about one wavefront per SIMD on
  // (the lean instantiation does better with immediate stores: 1 058 k against 1 003 k systems/s)
  const bool lean = a.lean != 0 && a.mode != MODE_SOLVE;
  const bool lean_solve = a.lean != 0 && a.back_rows != 0 && a.mode == MODE_SOLVE;
  const bool late = waves >= 1024 && !lean && !lean_solve;
  const bool commit = a.only_if_status != 0 && a.lad != nullptr && a.mode == MODE_NEWTON;   // behind fused ladder launches
  auto kern = commit ? (lean ? newton2_kernel_t<false, false, true, false, true> : newton2_kernel_t<false, false, false, false, true>)
            : lean_solve ? newton2_kernel_t<false, false, true, true>
            : lean ? newton2_kernel_t<false, false, true> : (late ? newton2_kernel_t<false, true, false> : newton2_kernel_t<false, false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)lds_bytes));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpb), lds_bytes, stream, P, a);
  return hipGetLastError();
}

// try_to_factorize on a staged plan: the inertia rule on the counts the forward launches summed (src/solver_types.jl:90-97)
__global__ void __launch_bounds__(256) staged_decide_kernel(const int* __restrict__ gcnt, int nvar, int batch, int32_t* success, int64_t* npos,
                                                            int64_t* nzero) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= batch) return;
  const int tp = gcnt[2 * b], tz = gcnt[2 * b + 1];
  success[b] = (tp == nvar && tz == 0) ? 1 : 0;
  if (npos) npos[b] = tp;
  if (nzero) nzero[b] = tz;
}

hipError_t launch_newton2_staged(const DevPlan2& P, int wpb, size_t lds_bytes, LaunchArgs a, const int32_t* stage_ptr, int nstages, hipStream_t stream) {
  if (wpb < 1 || wpb > 4) return hipErrorInvalidConfiguration;
  a.nquads = (a.batch + 3) / 4;
  // the bidirectional chain of mid-size batches (two large tasks per group of problems) fills the SIMDs like the single stream
  // does: L rows one front late there, immediate stores for the bushy trees of small batches (see LPend)
  // (measured: 4096 problems in two tasks of 500 fronts 881 k -> 958 k systems/s; 256 problems in 32 tasks of 58 fronts
  //  462 k -> 436 k: short tasks lose — their L rows would mostly be flushed at the task's end, in front of the hand-over)
  const int ntask0 = stage_ptr[1] - stage_ptr[0];
  // (3 072 problems = 1.5 wavefronts per SIMD: 925 k systems/s with immediate stores against 819 k one front late;
  //  4 096: 938 k against 966 k — late only when the first stage fills both slots of every SIMD)
  const bool late = (long long)a.nquads * ntask0 >= 1920 && P.nsuper >= 128 * ntask0;
  const bool lean = a.lean != 0 && a.mode != MODE_SOLVE;
  const bool lean_solve = a.lean != 0 && a.back_rows != 0 && a.mode == MODE_SOLVE;
  auto kern = lean_solve ? newton2_kernel_t<true, false, true, true>
            : lean ? (late ? newton2_kernel_t<true, true, true> : newton2_kernel_t<true, false, true>)
                   : (late ? newton2_kernel_t<true, true, false> : newton2_kernel_t<true, false, false>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)lds_bytes));
  if (e != hipSuccess) return e;
  const bool ladder = a.mode == MODE_NEWTON && a.lad_mode != 0 && a.lad && a.lgcnt && a.ldep && a.status_call;
  // counters of the call: [gcnt | lgcnt | lad | ldep | status | dep] is ONE allocation of the handle, zeroed with one memset
  // (lad_zero_ints > 0); views of a handle (SubBatch) bring gcnt alone
  const bool one_block = a.lad_zero_ints > 0;
  if (one_block || a.mode != MODE_SOLVE) {
    e = hipMemsetAsync(a.gcnt, 0, one_block ? (size_t)a.lad_zero_ints * sizeof(int) : (size_t)a.batch * 2 * sizeof(int), stream);
    if (e != hipSuccess) return e;
  }
  // newton: forward + backward; factorize: forward only, then the decision; solve: forward substitution + backward
  const int npass = a.mode == MODE_FACTOR ? 1 : 2;
  // Stages [0, s_df) run one launch per stage; the stages from s_df on — the top of the tree, whose wavefronts together are
  // few (at most df_waves) — run as ONE launch per phase in which a task waits for its children resp. its parent on device
  // counters.  Every launch bumps the counters, so that the top launch finds the lower stages' tasks done; s_df = 0 is the pure
  // dataflow execution of the smallest batches, s_df = nstages one launch per stage.
  const int ntasks_all = stage_ptr[nstages];
  int s_df = nstages;
  if (a.dep) {
    while (s_df > 0 && (long long)(ntasks_all - stage_ptr[s_df - 1]) * a.nquads <= (long long)a.df_waves) s_df--;
    // Measured: a mixed execution (lower stages one launch each, the top of the tree as one launch) is NOT faster than a launch
    // per stage — cfg3 64 problems 0.255 against 0.241 ms, cfg4 256 problems 1.51 M against 1.66 M systems/s: the fences of the
    // top launch cost what the saved launch boundaries gain.  So: all of the tree in one launch per phase, or none of it.
    if (s_df != 0) s_df = nstages;
    if (!one_block) {  // (not reached: a.dep comes with the block)
      e = hipMemsetAsync(a.dep, 0, 2 * (size_t)ntasks_all * (size_t)a.nquads * sizeof(int), stream);
      if (e != hipSuccess) return e;
    }
  }
  a.ntasks_all = ntasks_all;
  // the in-kernel ladder: every task of a group of problems on a wavefront of its own, all of a launch resident at once
  auto kern_f = lean ? newton2_kernel_t<true, false, true, false, true> : newton2_kernel_t<true, false, false, false, true>;
  if (ladder) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern_f), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)lds_bytes));
    if (e != hipSuccess) return e;
  }
  auto launch_ladder = [&](int first) {
    const int slots = std::max(1, a.lad_capacity / ntasks_all);
    for (int q0 = 0; q0 < a.nquads; q0 += slots) {
      a.phase = 2; a.task0 = 0; a.ntasks = ntasks_all; a.df_live = 1; a.lad_first = first;
      a.quad0 = q0; a.lad_slots = std::min(slots, a.nquads - q0);
      const long long waves = (long long)ntasks_all * a.lad_slots;
      hipLaunchKernelGGL(kern_f, dim3((unsigned)((waves + wpb - 1) / wpb)), dim3(64 * wpb), lds_bytes, stream, P, a);
    }
  };
  if (ladder && a.lad_mode == 2 && s_df == 0 && (long long)ntasks_all * a.nquads <= (long long)a.lad_capacity) {  // smallest batches: first attempt, ladder and backward sweeps in ONE launch
    launch_ladder(1);
    return hipGetLastError();
  }
  auto launch_range = [&](int pass, int t0, int t1, int live = 0) {
    a.phase = pass; a.task0 = t0; a.ntasks = t1 - t0; a.df_live = live;
    const long long waves = (long long)a.ntasks * a.nquads;
    if (waves > 0) hipLaunchKernelGGL(kern, dim3((unsigned)((waves + wpb - 1) / wpb)), dim3(64 * wpb), lds_bytes, stream, P, a);
  };
  // forward: children first
  for (int q = 0; q < s_df; q++) launch_range(0, stage_ptr[q], stage_ptr[q + 1]);
  if (s_df < nstages) launch_range(0, stage_ptr[s_df], ntasks_all, 1);
  if (npass == 2) {  // backward: parents first
    if (s_df < nstages) launch_range(1, stage_ptr[s_df], ntasks_all, 1);
    for (int q = s_df - 1; q >= 0; q--) launch_range(1, stage_ptr[q], stage_ptr[q + 1]);
  }
  if (ladder) launch_ladder(0);  // the problems that failed the attempt climb the rho ladder here (the other groups' wavefronts exit at once)
#ifdef CNL_DBG_TRACE
  if (ladder) {
    (void)hipStreamSynchronize(stream);
    static int hb[1 << 16];
    int n = 0;
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(cnl_trace_n), sizeof(int));
    (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(cnl_trace_buf), sizeof(hb));
    fprintf(stderr, "[trace] ladder launch: ntasks_all %d nquads %d batch %d, %d records\n", ntasks_all, a.nquads, a.batch, n / 4);
    for (int i = 0; i + 3 < n && i < (1 << 16) - 4; i += 4) {
      const int tag = hb[i] & 255, wg = hb[i] >> 8;
      if (tag == 1) fprintf(stderr, "[trace] wg %d START task %d widx %d nchild %d nfr %d parent %d\n", wg, hb[i + 1] & 0xffff, hb[i + 1] >> 16, hb[i + 2] & 0xffff, hb[i + 2] >> 16, hb[i + 3]);
      else if (tag == 2) fprintf(stderr, "[trace] wg %d task %d END OF RUNG %d tpos %d tzer %d\n", wg, hb[i + 1] & 0xffff, hb[i + 1] >> 16, hb[i + 2], hb[i + 3]);
      else if (tag == 3) fprintf(stderr, "[trace] wg %d task %d rung %d fin %d epoch %d\n", wg, hb[i + 1] & 0xffff, hb[i + 1] >> 16, hb[i + 2], hb[i + 3]);
      else if (tag == 9) fprintf(stderr, "[trace] wg %d GIVE UP at site %d: wants %d has %d\n", wg, hb[i + 3], hb[i + 1], hb[i + 2]);
      else fprintf(stderr, "[trace] wg %d tag %d: %d %d %d\n", wg, tag, hb[i + 1], hb[i + 2], hb[i + 3]);
    }
    n = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(cnl_trace_n), &n, sizeof(int));
  }
#endif
  if (a.mode == MODE_FACTOR)
    hipLaunchKernelGGL(staged_decide_kernel, dim3((a.batch + 255) / 256), dim3(256), 0, stream, a.gcnt, P.nvar, a.batch, a.success, a.npos, a.nzero);
  return hipGetLastError();
}

}  // namespace cnl
