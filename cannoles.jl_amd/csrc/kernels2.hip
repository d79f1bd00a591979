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
//    lower triangle (register a = row a), the pivot row is broadcast with
//    ds_bpermute / v_readlane, and the rank-1 update is one FMA per row.  No
//    LDS traffic, no barrier and no index decode inside the pivot loop.
//    Fronts of order <= 16 run four problems at once, <= 32 two, <= 64 one.
//  * the plan is a self-describing record stream (analysis.cpp) read with
//    coalesced loads into an LDS double buffer; the record of front s+2 and the
//    values of front s+1 are prefetched into registers while front s is being
//    eliminated, so no global-memory latency sits on the per-front critical path.
//  * update matrices wait for their parent on a per-problem LDS stack whose
//    offsets were fixed on the host; the few large ones (and the staging of
//    fronts of order > 32) use a per-problem global scratch instead.
//  * L rows are stored to HBM straight from registers at their pivot step.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace cnl {

namespace {

constexpr int RN = 2;    // record prefetch: RN x dwordx4 per lane = RN*256 words
constexpr int PVN = 8;   // value prefetch: PVN doubles per lane = PVN*16 entries per problem

__device__ __forceinline__ int tri2(int i) { return (i * (i + 1)) >> 1; }

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

// broadcast lane (group base + a) of a TE-lane group
template <int TE>
__device__ __forceinline__ double bcast(double v, int a, int grp4) {
  if (false) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), a);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), a);
    return __hiloint2double(hi, lo);
  } else {
    int lo = __builtin_amdgcn_ds_bpermute(grp4 + a * 4, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(grp4 + a * 4, __double2hiint(v));
    return __hiloint2double(hi, lo);
  }
}

// value of the lane whose byte address (4 * lane) is `addr4`
__device__ __forceinline__ double bcast_addr(double v, int addr4) {
  int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
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
//     r<k-1> = fma(-l[top-k], w, r<k>),   l = w / d broadcast with ds_bpermute.
// The rows are spelled as individual scalars through the X-macro lists of elim_lists.inc.
#include "elim_lists.inc"

#define CNL_DECL(k) double r##k;
// lanes b > row read past the row: harmless garbage in the unused upper triangle (staging is padded)
#define CNL_LOAD(k) { const int a_ = top - k > 0 ? top - k : 0; r##k = Fs[tri2(a_) + b]; }  /* rows below 0: unused copies of row 0 */
#define CNL_STEP(km1, k) r##km1 = fma(-bcast_addr(lv, base + (TE_ - 1 - k) * 4), w, r##k);
#define CNL_CHK(k) if (k >= i) goto rows_done;  /* rows i-k >= 1 only (row 0 is the unused rhs-row diagonal) */
#define CNL_USTG(k) { const int a_ = nupd - k; if (a_ >= 0) { if (b <= a_ && valid) Ug[tri2(a_) + b] = r##k; } }
#define CNL_USTL(k) { const int a_ = nupd - k; if (a_ >= 0) { if (b <= a_) Ul[tri2(a_) + b] = r##k; } }

#define CNL_DEFINE_ELIM(NAME, INL, TEV, GFS, ALL, STEPS)                                                               \
  __device__ INL void NAME(const DevPlan2& P, const Ctx2& c, int lane, int prob0, int pass, int f,        \
                                       int nupd, long long lptr, int uoff, int fsoff, bool uglob, double* pbase0,     \
                                       int* cnt, double eig_tol) {                                                     \
    constexpr int TE_ = TEV;                                                                                           \
    constexpr int PPW = 64 / TE_;                                                                                      \
    const int gp = pass * PPW + (TE_ == 64 ? 0 : lane / TE_);                                                          \
    const int b = lane % TE_;                                                                                          \
    const int grp4 = (lane - b) * 4;                                                                                   \
    const int prob = prob0 + gp;                                                                                       \
    const bool valid = prob < c.batch;                                                                                 \
    const long long pclamp = valid ? prob : prob0;                                                                     \
    double* pb = pbase0 + gp * P.prob_doubles;                                                                         \
    const double* Fs = GFS ? (c.gs + pclamp * P.gs_doubles + fsoff) : (pb + P.u2_peak);                                \
    double* Lp = c.L + pclamp * P.lsize + lptr;                                                                        \
    const int tu = tri2(1 + nupd);                                                                                     \
    const int top = f - 1;                                                                                             \
    ALL(CNL_DECL)                                                                                                      \
    ALL(CNL_LOAD)                                                                                                      \
    int npos = 0, nzer = 0;                                                                                            \
    int base = grp4 + (top - (TE_ - 1)) * 4;                                                                           \
    for (int i = top; i > nupd; i--) {                                                                                 \
      const double w = r0;                                                                                             \
      const double dpiv = bcast_addr(w, base + (TE_ - 1) * 4);                                                         \
      npos += dpiv > eig_tol;                                                                                          \
      nzer += fabs(dpiv) <= eig_tol;                                                                                   \
      const double lv = w / dpiv;                                                                                      \
      if (valid && b <= i) Lp[tri2(i) - tu + b] = (b == i) ? dpiv : lv;                                                \
      STEPS(CNL_STEP, CNL_CHK)                                                                                         \
    rows_done:                                                                                                         \
      base -= 4;                                                                                                       \
    }                                                                                                                  \
    if (b == 0) { cnt[gp * 2] += npos; cnt[gp * 2 + 1] += nzer; }                                                      \
    if (uglob) {                                                                                                       \
      double* Ug = c.gs + pclamp * P.gs_doubles + uoff;                                                                \
      ALL(CNL_USTG)                                                                                                    \
    } else {                                                                                                           \
      double* Ul = pb + uoff;                                                                                          \
      ALL(CNL_USTL)                                                                                                    \
    }                                                                                                                  \
  }

// the rare large classes are real calls so that their register needs do not leak into the hot path
CNL_DEFINE_ELIM(eliminate16, __forceinline__, 16, false, CNL_ALL16, CNL_STEPS16)
CNL_DEFINE_ELIM(eliminate32, __attribute__((noinline)), 32, false, CNL_ALL32, CNL_STEPS32)
CNL_DEFINE_ELIM(eliminate32g, __attribute__((noinline)), 32, true, CNL_ALL32, CNL_STEPS32)
CNL_DEFINE_ELIM(eliminate64, __attribute__((noinline)), 64, true, CNL_ALL64, CNL_STEPS64)

// backward substitution of one front for the problems of a pass
template <int TE>
__device__ __attribute__((noinline)) void back_front_call(const DevPlan2& P, const Ctx2& c, int lane, int prob0, int pass, const int* rec, int f,
                                                     int nupd, int npiv, long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag);

template <int TE>
__device__ __forceinline__ void back_front(const DevPlan2& P, const Ctx2& c, int lane, int prob0, int pass, const int* rec, int f,
                                           int nupd, int npiv, long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag) {
  constexpr int PPW = 64 / TE;
  const int gp = pass * PPW + (TE == 64 ? 0 : lane / TE);
  const int b = lane % TE;
  const int prob = prob0 + gp;
  const bool valid = prob < c.batch && okflag[gp] != 0;
  const long long pclamp = prob < c.batch ? prob : prob0;
  double* xs = pbase0 + gp * P.prob_doubles;
  const double* Lp = c.L + pclamp * P.lsize + lptr;
  const int tu = tri2(1 + nupd);
  double xb = 0.0;
  if (pxoff >= 0 && b >= 1 && b <= nupd) xb = xs[pxoff + rec[B_HDR + b]];
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
          if (valid) c.dout[pclamp * P.dstride + rec[B_HDR + 1 + nupd + k0 + k]] = -xi;
        }
      }
    }
  }
  if (b >= 1 && b < f) xs[xoff + b] = xb;
}

template <int TE>
__device__ __attribute__((noinline)) void back_front_call(const DevPlan2& P, const Ctx2& c, int lane, int prob0, int pass, const int* rec, int f,
                                                     int nupd, int npiv, long long lptr, int xoff, int pxoff, double* pbase0, const int* okflag) {
  back_front<TE>(P, c, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, okflag);
}

}  // namespace

// ==========================================================================================
__global__ void __launch_bounds__(256, 2) newton2_kernel(const DevPlan2 P, const LaunchArgs A) {
  const int WPB = blockDim.x >> 6;
  extern __shared__ double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane >> 4, l = lane & 15;
  const int prob0 = (blockIdx.x * WPB + wave) * 4;
  if (prob0 >= A.batch) return;
  const int prob = prob0 + g;
  const bool valid = prob < A.batch;
  const long long pclamp = valid ? prob : prob0;

  const int wave_doubles = P.reccap + 4 * P.prob_doubles + 8;
  double* wbase = smem + wave * wave_doubles;
  int* recbuf = reinterpret_cast<int*>(wbase);  // two buffers of reccap words
  double* pbase0 = wbase + P.reccap;
  int* cnt = reinterpret_cast<int*>(pbase0 + 4 * P.prob_doubles);
  double* myU = pbase0 + g * P.prob_doubles;
  double* myFs = myU + P.u2_peak;

  Ctx2 c;
  c.vals = A.vals; c.rhs = A.rhs; c.L = A.L; c.gs = A.scratch; c.dout = A.d; c.batch = A.batch;
  const double* myvals = A.vals + pclamp * P.vstride;
  const double* myrhs = (A.mode == MODE_FACTOR || !A.rhs) ? nullptr : A.rhs + pclamp * P.rstride;
  double* mygs = A.scratch + pclamp * P.gs_doubles;
  const double eig_tol = A.params[0];
  const int xpos = A.extra_pos ? A.extra_pos[pclamp] : 0, xzer = A.extra_zer ? A.extra_zer[pclamp] : 0;

  // per-problem ladder state, replicated over the 16 lanes of the group
  double rho = 0.0, wrote = 0.0;
  double rho_old = (A.mode == MODE_NEWTON) ? A.rho_old[pclamp] : 0.0;
  int nfact = 0;
  bool done = !valid, success = false, ovr = false;
  const double kdec = A.params[2], kinc = A.params[3], klarge = A.params[4], rho0 = A.params[5], rhomax = A.params[6],
               rhomin = A.params[7];

  while (true) {
    // ---------------- forward pass over the record stream ----------------
    if (l == 0) { cnt[g * 2] = xpos; cnt[g * 2 + 1] = xzer; }
    const int4* rstream = reinterpret_cast<const int4*>(P.rec);
    // prologue: record 0 -> buffer 0 (synchronous), then prefetch record 1 and the values of front 0
    int roff = 0;  // word offset of the current record
    {
      int len0 = P.rec[R_RECLEN];
      if (len0 > P.reccap) len0 = P.reccap;
      for (int w4 = lane; w4 * 4 < len0; w4 += 64) reinterpret_cast<int4*>(recbuf)[w4] = rstream[w4];
      wsync();
    }
    int4 R[RN];
    double pv[PVN];
    int nxt_off = roff + P.rec[R_RECLEN];  // offset of record 1
    {
#pragma unroll
      for (int k = 0; k < RN; k++) R[k] = rstream[(nxt_off >> 2) + lane + 64 * k];  // stream is padded: over-read is safe
      const int nasm0 = rfl(recbuf[R_NASM]), aoff0 = rfl(recbuf[R_ASM_OFF]);
#pragma unroll
      for (int j = 0; j < PVN; j++) {
        double v0 = 0.0;
        const int e = j * 16 + l;
        if (e < nasm0) {
          const int src = P.rec[aoff0 + e];
          if (src >= P.nnz) v0 = myrhs ? myrhs[src - P.nnz] : 0.0;
          else if (src >= 0) v0 = (ovr && src >= P.rho_begin) ? rho : myvals[src];
        }
        pv[j] = v0;
      }
    }
    int cur_next = 0;
    for (int s = 0; s < P.nsuper; s++) {
      const int* rec = recbuf + (s & 1) * P.reccap;
      const int npiv = rfl(rec[R_NPIV]), nupd = rfl(rec[R_NUPD]), nasm = rfl(rec[R_NASM]);
      const int nchild = rfl(rec[R_NCHILD]), uoff = rfl(rec[R_UOFF]), flags = rfl(rec[R_FLAGS]), fsoff = rfl(rec[R_FSOFF]);
      const int cls = rfl(rec[R_CLS]), aoff = rfl(rec[R_ASM_OFF]), coff = rfl(rec[R_CHILD_OFF]);
      const long long lptr = (long long)rfl(rec[R_LPTR_LO]) | ((long long)rfl(rec[R_LPTR_HI]) << 31);
      const int f = 1 + nupd + npiv;
      const int tf = tri2(f);
      const bool gfs = flags & RF_FS_GLOBAL;
      // (1) zero the staging triangle, (2) assemble (prefetched values first)
      if (!gfs) {
        for (int t = l; t < tf; t += 16) myFs[t] = 0.0;
        wsync();
#pragma unroll
        for (int j = 0; j < PVN; j++) {
          const int e = j * 16 + l;
          if (j * 16 < nasm) {
            if (e < nasm) {
              const int pos = rec[aoff + nasm + e];
              __hip_atomic_fetch_add(&myFs[pos], pv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
          }
        }
        for (int e = PVN * 16 + l; e < nasm; e += 16) {
          const int src = rec[aoff + e], pos = rec[aoff + nasm + e];
          double v = 0.0;
          if (src >= P.nnz) v = myrhs ? myrhs[src - P.nnz] : 0.0;
          else if (src >= 0) v = (ovr && src >= P.rho_begin) ? rho : myvals[src];
          __hip_atomic_fetch_add(&myFs[pos], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
      } else {
        double* Fg = mygs + fsoff;
        const int* grec = P.rec + roff;  // lists of a globally staged front are read from the stream itself
        for (int t = l; t < tf; t += 16) Fg[t] = 0.0;
        gsync();
        for (int e0 = 0; e0 < nasm; e0 += 16) {
          const int e = e0 + l;
          const int src = grec[aoff + e], pos = grec[aoff + nasm + e];
          double v = 0.0;
          if (e0 < PVN * 16) {
            // prefetched slot j = e0/16 (static index needed): fall through the unrolled select below
#pragma unroll
            for (int j = 0; j < PVN; j++) if (j * 16 == e0) v = pv[j];
          } else {
            if (src >= P.nnz) v = myrhs ? myrhs[src - P.nnz] : 0.0;
            else if (src >= 0) v = (ovr && src >= P.rho_begin) ? rho : myvals[src];
          }
          if (src != -1 && valid) Fg[pos] += v;
          gsync();
        }
      }
      // (3) next record into the other buffer; prefetch the one after and the next front's values
      int* nrec = recbuf + ((s + 1) & 1) * P.reccap;
      if (s + 1 < P.nsuper) {
        const int nlen = __builtin_amdgcn_readlane(R[0].z, 0);  // word R_RECLEN of the prefetched header
        const int clen = nlen < P.reccap ? nlen : P.reccap;     // globally staged fronts keep only their head in LDS
#pragma unroll
        for (int k = 0; k < RN; k++)
          if ((lane + 64 * k) * 4 < clen) reinterpret_cast<int4*>(nrec)[lane + 64 * k] = R[k];
        wsync();
        for (int w4 = RN * 64 + lane; w4 * 4 < clen; w4 += 64) reinterpret_cast<int4*>(nrec)[w4] = rstream[(nxt_off >> 2) + w4];
        wsync();
        const int nn_off = nxt_off + nlen;
        if (s + 2 < P.nsuper) {
#pragma unroll
          for (int k = 0; k < RN; k++) R[k] = rstream[(nn_off >> 2) + lane + 64 * k];
        }
        const int nasm1 = rfl(nrec[R_NASM]), aoff1 = rfl(nrec[R_ASM_OFF]);
        const bool ngfs = rfl(nrec[R_FLAGS]) & RF_FS_GLOBAL;
        const int* gnrec = P.rec + nxt_off;
#pragma unroll
        for (int j = 0; j < PVN; j++) {
          double v0 = 0.0;
          const int e = j * 16 + l;
          if (j * 16 < nasm1) {
            if (e < nasm1) {
              const int src = ngfs ? gnrec[aoff1 + e] : nrec[aoff1 + e];
              if (src >= P.nnz) v0 = myrhs ? myrhs[src - P.nnz] : 0.0;
              else if (src >= 0) v0 = (ovr && src >= P.rho_begin) ? rho : myvals[src];
            }
          }
          pv[j] = v0;
        }
        cur_next = nxt_off;
        nxt_off = nn_off;
      }
      // (4) extend-add the children's update matrices
      {
        int co = coff;
        for (int ci = 0; ci < nchild; ci++) {
          int cu, tuc, cfl;
          if (gfs) {
            const int* crec = P.rec + roff + co;
            cu = rfl(crec[C_UOFF]); tuc = rfl(crec[C_TUC]); cfl = rfl(crec[C_FLAGS]);
          } else {
            cu = rfl(rec[co + C_UOFF]); tuc = rfl(rec[co + C_TUC]); cfl = rfl(rec[co + C_FLAGS]);
          }
          if (!gfs && !cfl) {
            const int* dest = rec + co + C_HDR;
            const double* U = myU + cu;
            for (int t = l; t < tuc; t += 16)
              __hip_atomic_fetch_add(&myFs[dest[t]], U[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          } else if (!gfs) {
            const int* dest = rec + co + C_HDR;
            const double* Ug = mygs + cu;
            for (int t = l; t < tuc; t += 16)
              __hip_atomic_fetch_add(&myFs[dest[t]], Ug[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          } else {
            const int* dest = P.rec + roff + co + C_HDR;
            const double* Ul = myU + cu;
            const double* Ug = mygs + cu;
            double* Fg = mygs + fsoff;
            for (int t = l; t < tuc; t += 16) {
              const double u = cfl ? Ug[t] : Ul[t];
              if (valid) Fg[dest[t]] += u;
            }
            gsync();
          }
          co += C_HDR + ((tuc + 3) & ~3);
        }
      }
      if (gfs) gsync(); else wsync();
      // (5) eliminate in registers, store L rows and the update matrix
      const bool uglob = flags & RF_U_GLOBAL;
      if (cls == 16) {
        eliminate16(P, c, lane, prob0, 0, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
      } else if (cls == 32) {
        for (int pass = 0; pass < 2; pass++) {
          if (prob0 + pass * 2 >= A.batch) break;
          if (gfs) eliminate32g(P, c, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
          else eliminate32(P, c, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
        }
      } else {
        for (int pass = 0; pass < 4; pass++) {
          if (prob0 + pass >= A.batch) break;
          eliminate64(P, c, lane, prob0, pass, f, nupd, lptr, uoff, fsoff, uglob, pbase0, cnt, eig_tol);
        }
      }
      if (flags & (RF_U_GLOBAL | RF_FS_GLOBAL)) gsync(); else wsync();
      roff = cur_next;
    }
    // ---------------- inertia test and rho ladder (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047) ----
    wsync();
    const bool ok = cnt[g * 2] == P.nvar && cnt[g * 2 + 1] == 0;
    if (A.mode == MODE_FACTOR) {
      if (valid && l == 0) {
        A.success[prob] = ok ? 1 : 0;
        if (A.npos) A.npos[prob] = cnt[g * 2];
        if (A.nzero) A.nzero[prob] = cnt[g * 2 + 1];
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
  if (l == 0) cnt[8 + g] = (success && valid) ? 1 : 0;
  gsync();
  // ---------------- backward pass (d = -K^-1 rhs), only where the factorisation succeeded -----------
  // (problems that failed still walk the stream with the wave; their output is not stored)
  if (__any(success)) {
    Ctx2 cb = c;
    cb.batch = A.batch;
    const int4* bstream = reinterpret_cast<const int4*>(P.brec);
    int boff = 0;
    {
      const int len0 = P.brec[B_RECLEN];
      for (int w4 = lane; w4 * 4 < len0; w4 += 64) reinterpret_cast<int4*>(recbuf)[w4] = bstream[w4];
      wsync();
    }
    int nxt = rfl(recbuf[B_RECLEN]);
    int4 Rb = bstream[(nxt >> 2) + lane];  // padded stream
    for (int s = 0; s < P.nsuper; s++) {
      const int* rec = recbuf + (s & 1) * P.reccap;
      const int npiv = rfl(rec[B_NPIV]), nupd = rfl(rec[B_NUPD]), xoff = rfl(rec[B_XOFF]), pxoff = rfl(rec[B_PXOFF]);
      const int cls = rfl(rec[B_CLS]);
      const long long lptr = (long long)rfl(rec[B_LPTR_LO]) | ((long long)rfl(rec[B_LPTR_HI]) << 31);
      const int f = 1 + nupd + npiv;
      // next record
      int* nrec = recbuf + ((s + 1) & 1) * P.reccap;
      if (s + 1 < P.nsuper) {
        const int nlen = __builtin_amdgcn_readlane(Rb.z, 0);  // word B_RECLEN
        if (lane * 4 < nlen) reinterpret_cast<int4*>(nrec)[lane] = Rb;
        wsync();
        for (int w4 = 64 + lane; w4 * 4 < nlen; w4 += 64) reinterpret_cast<int4*>(nrec)[w4] = bstream[(nxt >> 2) + w4];
        const int nn = nxt + nlen;
        if (s + 2 < P.nsuper) Rb = bstream[(nn >> 2) + lane];
        nxt = nn;
      }
      if (cls == 16) {
        back_front<16>(P, cb, lane, prob0, 0, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, cnt + 8);
      } else if (cls == 32) {
        for (int pass = 0; pass < 2; pass++) {
          if (prob0 + pass * 2 >= A.batch) break;
          back_front_call<32>(P, cb, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, cnt + 8);
        }
      } else {
        for (int pass = 0; pass < 4; pass++) {
          if (prob0 + pass >= A.batch) break;
          back_front_call<64>(P, cb, lane, prob0, pass, rec, f, nupd, npiv, lptr, xoff, pxoff, pbase0, cnt + 8);
        }
      }
      wsync();
    }
  }
  if (valid && l == 0) {
    A.rho[prob] = rho;
    A.rho_old[prob] = rho_old;
    A.nfact[prob] = nfact;
    A.success[prob] = success ? 1 : 0;
  }
}

hipError_t launch_newton2(const DevPlan2& P, int wpb, size_t lds_bytes, const LaunchArgs& a, hipStream_t stream) {
  if (wpb < 1 || wpb > 4) return hipErrorInvalidConfiguration;
  const int waves = (a.batch + 3) / 4;
  const int grid = (waves + wpb - 1) / wpb;
  static size_t attr_set = 0;
  if (lds_bytes > attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(newton2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = lds_bytes;
  }
  hipLaunchKernelGGL(newton2_kernel, dim3(grid), dim3(64 * wpb), lds_bytes, stream, P, a);
  return hipGetLastError();
}

}  // namespace cnl
