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

struct Win {
  double S[15];   // band slots, packed lower triangle
  double X[5];    // border row
  double S55;
  double c[5];    // right-hand side
  double c5;
};

#define LDSD(off) (*reinterpret_cast<const double*>(myb + (off)))
#define LDSW(off) (*reinterpret_cast<double*>(myb + (off)))

// one forward step with enter slot PH, pivot slot (PH + 1) % 5
template <int PH>
__device__ __forceinline__ void fstep(Win& W, cptr st, char* myb, const double* __restrict__ gvals, const double* __restrict__ grhs,
                                      cptr borders, long long pv, long long pr, bool has_rhs, double rho, bool ovr, double tol,
                                      int& npos, int& nzer) {
  constexpr int es = PH, ps = (PH + 1) % BAND_NB;
  const int fl = st[BS_FLAGS];
  // ---- enter ----
  {
    double dg = (LDSD(st[BS_DG0]) + LDSD(st[BS_DG1])) + LDSD(st[BS_DG2]);
    double rv = LDSD(st[BS_RHO]);
    if (!(fl & (1 << 16))) rv = ovr ? rho : rv;
    W.S[sidx(es, es)] = dg + rv;
#pragma unroll
    for (int k = 1; k <= BAND_HW; k++) {
      const int s = (es - k + BAND_NB) % BAND_NB;
      W.S[sidx(es, s)] = LDSD(st[BS_OD + 2 * (k - 1)]) + LDSD(st[BS_OD + 2 * (k - 1) + 1]);
    }
    W.X[es] = LDSD(st[BS_BC0]) + LDSD(st[BS_BC1]);
    W.c[es] = LDSD(st[BS_RX]);
  }
  // ---- residual rows completed by the entering variable: products -J_a J_b / d_r, counted in the inertia ----
  const int nrows = (fl >> 8) & 255;
  for (int i = 0; i < nrows; i++) {
    cptr rb = st + BAND_SW + BAND_RW * i;
    const double dr = LDSD(rb[BR_DI]);
    npos += dr > tol;
    nzer += fabs(dr) <= tol;
    const double r = rrcp(dr);
    const double w = rdiv(-1.0, dr, r);
    double J[BAND_NB];
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) J[s] = LDSD(rb[BR_J0 + s]);
    const double tr = LDSD(rb[BR_RR]) * w;
#pragma unroll
    for (int a = 0; a < BAND_NB; a++) {
      const double ta = J[a] * w;
#pragma unroll
      for (int b = 0; b <= a; b++) W.S[sidx(a, b)] = fma(ta, J[b], W.S[sidx(a, b)]);
      W.c[a] = fma(tr, J[a], W.c[a]);
    }
  }
  // ---- border pivot ----
  if (fl & BF_PIVOT_B) {
    cptr bt = borders + BAND_BW * st[BS_BORDER];
    W.S55 += gvals[pv + bt[BB_DSRC]];
    W.c5 += has_rhs ? grhs[pr + bt[BB_RHS]] : 0.0;
    const double d = W.S55;
    npos += d > tol;
    nzer += fabs(d) <= tol;
    const double r = rrcp(d);
    double l[BAND_NB];
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) l[s] = rdiv(W.X[s], d, r);
    const double z = rdiv(W.c5, d, r);
#pragma unroll
    for (int a = 0; a < BAND_NB; a++) {
#pragma unroll
      for (int b = 0; b <= a; b++) W.S[sidx(a, b)] = fma(W.X[a], -l[b], W.S[sidx(a, b)]);
      W.c[a] = fma(W.X[a], -z, W.c[a]);
    }
    char* lo = myb + st[BS_LB];
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) reinterpret_cast<double*>(lo)[s] = l[s];
    reinterpret_cast<double*>(lo)[BAND_NB] = z;
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) W.X[s] = 0.0;
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
    for (int s = 0; s < BAND_NB; s++) { w[s] = W.S[sidx(s, ps)]; l[s] = rdiv(w[s], d, r); }
    const double w5 = W.X[ps], l5 = rdiv(w5, d, r), z = rdiv(W.c[ps], d, r);
#pragma unroll
    for (int a = 0; a < BAND_NB; a++) {
      if (a == ps) continue;
#pragma unroll
      for (int b = 0; b <= a; b++) {
        if (b == ps) continue;
        W.S[sidx(a, b)] = fma(w[a], -l[b], W.S[sidx(a, b)]);
      }
      W.X[a] = fma(w5, -l[a], W.X[a]);
      W.c[a] = fma(w[a], -z, W.c[a]);
    }
    W.S55 = fma(w5, -l5, W.S55);
    W.c5 = fma(w5, -z, W.c5);
    double* lo = reinterpret_cast<double*>(myb + st[BS_LX]);
    int k = 0;
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) {
      if (s == ps) continue;
      lo[k++] = l[s];
    }
    lo[4] = l5;
    lo[5] = z;
  }
}

// one backward step (pivot slot (PH + 1) % 5): x of the band pivot, then of the border pivot, then the residual components
template <int PH>
__device__ __forceinline__ void bstep(double (&xs)[6], cptr st, char* myb, cptr borders, double* __restrict__ gd,
                                      long long pd, bool okme) {
  constexpr int ps = (PH + 1) % BAND_NB;
  const int fl = st[BS_FLAGS];
  if (fl & BF_PIVOT_X) {
    const double* lo = reinterpret_cast<const double*>(myb + st[BS_LX]);
    double x = lo[5];
    int k = 0;
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) {
      if (s == ps) continue;
      x = fma(-lo[k++], xs[s], x);
    }
    x = fma(-lo[4], xs[5], x);
    xs[ps] = x;
    LDSW(st[BS_DX]) = -x;
  }
  if (fl & BF_PIVOT_B) {
    const double* lo = reinterpret_cast<const double*>(myb + st[BS_LB]);
    double x = lo[BAND_NB];
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) x = fma(-lo[s], xs[s], x);
    xs[5] = x;
    if (okme) gd[pd + borders[BAND_BW * st[BS_BORDER] + BB_DOUT]] = -x;
  }
  const int nrows = (fl >> 8) & 255;
  for (int i = 0; i < nrows; i++) {
    cptr rb = st + BAND_SW + BAND_RW * i;
    double acc = -LDSD(rb[BR_RR]);
#pragma unroll
    for (int s = 0; s < BAND_NB; s++) acc = fma(LDSD(rb[BR_J0 + s]), xs[s], acc);
    const double dr = LDSD(rb[BR_DI]);
    LDSW(rb[BR_DR]) = rdiv(acc, dr, rrcp(dr));
  }
  if (fl & BF_ENTER_B) xs[5] = 0.0;
}

}  // namespace

// control block of a workgroup in LDS: per problem [rho | flags], then one word "all done"
template <int NL>
__global__ void __launch_bounds__(128, (NL <= 8 ? 2 : 1)) band_newton_kernel(const BandDev P, const LaunchArgs Ain) {
  constexpr int NI = NL / 8;
  extern __shared__ double lds[];
  LaunchArgs A = Ain;
  A.vals = as_global(Ain.vals); A.rhs = as_global(Ain.rhs); A.d = as_global(Ain.d); A.L = as_global(Ain.L);
  A.rho_old = as_global(Ain.rho_old); A.rho = as_global(Ain.rho); A.nfact = as_global(Ain.nfact); A.success = as_global(Ain.success);
  A.npos = as_global(Ain.npos); A.nzero = as_global(Ain.nzero);
  const int lane = threadIdx.x & 63;
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lq = lane >> 3, le = lane & 7;
  const int prob0 = blockIdx.x * NL;
  const int batch = A.batch;
  const bool has_rhs = A.mode == MODE_NEWTON && A.rhs != nullptr;
  char* wblk = reinterpret_cast<char*>(lds + (size_t)part * NL * LANE_D);
  double* ctrl = lds + (size_t)P.nparts * NL * LANE_D;   // [NL] rho, [NL] flags (1 ovr, 2 ok), [1] all done
  cptr fops = as_const(P.fops[part]);
  cptr bops = as_const(P.bops[part]);
  cptr epochs = as_const(P.epochs[part]);
  cptr borders = as_const(P.borders[part]);
  const int nepochs = P.nepochs[part];
  // mover offsets: elements from the first problem of the workgroup (32-bit: NL problems span < 4 GB)
  const double* vbase = A.vals + (long long)prob0 * P.nnz;
  const double* rbase = has_rhs ? A.rhs + (long long)prob0 * P.N : A.vals;
  double* lbase_g = A.L + (long long)prob0 * P.lsize + P.loff[part];
  double* dbase = A.d ? A.d + (long long)prob0 * P.N : nullptr;
  // per-lane byte offsets of the mover (wave-uniform base pointer + 32-bit offset: one global_load / global_store each)
  unsigned voffb[NI], roffb[NI], loffb[NI], doffb[NI], ldsb[NI];
  bool movok[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    int pl = i * 8 + lq;
    movok[i] = prob0 + pl < batch;
    if (!movok[i]) pl = batch - 1 - prob0;
    voffb[i] = ((unsigned)pl * (unsigned)P.nnz + (unsigned)le) << 3;
    roffb[i] = ((unsigned)pl * (unsigned)P.N + (unsigned)le) << 3;
    loffb[i] = ((unsigned)pl * (unsigned)P.lsize + (unsigned)le) << 3;
    doffb[i] = roffb[i];
    ldsb[i] = ((unsigned)(i * 8 + lq) * (unsigned)LANE_D + (unsigned)le) << 3;
  }
  // compute lanes
  const bool clane = lane < NL;
  const int cprob = prob0 + (clane ? lane : 0);
  const bool valid = clane && cprob < batch;
  const int cpl = valid ? cprob - prob0 : batch - 1 - prob0;   // problem whose block this lane computes on
  char* myb = wblk + (size_t)(clane ? lane : 0) * LANE_D * 8;
  const long long pv = (long long)(prob0 + cpl) * P.nnz, pr = (long long)(prob0 + cpl) * P.N;
  // every block's zero cell
  for (int t = lane; t < NL; t += 64) *reinterpret_cast<double*>(wblk + ((size_t)t * LANE_D + BAND_ZERO_OFF) * 8) = 0.0;

  const double tol = A.params[0], kdec = A.params[2], kinc = A.params[3], klarge = A.params[4], rho0 = A.params[5], rhomax = A.params[6],
               rhomin = A.params[7];
  double rho = 0.0, wrote = 0.0;
  double rho_old = (A.mode == MODE_NEWTON && valid) ? A.rho_old[cprob] : 0.0;
  int nfact = 0;
  bool done = !valid, success = false, ovr = false;

  double stg[NPC][NI];   // operand pieces in flight
  unsigned pmask = 0;    // pieces of the epoch whose operands are in flight
  auto issue = [&](cptr E, int ofs) {
    pmask = 0;
#pragma unroll
    for (int k = 0; k < NPC; k++) {
      const int pc = E[ofs + k];
      if (pc >= 0) {
        pmask |= 1u << k;
        const int arr = pc >> 28;
        const long long base = (long long)(pc & ((1 << 28) - 1)) << 3;
        if (arr == 0) {
          const char* pb = reinterpret_cast<const char*>(vbase) + base;
#pragma unroll
          for (int i = 0; i < NI; i++) stg[k][i] = *reinterpret_cast<const double*>(pb + voffb[i]);
        } else if (arr == 1) {
          const char* pb = reinterpret_cast<const char*>(rbase) + base;
#pragma unroll
          for (int i = 0; i < NI; i++) stg[k][i] = has_rhs ? *reinterpret_cast<const double*>(pb + roffb[i]) : 0.0;
        } else {
          const char* pb = reinterpret_cast<const char*>(lbase_g) + base;
#pragma unroll
          for (int i = 0; i < NI; i++) stg[k][i] = *reinterpret_cast<const double*>(pb + loffb[i]);
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < NPC; k++)
      if (pmask & (1u << k)) {
#pragma unroll
        for (int i = 0; i < NI; i++) *reinterpret_cast<double*>(wblk + ldsb[i] + (BAND_IN_OFF + 8 * k) * 8) = stg[k][i];
      }
  };

  Win W;
  int npos = 0, nzer = 0;
  double lj[6], zj[4];   // junction factor (first wavefront)
  while (true) {
    // ================= forward: assembly, elimination, forward substitution =================
#pragma unroll
    for (int q = 0; q < 15; q++) W.S[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 5; q++) { W.X[q] = 0.0; W.c[q] = 0.0; }
    W.S55 = 0.0; W.c5 = 0.0;
    npos = 0; nzer = 0;
    issue(epochs, BE_FP);
    int o = 0, u = 0;
    for (int e = 0; e < nepochs; e++) {
      cptr E = epochs + e * BAND_EW;
      commit();
      if (e + 1 < nepochs) issue(E + BAND_EW, BE_FP);
      const int nst = E[BE_NSTEP];
      if (clane) {
        for (int t = 0; t < nst; t++) {
          cptr st = fops + o;
          const int nrows = (st[BS_FLAGS] >> 8) & 255;
          switch (u % BAND_NB) {
            case 0: fstep<0>(W, st, myb, A.vals, A.rhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer); break;
            case 1: fstep<1>(W, st, myb, A.vals, A.rhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer); break;
            case 2: fstep<2>(W, st, myb, A.vals, A.rhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer); break;
            case 3: fstep<3>(W, st, myb, A.vals, A.rhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer); break;
            default: fstep<4>(W, st, myb, A.vals, A.rhs, borders, pv, pr, has_rhs, rho, ovr, tol, npos, nzer); break;
          }
          o += BAND_SW + BAND_RW * nrows;
          u++;
        }
      } else {
        for (int t = 0; t < nst; t++) { o += BAND_SW + BAND_RW * ((fops[o + BS_FLAGS] >> 8) & 255); u++; }
      }
      // factor records of the epoch: 64-byte pieces from the out ring
      const int lb = E[BE_LBASE], lc = E[BE_LCNT];
      char* lout = reinterpret_cast<char*>(lbase_g) + ((long long)lb << 3);
#pragma unroll
      for (int cpc = 0; cpc < BAND_LOUT_MAX / 8; cpc++) {
        if (cpc * 8 < lc) {
#pragma unroll
          for (int i = 0; i < NI; i++) {
            const double x = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_LOUT_OFF + 8 * cpc) * 8);
            if (movok[i] && cpc * 8 + le < lc) *reinterpret_cast<double*>(lout + loffb[i] + 64 * cpc) = x;
          }
        }
      }
    }
    // ================= junction + inertia rule + rho ladder (src/solver_types.jl:90-97, src/CaNNOLeS.jl:1023-1047) ==========
    int tpos = npos, tzer = nzer;
    if (P.nparts == 2) {
      // windows to LDS (slot order): the junction reads both with run-time slot numbers
      if (clane) {
        double* ex = reinterpret_cast<double*>(myb + BAND_DX_OFF * 8);
#pragma unroll
        for (int q = 0; q < 15; q++) ex[q] = W.S[q];
#pragma unroll
        for (int q = 0; q < 5; q++) ex[15 + q] = W.c[q];
        ex[20] = (double)npos; ex[21] = (double)nzer;
      }
      __syncthreads();
      if (part == 0 && clane) {
        const double* exL = reinterpret_cast<const double*>(myb + BAND_DX_OFF * 8);
        const double* exR = reinterpret_cast<const double*>(myb + (size_t)NL * LANE_D * 8 + BAND_DX_OFF * 8);
        const int tL = P.m0 % BAND_NB, tR = (P.n - 1 - P.m0) % BAND_NB;
        double SJ[10], cJ[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int aL = (tL + i) % BAND_NB, aR = (tR - i + BAND_NB) % BAND_NB;
#pragma unroll
          for (int j = 0; j <= i; j++) {
            const int bL = (tL + j) % BAND_NB, bR = (tR - j + BAND_NB) % BAND_NB;
            const int iL = aL >= bL ? aL * (aL + 1) / 2 + bL : bL * (bL + 1) / 2 + aL;
            const int iR = aR >= bR ? aR * (aR + 1) / 2 + bR : bR * (bR + 1) / 2 + aR;
            SJ[i * (i + 1) / 2 + j] = exL[iL] + exR[iR];
          }
          cJ[i] = exL[15 + aL] + exR[15 + aR];
        }
        tpos += (int)exR[20]; tzer += (int)exR[21];
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
      const bool ok = tpos == P.nvar && tzer == 0;
      if (A.mode == MODE_FACTOR) {
        if (valid) {
          A.success[cprob] = ok ? 1 : 0;
          if (A.npos) A.npos[cprob] = tpos;
          if (A.nzero) A.nzero[cprob] = tzer;
        }
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
  if (A.mode == MODE_FACTOR) return;
  // ================= backward: d = -K^-1 rhs where the factorisation succeeded =================
  {
    double xs[6];
#pragma unroll
    for (int q = 0; q < 6; q++) xs[q] = 0.0;
    if (P.nparts == 2) {
      if (part == 0 && clane) {
        double xj[4];
        xj[3] = zj[3];
        xj[2] = fma(-lj[sidx(2, 2)], xj[3], zj[2]);
        xj[1] = fma(-lj[sidx(2, 1)], xj[3], fma(-lj[sidx(1, 1)], xj[2], zj[1]));
        xj[0] = fma(-lj[sidx(2, 0)], xj[3], fma(-lj[sidx(1, 0)], xj[2], fma(-lj[sidx(0, 0)], xj[1], zj[0])));
        double* ex = reinterpret_cast<double*>(myb + BAND_DX_OFF * 8);
        double* exR = reinterpret_cast<double*>(myb + (size_t)NL * LANE_D * 8 + BAND_DX_OFF * 8);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ex[i] = xj[i]; exR[i] = xj[i];
          if (valid && success) A.d[(long long)cprob * P.N + P.m0 + i] = -xj[i];
        }
      }
      __syncthreads();
      if (clane) {
        const double* ex = reinterpret_cast<const double*>(myb + BAND_DX_OFF * 8);
        const int t0 = part == 0 ? P.m0 % BAND_NB : (P.n - 1 - P.m0) % BAND_NB;
#pragma unroll
        for (int s = 0; s < BAND_NB; s++) {
          // junction variable i sits in slot (t0 + i) % 5 (first part) / (t0 - i) % 5 (second part)
          const int i = part == 0 ? (s - t0 + BAND_NB) % BAND_NB : (t0 - s + BAND_NB) % BAND_NB;
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
    const long long pd = (long long)(prob0 + cpl) * P.N;
    issue(epochs + (nepochs - 1) * BAND_EW, BE_BP);
    int o = 0, u = P.nsteps[part] - 1;
    for (int e = nepochs - 1; e >= 0; e--) {
      cptr E = epochs + e * BAND_EW;
      commit();
      if (e > 0) issue(E - BAND_EW, BE_BP);
      const int nst = E[BE_NSTEP];
      if (clane) {
        for (int t = 0; t < nst; t++) {
          cptr st = bops + o;
          const int nrows = (st[BS_FLAGS] >> 8) & 255;
          switch (u % BAND_NB) {
            case 0: bstep<0>(xs, st, myb, borders, A.d, pd, okme); break;
            case 1: bstep<1>(xs, st, myb, borders, A.d, pd, okme); break;
            case 2: bstep<2>(xs, st, myb, borders, A.d, pd, okme); break;
            case 3: bstep<3>(xs, st, myb, borders, A.d, pd, okme); break;
            default: bstep<4>(xs, st, myb, borders, A.d, pd, okme); break;
          }
          o += BAND_SW + BAND_RW * nrows;
          u--;
        }
      } else {
        for (int t = 0; t < nst; t++) { o += BAND_SW + BAND_RW * ((bops[o + BS_FLAGS] >> 8) & 255); u--; }
      }
      // solution components of the epoch
      const int xlo = E[BE_DXLO], xc = E[BE_DXCNT], rlo = E[BE_DRLO], rc = E[BE_DRCNT];
      char* dxo = reinterpret_cast<char*>(dbase) + ((long long)xlo << 3);
      char* dro = reinterpret_cast<char*>(dbase) + ((long long)rlo << 3);
#pragma unroll
      for (int cpc = 0; cpc < 2; cpc++) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
          if (cpc * 8 < xc) {
            const double x = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_DX_OFF + 8 * cpc) * 8);
            if (movst[i] && cpc * 8 + le < xc) *reinterpret_cast<double*>(dxo + doffb[i] + 64 * cpc) = x;
          }
          if (cpc * 8 < rc) {
            const double x = *reinterpret_cast<const double*>(wblk + ldsb[i] + (BAND_DR_OFF + 8 * cpc) * 8);
            if (movst[i] && cpc * 8 + le < rc) *reinterpret_cast<double*>(dro + doffb[i] + 64 * cpc) = x;
          }
        }
      }
    }
  }
  // ================= outputs of newton_system! =================
  if (part == 0) {
    if (nfact > 1 && rho <= rhomax) rho_old = rho;
    if (valid) {
      A.rho[cprob] = rho;
      A.rho_old[cprob] = rho_old;
      A.nfact[cprob] = nfact;
      A.success[cprob] = success ? 1 : 0;
    }
    // rho slots of the problems that climbed (src/CaNNOLeS.jl:1031,1038,1044-1046): the last nvar entries of vals
    for (int q = 0; q < NL; q++) {
      const int nf = __builtin_amdgcn_readlane(nfact, q);
      const int vq = __builtin_amdgcn_readlane((int)valid, q);
      if (nf > 1 && vq) {
        const double wq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wrote), q), __builtin_amdgcn_readlane(__double2loint(wrote), q));
        double* vt = A.vals + (long long)(prob0 + q) * P.nnz + (P.nnz - P.nvar);
        for (int i = lane; i < P.nvar; i += 64) vt[i] = wq;
      }
    }
  }
}

size_t band_lds_bytes(int nparts, int nl) { return ((size_t)nparts * nl * LANE_D + 2 * nl + 8) * sizeof(double); }

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
