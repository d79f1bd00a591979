// outer_step.hip — the per-problem bookkeeping of the batched outer / inner loop, as four device kernels.
//
// cannoles.jl_amd/device_loop.py runs B problems of one model family through the reference's `solve!`
// (/root/reference/src/CaNNOLeS.jl:612-864) in lockstep, every quantity a [B, ...] array in HBM and every branch a mask.  Until
// round 4 the masks and the masked state updates were ~150 framework launches per global step (2.3 ms of launch chains for
// batches of any size); here they are four kernels working IN PLACE on the state the caller describes with `cnl_outer_state`:
//   cnl_outer_begin_dev        start of an outer iteration (:612-626), who needs a Newton system, the step's branch flags
//   cnl_outer_newton_done_dev  takes the Newton system's results (:633-652: d, rho_old, counters, `broken`), extrapolation eps (:659)
//   cnl_outer_trial_done_dev   acceptance test at the trial point and the state update (:733-763), end-of-inner-loop tests (:765-800)
//   cnl_outer_end_dev          statuses, outer-iteration counters (:800-857)
// The arithmetic of every test is the reference's, in its operation order; minimum / maximum propagate NaN as the framework's
// (and Julia's) do.  The model callbacks, the line search and the rare small-residual branch stay with the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/cannoles_hip.h"

// every product and sum rounded separately, as the scalar (and the framework's element-wise) code the decisions are compared with
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ double tmax(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }
__device__ __forceinline__ double tmin(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }

// sum over the workgroup (256 threads), result in every thread; fixed order
__device__ double block_sum(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ void __launch_bounds__(256) outer_begin_kernel(const cnl_outer_state S) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= S.B) return;
  const bool act = S.status[b] == 0;
  const bool so = act && S.phase0[b];
  long long inner = S.inner[b];
  if (so) {  // start of an outer iteration, src/CaNNOLeS.jl:612-626
    const double nd = S.normdual[b], np_ = S.normprimal[b];
    const double comb = nd + np_;
    S.combined[b] = comb;
    S.delta[b] = tmax(tmin(S.delta_dec * S.delta[b], comb), S.dmin);
    inner = 0;
    S.inner[b] = 0;
    S.combined_hat[b] = INFINITY;
    S.ndh[b] = nd; S.nph[b] = np_;
    S.phase0[b] = 0;
  }
  const bool need = act && inner != 1;   // the iteration right behind a rejected extrapolation reuses d (:627)
  S.act[b] = act; S.need[b] = need; S.brk[b] = 0;
  if (act) atomicOr(S.flags + 0, 1);
  if (need) atomicOr(S.flags + 1, 1);
  if (act && inner == 0) atomicOr(S.flags + 2, 1);
  if (act && inner > 0) atomicOr(S.flags + 3, 1);
}

// one workgroup per problem
__global__ void __launch_bounds__(256) outer_newton_done_kernel(const cnl_outer_state S, int did_newton) {
  __shared__ double sh[4];
  const long long b = blockIdx.x;
  const int t = threadIdx.x;
  bool act = S.act[b] != 0;
  if (did_newton) {
    const bool need = S.need[b] != 0;
    const double* dn = S.d_new + b * S.N;
    double bad = 0.0;
    for (long long k = t; k < S.N; k += 256) bad += isfinite(dn[k]) ? 0.0 : 1.0;
    bad = block_sum(bad, sh);
    if (need) {
      double* d = S.d + b * S.N;
      for (long long k = t; k < S.N; k += 256) d[k] = dn[k];
    }
    if (t == 0) {
      if (need) {
        S.rho_old[b] = S.ro_tmp[b];
        S.nfact[b] += S.nf_new[b];
        S.nlin[b] += 1;
      }
      // `broken` (:638-652): the inner loop is left at once, the end-of-iteration tests still run for the problem
      const bool brk = need && (S.rho_new[b] > S.rhomax || S.ok_new[b] == 0 || bad != 0.0 || S.fx[b] >= 1e60);
      S.brk[b] = brk;
      if (brk) { act = false; S.act[b] = 0; }
    }
    __syncthreads();
    act = S.act[b] != 0;
  }
  // multipliers of the line search's merit function, lam - c / delta (:1066), for every problem (used where lsm)
  if (S.lam_ls) {
    const double dl = S.delta[b];
    for (long long k = t; k < S.P; k += 256) S.lam_ls[b * S.P + k] = S.p > 0 ? S.lam[b * S.P + k] - S.cx[b * S.P + k] / dl : S.lam[b * S.P + k];
  }
  if (t == 0) {
    const long long inner = S.inner[b];
    const bool ext = act && inner == 0, lsm = act && inner > 0;
    S.ext[b] = ext; S.lsm[b] = lsm;
    if (ext) {  // :659
      const double e = S.epsk[b];
      S.epsk[b] = tmax(tmin(1e3 * S.delta[b], 99 * e / 100), 9 * e / 10);
    }
  }
}

// the extrapolation's trial point (cnl_trial_point_dev wrote xt_e, rt_e, lamt_e for every problem) goes to the problems of `ext`
__global__ void __launch_bounds__(256) outer_extrapolated_kernel(const cnl_outer_state S) {
  const long long b = blockIdx.x;
  if (!S.ext[b]) return;
  const int t = threadIdx.x;
  for (long long k = t; k < S.n; k += 256) S.xt[b * S.n + k] = S.xt_e[b * S.n + k];
  for (long long k = t; k < S.m; k += 256) S.rt[b * S.m + k] = S.rt_e[b * S.m + k];
  for (long long k = t; k < S.P; k += 256) S.lamt[b * S.P + k] = S.lamt_e[b * S.P + k];
}

__global__ void __launch_bounds__(256) outer_trial_done_kernel(const cnl_outer_state S) {
  __shared__ double sh[4];
  __shared__ int dec[4];
  const long long b = blockIdx.x;
  const int t = threadIdx.x;
  const bool act = S.act[b] != 0, brk = S.brk[b] != 0;
  const long long inner0 = S.inner[b];
  // f(xt) = |F(xt)|^2 / 2
  const double* Ft = S.Ft + b * S.m;
  double ss = 0.0;
  for (long long k = t; k < S.m; k += 256) ss += Ft[k] * Ft[k];
  ss = block_sum(ss, sh);
  if (t == 0) {
    double ndh = S.ndh[b], nph = S.nph[b], chat = S.combined_hat[b];
    if (act) { ndh = S.nrm_t[2 * b]; nph = S.nrm_t[2 * b + 1]; chat = ndh + nph; }   // optimality measures at the trial point, :722-732
    S.ndh[b] = ndh; S.nph[b] = nph; S.combined_hat[b] = chat;
    const double epsk = S.epsk[b];
    const bool good = chat <= 0.99 * S.combined[b] + epsk;                           // :733
    const bool acc_state = act && (inner0 > 0 || good);
    const bool acc_lam = act && good;
    if (acc_state) S.fx[b] = 0.5 * ss;
    const double delta = S.delta[b];
    double delta_next = delta;
    if (S.p > 0) {                                                                   // :758-763
      const bool dr = act && inner0 > 0 && (ndh <= 0.99 * S.normdual[b] + epsk / 2) && (nph > 0.99 * S.normprimal[b] + epsk / 2);
      if (dr) delta_next = tmax(delta / 10, S.dmin);
    }
    const long long inner = inner0 + (act ? 1 : 0);
    S.inner[b] = inner;
    const bool tired = inner > S.max_inner;
    const bool done_in = (act && (good || tired)) || brk;
    if (done_in) { S.normdual[b] = ndh; S.normprimal[b] = nph; }
    S.delta[b] = delta_next;
    const bool rej = act && !good;
    S.rej[b] = rej; S.done_in[b] = done_in; S.tired[b] = tired;
    dec[0] = acc_state; dec[1] = acc_lam; dec[2] = done_in;
    if (rej) atomicOr(S.flags + 4, 1);
  }
  __syncthreads();
  const bool acc_state = dec[0] != 0, acc_lam = dec[1] != 0, done_in = dec[2] != 0;
  if (acc_state) {
    for (long long k = t; k < S.n; k += 256) S.x[b * S.n + k] = S.xt[b * S.n + k];
    for (long long k = t; k < S.m; k += 256) { S.r[b * S.m + k] = S.rt[b * S.m + k]; S.Fx[b * S.m + k] = S.Ft[b * S.m + k]; }
    for (long long k = t; k < S.P; k += 256) S.cx[b * S.P + k] = S.ct[b * S.P + k];
    for (long long k = t; k < S.nnzjF; k += 256) S.Jv[b * S.nnzjF + k] = S.Jt[b * S.nnzjF + k];
    if (S.Jcv != S.Jct) for (long long k = t; k < S.nnzjc; k += 256) S.Jcv[b * S.nnzjc + k] = S.Jct[b * S.nnzjc + k];
  }
  if (acc_lam) for (long long k = t; k < S.P; k += 256) S.lam[b * S.P + k] = S.lamt[b * S.P + k];
  if (act) for (long long k = t; k < S.N; k += 256) S.rhs_cur[b * S.N + k] = S.rhs_t[b * S.N + k];
  __syncthreads();
  if (t == 0) {
    // end of the inner loop -> end of the outer iteration, :765-800
    double sl = 0.0, sc = 0.0;
    for (long long k = 0; k < S.p; k++) { sl += fabs(S.lam[b * S.P + k]); const double c = S.cx[b * S.P + k]; sc += c * c; }
    const double ds = S.p > 0 ? tmax(sl / (double)S.p, S.smax) / S.smax : 1.0;
    const bool first_order = tmax(S.normdual[b] / ds, S.normprimal[b]) <= S.epstol[b];
    const bool small_res = (2 * sqrt(S.fx[b]) <= S.epsF[b]) && (sqrt(sc) <= S.epsc[b]);
    S.small_res[b] = small_res;
    const bool chk = done_in && small_res && !first_order;
    S.chk[b] = chk;
    if (chk) atomicOr(S.flags + 5, 1);
  }
}

// ---- Armijo line search on the merit function phi(x) = |F|^2 / 2 - lam'c + eta |c|^2 / 2, src/CaNNOLeS.jl:1054-1112 ----------------
// The model callbacks (F, c at the trial points) are the caller's; these kernels do the rest of a round for every problem at once.
__device__ double merit(const cnl_outer_state& S, long long b, const double* F, const double* c, double eta, double* sh) {
  const int t = threadIdx.x;
  double sf = 0.0, slc = 0.0, scc = 0.0;
  for (long long k = t; k < S.m; k += 256) sf += F[b * S.m + k] * F[b * S.m + k];
  if (S.p > 0)
    for (long long k = t; k < S.p; k += 256) { const double cv = c[b * S.P + k]; slc += S.lam[b * S.P + k] * cv; scc += cv * cv; }
  sf = block_sum(sf, sh);
  slc = block_sum(slc, sh);
  scc = block_sum(scc, sh);
  double phi = 0.5 * sf;
  if (S.p > 0) { phi = phi - slc; phi = phi + eta * scc / 2; }
  return phi;
}

// Dphi = g'dx with g = Jx'F - Jc'(lam - c / delta) (the dual part cnl_residual_vectors_dev left in ls_g), eta, phi(x), alpha = 1,
// first trial point xl = x + dx
__global__ void __launch_bounds__(256) outer_ls_begin_kernel(const cnl_outer_state S) {
  __shared__ double sh[4];
  const long long b = blockIdx.x;
  const int t = threadIdx.x;
  const double* g = S.ls_g + b * S.N;
  const double* dx = S.d + b * S.N;
  double dp = 0.0;
  for (long long k = t; k < S.n; k += 256) dp += g[k] * dx[k];
  dp = block_sum(dp, sh);
  double eta = S.eta[b];
  if (S.p > 0 && S.lsm[b]) eta = 1.0 / S.delta[b];
  const double phix = merit(S, b, S.Fx, S.cx, eta, sh);
  if (t == 0) { S.Dphi[b] = dp; S.eta[b] = eta; S.phix[b] = phix; S.alpha[b] = 1.0; }
  for (long long k = t; k < S.n; k += 256) S.xl[b * S.n + k] = S.x[b * S.n + k] + dx[k];
}

// Armijo test at (Fl, cl) = (F(xl), c(xl)); first != 0: the first test (every lsm problem), else a backtracking round's
__global__ void __launch_bounds__(256) outer_ls_test_kernel(const cnl_outer_state S, int first) {
  __shared__ double sh[4];
  const long long b = blockIdx.x;
  const bool cand = first ? S.lsm[b] != 0 : S.bt[b] != 0;
  if (!cand) { if (first && threadIdx.x == 0) S.bt[b] = 0; return; }
  const double phil = merit(S, b, S.Fl, S.cl, S.eta[b], sh);
  if (threadIdx.x == 0) {
    const double alpha = S.alpha[b];
    bool bt = !(phil <= S.phix[b] + S.gammaA * alpha * S.Dphi[b]);
    if (!first) bt = bt && (alpha >= S.eps2);
    S.bt[b] = bt;
    if (bt) atomicOr(S.flags + 6, 1);
  }
}

// one backtracking step for the problems of bt: alpha / 4, xl = x + alpha dx (:1098-1105)
__global__ void __launch_bounds__(256) outer_ls_step_kernel(const cnl_outer_state S) {
  const long long b = blockIdx.x;
  if (!S.bt[b]) return;
  const int t = threadIdx.x;
  const double alpha = S.alpha[b] / 4;
  const double* dx = S.d + b * S.N;
  for (long long k = t; k < S.n; k += 256) S.xl[b * S.n + k] = S.x[b * S.n + k] + alpha * dx[k];
  if (t == 0) { S.alpha[b] = alpha; S.nbk[b] += 1; }
}

// the accepted point of the line search becomes the trial point of the problems of lsm
__global__ void __launch_bounds__(256) outer_ls_take_kernel(const cnl_outer_state S) {
  const long long b = blockIdx.x;
  if (!S.lsm[b]) return;
  const int t = threadIdx.x;
  for (long long k = t; k < S.n; k += 256) S.xt[b * S.n + k] = S.xl[b * S.n + k];
  for (long long k = t; k < S.m; k += 256) S.rt[b * S.m + k] = S.Fl[b * S.m + k];
  for (long long k = t; k < S.P; k += 256) S.lamt[b * S.P + k] = S.lam_ls[b * S.P + k];
}

__global__ void __launch_bounds__(256) outer_end_kernel(const cnl_outer_state S) {
  const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
  if (b >= S.B) return;
  const bool done_in = S.done_in[b] != 0;
  if (!done_in) return;
  double sl = 0.0;
  for (long long k = 0; k < S.p; k++) sl += fabs(S.lam[b * S.P + k]);
  const double ds = S.p > 0 ? tmax(sl / (double)S.p, S.smax) / S.smax : 1.0;
  const bool first_order = tmax(S.normdual[b] / ds, S.normprimal[b]) <= S.epstol[b];
  S.it[b] += 1;
  // tired = inner > max_inner: the reference hands that to get_status as `stalled` (src/CaNNOLeS.jl:846) -> 5; 4 (max_eval) is an
  // evaluation-count limit, which this loop does not have
  S.status[b] = first_order ? 1 : (S.small_res[b] ? 2 : (S.brk[b] ? 3 : (S.tired[b] ? 5 : 0)));
  S.phase0[b] = 1;
}

// every array of the state the kernels dereference must be there (the header promises CNL_ERR_ARG for a missing array, not an
// asynchronous fault).  The members are named one by one (walking the struct as an array of pointers would read a scalar
// member added later as a pointer).  The multiplier / constraint arrays (lam, cx, ct, lamt, lamt_e, cl, lam_ls) have rows of
// P = max(p, 1) entries and are walked over P: they are required also when p == 0 (include/cannoles_hip.h says so); only the
// constraint-Jacobian value arrays Jcv / Jct may be NULL when nnzjc == 0.  The line-search entries also need theirs (check_ls).
int check(const cnl_outer_state* st) {
  if (!st || st->B <= 0) return CNL_ERR_ARG;
#define CNL_NEED(m) if (!st->m) return CNL_ERR_ARG;
  CNL_NEED(status) CNL_NEED(it) CNL_NEED(flags) CNL_NEED(nf_new) CNL_NEED(ok_new)
  CNL_NEED(inner) CNL_NEED(nfact) CNL_NEED(nlin)
  CNL_NEED(phase0) CNL_NEED(act) CNL_NEED(need) CNL_NEED(brk) CNL_NEED(ext) CNL_NEED(lsm) CNL_NEED(rej) CNL_NEED(chk) CNL_NEED(done_in) CNL_NEED(tired) CNL_NEED(small_res)
  CNL_NEED(normdual) CNL_NEED(normprimal) CNL_NEED(combined) CNL_NEED(combined_hat) CNL_NEED(delta) CNL_NEED(ndh) CNL_NEED(nph) CNL_NEED(fx)
  CNL_NEED(epsk) CNL_NEED(epstol) CNL_NEED(epsF) CNL_NEED(epsc) CNL_NEED(rho_old)
  CNL_NEED(d) CNL_NEED(d_new) CNL_NEED(ro_tmp) CNL_NEED(rho_new)
  CNL_NEED(x) CNL_NEED(r) CNL_NEED(Fx) CNL_NEED(Jv) CNL_NEED(rhs_cur)
  CNL_NEED(xt) CNL_NEED(rt) CNL_NEED(Ft) CNL_NEED(Jt) CNL_NEED(rhs_t) CNL_NEED(nrm_t)
  CNL_NEED(xt_e) CNL_NEED(rt_e)
  CNL_NEED(cx) CNL_NEED(lam) CNL_NEED(ct) CNL_NEED(lamt) CNL_NEED(lamt_e)
  if (st->nnzjc > 0) { CNL_NEED(Jcv) CNL_NEED(Jct) }
  return 0;
}
int check_ls(const cnl_outer_state* st) {
  if (check(st)) return CNL_ERR_ARG;
  CNL_NEED(ls_g) CNL_NEED(xl) CNL_NEED(Fl) CNL_NEED(cl) CNL_NEED(lam_ls) CNL_NEED(alpha) CNL_NEED(Dphi) CNL_NEED(phix) CNL_NEED(eta) CNL_NEED(nbk) CNL_NEED(bt)
#undef CNL_NEED
  return 0;
}
int done() { return hipGetLastError() == hipSuccess ? CNL_OK : CNL_ERR_HIP; }

}  // namespace

extern "C" {

int cnl_outer_begin_dev(const cnl_outer_state* st, void* stream) {
  if (check(st)) return CNL_ERR_ARG;
  if (hipMemsetAsync(st->flags, 0, 8 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return CNL_ERR_HIP;
  hipLaunchKernelGGL(outer_begin_kernel, dim3((unsigned)((st->B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_newton_done_dev(const cnl_outer_state* st, int did_newton, void* stream) {
  if (check(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_newton_done_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st, did_newton);
  return done();
}

int cnl_outer_extrapolated_dev(const cnl_outer_state* st, void* stream) {
  if (check(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_extrapolated_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_trial_done_dev(const cnl_outer_state* st, void* stream) {
  if (check(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_trial_done_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_ls_begin_dev(const cnl_outer_state* st, void* stream) {
  if (check_ls(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_ls_begin_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_ls_test_dev(const cnl_outer_state* st, int first, void* stream) {
  if (check_ls(st)) return CNL_ERR_ARG;
  if (hipMemsetAsync(st->flags + 6, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return CNL_ERR_HIP;
  hipLaunchKernelGGL(outer_ls_test_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st, first);
  return done();
}

int cnl_outer_ls_step_dev(const cnl_outer_state* st, void* stream) {
  if (check_ls(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_ls_step_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_ls_take_dev(const cnl_outer_state* st, void* stream) {
  if (check_ls(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_ls_take_kernel, dim3((unsigned)st->B), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

int cnl_outer_end_dev(const cnl_outer_state* st, void* stream) {
  if (check(st)) return CNL_ERR_ARG;
  hipLaunchKernelGGL(outer_end_kernel, dim3((unsigned)((st->B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *st);
  return done();
}

}  // extern "C"
