// dense.hip — dense-residual-block backend (see dense.h).  Panel kernels are written here; the two GEMM-shaped steps
// (J' W J and the trailing updates of the blocked LDL^T) go to rocBLAS dgemm, i.e. to the fp64 matrix cores.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>

#include "dense.h"

namespace cnl {

namespace {

constexpr int NB = 16;  // panel width of the blocked factorisation (measured on MI355X, n = 1000, fused panel step: 8 -> 2.14 ms, 16 -> 1.79 ms, 32 -> 2.01 ms per system)

#define DCHK(x)                                                                                 \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return 4; }   \
  } while (0)
#define BCHK(x)                                                                                 \
  do {                                                                                          \
    rocblas_status s_ = (x);                                                                    \
    if (s_ != rocblas_status_success) { err = std::string(#x) + ": rocBLAS status " + std::to_string((int)s_); return 4; } \
  } while (0)

// Jd(i, j) = vals[jslot], JW = diag(w) Jd with w_i = -1 / d_r(i); inertia of the residual pivots
__global__ void __launch_bounds__(256) gather_kernel(const double* __restrict__ vals, const int* __restrict__ jslot,
                                                     const int* __restrict__ dslot, int m, int n, double* __restrict__ Jd,
                                                     double* __restrict__ JW, double* __restrict__ w) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)m * n) return;
  const int i = (int)(t % m);
  const double wi = -1.0 / vals[dslot[i]];
  const double v = vals[jslot[t]];
  Jd[t] = v;
  JW[t] = wi * v;
  if (t < m) w[i] = wi;
}

__global__ void __launch_bounds__(256) rinertia_kernel(const double* __restrict__ vals, const int* __restrict__ dslot, int m,
                                                       double eig_tol, int* cnt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const double dv = vals[dslot[i]];
  if (dv > eig_tol) atomicAdd(&cnt[0], 1);
  if (fabs(dv) <= eig_tol) atomicAdd(&cnt[1], 1);
}

__global__ void __launch_bounds__(256) hscatter_kernel(const double* __restrict__ vals, const int* __restrict__ hslot,
                                                       const int* __restrict__ hpos, int nh, double* S0) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < nh) atomicAdd(&S0[hpos[e]], vals[hslot[e]]);
}

// S = S0 + diag(rho): rho from the slots of vals (first attempt, as given) or one value for all (ladder retries)
__global__ void __launch_bounds__(256) shift_kernel(const double* __restrict__ S0, double* __restrict__ S, int n, int nv,
                                                    const double* __restrict__ rho_slots, double rho, int use_slots) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)n * n) return;
  const int i = (int)(t % n), j = (int)(t / n);
  double v = S0[t];
  if (i == j && i < nv) v += use_slots ? rho_slots[i] : rho;  // rho I on the variables only
  S[t] = v;
}

// One panel step in ONE launch: every workgroup factorises the nb x nb diagonal block at (k0, k0) in LDS (unblocked
// LDL^T, redundantly: 16 x 16 is cheaper than a launch), then serves 256 rows below it, thread per row:
//   (l_ij d_j) = a_ij - sum_q (l_iq d_q) L11(j, q)  -> W21 (for the trailing GEMM),   l_ij -> S.
// Workgroup 0 writes the factorised diagonal block back and counts its pivots.
__global__ void __launch_bounds__(256) panel_l21_kernel(double* S, int n, int k0, int nb, double* __restrict__ W21, int n2,
                                                        double eig_tol, int* cnt) {
  __shared__ double a[NB][NB + 1];
  __shared__ double dd[NB];
  const int t = threadIdx.x;
  for (int q = t; q < nb * nb; q += 256) { const int i = q % nb, j = q / nb; a[i][j] = S[(size_t)(k0 + i) + (size_t)n * (k0 + j)]; }
  __syncthreads();
  int np = 0, nz = 0;
  for (int j = 0; j < nb; j++) {
    const double dj = a[j][j];
    if (t == 0) { np += dj > eig_tol; nz += fabs(dj) <= eig_tol; }
    __syncthreads();
    if (t > j && t < nb) a[t][j] = a[t][j] / dj;  // l_tj
    __syncthreads();
    // a(i, k) -= l_ij d_j l_kj for j < k <= i
    for (int q = t; q < nb * nb; q += 256) {
      const int i = q % nb, k = q / nb;
      if (k > j && i >= k) a[i][k] -= a[i][j] * dj * a[k][j];
    }
    __syncthreads();
  }
  if (t < nb) dd[t] = a[t][t];
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int q = t; q < nb * nb; q += 256) { const int i = q % nb, j = q / nb; if (i >= j) S[(size_t)(k0 + i) + (size_t)n * (k0 + j)] = a[i][j]; }
    if (t == 0) { if (np) atomicAdd(&cnt[0], np); if (nz) atomicAdd(&cnt[1], nz); }
  }
  const int r = blockIdx.x * 256 + t;  // row of the trailing part
  if (r >= n2) return;
  double* row = S + (size_t)(k0 + nb + r) + (size_t)n * k0;  // entry (row, k0 + j) at row[j * n]
  double wv[NB];
#pragma unroll
  for (int j = 0; j < NB; j++) {
    if (j < nb) {
      double v = row[(size_t)j * n];
#pragma unroll
      for (int q = 0; q < j; q++) v -= wv[q] * a[j][q];
      wv[j] = v;
      W21[(size_t)r + (size_t)n2 * j] = v;
      row[(size_t)j * n] = v / dd[j];
    }
  }
}

__global__ void __launch_bounds__(256) scale_by_diag_kernel(double* __restrict__ y, const double* __restrict__ S, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = y[i] / S[(size_t)i + (size_t)n * i];
}

__global__ void __launch_bounds__(256) mul_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

__global__ void __launch_bounds__(256) out_kernel(const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ w,
                                                  double* __restrict__ d, int n, int m, int p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = -x[i];
  else if (i < n + m) d[i] = w[i - n] * u[i - n];  // d_r = -(rhs_r - J x) / d_r
  else if (i < n + m + p) d[i] = -x[n + (i - n - m)];  // multipliers: the last p unknowns of the dense system
}

__global__ void __launch_bounds__(256) fill_kernel(double* p, double v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// general condensed system: S0(pos[s]) = slot s (positions are unique)
__global__ void __launch_bounds__(256) slot_scatter_kernel(const double* __restrict__ slots, const int* __restrict__ pos, int nslots,
                                                           double* __restrict__ S0) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < nslots) S0[pos[e]] = slots[e];
}

__global__ void __launch_bounds__(256) neg_kernel(const double* __restrict__ y, double* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = -y[i];
}

inline int blocks(long long n) { return (int)((n + 255) / 256); }

}  // namespace

struct DenseState {
  rocblas_handle blas = nullptr;
  int64_t batch = 1;
  int *jslot = nullptr, *dslot = nullptr, *hslot = nullptr, *hpos = nullptr, *cnt = nullptr;
  double *Jd = nullptr, *JW = nullptr, *w = nullptr, *S0 = nullptr, *W21 = nullptr, *y = nullptr, *u = nullptr, *tmp = nullptr;
  double* S = nullptr;  // [batch][n * n] factors (unit lower L below the diagonal, D on it)
  const double* last_vals = nullptr;
  std::vector<void*> allocs;
};

bool detect_dense(DensePlan& D, int64_t N, int64_t nnz, const int64_t* rows1, const int64_t* cols1, int64_t nvar, int64_t nequ,
                  int64_t ncon) {
  D.active = false;
  if (nequ <= 0 || nvar < 32 || N != nvar + nequ + ncon) return false;
  if ((long long)nvar * nequ > (1ll << 28) || nnz > (1ll << 30) || (nvar + ncon) > 30000) return false;
  D.n = (int32_t)nvar; D.m = (int32_t)nequ; D.p = (int32_t)ncon; D.nnz = (int32_t)nnz;
  const int64_t ns = nvar + ncon;  // order of the dense system
  D.jslot.assign((size_t)nvar * nequ, -1);
  D.dslot.assign(nequ, -1);
  D.hslot.clear(); D.hpos.clear();
  const int64_t rho_begin = nnz - nvar;
  for (int64_t e = 0; e < nnz; e++) {
    const int64_t i = rows1[e] - 1, j = cols1[e] - 1;
    if (i < nvar) {  // H_F / H_c entry or rho slot (lower triangle, column j <= row i)
      if (e >= rho_begin) continue;  // rho slots are applied by shift_kernel
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

template <class T>
static int dalloc_(DenseState* st, T** p, size_t count, std::string& err) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return 4; }
  st->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

int dense_create(DenseState** out, const DensePlan& D, int64_t batch, std::string& err) {
  DenseState* st = new DenseState();
  st->batch = batch;
  *out = st;
  const size_t n = D.n, m = D.m, ns = (size_t)D.n + D.p;
  int rc;
  if ((rc = dalloc_(st, &st->jslot, n * m, err))) return rc;
  if ((rc = dalloc_(st, &st->dslot, m, err))) return rc;
  if ((rc = dalloc_(st, &st->hslot, D.hslot.size(), err))) return rc;
  if ((rc = dalloc_(st, &st->hpos, D.hpos.size(), err))) return rc;
  if ((rc = dalloc_(st, &st->cnt, 4, err))) return rc;
  if ((rc = dalloc_(st, &st->Jd, n * m, err))) return rc;
  if ((rc = dalloc_(st, &st->JW, n * m, err))) return rc;
  if ((rc = dalloc_(st, &st->w, m, err))) return rc;
  if ((rc = dalloc_(st, &st->S0, ns * ns, err))) return rc;
  if ((rc = dalloc_(st, &st->W21, ns * NB, err))) return rc;
  if ((rc = dalloc_(st, &st->y, ns, err))) return rc;
  if ((rc = dalloc_(st, &st->u, m, err))) return rc;
  if ((rc = dalloc_(st, &st->tmp, m, err))) return rc;
  if ((rc = dalloc_(st, &st->S, (size_t)batch * ns * ns, err))) return rc;
  DCHK(hipMemcpy(st->jslot, D.jslot.data(), n * m * sizeof(int), hipMemcpyHostToDevice));
  DCHK(hipMemcpy(st->dslot, D.dslot.data(), m * sizeof(int), hipMemcpyHostToDevice));
  if (!D.hslot.empty()) {
    DCHK(hipMemcpy(st->hslot, D.hslot.data(), D.hslot.size() * sizeof(int), hipMemcpyHostToDevice));
    DCHK(hipMemcpy(st->hpos, D.hpos.data(), D.hpos.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  BCHK(rocblas_create_handle(&st->blas));
  return 0;
}

void dense_destroy(DenseState* st) {
  if (!st) return;
  if (st->blas) rocblas_destroy_handle(st->blas);
  for (void* p : st->allocs) (void)hipFree(p);
  delete st;
}

namespace {

// S_b = L D L^T of (S0 + shift); counts of the dense pivots go to st->cnt[0..1] (added to what is there)
int factor_ns(DenseState* st, int n, int nv, double* S, const double* rho_slots, double rho, int use_slots, double eig_tol,
              hipStream_t stream, std::string& err) {
  hipLaunchKernelGGL(shift_kernel, dim3(blocks((long long)n * n)), dim3(256), 0, stream, st->S0, S, n, nv, rho_slots, rho, use_slots);
  const double one = 1.0, mone = -1.0;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int nb = std::min(NB, n - k0), n2 = n - k0 - nb;
    hipLaunchKernelGGL(panel_l21_kernel, dim3(std::max(1, blocks(n2))), dim3(256), 0, stream, S, n, k0, nb, st->W21, n2, eig_tol, st->cnt);
    if (n2 <= 0) break;
    // S22 -= L21 (L21 D)^T
    BCHK(rocblas_dgemm(st->blas, rocblas_operation_none, rocblas_operation_transpose, n2, n2, nb, &mone,
                       S + (size_t)(k0 + nb) + (size_t)n * k0, n, st->W21, n2, &one, S + (size_t)(k0 + nb) + (size_t)n * (k0 + nb), n));
  }
  DCHK(hipGetLastError());
  return 0;
}

int factor(DenseState* st, const DensePlan& D, double* S, const double* rho_slots, double rho, int use_slots, double eig_tol,
           hipStream_t stream, std::string& err) {
  return factor_ns(st, D.n + D.p, D.n, S, rho_slots, rho, use_slots, eig_tol, stream, err);
}

int build(DenseState* st, const DensePlan& D, const double* vals, double eig_tol, hipStream_t stream, std::string& err) {
  const int n = D.n, m = D.m;
  DCHK(hipMemsetAsync(st->cnt, 0, 4 * sizeof(int), stream));
  hipLaunchKernelGGL(gather_kernel, dim3(blocks((long long)m * n)), dim3(256), 0, stream, vals, st->jslot, st->dslot, m, n, st->Jd, st->JW, st->w);
  hipLaunchKernelGGL(rinertia_kernel, dim3(blocks(m)), dim3(256), 0, stream, vals, st->dslot, m, eig_tol, st->cnt + 2);
  const int ns = n + D.p;
  DCHK(hipMemsetAsync(st->S0, 0, (size_t)ns * ns * sizeof(double), stream));
  if (!D.hslot.empty())
    hipLaunchKernelGGL(hscatter_kernel, dim3(blocks((long long)D.hslot.size())), dim3(256), 0, stream, vals, st->hslot, st->hpos, (int)D.hslot.size(), st->S0);
  const double one = 1.0;
  // top-left block: S0(1:n, 1:n) += J' (W J)
  BCHK(rocblas_dgemm(st->blas, rocblas_operation_transpose, rocblas_operation_none, n, n, m, &one, st->Jd, m, st->JW, m, &one, st->S0, ns));
  return 0;
}

// d = -K^-1 rhs with the factor in S (Jd, w of the same problem must be in place)
int solve(DenseState* st, const DensePlan& D, const double* S, const double* rhs, double* d, hipStream_t stream, std::string& err) {
  const int n = D.n, m = D.m, p = D.p, ns = D.n + D.p;
  const double one = 1.0, mone = -1.0;
  // y = [rhs_x + J' (w .* rhs_r) ; rhs_c]      (rhs_x - J' (rhs_r / d_r) on the variables)
  hipLaunchKernelGGL(mul_kernel, dim3(blocks(m)), dim3(256), 0, stream, st->w, rhs + n, st->tmp, m);
  DCHK(hipMemcpyAsync(st->y, rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  if (p > 0) DCHK(hipMemcpyAsync(st->y + n, rhs + n + m, (size_t)p * sizeof(double), hipMemcpyDeviceToDevice, stream));
  BCHK(rocblas_dgemv(st->blas, rocblas_operation_transpose, m, n, &one, st->Jd, m, st->tmp, 1, &one, st->y, 1));
  BCHK(rocblas_dtrsv(st->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, ns, S, ns, st->y, 1));
  hipLaunchKernelGGL(scale_by_diag_kernel, dim3(blocks(ns)), dim3(256), 0, stream, st->y, S, ns);
  BCHK(rocblas_dtrsv(st->blas, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_unit, ns, S, ns, st->y, 1));
  // u = rhs_r - J x
  DCHK(hipMemcpyAsync(st->u, rhs + n, (size_t)m * sizeof(double), hipMemcpyDeviceToDevice, stream));
  BCHK(rocblas_dgemv(st->blas, rocblas_operation_none, m, n, &mone, st->Jd, m, st->y, 1, &one, st->u, 1));
  hipLaunchKernelGGL(out_kernel, dim3(blocks(n + m + p)), dim3(256), 0, stream, st->y, st->u, st->w, d, n, m, p);
  DCHK(hipGetLastError());
  return 0;
}

}  // namespace

int dense_run(DenseState* st, const DensePlan& D, int mode, double* vals, const double* rhs, double* d, double* rho_old_d,
              double* rho_d, int32_t* nfact_d, int32_t* success_d, int64_t* npos_d, int64_t* nzero_d, const double params[9],
              hipStream_t stream, std::string& err) {
  const int n = D.n, m = D.m;
  const size_t N = (size_t)n + m + D.p, ns = (size_t)n + D.p;
  BCHK(rocblas_set_stream(st->blas, stream));
  const double eig_tol = params[0], kdec = params[2], kinc = params[3], klarge = params[4], rho0 = params[5], rhomax = params[6],
               rhomin = params[7];
  if (mode == 2 && !st->last_vals) { err = "cnl_solve before cnl_factorize"; return 5; }
  for (int64_t b = 0; b < st->batch; b++) {
    double* S = st->S + b * ns * ns;
    double* vb = vals ? vals + (size_t)b * D.nnz : nullptr;
    int rc;
    if (mode == 2) {
      // Jd / w of this problem again (they are shared scratch), then the solve with the stored factor
      hipLaunchKernelGGL(gather_kernel, dim3(blocks((long long)m * n)), dim3(256), 0, stream, st->last_vals + (size_t)b * D.nnz, st->jslot,
                         st->dslot, m, n, st->Jd, st->JW, st->w);
      if ((rc = solve(st, D, S, rhs + b * N, d + b * N, stream, err))) return rc;
      continue;
    }
    if ((rc = build(st, D, vb, eig_tol, stream, err))) return rc;
    int cnt[4];
    auto attempt = [&](double rho, int use_slots) -> int {
      DCHK(hipMemsetAsync(st->cnt, 0, 2 * sizeof(int), stream));
      int r2 = factor(st, D, S, vb + (D.nnz - n), rho, use_slots, eig_tol, stream, err);
      if (r2) return r2;
      DCHK(hipMemcpyAsync(cnt, st->cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
      DCHK(hipStreamSynchronize(stream));
      return 0;
    };
    auto ok = [&]() { return cnt[0] + cnt[2] == n && cnt[1] + cnt[3] == 0; };  // src/solver_types.jl:90-97
    if (mode == 1) {
      if ((rc = attempt(0.0, 1))) return rc;
      const int32_t s32 = ok() ? 1 : 0;
      DCHK(hipMemcpyAsync(success_d + b, &s32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
      if (npos_d) { const int64_t v = cnt[0] + cnt[2]; DCHK(hipMemcpyAsync(npos_d + b, &v, sizeof(int64_t), hipMemcpyHostToDevice, stream)); }
      if (nzero_d) { const int64_t v = cnt[1] + cnt[3]; DCHK(hipMemcpyAsync(nzero_d + b, &v, sizeof(int64_t), hipMemcpyHostToDevice, stream)); }
      DCHK(hipStreamSynchronize(stream));
      continue;
    }
    // newton_system!, src/CaNNOLeS.jl:1008-1052
    double rho_old = 0.0;
    DCHK(hipMemcpyAsync(&rho_old, rho_old_d + b, sizeof(double), hipMemcpyDeviceToHost, stream));
    DCHK(hipStreamSynchronize(stream));
    double rho = 0.0, wrote = 0.0;
    int nfact = 1;
    if ((rc = attempt(0.0, 1))) return rc;
    bool success = ok();
    if (!success) {
      rho = rho_old == 0.0 ? rho0 : std::max(rhomin, kdec * rho_old);
      wrote = rho;
      if ((rc = attempt(rho, 0))) return rc;
      success = ok();
      nfact++;
      while (!success && rho <= rhomax) {
        rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
        if (rho <= rhomax) {
          wrote = rho;
          if ((rc = attempt(rho, 0))) return rc;
          success = ok();
          nfact++;
        }
      }
      if (rho <= rhomax) rho_old = rho;
      hipLaunchKernelGGL(fill_kernel, dim3(blocks(n)), dim3(256), 0, stream, vb + (D.nnz - n), wrote, n);
    }
    if (success && (rc = solve(st, D, S, rhs + b * N, d + b * N, stream, err))) return rc;
    const int32_t nf32 = nfact, s32 = success ? 1 : 0;
    DCHK(hipMemcpyAsync(rho_d + b, &rho, sizeof(double), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(rho_old_d + b, &rho_old, sizeof(double), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(nfact_d + b, &nf32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(success_d + b, &s32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    DCHK(hipStreamSynchronize(stream));
  }
  if (mode == 1) st->last_vals = vals;
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
int dense_create_general(DenseState** out, int32_t ns, int64_t batch, std::string& err) {
  DenseState* st = new DenseState();
  st->batch = batch;
  *out = st;
  int rc;
  const size_t n = (size_t)ns;
  if ((rc = dalloc_(st, &st->cnt, 4, err))) return rc;
  if ((rc = dalloc_(st, &st->S0, n * n, err))) return rc;
  if ((rc = dalloc_(st, &st->W21, n * NB, err))) return rc;
  if ((rc = dalloc_(st, &st->y, n, err))) return rc;
  if ((rc = dalloc_(st, &st->S, (size_t)batch * n * n, err))) return rc;
  BCHK(rocblas_create_handle(&st->blas));
  return 0;
}

int dense_run_general(DenseState* st, const GeneralOps& G, int mode, const double* cbuf, const int* xpos, const int* xzer, double* d2,
                      double* vals_rho0, int64_t vals_stride, double* rho_old_d, double* rho_d, int32_t* nfact_d, int32_t* success_d,
                      int64_t* npos_d, int64_t* nzero_d, const double params[9], hipStream_t stream, std::string& err) {
  const int ns = G.ns, nv = G.nv;
  BCHK(rocblas_set_stream(st->blas, stream));
  const double eig_tol = params[0], kdec = params[2], kinc = params[3], klarge = params[4], rho0 = params[5], rhomax = params[6],
               rhomin = params[7];
  for (int64_t b = 0; b < st->batch; b++) {
    double* S = st->S + (size_t)b * ns * ns;
    const double* cb = cbuf + (size_t)b * G.cstride;
    int rc = 0;
    auto solve_b = [&]() -> int {
      // x = S^-1 crhs;  d2 = -x
      DCHK(hipMemcpyAsync(st->y, cb + G.nslots + nv, (size_t)ns * sizeof(double), hipMemcpyDeviceToDevice, stream));
      BCHK(rocblas_dtrsv(st->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_unit, ns, S, ns, st->y, 1));
      hipLaunchKernelGGL(scale_by_diag_kernel, dim3(blocks(ns)), dim3(256), 0, stream, st->y, S, ns);
      BCHK(rocblas_dtrsv(st->blas, rocblas_fill_lower, rocblas_operation_transpose, rocblas_diagonal_unit, ns, S, ns, st->y, 1));
      hipLaunchKernelGGL(neg_kernel, dim3(blocks(ns)), dim3(256), 0, stream, st->y, d2 + (size_t)b * ns, ns);
      DCHK(hipGetLastError());
      return 0;
    };
    if (mode == 2) { if ((rc = solve_b())) return rc; continue; }
    DCHK(hipMemsetAsync(st->S0, 0, (size_t)ns * ns * sizeof(double), stream));
    hipLaunchKernelGGL(slot_scatter_kernel, dim3(blocks(G.nslots)), dim3(256), 0, stream, cb, G.d_pos, G.nslots, st->S0);
    DCHK(hipMemcpyAsync(st->cnt + 2, xpos + b, sizeof(int), hipMemcpyDeviceToDevice, stream));
    DCHK(hipMemcpyAsync(st->cnt + 3, xzer + b, sizeof(int), hipMemcpyDeviceToDevice, stream));
    int cnt[4];
    auto attempt = [&](double rho, int use_slots) -> int {
      DCHK(hipMemsetAsync(st->cnt, 0, 2 * sizeof(int), stream));
      int r2 = factor_ns(st, ns, nv, S, cb + G.nslots, rho, use_slots, eig_tol, stream, err);
      if (r2) return r2;
      DCHK(hipMemcpyAsync(cnt, st->cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
      DCHK(hipStreamSynchronize(stream));
      return 0;
    };
    auto ok = [&]() { return cnt[0] + cnt[2] == nv && cnt[1] + cnt[3] == 0; };  // src/solver_types.jl:90-97
    if (mode == 1) {
      if ((rc = attempt(0.0, 1))) return rc;
      const int32_t s32 = ok() ? 1 : 0;
      DCHK(hipMemcpyAsync(success_d + b, &s32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
      if (npos_d) { const int64_t v = cnt[0] + cnt[2]; DCHK(hipMemcpyAsync(npos_d + b, &v, sizeof(int64_t), hipMemcpyHostToDevice, stream)); }
      if (nzero_d) { const int64_t v = cnt[1] + cnt[3]; DCHK(hipMemcpyAsync(nzero_d + b, &v, sizeof(int64_t), hipMemcpyHostToDevice, stream)); }
      DCHK(hipStreamSynchronize(stream));
      continue;
    }
    // newton_system!, src/CaNNOLeS.jl:1008-1052
    double rho_old = 0.0;
    DCHK(hipMemcpyAsync(&rho_old, rho_old_d + b, sizeof(double), hipMemcpyDeviceToHost, stream));
    DCHK(hipStreamSynchronize(stream));
    double rho = 0.0, wrote = 0.0;
    int nfact = 1;
    if ((rc = attempt(0.0, 1))) return rc;
    bool success = ok();
    if (!success) {
      rho = rho_old == 0.0 ? rho0 : std::max(rhomin, kdec * rho_old);
      wrote = rho;
      if ((rc = attempt(rho, 0))) return rc;
      success = ok();
      nfact++;
      while (!success && rho <= rhomax) {
        rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
        if (rho <= rhomax) {
          wrote = rho;
          if ((rc = attempt(rho, 0))) return rc;
          success = ok();
          nfact++;
        }
      }
      if (rho <= rhomax) rho_old = rho;
      hipLaunchKernelGGL(fill_kernel, dim3(blocks(nv)), dim3(256), 0, stream, vals_rho0 + (size_t)b * vals_stride, wrote, nv);
    }
    if (success && (rc = solve_b())) return rc;
    const int32_t nf32 = nfact, s32 = success ? 1 : 0;
    DCHK(hipMemcpyAsync(rho_d + b, &rho, sizeof(double), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(rho_old_d + b, &rho_old, sizeof(double), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(nfact_d + b, &nf32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    DCHK(hipMemcpyAsync(success_d + b, &s32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    DCHK(hipStreamSynchronize(stream));
  }
  return 0;
}

}  // namespace cnl
