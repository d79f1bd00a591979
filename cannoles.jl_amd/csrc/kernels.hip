// kernels.hip — gfx950 (MI355X) kernels of the Newton-system path.
//
// One kernel executes the whole reference call
//   newton_system!(d, nvar, nequ, ncon, rhs, vals, LDLT, rho_old, params)
//   (/root/reference/src/CaNNOLeS.jl:1008-1052)
// for a batch of problems that share one sparsity pattern: KKT assembly with
// COO-order duplicate summation (set_vals!, src/solver_types.jl:53-59), LDL^T
// factorisation, inertia test (src/solver_types.jl:90-97), the per-problem rho
// ladder and the solve d = -K^-1 rhs (src/solver_types.jl:69-77).
//
// Mapping: a group of TPP threads (a wavefront, a fraction of one, or a whole
// workgroup) owns one problem and walks the static multifrontal plan built on
// the host (analysis.cpp).  Fronts are packed lower triangles addressed in
// REVERSED order: local index 0 is the right-hand-side row, 1..nupd the update
// rows, the pivots come last and are eliminated from the highest index down,
// so that after the pivots are gone the update matrix is the packed prefix of
// the front and the L panel is its packed suffix (one contiguous, coalesced
// store to HBM).  The update-matrix stack lives in LDS at offsets computed on
// the host; there is no inter-workgroup communication and no atomic.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace cnl {

__device__ __forceinline__ int tri_i(int i) { return (int)(((unsigned)i * (unsigned)(i + 1)) >> 1); }

__device__ __forceinline__ void tri_decode(int t, int& a, int& b) {
  a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (tri_i(a + 1) <= t) a++;
  while (tri_i(a) > t) a--;
  b = t - tri_i(a);
}

// synchronise the TPP threads that own one problem
template <int TPP, bool LDSW>
__device__ __forceinline__ void psync() {
  if (TPP <= 64) {
    // one wavefront (or part of one): LDS operations of a wave execute in
    // order, so a compiler-level fence is all that is needed; work areas in
    // global memory additionally need the stores drained (workgroup scope).
    if (LDSW) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// orders the factor phase (L panel stores) before the solve phase (panel loads by other lanes)
template <int TPP>
__device__ __forceinline__ void phase_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (TPP <= 64) __builtin_amdgcn_wave_barrier();
  else __syncthreads();
}

template <int TPP>
__device__ __forceinline__ double psum(double v, double* red, int tid) {
  if (TPP <= 64) {
#pragma unroll
    for (int o = TPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, TPP);
    return v;
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < TPP / 64; w++) s += red[w];
    return s;
  }
}

struct ProblemCtx {
  const double* vals;  // this problem's COO values
  const double* rhs;
  double* L;
  double* W;           // work area (LDS or global)
  double* red;         // cross-wave reduction scratch (TPP > 64)
};

// ---------------------------------------------------------------------------
// forward pass over one front: assemble, extend-add, eliminate the pivots,
// store the L panel, leave the update matrix for the parent.
template <int TPP, bool LDSW, bool WITH_K>
__device__ __forceinline__ void front_forward(const DevPlan& P, const FrontHdr& H, const ProblemCtx& c, int tid,
                                              bool rho_override, double rho, double eig_tol, int& npos, int& nzer) {
  const int nupd = H.nupd, npiv = H.npiv;
  const int f = 1 + nupd + npiv;
  const int tf = tri_i(f);
  const int tu = tri_i(1 + nupd);
  double* F = c.W + H.foff;
  double* Lp = c.L + ((long long)H.lptr_lo | ((long long)H.lptr_hi << 31));

  if (WITH_K) {
    for (int t = tid; t < tf; t += TPP) F[t] = 0.0;
    psync<TPP, LDSW>();
    // assembly rounds (round r holds the r-th duplicate of every slot: COO-order sums)
    for (int r = H.seg_begin; r < H.seg_end; r++) {
      const int e0 = P.seg_ptr[r], e1 = P.seg_ptr[r + 1];
      for (int e = e0 + tid; e < e1; e += TPP) {
        const int src = P.asm_src[e], pos = P.asm_pos[e];
        double v;
        if (src >= P.nnz) v = c.rhs ? c.rhs[src - P.nnz] : 0.0;
        else if (rho_override && src >= P.rho_begin) v = rho;
        else v = c.vals[src];
        F[pos] += v;
      }
      psync<TPP, LDSW>();
    }
    // extend-add of the children's update matrices
    for (int ci = H.child_begin; ci < H.child_end; ci++) {
      const FrontHdr& C = P.fronts[P.child_idx[ci]];
      const int tuc = tri_i(1 + C.nupd);
      const double* U = c.W + C.ubase;
      const int* rel = P.rel_idx + C.rel_begin;
      if (tid < tuc) {
        int a, b;
        tri_decode(tid, a, b);
        for (int t = tid; t < tuc; t += TPP) {
          const int ra = rel[a], rb = rel[b];
          F[tri_i(ra) + rb] += U[t];
          b += TPP;
          while (b > a) { b -= a + 1; a++; }
        }
      }
      psync<TPP, LDSW>();
    }
  } else {
    // vector-only forward solve (cnl_solve): F is the front's right-hand side vector
    for (int t = tid; t < f; t += TPP) {
      double v = 0.0;
      if (t > nupd) v = c.rhs[P.perm[H.first_piv + (f - 1 - t)]];
      F[t] = v;
    }
    psync<TPP, LDSW>();
    for (int ci = H.child_begin; ci < H.child_end; ci++) {
      const FrontHdr& C = P.fronts[P.child_idx[ci]];
      const double* U = c.W + C.ubase;
      const int* rel = P.rel_idx + C.rel_begin;
      for (int a = 1 + tid; a <= C.nupd; a += TPP) F[rel[a]] += U[a];
      psync<TPP, LDSW>();
    }
  }

  if (WITH_K) {
    double* wv0 = c.W + P.wv_off;
    const int idep = f - H.indep;  // pivots with local index >= idep are mutually independent
    for (int i = f - 1; i > nupd; i--) {
      const int ulim = i >= idep ? idep : i;
      double* rowi = F + tri_i(i);
      double* wv = wv0 + ((i & 1) ? P.fmax : 0);
      const double dpiv = rowi[i];
      npos += dpiv > eig_tol;
      nzer += fabs(dpiv) <= eig_tol;
      for (int j = tid; j < ulim; j += TPP) {
        const double w = rowi[j];
        wv[j] = w;
        rowi[j] = w / dpiv;
      }
      psync<TPP, LDSW>();
      const int tul = tri_i(ulim);
      if (tid < tul) {
        int a, b;
        tri_decode(tid, a, b);
        for (int t = tid; t < tul; t += TPP) {
          F[t] -= rowi[a] * wv[b];
          b += TPP;
          while (b > a) { b -= a + 1; a++; }
        }
      }
      psync<TPP, LDSW>();
    }
    // L panel (packed suffix of the front) -> HBM, coalesced
    for (int t = tu + tid; t < tf; t += TPP) Lp[t - tu] = F[t];
  } else {
    // forward substitution with the stored panel: z_i = w_i / d_i, v_j -= l_ij w_i
    const int idep = f - H.indep;
    double* PB = c.W + P.pb_off;
    const int plen = tf - tu;
    for (int t = tid; t < plen; t += TPP) PB[t] = Lp[t];
    psync<TPP, LDSW>();
    for (int i = f - 1; i > nupd; i--) {
      const int ulim = i >= idep ? idep : i;
      const double* rowi = PB + (tri_i(i) - tu);
      const double w = F[i];
      psync<TPP, LDSW>();
      for (int j = 1 + tid; j < ulim; j += TPP) F[j] -= rowi[j] * w;
      if (tid == 0) Lp[tri_i(i) - tu] = w / rowi[i];
      psync<TPP, LDSW>();
    }
  }

  // hand the update matrix (or vector) to the parent: move it down the stack
  const int ulen = WITH_K ? tu : 1 + nupd;
  if (H.ubase != H.foff) {
    double* U = c.W + H.ubase;
    for (int t0 = 0; t0 < ulen; t0 += TPP) {
      const int t = t0 + tid;
      double v = 0.0;
      if (t < ulen) v = F[t];
      psync<TPP, LDSW>();
      if (t < ulen) U[t] = v;
      psync<TPP, LDSW>();
    }
  } else {
    psync<TPP, LDSW>();
  }
}

// backward pass over one front: x_i = z_i - sum_j l_ij x_j ;  d = -x
template <int TPP, bool LDSW>
__device__ __forceinline__ void front_backward(const DevPlan& P, const FrontHdr& H, const ProblemCtx& c, int tid, double* dout) {
  const int nupd = H.nupd, npiv = H.npiv;
  const int f = 1 + nupd + npiv;
  const int tu = tri_i(1 + nupd);
  const int plen = tri_i(f) - tu;
  double* X = c.W + H.xoff;
  const double* Lp = c.L + ((long long)H.lptr_lo | ((long long)H.lptr_hi << 31));
  double* PB = c.W + P.pb_off;
  for (int t = tid; t < plen; t += TPP) PB[t] = Lp[t];
  if (H.parent >= 0) {
    const double* Xp = c.W + P.fronts[H.parent].xoff;
    const int* rel = P.rel_idx + H.rel_begin;
    for (int j0 = 1; j0 <= nupd; j0 += TPP) {
      const int j = j0 + tid;
      double v = 0.0;
      if (j <= nupd) v = Xp[rel[j]];
      psync<TPP, LDSW>();
      if (j <= nupd) X[j] = v;
      psync<TPP, LDSW>();
    }
  }
  psync<TPP, LDSW>();
  for (int i = nupd + 1; i < f; i++) {
    const double* rowi = PB + (tri_i(i) - tu);
    double part = 0.0;
    for (int j = 1 + tid; j < i; j += TPP) part += rowi[j] * X[j];
    const double acc = psum<TPP>(part, c.red, tid);
    const double xi = rowi[0] - acc;
    if (tid == 0) {
      X[i] = xi;
      dout[P.perm[H.first_piv + (f - 1 - i)]] = -xi;
    }
    psync<TPP, LDSW>();
  }
}

template <int TPP, bool LDSW>
__device__ __forceinline__ bool factor_attempt(const DevPlan& P, const ProblemCtx& c, int tid, bool rho_override, double rho,
                                               double eig_tol, int& npos_out, int& nzer_out, int xpos, int xzer) {
  int npos = xpos, nzer = xzer;
  for (int s = 0; s < P.nsuper; s++) front_forward<TPP, LDSW, true>(P, P.fronts[s], c, tid, rho_override, rho, eig_tol, npos, nzer);
  npos_out = npos; nzer_out = nzer;
  return npos == P.nvar && nzer == 0;  // src/solver_types.jl:96
}

template <class T>
__device__ __forceinline__ T* as_global(T* p) {  // struct members are generic pointers to the compiler: mark them global
  return (T*)(__attribute__((address_space(1))) T*)p;
}

template <int TPP, int PPB, bool LDSW>
__global__ void __launch_bounds__(TPP* PPB) newton_kernel(const DevPlan Pin, const LaunchArgs Ain) {
  DevPlan P = Pin;
  P.fronts = as_global(Pin.fronts); P.seg_ptr = as_global(Pin.seg_ptr); P.asm_pos = as_global(Pin.asm_pos);
  P.asm_src = as_global(Pin.asm_src); P.child_idx = as_global(Pin.child_idx); P.rel_idx = as_global(Pin.rel_idx);
  P.perm = as_global(Pin.perm);
  LaunchArgs A = Ain;
  A.vals = as_global(Ain.vals); A.rhs = as_global(Ain.rhs); A.d = as_global(Ain.d); A.L = as_global(Ain.L);
  A.scratch = as_global(Ain.scratch); A.rho_old = as_global(Ain.rho_old); A.rho = as_global(Ain.rho);
  A.nfact = as_global(Ain.nfact); A.success = as_global(Ain.success); A.npos = as_global(Ain.npos); A.nzero = as_global(Ain.nzero);
  A.extra_pos = as_global(Ain.extra_pos); A.extra_zer = as_global(Ain.extra_zer);
  extern __shared__ double smem[];
  const int gl = threadIdx.x / TPP;     // problem slot inside the workgroup
  const int tid = threadIdx.x % TPP;
  const int b = blockIdx.x * PPB + gl;
  if (b >= A.batch) return;             // whole owner group exits together (no block barrier when PPB > 1)
  ProblemCtx c;
  c.vals = A.vals ? A.vals + (long long)b * P.vstride : nullptr;
  c.rhs = A.rhs ? A.rhs + (long long)b * P.rstride : nullptr;
  c.L = A.L + (long long)b * P.lsize;
  double* redbase = smem;               // 16 doubles for cross-wave sums (TPP > 64)
  c.red = redbase;
  c.W = LDSW ? (smem + 16 + (long long)gl * P.work_doubles) : (A.scratch + (long long)b * P.work_doubles);
  double* dout = A.d ? A.d + (long long)b * P.dstride : nullptr;
  const double eig_tol = A.params[0];
  const int xpos = A.extra_pos ? A.extra_pos[b] : 0, xzer = A.extra_zer ? A.extra_zer[b] : 0;

  if (A.mode == MODE_SOLVE) {
    int np = 0, nz = 0;
    for (int s = 0; s < P.nsuper; s++) front_forward<TPP, LDSW, false>(P, P.fronts[s], c, tid, false, 0.0, eig_tol, np, nz);
    phase_fence<TPP>();
    for (int s = P.nsuper - 1; s >= 0; s--) front_backward<TPP, LDSW>(P, P.fronts[s], c, tid, dout);
    return;
  }
  if (A.mode == MODE_FACTOR) {
    int np, nz;
    ProblemCtx cf = c; cf.rhs = nullptr;
    const bool ok = factor_attempt<TPP, LDSW>(P, cf, tid, false, 0.0, eig_tol, np, nz, xpos, xzer);
    if (tid == 0) {
      A.success[b] = ok ? 1 : 0;
      if (A.npos) A.npos[b] = np;
      if (A.nzero) A.nzero[b] = nz;
    }
    return;
  }
  // ---- newton_system!: src/CaNNOLeS.jl:1019-1051 ----
  const double kdec = A.params[2], kinc = A.params[3], klarge = A.params[4], rho0 = A.params[5],
               rhomax = A.params[6], rhomin = A.params[7];
  double rho_old = A.rho_old[b];
  double rho = 0.0, wrote = 0.0;
  int nfact = 0, np, nz;
  bool success = factor_attempt<TPP, LDSW>(P, c, tid, false, 0.0, eig_tol, np, nz, xpos, xzer);
  nfact++;
  if (!success) {
    rho = rho_old == 0.0 ? rho0 : fmax(rhomin, kdec * rho_old);
    wrote = rho;
    success = factor_attempt<TPP, LDSW>(P, c, tid, true, rho, eig_tol, np, nz, xpos, xzer);
    nfact++;
    while (!success && rho <= rhomax) {
      rho = rho_old == 0.0 ? klarge * rho : kinc * rho;
      if (rho <= rhomax) {
        wrote = rho;
        success = factor_attempt<TPP, LDSW>(P, c, tid, true, rho, eig_tol, np, nz, xpos, xzer);
        nfact++;
      }
    }
    if (rho <= rhomax) rho_old = rho;
    // the reference leaves the last rho tried in the rho slots of vals
    double* vt = A.vals + (long long)b * P.vstride + P.rho_begin;
    for (int i = tid; i < P.nvar; i += TPP) vt[i] = wrote;
  }
  phase_fence<TPP>();
  if (success)
    for (int s = P.nsuper - 1; s >= 0; s--) front_backward<TPP, LDSW>(P, P.fronts[s], c, tid, dout);
  if (tid == 0) {
    A.rho[b] = rho;
    A.rho_old[b] = rho_old;
    A.nfact[b] = nfact;
    A.success[b] = success ? 1 : 0;
  }
}

size_t max_lds_bytes() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
  return (size_t)v;
}

template <int TPP, int PPB, bool LDSW>
static hipError_t launch_t(const DevPlan& P, const KernelConfig& cfg, const LaunchArgs& a, hipStream_t stream) {
  auto kfn = newton_kernel<TPP, PPB, LDSW>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)cfg.lds_bytes));
  if (e != hipSuccess) return e;
  const int grid = (a.batch + PPB - 1) / PPB;
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(TPP * PPB), cfg.lds_bytes, stream, P, a);
  return hipGetLastError();
}

hipError_t launch_newton(const DevPlan& P, const KernelConfig& cfg, const LaunchArgs& a, hipStream_t stream) {
#define CNL_CASE(T, B, W) \
  if (cfg.tpp == T && cfg.ppb == B && (cfg.lds_work != 0) == W) return launch_t<T, B, W>(P, cfg, a, stream);
  CNL_CASE(64, 16, true) CNL_CASE(64, 8, true) CNL_CASE(64, 4, true) CNL_CASE(64, 2, true) CNL_CASE(64, 1, true)
  CNL_CASE(32, 32, true) CNL_CASE(32, 16, true) CNL_CASE(32, 8, true) CNL_CASE(32, 2, true)
  CNL_CASE(16, 64, true) CNL_CASE(16, 32, true) CNL_CASE(16, 16, true) CNL_CASE(16, 4, true)
  CNL_CASE(256, 1, true) CNL_CASE(1024, 1, true)
  CNL_CASE(64, 4, false) CNL_CASE(256, 1, false) CNL_CASE(1024, 1, false)
#undef CNL_CASE
  return hipErrorInvalidConfiguration;
}

}  // namespace cnl
