// kernels_aux.hip — thread-parallel pre/post passes around the multifrontal kernels (see condense.h):
//   condense_kernel : vals/rhs -> condensed slots [K2 entries | rho tail | condensed rhs]
//   cond_inertia    : inertia contribution of the condensed residual pivots d_r
//   expand_kernel   : solution of the condensed system -> d of the full system, r components recovered
// They stream over independent (problem, entry) pairs at full occupancy; no dependency chains.
#include <hip/hip_runtime.h>

#include "band.h"
#include "condense.h"
#include "kernels.h"

namespace cnl {

template <class T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(__attribute__((address_space(1))) T*)p; }
__device__ __forceinline__ DevCond globalize(const DevCond& Cin) {
  DevCond C = Cin;
  C.c_order = as_global(Cin.c_order);
  C.ch_slot = as_global(Cin.ch_slot); C.ch_rng = as_global(Cin.ch_rng); C.ch_tile = as_global(Cin.ch_tile);
  C.rng_start = as_global(Cin.rng_start); C.rng_len = as_global(Cin.rng_len);
  C.c_la = as_global(Cin.c_la); C.c_lb = as_global(Cin.c_lb); C.c_ld = as_global(Cin.c_ld);
  C.ch_tptr = as_global(Cin.ch_tptr); C.tile_src = as_global(Cin.tile_src); C.c_pack = as_global(Cin.c_pack);
  C.c_ptr = as_global(Cin.c_ptr); C.c_a = as_global(Cin.c_a); C.c_b = as_global(Cin.c_b); C.c_d = as_global(Cin.c_d);
  C.r_dsrc = as_global(Cin.r_dsrc); C.r_ptr = as_global(Cin.r_ptr); C.r_jsrc = as_global(Cin.r_jsrc); C.r_jx = as_global(Cin.r_jx);
  C.red_of = as_global(Cin.red_of); C.cidx_of = as_global(Cin.cidx_of);
  C.orig_of = as_global(Cin.orig_of); C.r_orig = as_global(Cin.r_orig);
  return C;
}

// refined reciprocal division (same as the multifrontal kernel's): shorter than the IEEE expansion
__device__ __forceinline__ double fast_div_aux(double w, double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  const double q = w * r;
  return fma(fma(-d, q, w), r, q);
}

// One thread forms one slot for CPB problems: the contribution indices are read once and reused for
// every problem, and the CPB independent gathers per index give the memory system some parallelism.
constexpr int CPB = 4;
__global__ void __launch_bounds__(256) condense_kernel(const DevCond Cin, const double* __restrict__ vals,
                                                       const double* __restrict__ rhs, double* __restrict__ cbuf,
                                                       int slot_begin, int slot_end, int batch) {
  const DevCond C = globalize(Cin);
  const int t = slot_begin + blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * CPB;
  if (t >= slot_end || b0 >= batch) return;
  const int s = C.c_order[t];  // slots of equal contribution count sit together: no divergence inside a wavefront
  const double* v[CPB];
  const double* r[CPB];
#pragma unroll
  for (int q = 0; q < CPB; q++) {
    const int b = b0 + q < batch ? b0 + q : batch - 1;
    v[q] = vals + (long long)b * C.nnz;
    r[q] = rhs ? rhs + (long long)b * C.N - C.nnz : v[q];  // entries >= nnz address the right-hand side
  }
  double acc[CPB];
#pragma unroll
  for (int q = 0; q < CPB; q++) acc[q] = 0.0;
  const int c0 = C.c_ptr[s], c1 = C.c_ptr[s + 1];
  for (int c = c0; c < c1; c++) {
    const int a = C.c_a[c], bb = C.c_b[c];
    if (bb < 0) {
#pragma unroll
      for (int q = 0; q < CPB; q++) {
        const double xa = a < C.nnz ? v[q][a] : (rhs ? r[q][a] : 0.0);
        acc[q] += xa;
      }
    } else {
      const int dd = C.c_d[c];
#pragma unroll
      for (int q = 0; q < CPB; q++) {
        const double xa = v[q][a];
        const double xb = bb < C.nnz ? v[q][bb] : (rhs ? r[q][bb] : 0.0);
        acc[q] -= fast_div_aux(xa * xb, v[q][dd]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < CPB; q++)
    if (b0 + q < batch) cbuf[(long long)(b0 + q) * C.cstride + s] = acc[q];
}

// LDS-tiled condense: a workgroup forms the slots of one chunk (a column range: its matrix, rho and rhs
// slots) for TPB problems.  The sources the chunk needs are staged in LDS with coalesced loads (a line of
// vals is fetched about once per problem), then every thread forms slots for the TPB problems from LDS
// (contribution indices are read once per slot and reused for every problem).  `mask` selects the slot
// ranges to produce: 1 matrix, 2 rho, 4 right-hand side.
constexpr int TPB = CONDENSE_TILED_PROBLEMS;
__global__ void __launch_bounds__(256) condense_tiled_kernel(const DevCond Cin, const double* __restrict__ vals,
                                                             const double* __restrict__ rhs, double* __restrict__ cbuf,
                                                             int mask, int batch) {
  extern __shared__ double tile[];
  const DevCond C = globalize(Cin);
  const int ch = blockIdx.x;
  const int b0 = blockIdx.y * TPB;
  const int T = C.ch_tile[ch];
  const int tid = threadIdx.x;
  {
    const int* tsrc = C.tile_src + C.ch_tptr[ch];
    for (int t = tid; t < T; t += 256) {
      const int gsrc = tsrc[t];
#pragma unroll
      for (int q = 0; q < TPB; q++) {
        const int b = b0 + q < batch ? b0 + q : batch - 1;
        double v;
        if (gsrc < C.nnz) v = vals[(long long)b * C.nnz + gsrc];
        else v = rhs ? rhs[(long long)b * C.N + (gsrc - C.nnz)] : 0.0;
        tile[q * T + t] = v;
      }
    }
  }
  // stage the contribution lists of the chunk too: the compute phase then touches global memory only to store
  unsigned long long* cpk = reinterpret_cast<unsigned long long*>(tile + TPB * C.tile_max);
  int* cpl = reinterpret_cast<int*>(cpk + C.chunk_ncon_max);  // local contribution pointer of every slot of the chunk (+1 per range)
  const int* rg = C.ch_slot + 6 * ch;
  int coff = 0, soff = 0;
  for (int r = 0; r < 3; r++) {
    const int s0 = rg[2 * r], len = rg[2 * r + 1];
    const int cb = C.c_ptr[s0], ce = C.c_ptr[s0 + len];
    for (int i = tid; i < ce - cb; i += 256) cpk[coff + i] = C.c_pack[cb + i];
    for (int i = tid; i <= len; i += 256) cpl[soff + i] = C.c_ptr[s0 + i] - cb + coff;
    coff += ce - cb;
    soff += len + 1;
  }
  __syncthreads();
  soff = 0;
  for (int r = 0; r < 3; r++) {
    const int s0 = rg[2 * r], len = rg[2 * r + 1];
    if (mask & (1 << r)) {
      for (int t = tid; t < len; t += 256) {
        double acc[TPB];
#pragma unroll
        for (int q = 0; q < TPB; q++) acc[q] = 0.0;
        const int c0 = cpl[soff + t], c1 = cpl[soff + t + 1];
        for (int c = c0; c < c1; c++) {
          const unsigned long long pk = cpk[c];
          const int la = (int)(pk & 0xffff), lb = (int)((pk >> 16) & 0xffff) - 1;
          if (lb < 0) {
#pragma unroll
            for (int q = 0; q < TPB; q++) acc[q] += tile[q * T + la];
          } else {
            const int ld = (int)((pk >> 32) & 0xffff) - 1;
#pragma unroll
            for (int q = 0; q < TPB; q++) acc[q] -= fast_div_aux(tile[q * T + la] * tile[q * T + lb], tile[q * T + ld]);
          }
        }
#pragma unroll
        for (int q = 0; q < TPB; q++)
          if (b0 + q < batch) cbuf[(long long)(b0 + q) * C.cstride + s0 + t] = acc[q];
      }
    }
    soff += len + 1;
  }
}

// pos_r = #{d_r > eig_tol}, zer_r = #{|d_r| <= eig_tol} over the condensed pivots (src/solver_types.jl:90-95).
// One workgroup per problem; the counts are written, not accumulated (no memset before the launch).
__global__ void __launch_bounds__(256) cond_inertia_kernel(const DevCond Cin, const double* __restrict__ vals, int* extra_pos,
                                                           int* extra_zer, double eig_tol, int batch) {
  const DevCond C = globalize(Cin);
  const int b = blockIdx.x;
  __shared__ int sp[4], sz[4];
  const double* v = vals + (long long)b * C.nnz;
  int pos = 0, zer = 0;
  for (int q = threadIdx.x; q < C.ncond; q += 256) {
    const double d = v[C.r_dsrc[q]];
    pos += d > eig_tol;
    zer += fabs(d) <= eig_tol;
  }
  for (int o = 32; o > 0; o >>= 1) { pos += __shfl_xor(pos, o, 64); zer += __shfl_xor(zer, o, 64); }
  if ((threadIdx.x & 63) == 0) { sp[threadIdx.x >> 6] = pos; sz[threadIdx.x >> 6] = zer; }
  __syncthreads();
  if (threadIdx.x == 0) {
    extra_pos[b] = sp[0] + sp[1] + sp[2] + sp[3];
    extra_zer[b] = sz[0] + sz[1] + sz[2] + sz[3];
  }
}

// d_full from the condensed solution d2 (= -K2^-1 crhs): kept nodes copy; condensed r:
//   sol_r = (rhs_r - sum_k J_rk sol_xk) / d_r,  d_r_out = -sol_r = -(rhs_r + sum_k J_rk d2_xk) / d_r
// Workgroups [0, nb_copy) copy 256 kept entries each.  The others recover XRB residual components each: the
// products J_rk d2_xk are formed entry-per-thread (consecutive threads read consecutive Jacobian entries; a
// thread-per-row loop reads them with a stride of the row length and re-fetches every line from L2 once per
// column), parked in LDS and summed row-per-thread.
constexpr int XRB = 256;    // residual rows per workgroup
constexpr int XEMAX = 2048; // Jacobian entries staged per workgroup (longer chunks take the row-per-thread loop)
constexpr int XPB = 8;      // problems per workgroup: the index lists are read once and reused
// d2 == nullptr: the multifrontal kernel has already written the kept components into dout (caller's numbering);
// only the residual components are recovered, reading the x components from dout itself.
__global__ void __launch_bounds__(256) expand_kernel(const DevCond Cin, double* __restrict__ vals, const double* __restrict__ rhs,
                                                     const double* d2, const double* __restrict__ cbuf,
                                                     double* dout, const int* __restrict__ success,
                                                     int copy_rho_tail, int nb_copy, int batch) {
  const DevCond C = globalize(Cin);
  const int b0 = blockIdx.y * XPB;
  const int t = threadIdx.x;
  __shared__ double prod[2 * XEMAX];
  if ((int)blockIdx.x < nb_copy) {
    const int j = blockIdx.x * 256 + t;
    const int oj = j < C.N2 ? C.orig_of[j] : 0;
    for (int q = 0; q < XPB && b0 + q < batch; q++) {
      const long long b = b0 + q;
      if (copy_rho_tail && j < C.nvar) vals[b * C.nnz + (C.nnz - C.nvar) + j] = cbuf[b * C.cstride + C.ncs + j];
      if (success && !success[b]) continue;
      if (j < C.N2) dout[b * C.N + oj] = d2[b * C.N2 + j];
    }
    return;
  }
  const int q0 = (blockIdx.x - nb_copy) * XRB;
  const int q1 = q0 + XRB < C.ncond ? q0 + XRB : C.ncond;
  const int e0 = C.r_ptr[q0], e1 = C.r_ptr[q1];
  const bool staged = e1 - e0 <= XEMAX;  // workgroup-uniform
  // this thread's share of the entry list and its row, read once for all problems of the workgroup
  constexpr int XE = XEMAX / 256;
  int js[XE], jx[XE];
#pragma unroll
  for (int k = 0; k < XE; k++) {
    const int e = e0 + t + 256 * k;
    js[k] = (staged && e < e1) ? C.r_jsrc[e] : 0;
    jx[k] = (staged && e < e1) ? C.r_jx[e] : 0;
  }
  const int qr = q0 + t;
  const bool has_row = qr < q1;
  const int i = has_row ? C.r_orig[qr] : 0;
  const int k0 = has_row ? C.r_ptr[qr] : 0, k1 = has_row ? C.r_ptr[qr + 1] : 0;
  const int dsrc = has_row ? C.r_dsrc[qr] : 0;
  const int nq = batch - b0 < XPB ? batch - b0 : XPB;
  if (staged) {
    // Problems are pipelined: the operands of problem q + 1 are loaded into registers before the products of problem q are
    // parked and summed (two LDS buffers, one barrier per problem), so the loads of the next problem are in flight while
    // this one is reduced.
    double jv[XE], xv[XE], rh = 0.0, dv = 1.0;
    bool ok_cur = false;
    auto load = [&](int q, double (&jv_)[XE], double (&xv_)[XE], double& rh_, double& dv_) -> bool {
      const long long b = b0 + q;
      if (q >= nq || (success && !success[b])) return false;  // workgroup-uniform
      const double* v = vals + b * C.nnz;
      const double* x2 = d2 ? d2 + b * C.N2 : dout + b * C.N;
#pragma unroll
      for (int k = 0; k < XE; k++) {
        const bool in = t + 256 * k < e1 - e0;
        jv_[k] = in ? v[js[k]] : 0.0;
        xv_[k] = in ? x2[jx[k]] : 0.0;
      }
      rh_ = has_row ? rhs[b * C.N + i] : 0.0;
      dv_ = has_row ? v[dsrc] : 1.0;
      return true;
    };
    ok_cur = load(0, jv, xv, rh, dv);
    for (int q = 0; q < nq; q++) {
      double jn[XE], xn[XE], rhn = 0.0, dvn = 1.0;
      const bool ok_next = load(q + 1, jn, xn, rhn, dvn);
      if (ok_cur) {
        double* pb = prod + (q & 1) * XEMAX;
#pragma unroll
        for (int k = 0; k < XE; k++) {
          const int e = t + 256 * k;
          if (e < e1 - e0) pb[e] = jv[k] * xv[k];
        }
      }
      __syncthreads();  // (also orders the reads of this buffer two problems ago before the writes above)
      if (ok_cur && has_row) {
        const double* pb = prod + (q & 1) * XEMAX;
        double s = rh;
        for (int k = k0; k < k1; k++) s += pb[k - e0];
        dout[(long long)(b0 + q) * C.N + i] = -fast_div_aux(s, dv);
      }
#pragma unroll
      for (int k = 0; k < XE; k++) { jv[k] = jn[k]; xv[k] = xn[k]; }
      rh = rhn; dv = dvn; ok_cur = ok_next;
    }
    return;
  }
  for (int q = 0; q < nq; q++) {
    const long long b = b0 + q;
    if (success && !success[b]) continue;  // workgroup-uniform
    const double* v = vals + b * C.nnz;
    // x components: reduced index == caller's index for the variables (they are never condensed)
    const double* x2 = d2 ? d2 + b * C.N2 : dout + b * C.N;
    if (has_row) {
      double s = rhs[b * C.N + i];
      for (int k = k0; k < k1; k++) s = fma(v[C.r_jsrc[k]], x2[C.r_jx[k]], s);
      dout[b * C.N + i] = -fast_div_aux(s, v[dsrc]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row f1 of the scope table: the vectors either side of the Newton system, kept on the device.
//   dual   = Jx' r - Jc' lambda      /root/reference/src/CaNNOLeS.jl:507-508, 519-521 (and 722-724 at the trial point)
//   primal = [F - r ; c]             :522-524 (725-726)
//   rhs    = [dual ; primal]         :631-632
//   norms  = (||dual||_inf, ||primal||_inf)   :528-529 (730-731)
// The Jacobian values are read from the J_F / J_c segments of `vals` (prepare_newton_system! copies them there,
// :953-967).  Both transposed products are accumulated per column in COO order, which is the order of the
// reference's COO mul! (y[col[k]] += val[k] * x[row[k]] for k = 1, 2, ...), and subtracted afterwards as the
// reference does: the results are bit-identical to that recipe.
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
  // the bit patterns of non-negative doubles (and of NaN, which must win like in norm(., Inf)) are ordered as integers
  atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

// One thread serves entry i of RPT problems: the index lists are read once and the launch has RPT times fewer
// wavefronts (with one problem per thread the kernel is bound by the wavefront launch rate).
constexpr int RPT = 4;
// Entries of a column whose operands are loaded TOGETHER, ahead of the ordered sum.  A per-entry loop (index -> value -> add, with a
// trip count the compiler does not know) serialises two memory round trips per entry: twelve for a band column, which is what
// the kernel spent its time on (2.4 ms at 8192 systems of cfg3's size, 0.34 of the HBM rate; a workgroup walking several groups of
// problems, or the slots staged through LDS, changed nothing).  The sum itself stays sequential in COO order.
constexpr int RV_KF = 6, RV_KC = 2;
__global__ void __launch_bounds__(256) residual_vectors_kernel(const DevJt Jin, const JacSrc S,
                                                               const double* __restrict__ r, const double* __restrict__ lambda,
                                                               const double* __restrict__ Fx, const double* __restrict__ cx,
                                                               double* __restrict__ rhs, double* __restrict__ norms, int batch) {
#pragma clang fp contract(off)  // separately rounded multiply and add, as the reference's scalar loops
  DevJt J = Jin;
  J.ptrF = as_global(Jin.ptrF); J.slotF = as_global(Jin.slotF); J.idxF = as_global(Jin.idxF);
  J.ptrC = as_global(Jin.ptrC); J.slotC = as_global(Jin.slotC); J.idxC = as_global(Jin.idxC);
  const double* __restrict__ vF = S.vF; const double* __restrict__ vC = S.vC;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * RPT;
  long long bq[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) bq[q] = b0 + q < batch ? b0 + q : batch - 1;
  double out[RPT];
#pragma unroll
  for (int q = 0; q < RPT; q++) out[q] = 0.0;
  const bool is_dual = i < J.nvar;
  if (i < J.N) {
    if (is_dual) {
      double s1[RPT], s2[RPT];
#pragma unroll
      for (int q = 0; q < RPT; q++) s1[q] = s2[q] = 0.0;
      const int kf0 = J.ptrF[i], kf1 = J.ptrF[i + 1], kc0 = J.ptrC[i], kc1 = J.ptrC[i + 1];
      {
        // the first RV_KF entries: indices, then all operands, in flight together (entries past the end read a slot that exists —
        // JacSrc::safeF / safeC — and index 0, and are not added)
        int sl[RV_KF], ix[RV_KF];
#pragma unroll
        for (int u = 0; u < RV_KF; u++) { const bool on = kf0 + u < kf1; sl[u] = on ? J.slotF[kf0 + u] : S.safeF; ix[u] = on ? J.idxF[kf0 + u] : 0; }
        double jv[RV_KF][RPT], xv[RV_KF][RPT];
#pragma unroll
        for (int u = 0; u < RV_KF; u++)
#pragma unroll
          for (int q = 0; q < RPT; q++) { jv[u][q] = vF[bq[q] * S.sF + sl[u]]; xv[u][q] = r[bq[q] * J.nequ + ix[u]]; }
#pragma unroll
        for (int u = 0; u < RV_KF; u++)
          if (kf0 + u < kf1) {
#pragma unroll
            for (int q = 0; q < RPT; q++) { const double t_ = jv[u][q] * xv[u][q]; s1[q] = s1[q] + t_; }
          }
      }
      for (int k = kf0 + RV_KF; k < kf1; k++) {
        const int sl = J.slotF[k], ix = J.idxF[k];
#pragma unroll
        for (int q = 0; q < RPT; q++) { const double t_ = vF[bq[q] * S.sF + sl] * r[bq[q] * J.nequ + ix]; s1[q] = s1[q] + t_; }
      }
      {
        int sl[RV_KC], ix[RV_KC];
#pragma unroll
        for (int u = 0; u < RV_KC; u++) { const bool on = kc0 + u < kc1; sl[u] = on ? J.slotC[kc0 + u] : S.safeC; ix[u] = on ? J.idxC[kc0 + u] : 0; }
        double jv[RV_KC][RPT], xv[RV_KC][RPT];
#pragma unroll
        for (int u = 0; u < RV_KC; u++)
#pragma unroll
          for (int q = 0; q < RPT; q++) { jv[u][q] = vC[bq[q] * S.sC + sl[u]]; xv[u][q] = kc0 + u < kc1 ? lambda[bq[q] * J.ncon + ix[u]] : 0.0; }
#pragma unroll
        for (int u = 0; u < RV_KC; u++)
          if (kc0 + u < kc1) {
#pragma unroll
            for (int q = 0; q < RPT; q++) { const double t_ = jv[u][q] * xv[u][q]; s2[q] = s2[q] + t_; }
          }
      }
      for (int k = kc0 + RV_KC; k < kc1; k++) {
        const int sl = J.slotC[k], ix = J.idxC[k];
#pragma unroll
        for (int q = 0; q < RPT; q++) { const double t_ = vC[bq[q] * S.sC + sl] * lambda[bq[q] * J.ncon + ix]; s2[q] = s2[q] + t_; }
      }
#pragma unroll
      for (int q = 0; q < RPT; q++) out[q] = s1[q] - s2[q];
    } else if (i < J.nvar + J.nequ) {
#pragma unroll
      for (int q = 0; q < RPT; q++) out[q] = Fx[bq[q] * J.nequ + (i - J.nvar)] - r[bq[q] * J.nequ + (i - J.nvar)];
    } else {
#pragma unroll
      for (int q = 0; q < RPT; q++) out[q] = cx[bq[q] * J.ncon + (i - J.nvar - J.nequ)];
    }
#pragma unroll
    for (int q = 0; q < RPT; q++)
      if (b0 + q < batch) rhs[bq[q] * J.N + i] = out[q];
  }
  // infinity norms: a workgroup holds entries of one kind except the (at most two) that straddle a boundary.
  // NaN must propagate as in norm(., Inf): fmax drops it, so the integer images are compared (they are ordered like
  // the non-negative doubles, NaN above infinity).
#pragma unroll
  for (int q = 0; q < RPT; q++) {
    const double ad = (i < J.N && is_dual) ? fabs(out[q]) : 0.0, ap = (i < J.N && !is_dual) ? fabs(out[q]) : 0.0;
    unsigned long long ud = (unsigned long long)__double_as_longlong(ad), up = (unsigned long long)__double_as_longlong(ap);
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long od = __shfl_xor(ud, o, 64), op = __shfl_xor(up, o, 64);
      ud = od > ud ? od : ud;
      up = op > up ? op : up;
    }
    if ((threadIdx.x & 63) == 0 && b0 + q < batch) {
      if (ud) atomicMax(reinterpret_cast<unsigned long long*>(norms + 2 * bq[q]), ud);
      if (up) atomicMax(reinterpret_cast<unsigned long long*>(norms + 2 * bq[q] + 1), up);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (round 5) Row f1 through COLUMN TILES.  The gather kernel above spends its time in the texture-address unit: a wavefront's load of
// "entry u of 64 consecutive columns" touches 64 addresses 40 bytes apart (a band Jacobian in row-major COO order), i.e. 20 cache
// lines per instruction for 512 useful bytes, and every line is touched by five such instructions.  Here a workgroup serves RVT_COLS
// consecutive columns: the slot range of `vals` that holds their J_F entries, the range of r they multiply, and the same for J_c /
// lambda are CONTIGUOUS ranges for band and block patterns (DevJt::rv_tiles, found at handle creation), so they are streamed with
// 16-byte loads per lane — one problem ahead, through registers — into LDS, and each thread forms the sums of its column from LDS
// with the offsets of its entries held in registers for all the problems of the workgroup.  Same arithmetic as above: per-column
// sums in COO order, multiply and add rounded separately, the two products subtracted afterwards (bit-identical results; the
// parity tests compare the two kernels bit for bit).  The residual rows a tile has in LDS are also the rows whose primal entry
// F - r it writes (r is read once).
//
// A window of w doubles at g (8-byte aligned) is loaded in 16-byte chunks from the 16-byte boundary at or below g: the chunk may
// start one double before the window and end one double behind it — inside the 16-byte granule of a valid address, never across
// a page — and element k of the window lands in lds[par + k], par = the parity of g.
typedef double rvt_d2 __attribute__((ext_vector_type(2)));   // (HIP's double2 class keeps arrays of it out of registers)
template <int NCH, bool NT = false>
__device__ __forceinline__ void rvt_issue(const double* __restrict__ g, int w, int t, rvt_d2 (&v)[NCH]) {
  const int par = (int)((reinterpret_cast<uintptr_t>(g) >> 3) & 1);
  const rvt_d2* g2 = reinterpret_cast<const rvt_d2*>(g - par);
  const int nch = (w + par + 1) >> 1;
#pragma unroll
  for (int u = 0; u < NCH; u++)
    if (u * 256 < nch) {                      // wavefront-uniform
      const int i = u * 256 + t;
      if (i < nch) v[u] = NT ? __builtin_nontemporal_load(g2 + i) : g2[i];
    }
}
template <int NCH>
__device__ __forceinline__ void rvt_commit(double* lds, int w, int par, int t, const rvt_d2 (&v)[NCH]) {
  const int nch = (w + par + 1) >> 1;
  rvt_d2* l2 = reinterpret_cast<rvt_d2*>(lds);
#pragma unroll
  for (int u = 0; u < NCH; u++)
    if (u * 256 < nch) {
      const int i = u * 256 + t;
      if (i < nch) l2[i] = v[u];
    }
}
__device__ __forceinline__ int rvt_par(const double* g) { return (int)((reinterpret_cast<uintptr_t>(g) >> 3) & 1); }
__device__ __forceinline__ int rvt_even(int w) { return (w + 3) & ~1; }
constexpr int RVT_NFX = (RVT_MAXR + 255) / 256;   // primal rows of a tile per thread

__global__ void __launch_bounds__(256) residual_vectors_tiled_kernel(const DevJt Jin, const JacSrc S,
                                                                     const double* __restrict__ r, const double* __restrict__ lambda,
                                                                     const double* __restrict__ Fx, const double* __restrict__ cx,
                                                                     double* __restrict__ rhs, double* __restrict__ norms, int batch, int pb) {
#pragma clang fp contract(off)
  constexpr bool NT = true;   // streamed once: non-temporal loads and stores (1.29 -> 1.27 ms)
  extern __shared__ rvt_d2 rvt_lds2[];
  double* lds = reinterpret_cast<double*>(rvt_lds2);
  __shared__ unsigned long long red[2];
  DevJt J = Jin;
  J.ptrF = as_global(Jin.ptrF); J.slotF = as_global(Jin.slotF); J.idxF = as_global(Jin.idxF);
  J.ptrC = as_global(Jin.ptrC); J.slotC = as_global(Jin.slotC); J.idxC = as_global(Jin.idxC);
  J.rv_tiles = as_global(Jin.rv_tiles); J.rv_table = as_global(Jin.rv_table);
  const int t = threadIdx.x;
  const int tile = blockIdx.x;
  const int b_begin = blockIdx.y * pb, b_end = min(batch, b_begin + pb);
  if (t < 2) red[t] = 0ull;
  if (tile >= J.rv_ntiles) {
    // rows of the primal part that no column tile owns (the column tiles' row ranges are not an ordered cover of 0 .. nequ)
    const int i0 = (tile - J.rv_ntiles) * RVT_PROWS;
    for (int b = b_begin; b < b_end; b++) {
      unsigned long long up = 0ull;
#pragma unroll
      for (int k = 0; k < RVT_PROWS / 256; k++) {
        const int i = i0 + k * 256 + t;
        if (i < J.nequ) {
          const double o = Fx[(long long)b * J.nequ + i] - r[(long long)b * J.nequ + i];
          rhs[(long long)b * J.N + J.nvar + i] = o;
          const unsigned long long a = (unsigned long long)__double_as_longlong(fabs(o));
          up = a > up ? a : up;
        }
      }
      for (int o = 32; o > 0; o >>= 1) { const unsigned long long x = __shfl_xor(up, o, 64); up = x > up ? x : up; }
      if ((t & 63) == 0 && up) atomicMax(reinterpret_cast<unsigned long long*>(norms + 2 * (long long)b + 1), up);
    }
    return;
  }
  const int32_t* T = J.rv_tiles + tile * RVT_TW;
  const int fslo = T[RVT_FSLO], wF = T[RVT_WF], rlo = T[RVT_RLO], wR = T[RVT_WR], cslo = T[RVT_CSLO], wC = T[RVT_WC], llo = T[RVT_LLO],
            wL = T[RVT_WL], own_lo = T[RVT_OWNLO], own_hi = T[RVT_OWNHI];
  const int oF = 0, oR = rvt_even(wF), oC = oR + rvt_even(wR), oL = oC + rvt_even(wC);
  // the column(s) of this thread: window offsets of their first entries, counts
  constexpr int CPT = RVT_COLS / 256;
  uint32_t tab[CPT][RVT_KF + RVT_KC + 1];
#pragma unroll
  for (int j = 0; j < CPT; j++)
#pragma unroll
    for (int w = 0; w < RVT_KF + RVT_KC + 1; w++)
      tab[j][w] = J.rv_table[((size_t)tile * (RVT_KF + RVT_KC + 1) + w) * RVT_COLS + j * 256 + t];
  rvt_d2 pF[(RVT_MAXF + 2 + 511) / 512], pR[(RVT_MAXR + 2 + 511) / 512], pC[(RVT_MAXC + 2 + 511) / 512], pL[(RVT_MAXL + 2 + 511) / 512];
  double pX[RVT_NFX];
  // (a macro: a lambda that captures the register arrays by reference puts them in scratch memory)
#define RVT_ISSUE(B_)                                                                              \
  {                                                                                                \
    const long long b_ = (B_);                                                                     \
    rvt_issue<4, NT>(S.vF + b_ * S.sF + fslo, wF, t, pF);                                             \
    rvt_issue<1, NT>(r + b_ * J.nequ + rlo, wR, t, pR);                                                  \
    if (wC) {                                                                                      \
      rvt_issue<1, NT>(S.vC + b_ * S.sC + cslo, wC, t, pC);                                           \
      rvt_issue<1, NT>(lambda + b_ * J.ncon + llo, wL, t, pL);                                           \
    }                                                                                              \
    _Pragma("unroll") for (int k = 0; k < RVT_NFX; k++) {                                          \
      const int i = own_lo + k * 256 + t;                                                          \
      if (i < own_hi) pX[k] = __builtin_nontemporal_load(Fx + b_ * J.nequ + i);                    \
    }                                                                                              \
  }
  RVT_ISSUE(b_begin)
  for (int b = b_begin; b < b_end; b++) {
    const int parF = rvt_par(S.vF + (long long)b * S.sF + fslo), parR = rvt_par(r + (long long)b * J.nequ + rlo);
    const int parC = wC ? rvt_par(S.vC + (long long)b * S.sC + cslo) : 0, parL = wC ? rvt_par(lambda + (long long)b * J.ncon + llo) : 0;
    rvt_commit(lds + oF, wF, parF, t, pF);
    rvt_commit(lds + oR, wR, parR, t, pR);
    if (wC) { rvt_commit(lds + oC, wC, parC, t, pC); rvt_commit(lds + oL, wL, parL, t, pL); }
    double fx[RVT_NFX];
#pragma unroll
    for (int k = 0; k < RVT_NFX; k++) fx[k] = pX[k];
    __syncthreads();
    if (b + 1 < b_end) RVT_ISSUE(b + 1)
    unsigned long long ud = 0ull, up = 0ull;
    const double* lF = lds + oF + parF; const double* lR = lds + oR + parR; const double* lC = lds + oC + parC; const double* lL = lds + oL + parL;
#pragma unroll
    for (int j = 0; j < CPT; j++) {
      const int col = tile * RVT_COLS + j * 256 + t;
      const int nF = (int)(tab[j][RVT_KF + RVT_KC] & 255u), nC = (int)((tab[j][RVT_KF + RVT_KC] >> 8) & 255u);
      double jv[RVT_KF], xv[RVT_KF], cv[RVT_KC], lv[RVT_KC];
#pragma unroll
      for (int u = 0; u < RVT_KF; u++) { jv[u] = lF[tab[j][u] & 0xffffu]; xv[u] = lR[tab[j][u] >> 16]; }
      if (wC) {
#pragma unroll
        for (int u = 0; u < RVT_KC; u++) { cv[u] = lC[tab[j][RVT_KF + u] & 0xffffu]; lv[u] = lL[tab[j][RVT_KF + u] >> 16]; }
      }
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int u = 0; u < RVT_KF; u++)
        if (u < nF) { const double t_ = jv[u] * xv[u]; s1 = s1 + t_; }
      if (nF > RVT_KF) {
        const int q0 = J.ptrF[col];
        for (int k = RVT_KF; k < nF; k++) { const double t_ = S.vF[(long long)b * S.sF + J.slotF[q0 + k]] * r[(long long)b * J.nequ + J.idxF[q0 + k]]; s1 = s1 + t_; }
      }
      if (wC) {
#pragma unroll
        for (int u = 0; u < RVT_KC; u++)
          if (u < nC) { const double t_ = cv[u] * lv[u]; s2 = s2 + t_; }
        if (nC > RVT_KC) {
          const int q0 = J.ptrC[col];
          for (int k = RVT_KC; k < nC; k++) { const double t_ = S.vC[(long long)b * S.sC + J.slotC[q0 + k]] * lambda[(long long)b * J.ncon + J.idxC[q0 + k]]; s2 = s2 + t_; }
        }
      }
      if (col < J.nvar) {
        const double o = s1 - s2;
        if (NT) __builtin_nontemporal_store(o, rhs + (long long)b * J.N + col); else rhs[(long long)b * J.N + col] = o;
        const unsigned long long a = (unsigned long long)__double_as_longlong(fabs(o));
        ud = a > ud ? a : ud;
      }
    }
#pragma unroll
    for (int k = 0; k < RVT_NFX; k++) {
      const int i = own_lo + k * 256 + t;
      if (i < own_hi) {
        const double o = fx[k] - lR[i - rlo];
        if (NT) __builtin_nontemporal_store(o, rhs + (long long)b * J.N + J.nvar + i); else rhs[(long long)b * J.N + J.nvar + i] = o;
        const unsigned long long a = (unsigned long long)__double_as_longlong(fabs(o));
        up = a > up ? a : up;
      }
    }
    if (tile == 0)
      for (int k = t; k < J.ncon; k += 256) {
        const double o = cx[(long long)b * J.ncon + k];
        rhs[(long long)b * J.N + J.nvar + J.nequ + k] = o;
        const unsigned long long a = (unsigned long long)__double_as_longlong(fabs(o));
        up = a > up ? a : up;
      }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long xd = __shfl_xor(ud, o, 64), xp = __shfl_xor(up, o, 64);
      ud = xd > ud ? xd : ud;
      up = xp > up ? xp : up;
    }
    if ((t & 63) == 0) {
      if (ud) atomicMax(&red[0], ud);
      if (up) atomicMax(&red[1], up);
    }
    __syncthreads();
    if (t < 2) {
      const unsigned long long v = red[t];
      if (v) atomicMax(reinterpret_cast<unsigned long long*>(norms + 2 * (long long)b + t), v);
      red[t] = 0ull;
    }
  }
}

// Trial point of the extrapolation step, /root/reference/src/CaNNOLeS.jl:661-668 with dlambda = -d[n+m+1:N] (:654):
//   xt = x + dx, rt = r + dr, dlambda capped at ||dlambda||_2 <= max_dlambda (1e4), lambdat = lambda + dlambda.
// Workgroup (c, b): elements [1024 c, 1024 c + 1024) of problem b's [x | r] (four per thread, 256 apart, loads in front of the stores,
// non-temporal); the workgroups c = 0 also form dlambda (one workgroup per problem: the norm is summed in a fixed order).
// (Round 5: one workgroup per problem walking all three vectors ran at 0.63 of the HBM rate.)
constexpr int TP_UNROLL = 4;
__global__ void __launch_bounds__(256) trial_point_kernel(const DevJt J, const double* __restrict__ x, const double* __restrict__ r,
                                                          const double* __restrict__ lambda, const double* __restrict__ d,
                                                          double max_dlambda, double* __restrict__ xt, double* __restrict__ rt,
                                                          double* __restrict__ lambdat, double* __restrict__ dlambda, int batch) {
  const int t = threadIdx.x;
  __shared__ double part[4];
  // one grid dimension (a second one ends at 65 535 problems): workgroup = (problem, chunk)
  const int nch_ = (J.nvar + J.nequ + 256 * TP_UNROLL - 1) / (256 * TP_UNROLL) > 0 ? (J.nvar + J.nequ + 256 * TP_UNROLL - 1) / (256 * TP_UNROLL) : 1;
  const long long b = blockIdx.x / nch_;
  const int chunk = blockIdx.x % nch_;
  const double* db = d + b * J.N;
  {
    const int nxr = J.nvar + J.nequ;   // d[0 .. nvar + nequ) = [dx | dr] lines up with [x | r]
    double a[TP_UNROLL], c[TP_UNROLL];
#pragma unroll
    for (int j = 0; j < TP_UNROLL; j++) {
      const int k = chunk * (256 * TP_UNROLL) + j * 256 + t;
      a[j] = 0.0; c[j] = 0.0;
      if (k < nxr) {
        a[j] = __builtin_nontemporal_load(k < J.nvar ? x + b * J.nvar + k : r + b * J.nequ + (k - J.nvar));
        c[j] = __builtin_nontemporal_load(db + k);
      }
    }
#pragma unroll
    for (int j = 0; j < TP_UNROLL; j++) {
      const int k = chunk * (256 * TP_UNROLL) + j * 256 + t;
      if (k < nxr) __builtin_nontemporal_store(a[j] + c[j], k < J.nvar ? xt + b * J.nvar + k : rt + b * J.nequ + (k - J.nvar));
    }
  }
  if (chunk != 0) return;
  double ss = 0.0;
  for (int k = t; k < J.ncon; k += 256) { const double v = db[J.nvar + J.nequ + k]; ss += v * v; }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if ((t & 63) == 0) part[t >> 6] = ss;
  __syncthreads();
  const double nrm = sqrt(part[0] + part[1] + part[2] + part[3]);
  for (int k = t; k < J.ncon; k += 256) {
    double dl = -db[J.nvar + J.nequ + k];
    if (nrm > max_dlambda) dl = dl * max_dlambda / nrm;  // same operation order as dλ .= dλ .* Mdλ ./ norm(dλ)
    dlambda[b * J.ncon + k] = dl;
    lambdat[b * J.ncon + k] = lambda[b * J.ncon + k] + dl;
  }
}

// prepare_newton_system! (/root/reference/src/CaNNOLeS.jl:947-981) for a batch, on the device: copies of the model's
// value arrays into the segments of `vals` ([H_F | H_c | J_F | J_c | -I | -delta I | rho I], :256-315):
//   H_F <- hF (left alone when hF == nullptr: Gauss-Newton variants, update_newton_hessian! no-op, hessian_approx.jl:46)
//   H_c <- -hc (:971-972), J_F <- Jx (:968-969), J_c <- Jcx (:973-974), -delta I <- -delta[b] (:975-976), rho I <- 0 (:978-979);
//   the -I segment is never written (:306).
constexpr int PREP_UNROLL = 4;
__global__ void __launch_bounds__(256) prepare_kernel(int nnzhF, int nnzhc, int nnzjF, int nnzjc, int nvar, int nequ, int ncon,
                                                      const double* __restrict__ hF, const double* __restrict__ hc,
                                                      const double* __restrict__ Jx, const double* __restrict__ Jcx,
                                                      const double* __restrict__ delta, double* __restrict__ vals, int batch) {
  // four slots per thread, 256 apart (coalesced 8-byte accesses), every load issued before the first store; the data are touched
  // once: non-temporal.  (One slot per thread: 0.62 of the HBM rate on 1.68 MB per system, bound by the number of workgroups.)
  const int o1 = nnzhF, o2 = o1 + nnzhc, o3 = o2 + nnzjF, o4 = o3 + nnzjc, o5 = o4 + nequ, o6 = o5 + ncon, nnz = o6 + nvar;
  // one grid dimension (a second one ends at 65 535 problems): workgroup = (problem, chunk of 1 024 slots)
  const int nch_ = (nnz + 256 * PREP_UNROLL - 1) / (256 * PREP_UNROLL);
  const long long b = blockIdx.x / nch_;
  const int chunk = blockIdx.x % nch_;
  double* v = vals + b * nnz;
  double x[PREP_UNROLL];
  bool st[PREP_UNROLL];
  // one predicated load per slot — source, validity and sign are SELECTED (round 6: loads inside the branches of an if-chain were
  // waited for one by one)
#pragma unroll
  for (int j = 0; j < PREP_UNROLL; j++) {
    const int k = chunk * (256 * PREP_UNROLL) + j * 256 + threadIdx.x;
    const int seg = k < o1 ? 0 : k < o2 ? 1 : k < o3 ? 2 : k < o4 ? 3 : k < o5 ? 4 : k < o6 ? 5 : 6;
    const double* sp = seg == 0 ? hF + b * nnzhF + k : seg == 1 ? hc + b * nnzhc + (k - o1) : seg == 2 ? Jx + b * nnzjF + (k - o2)
                     : seg == 3 ? Jcx + b * nnzjc + (k - o3) : delta + b;
    const bool rd = k < nnz && (seg == 0 ? hF != nullptr : seg == 1 || seg == 3 ? ncon > 0 : seg == 2 || seg == 5);
    const double y = rd ? __builtin_nontemporal_load(sp) : 0.0;
    x[j] = (seg == 1 || seg == 5) ? -y : y;
    st[j] = rd || (seg == 6 && k < nnz);   // (rho I <- 0; -I and the segments without a source are left alone)
  }
#pragma unroll
  for (int j = 0; j < PREP_UNROLL; j++) {
    const int k = chunk * (256 * PREP_UNROLL) + j * 256 + threadIdx.x;
    if (st[j]) __builtin_nontemporal_store(x[j], v + k);
  }
}

// ---- cnl_options.batch_layout = CNL_LAYOUT_INTERLEAVED (band.h: band_il_index) ----------------------------------------------------
// A workgroup moves a tile of 32 problems x 128 elements through LDS: on the problem-major side a problem's 128 elements are 1 KB
// contiguous, on the interleaved side the tile is 16 blocks x (32 problems x 8 doubles) = 32 KB contiguous.
constexpr int IL_TILE = 128, IL_PITCH = IL_TILE + 8;   // (pitch: the 4 x 8 lanes of a half-wavefront hit 32 different bank pairs)
__device__ __forceinline__ long long il_tile_dst(long long g, long long nb, int chunk, int idx, int& p, int& kk) {
  const int bl = idx >> 8, j = idx & 7;
  p = (idx >> 3) & 31;
  kk = bl * 8 + j;
  return ((g * nb + (long long)chunk * (IL_TILE / 8) + bl) * BAND_IL_GROUP + p) * 8 + j;
}
__global__ void __launch_bounds__(256) interleave_kernel(const double* __restrict__ src, double* __restrict__ dst, int batch, long long len,
                                                         int to_interleaved) {
  __shared__ double tile[BAND_IL_GROUP * IL_PITCH];
  const long long g = blockIdx.y, nb = band_il_blocks(len);
  const int chunk = blockIdx.x, t = threadIdx.x;
  if (to_interleaved) {
    double x[BAND_IL_GROUP / 2];   // (every load before the first LDS store)
    const int kk = t & (IL_TILE - 1);
    const long long k = (long long)chunk * IL_TILE + kk;
#pragma unroll
    for (int it = 0; it < BAND_IL_GROUP / 2; it++) {
      const long long prob = g * BAND_IL_GROUP + 2 * it + (t >> 7);
      x[it] = (prob < batch && k < len) ? __builtin_nontemporal_load(src + prob * len + k) : 0.0;
    }
#pragma unroll
    for (int it = 0; it < BAND_IL_GROUP / 2; it++) tile[(2 * it + (t >> 7)) * IL_PITCH + kk] = x[it];
    __syncthreads();
    for (int i = 0; i < BAND_IL_GROUP * IL_TILE / 256; i++) {
      int p, kk;
      const long long o = il_tile_dst(g, nb, chunk, i * 256 + t, p, kk);
      if ((long long)chunk * (IL_TILE / 8) + kk / 8 < nb) __builtin_nontemporal_store(tile[p * IL_PITCH + kk], dst + o);   // (pads and the spare block: zeros)
    }
  } else {
    for (int i = 0; i < BAND_IL_GROUP * IL_TILE / 256; i++) {
      int p, kk;
      const long long o = il_tile_dst(g, nb, chunk, i * 256 + t, p, kk);
      tile[p * IL_PITCH + kk] = ((long long)chunk * (IL_TILE / 8) + kk / 8 < nb) ? __builtin_nontemporal_load(src + o) : 0.0;
    }
    __syncthreads();
    for (int it = 0; it < BAND_IL_GROUP / 2; it++) {
      const int pi = 2 * it + (t >> 7), kk = t & (IL_TILE - 1);
      const long long prob = g * BAND_IL_GROUP + pi, k = (long long)chunk * IL_TILE + kk;
      if (prob < batch && k < len) __builtin_nontemporal_store(tile[pi * IL_PITCH + kk], dst + prob * len + k);
    }
  }
}

hipError_t launch_interleave(const double* src, double* dst, int batch, long long len, int to_interleaved, hipStream_t stream) {
  const long long groups = (batch + BAND_IL_GROUP - 1) / BAND_IL_GROUP, chunks = (band_il_blocks(len) * 8 + IL_TILE - 1) / IL_TILE;
  if (batch <= 0 || len <= 0 || groups > 65535 || chunks > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)chunks, (unsigned)groups), dim3(256), 0, stream, src, dst, batch, len, to_interleaved);
  return hipGetLastError();
}

// prepare_newton_system! writing `vals` interleaved (the values and the "left alone" rules of prepare_kernel above, the tiles of
// interleave_kernel): the model's arrays are read 1 KB per problem, `vals` is written 32 KB contiguous per workgroup
__global__ void __launch_bounds__(256) prepare_il_kernel(int nnzhF, int nnzhc, int nnzjF, int nnzjc, int nvar, int nequ, int ncon,
                                                         const double* __restrict__ hF, const double* __restrict__ hc,
                                                         const double* __restrict__ Jx, const double* __restrict__ Jcx,
                                                         const double* __restrict__ delta, double* __restrict__ vals, int batch) {
  __shared__ double tile[BAND_IL_GROUP * IL_PITCH];
  const int o1 = nnzhF, o2 = o1 + nnzhc, o3 = o2 + nnzjF, o4 = o3 + nnzjc, o5 = o4 + nequ, o6 = o5 + ncon, nnz = o6 + nvar;
  const long long g = blockIdx.y, nb = band_il_blocks(nnz);
  const int chunk = blockIdx.x, t = threadIdx.x;
  // every load is issued before the first LDS store (one predicated load per element: source pointer, validity and sign are selected,
  // not branched on — loads inside the branches of prepare_kernel's chain were waited for one by one: 0.51 of the HBM rate against 0.65)
  double x[BAND_IL_GROUP / 2];
  const int kk0 = t & (IL_TILE - 1), k0 = chunk * IL_TILE + kk0;
  const int seg = k0 < o1 ? 0 : k0 < o2 ? 1 : k0 < o3 ? 2 : k0 < o4 ? 3 : k0 < o5 ? 4 : k0 < o6 ? 5 : 6;
  const double* sbase = seg == 0 ? hF : seg == 1 ? hc : seg == 2 ? Jx : seg == 3 ? Jcx : delta;
  const long long sstr = seg == 0 ? nnzhF : seg == 1 ? nnzhc : seg == 2 ? nnzjF : seg == 3 ? nnzjc : 1;
  const int soff = seg == 0 ? k0 : seg == 1 ? k0 - o1 : seg == 2 ? k0 - o2 : seg == 3 ? k0 - o3 : 0;
  const bool rd = k0 < nnz && (seg == 0 ? hF != nullptr : seg == 1 || seg == 3 ? ncon > 0 : seg == 2 || seg == 5);
  const bool neg = seg == 1 || seg == 5;
#pragma unroll
  for (int it = 0; it < BAND_IL_GROUP / 2; it++) {
    const long long b = g * BAND_IL_GROUP + 2 * it + (t >> 7);
    x[it] = (rd && b < batch) ? __builtin_nontemporal_load(sbase + b * sstr + soff) : 0.0;
  }
#pragma unroll
  for (int it = 0; it < BAND_IL_GROUP / 2; it++) tile[(2 * it + (t >> 7)) * IL_PITCH + kk0] = neg ? -x[it] : x[it];
  __syncthreads();
  for (int i = 0; i < BAND_IL_GROUP * IL_TILE / 256; i++) {
    int p, kk;
    const long long o = il_tile_dst(g, nb, chunk, i * 256 + t, p, kk);
    const int k = chunk * IL_TILE + kk;
    // the slots prepare_newton_system! leaves alone: H_F without a Hessian, the constraint segments without constraints, -I; and the pads
    const bool wr = k < o1 ? hF != nullptr : k < o2 ? ncon > 0 : k < o3 ? true : k < o4 ? ncon > 0 : k < o5 ? false : k < nnz;
    if (wr && g * BAND_IL_GROUP + p < batch) __builtin_nontemporal_store(tile[p * IL_PITCH + kk], vals + o);
  }
}

// Row f4 of the scope table: least-squares multiplier estimate  min || Jc' lambda - Jx' r ||  by CGLS, as the reference
// obtains it from Krylov.jl (`krylov_solve!(cgls_workspace, Jcx', Jxtr)`, /root/reference/src/CaNNOLeS.jl:507-518 and
// :880-882; Krylov.jl is an un-vendored dependency, Project.toml compat "0.10": its cgls is the textbook recurrence
//   r = b, s = A'r, p = s, gamma = ||s||^2;  q = A p, alpha = gamma / ||q||^2, x += alpha p, r -= alpha q,
//   s = A'r, beta = ||s||^2 / gamma, p = s + beta p;   stop when ||s|| <= atol + rtol ||s_0||  or after itmax steps
// with A = Jc' (nvar x ncon) and b = Jx' r).  One workgroup per problem; the ncon-vectors live in LDS, the two
// nvar-vectors (residual, q) in a global workspace; reductions in a fixed order (deterministic).
constexpr int CGLS_PMAX = 1024;
__device__ __forceinline__ double block_sum(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) cgls_kernel(const DevJt Jin, const JacSrc S, const double* __restrict__ r,
                                                   double* __restrict__ lambda, double* __restrict__ Jxtr, double* __restrict__ ws,
                                                   int* __restrict__ iters, double atol, double rtol, int itmax, int ones_if_zero) {
  DevJt J = Jin;
  J.ptrF = as_global(Jin.ptrF); J.slotF = as_global(Jin.slotF); J.idxF = as_global(Jin.idxF);
  J.ptrC = as_global(Jin.ptrC); J.slotC = as_global(Jin.slotC); J.idxC = as_global(Jin.idxC);
  J.rptrC = as_global(Jin.rptrC); J.rslotC = as_global(Jin.rslotC); J.rcolC = as_global(Jin.rcolC);
  __shared__ double x[CGLS_PMAX], sv[CGLS_PMAX], pv[CGLS_PMAX];
  __shared__ double red[4];
  const long long b = blockIdx.x;
  const int t = threadIdx.x, n = J.nvar, p = J.ncon;
  const double* __restrict__ vF = S.vF + b * S.sF;
  const double* __restrict__ vC = S.vC + b * S.sC;
  const double* rb = r + b * J.nequ;
  double* res = ws + b * 2 * n;  // residual of the least-squares problem
  double* q = res + n;
  // b = Jx' r  (mul!(Jxtr, Jx', r), per-column COO-order sums as in residual_vectors_kernel)
  {
#pragma clang fp contract(off)  // separately rounded multiply and add, as the reference's scalar loop
    for (int j = t; j < n; j += 256) {
      double s1 = 0.0;
      for (int k = J.ptrF[j]; k < J.ptrF[j + 1]; k++) { const double t_ = vF[J.slotF[k]] * rb[J.idxF[k]]; s1 = s1 + t_; }
      res[j] = s1;
      if (Jxtr) Jxtr[b * n + j] = s1;
    }
  }
  for (int k = t; k < p; k += 256) x[k] = 0.0;
  __syncthreads();
  // s = A' res = Jc res: one wavefront per constraint row
  auto at_res = [&]() {
    for (int k = t >> 6; k < p; k += 4) {
      double acc = 0.0;
      for (int e = J.rptrC[k] + (t & 63); e < J.rptrC[k + 1]; e += 64) acc += vC[J.rslotC[e]] * res[J.rcolC[e]];
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if ((t & 63) == 0) sv[k] = acc;
    }
    __syncthreads();
  };
  at_res();
  for (int k = t; k < p; k += 256) pv[k] = sv[k];
  __syncthreads();
  double part = 0.0;
  for (int k = t; k < p; k += 256) part += sv[k] * sv[k];
  double gamma = block_sum(part, red);
  const double eps = atol + rtol * sqrt(gamma);
  int it = 0;
  while (it < itmax && sqrt(gamma) > eps) {
    // q = A p = Jc' p: thread per variable
    part = 0.0;
    for (int j = t; j < n; j += 256) {
      double s2 = 0.0;
      for (int k = J.ptrC[j]; k < J.ptrC[j + 1]; k++) s2 += vC[J.slotC[k]] * pv[J.idxC[k]];
      q[j] = s2;
      part += s2 * s2;
    }
    const double delta = block_sum(part, red);
    if (delta == 0.0) break;
    const double alpha = gamma / delta;
    for (int k = t; k < p; k += 256) x[k] += alpha * pv[k];
    for (int j = t; j < n; j += 256) res[j] -= alpha * q[j];
    __syncthreads();
    at_res();
    part = 0.0;
    for (int k = t; k < p; k += 256) part += sv[k] * sv[k];
    const double gnext = block_sum(part, red);
    const double beta = gnext / gamma;
    for (int k = t; k < p; k += 256) pv[k] = sv[k] + beta * pv[k];
    __syncthreads();
    gamma = gnext;
    it++;
  }
  // if norm(lambda) == 0: lambda .= 1   (src/CaNNOLeS.jl:515-517)
  part = 0.0;
  for (int k = t; k < p; k += 256) part += x[k] * x[k];
  const double xn = block_sum(part, red);
  for (int k = t; k < p; k += 256) lambda[b * p + k] = (ones_if_zero && xn == 0.0) ? 1.0 : x[k];
  if (t == 0 && iters) iters[b] = it;
}

hipError_t launch_cgls(const DevJt& J, const JacSrc& S, const double* r, double* lambda, double* Jxtr, double* ws, int32_t* iters,
                       double atol, double rtol, int itmax, int ones_if_zero, int batch, hipStream_t stream) {
  if (J.ncon > CGLS_PMAX) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cgls_kernel, dim3(batch), dim3(256), 0, stream, J, S, r, lambda, Jxtr, ws, iters, atol, rtol, itmax, ones_if_zero);
  return hipGetLastError();
}

// one linear grid dimension (problem, chunk): the product must fit a 31-bit grid (a larger one would truncate silently)
static inline bool linear_grid_ok(long long nch, long long batch) { return nch > 0 && batch > 0 && nch * batch <= 0x7fffffffLL; }

hipError_t launch_prepare(int nnzhF, int nnzhc, int nnzjF, int nnzjc, int nvar, int nequ, int ncon, const double* hF, const double* hc,
                          const double* Jx, const double* Jcx, const double* delta, double* vals, int batch, int interleaved, hipStream_t stream) {
  const int nnz = nnzhF + nnzhc + nnzjF + nnzjc + nequ + ncon + nvar;
  if (interleaved) {
    const long long groups = (batch + BAND_IL_GROUP - 1) / BAND_IL_GROUP, chunks = (nnz + IL_TILE - 1) / IL_TILE;
    if (batch <= 0 || nnz <= 0 || groups > 65535) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL(prepare_il_kernel, dim3((unsigned)chunks, (unsigned)groups), dim3(256), 0, stream, nnzhF, nnzhc, nnzjF, nnzjc, nvar, nequ,
                       ncon, hF, hc, Jx, Jcx, delta, vals, batch);
    return hipGetLastError();
  }
  const long long nch = (nnz + 256 * PREP_UNROLL - 1) / (256 * PREP_UNROLL);   // (0 when nnz == 0: the kernel divides by it)
  if (!linear_grid_ok(nch, batch)) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(prepare_kernel, dim3((unsigned)(nch * batch)), dim3(256), 0, stream, nnzhF, nnzhc, nnzjF, nnzjc, nvar, nequ, ncon,
                     hF, hc, Jx, Jcx, delta, vals, batch);
  return hipGetLastError();
}

hipError_t launch_residual_vectors(const DevJt& J, const JacSrc& S, const double* r, const double* lambda, const double* Fx,
                                   const double* cx, double* rhs, double* norms, int batch, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(norms, 0, sizeof(double) * 2 * (size_t)batch, stream);
  if (e != hipSuccess) return e;
  // problems per workgroup of the tiled kernel: the table of a tile is read once per workgroup, and a workgroup streams one problem
  // ahead (measured at 8192 systems of cfg3's pattern: 1 -> 1.40 ms, 2 -> 1.30, 4 -> 1.27, 16 -> 1.32)
  const int pb = J.rv_ntiles > 0 ? (batch >= 512 ? 4 : batch >= 64 ? 2 : 1) : RPT;
  // the second grid dimension holds at most 65535 groups of problems: larger batches go in slices of that many groups, every
  // per-problem pointer advanced to the slice's first problem (row strides: the Jacobian sources', nequ, ncon, N, 2)
  const long long slice = 65535LL * pb;
  for (long long b0 = 0; b0 < batch; b0 += slice) {
    const int nb = (int)std::min<long long>(slice, batch - b0);
    const JacSrc v_{S.vF + b0 * S.sF, S.sF, S.vC + b0 * S.sC, S.sC, S.safeF, S.safeC};
    const double* r_ = r + b0 * J.nequ;
    const double* l_ = lambda ? lambda + b0 * J.ncon : nullptr;
    const double* f_ = Fx + b0 * J.nequ;
    const double* c_ = cx ? cx + b0 * J.ncon : nullptr;
    double* o_ = rhs + b0 * J.N;
    double* n_ = norms + 2 * b0;
    if (J.rv_ntiles > 0)
      hipLaunchKernelGGL(residual_vectors_tiled_kernel, dim3(J.rv_ntiles + J.rv_primal_tiles, (nb + pb - 1) / pb), dim3(256),
                         (size_t)J.rv_lds_doubles * sizeof(double), stream, J, v_, r_, l_, f_, c_, o_, n_, nb, pb);
    else
      hipLaunchKernelGGL(residual_vectors_kernel, dim3((J.N + 255) / 256, (nb + RPT - 1) / RPT), dim3(256), 0, stream, J, v_, r_, l_, f_, c_, o_, n_, nb);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_trial_point(const DevJt& J, const double* x, const double* r, const double* lambda, const double* d,
                              double max_dlambda, double* xt, double* rt, double* lambdat, double* dlambda, int batch,
                              hipStream_t stream) {
  const int nch = std::max(1, (J.nvar + J.nequ + 256 * TP_UNROLL - 1) / (256 * TP_UNROLL));
  if (!linear_grid_ok(nch, batch)) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(trial_point_kernel, dim3((unsigned)((long long)nch * batch)), dim3(256), 0, stream, J, x, r, lambda, d, max_dlambda, xt, rt, lambdat, dlambda,
                     batch);
  return hipGetLastError();
}

hipError_t launch_condense(const DevCond& C, const double* vals, const double* rhs, double* cbuf, int slot_begin, int slot_end,
                           int batch, hipStream_t stream) {
  const int n = slot_end - slot_begin;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(condense_kernel, dim3((n + 255) / 256, (batch + CPB - 1) / CPB), dim3(256), 0, stream, C, vals, rhs, cbuf, slot_begin, slot_end, batch);
  return hipGetLastError();
}

hipError_t launch_condense_tiled(const DevCond& C, const double* vals, const double* rhs, double* cbuf, int mask, int nchunks,
                                 int batch, hipStream_t stream) {
  if (nchunks <= 0) return hipSuccess;
  const size_t lds = (size_t)TPB * (size_t)C.tile_max * sizeof(double) + (size_t)C.chunk_ncon_max * 8 + ((size_t)C.chunk_nslot_max + 8) * 4;
  {  // per device and cheap: set on every launch (a process may drive several devices from several threads)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(condense_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attr_cap((int)lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(condense_tiled_kernel, dim3(nchunks, (batch + TPB - 1) / TPB), dim3(256), lds, stream, C, vals, rhs, cbuf, mask, batch);
  return hipGetLastError();
}

hipError_t launch_cond_inertia(const DevCond& C, const double* vals, int* extra_pos, int* extra_zer, double eig_tol, int batch,
                               hipStream_t stream) {
  hipLaunchKernelGGL(cond_inertia_kernel, dim3(batch), dim3(256), 0, stream, C, vals, extra_pos, extra_zer, eig_tol, batch);
  return hipGetLastError();
}

hipError_t launch_expand(const DevCond& C, double* vals, const double* rhs, const double* d2, const double* cbuf, double* dout,
                         const int* success, int copy_rho_tail, int batch, hipStream_t stream) {
  const int nb_copy = d2 ? ((int)C.N2 + 255) / 256 : 0, nb_r = ((int)C.ncond + XRB - 1) / XRB;
  hipLaunchKernelGGL(expand_kernel, dim3(nb_copy + nb_r, (batch + XPB - 1) / XPB), dim3(256), 0, stream, C, vals, rhs, d2, cbuf, dout, success,
                     copy_rho_tail, nb_copy, batch);
  return hipGetLastError();
}

// rho slots of the problems marked active: vals[b][nnz - nvar ..] = rho[b] (the host-driven ladder of the small-batch host call)
__global__ void __launch_bounds__(256) fill_rho_kernel(double* __restrict__ vals, long long nnz, int nvar, const double* __restrict__ rho,
                                                       const int* __restrict__ active, int batch) {
  for (int b = blockIdx.y; b < batch; b += gridDim.y) {  // (gridDim.y is clamped to the launch limit of 65535)
    if (!active[b]) continue;
    const double r = rho[b];
    double* t = vals + (long long)b * nnz + (nnz - nvar);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvar; i += gridDim.x * 256) t[i] = r;
  }
}
// debugging aid (capi.cpp, env CNL_DBG_LDSFILL): leaves a byte pattern in the LDS of every CU, so that a kernel that reads LDS it
// has not written — whatever the previous kernel left there — fails reproducibly instead of depending on the process's history
__global__ void __launch_bounds__(256) lds_fill_kernel(int pattern, int* sink) {
  extern __shared__ int lds_fill_buf[];
  const int n = 40 * 1024 / 4;
  for (int i = threadIdx.x; i < n; i += 256) lds_fill_buf[i] = pattern;
  __syncthreads();
  if (sink && lds_fill_buf[(threadIdx.x * 37) % n] == 12345) sink[0] = 1;
}
// the same for private (scratch) memory: spilled registers and out-of-line frames of the kernels that use scratch start from
// whatever the previous scratch user left
__global__ void __launch_bounds__(256) scratch_fill_kernel(int pattern, int n, int* sink) {
  volatile int a[640];
  for (int i = 0; i < 640; i++) a[i] = pattern;
  int acc = 0;
  for (int i = 0; i < n; i++) acc += a[(i * 37 + threadIdx.x) % 640];
  if (sink && acc == 12345) sink[0] = acc;
}
// ... and for the vector registers (a wavefront starts with whatever the previous owner of its registers left in them)
__global__ void __launch_bounds__(256) vgpr_fill_kernel(int pattern) {
  asm volatile(".irp r,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223,224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255\n v_mov_b32 v\\r, %0\n .endr" :: "v"(pattern) : "memory", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
hipError_t launch_lds_fill(int pattern, hipStream_t stream) {
  if (getenv("CNL_DBG_SCRATCHFILL")) {
    hipLaunchKernelGGL(scratch_fill_kernel, dim3(256 * 8), dim3(256), 0, stream, pattern, 0, (int*)nullptr);
    hipLaunchKernelGGL(vgpr_fill_kernel, dim3(256 * 8), dim3(256), 0, stream, pattern);
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lds_fill_kernel, dim3(256 * 16), dim3(256), 40 * 1024, stream, pattern, (int*)nullptr);   // four workgroups of 40 KB per CU, several rounds
  return hipGetLastError();
}

hipError_t launch_fill_rho(double* vals, long long nnz, int nvar, const double* rho, const int* active, int batch, hipStream_t stream) {
  if (nvar <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_rho_kernel, dim3(std::min(64, (nvar + 255) / 256), std::min(batch, 65535)), dim3(256), 0, stream, vals, nnz, nvar, rho, active, batch);
  return hipGetLastError();
}

}  // namespace cnl
