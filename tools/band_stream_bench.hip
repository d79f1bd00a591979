// band_stream_bench.hip — can a "lane = problem" band sweep be fed from problem-major arrays at the HBM rate?
// Skeleton of the band kernel's data movement (round 5), no real arithmetic: a workgroup of two wavefronts serves NL problems; wave w is
// PART w of every problem's chain (part 1 walks its half backwards).  Per epoch of 8 pivot steps every input stream is loaded
// coalesced (8 lanes x 8 bytes = one 64-byte piece per problem and instruction), kept in registers for one epoch (the registers are the
// look-ahead buffer), written to the problem's LDS ring, and consumed by the compute lanes (lane = problem) with ds_read_b64; the factor
// rows (7 doubles per step) go through an LDS ring back to 64-byte pieces.  Stream rates are those of BASELINE config 3 (per step:
// J_F 5, H_F 3, six single streams).  FL = dependent FMAs per step standing in for the elimination.
// Build: hipcc -O3 --offload-arch=gfx950 -o band_stream_bench band_stream_bench.hip ; prints one JSON line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int NS = 8;                       // input streams
__device__ constexpr int RATE[NS] = {5, 3, 1, 1, 1, 1, 1, 1};
__device__ constexpr int ROWS[NS] = {5, 3, 1, 1, 1, 1, 1, 1};   // 8-double pieces per epoch and stream
constexpr int NROWS = 14;
constexpr int OUTR = 7;                     // factor doubles per step
__device__ constexpr int ISRHS[NS] = {0, 0, 0, 0, 0, 0, 1, 1};

struct Args {
  const double* vals; const double* rhs; double* L; double* sink;
  long long vstride, rstride, lstride;
  int seg[NS];      // first double of the stream inside vals / rhs
  int len[NS];      // doubles of the stream
  int nsteps;       // steps per part
  int batch;
  int flags;   // 1: no loads after the first epoch, 2: no stores, 4: INTERLEAVED layout (blocks of eight doubles of the workgroup's problems side by side: every instruction moves 512 contiguous bytes)
};

template <int NL, int FL, int E, int DEPTH, int PAIR, int W16 = 0>
__global__ void __launch_bounds__(128, 1) stream_kernel(const Args A) {
  constexpr int NI = W16 ? NL / 16 : NL / 8;
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int RING = (5 + 3 + 6) * E;   // one epoch of every stream
  constexpr int EB = E / 8;
  constexpr int TOT = (RING + E * OUTR) | 1;   // per-problem LDS block (doubles), odd
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int lq = W16 ? lane >> 2 : lane >> 3, le = W16 ? 2 * (lane & 3) : lane & 7;
  constexpr int PPI = W16 ? 16 : 8;   // problems per instruction
  const int prob0 = blockIdx.x * NL;
  double* blk = lds + (size_t)part * NL * TOT;
  const double* vals = A.vals;
  const double* rhs = A.rhs;
  // per-lane element offsets of the loader (problem lq of every group of eight, element le)
  long long voff[NI], roff[NI], loff[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    int p = prob0 + i * PPI + lq;
    if (p >= A.batch) p = A.batch - 1;
    voff[i] = (long long)p * A.vstride + le;
    roff[i] = (long long)p * A.rstride + le;
    loff[i] = (long long)p * A.lstride + le;
  }
  double st[NROWS * EB][NI], st2[NROWS * EB][NI], stb[NROWS * EB][NI];   // stb: second double of a 16-byte load
  double sp[6][NI];   // PAIR: second halves of the single-rate streams' 128-byte loads
  auto issue = [&](int ep, double (&st)[NROWS * EB][NI]) {
    int r = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int g = ROWS[s] * E;
      long long pos = part == 0 ? (long long)ep * g : (long long)A.len[s] - (long long)(ep + 1) * g;
      if (pos < 0) pos = 0;
      if (pos + g > A.len[s]) pos = A.len[s] - g;
      const double* base = (ISRHS[s] ? rhs : vals) + A.seg[s] + pos;
#pragma unroll
      for (int c = 0; c < ROWS[s] * EB; c++) {
        if (PAIR && s >= 2) {
          // even epochs: both 64-byte halves of the line (this epoch's and the next one's); odd epochs: the half kept in registers
          if ((ep & 1) == 0) {
#pragma unroll
            for (int i = 0; i < NI; i++) { st[r][i] = base[(ISRHS[s] ? roff[i] : voff[i])]; sp[s - 2][i] = base[(ISRHS[s] ? roff[i] : voff[i]) + 8]; }
          } else {
#pragma unroll
            for (int i = 0; i < NI; i++) st[r][i] = sp[s - 2][i];
          }
        } else if (W16) {
#pragma unroll
          for (int i = 0; i < NI; i++) { const d2 v = *reinterpret_cast<const d2*>(base + (ISRHS[s] ? roff[i] : voff[i]) + c * 8); st[r][i] = v.x; stb[r][i] = v.y; }
        } else if (A.flags & 4) {
          const long long e0 = (long long)A.seg[s] + pos + c * 8 + le;          // element of the lane inside a problem's array
          const double* wb = (ISRHS[s] ? rhs + (long long)prob0 * A.rstride : vals + (long long)prob0 * A.vstride);
#pragma unroll
          for (int i = 0; i < NI; i++) st[r][i] = wb[((e0 >> 3) * NL + (i * PPI + lq)) * 8 + (e0 & 7)];
        } else {
#pragma unroll
          for (int i = 0; i < NI; i++) st[r][i] = base[(ISRHS[s] ? roff[i] : voff[i]) + c * 8];
        }
        r++;
      }
    }
  };
  auto commit = [&](double (&st)[NROWS * EB][NI]) {
#pragma unroll
    for (int r = 0; r < NROWS * EB; r++)
#pragma unroll
      for (int i = 0; i < NI; i++) { blk[(i * PPI + lq) * TOT + r * 8 + le] = st[r][i]; if (W16) blk[(i * PPI + lq) * TOT + r * 8 + le + 1] = stb[r][i]; }
  };
  const int nep = A.nsteps / E;
  issue(0, st);
  if (DEPTH == 2) issue(1, st2);
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 1.0 + k;
  double* outbase = A.L + (long long)part * (A.lstride / 2);
  auto epoch = [&](int ep) {
    // compute lanes: lane = problem
    if (lane < NL) {
      const double* my = blk + lane * TOT;
      double* myout = blk + lane * TOT + RING;
#pragma unroll 2
      for (int t = 0; t < E; t++) {
        double v[14];
#pragma unroll
        for (int k = 0; k < 5; k++) v[k] = my[t * 5 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) v[5 + k] = my[5 * E + t * 3 + k];
#pragma unroll
        for (int k = 0; k < 6; k++) v[8 + k] = my[8 * E + k * E + t];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = fma(acc[k], 0.5, v[k] + v[(k + 8) % 14]);
#pragma unroll
        for (int f = 0; f < FL; f++) acc[f & 7] = fma(acc[(f + 1) & 7], 0.999, acc[f & 7]);
#pragma unroll
        for (int k = 0; k < OUTR; k++) myout[t * OUTR + k] = acc[k];
      }
    }
    // factor rows of the epoch: 56 doubles per problem = 7 pieces
#pragma unroll
    for (int c = 0; c < OUTR * EB; c++)
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const double x = blk[(i * PPI + lq) * TOT + RING + c * 8 + le];
        if (prob0 + i * PPI + lq < A.batch && !(A.flags & 2)) {
          if (W16) { d2 v; v.x = x; v.y = blk[(i * PPI + lq) * TOT + RING + c * 8 + le + 1]; *reinterpret_cast<d2*>(outbase + loff[i] + (long long)ep * (E * OUTR) + c * 8) = v; }
          else if (A.flags & 4) {
            const long long e0 = (long long)part * (A.lstride / 2) + (long long)ep * (E * OUTR) + c * 8 + le;
            (A.L + (long long)prob0 * A.lstride)[((e0 >> 3) * NL + (i * PPI + lq)) * 8 + (e0 & 7)] = x;
          } else outbase[loff[i] + (long long)ep * (E * OUTR) + c * 8] = x;
        }
      }
  };
  if (DEPTH == 1) {
    for (int ep = 0; ep < nep; ep++) {
      commit(st);
      if (ep + 1 < nep && !(A.flags & 1)) issue(ep + 1, st);
      epoch(ep);
    }
  } else {
    for (int ep = 0; ep < nep; ep += 2) {
      commit(st);
      if (ep + 2 < nep) issue(ep + 2, st);
      epoch(ep);
      commit(st2);
      if (ep + 3 < nep) issue(ep + 3, st2);
      epoch(ep + 1);
    }
  }
  if (acc[0] == 123.456) A.sink[blockIdx.x] = acc[1];
}

template <int NL, int FL, int E, int DEPTH, int PAIR, int W16 = 0>
int run(const char* name, Args A) {
  constexpr int RING = (5 + 3 + 6) * E;
  constexpr int TOT = (RING + E * OUTR) | 1;
  const size_t ldsb = (size_t)2 * NL * TOT * sizeof(double);
  auto kern = stream_kernel<NL, FL, E, DEPTH, PAIR, W16>;
  CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  const int grid = (A.batch + NL - 1) / NL;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128), ldsb, 0, A);
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(128), ldsb, 0, A);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double in_bytes = (double)A.batch * 2.0 * A.nsteps * 14.0 * 8.0, out_bytes = (double)A.batch * 2.0 * A.nsteps * 7.0 * 8.0;
  printf("{\"case\": \"%s\", \"NL\": %d, \"FL\": %d, \"E\": %d, \"depth\": %d, \"pair\": %d, \"w16\": %d, \"aligned\": %d, \"flags\": %d, \"batch\": %d, \"grid\": %d, \"lds_bytes\": %zu, \"ms\": %.3f, \"TBps\": %.2f, \"ns_per_step\": %.1f}\n", name, NL, FL, E, DEPTH, PAIR, W16, (int)(A.vstride % 8 == 0), A.flags, A.batch,
         grid, ldsb, best, (in_bytes + out_bytes) / (best * 1e-3) / 1e12, best * 1e6 / A.nsteps);
  return 0;
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 8192;
  Args A{};
  const int n = 10000;
  const int seglen[NS] = {49994, 29997, 10000, 10000, 10000, 10000, 10000, 10000};   // J_F, H_F, H_c, J_c, -I, rho, rhs_x, rhs_r
  const int segoff[NS] = {39997, 0, 29997, 89991, 99991, 110041, 0, 10000};
  for (int s = 0; s < NS; s++) { A.seg[s] = segoff[s]; A.len[s] = seglen[s]; }
  A.vstride = 120041; A.rstride = 20050; A.lstride = 2 * 5000 * 7 + 64; A.batch = batch; A.nsteps = 5000;
  (void)n;
  double *vals, *rhs, *L, *sink;
  CHK(hipMalloc(&vals, (size_t)2 * batch * 120064 * 8 + 4096));
  CHK(hipMalloc(&rhs, (size_t)2 * batch * 20056 * 8 + 4096));
  CHK(hipMalloc(&L, (size_t)2 * batch * A.lstride * 8 + 4096));
  CHK(hipMalloc(&sink, 1 << 20));
  CHK(hipMemset(vals, 0, (size_t)batch * A.vstride * 8));
  CHK(hipMemset(rhs, 0, (size_t)batch * A.rstride * 8));
  A.vals = vals; A.rhs = rhs; A.L = L; A.sink = sink;
  // what bounds the skeleton: every problem aliased onto problem 0 (cache hits), loads only, stores only
  for (int alias = 0; alias < 2; alias++) {
    if (alias) { A.vstride = 0; A.rstride = 0; }
    for (int fl : {0, 1, 2, 4, 6}) {
      if (alias && fl >= 4) continue;
      A.flags = fl;
      A.batch = 512 * 24;
      if (run<24, 60, 8, 1, 0, 0>(alias ? "two workgroups per CU, ALL PROBLEMS READ PROBLEM 0" : "two workgroups per CU", A)) return 1;
    }
  }
  return 0;
}
