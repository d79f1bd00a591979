// seg_bench.hip — what the memory system of an MI355X gives a kernel that reads SEGMENTS of S bytes from many far-apart places
// (round 5: the band kernels read 64-byte pieces, one per problem and stream; is the piece size the bound?).
// A wavefront serves NP = 512 / S problems per load instruction (S / 8 lanes x 8 bytes each; S = 1024: 16 bytes per lane, one
// problem); every problem has K streams (offsets inside its 960 KB block of `vals`), a stream advances by S bytes per round; U
// loads are in flight per wavefront.  One wavefront per SIMD (1024 wavefronts) or two.  Prints one JSON line per case.
// Build: hipcc -O3 --offload-arch=gfx950 -o seg_bench seg_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int S, int K, int W>   // segment bytes, streams per problem, bytes per lane (8 or 16)
__global__ void __launch_bounds__(64) seg_kernel(const double* __restrict__ base, long long pstride, long long sstride, int rounds, int nprob, double* sink) {
  constexpr int LPS = S / W;          // lanes per segment
  constexpr int NP = 64 / LPS;        // problems per instruction
  const int lane = threadIdx.x;
  const long long prob = ((long long)blockIdx.x * NP + lane / LPS) % nprob;
  const double* p = base + prob * pstride + (lane % LPS) * (W / 8);
  double acc = 0.0;
  for (int r = 0; r < rounds; r++) {
    double v[K], v2[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const double* q = p + k * sstride + (long long)r * (S / 8);
      if (W == 16) { typedef double d2 __attribute__((ext_vector_type(2), aligned(8))); const d2 t = *reinterpret_cast<const d2*>(q); v[k] = t.x; v2[k] = t.y; }
      else { v[k] = *q; v2[k] = 0.0; }
    }
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k] + v2[k];
  }
  if (acc == 123.456) sink[blockIdx.x] = acc;
}

template <int S, int K, int W>
int run(const double* base, long long pstride, int nprob, int waves, double* sink) {
  constexpr int NP = 64 / (S / W);
  const long long sstride = 15000;          // doubles between the streams of a problem (120 KB)
  const int rounds = (int)(sstride * 8 / S) / 2;   // half a stream
  const int grid = waves;                   // one wavefront per workgroup
  (void)nprob;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((seg_kernel<S, K, W>), dim3(grid), dim3(64), 0, 0, base, pstride, sstride, rounds, grid * NP, sink);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double bytes = (double)grid * 64.0 * W * K * rounds;
  printf("{\"segment_bytes\": %d, \"streams\": %d, \"bytes_per_lane\": %d, \"wavefronts\": %d, \"problems\": %d, \"rounds\": %d, \"ms\": %.3f, \"TBps\": %.2f}\n", S, K, W, grid, grid * NP, rounds, best,
         bytes / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  const long long pstride = 120041;   // doubles per problem (cfg3's vals)
  const size_t maxprob = 2048 * 8 + 8;
  double *base, *sink;
  CHK(hipMalloc(&base, maxprob * pstride * 8));
  CHK(hipMalloc(&sink, 1 << 20));
  CHK(hipMemset(base, 0, maxprob * pstride * 8));
  for (int waves : {1024, 2048}) {
    if (run<64, 8, 8>(base, pstride, 0, waves, sink)) return 1;
    if (run<64, 16, 8>(base, pstride, 0, waves, sink)) return 1;
    if (run<128, 8, 8>(base, pstride, 0, waves, sink)) return 1;
    if (run<128, 8, 16>(base, pstride, 0, waves, sink)) return 1;
    if (run<256, 8, 8>(base, pstride, 0, waves, sink)) return 1;
    if (run<256, 8, 16>(base, pstride, 0, waves, sink)) return 1;
    if (run<512, 8, 8>(base, pstride, 0, waves, sink)) return 1;
    if (run<512, 8, 16>(base, pstride, 0, waves, sink)) return 1;
    if (run<1024, 8, 16>(base, pstride, 0, waves, sink)) return 1;
  }
  return 0;
}
