// vmem_bench.hip — what does one vector-memory load instruction cost a CU on gfx950, in the shapes the register-front kernel issues?
// 2048 wavefronts (8 per CU, one per workgroup), each issues rounds of 8 independent loads and sums them.  A wavefront serves four
// "problems" (16 lanes each, regions `stride` bytes apart) like newton2_kernel; per load a problem reads 16 consecutive doubles.
//   mode 0: 8 bytes per lane, all 64 lanes        mode 1: 8 bytes per lane, lanes l < 10 of each 16 (the others masked off)
//   mode 2: 16 bytes per lane, all lanes           mode 3: 8 bytes per lane, the 64 lanes read 512 consecutive bytes (one region)
//   mode 4: 8 bytes per lane, lane stride 40 bytes (the row-form gathers: one operand of each of 16 rows of 5 entries)
// footprint: bytes each problem cycles through (small = L2 / L1 hits, large = HBM).
// Build: hipcc -O3 --offload-arch=gfx950 -o vmem_bench vmem_bench.hip ; prints one JSON line per case.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(64) loads(const char* base, long long stride, unsigned foot, int rounds, double* out) {
  const int lane = threadIdx.x, l = lane & 15, g = lane >> 4;
  const long long wave = blockIdx.x;
  const char* p = base + (wave * 4 + (MODE == 3 ? 0 : g)) * stride;
  unsigned off = MODE == 3 ? lane * 8u : (MODE == 2 ? l * 16u : (MODE == 4 ? l * 40u : l * 8u));
  const unsigned step = MODE == 3 ? 512u : (MODE == 2 ? 256u : (MODE == 4 ? 640u : 128u));
  double acc = 0.0;
  unsigned pos = 0;
  for (int r = 0; r < rounds; r++) {
    double v[8];
    d2 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const unsigned o = off + pos;
      if (MODE == 2) w[k] = *reinterpret_cast<const d2*>(p + o);
      else if (MODE == 1) { v[k] = 0.0; if (l < 10) v[k] = *reinterpret_cast<const double*>(p + o); }
      else v[k] = *reinterpret_cast<const double*>(p + o);
      pos += step;
      if (pos >= foot) pos = 0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) acc += MODE == 2 ? w[k].x + w[k].y : v[k];
  }
  if (acc == 123.456) out[blockIdx.x] = acc;
}

template <int MODE>
int run(const char* name, const char* buf, long long stride, unsigned foot, int rounds, double* out) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int waves = 2048;
  hipLaunchKernelGGL(loads<MODE>, dim3(waves), dim3(64), 0, 0, buf, stride, foot, rounds, out);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(loads<MODE>, dim3(waves), dim3(64), 0, 0, buf, stride, foot, rounds, out);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_cu = (double)waves / 256.0 * rounds * 8.0;
  const double ns = ms * 1e6 / instr_per_cu;
  const double lane_bytes = MODE == 2 ? 1024.0 : (MODE == 1 ? 320.0 : 512.0);
  printf("{\"case\": \"%s\", \"footprint\": %u, \"ms\": %.3f, \"ns_per_load_per_cu\": %.2f, \"cycles_at_2.4GHz\": %.1f, \"useful_TBps\": %.2f}\n", name, foot, ms, ns,
         ns * 2.4, (double)waves * rounds * 8.0 * lane_bytes / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  const long long stride = 1 << 20;   // 1 MB per problem region
  const size_t total = (size_t)2048 * 4 * stride + (1 << 20);
  char* buf;
  double* out;
  CHK(hipMalloc(&buf, total));
  CHK(hipMemset(buf, 0, total));
  CHK(hipMalloc(&out, 4096 * sizeof(double)));
  for (unsigned foot : {4096u, 1u << 20}) {
    const int rounds = foot <= 4096 ? 2000 : 800;
    if (run<0>("8B x 64 lanes, 4 regions", buf, stride, foot, rounds, out)) return 1;
    if (run<1>("8B x 40 lanes (10 of 16), 4 regions", buf, stride, foot, rounds, out)) return 1;
    if (run<2>("16B x 64 lanes, 4 regions", buf, stride, foot, rounds, out)) return 1;
    if (run<3>("8B x 64 lanes, 512 consecutive bytes", buf, stride, foot, rounds, out)) return 1;
    if (run<4>("8B x 64 lanes, lane stride 40B, 4 regions", buf, stride, foot, rounds, out)) return 1;
  }
  return 0;
}
