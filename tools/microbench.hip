// microbench.hip — gfx950 numbers the dense fp64 path is designed against (not in the local guides):
//   fp64 MFMA (v_mfma_f64_16x16x4_f64) issue rate and chip throughput, v_fma_f64 issue rate for one wave alone,
//   dependent-chain latencies (fma, rcp, LDS round trip) and the cost of a dependent kernel boundary.
// Build: hipcc -O3 --offload-arch=gfx950 -o microbench microbench.hip ; run on the GPU box, prints one JSON object.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, long long* cyc) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void __launch_bounds__(256) fma_loop(double* out, int iters, long long* cyc) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-9 + i;
  double a = 1.0000001, b = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], a, b);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) chain_kernel(double* out, int iters, long long* cyc) {
  __shared__ double lds[128];
  const int l = threadIdx.x;
  double x = 1.0 + l * 1e-6;
  // 0: dependent fma chain
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) x = fma(x, 1.0000001, 1e-9);
  long long t1 = __builtin_amdgcn_s_memtime();
  // 1: dependent rcp chain (v_rcp_f64)
  double y = 1.5 + l * 1e-6;
  for (int it = 0; it < iters; it++) y = __builtin_amdgcn_rcp(y) + 0.5;
  long long t2 = __builtin_amdgcn_s_memtime();
  // 2: LDS round trip: write own slot, read neighbour's slot (wave-synchronous)
  double z = x;
  for (int it = 0; it < iters; it++) {
    lds[l] = z;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    z = lds[(l + 1) & 63] + 1e-9;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  long long t3 = __builtin_amdgcn_s_memtime();
  // 3: full IEEE division chain
  double w = 1.5 + l * 1e-6;
  for (int it = 0; it < iters; it++) w = 1.0 / w + 0.5;
  long long t4 = __builtin_amdgcn_s_memtime();
  out[l] = x + y + z + w;
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}

__global__ void __launch_bounds__(256) tiny_kernel(double* p, int k) {
  if (threadIdx.x == 0) p[blockIdx.x] += k;
}
struct BigArgs { double* p[40]; int v[24]; };  // ~416 bytes of kernel arguments, like the dense backend's DnDev
__global__ void __launch_bounds__(256) tiny_big(BigArgs a, int k) {
  if (threadIdx.x == 0) a.p[k & 7][blockIdx.x] += a.v[k & 15];
}
__global__ void __launch_bounds__(1024) tiny_lds(BigArgs a, int k) {
  __shared__ double buf[5000];
  buf[threadIdx.x] = k;
  __syncthreads();
  if (threadIdx.x == 0) a.p[k & 7][blockIdx.x] += buf[(k * 7) & 1023];
}
__global__ void __launch_bounds__(256) tiny_empty(int k) {
  if (k == 0x7fffffff) __builtin_trap();
}
// reads a value the previous kernel wrote (the dependent chain of the dense steps) from a different 2 MB region each time
__global__ void __launch_bounds__(256) tiny_dep(double* base, int k) {
  double* p = base + ((size_t)(k & 63) << 18);
  if (threadIdx.x == 0) p[blockIdx.x + 256] = p[blockIdx.x] + 1.0;
}

int main() {
  int dev = 0;
  CHK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  double* out; long long* cyc;
  CHK(hipMalloc(&out, 1 << 24));
  CHK(hipMalloc(&cyc, 1 << 20));
  CHK(hipMemset(out, 0, 1 << 24));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.name, ncu, prop.clockRate / 1000);
  std::vector<long long> hc(4096);
  // ---- MFMA f64: blocks of 256 threads (one wave per SIMD) x blocks per CU ----
  for (int bpc : {1, 2}) {
    const int iters = 20000, nacc = 4;
    const int grid = ncu * bpc;
    hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, 100, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(hc.data(), cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost));
    const double nm = (double)grid * 4 * iters * nacc;  // MFMAs
    const double tf = nm * 2048.0 / (ms * 1e-3) / 1e12;
    printf(", \"mfma_f64_16x16x4_wavesPerSimd%d\": {\"TFLOPs\": %.2f, \"ms\": %.3f, \"memtime_ticks_per_mfma_per_wave\": %.2f}", bpc, tf, ms,
           (double)hc[0] / (iters * nacc));
  }
  // sweep: accumulators per wave x waves per SIMD
  {
    printf(", \"mfma_f64_sweep_TFLOPs\": {");
    bool first = true;
    for (int nacc : {1, 2, 8}) for (int bpc : {1, 2, 4}) {
      const int iters = 8000;
      const int grid = ncu * bpc;
      auto launch = [&](int it) {
        if (nacc == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, it, cyc);
        else if (nacc == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(grid), dim3(256), 0, 0, out, it, cyc);
        else hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, it, cyc);
      };
      launch(50);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      launch(iters);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      const double nm = (double)grid * 4 * iters * nacc;
      printf("%s\"acc%d_w%d\": %.1f", first ? "" : ", ", nacc, bpc, nm * 2048.0 / (ms * 1e-3) / 1e12);
      first = false;
    }
    printf("}");
  }
  // one accumulator: dependent MFMA latency
  {
    const int iters = 20000;
    hipLaunchKernelGGL(mfma_loop<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<1>, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemcpy(hc.data(), cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost));
    printf(", \"mfma_f64_dependent\": {\"ns_per_mfma\": %.2f, \"memtime_ticks\": %.2f}", ms * 1e6 / iters, (double)hc[0] / iters);
  }
  // ---- v_fma_f64 ----
  for (int bpc : {1, 4}) {
    const int iters = 20000, nacc = 8;
    const int grid = ncu * bpc;
    hipLaunchKernelGGL(fma_loop<8>, dim3(grid), dim3(256), 0, 0, out, 100, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(fma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double nf = (double)grid * 256 * iters * nacc * 2.0;
    printf(", \"v_fma_f64_wavesPerSimd%d\": {\"TFLOPs\": %.2f, \"ns_per_wave_instr\": %.3f}", bpc, nf / (ms * 1e-3) / 1e12,
           ms * 1e6 / ((double)iters * nacc * bpc));
  }
  // ---- chains ----
  {
    const int iters = 4000;
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, out, iters, cyc);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(hc.data(), cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost));
    printf(", \"chain_memtime_ticks\": {\"fma_f64\": %.1f, \"rcp_f64_plus_add\": %.1f, \"lds_write_read_roundtrip\": %.1f, \"ieee_div_plus_add\": %.1f}",
           (double)hc[0] / iters, (double)hc[1] / iters, (double)hc[2] / iters, (double)hc[3] / iters);
  }
  // ---- kernel boundary ----
  for (int grid : {1, 136, 1024}) {
    const int n = 2000;
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, 0, out, i);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < n; i++) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(256), 0, 0, out, i);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf(", \"kernel_boundary_us_grid%d\": %.3f", grid, ms * 1e3 / n);
  }
  {
    double* depbuf = nullptr;
    CHK(hipMalloc(&depbuf, ((size_t)64 << 21) + (1 << 20)));  // 64 regions of 2 MB
    CHK(hipMemset(depbuf, 0, ((size_t)64 << 21) + (1 << 20)));
    BigArgs ba; for (int i = 0; i < 40; i++) ba.p[i] = out + 4096 * i; for (int i = 0; i < 24; i++) ba.v[i] = i;
    const int n = 2000;
    for (int variant = 0; variant < 3; variant++) {
      auto go = [&](int i) {
        if (variant == 0) hipLaunchKernelGGL(tiny_empty, dim3(136), dim3(256), 0, 0, i);
        else if (variant == 1) hipLaunchKernelGGL(tiny_big, dim3(136), dim3(256), 0, 0, ba, i);
        else hipLaunchKernelGGL(tiny_dep, dim3(136), dim3(256), 0, 0, depbuf, i);
      };
      for (int i = 0; i < 10; i++) go(i);
      CHK(hipDeviceSynchronize());
      CHK(hipEventRecord(e0));
      for (int i = 0; i < n; i++) go(i);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      printf(", \"kernel_boundary_us_%s\": %.3f", variant == 0 ? "no_memory" : (variant == 1 ? "416B_args" : "dependent_load_store"), ms * 1e3 / n);
    }
  }
  // the same through a captured graph (no host launch cost in the timed region)
  {
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(tiny_kernel, dim3(136), dim3(256), 0, st, out, i);
    CHK(hipStreamEndCapture(st, &g));
    {
      BigArgs ba; for (int i = 0; i < 40; i++) ba.p[i] = out + 4096 * i; for (int i = 0; i < 24; i++) ba.v[i] = i;
      for (int variant = 0; variant < 2; variant++) {
        hipGraph_t g2; hipGraphExec_t ge2;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; i++) {
          if (variant == 0) hipLaunchKernelGGL(tiny_big, dim3(136), dim3(256), 0, st, ba, i);
          else hipLaunchKernelGGL(tiny_lds, dim3(136), dim3(1024), 0, st, ba, i);
        }
        CHK(hipStreamEndCapture(st, &g2));
        CHK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ge2, st)); CHK(hipStreamSynchronize(st));
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; r++) CHK(hipGraphLaunch(ge2, st));
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        float ms2; CHK(hipEventElapsedTime(&ms2, e0, e1));
        printf(", \"graph_kernel_boundary_us_%s\": %.3f", variant == 0 ? "416B_args" : "416B_args_1024thr_40KB_lds", ms2 * 1e3 / 2000);
      }
    }
    CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; r++) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf(", \"graph_kernel_boundary_us_grid136\": %.3f", ms * 1e3 / 2000);
  }
  printf("}\n");
  return 0;
}
