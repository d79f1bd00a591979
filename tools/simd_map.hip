// Which SIMD does wavefront w of a workgroup run on?  (round 6: role layout of band_newton_mw_kernel)  Reads HW_REG_HW_ID.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/simd_map tools/simd_map.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  extern __shared__ double lds[];
  const int w = threadIdx.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = hw;
  // keep the workgroup resident for a while so that two workgroups share a CU
  double a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0000001 + 1e-9;
  if (a == 123.456) lds[0] = a;
}
int main(int argc, char** argv) {
  const int wgs = 512, threads = argc > 1 ? atoi(argv[1]) : 512, nw = threads / 64;
  const size_t ldsb = argc > 2 ? (size_t)atoi(argv[2]) : 70000;
  unsigned* d; hipMalloc(&d, wgs * nw * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), ldsb, 0, d, 200000);
  std::vector<unsigned> h(wgs * nw);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  for (int b : {0, 1, 2, 8, 255, 256, 257, 511}) {
    printf("wg %3d:", b);
    for (int w = 0; w < nw; w++) { unsigned x = h[b * nw + w]; printf(" [w%d simd %u wave %u cu %u se %u raw %08x]", w, (x >> 4) & 3, x & 15, (x >> 8) & 15, (x >> 13) & 7, x); }
    printf("\n");
  }
  // histogram of the pattern simd(w) over all workgroups
  int pat[4][16] = {};
  for (int b = 0; b < wgs; b++) for (int w = 0; w < nw; w++) pat[(h[b * nw + w] >> 4) & 3][w]++;
  for (int s = 0; s < 4; s++) { printf("simd %d:", s); for (int w = 0; w < nw; w++) printf(" w%d=%d", w, pat[s][w]); printf("\n"); }
  return 0;
}
