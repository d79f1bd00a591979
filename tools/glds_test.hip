// checks on the GPU: global_load_lds_dwordx4 from an 8-byte-aligned (not 16) source through inline asm, saddr form, M0 = LDS byte address
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const char* base, unsigned off, double* dst) {
  const unsigned a = (unsigned)(unsigned long long)(lptr_t)dst;
  const unsigned au = __builtin_amdgcn_readfirstlane(a);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(base), "s"(au) : "memory");
}
__global__ void k(const double* __restrict__ g, double* out, int nbytes, int shift) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) lds[i] = -1.0;
  __syncthreads();
  const char* src = reinterpret_cast<const char*>(g) + 8 * shift;
  if (lane * 16 < nbytes) glds16(src, lane * 16, lds + 32);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; i++) h[i] = i;
  double *g, *o;
  hipMalloc(&g, 8192); hipMalloc(&o, 4096);
  hipMemcpy(g, h.data(), 8192, hipMemcpyHostToDevice);
  int bad = 0;
  for (int shift = 0; shift < 4; shift++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, 920, shift);
    std::vector<double> r(512);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 512; i++) {
      double want = -1.0;
      if (i >= 32 && i < 32 + 116) want = (i - 32) + shift;   // 920 bytes = 115 doubles -> 58 lanes * 2 = 116 doubles
      if (r[i] != want) { if (bad < 10) printf("shift %d i %d got %g want %g\n", shift, i, r[i], want); bad++; }
    }
  }
  printf("glds test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
