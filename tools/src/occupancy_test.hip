// Diagnostic: how many 64-thread workgroups with a given dynamic LDS size really run concurrently.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void spin(double *out, unsigned long long *when, int iters) {
  extern __shared__ double sm[];
  const unsigned long long w0 = wall_clock64();
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
  sm[threadIdx.x] = a;
  for (int i = 0; i < iters; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
  out[blockIdx.x * 64 + threadIdx.x] = a + sm[63 - threadIdx.x];
  if (threadIdx.x == 0) { when[2 * blockIdx.x] = w0; when[2 * blockIdx.x + 1] = wall_clock64(); }
}
int main() {
  double *out; unsigned long long *when;
  CK(hipMalloc(&out, 4096 * 64 * 8)); CK(hipMalloc(&when, 4096 * 16));
  static unsigned long long h[8192];
  const int lds[] = {54208, 53248, 52224, 49152, 40960, 32768, 16384};
  for (int l : lds) {
    CK(hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, l));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)spin, 64, l));
    for (int blocks : {768, 1024, 2048}) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), l, 0, out, when, 400000);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, when, blocks * 16, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0; for (int b = 0; b < blocks; ++b) { if (h[2 * b] < t0) t0 = h[2 * b]; if (h[2 * b + 1] > t1) t1 = h[2 * b + 1]; }
      // blocks that started within the first 10 % of one block's duration = resident in the first round
      const unsigned long long dur = h[1] - h[0]; int first = 0;
      for (int b = 0; b < blocks; ++b) if (h[2 * b] - t0 < dur / 10) ++first;
      printf("lds %6d B: occupancy API %d per CU; %4d blocks: %4d started at once, kernel = %.2f block durations\n", l, per_cu, blocks, first, (double)(t1 - t0) / dur);
    }
  }
  return 0;
}
