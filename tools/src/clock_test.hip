// Diagnostic: how fast a dependent FP64 chain runs with all lanes active vs one lane active, and what the
// s_memtime counter does meanwhile (wall_clock64 = constant 100 MHz reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
// mode 0: all lanes; 1: lane 0 only; 2: even blocks all lanes, odd blocks lane 0 only
__global__ void spin(double *out, unsigned long long *ticks, int iters, int mode) {
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9;
  const bool single = mode == 1 || (mode == 2 && (blockIdx.x & 1));
  const unsigned long long w0 = wall_clock64(), t0 = __builtin_amdgcn_s_memtime();
  if (!single) {
    for (int i = 0; i < iters; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
  } else {
    if (threadIdx.x == 0) for (int i = 0; i < iters; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
  const int cfg[][3] = {{768, 64, 0}, {768, 64, 1}, {768, 64, 2}, {768, 64, 0}, {256, 64, 1}, {256, 64, 0}};
  double *out; unsigned long long *ticks;
  CK(hipMalloc(&out, 8192 * 256 * 8)); CK(hipMalloc(&ticks, 8192 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static unsigned long long h[8192 * 2];
  for (auto &c : cfg) {
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 3000000;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(spin, dim3(c[0]), dim3(c[1]), 0, 0, out, ticks, iters, c[2]);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h, ticks, c[0] * 16, hipMemcpyDeviceToHost));
      double se = 0, so = 0, we = 0, wo = 0;
      for (int b = 0; b < c[0]; ++b) { if (b & 1) { so += h[2 * b]; wo += h[2 * b + 1]; } else { se += h[2 * b]; we += h[2 * b + 1]; } }
      const double n2 = c[0] / 2.0, nf = 4.0 * iters;
      printf("blocks %4d x %3d mode %d: kernel %.1f ms | even blocks: %.2f ns per fma, %.2f memtime ticks per fma | odd blocks: %.2f ns, %.2f ticks\n",
             c[0], c[1], c[2], ms, we / n2 * 10.0 / nf, se / n2 / nf, wo / n2 * 10.0 / nf, so / n2 / nf);
    }
  }
  return 0;
}
