// Study for the structured collocation elimination (docs/notebook.md, round 4): ONE wavefront factors a dense n x n block with partial
// pivoting and solves for nrhs right-hand sides, the augmented matrix [A | B] in LDS, lane j = column j (and j + 64) of the row being
// updated.  Several blocks per workgroup (one per wavefront), as many workgroups per CU as the LDS allows.  Checked against a host
// elimination; prints the time per block.   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_lu tools/src/wave_lu_bench.hip && /tmp/wave_lu [n] [nrhs] [blocks] [waves per workgroup]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline void wsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// M: [n][ld] row-major in LDS, ld = n + nrhs (<= 128); on return the last nrhs columns hold A^-1 B (rows in pivot order restored)
__device__ int wave_lu(double *M, int n, int ld, int lane) {
  for (int k = 0; k < n; ++k) {
    // pivot: largest |M[i][k]|, i >= k (lanes i and i + 64 are both candidates for n > 64)
    double best = -1.0; int bi = k;
    for (int i = k + lane; i < n; i += 64) { const double a = fabs(M[i * ld + k]); if (a > best) { best = a; bi = i; } }
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!(best > 0.0)) return 1;
    if (bi != k) {
      for (int j = lane; j < ld; j += 64) { const double t = M[k * ld + j]; M[k * ld + j] = M[bi * ld + j]; M[bi * ld + j] = t; }
      wsync();
    }
    const double inv = 1.0 / M[k * ld + k];
    // row i -= l_i row k, columns k+1 .. ld-1: lane j takes columns j, j + 64; the multiplier is a broadcast read
    double u0 = 0.0, u1 = 0.0;
    const int j0 = lane, j1 = lane + 64;
    if (j0 > k && j0 < ld) u0 = M[k * ld + j0];
    if (j1 > k && j1 < ld) u1 = M[k * ld + j1];
    // rows in batches of CH: the CH multipliers and the 2 CH entries are read together, updated, written together -- one LDS round trip
    // per batch instead of one per row (measured: 480 us per 64 x 126 block row by row)
    constexpr int CH = 8;
    const bool a0 = j0 > k && j0 < ld, a1 = j1 > k && j1 < ld;
    for (int i0 = k + 1; i0 < n; i0 += CH) {
      double l[CH], x0[CH], x1[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = i0 + c < n ? i0 + c : n - 1;
        l[c] = M[i * ld + k];
        x0[c] = a0 ? M[i * ld + j0] : 0.0;
        x1[c] = a1 ? M[i * ld + j1] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = i0 + c;
        if (i < n) {
          const double m_ = l[c] * inv;
          if (a0) M[i * ld + j0] = x0[c] - m_ * u0;
          if (a1) M[i * ld + j1] = x1[c] - m_ * u1;
          if (lane == 0) M[i * ld + k] = m_;
        }
      }
    }
    wsync();
  }
  // back substitution on the right-hand sides: lane c takes column n + c (and n + c + 64)
  // back substitution, lane c = right-hand side c (two passes for more than 64): x_k = b_k / u_kk, then b_j -= u_jk x_k for j < k in
  // batches of CH rows
  for (int c = n + lane; c < ld + 63 - (ld + 63 - n) % 64; c += 64) {
    const bool act = c < ld;
    for (int k = n - 1; k >= 0; --k) {
      const double xk = act ? M[k * ld + c] / M[k * ld + k] : 0.0;
      if (act) M[k * ld + c] = xk;
      constexpr int CH = 8;
      for (int j0_ = 0; j0_ < k; j0_ += CH) {
        double u[CH], b[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int j = j0_ + q < k ? j0_ + q : k - 1; u[q] = M[j * ld + k]; b[q] = act ? M[j * ld + c] : 0.0; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int j = j0_ + q; if (j < k && act) M[j * ld + c] = b[q] - u[q] * xk; }
      }
    }
  }
  wsync();
  return 0;
}

__global__ void lu_kernel(const double *A, const double *B, double *X, int n, int nrhs, int nblocks, int *fail, unsigned long long *cyc) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6, ld = n + nrhs;
  const int blk = blockIdx.x * wpb + wave;
  if (blk >= nblocks) return;
  double *M = lds + (size_t)wave * n * ld;
  for (int i = 0; i < n; ++i) for (int j = lane; j < ld; j += 64) M[i * ld + j] = j < n ? A[((size_t)blk * n + i) * n + j] : B[((size_t)blk * n + i) * nrhs + (j - n)];
  wsync();
  const long long t0 = wall_clock64();
  const int f = wave_lu(M, n, ld, lane);
  const long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));  // 100 MHz ticks spent in the elimination itself (the copies around it are not the subject)
  if (f && lane == 0) atomicAdd(fail, 1);
  for (int i = 0; i < n; ++i) for (int j = lane; j < nrhs; j += 64) X[((size_t)blk * n + i) * nrhs + j] = M[i * ld + n + j];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, nrhs = argc > 2 ? atoi(argv[2]) : 62, nb = argc > 3 ? atoi(argv[3]) : 50 * 256, wpb = argc > 4 ? atoi(argv[4]) : 2;
  if (n + nrhs > 128) { fprintf(stderr, "n + nrhs <= 128\n"); return 1; }
  std::vector<double> A((size_t)nb * n * n), B((size_t)nb * n * nrhs), X((size_t)nb * n * nrhs);
  srand(1);
  for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
  double *dA, *dB, *dX; int *df; unsigned long long *dc;
  OK(hipMalloc(&dA, A.size() * 8)); OK(hipMalloc(&dB, B.size() * 8)); OK(hipMalloc(&dX, X.size() * 8)); OK(hipMalloc(&df, 4)); OK(hipMalloc(&dc, 8));
  OK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); OK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); OK(hipMemset(df, 0, 4));
  const size_t lds = (size_t)wpb * n * (n + nrhs) * 8;
  OK(hipFuncSetAttribute((const void *)lu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (nb + wpb - 1) / wpb;
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    OK(hipMemset(dc, 0, 8));
    OK(hipEventRecord(e0));
    hipLaunchKernelGGL(lu_kernel, dim3(grid), dim3(64 * wpb), lds, 0, dA, dB, dX, n, nrhs, nb, df, dc);
    OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1)); OK(hipEventElapsedTime(&ms, e0, e1));
  }
  OK(hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost));
  int fails = 0; OK(hipMemcpy(&fails, df, 4, hipMemcpyDeviceToHost));
  unsigned long long ticks = 0; OK(hipMemcpy(&ticks, dc, 8, hipMemcpyDeviceToHost));
  // host check of a few blocks: residual A x - b
  double worst = 0.0;
  for (int blk = 0; blk < nb; blk += nb / 7 + 1)
    for (int c = 0; c < nrhs; c += 5)
      for (int i = 0; i < n; ++i) {
        double s = -B[((size_t)blk * n + i) * nrhs + c];
        for (int j = 0; j < n; ++j) s += A[((size_t)blk * n + i) * n + j] * X[((size_t)blk * n + j) * nrhs + c];
        worst = fmax(worst, fabs(s));
      }
  int cus = 0; OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const double per_cu_concurrent = floor(160.0 * 1024 / lds) * wpb;
  printf("n %d nrhs %d blocks %d, %d wavefronts per workgroup, LDS %zu B per workgroup: %.3f ms = %.2f us per block per CU-slot (%.0f blocks in flight per CU), "
         "elimination alone %.1f us per block (wall clock inside the kernel), failed %d, worst residual %.2e\n", n, nrhs, nb, wpb, lds, ms,
         ms * 1e3 / ((double)nb / (cus * per_cu_concurrent)), per_cu_concurrent, ticks * 0.01 / nb, fails / 3, worst);
  return 0;
}
