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

// M: [N][LD] row-major in LDS (compile-time sizes, LD <= 128); on return the last LD - N columns hold A^-1 B.  Branch-free inner loops:
// a lane whose column is not updated writes back what it read.
template <int N, int LD>
__device__ int wave_lu(double *M, int lane) {
  constexpr int CH = 8;
  for (int k = 0; k < N; ++k) {
    double best = (lane >= k && lane < N) ? fabs(M[lane * LD + k]) : -1.0; int bi = lane;  // N <= 64: one candidate per lane
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!(best > 0.0)) return 1;
    if (bi != k) {
      const double t0 = M[k * LD + lane], t1 = M[bi * LD + lane];
      M[k * LD + lane] = t1; M[bi * LD + lane] = t0;
      if (LD > 64) { const int j = lane + 64 < LD ? lane + 64 : LD - 1; const double s0 = M[k * LD + j], s1 = M[bi * LD + j]; wsync(); M[k * LD + j] = s1; M[bi * LD + j] = s0; }
      wsync();
    }
    const double inv = 1.0 / M[k * LD + k];
    const int j0 = lane, j1 = lane + 64 < LD ? lane + 64 : LD - 1;
    const bool a0 = j0 > k, a1 = true;  // (lanes beyond the last column repeat it: same values computed and written again)
    const double u0 = M[k * LD + j0], u1 = M[k * LD + j1];
    for (int i0 = k + 1; i0 < N; i0 += CH) {
      double l[CH], x0[CH], x1[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = i0 + c < N ? i0 + c : N - 1;
        l[c] = M[i * LD + k]; x0[c] = M[i * LD + j0]; x1[c] = M[i * LD + j1];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = i0 + c < N ? i0 + c : N - 1;  // (a short batch repeats its last row: same value written twice)
        const double m_ = l[c] * inv;  // (the repeated row computes and writes the same values again)
        M[i * LD + j0] = a0 ? x0[c] - m_ * u0 : (j0 == k ? m_ : x0[c]);
        if (LD > 64) M[i * LD + j1] = a1 ? x1[c] - m_ * u1 : x1[c];
      }
      wsync();
    }
  }
  // back substitution, lane c = right-hand side c
  const int c = N + lane < LD ? N + lane : LD - 1;
  for (int k = N - 1; k >= 0; --k) {
    const double xk = M[k * LD + c] / M[k * LD + k];
    M[k * LD + c] = xk;
    for (int j0_ = 0; j0_ < k; j0_ += CH) {
      double u[CH], b[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) { const int j = j0_ + q < k ? j0_ + q : k - 1; u[q] = M[j * LD + k]; b[q] = M[j * LD + c]; }
#pragma unroll
      for (int q = 0; q < CH; ++q) { const int j = j0_ + q < k ? j0_ + q : k - 1; M[j * LD + c] = b[q] - u[q] * xk; }
      wsync();
    }
  }
  wsync();
  return 0;
}

// ---- the same with the block in REGISTERS: lane r holds row r of [A | B] (N + R doubles), the pivot by a DPP-free butterfly over the
// candidates, the pivot row broadcast entry by entry with v_readlane; every loop fully unrolled (static register indices).
__device__ __forceinline__ double lane_get(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int N, int R>
__device__ __forceinline__ int wave_lu_regs(double (&a)[N + R], int lane, int &ord) {
  bool done = lane >= N;
  ord = -1;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double best = done ? -1.0 : fabs(a[k]);
    double m = best;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
    if (!(m > 0.0)) return 1;
    const int pl = (int)__builtin_ctzll(__ballot(best == m));  // first lane holding the largest entry
    const double inv = 1.0 / lane_get(a[k], pl);
    const bool mine = lane == pl;
    const double l = (done || mine) ? 0.0 : a[k] * inv;
    if (mine) { done = true; ord = k; }
    if (l != 0.0) a[k] = l;  // rows still to be eliminated keep their multiplier (not needed again here), the pivot row keeps U
#pragma unroll
    for (int j = k + 1; j < N + R; ++j) a[j] -= l * lane_get(a[j], pl);
  }
  // back substitution: x_k sits in the lane whose row was the k-th pivot
#pragma unroll
  for (int k = N - 1; k >= 0; --k) {
    const int pl = (int)__builtin_ctzll(__ballot(ord == k));
    const double inv = 1.0 / lane_get(a[k], pl);
    const double u = (ord >= 0 && ord < k) ? a[k] : 0.0;  // U entry (row ord, column k) of the rows pivoted before k
#pragma unroll
    for (int c = 0; c < R; ++c) {
      const double x = lane_get(a[N + c], pl) * inv;
      a[N + c] = lane == pl ? x : a[N + c] - u * x;
    }
  }
  return 0;
}

template <int N, int R>
__global__ __launch_bounds__(512) void lu_regs_kernel(const double *A, const double *B, double *X, int nblocks, int *fail, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int blk = blockIdx.x * wpb + wave;
  if (blk >= nblocks) return;
  double a[N + R];
  const int r = lane < N ? lane : N - 1;
#pragma unroll
  for (int j = 0; j < N + R; ++j) a[j] = j < N ? A[((size_t)blk * N + r) * N + j] : B[((size_t)blk * N + r) * R + (j - N)];
  int ord;
  const long long t0 = wall_clock64();
  const int f = wave_lu_regs<N, R>(a, lane, ord);
  const long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
  if (f && lane == 0) atomicAdd(fail, 1);
  if (lane < N && ord >= 0) {
#pragma unroll
    for (int c = 0; c < R; ++c) X[((size_t)blk * N + ord) * R + c] = a[N + c];
  }
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
  const int f = (n == 64 && ld == 80) ? wave_lu<64, 80>(M, lane) : ((n == 64 && ld == 126) ? wave_lu<64, 126>(M, lane) : ((n == 48 && ld == 64) ? wave_lu<48, 64>(M, lane) : 1));
  const long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));  // 100 MHz ticks spent in the elimination itself (the copies around it are not the subject)
  if (f && lane == 0) atomicAdd(fail, 1);
  for (int i = 0; i < n; ++i) for (int j = lane; j < nrhs; j += 64) X[((size_t)blk * n + i) * nrhs + j] = M[i * ld + n + j];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, nrhs = argc > 2 ? atoi(argv[2]) : 62, nb = argc > 3 ? atoi(argv[3]) : 50 * 256, wpb = argc > 4 ? atoi(argv[4]) : 2;
  const bool regs = argc > 5;  // any sixth argument: the register version (64 x 64 + 16 only)
  if (!((n == 64 && (nrhs == 16 || nrhs == 62 || (nrhs == 22 && argc > 5))) || (n == 48 && nrhs == 16))) { fprintf(stderr, "compiled sizes: 64 16, 64 62, 48 16\n"); return 1; }
  std::vector<double> A((size_t)nb * n * n), B((size_t)nb * n * nrhs), X((size_t)nb * n * nrhs);
  srand(1);
  for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
  double *dA, *dB, *dX; int *df; unsigned long long *dc;
  OK(hipMalloc(&dA, A.size() * 8)); OK(hipMalloc(&dB, B.size() * 8)); OK(hipMalloc(&dX, X.size() * 8)); OK(hipMalloc(&df, 4)); OK(hipMalloc(&dc, 8));
  OK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); OK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice)); OK(hipMemset(df, 0, 4));
  const size_t lds = (size_t)wpb * n * (n + nrhs) * 8;
  if (!regs) OK(hipFuncSetAttribute((const void *)lu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (nb + wpb - 1) / wpb;
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    OK(hipMemset(dc, 0, 8));
    OK(hipEventRecord(e0));
    if (regs && nrhs == 22) hipLaunchKernelGGL((lu_regs_kernel<64, 22>), dim3(grid), dim3(64 * wpb), 0, 0, dA, dB, dX, nb, df, dc);
    else if (regs) hipLaunchKernelGGL((lu_regs_kernel<64, 16>), dim3(grid), dim3(64 * wpb), 0, 0, dA, dB, dX, nb, df, dc);
    else hipLaunchKernelGGL(lu_kernel, dim3(grid), dim3(64 * wpb), lds, 0, dA, dB, dX, n, nrhs, nb, df, dc);
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
  const double per_cu_concurrent = regs ? 8.0 : floor(160.0 * 1024 / lds) * wpb;  // register version: eight wavefronts of 256 VGPRs per CU
  if (regs) printf("REGISTER version: ");
  printf("n %d nrhs %d blocks %d, %d wavefronts per workgroup, LDS %zu B per workgroup: %.3f ms = %.2f us per block per CU-slot (%.0f blocks in flight per CU), "
         "elimination alone %.1f us per block (wall clock inside the kernel), failed %d, worst residual %.2e\n", n, nrhs, nb, wpb, lds, ms,
         ms * 1e3 / ((double)nb / (cus * per_cu_concurrent)), per_cu_concurrent, ticks * 0.01 / nb, fails / 3, worst);
  return 0;
}
