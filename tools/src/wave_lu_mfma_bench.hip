// Go / no-go study (docs/notebook.md, round 5): the register-resident block elimination of the planning kernels (cfz_struct.inl
// wave_lu_regs: lane = row, the pivot row broadcast entry by entry with v_readlane, ~330 vector instructions per pivot of a 64 x 96 block)
// against a BLOCKED elimination on the matrix cores: the block in the accumulator layout of v_mfma_f64_16x16x4_f64 (tile (I, J), register g
// of lane l = row 16 I + 4 g + (l >> 4), column 16 J + (l & 15)), panels of four pivots factored in a lane = row copy of the panel's columns
// (through LDS), rows never moved (implicit pivoting: the multipliers of finished rows are zero), the trailing update as one matrix
// instruction per tile and panel, the back-substitution four unknowns at a time the same way.  Same pivots, same products in the same
// order (the matrix instruction accumulates its four k in ascending order, one fused multiply-add each): the result is compared BIT FOR BIT.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_lu_mfma tools/src/wave_lu_mfma_bench.hip && /tmp/wave_lu_mfma [blocks] [wavefronts per workgroup]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v16d __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) double lds_f64;

template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_get(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_mov<0xB1>(v));
  v = fmax(v, dpp_mov<0x4E>(v));
  v = fmax(v, dpp_mov<0x141>(v));
  v = fmax(v, dpp_mov<0x140>(v));
  return fmax(fmax(lane_get(v, 0), lane_get(v, 16)), fmax(lane_get(v, 32), lane_get(v, 48)));
}

// ---- the product code's elimination (cfz_struct.inl), the baseline -----------------------------------------------------------------
template <int NB, int RB>
__device__ __forceinline__ int wave_lu_regs(double (&a)[NB + RB], int lane, int &ord) {
  bool done = lane >= NB;
  ord = -1;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const double best = done ? -1.0 : fabs(a[k]);
    const double m = wave_max(best);
    if (!(m > 0.0)) return 1;
    const int pl = (int)__builtin_ctzll(__ballot(best == m));
    const double inv = 1.0 / lane_get(a[k], pl);
    const bool mine = lane == pl;
    const double l = (done || mine) ? 0.0 : a[k] * inv;
    if (mine) { done = true; ord = k; }
#pragma unroll
    for (int j = k + 1; j < NB + RB; ++j) a[j] -= l * lane_get(a[j], pl);
  }
#pragma unroll
  for (int k = NB - 1; k >= 0; --k) {
    const int pl = (int)__builtin_ctzll(__ballot(ord == k));
    const double inv = 1.0 / lane_get(a[k], pl);
    const double u = (ord >= 0 && ord < k) ? a[k] : 0.0;
#pragma unroll
    for (int c = 0; c < RB; ++c) {
      const double x = lane_get(a[NB + c], pl) * inv;
      a[NB + c] = lane == pl ? x : a[NB + c] - u * x;
    }
  }
  return 0;
}

template <int RB>
__global__ __launch_bounds__(512) void lu_regs_kernel(const double *A, const double *B, double *X, int nblocks, int *fail, unsigned long long *cyc) {
  constexpr int N = 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int blk = blockIdx.x * wpb + wave;
  if (blk >= nblocks) return;
  double a[N + RB];
#pragma unroll
  for (int j = 0; j < N + RB; ++j) a[j] = j < N ? A[((size_t)blk * N + lane) * N + j] : B[((size_t)blk * N + lane) * RB + (j - N)];
  int ord;
  const long long t0 = wall_clock64();
  const int f = wave_lu_regs<N, RB>(a, lane, ord);
  const long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
  if (f && lane == 0) atomicAdd(fail, 1);
  if (ord >= 0) {
#pragma unroll
    for (int c = 0; c < RB; ++c) X[((size_t)blk * N + ord) * RB + c] = a[N + c];
  }
}

// ---- the blocked elimination on the matrix cores -----------------------------------------------------------------------------------
// Column tile J of a lane is ONE vector of sixteen doubles, T[J][4 I + g] = row 16 I + 4 g + (l >> 4), column 16 J + (l & 15): a pivot row p
// is element p >> 2 of the lanes (l >> 4) == (p & 3) -- a dynamic but UNIFORM element index, which the compiler turns into s_set_gpr_idx /
// v_mov (no branch, no copy of the tile); register 4 I .. 4 I + 3 of the vector are the accumulator of tile (I, J).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ void tile_mfma(v16d &t, int I, double a, double b) {
  v4d c = {t[4 * I], t[4 * I + 1], t[4 * I + 2], t[4 * I + 3]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  t[4 * I] = c[0]; t[4 * I + 1] = c[1]; t[4 * I + 2] = c[2]; t[4 * I + 3] = c[3];
}
// element e (uniform) of the column tiles J0 .. J1 - 1: the compiler's s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off.  (A scalar branch over
// the sixteen registers is turned back into exactly this by the optimiser; with the moves as inline assembly it stays a branch, costs the
// same 47 us and seven MINUTES of compilation per instantiation.  With the tiles as a [4][NT] array of 4-vectors indexed from a loop over
// the panels, they live in scratch memory: 143 us.)
template <int NT, int J0, int J1>
__device__ __forceinline__ void tile_row(const v16d (&T)[NT], int e, double (&u)[NT]) {
#pragma unroll
  for (int J = J0; J < J1; ++J) u[J] = T[J][e];
}
__device__ __forceinline__ double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// T: 64 x (64 + RB) in tiles; lds: 2 KB (panel, lane = row) + 4 x 16 (NT) doubles (pivot rows / right-hand sides of a panel).  The solution X
// (row = unknown) goes to out[k * ostride + c].  0 = ok.  One function per panel (P a template parameter, not a loop variable: a sixteen-
// trip loop of this size is beyond the unroller's budget, and a tile array indexed by a loop variable stays in scratch memory).
// Two trips through LDS per panel: (1) the panel's four columns to a lane = row copy; (2) the multipliers to the matrix instruction's
// first-operand layout and the four pivot rows to every lane of their column.
template <int RB, int P>
__device__ __forceinline__ int lu_forward_panel(v16d (&T)[4 + RB / 16], lds_f64 *Pb, lds_f64 *Ub, bool &done, int &ord) {
  constexpr int NT = 4 + RB / 16, k0 = 4 * P, Jp = P >> 2, c0 = k0 & 15;
  const int lane = threadIdx.x & 63, cj = lane & 15, rg = lane >> 4;
  if (cj >= c0 && cj < c0 + 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Pb[(4 * e + rg) * 4 + (cj - c0)] = T[Jp][e];
  }
  wave_sync();
  double a[4], l[4];
  int pl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = Pb[lane * 4 + q];
  wave_sync();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double best = done ? -1.0 : fabs(a[q]);
    const double m = wave_max(best);
    if (!(m > 0.0)) return 1;
    pl[q] = (int)__builtin_ctzll(__ballot(best == m));
    const double inv = 1.0 / lane_get(a[q], pl[q]);
    const bool mine = lane == pl[q];
    l[q] = (done || mine) ? 0.0 : a[q] * inv;
    if (mine) { done = true; ord = k0 + q; Ub[4 * 16 * NT + k0 + q] = inv; }  // (the reciprocal pivot, kept for the back-substitution)
#pragma unroll
    for (int s = q + 1; s < 4; ++s) a[s] -= l[q] * lane_get(a[s], pl[q]);
  }
  // the multipliers (negated) and the four pivot rows as they stand (before the panel's own eliminations)
#pragma unroll
  for (int q = 0; q < 4; ++q) Pb[lane * 4 + q] = -l[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = __builtin_amdgcn_readfirstlane(pl[q] >> 2);
    double u[NT];
    tile_row<NT, Jp, NT>(T, e, u);
    if (rg == (pl[q] & 3)) {
#pragma unroll
      for (int J = Jp; J < NT; ++J) Ub[q * 16 * NT + 16 * J + cj] = u[J];
    }
  }
  wave_sync();
  double Aop[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) Aop[I] = Pb[(16 * I + cj) * 4 + rg];
  // every lane brings the four rows of its column up to date (row q with the pivots before it in the panel, in their order), zeroes the
  // columns up to each row's own pivot, and keeps the row of its lane group: the matrix instruction's second operand.  A column tile at a
  // time (four values live, not 4 NT)
  const double l10 = lane_get(l[0], pl[1]), l20 = lane_get(l[0], pl[2]), l21 = lane_get(l[1], pl[2]);
  const double l30 = lane_get(l[0], pl[3]), l31 = lane_get(l[1], pl[3]), l32 = lane_get(l[2], pl[3]);
#pragma unroll
  for (int J = Jp; J < NT; ++J) {
    double u0 = Ub[0 * 16 * NT + 16 * J + cj], u1 = Ub[1 * 16 * NT + 16 * J + cj], u2 = Ub[2 * 16 * NT + 16 * J + cj], u3 = Ub[3 * 16 * NT + 16 * J + cj];
    u1 -= l10 * u0;
    u2 -= l20 * u0; u2 -= l21 * u1;
    u3 -= l30 * u0; u3 -= l31 * u1; u3 -= l32 * u2;
    if (J == Jp) {
      if (cj <= c0) u0 = 0.0;
      if (cj <= c0 + 1) u1 = 0.0;
      if (cj <= c0 + 2) u2 = 0.0;
      if (cj <= c0 + 3) u3 = 0.0;
    }
    const double b = rg == 0 ? u0 : (rg == 1 ? u1 : (rg == 2 ? u2 : u3));
#pragma unroll
    for (int I = 0; I < 4; ++I) tile_mfma(T[J], I, Aop[I], b);
  }
  wave_sync();
  return 0;
}
// back-substitution, four unknowns at a time
template <int RB, int P>
__device__ __forceinline__ void lu_backward_panel(v16d (&T)[4 + RB / 16], lds_f64 *Pb, lds_f64 *Ub, int ord, double *out, int ostride) {
  constexpr int NT = 4 + RB / 16, NR = RB / 16, k0 = 4 * P, Jp = P >> 2, c0 = k0 & 15;
  const int lane = threadIdx.x & 63, cj = lane & 15, rg = lane >> 4;
  if (cj >= c0 && cj < c0 + 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Pb[(4 * e + rg) * 4 + (cj - c0)] = T[Jp][e];
  }
  wave_sync();
  double a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = Pb[lane * 4 + q];
  wave_sync();
  int pk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pk[i] = (int)__builtin_ctzll(__ballot(ord == k0 + i));
    const int e = __builtin_amdgcn_readfirstlane(pk[i] >> 2);
    double u[NT];
    tile_row<NT, 4, NT>(T, e, u);
    if (rg == (pk[i] & 3)) {
#pragma unroll
      for (int Jr = 0; Jr < NR; ++Jr) Ub[i * 16 * NT + 16 * Jr + cj] = u[4 + Jr];
    }
  }
  // rows pivoted before this panel: right-hand sides -= (their entries in the panel's columns) x (the four solutions), last column first
  const bool early = ord < k0;
#pragma unroll
  for (int i = 0; i < 4; ++i) Pb[lane * 4 + i] = early ? -a[3 - i] : 0.0;
  wave_sync();
  double Aop[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) Aop[I] = Pb[(16 * I + cj) * 4 + rg];
  // the panel's 4 x 4 triangle (entries of the pivot rows in the panel's columns) and the reciprocal pivots: uniform
  const double i3 = uni(Ub[4 * 16 * NT + k0 + 3]), i2 = uni(Ub[4 * 16 * NT + k0 + 2]), i1 = uni(Ub[4 * 16 * NT + k0 + 1]), i0 = uni(Ub[4 * 16 * NT + k0]);
  const double u23 = lane_get(a[3], pk[2]), u13 = lane_get(a[3], pk[1]), u12 = lane_get(a[2], pk[1]);
  const double u03 = lane_get(a[3], pk[0]), u02 = lane_get(a[2], pk[0]), u01 = lane_get(a[1], pk[0]);
#pragma unroll
  for (int Jr = 0; Jr < NR; ++Jr) {  // a column tile of right-hand sides at a time (four values live)
    double x0 = Ub[0 * 16 * NT + 16 * Jr + cj], x1 = Ub[1 * 16 * NT + 16 * Jr + cj], x2 = Ub[2 * 16 * NT + 16 * Jr + cj], x3 = Ub[3 * 16 * NT + 16 * Jr + cj];
    x3 *= i3;
    x2 -= u23 * x3; x2 *= i2;
    x1 -= u13 * x3; x1 -= u12 * x2; x1 *= i1;
    x0 -= u03 * x3; x0 -= u02 * x2; x0 -= u01 * x1; x0 *= i0;
    out[(size_t)(k0 + rg) * ostride + 16 * Jr + cj] = rg == 0 ? x0 : (rg == 1 ? x1 : (rg == 2 ? x2 : x3));
    if (P > 0) {
      const double b = rg == 3 ? x0 : (rg == 2 ? x1 : (rg == 1 ? x2 : x3));
#pragma unroll
      for (int I = 0; I < 4; ++I) tile_mfma(T[4 + Jr], I, Aop[I], b);
    }
  }
  wave_sync();
}
template <int RB>
__device__ __forceinline__ int wave_lu_mfma(v16d (&T)[4 + RB / 16], lds_f64 *lds, double *out, int ostride) {
  lds_f64 *Pb = lds, *Ub = lds + 256;
  bool done = false;
  int ord = -1;
#define CFZ_FWD(P) if (lu_forward_panel<RB, P>(T, Pb, Ub, done, ord)) return 1;
  CFZ_FWD(0) CFZ_FWD(1) CFZ_FWD(2) CFZ_FWD(3) CFZ_FWD(4) CFZ_FWD(5) CFZ_FWD(6) CFZ_FWD(7)
  CFZ_FWD(8) CFZ_FWD(9) CFZ_FWD(10) CFZ_FWD(11) CFZ_FWD(12) CFZ_FWD(13) CFZ_FWD(14) CFZ_FWD(15)
#undef CFZ_FWD
#define CFZ_BWD(P) lu_backward_panel<RB, P>(T, Pb, Ub, ord, out, ostride);
  CFZ_BWD(15) CFZ_BWD(14) CFZ_BWD(13) CFZ_BWD(12) CFZ_BWD(11) CFZ_BWD(10) CFZ_BWD(9) CFZ_BWD(8)
  CFZ_BWD(7) CFZ_BWD(6) CFZ_BWD(5) CFZ_BWD(4) CFZ_BWD(3) CFZ_BWD(2) CFZ_BWD(1) CFZ_BWD(0)
#undef CFZ_BWD
  return 0;
}

template <int RB>
__global__ __launch_bounds__(512) void lu_mfma_kernel(const double *A, const double *B, double *X, int nblocks, int *fail, unsigned long long *cyc) {
  constexpr int N = 64, NT = 4 + RB / 16;
  __shared__ double lds_all[8 * (256 + 4 * 16 * NT + 64)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int blk = blockIdx.x * wpb + wave;
  if (blk >= nblocks) return;
  const int cj = lane & 15, rg = lane >> 4;
  v16d T[NT];
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * I + 4 * g + rg, col = 16 * J + cj;
        T[J][4 * I + g] = J < 4 ? A[((size_t)blk * N + row) * N + col] : B[((size_t)blk * N + row) * RB + (col - N)];
      }
  const long long t0 = wall_clock64();
  const int f = wave_lu_mfma<RB>(T, (lds_f64 *)(lds_all + wave * (256 + 4 * 16 * NT + 64)), X + (size_t)blk * N * RB, RB);
  const long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(cyc, (unsigned long long)(t1 - t0));
  if (f && lane == 0) atomicAdd(fail, 1);
}

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 8 * 256 * 4, wpb = argc > 2 ? atoi(argv[2]) : 8;
  constexpr int n = 64, RB = 32;
  std::vector<double> A((size_t)nb * n * n), B((size_t)nb * n * RB), X0((size_t)nb * n * RB), X1((size_t)nb * n * RB);
  srand(1);
  for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (size_t b = 0; b < (size_t)nb; b += 3)  // a third of the blocks: a KKT-like pattern (zero diagonal block)
    for (int i = 32; i < n; ++i) for (int j = 32; j < n; ++j) A[(b * n + i) * n + j] = (i == j) ? -1e-7 : 0.0;
  for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
  double *dA, *dB, *dX; int *df; unsigned long long *dc;
  OK(hipMalloc(&dA, A.size() * 8)); OK(hipMalloc(&dB, B.size() * 8)); OK(hipMalloc(&dX, X0.size() * 8)); OK(hipMalloc(&df, 4)); OK(hipMalloc(&dc, 8));
  OK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); OK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  const int grid = (nb + wpb - 1) / wpb;
  hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
  int cus = 0; OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  for (int variant = 0; variant < 2; ++variant) {
    float ms = 0;
    int fails = 0;
    unsigned long long ticks = 0;
    OK(hipMemset(dX, 0, X0.size() * 8));
    for (int rep = 0; rep < 3; ++rep) {
      OK(hipMemset(dc, 0, 8)); OK(hipMemset(df, 0, 4));
      OK(hipEventRecord(e0));
      if (variant == 0) hipLaunchKernelGGL((lu_regs_kernel<RB>), dim3(grid), dim3(64 * wpb), 0, 0, dA, dB, dX, nb, df, dc);
      else hipLaunchKernelGGL((lu_mfma_kernel<RB>), dim3(grid), dim3(64 * wpb), 0, 0, dA, dB, dX, nb, df, dc);
      OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1)); OK(hipEventElapsedTime(&ms, e0, e1));
    }
    OK(hipMemcpy((variant ? X1 : X0).data(), dX, X0.size() * 8, hipMemcpyDeviceToHost));
    OK(hipMemcpy(&fails, df, 4, hipMemcpyDeviceToHost)); OK(hipMemcpy(&ticks, dc, 8, hipMemcpyDeviceToHost));
    const std::vector<double> &X = variant ? X1 : X0;
    double worst = 0.0;
    for (int blk = 0; blk < nb; blk += nb / 7 + 1)
      for (int c = 0; c < RB; c += 5)
        for (int i = 0; i < n; ++i) {
          double s = -B[((size_t)blk * n + i) * RB + c];
          for (int j = 0; j < n; ++j) s += A[((size_t)blk * n + i) * n + j] * X[((size_t)blk * n + j) * RB + c];
          worst = fmax(worst, fabs(s));
        }
    printf("%s: 64 x 64 with %d right-hand sides, %d blocks, %d wavefronts per workgroup: launch %.3f ms = %.2f us per block per wavefront slot (%d CUs x %d), "
           "elimination alone %.1f us per block, failed %d, worst residual %.2e\n", variant ? "matrix cores" : "lane = row   ", RB, nb, wpb, ms,
           ms * 1e3 / ((double)nb / (cus * (double)wpb)), cus, wpb, ticks * 0.01 / nb, fails, worst);
  }
  size_t diff = 0;
  double dmax = 0.0;
  for (size_t i = 0; i < X0.size(); ++i) if (memcmp(&X0[i], &X1[i], 8) != 0) { ++diff; dmax = fmax(dmax, fabs(X0[i] - X1[i])); }
  printf("solutions that differ in any bit: %zu of %zu (largest difference %.3e)\n", diff, X0.size(), dmax);
  return 0;
}
