// Go / no-go for the MPC kernel's backward Riccati sweep on the matrix cores (docs/notebook.md, round 5).
//
// Today (cfz_solver.inl riccati_backward): ONE lane runs the 29 dependent stages, ~336 scalar FP64 instructions each, upper-triangle
// algebra on the structure of A = I + S: 50-56 k of an iteration's ~200 k cycles.  Here the stage is written in homogeneous coordinates,
//   [z+; 1] = T [z; 1; u],  T = [[A d B], [0 1 0]]  (6 x 8),   V+(z+) = 1/2 [z+; 1]' Pt [z+; 1],  Pt = [[P p], [p' 0]]  (6 x 6),
//   M = T' Pt T + Ht  (8 x 8: Ht the stage's Hessian and gradient),  Pt <- M_kk - M_ke M_ee^-1 M_ek  (k = z and 1, e = u),
// on v_mfma_f64_16x16x4_f64 (one wavefront, every lane takes part):
//   Y = Pt T      two k-steps; A operand = Pt: the accumulator registers of the stage before as they stand (lane l holds row (l >> 4) + 4 r,
//                 column l & 15 in register r; Pt is symmetric, so register s is A[row l & 15][k = (l >> 4) + 4 s]);
//   M = Tt' Y + Ht   two k-steps; B operand = Y's accumulator registers as they stand (register s is B[k = (l >> 4) + 4 s][column]);
//                 the accumulator starts as Ht; Tt' repeats the rows of u at rows 8, 9, 12, 13 so that every 16-lane group that needs
//                 M's rows of u holds them in its own registers (no movement between the groups):
//                     rows 6, 7 (groups 2, 3, register 1)  the diagonal block M_ee read by v_readlane;
//                     rows 8, 9 (groups 0, 1, register 2) and rows 12, 13 (groups 0, 1, register 3), u0 and u1 crossed over;
//   Pt <- M - U V    one k-step (k = 0, 1): A = -M_ke (groups 0 / 1: register 2), B = V = M_ee^-1 M_ek from registers 2 and 3.
// Five dependent matrix instructions, six v_readlane and one 2 x 2 inverse per stage; the gains are -V (groups 0, 1, columns 0..5).
// Prints cycles per stage of both versions and the largest difference of the gains.   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ricc tools/src/riccati_mfma_bench.hip && /tmp/ricc
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NS = 30;  // stages
// per stage in "LDS": T (6 x 8, row-major: 48), Ht (8 x 8 symmetric: 64) = 112 doubles; gains out 2 x 6

__device__ __forceinline__ double rl(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// variable of row i of Tt' / Ht's rows: 0..5 kept, 6 u0, 7 u1, then the repeats; -1 = a zero row
__device__ __forceinline__ int var_of_row(int i) { return i < 8 ? i : (i == 8 ? 6 : i == 9 ? 7 : i == 12 ? 7 : i == 13 ? 6 : -1); }

// reference: dense algebra of the same recursion on one lane
__device__ void stage_scalar(const double *T, const double *H, double P[6][6], double K[2][6]) {
  double Y[6][8], M[8][8];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += P[i][k] * T[k * 8 + j]; Y[i][j] = s; }
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s = H[i * 8 + j]; for (int k = 0; k < 6; ++k) s += T[k * 8 + i] * Y[k][j]; M[i][j] = s; }
  const double idet = 1.0 / (M[6][6] * M[7][7] - M[6][7] * M[7][6]);
  const double i00 = M[7][7] * idet, i01 = -M[6][7] * idet, i10 = -M[7][6] * idet, i11 = M[6][6] * idet;
  for (int j = 0; j < 6; ++j) { K[0][j] = -(i00 * M[6][j] + i01 * M[7][j]); K[1][j] = -(i10 * M[6][j] + i11 * M[7][j]); }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P[i][j] = M[i][j] + M[i][6] * K[0][j] + M[i][7] * K[1][j];
}

// the same stage with every sum in the order the matrix instructions are believed to use (k ascending, one fused multiply-add per k
// onto the accumulator) and the operands they really get (Pt read transposed; the rows of u taken from M's rows 6, 7): bitwise
// equality with the matrix-core sweep is what lets the CPU port mirror it
__device__ void stage_emul(const double *T, const double *H, double P[8][8], double K[2][6]) {
  double Y[8][8], M[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s = 0.0; for (int k = 0; k < 6; ++k) s = fma(P[k][i], T[k * 8 + j], s); Y[i][j] = s; }
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s = H[i * 8 + j]; for (int k = 0; k < 6; ++k) s = fma(T[k * 8 + i], Y[k][j], s); M[i][j] = s; }
  const double idet = 1.0 / fma(M[6][6], M[7][7], -(M[6][7] * M[7][6]));
  const double i00 = M[7][7] * idet, i01 = -M[6][7] * idet, i10 = -M[7][6] * idet, i11 = M[6][6] * idet;
  double V[2][8];
  for (int j = 0; j < 8; ++j) { V[0][j] = fma(i00, M[6][j], i01 * M[7][j]); V[1][j] = fma(i10, M[6][j], i11 * M[7][j]); }
  for (int j = 0; j < 6; ++j) { K[0][j] = -V[0][j]; K[1][j] = -V[1][j]; }
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) P[i][j] = fma(-M[7][i], V[1][j], fma(-M[6][i], V[0][j], M[i][j]));
}

__global__ __launch_bounds__(64) void bench(const double *data, double *gains_s, double *gains_m, double *gains_e, long long *cyc, int reps) {
  __shared__ double sd[NS * 112];
  __shared__ double kout[NS * 12];
  const int lane = threadIdx.x, lo = lane & 15, g = lane >> 4;
  for (int i = lane; i < NS * 112; i += 64) sd[i] = data[(size_t)blockIdx.x * NS * 112 + i];
  __syncthreads();
  // ---- scalar reference (lane 0) ----
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if (lane == 0) {
      double P[6][6];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P[i][j] = (i == j && i < 5) ? 1.0 : 0.0;
      for (int k = NS - 1; k >= 0; --k) {
        double K[2][6];
        stage_scalar(sd + k * 112, sd + k * 112 + 48, P, K);
        for (int j = 0; j < 6; ++j) { kout[k * 12 + j] = K[0][j]; kout[k * 12 + 6 + j] = K[1][j]; }
      }
    }
    __syncthreads();
  }
  long long t1 = clock64();
  for (int i = lane; i < NS * 12; i += 64) gains_s[(size_t)blockIdx.x * NS * 12 + i] = kout[i];
  __syncthreads();
  if (lane == 0) {
    double P[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) P[i][j] = (i == j && i < 5) ? 1.0 : 0.0;
    for (int k = NS - 1; k >= 0; --k) {
      double K[2][6];
      stage_emul(sd + k * 112, sd + k * 112 + 48, P, K);
      for (int j = 0; j < 6; ++j) { gains_e[(size_t)blockIdx.x * NS * 12 + k * 12 + j] = K[0][j]; gains_e[(size_t)blockIdx.x * NS * 12 + k * 12 + 6 + j] = K[1][j]; }
    }
  }
  __syncthreads();
  // ---- matrix cores ----
  // this lane's operand addresses inside a stage's 112 doubles (fixed over the sweep): B operand of Y = Pt T, A operand of M = Tt' Y,
  // the four accumulator entries of Ht
  int oB[2], oA[2], oH[4];
  for (int s = 0; s < 2; ++s) {
    const int k = g + 4 * s;
    oB[s] = (k < 6 && lo < 8) ? k * 8 + lo : -1;                       // T[k][lo]
    const int vr = var_of_row(lo);
    oA[s] = (k < 6 && vr >= 0) ? k * 8 + vr : -1;                      // Tt'[lo][k] = T[k][var(lo)]
  }
  for (int r = 0; r < 4; ++r) { const int vr = var_of_row(g + 4 * r); oH[r] = (vr >= 0 && lo < 8) ? 48 + vr * 8 + lo : -1; }
  long long t2 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
    v4d P = {0.0, 0.0, 0.0, 0.0};  // Pt in accumulator layout: rows g, g + 4 (registers 0, 1), column lo
    if (lo == g && g < 4) P[0] = 1.0;            // rows 0..3: identity
    if (lo == g + 4 && g == 0) P[1] = 1.0;       // row 4
    const double *st = sd + (NS - 1) * 112;
    double b0 = oB[0] >= 0 ? st[oB[0]] : 0.0, b1 = oB[1] >= 0 ? st[oB[1]] : 0.0, a0 = oA[0] >= 0 ? st[oA[0]] : 0.0, a1 = oA[1] >= 0 ? st[oA[1]] : 0.0;
    v4d H = {oH[0] >= 0 ? st[oH[0]] : 0.0, oH[1] >= 0 ? st[oH[1]] : 0.0, oH[2] >= 0 ? st[oH[2]] : 0.0, oH[3] >= 0 ? st[oH[3]] : 0.0};
    for (int k = NS - 1; k >= 0; --k) {
#ifndef RIC_VARIANT
#define RIC_VARIANT 0
#endif
      const v4d Z4 = {0.0, 0.0, 0.0, 0.0};
#if RIC_VARIANT == 1
      const v4d Ya = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], b0, Z4, 0, 0, 0);
      const v4d Yb = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], b1, Z4, 0, 0, 0);
      const v4d Y = Ya + Yb;
      const v4d Ma = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Y[0], H, 0, 0, 0);
      const v4d Mb = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Y[1], Z4, 0, 0, 0);
      v4d M = Ma + Mb;
#else
      v4d Y = Z4;
      Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], b0, Y, 0, 0, 0);
      Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], b1, Y, 0, 0, 0);
      v4d M = H;
      M = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Y[0], M, 0, 0, 0);
      M = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Y[1], M, 0, 0, 0);
#endif
      // the next stage's operands: issued now, used after this stage's last matrix instruction
      const double *sn = sd + (k > 0 ? k - 1 : 0) * 112;
      const double nb0 = oB[0] >= 0 ? sn[oB[0]] : 0.0, nb1 = oB[1] >= 0 ? sn[oB[1]] : 0.0, na0 = oA[0] >= 0 ? sn[oA[0]] : 0.0, na1 = oA[1] >= 0 ? sn[oA[1]] : 0.0;
      const v4d nH = {oH[0] >= 0 ? sn[oH[0]] : 0.0, oH[1] >= 0 ? sn[oH[1]] : 0.0, oH[2] >= 0 ? sn[oH[2]] : 0.0, oH[3] >= 0 ? sn[oH[3]] : 0.0};
      // M_ee: rows 6, 7 = groups 2, 3, register 1; columns 6, 7
#if RIC_VARIANT == 3
      const double m66 = 2.0 + M[1] * 1e-300, m67 = 0.1, m76 = 0.1, m77 = 3.0;
#else
      const double m66 = rl(M[1], 38), m67 = rl(M[1], 39), m76 = rl(M[1], 54), m77 = rl(M[1], 55);
#endif
#if RIC_VARIANT == 2
      const double idet = fma(m66, m77, -(m67 * m76)) * 0.25;
#elif RIC_VARIANT == 4
      const double det_ = fma(m66, m77, -(m67 * m76));
      double idet = __builtin_amdgcn_rcp(det_);
      idet = fma(fma(-det_, idet, 1.0), idet, idet); idet = fma(fma(-det_, idet, 1.0), idet, idet);
#else
      const double idet = 1.0 / fma(m66, m77, -(m67 * m76));
#endif
      const double i00 = m77 * idet, i01 = -m67 * idet, i10 = -m76 * idet, i11 = m66 * idet;
      // groups 0, 1: register 2 = row 8 / 9 (u0 / u1), register 3 = row 12 / 13 (u1 / u0)
      const double m6 = g == 0 ? M[2] : M[3], m7 = g == 0 ? M[3] : M[2];
      const double V = g == 0 ? fma(i00, m6, i01 * m7) : (g == 1 ? fma(i10, m6, i11 * m7) : 0.0);  // V[k = g][column lo]
      const double U = g < 2 ? -M[2] : 0.0;                                                 // -M_ke[row lo][k = g] = -M[6 + g][lo] (symmetry)
      M = __builtin_amdgcn_mfma_f64_16x16x4f64(U, V, M, 0, 0, 0);
      if (g < 2 && lo < 6) kout[k * 12 + g * 6 + lo] = -V;
      P = M; b0 = nb0; b1 = nb1; a0 = na0; a1 = na1; H = nH;
    }
    __syncthreads();
  }
  long long t3 = clock64();
  for (int i = lane; i < NS * 12; i += 64) gains_m[(size_t)blockIdx.x * NS * 12 + i] = kout[i];
  if (lane == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t3 - t2; }
}

int main(int argc, char **argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<double> data((size_t)blocks * NS * 112);
  srand(1);
  auto rnd = []() { return rand() / (double)RAND_MAX - 0.5; };
  for (int b = 0; b < blocks; ++b)
    for (int k = 0; k < NS; ++k) {
      double *T = data.data() + ((size_t)b * NS + k) * 112, *H = T + 48;
      for (int i = 0; i < 48; ++i) T[i] = 0.0;
      // A = I + small S, d, B as in the kinematic bicycle's sensitivities
      for (int i = 0; i < 5; ++i) T[i * 8 + i] = 1.0;
      T[0 * 8 + 2] = 0.1 * rnd(); T[0 * 8 + 3] = 0.1 + 0.02 * rnd(); T[0 * 8 + 4] = 0.01 * rnd();
      T[1 * 8 + 2] = 0.1 * rnd(); T[1 * 8 + 3] = 0.02 * rnd(); T[1 * 8 + 4] = 0.01 * rnd();
      T[2 * 8 + 3] = 0.03 * rnd(); T[2 * 8 + 4] = 0.05 + 0.01 * rnd();
      for (int i = 0; i < 5; ++i) T[i * 8 + 5] = 0.05 * rnd();                      // d
      T[5 * 8 + 5] = 1.0;                                                           // the constant stays the constant
      T[0 * 8 + 6] = 0.005 * rnd(); T[1 * 8 + 6] = 0.005 * rnd(); T[2 * 8 + 6] = 0.002 * rnd(); T[3 * 8 + 6] = 0.1;
      T[0 * 8 + 7] = 0.001 * rnd(); T[1 * 8 + 7] = 0.001 * rnd(); T[2 * 8 + 7] = 0.005 * rnd(); T[4 * 8 + 7] = 0.1;
      // Ht: positive definite Hessian over (z, u), gradient in row / column 5
      double L[8][8];
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) L[i][j] = (i == 5 || j == 5) ? 0.0 : (i == j ? 1.0 + fabs(rnd()) : (j < i ? 0.3 * rnd() : 0.0));
      for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s = 0; for (int q = 0; q < 8; ++q) s += L[i][q] * L[j][q]; H[i * 8 + j] = s; }
      for (int i = 0; i < 8; ++i) if (i != 5) { const double gi = rnd(); H[i * 8 + 5] = gi; H[5 * 8 + i] = gi; }
      H[5 * 8 + 5] = 0.0;
    }
  double *dd, *gs, *gm, *ge; long long *dc;
  OK(hipMalloc(&dd, data.size() * 8)); OK(hipMalloc(&gs, (size_t)blocks * NS * 12 * 8)); OK(hipMalloc(&gm, (size_t)blocks * NS * 12 * 8)); OK(hipMalloc(&ge, (size_t)blocks * NS * 12 * 8)); OK(hipMalloc(&dc, blocks * 16));
  OK(hipMemcpy(dd, data.data(), data.size() * 8, hipMemcpyHostToDevice));
  for (int it = 0; it < 2; ++it) { hipLaunchKernelGGL(bench, dim3(blocks), dim3(64), 0, 0, dd, gs, gm, ge, dc, reps); OK(hipDeviceSynchronize()); }
  std::vector<double> hs((size_t)blocks * NS * 12), hm(hs.size()), he(hs.size()); std::vector<long long> hc(blocks * 2);
  OK(hipMemcpy(hs.data(), gs, hs.size() * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(hm.data(), gm, hm.size() * 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(he.data(), ge, he.size() * 8, hipMemcpyDeviceToHost));
  OK(hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost));
  double dmax = 0.0, gmax = 0.0;
  for (size_t i = 0; i < hs.size(); ++i) { dmax = fmax(dmax, fabs(hs[i] - hm[i])); gmax = fmax(gmax, fabs(hs[i])); }
  double emax = 0.0; size_t nbit = 0;
  for (size_t i = 0; i < hs.size(); ++i) { emax = fmax(emax, fabs(he[i] - hm[i])); nbit += he[i] != hm[i]; }
  printf("gains: largest |scalar - mfma| %.3e (largest gain %.3e); against the emulation of the matrix instructions' order (k ascending, fused multiply-adds): %.3e, %zu of %zu values differ\n", dmax, gmax, emax, nbit, hs.size());
  printf("block 0: scalar dense reference %.0f cycles per stage, matrix cores %.0f cycles per stage (%d stages, %d repetitions; today's hand-written scalar sweep: ~1850)\n",
         hc[0] / (double)(reps * NS), hc[1] / (double)(reps * NS), NS, reps);
  return 0;
}
