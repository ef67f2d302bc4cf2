// Dense block eliminations of the structured planning paths (included by cfz_colloc.inl; used by cfz_jstruct.inl).
//
// Round 4's structured elimination of the single-vehicle plan lived here (interiors of 64 unknowns, separators of 14-31, a separator
// recursion whose wavefronts handed blocks over through global memory behind wavefront fences); since round 5 every plan goes through
// cfz_jstruct.inl's scheme (separators of at most 15 unknowns, every hand-off in registers) and round 6 removed the old path (VERDICT r5
// item 6: its fenced hand-offs were the idiom of a race that had been "caught, not understood").  What remains is what the joint scheme
// builds on: the serial dense solve of the CPU build, the register elimination (lane = row, v_readlane broadcasts) and the 64-row
// elimination on the matrix cores (lu64_*; tools/src/wave_lu_mfma_bench.hip, docs/notebook.md round 5).
#pragma once

namespace cfzc {

constexpr int kSI = 64;             // unknowns of an interior
CFZP_FN double band_at(const Band &B, int n, int i, int j) {
  const int dd = i - j;
  return (i >= 0 && j >= 0 && i < n && j < n && dd <= B.kb && -dd <= B.kb) ? B.ab[(size_t)j * B.ld + (B.off + dd)] : 0.0;
}

// dense elimination with partial pivoting of the n x n block in aug[n][ld] with nrhs right-hand sides behind it (ld >= n + nrhs);
// the solution replaces the right-hand sides.  0 = ok.  (The CPU build, and the GPU's reference path: ONE lane runs it.)
CFZP_FN int block_solve_serial(double *aug, int n, int ld, int nrhs) {
  for (int k = 0; k < n; ++k) {
    int p = k; double best = fabs(aug[k * ld + k]);
    for (int i = k + 1; i < n; ++i) { const double a = fabs(aug[i * ld + k]); if (a > best) { best = a; p = i; } }
    if (!(best > 0.0)) return 1;
    if (p != k) for (int j = k; j < n + nrhs; ++j) { const double t = aug[k * ld + j]; aug[k * ld + j] = aug[p * ld + j]; aug[p * ld + j] = t; }
    const double inv = 1.0 / aug[k * ld + k];
    for (int i = k + 1; i < n; ++i) {
      const double l = aug[i * ld + k] * inv;
      if (l != 0.0) for (int j = k + 1; j < n + nrhs; ++j) aug[i * ld + j] -= l * aug[k * ld + j];
    }
  }
  for (int c = n; c < n + nrhs; ++c)
    for (int k = n - 1; k >= 0; --k) {
      double t = aug[k * ld + c];
      for (int j = k + 1; j < n; ++j) t -= aug[k * ld + j] * aug[j * ld + c];
      aug[k * ld + c] = t / aug[k * ld + k];
    }
  return 0;
}

// LDS of the 64-row eliminations on the matrix cores (lu64_build below; the host sizes the dynamic LDS with it: cfz_planning.hip)
constexpr int kLuStage = 64 * 17;              // doubles: lane = row -> tiles, sixteen columns at a time (rows padded to 17: no bank conflicts)
constexpr int kLuWork = 256 + 4 * 16 * 6 + 64;  // the panel copy, four pivot rows / right-hand sides of six column tiles, the reciprocal pivots
constexpr int kLuLdsWave = kLuStage > kLuWork ? kLuStage : kLuWork;  // doubles of LDS per wavefront (8.5 KB: the elimination's work area takes the staging area's place)
#if defined(__HIP_DEVICE_COMPILE__)
// The GPU's block elimination: lane r holds row r of [A | B] in registers (NB + RB doubles), every loop unrolled (static register
// indices), the pivot by a butterfly over the candidates, the pivot row broadcast entry by entry with v_readlane -- no memory inside
// a pivot step (tools/src/wave_lu_bench.hip: 80-120 us for 64 x 64 with 22 right-hand sides, against 230 us with the block in LDS and
// 3.4 us per pivot of the band elimination).  Functions of their own (their registers are theirs); they name no LDS.
__device__ __forceinline__ double struct_lane_get(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int NB, int RB>
__device__ __forceinline__ int wave_lu_regs(double (&a)[NB + RB], int lane, int &ord) {
  bool done = lane >= NB;
  ord = -1;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const double best = done ? -1.0 : fabs(a[k]);
    const double m = cfz::wave_reduce<1>(best);  // DPP row operations and four v_readlane: no trip through the LDS crossbar
    if (!(m > 0.0)) return 1;
    const int pl = (int)__builtin_ctzll(__ballot(best == m));  // the first row holding the largest entry, as the serial search
    const double inv = 1.0 / struct_lane_get(a[k], pl);
    const bool mine = lane == pl;
    const double l = (done || mine) ? 0.0 : a[k] * inv;
    if (mine) { done = true; ord = k; }
#pragma unroll
    for (int j = k + 1; j < NB + RB; ++j) a[j] -= l * struct_lane_get(a[j], pl);
  }
#pragma unroll
  for (int k = NB - 1; k >= 0; --k) {
    const int pl = (int)__builtin_ctzll(__ballot(ord == k));
    const double inv = 1.0 / struct_lane_get(a[k], pl);
    const double u = (ord >= 0 && ord < k) ? a[k] : 0.0;
#pragma unroll
    for (int c = 0; c < RB; ++c) {
      const double x = struct_lane_get(a[NB + c], pl) * inv;
      a[NB + c] = lane == pl ? x : a[NB + c] - u * x;
    }
  }
  return 0;
}
// ---- The 64-row eliminations on the matrix cores (round 5; tools/src/wave_lu_mfma_bench.hip is the study: 80 -> 45 us for a wavefront alone
// on its CU, 122 -> 65 us with two per SIMD, the solutions equal BIT FOR BIT).  The same elimination -- the same pivots (largest entry of
// the column among the rows not yet used, the first of equals), the same multipliers, every product applied in the same order -- BLOCKED
// four pivots at a time, the block [A | B] in the accumulator layout of v_mfma_f64_16x16x4_f64: column tile J of lane l is ONE vector of
// sixteen doubles, T[J][4 I + g] = row 16 I + 4 g + (l >> 4), column 16 J + (l & 15) (registers 4 I .. 4 I + 3 are the accumulator of tile
// (I, J)).  Rows never move (the multipliers of finished rows are zero).  Per panel:
//   (1) the panel's four columns go through LDS to a lane = row copy, where the four pivots are chosen and the panel factored as in
//       wave_lu_regs (v_readlane of four values instead of ninety);
//   (2) the negated multipliers go through LDS to the matrix instruction's first-operand layout (lane l: row 16 I + (l & 15), pivot l >> 4);
//       pivot row p is element p >> 2 of the lanes (l >> 4) == (p & 3) of every column tile -- a dynamic but UNIFORM element index
//       (s_set_gpr_idx_on / v_mov) -- and goes through LDS to all lanes of its column, which bring the four rows up to date with the
//       pivots before them in the panel, zero the columns up to each row's own pivot and keep the row of their lane group: the second
//       operand;
//   (3) one matrix instruction per tile: c - l0 u0 - l1 u1 - l2 u2 - l3 u3, accumulated in that order with one rounding per product
//       (as the fused multiply-adds of the unblocked loop).
// The back-substitution takes four unknowns at a time from the last the same way (the panel's triangle solved in every lane, the rows
// pivoted earlier updated by one matrix instruction per tile with the operands in descending order of the unknowns); the solution goes
// straight to memory, row = unknown.  A panel is a function of a template parameter, not a loop iteration: a sixteen-trip loop of this
// size is beyond the unroller's budget, and a tile array indexed by a loop variable would live in scratch memory.
typedef double lu_v4 __attribute__((ext_vector_type(4)));
typedef double lu_v16 __attribute__((ext_vector_type(16)));
// the lane index as a value the optimiser cannot see through: inside a loop over tasks it keeps per-lane constants of a builder (sixty-four
// "is this my diagonal entry" doubles ...) from being hoisted out of the loop and spilled for the whole function
__device__ __forceinline__ int lu_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void lu_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ double lu_uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void lu_tile_mfma(lu_v16 &t, int I, double a, double b) {
  lu_v4 c = {t[4 * I], t[4 * I + 1], t[4 * I + 2], t[4 * I + 3]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  t[4 * I] = c[0]; t[4 * I + 1] = c[1]; t[4 * I + 2] = c[2]; t[4 * I + 3] = c[3];
}
template <int RB, int P>
__device__ __forceinline__ int lu64_forward_panel(lu_v16 (&T)[4 + RB / 16], cfzb::lds_f64 *Pb, cfzb::lds_f64 *Ub, bool &done, int &ord) {
  constexpr int NT = 4 + RB / 16, k0 = 4 * P, Jp = P >> 2, c0 = k0 & 15;
  const int lane = threadIdx.x & 63, cj = lane & 15, rg = lane >> 4;
  if (cj >= c0 && cj < c0 + 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Pb[(4 * e + rg) * 4 + (cj - c0)] = T[Jp][e];
  }
  lu_wave_sync();
  double a[4], l[4];
  int pl[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = Pb[lane * 4 + q];
  lu_wave_sync();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double best = done ? -1.0 : fabs(a[q]);
    const double m = cfz::wave_reduce<1>(best);
    if (!(m > 0.0)) return 1;
    pl[q] = (int)__builtin_ctzll(__ballot(best == m));
    const double inv = 1.0 / struct_lane_get(a[q], pl[q]);
    const bool mine = lane == pl[q];
    l[q] = (done || mine) ? 0.0 : a[q] * inv;
    if (mine) { done = true; ord = k0 + q; Ub[4 * 16 * 6 + k0 + q] = inv; }  // (the reciprocal pivot, kept for the back-substitution)
#pragma unroll
    for (int s = q + 1; s < 4; ++s) a[s] -= l[q] * struct_lane_get(a[s], pl[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) Pb[lane * 4 + q] = -l[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = __builtin_amdgcn_readfirstlane(pl[q] >> 2);
    double u[NT];
#pragma unroll
    for (int J = Jp; J < NT; ++J) u[J] = T[J][e];
    if (rg == (pl[q] & 3)) {
#pragma unroll
      for (int J = Jp; J < NT; ++J) Ub[q * 16 * NT + 16 * J + cj] = u[J];
    }
  }
  lu_wave_sync();
  double Aop[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) Aop[I] = Pb[(16 * I + cj) * 4 + rg];
  const double l10 = struct_lane_get(l[0], pl[1]), l20 = struct_lane_get(l[0], pl[2]), l21 = struct_lane_get(l[1], pl[2]);
  const double l30 = struct_lane_get(l[0], pl[3]), l31 = struct_lane_get(l[1], pl[3]), l32 = struct_lane_get(l[2], pl[3]);
#pragma unroll
  for (int J = Jp; J < NT; ++J) {  // a column tile at a time: four values live
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
    for (int I = 0; I < 4; ++I) lu_tile_mfma(T[J], I, Aop[I], b);
  }
  lu_wave_sync();
  return 0;
}
// out[k * SK + c * SC]: unknown k, column c of the solution (OUT: a global or an LDS pointer)
template <int RB, int P, int SK, int SC, class OUT>
__device__ __forceinline__ void lu64_backward_panel(lu_v16 (&T)[4 + RB / 16], cfzb::lds_f64 *Pb, cfzb::lds_f64 *Ub, int ord, OUT *out) {
  constexpr int NT = 4 + RB / 16, NR = RB / 16, k0 = 4 * P, Jp = P >> 2, c0 = k0 & 15;
  const int lane = threadIdx.x & 63, cj = lane & 15, rg = lane >> 4;
  if (cj >= c0 && cj < c0 + 4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Pb[(4 * e + rg) * 4 + (cj - c0)] = T[Jp][e];
  }
  lu_wave_sync();
  double a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = Pb[lane * 4 + q];
  lu_wave_sync();
  int pk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pk[i] = (int)__builtin_ctzll(__ballot(ord == k0 + i));
    const int e = __builtin_amdgcn_readfirstlane(pk[i] >> 2);
    double u[NT];
#pragma unroll
    for (int Jr = 0; Jr < NR; ++Jr) u[4 + Jr] = T[4 + Jr][e];
    if (rg == (pk[i] & 3)) {
#pragma unroll
      for (int Jr = 0; Jr < NR; ++Jr) Ub[i * 16 * NT + 16 * Jr + cj] = u[4 + Jr];
    }
  }
  // rows pivoted before this panel: right-hand sides -= (their entries in the panel's columns) x (the four solutions), last column first
  const bool early = ord < k0;
#pragma unroll
  for (int i = 0; i < 4; ++i) Pb[lane * 4 + i] = early ? -a[3 - i] : 0.0;
  lu_wave_sync();
  double Aop[4];
#pragma unroll
  for (int I = 0; I < 4; ++I) Aop[I] = Pb[(16 * I + cj) * 4 + rg];
  const double i3 = lu_uni(Ub[4 * 16 * 6 + k0 + 3]), i2 = lu_uni(Ub[4 * 16 * 6 + k0 + 2]), i1 = lu_uni(Ub[4 * 16 * 6 + k0 + 1]), i0 = lu_uni(Ub[4 * 16 * 6 + k0]);
  const double u23 = struct_lane_get(a[3], pk[2]), u13 = struct_lane_get(a[3], pk[1]), u12 = struct_lane_get(a[2], pk[1]);
  const double u03 = struct_lane_get(a[3], pk[0]), u02 = struct_lane_get(a[2], pk[0]), u01 = struct_lane_get(a[1], pk[0]);
#pragma unroll
  for (int Jr = 0; Jr < NR; ++Jr) {
    double x0 = Ub[0 * 16 * NT + 16 * Jr + cj], x1 = Ub[1 * 16 * NT + 16 * Jr + cj], x2 = Ub[2 * 16 * NT + 16 * Jr + cj], x3 = Ub[3 * 16 * NT + 16 * Jr + cj];
    x3 *= i3;
    x2 -= u23 * x3; x2 *= i2;
    x1 -= u13 * x3; x1 -= u12 * x2; x1 *= i1;
    x0 -= u03 * x3; x0 -= u02 * x2; x0 -= u01 * x1; x0 *= i0;
    out[(k0 + rg) * SK + (16 * Jr + cj) * SC] = rg == 0 ? x0 : (rg == 1 ? x1 : (rg == 2 ? x2 : x3));
    if (P > 0) {
      const double b = rg == 3 ? x0 : (rg == 2 ? x1 : (rg == 1 ? x2 : x3));
#pragma unroll
      for (int I = 0; I < 4; ++I) lu_tile_mfma(T[4 + Jr], I, Aop[I], b);
    }
  }
  lu_wave_sync();
}
template <int RB, int SK, int SC, class OUT>
__device__ __forceinline__ int lu64_tiles(lu_v16 (&T)[4 + RB / 16], cfzb::lds_f64 *work, OUT *out) {
  static_assert(RB == 16 || RB == 32, "one or two column tiles of right-hand sides");
  cfzb::lds_f64 *Pb = work, *Ub = work + 256;
  bool done = false;
  int ord = -1;
#define CFZ_LU_FWD(P) if (lu64_forward_panel<RB, P>(T, Pb, Ub, done, ord)) return 1;
  CFZ_LU_FWD(0) CFZ_LU_FWD(1) CFZ_LU_FWD(2) CFZ_LU_FWD(3) CFZ_LU_FWD(4) CFZ_LU_FWD(5) CFZ_LU_FWD(6) CFZ_LU_FWD(7)
  CFZ_LU_FWD(8) CFZ_LU_FWD(9) CFZ_LU_FWD(10) CFZ_LU_FWD(11) CFZ_LU_FWD(12) CFZ_LU_FWD(13) CFZ_LU_FWD(14) CFZ_LU_FWD(15)
#undef CFZ_LU_FWD
#define CFZ_LU_BWD(P) lu64_backward_panel<RB, P, SK, SC, OUT>(T, Pb, Ub, ord, out);
  CFZ_LU_BWD(15) CFZ_LU_BWD(14) CFZ_LU_BWD(13) CFZ_LU_BWD(12) CFZ_LU_BWD(11) CFZ_LU_BWD(10) CFZ_LU_BWD(9) CFZ_LU_BWD(8)
  CFZ_LU_BWD(7) CFZ_LU_BWD(6) CFZ_LU_BWD(5) CFZ_LU_BWD(4) CFZ_LU_BWD(3) CFZ_LU_BWD(2) CFZ_LU_BWD(1) CFZ_LU_BWD(0)
#undef CFZ_LU_BWD
  return 0;
}
// ... from a lane = row builder: fill(J, c) leaves the sixteen entries of row = lane in columns 16 J .. 16 J + 15 of [A | B] in c (J a
// std::integral_constant); they go through LDS into the tile layout, a column tile at a time (sixteen values live, never the whole row beside
// the tiles), then the elimination; out[k * SK + c * SC] = (A^-1 B)[k][c] (default: column-major, 64 rows).  lds: this wavefront's
// kLuLdsWave doubles.
template <int J, int NT, class F>
__device__ __forceinline__ void lu64_fill(lu_v16 (&T)[NT], cfzb::lds_f64 *lds, F &fill) {
  if constexpr (J < NT) {
    const int lane = threadIdx.x & 63, cj = lane & 15, rg = lane >> 4;
    double c[16];
    fill(std::integral_constant<int, J>{}, c);
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[lane * 17 + k] = c[k];
    lu_wave_sync();
#pragma unroll
    for (int e = 0; e < 16; ++e) T[J][e] = lds[(4 * e + rg) * 17 + cj];
    lu_wave_sync();
    lu64_fill<J + 1, NT>(T, lds, fill);
  }
}
template <int RB, int SK = 1, int SC = 64, class OUT = cfzb::glb_f64, class F>
__device__ __forceinline__ int lu64_build(cfzb::lds_f64 *lds, OUT *out, F fill) {
  lu_v16 T[4 + RB / 16];
  lu64_fill<0, 4 + RB / 16>(T, lds, fill);
  return lu64_tiles<RB, SK, SC, OUT>(T, lds, out);  // (the staging area is free: every lane has read its tiles)
}
#endif

// which wavefront / lane of the workgroup this is (the CPU build: one "wavefront" of one lane)
#if defined(__HIP_DEVICE_COMPILE__)
#define CFZS_WAVE ((int)(threadIdx.x >> 6))
#define CFZS_NW ((int)(blockDim.x >> 6))
#define CFZS_LANE ((int)(threadIdx.x & 63))
#define CFZS_FIRST_OF_WAVE (CFZS_LANE == 0)
#else
#define CFZS_WAVE 0
#define CFZS_NW 1
#define CFZS_LANE 0
#define CFZS_FIRST_OF_WAVE true
#endif

}  // namespace cfzc
