// Structured elimination of the single-vehicle collocation plan's Newton system (included by cfz_colloc.inl; docs/notebook.md, round 4).
//
// The band matrix `assemble` fills is eliminated interval by interval instead of pivot by pivot: in the band ordering an interval is
// [continuity | tube | pt0] [pt1 pt2 | 30 ODE rows | pt3 pt4 pt5]; with the steering rate of pt5 counted to the NEXT interval's head the
// second bracket is an INTERIOR of exactly 64 unknowns that couples to 7 unknowns on its left (pt0) and at most 14 on its right (that
// steering rate, the next continuity rows; for the last interval the end point's tube rows and the terminal rows), and what lies
// between two interiors is a SEPARATOR of 14-31 unknowns.  Phase 1: every interior independently, dense with partial pivoting, with its
// coupling columns and the two right-hand sides as right-hand sides.  Phase 2: the Schur complements onto the separators.  Phase 3: the
// separator system is block tridiagonal: a recursion over the intervals (dense, pivoted within a block).  Phase 4: back-substitution.
// Another elimination ORDER of the same matrix: tools/colloc_condense_study.py and tests/test_colloc.py pin pattern and accuracy.
#pragma once

namespace cfzc {

constexpr int kSI = 64;             // unknowns of an interior
constexpr int kSL = 7, kSRt = 14;   // coupling columns on the left / on the right
constexpr int kSR = 24;             // right-hand sides of an interior: 7 + 14 coupling columns, 2 right-hand sides, 1 spare
constexpr int kSS = 32;             // separator block, padded (<= 31 unknowns)
constexpr int kSZ = 16;             // right-hand sides of a separator block: <= 14 coupling columns + 2
constexpr int kSWi = kSI * (kSL + kSRt) + kSI * kSR;  // doubles per interval: C (64 x 21), W (64 x 24)
constexpr int kSWs = kSS * kSS + 4 * kSS * kSZ;       // doubles per separator: D (32 x 32), UR, Z and their bottom-up twins (32 x 16 each)

struct SWork { double *Ci, *Wi, *Ds, *Us, *Zs, *Ub, *Zb, *aug, *flag; int *cl, *ps; };

CFZP_FN size_t struct_doubles(const CSpec &sp) {
  if (sp.V != 1) return 0;
  const int N = sp.N[0];
  return (size_t)N * kSWi + (size_t)(N + 1) * kSWs + (size_t)kSI * (kSI + kSR) * 8 + (size_t)(N * 24 + 2 * (N + 2) + 3) / 2 + 16;
}
// doubles the structured elimination of one single-vehicle Newton system moves between its phases (what bench.py prices the kernel's HBM
// traffic with): the band cleared and gathered once; C written and read by the Schur products; W written, read by the Schur products and by
// the back-substitution; the separator blocks, their right-hand sides and solutions written and read once, the upward twins for half of them
CFZP_FN size_t struct_alg_doubles(const CSpec &sp, size_t nk, size_t ld) {
  const size_t N = sp.N[0];
  return 2 * nk * ld + 2 * N * kSI * (kSL + kSRt) + 3 * N * kSI * kSR + 2 * (N + 1) * kSS * kSS + 4 * (N + 1) * kSS * kSZ + (N + 1) * kSS * kSZ + 4 * nk;
}
CFZP_FN SWork struct_carve(const CSpec &sp, double *p) {
  const int N = sp.N[0];
  SWork s;
  s.Ci = p; p += (size_t)N * kSI * (kSL + kSRt); s.Wi = p; p += (size_t)N * kSI * kSR;
  s.Ds = p; p += (size_t)(N + 1) * kSS * kSS; s.Us = p; p += (size_t)(N + 1) * kSS * kSZ; s.Zs = p; p += (size_t)(N + 1) * kSS * kSZ;
  s.Ub = p; p += (size_t)(N + 1) * kSS * kSZ; s.Zb = p; p += (size_t)(N + 1) * kSS * kSZ;
  s.aug = p; p += (size_t)kSI * (kSI + kSR) * 8;  // one staging area per wavefront (eight)
  s.flag = p; p += 2;
  s.cl = reinterpret_cast<int *>(p); s.ps = s.cl + N * 24;
  return s;
}

// positions: ps[i] = first position of separator i (i = 0..N), ps[N + 1] = nk; the interior of interval i is [pi, pi + 64), pi = ps[i + 1] - 64
// cl[24 i + q]: q < 7 the positions of pt0 of interval i; 7 <= q < 21 the coupled positions of separator i + 1, -1 = none
CFZP_FN void struct_setup(const CSpec &sp, const CDims &d, const CWork &w, const SWork &s) {
  const int N = sp.N[0];
  CFZP_LANE_FOR(one, 0, 0) s.flag[1] = 0.0;
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, N - 1) {
    const int pe = w.posx[7 * (kPts * i + 5) + 6];  // the steering rate of the interval's last point: first position of separator i + 1
    s.ps[i + 1] = pe;
    if (i == 0) { s.ps[0] = 0; s.ps[N + 1] = d.nk; }
    int *cl = s.cl + 24 * i;
    for (int c = 0; c < 7; ++c) cl[c] = w.posx[7 * (kPts * i) + c];
    int q = 7;
    cl[q++] = pe;
    if (i + 1 < N) for (int c = 0; c < 7; ++c) cl[q++] = w.posc[d.rC + 7 * i + c];
    else {
      for (int T = d.coff[0]; T < d.coff[1]; ++T)
        if (chk_point(sp, d, T) == kPts * N - 1) for (int r = 0; r < 8; ++r) if (q < 21) cl[q++] = w.posc[d.rT + 8 * T + r];
      for (int c = 0; c < 5; ++c) if (w.posc[d.rF + c] >= 0 && q < 21) cl[q++] = w.posc[d.rF + c];
    }
    while (q < 24) cl[q++] = -1;
    // the layout the phases below assume (a changed ordering or row set must not turn into a wrong Newton step): interiors of
    // exactly 64 unknowns behind pt0, separators of 1..31 unknowns, the right coupling list complete (the loops above stop at 21 entries)
    bool ok = pe == w.posx[7 * (kPts * i + 1)] + kSI && cl[0] == w.posx[7 * (kPts * i + 1)] - 7 && cl[6] == cl[0] + 6;
    const int lo = i == 0 ? 0 : w.posx[7 * (kPts * (i - 1) + 5) + 6];  // first position of separator i
    ok = ok && cl[0] - lo + 7 >= 1 && cl[0] - lo + 7 <= kSS - 1;
    if (i + 1 == N) {
      int need = 1;
      for (int T = d.coff[0]; T < d.coff[1]; ++T) if (chk_point(sp, d, T) == kPts * N - 1) need += 8;
      for (int c = 0; c < 5; ++c) if (w.posc[d.rF + c] >= 0) ++need;
      ok = ok && need <= kSRt && d.nk - pe >= 1 && d.nk - pe <= kSS - 1;
    }
    if (!ok) s.flag[1] = 1.0;
  }
  CFZP_SYNC();
}

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
// interior of one interval: rows gathered from the band, coupling columns kept in C, K_II^-1 [C | b1 b2] to W (row = unknown).
// NC coupling columns are live (15 for all intervals but the last: pt0, the steering rate, the next continuity rows; 21 for the last);
// the right-hand sides follow them in the registers and go to W's columns 21 and 22 either way.
template <int NC>
__device__ __attribute__((noinline)) int struct_interior(const cfzb::glb_f64 *ab, int kb, int ld, int off, int pi, const cfzb::glb_i32 *cl,
                                                         const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *C, cfzb::glb_f64 *W) {
  constexpr int RB = NC + 2 + (NC & 1);  // (even, as the elimination's template was measured)
  const int lane = threadIdx.x & 63, r = pi + lane;
  double a[kSI + RB];
#pragma unroll
  for (int j = 0; j < kSI; ++j) { const int c = pi + j, dd = r - c; a[j] = (dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0; }
#pragma unroll
  for (int q = 0; q < kSL + kSRt; ++q) {
    const int c = q < NC ? cl[q] : -1, dd = r - c;
    const double v = (c >= 0 && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0;
    if (q < NC) a[kSI + q] = v;
    C[q * kSI + lane] = v;
  }
  a[kSI + NC] = b1[r]; a[kSI + NC + 1] = b2[r];
  if (NC & 1) a[kSI + NC + 2] = 0.0;
  int ord;
  if (wave_lu_regs<kSI, RB>(a, lane, ord)) return 1;
#pragma unroll
  for (int q = 0; q < kSL + kSRt; ++q) W[q * kSI + ord] = q < NC ? a[kSI + q] : 0.0;
  W[21 * kSI + ord] = a[kSI + NC]; W[22 * kSI + ord] = a[kSI + NC + 1];
  return 0;
}
// one separator block: D (32 x 32 in memory, identity-padded), [U | r1 r2] (32 x 16) -> Z = D^-1 [U | r].  NS = 16 for the separators of
// at most 15 unknowns (four in five): half the pivot steps
template <int NS>
__device__ __attribute__((noinline)) int struct_separator_n(const cfzb::glb_f64 *D, const cfzb::glb_f64 *U, cfzb::glb_f64 *Z) {
  const int lane = threadIdx.x & 63, r = lane < NS ? lane : NS - 1;
  double a[NS + kSZ];
#pragma unroll
  for (int j = 0; j < NS; ++j) a[j] = D[r * kSS + j];
#pragma unroll
  for (int q = 0; q < kSZ; ++q) a[NS + q] = U[r * kSZ + q];
  int ord;
  if (wave_lu_regs<NS, kSZ>(a, lane, ord)) return 1;
  if (lane < NS) {
#pragma unroll
    for (int q = 0; q < kSZ; ++q) Z[ord * kSZ + q] = a[NS + q];
  }
  return 0;
}
__device__ __forceinline__ int struct_separator(const cfzb::glb_f64 *D, const cfzb::glb_f64 *U, cfzb::glb_f64 *Z, int ns) {
  if (ns <= 16) {
    const int f = struct_separator_n<16>(D, U, Z);
    if (!f) { const int lane = threadIdx.x & 63; if (lane >= 16 && lane < kSS) { for (int q = 0; q < kSZ; ++q) Z[lane * kSZ + q] = 0.0; } }  // the padding rows' (zero) solution
    return f;
  }
  return struct_separator_n<kSS>(D, U, Z);
}
#endif

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

// The whole solve: on return b1, b2 (band positions) hold the two solutions.  0 = ok, 1 = a block was singular.
CFZP_FN int struct_solve(const CSpec &sp, const CDims &d, const CWork &w, const SWork &s, const Band &B, double *b1, double *b2, long long *ptk) {
  double *flag = s.flag;
  long long tp = tick();  // ptk[0..2]: interiors, Schur complements, separator recursion + back-substitution (device clock)
  const int N = sp.N[0], nk = d.nk, ldi = kSI + kSR;
  CFZP_LANE_FOR(one, 0, 0) flag[0] = 0.0;
  CFZP_SYNC();
  // ---- phase 1: interiors (a wavefront each) ---------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = CFZS_WAVE; i < N; i += CFZS_NW) {
    const cfzb::glb_f64 *ab_ = (const cfzb::glb_f64 *)B.ab, *b1_ = (const cfzb::glb_f64 *)b1, *b2_ = (const cfzb::glb_f64 *)b2;
    const cfzb::glb_i32 *cl_ = (const cfzb::glb_i32 *)(s.cl + 24 * i);
    cfzb::glb_f64 *C_ = (cfzb::glb_f64 *)(s.Ci + (size_t)i * kSI * (kSL + kSRt)), *W_ = (cfzb::glb_f64 *)(s.Wi + (size_t)i * kSI * kSR);
    const int f = i + 1 < N ? struct_interior<15>(ab_, B.kb, B.ld, B.off, s.ps[i + 1] - kSI, cl_, b1_, b2_, C_, W_)
                            : struct_interior<kSL + kSRt>(ab_, B.kb, B.ld, B.off, s.ps[i + 1] - kSI, cl_, b1_, b2_, C_, W_);
    if (f && CFZS_LANE == 0) flag[0] = 1.0;
  }
#else
  for (int i = CFZS_WAVE; i < N; i += CFZS_NW) {
    const int pi = s.ps[i + 1] - kSI;
    const int *cl = s.cl + 24 * i;
    double *aug = s.aug + (size_t)CFZS_WAVE * kSI * ldi, *C = s.Ci + (size_t)i * kSI * (kSL + kSRt), *W = s.Wi + (size_t)i * kSI * kSR;
#if defined(__HIP_DEVICE_COMPILE__)
    for (int r = CFZS_LANE; r < kSI; r += 64)
#else
    for (int r = 0; r < kSI; ++r)
#endif
    {
      for (int j = 0; j < kSI; ++j) aug[r * ldi + j] = band_at(B, nk, pi + r, pi + j);
      for (int q = 0; q < kSL + kSRt; ++q) { const double v = cl[q] >= 0 ? band_at(B, nk, pi + r, cl[q]) : 0.0; aug[r * ldi + kSI + q] = v; C[q * kSI + r] = v; }
      aug[r * ldi + kSI + 21] = b1[pi + r]; aug[r * ldi + kSI + 22] = b2[pi + r]; aug[r * ldi + kSI + 23] = 0.0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
#endif
    if (CFZS_FIRST_OF_WAVE) { if (block_solve_serial(aug, kSI, ldi, kSR)) flag[0] = 1.0; }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int r = CFZS_LANE; r < kSI; r += 64)
#else
    for (int r = 0; r < kSI; ++r)
#endif
      for (int q = 0; q < kSR; ++q) W[q * kSI + r] = aug[r * ldi + kSI + q];
  }
#endif
  CFZP_SYNC();
  { const long long t1 = tick(); ptk[0] += t1 - tp; tp = t1; }
  if (flag[0] != 0.0) return 1;
  // ---- phase 2: separator blocks from the band, minus the interiors' Schur complements ------------------------------------------
  CFZP_LANE_FOR(t, 0, (N + 1) * kSS - 1) {  // one column of a separator block per lane at a time: contiguous in the band
    const int i = t / kSS, b = t % kSS, ns = (i < N ? s.ps[i + 1] - kSI : nk) - s.ps[i];
    double *Di = s.Ds + (size_t)i * kSS * kSS;
    for (int a = 0; a < kSS; ++a) Di[a * kSS + b] = (a < ns && b < ns) ? band_at(B, nk, s.ps[i] + a, s.ps[i] + b) : (a == b ? 1.0 : 0.0);
  }
  CFZP_LANE_FOR(t, 0, (N + 1) * kSS * kSZ - 1) {
    const int i = t / (kSS * kSZ), a = (t / kSZ) % kSS, q = t % kSZ, ns = (i < N ? s.ps[i + 1] - kSI : nk) - s.ps[i];
    s.Us[t] = (q >= 14 && a < ns) ? (q == 14 ? b1[s.ps[i] + a] : b2[s.ps[i] + a]) : 0.0;
  }
  CFZP_SYNC();
  // M = C' W of interval i: rows = coupling columns (7 left, 14 right), columns = W's first 23; scattered into D_i, U_i, D_{i+1} and the
  // right-hand sides.  One (interval, row, column) triple per lane at a time, all intervals at once: what two neighbouring intervals
  // write into the same separator never coincides (interval i: the rows / columns of its right coupling list, interval i + 1: pt0).
  CFZP_LANE_FOR(tt, 0, N * 21 * 23 - 1) {
    const int i = tt / (21 * 23), t = tt - i * (21 * 23);
    const int *cl = s.cl + 24 * i;
    const double *C = s.Ci + (size_t)i * kSI * (kSL + kSRt), *W = s.Wi + (size_t)i * kSI * kSR;
    double *Di = s.Ds + (size_t)i * kSS * kSS, *Dn = s.Ds + (size_t)(i + 1) * kSS * kSS, *Ui = s.Us + (size_t)i * kSS * kSZ, *Un = s.Us + (size_t)(i + 1) * kSS * kSZ;
    const int a = t / 23, q = t % 23;  // coupling column a against W's column q (q < 21: coupling column q, 21 / 22: the right-hand sides)
    if (cl[a] < 0 || (q < 21 && cl[q] < 0)) continue;
    if (a >= 7 && q < 7) continue;  // (the transpose of a block that is kept)
    double m_ = 0.0;
#pragma unroll 16
    for (int r = 0; r < kSI; ++r) m_ += C[a * kSI + r] * W[q * kSI + r];  // (both stored column by column: contiguous in r)
    const int la = a < 7 ? cl[a] - s.ps[i] : cl[a] - s.ps[i + 1];  // local index in its separator
    if (q >= 21) { (a < 7 ? Ui : Un)[la * kSZ + 14 + (q - 21)] -= m_; continue; }
    const int lq = q < 7 ? cl[q] - s.ps[i] : cl[q] - s.ps[i + 1];
    if (a < 7 && q < 7) Di[la * kSS + lq] -= m_;
    else if (a >= 7 && q >= 7) Dn[la * kSS + lq] -= m_;
    else Ui[la * kSZ + (q - 7)] = -m_;  // coupling of separator i (row la) with separator i + 1 (its q-th coupled unknown)
  }
  CFZP_SYNC();
  { const long long t1 = tick(); ptk[1] += t1 - tp; tp = t1; }
  // ---- phase 3: the recursion over the separators (the first wavefront; everybody waits) -------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  {
    // From both ends: wavefront 0 eliminates separators 0 .. mid - 1 downwards (separator i into i + 1), wavefront 1 separators N .. mid + 1
    // upwards (j into j - 1); then wavefront 0 solves separator mid, which has received both, and the two back-substitute outwards.
    const int lane = CFZS_LANE, wv = CFZS_WAVE, mid = (4 * (N + 1)) / 7;  // (an upward step costs more than a downward one)
#define CFZS_WFENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
    // the upward steps' right-hand sides: columns 0..6 of Ub_j = U_{j-1}' (row = unknown of separator j, column = pt0 unknown of j - 1)
    for (int t = (int)threadIdx.x; t < (N - mid) * kSS * 14; t += (int)blockDim.x) { const int j = mid + 1 + t / (kSS * 14), e = t % (kSS * 14); s.Ub[(size_t)j * kSS * kSZ + (e / 14) * kSZ + (e % 14)] = 0.0; }
    __syncthreads();
    for (int t = (int)threadIdx.x; t < (N - mid) * 14 * 7; t += (int)blockDim.x) {
      const int j = mid + 1 + t / 98, e = t % 98, bq = e / 7, a = e % 7;
      const int *cl = s.cl + 24 * (j - 1);
      if (cl[7 + bq] >= 0) s.Ub[(size_t)j * kSS * kSZ + (cl[7 + bq] - s.ps[j]) * kSZ + a] = s.Us[(size_t)(j - 1) * kSS * kSZ + (cl[a] - s.ps[j - 1]) * kSZ + bq];
    }
    __syncthreads();
    if (wv == 0) {
      bool bad = false;
      for (int i = 0; i < mid && !bad; ++i) {
        const double *Ui = s.Us + (size_t)i * kSS * kSZ;
        double *Zi = s.Zs + (size_t)i * kSS * kSZ;
        CFZS_WFENCE();
        if (struct_separator((const cfzb::glb_f64 *)(s.Ds + (size_t)i * kSS * kSS), (const cfzb::glb_f64 *)Ui, (cfzb::glb_f64 *)Zi, s.ps[i + 1] - kSI - s.ps[i])) { bad = true; break; }
        CFZS_WFENCE();
        const int *cl = s.cl + 24 * i;
        double *Dn = s.Ds + (size_t)(i + 1) * kSS * kSS, *Un = s.Us + (size_t)(i + 1) * kSS * kSZ;
        for (int t = lane; t < 14 * kSZ; t += 64) {
          const int bq = t / kSZ, q = t % kSZ;
          if (cl[7 + bq] < 0 || (q < 14 && cl[7 + q] < 0)) continue;
          const int lb = cl[7 + bq] - s.ps[i + 1];
          double m_ = 0.0;
          for (int a = 0; a < 7; ++a) { const int la = cl[a] - s.ps[i]; m_ += Ui[la * kSZ + bq] * Zi[la * kSZ + q]; }
          if (q < 14) Dn[lb * kSS + (cl[7 + q] - s.ps[i + 1])] -= m_; else Un[lb * kSZ + q] -= m_;
        }
      }
      if (bad && lane == 0) flag[0] = 1.0;
    } else if (wv == 1) {
      bool bad = false;
      for (int j = N; j > mid && !bad; --j) {
        // right-hand sides of block j for the upward step: columns 0..6 = U_{j-1}' (row R_b of separator j, column a of pt0_{j-1}), 14 / 15 = r_j
        const int *cl = s.cl + 24 * (j - 1);
        const double *Up = s.Us + (size_t)(j - 1) * kSS * kSZ;  // coupling of separator j - 1 (rows) with separator j (its coupled unknowns, columns)
        double *Ubj = s.Ub + (size_t)j * kSS * kSZ, *Zbj = s.Zb + (size_t)j * kSS * kSZ;
        const double *Uj = s.Us + (size_t)j * kSS * kSZ;
        CFZS_WFENCE();
        for (int t = lane; t < kSS * 2; t += 64) Ubj[(t >> 1) * kSZ + 14 + (t & 1)] = Uj[(t >> 1) * kSZ + 14 + (t & 1)];  // r_j as it stands now (U' was laid out before the recursion)
        CFZS_WFENCE();
        if (struct_separator((const cfzb::glb_f64 *)(s.Ds + (size_t)j * kSS * kSS), (const cfzb::glb_f64 *)Ubj, (cfzb::glb_f64 *)Zbj, (j < N ? s.ps[j + 1] - kSI : nk) - s.ps[j])) { bad = true; break; }
        CFZS_WFENCE();
        // D_{j-1}[L, L] -= U_{j-1} Zb_j[R, 0:7];  r_{j-1}[L] -= U_{j-1} Zb_j[R, 14:16]
        double *Dp = s.Ds + (size_t)(j - 1) * kSS * kSS; double *Upw = s.Us + (size_t)(j - 1) * kSS * kSZ;
        for (int t = lane; t < 7 * 9; t += 64) {
          const int a = t / 9, q = t % 9, la = cl[a] - s.ps[j - 1];
          double m_ = 0.0;
          for (int bq = 0; bq < 14; ++bq) { if (cl[7 + bq] < 0) continue; m_ += Up[la * kSZ + bq] * Zbj[(cl[7 + bq] - s.ps[j]) * kSZ + (q < 7 ? q : 14 + (q - 7))]; }
          if (q < 7) Dp[la * kSS + (cl[q] - s.ps[j - 1])] -= m_; else Upw[la * kSZ + 14 + (q - 7)] -= m_;
        }
      }
      if (bad && lane == 0) flag[0] = 1.0;
    }
    __syncthreads();
    if (flag[0] == 0.0 && wv == 0) {  // the middle block: everything above and below has been folded into it
      double *Zm = s.Zs + (size_t)mid * kSS * kSZ;
      if (struct_separator((const cfzb::glb_f64 *)(s.Ds + (size_t)mid * kSS * kSS), (const cfzb::glb_f64 *)(s.Us + (size_t)mid * kSS * kSZ), (cfzb::glb_f64 *)Zm, (mid < N ? s.ps[mid + 1] - kSI : nk) - s.ps[mid])) { if (lane == 0) flag[0] = 1.0; }
      CFZS_WFENCE();
      if (mid < N) for (int t = lane; t < kSS * 2; t += 64) s.Zb[(size_t)mid * kSS * kSZ + (t / 2) * kSZ + 14 + (t & 1)] = Zm[(t / 2) * kSZ + 14 + (t & 1)];  // x_mid, for the downward pass
    }
    __syncthreads();
    if (flag[0] == 0.0) {
      if (wv == 0) {  // upwards: x_i = Z_i[:, rhs] - Z_i[:, U columns] x_{i+1}[R]
        for (int i = mid - 1; i >= 0; --i) {
          const int *cl = s.cl + 24 * i;
          double *Zi = s.Zs + (size_t)i * kSS * kSZ; const double *Zn = s.Zs + (size_t)(i + 1) * kSS * kSZ;
          CFZS_WFENCE();
          if (lane < kSS) {
            double x1 = Zi[lane * kSZ + 14], x2 = Zi[lane * kSZ + 15];
            for (int bq = 0; bq < 14; ++bq) {
              if (cl[7 + bq] < 0) continue;
              const int lb = cl[7 + bq] - s.ps[i + 1];
              x1 -= Zi[lane * kSZ + bq] * Zn[lb * kSZ + 14]; x2 -= Zi[lane * kSZ + bq] * Zn[lb * kSZ + 15];
            }
            Zi[lane * kSZ + 14] = x1; Zi[lane * kSZ + 15] = x2;
          }
        }
      } else if (wv == 1) {  // downwards: x_j = Zb_j[:, rhs] - Zb_j[:, 0:7] x_{j-1}[pt0]; results also into Z (phase 4 reads Z)
        for (int j = mid + 1; j <= N; ++j) {
          const int *cl = s.cl + 24 * (j - 1);
          double *Zbj = s.Zb + (size_t)j * kSS * kSZ; const double *Zbp = s.Zb + (size_t)(j - 1) * kSS * kSZ;
          CFZS_WFENCE();
          if (lane < kSS) {
            double x1 = Zbj[lane * kSZ + 14], x2 = Zbj[lane * kSZ + 15];
            for (int a = 0; a < 7; ++a) { const int la = cl[a] - s.ps[j - 1]; x1 -= Zbj[lane * kSZ + a] * Zbp[la * kSZ + 14]; x2 -= Zbj[lane * kSZ + a] * Zbp[la * kSZ + 15]; }
            Zbj[lane * kSZ + 14] = x1; Zbj[lane * kSZ + 15] = x2;
            s.Zs[(size_t)j * kSS * kSZ + lane * kSZ + 14] = x1; s.Zs[(size_t)j * kSS * kSZ + lane * kSZ + 15] = x2;
          }
        }
      }
    }
#undef CFZS_WFENCE
  }
#else
  CFZP_LANE_FOR(one, 0, 0) {
    double *aug = s.aug;
    const int lds = kSS + kSZ;
    for (int i = 0; i <= N; ++i) {
      const double *Di = s.Ds + (size_t)i * kSS * kSS, *Ui = s.Us + (size_t)i * kSS * kSZ;
      double *Zi = s.Zs + (size_t)i * kSS * kSZ;
      for (int a = 0; a < kSS; ++a) { for (int b = 0; b < kSS; ++b) aug[a * lds + b] = Di[a * kSS + b]; for (int q = 0; q < kSZ; ++q) aug[a * lds + kSS + q] = Ui[a * kSZ + q]; }
      if (block_solve_serial(aug, kSS, lds, kSZ)) { flag[0] = 1.0; break; }
      for (int a = 0; a < kSS; ++a) for (int q = 0; q < kSZ; ++q) Zi[a * kSZ + q] = aug[a * lds + kSS + q];
      if (i < N) {  // D_{i+1}[R, R] -= U_i' Z_i[:, U columns];  right-hand sides of separator i + 1 likewise
        const int *cl = s.cl + 24 * i;
        double *Dn = s.Ds + (size_t)(i + 1) * kSS * kSS, *Un = s.Us + (size_t)(i + 1) * kSS * kSZ;
        for (int bq = 0; bq < 14; ++bq) {
          if (cl[7 + bq] < 0) continue;
          const int lb = cl[7 + bq] - s.ps[i + 1];
          for (int q = 0; q < kSZ; ++q) {
            if (q < 14 && cl[7 + q] < 0) continue;
            double m_ = 0.0;
            for (int a = 0; a < 7; ++a) { const int la = cl[a] - s.ps[i]; m_ += Ui[la * kSZ + bq] * Zi[la * kSZ + q]; }
            if (q < 14) Dn[lb * kSS + (cl[7 + q] - s.ps[i + 1])] -= m_; else Un[lb * kSZ + q] -= m_;
          }
        }
      }
    }
    // backward: x_i = Z_i[:, rhs] - Z_i[:, U columns] x_{i+1}[R]   (x kept in Z's right-hand-side columns)
    if (flag[0] == 0.0)
      for (int i = N - 1; i >= 0; --i) {
        const int *cl = s.cl + 24 * i;
        double *Zi = s.Zs + (size_t)i * kSS * kSZ; const double *Zn = s.Zs + (size_t)(i + 1) * kSS * kSZ;
        for (int a = 0; a < kSS; ++a)
          for (int bq = 0; bq < 14; ++bq) {
            if (cl[7 + bq] < 0) continue;
            const int lb = cl[7 + bq] - s.ps[i + 1];
            Zi[a * kSZ + 14] -= Zi[a * kSZ + bq] * Zn[lb * kSZ + 14]; Zi[a * kSZ + 15] -= Zi[a * kSZ + bq] * Zn[lb * kSZ + 15];
          }
      }
  }
#endif
  CFZP_SYNC();
  if (flag[0] != 0.0) return 1;
  // ---- phase 4: the separators' and the interiors' unknowns back to band positions ----------------------------------------------
  CFZP_LANE_FOR(t, 0, (N + 1) * kSS - 1) {
    const int i = t / kSS, a = t % kSS, ns = (i < N ? s.ps[i + 1] - kSI : nk) - s.ps[i];
    if (a < ns) { b1[s.ps[i] + a] = s.Zs[(size_t)t * kSZ + 14]; b2[s.ps[i] + a] = s.Zs[(size_t)t * kSZ + 15]; }
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(t, 0, N * kSI - 1) {
    const int i = t / kSI, r = t % kSI, pi = s.ps[i + 1] - kSI;
    const int *cl = s.cl + 24 * i;
    const double *W = s.Wi + (size_t)i * kSI * kSR + r;  // column q of the interval's W at W[q * 64]
    double y1 = W[21 * kSI], y2 = W[22 * kSI];
    for (int q = 0; q < 21; ++q) if (cl[q] >= 0) { y1 -= W[q * kSI] * b1[cl[q]]; y2 -= W[q * kSI] * b2[cl[q]]; }
    // (the separators' values were written above: the barrier before this loop orders them)
    s.Ci[(size_t)i * kSI * (kSL + kSRt) + r] = y1; s.Ci[(size_t)i * kSI * (kSL + kSRt) + kSI + r] = y2;  // parked: b1 / b2 at interior positions are still inputs of nobody, but keep reads and writes apart
    (void)pi; (void)r;
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(t, 0, N * kSI - 1) {
    const int i = t / kSI, r = t % kSI, pi = s.ps[i + 1] - kSI;
    b1[pi + r] = s.Ci[(size_t)i * kSI * (kSL + kSRt) + r]; b2[pi + r] = s.Ci[(size_t)i * kSI * (kSL + kSRt) + kSI + r];
  }
  CFZP_SYNC();
  { const long long t1 = tick(); ptk[2] += t1 - tp; }
  return 0;
}

}  // namespace cfzc
