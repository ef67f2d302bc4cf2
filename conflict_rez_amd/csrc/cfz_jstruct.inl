// Structured elimination of the JOINT collocation plan's Newton system (included by cfz_colloc.inl; docs/notebook.md, round 5;
// reference confrez/control/multi_vehicle_planner.py:343-480 hands this system to MA97, :455-465).
//
// Per vehicle the matrix is the single-vehicle plan's (cfz_struct.inl): per Radau interval an INTERIOR of 64 unknowns [pt1 pt2 | 30 ODE
// rows | pt3 pt4 pt5 without its steering rate] between SEPARATORS -- and with the tube slacks and rows condensed into the pose they touch
// (assemble) a separator is [steering rate of the last point | 7 continuity rows | pt0] = 15 unknowns (14 at the start: initial rows and
// pt0; at most 6 at the end: steering rate and terminal rows), an interior couples to 7 unknowns on its left (pt0) and at most 7 on its
// right (the steering rate, six continuity rows; at the end the steering rate and up to four terminal rows).  The vehicles couple through
// the condensed pair blocks only: 6 x 6 blocks on the poses (x, y, psi) of two vehicles at the same (interval, point), kept beside the
// band (CWork::pm).  Poses of pt0 lie in separators; the poses of points 1..5 make the interiors of equal interval index t a coupled
// system  blockdiag(K_a) + E M E'  with E the 15 pose unknowns per vehicle and M the complete pair blocks (diagonal and off-diagonal parts:
// splitting them would cancel), solved by the capacitance system  (I + G M) y = E' K^-1 (...),  G = blockdiag(E' K_a^-1 E):
//   phase 1  every (vehicle, interval) interior by itself: K_a^-1 [C | b1 b2 | E]  (64 x 64, 32 right-hand sides, one wavefront, rows in
//            registers: wave_lu_regs of cfz_struct.inl);
//   phase 2  per interval index: M, the capacitance matrix (64 x 64: 16 rows per vehicle, identity-padded), Y = (I + G M)^-1 E'K^-1 [C | b],
//            Z = M Y; then the Schur complements onto the joint separators (64 x 64: 16 rows per vehicle);
//   phase 3  the joint separators are block tridiagonal: a recursion over the interval index;
//   phase 4  back-substitution of the interiors.
// Another elimination ORDER of the matrix the band path factors (tube rows condensed first): tools/joint_condense_study.py (numpy) and
// tests/test_colloc.py pin pattern and accuracy.  Every dense block is 64 x 64 with at most 32 right-hand sides: one register
// elimination serves all three phases.
#pragma once

namespace cfzc {

constexpr int kJR = 32;   // right-hand sides of an interior: 14 coupling columns, b1, b2, 15 unit vectors of its pair-coupled poses, 1 spare
constexpr int kJC = 14;   // coupling columns of an interior: 7 on the left (pt0), up to 7 on the right
constexpr int kJB = 64;   // rows of a capacitance / joint separator block: 16 per vehicle
constexpr int kJU = 32;   // right-hand sides of a joint separator: 7 right-coupled unknowns per vehicle, b1, b2, 2 spare
constexpr int kJMt = kMaxVeh * kMaxVeh * 5 * 9;  // pair blocks of one interval index: [a][b][point 1..5][3][3]
// local row of pose j (x, y, psi of points 1..5) inside an interior: pt1 0.., pt2 7.., 30 ODE rows, pt3 44.., pt4 51.., pt5 58..
CFZP_FN int jprow(int j) { const int k = j / 3, c = j - 3 * k; return (k == 0 ? 0 : k == 1 ? 7 : k == 2 ? 44 : k == 3 ? 51 : 58) + c; }
// column of Y / Z: coupling column q of vehicle b, or right-hand side s (two halves of 32: vehicles 0, 1 | vehicles 2, 3, b1, b2)
CFZP_FN int jycol(int b, int q) { return 32 * (b >> 1) + kJC * (b & 1) + q; }
constexpr int kJYrhs = 60;

struct JWork {
  double *W, *Cc, *CW, *Mt, *Cap, *Yh, *Y, *Z, *Ds, *Us, *Zs, *xs, *aug, *flag;
  int *cl, *bs;
  int Nmax, NI;
};

CFZP_FN int jstruct_nmax(const CSpec &sp) { int m = 0; for (int a = 0; a < sp.V; ++a) m = sp.N[a] > m ? sp.N[a] : m; return m; }
CFZP_FN size_t jstruct_doubles(const CSpec &sp) {
  if (!jstruct_mode(sp)) return 0;
  size_t NI = 0;
  for (int a = 0; a < sp.V; ++a) NI += sp.N[a];
  const size_t Nm = jstruct_nmax(sp);
  return NI * (kSI * kJR + kSI * kJC + kJC * kJR) + Nm * (kJMt + 4 * kJB * kJB) + (Nm + 1) * (kJB * kJB + 2 * kJB * kJU + 2 * kJB) +
         (size_t)kJB * (kJB + 64) + 8 + (NI * 16 + kMaxVeh + 3) / 2 + 16;
}
CFZP_FN JWork jstruct_carve(const CSpec &sp, double *p) {
  JWork s;
  size_t NI = 0;
  for (int a = 0; a < sp.V; ++a) NI += sp.N[a];
  const size_t Nm = jstruct_nmax(sp);
  s.Nmax = (int)Nm; s.NI = (int)NI;
  s.W = p; p += NI * kSI * kJR; s.Cc = p; p += NI * kSI * kJC; s.CW = p; p += NI * kJC * kJR;
  s.Mt = p; p += Nm * kJMt; s.Cap = p; p += Nm * kJB * kJB; s.Yh = p; p += Nm * kJB * kJB; s.Y = p; p += Nm * kJB * kJB; s.Z = p; p += Nm * kJB * kJB;
  s.Ds = p; p += (Nm + 1) * kJB * kJB; s.Us = p; p += (Nm + 1) * kJB * kJU; s.Zs = p; p += (Nm + 1) * kJB * kJU; s.xs = p; p += (Nm + 1) * 2 * kJB;
  s.aug = p; p += (size_t)kJB * (kJB + 64);
  s.flag = p; p += 8;
  s.cl = reinterpret_cast<int *>(p); s.bs = s.cl + NI * 16;
  return s;
}

// first position and size of separator i of vehicle a (i = 0 .. N_a)
CFZP_FN int jsep_start(const JWork &s, int a, int i) { return i == 0 ? s.bs[a] : s.bs[a] + 79 * i - 1; }
CFZP_FN int jsep_size(const CSpec &sp, int a, int i) { return i == 0 ? 14 : (i < sp.N[a] ? 15 : 5 + (sp.has_final[a] ? 1 : 0)); }

// cl[16 it + q], it = off[a] + t: q < 7 the positions of pt0 of the interval, 7 <= q < 14 the coupled positions of separator t + 1
// (-1 = none); flag[1] != 0: the ordering is not the one this file assumes (the caller falls back on nothing: status 3)
CFZP_FN void jstruct_setup(const CSpec &sp, const CDims &d, const CWork &w, const JWork &s) {
  CFZP_LANE_FOR(one, 0, 0) { s.flag[1] = 0.0; for (int a = 0; a < sp.V; ++a) s.bs[a] = w.posc[7 * a]; }
  CFZP_SYNC();
  CFZP_LANE_FOR(it, 0, d.NI - 1) {
    const int a = veh_of_interval(d, it), t = it - d.off[a];
    int *cl = s.cl + 16 * it;
    for (int c = 0; c < 7; ++c) cl[c] = w.posx[7 * (kPts * it) + c];
    int q = 7;
    cl[q++] = w.posx[7 * (kPts * it + 5) + 6];
    if (t + 1 < sp.N[a]) for (int c = 0; c < 6; ++c) cl[q++] = w.posc[d.rC + 7 * (it - a) + c];
    else {
      for (int c = 0; c < 3; ++c) cl[q++] = w.posc[d.rF + 5 * a + c];
      if (sp.has_final[a]) cl[q++] = w.posc[d.rF + 5 * a + 4];
    }
    while (q < 16) cl[q++] = -1;
    // the layout this file computes positions from
    const int pi = w.posc[7 * a] + 79 * t + 14;
    bool ok = w.posx[7 * (kPts * it + 1)] == pi && w.posx[7 * (kPts * it + 5) + 5] == pi + 63 && cl[7] == pi + 64 && cl[0] == pi - 7;
    if (t + 1 < sp.N[a]) ok = ok && cl[8] == pi + 65;
    if (!ok) s.flag[1] = 1.0;
  }
  CFZP_SYNC();
}

#if defined(__HIP_DEVICE_COMPILE__)
// interior of (vehicle, interval): rows gathered from the band, coupling columns kept in C, K^-1 [C | b1 b2 | E] to W (row = unknown)
__device__ __attribute__((noinline)) int jstruct_interior(const cfzb::glb_f64 *ab, int kb, int ld, int pi, const cfzb::glb_i32 *cl,
                                                          const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *C, cfzb::glb_f64 *W) {
  const int lane = threadIdx.x & 63, r = pi + lane;
  double a[kSI + kJR];
#pragma unroll
  for (int j = 0; j < kSI; ++j) { const int c = pi + j, dd = r - c; a[j] = (dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (2 * kb + dd)] : 0.0; }
#pragma unroll
  for (int q = 0; q < kJC; ++q) {
    const int c = cl[q], dd = r - c;
    const double v = (c >= 0 && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (2 * kb + dd)] : 0.0;
    a[kSI + q] = v;
    C[q * kSI + lane] = v;
  }
  a[kSI + 14] = b1[r]; a[kSI + 15] = b2[r];
#pragma unroll
  for (int j = 0; j < 15; ++j) a[kSI + 16 + j] = lane == jprow(j) ? 1.0 : 0.0;
  a[kSI + 31] = 0.0;
  int ord;
  if (wave_lu_regs<kSI, kJR>(a, lane, ord)) return 1;
#pragma unroll
  for (int q = 0; q < kJR; ++q) W[q * kSI + ord] = a[kSI + q];
  return 0;
}
// a 64 x 64 block (column-major in memory) with 32 right-hand sides (column-major, 64 rows): Z = A^-1 R
__device__ __attribute__((noinline)) int jstruct_block(const cfzb::glb_f64 *A, const cfzb::glb_f64 *R, cfzb::glb_f64 *Z) {
  const int lane = threadIdx.x & 63;
  double a[kJB + 32];
#pragma unroll
  for (int j = 0; j < kJB; ++j) a[j] = A[j * kJB + lane];
#pragma unroll
  for (int q = 0; q < 32; ++q) a[kJB + q] = R[q * kJB + lane];
  int ord;
  if (wave_lu_regs<kJB, 32>(a, lane, ord)) return 1;
#pragma unroll
  for (int q = 0; q < 32; ++q) Z[q * kJB + ord] = a[kJB + q];
  return 0;
}
#endif

// CPU build (and the definition of what the register eliminations compute): A (n x n, column-major, ld 64), nrhs columns R -> Z
CFZP_FN int jstruct_block_serial(double *aug, const double *A, const double *R, double *Z, int nrhs) {
  const int ld = kJB + 64;
  for (int r = 0; r < kJB; ++r) { for (int j = 0; j < kJB; ++j) aug[r * ld + j] = A[j * kJB + r]; for (int q = 0; q < nrhs; ++q) aug[r * ld + kJB + q] = R[q * kJB + r]; }
  if (block_solve_serial(aug, kJB, ld, nrhs)) return 1;
  for (int r = 0; r < kJB; ++r) for (int q = 0; q < nrhs; ++q) Z[q * kJB + r] = aug[r * ld + kJB + q];
  return 0;
}

// The whole solve: on return b1, b2 (positions of build_order_vm) hold the two solutions.  0 = ok, 1 = a block was singular.
// ptk[0..2]: interiors; capacitance systems and Schur complements; separator recursion and back-substitution (device clock)
CFZP_FN int jstruct_solve(const CSpec &sp, const CDims &d, const CWork &w, const JWork &s, const Band &B, double *b1, double *b2, long long *ptk) {
  double *flag = s.flag;
  long long tp = tick(), ts;
#define CFZJ_TICK(k) do { const long long t1_ = tick(); ptk[k] += t1_ - ts; ts = t1_; } while (0)  // ptk[3..10]: sub-phases
  const int Nm = s.Nmax, V = sp.V;
  if (flag[1] != 0.0) return 1;
  CFZP_LANE_FOR(one, 0, 0) flag[0] = 0.0;
  CFZP_SYNC();
  // ---- phase 1: interiors ---------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  for (int it = CFZS_WAVE; it < d.NI; it += CFZS_NW) {
    const int a = veh_of_interval(d, it), t = it - d.off[a];
    const int f = jstruct_interior((const cfzb::glb_f64 *)B.ab, B.kb, B.ld, s.bs[a] + 79 * t + 14, (const cfzb::glb_i32 *)(s.cl + 16 * it), (const cfzb::glb_f64 *)b1,
                                   (const cfzb::glb_f64 *)b2, (cfzb::glb_f64 *)(s.Cc + (size_t)it * kSI * kJC), (cfzb::glb_f64 *)(s.W + (size_t)it * kSI * kJR));
    if (f && CFZS_LANE == 0) flag[0] = 1.0;
  }
#else
  for (int it = 0; it < d.NI; ++it) {
    const int a = veh_of_interval(d, it), t = it - d.off[a], pi = s.bs[a] + 79 * t + 14, ld = kSI + kJR;
    const int *cl = s.cl + 16 * it;
    double *aug = s.aug, *C = s.Cc + (size_t)it * kSI * kJC, *W = s.W + (size_t)it * kSI * kJR;
    for (int r = 0; r < kSI; ++r) {
      for (int j = 0; j < kSI; ++j) aug[r * ld + j] = band_at(B, d.nk, pi + r, pi + j);
      for (int q = 0; q < kJC; ++q) { const double v = cl[q] >= 0 ? band_at(B, d.nk, pi + r, cl[q]) : 0.0; aug[r * ld + kSI + q] = v; C[q * kSI + r] = v; }
      aug[r * ld + kSI + 14] = b1[pi + r]; aug[r * ld + kSI + 15] = b2[pi + r];
      for (int j = 0; j < 15; ++j) aug[r * ld + kSI + 16 + j] = r == jprow(j) ? 1.0 : 0.0;
      aug[r * ld + kSI + 31] = 0.0;
    }
    if (block_solve_serial(aug, kSI, ld, kJR)) flag[0] = 1.0;
    for (int r = 0; r < kSI; ++r) for (int q = 0; q < kJR; ++q) W[q * kSI + r] = aug[r * ld + kSI + q];
  }
#endif
  CFZP_SYNC();
  { const long long t1 = tick(); ptk[0] += t1 - tp; tp = t1; ts = t1; }
  if (flag[0] != 0.0) return 1;
  // ---- phase 2a: C'W of every interior; the pair blocks and the capacitance matrix of every interval index --------------------------
  CFZP_LANE_FOR(tt, 0, d.NI * kJC * 31 - 1) {
    const int it = tt / (kJC * 31), e = tt - it * (kJC * 31), al = e / 31, q = e - al * 31;
    const double *C = s.Cc + (size_t)it * kSI * kJC + al * kSI, *W = s.W + (size_t)it * kSI * kJR + q * kSI;
    double m_ = 0.0;
    if (s.cl[16 * it + al] >= 0) {
#pragma unroll 16
      for (int r = 0; r < kSI; ++r) m_ += C[r] * W[r];
    }
    s.CW[(size_t)it * kJC * kJR + al * kJR + q] = m_;
  }
  CFZP_LANE_FOR(tt, 0, Nm * kJMt - 1) {  // Mt[t][a][b][point][i][j]: owner computes (a diagonal block sums over the pairs of its vehicle)
    const int t = tt / kJMt, e = tt - t * kJMt, a = e / (kMaxVeh * 45), b = (e / 45) % kMaxVeh, kk = (e / 9) % 5, i = (e / 3) % 3, j = e % 3;
    double v = 0.0;
    if (a < V && b < V && t < sp.N[a] && t < sp.N[b])
      for (int pe = 0; pe < sp.n_pairs; ++pe) {
        const int pa = sp.pair_a[pe], pb = sp.pair_b[pe];
        if (t >= sp.N[pa] || t >= sp.N[pb]) continue;
        const double *pm = w.pm + (size_t)(d.poff[pe] + kPts * t + kk + 1) * 36;
        if (a == b) { if (pa == a) v += pm[6 * i + j]; else if (pb == a) v += pm[6 * (3 + i) + 3 + j]; }
        else if (pa == a && pb == b) v += pm[6 * i + 3 + j];
        else if (pa == b && pb == a) v += pm[6 * (3 + i) + j];
      }
    s.Mt[tt] = v;
  }
  CFZP_SYNC();
  CFZJ_TICK(3);
  CFZP_LANE_FOR(tt, 0, Nm * kJB * kJB - 1) {  // Cap = I + G M (column-major), Yh = E' K^-1 [C | b] (column-major)
    const int t = tt / (kJB * kJB), e = tt - t * (kJB * kJB), col = e / kJB, row = e - col * kJB;
    const int a = row >> 4, i = row & 15, b = col >> 4, j = col & 15;
    double cap = row == col ? 1.0 : 0.0, yh = 0.0;
    if (a < V && t < sp.N[a] && i < 15) {
      const double *W = s.W + (size_t)(d.off[a] + t) * kSI * kJR;
      const int pr = jprow(i);
      if (b < V && t < sp.N[b] && j < 15) {
        const int kk = j / 3, jj = j - 3 * kk;
        const double *M = s.Mt + (size_t)t * kJMt + ((a * kMaxVeh + b) * 5 + kk) * 9;
        for (int c = 0; c < 3; ++c) cap += W[(16 + 3 * kk + c) * kSI + pr] * M[3 * c + jj];
      }
      // column `col` of Yh: coupling column q of vehicle vb, or a right-hand side
      const int h = col >> 5, lc = col & 31;
      if (col >= kJYrhs) { if (col < kJYrhs + 2) yh = W[(14 + col - kJYrhs) * kSI + pr]; }
      else if (lc < 2 * kJC) { const int vb = 2 * h + lc / kJC, q = lc % kJC; if (vb == a) yh = W[q * kSI + pr]; }
    }
    s.Cap[tt] = cap; s.Yh[tt] = yh;
  }
  CFZP_SYNC();
  CFZJ_TICK(4);
  // ---- phase 2b: Y = Cap^-1 Yh ------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  for (int k = CFZS_WAVE; k < 2 * Nm; k += CFZS_NW) {
    const int t = k >> 1, h = k & 1;
    const int f = jstruct_block((const cfzb::glb_f64 *)(s.Cap + (size_t)t * kJB * kJB), (const cfzb::glb_f64 *)(s.Yh + (size_t)t * kJB * kJB + h * 32 * kJB),
                                (cfzb::glb_f64 *)(s.Y + (size_t)t * kJB * kJB + h * 32 * kJB));
    if (f && CFZS_LANE == 0) flag[0] = 1.0;
  }
#else
  for (int t = 0; t < Nm; ++t)
    if (jstruct_block_serial(s.aug, s.Cap + (size_t)t * kJB * kJB, s.Yh + (size_t)t * kJB * kJB, s.Y + (size_t)t * kJB * kJB, 64)) flag[0] = 1.0;
#endif
  CFZP_SYNC();
  CFZJ_TICK(5);
  if (flag[0] != 0.0) return 1;
  // ---- phase 2c: Z = M Y ------------------------------------------------------------------------------------------------------------
  CFZP_LANE_FOR(tt, 0, Nm * kJB * kJB - 1) {
    const int t = tt / (kJB * kJB), e = tt - t * (kJB * kJB), col = e / kJB, row = e - col * kJB, a = row >> 4, i = row & 15;
    double z = 0.0;
    if (a < V && t < sp.N[a] && i < 15) {
      const int kk = i / 3, ii = i - 3 * kk;
      const double *Yc = s.Y + (size_t)t * kJB * kJB + col * kJB;
      for (int b = 0; b < V; ++b) {
        if (t >= sp.N[b]) continue;
        const double *M = s.Mt + (size_t)t * kJMt + ((a * kMaxVeh + b) * 5 + kk) * 9 + 3 * ii;
        for (int c = 0; c < 3; ++c) z += M[c] * Yc[16 * b + 3 * kk + c];
      }
    }
    s.Z[tt] = z;
  }
  CFZP_SYNC();
  CFZJ_TICK(6);
  // ---- phase 2d: the joint separator blocks: band entries, pair blocks of pt0, identity padding; right-hand sides ------------------------
  CFZP_LANE_FOR(tt, 0, (Nm + 1) * kJB * kJB - 1) {
    const int i = tt / (kJB * kJB), e = tt - i * (kJB * kJB), col = e / kJB, row = e - col * kJB, a = row >> 4, la = row & 15, b = col >> 4, lb = col & 15;
    double v = row == col ? 1.0 : 0.0;
    if (a < V && b < V && i <= sp.N[a] && i <= sp.N[b] && la < jsep_size(sp, a, i) && lb < jsep_size(sp, b, i)) {
      if (a == b) v = band_at(B, d.nk, jsep_start(s, a, i) + la, jsep_start(s, a, i) + lb);
      else {
        v = 0.0;
        const int p0a = i == 0 ? 7 : 8, p0b = p0a;  // pt0's first local index
        if (i < sp.N[a] && i < sp.N[b] && la >= p0a && la < p0a + 3 && lb >= p0b && lb < p0b + 3)
          for (int pe = 0; pe < sp.n_pairs; ++pe) {
            const int pa = sp.pair_a[pe], pb = sp.pair_b[pe];
            const double *pm = w.pm + (size_t)(d.poff[pe] + kPts * i) * 36;
            if (pa == a && pb == b) v += pm[6 * (la - p0a) + 3 + (lb - p0b)];
            else if (pa == b && pb == a) v += pm[6 * (3 + la - p0a) + (lb - p0b)];
          }
      }
      if (a == b && i < sp.N[a]) {  // the diagonal parts of the pair blocks at pt0
        const int p0 = i == 0 ? 7 : 8;
        if (la >= p0 && la < p0 + 3 && lb >= p0 && lb < p0 + 3)
          for (int pe = 0; pe < sp.n_pairs; ++pe) {
            const int pa = sp.pair_a[pe], pb = sp.pair_b[pe];
            if (i >= sp.N[pa] || i >= sp.N[pb]) continue;
            const double *pm = w.pm + (size_t)(d.poff[pe] + kPts * i) * 36;
            if (pa == a) v += pm[6 * (la - p0) + (lb - p0)]; else if (pb == a) v += pm[6 * (3 + la - p0) + 3 + (lb - p0)];
          }
      }
    }
    s.Ds[tt] = v;
  }
  CFZP_LANE_FOR(tt, 0, (Nm + 1) * kJB * kJU - 1) {
    const int i = tt / (kJB * kJU), e = tt - i * (kJB * kJU), col = e / kJB, row = e - col * kJB, a = row >> 4, la = row & 15;
    double v = 0.0;
    if (col >= 28 && col < 30 && a < V && i <= sp.N[a] && la < jsep_size(sp, a, i)) v = (col == 28 ? b1 : b2)[jsep_start(s, a, i) + la];
    s.Us[tt] = v;
  }
  CFZP_SYNC();
  CFZJ_TICK(7);
  // Schur complements of the interiors of interval index t onto separators t (rows / columns of pt0) and t + 1 (the right-coupled
  // unknowns):  S[(a, al), (b, be)] -= [a == b] C_a'W_a[al, be] - sum_j C_a'K_a^-1 E[al, j] Z[(a, j), (b, be)];  owner computes
  CFZP_LANE_FOR(tt, 0, Nm * kMaxVeh * kJC * (kMaxVeh * kJC + 2) - 1) {
    const int per = kMaxVeh * kJC + 2;
    const int t = tt / (kMaxVeh * kJC * per), e = tt - t * (kMaxVeh * kJC * per), a = e / (kJC * per), al = (e / per) % kJC, tb = e % per;
    if (a >= V || t >= sp.N[a]) continue;
    const int it = d.off[a] + t;
    const int *cl = s.cl + 16 * it;
    if (cl[al] < 0) continue;
    const double *CW = s.CW + (size_t)it * kJC * kJR + al * kJR;
    const double *Zt = s.Z + (size_t)t * kJB * kJB;
    const bool aleft = al < 7;
    const int ra = 16 * a + (aleft ? cl[al] - jsep_start(s, a, t) : cl[al] - jsep_start(s, a, t + 1));  // row in separator t (left) or t + 1
    if (tb >= kMaxVeh * kJC) {  // the right-hand sides
      const int sr = tb - kMaxVeh * kJC;
      double m_ = CW[14 + sr];
      for (int j = 0; j < 15; ++j) m_ -= CW[16 + j] * Zt[(kJYrhs + sr) * kJB + 16 * a + j];
      s.Us[(size_t)(aleft ? t : t + 1) * kJB * kJU + (28 + sr) * kJB + ra] -= m_;
      continue;
    }
    const int b = tb / kJC, be = tb - b * kJC;
    if (b >= V || t >= sp.N[b]) continue;
    const int *clb = s.cl + 16 * (d.off[b] + t);
    if (clb[be] < 0) continue;
    const bool bleft = be < 7;
    if (!aleft && bleft) continue;  // (the transpose of a block that is kept)
    double m_ = a == b ? CW[be] : 0.0;
    for (int j = 0; j < 15; ++j) m_ -= CW[16 + j] * Zt[jycol(b, be) * kJB + 16 * a + j];
    const int cb = 16 * b + (bleft ? clb[be] - jsep_start(s, b, t) : clb[be] - jsep_start(s, b, t + 1));
    if (aleft && bleft) s.Ds[(size_t)t * kJB * kJB + cb * kJB + ra] -= m_;
    else if (!aleft && !bleft) s.Ds[(size_t)(t + 1) * kJB * kJB + cb * kJB + ra] -= m_;
    else s.Us[(size_t)t * kJB * kJU + (7 * b + be - 7) * kJB + ra] = -m_;  // coupling of separator t (row) with separator t + 1 (vehicle b's be-th coupled unknown)
  }
  CFZP_SYNC();
  CFZJ_TICK(8);
  { const long long t1 = tick(); ptk[1] += t1 - tp; tp = t1; }
  // ---- phase 3: recursion over the joint separators -------------------------------------------------------------------------------------
  for (int i = 0; i <= Nm; ++i) {
    double *Di = s.Ds + (size_t)i * kJB * kJB, *Ui = s.Us + (size_t)i * kJB * kJU, *Zi = s.Zs + (size_t)i * kJB * kJU;
#if defined(__HIP_DEVICE_COMPILE__)
    if (CFZS_WAVE == 0) { if (jstruct_block((const cfzb::glb_f64 *)Di, (const cfzb::glb_f64 *)Ui, (cfzb::glb_f64 *)Zi) && CFZS_LANE == 0) flag[0] = 1.0; }
#else
    if (jstruct_block_serial(s.aug, Di, Ui, Zi, kJU)) flag[0] = 1.0;
#endif
    CFZP_SYNC();
    if (flag[0] != 0.0) return 1;
    if (i == Nm) break;
    // D_{i+1}[R, R] -= U_i' Z_i[:, U columns],  right-hand sides of separator i + 1 likewise; U_i's rows are pt0 rows only
    double *Dn = s.Ds + (size_t)(i + 1) * kJB * kJB, *Un = s.Us + (size_t)(i + 1) * kJB * kJU;
    CFZP_LANE_FOR(tt, 0, 28 * 30 - 1) {
      const int cu = tt / 30, q = tt - cu * 30, b = cu / 7, be = cu - 7 * b;  // U column cu = (vehicle b, right-coupled unknown be) against Z column q
      if (b >= V || i >= sp.N[b]) continue;
      const int *clb = s.cl + 16 * (d.off[b] + i);
      if (clb[7 + be] < 0) continue;
      const int rb = 16 * b + clb[7 + be] - jsep_start(s, b, i + 1);
      int cq = -1;
      if (q < 28) {
        const int b2_ = q / 7, be2 = q - 7 * b2_;
        if (b2_ >= V || i >= sp.N[b2_]) continue;
        const int *cl2 = s.cl + 16 * (d.off[b2_] + i);
        if (cl2[7 + be2] < 0) continue;
        cq = 16 * b2_ + cl2[7 + be2] - jsep_start(s, b2_, i + 1);
      }
      double m_ = 0.0;
      for (int a = 0; a < V; ++a) {
        if (i >= sp.N[a]) continue;
        const int r0 = 16 * a + (i == 0 ? 7 : 8);
        for (int c = 0; c < 7; ++c) m_ += Ui[cu * kJB + r0 + c] * Zi[q * kJB + r0 + c];
      }
      if (q < 28) Dn[cq * kJB + rb] -= m_; else Un[q * kJB + rb] -= m_;
    }
    CFZP_SYNC();
  }
  // backward: x_i = Z_i[:, b] - Z_i[:, U columns] x_{i+1}[R]
  CFZP_LANE_FOR(tt, 0, 2 * kJB - 1) s.xs[(size_t)Nm * 2 * kJB + tt] = s.Zs[(size_t)Nm * kJB * kJU + (28 + tt / kJB) * kJB + (tt % kJB)];
  CFZP_SYNC();
  for (int i = Nm - 1; i >= 0; --i) {
    const double *Zi = s.Zs + (size_t)i * kJB * kJU, *xn = s.xs + (size_t)(i + 1) * 2 * kJB;
    CFZP_LANE_FOR(tt, 0, 2 * kJB - 1) {
      const int sr = tt / kJB, row = tt - sr * kJB;
      double x = Zi[(28 + sr) * kJB + row];
      for (int b = 0; b < V; ++b) {
        if (i >= sp.N[b]) continue;
        const int *clb = s.cl + 16 * (d.off[b] + i);
        for (int be = 0; be < 7; ++be) if (clb[7 + be] >= 0) x -= Zi[(7 * b + be) * kJB + row] * xn[sr * kJB + 16 * b + clb[7 + be] - jsep_start(s, b, i + 1)];
      }
      s.xs[(size_t)i * 2 * kJB + tt] = x;
    }
    CFZP_SYNC();
  }
  CFZJ_TICK(9);
  // ---- phase 4: the separators' and the interiors' unknowns back to their positions --------------------------------------------------------
  CFZP_LANE_FOR(tt, 0, (Nm + 1) * kJB - 1) {
    const int i = tt / kJB, row = tt - i * kJB, a = row >> 4, la = row & 15;
    if (a < V && i <= sp.N[a] && la < jsep_size(sp, a, i)) { const int p = jsep_start(s, a, i) + la; b1[p] = s.xs[(size_t)i * 2 * kJB + row]; b2[p] = s.xs[(size_t)i * 2 * kJB + kJB + row]; }
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(tt, 0, Nm * 2 * kJB - 1) {  // z = M y of every interval index: Z[:, b] - Z[:, coupling columns] s   (kept in Yh's first two columns)
    const int t = tt / (2 * kJB), e = tt - t * (2 * kJB), sr = e / kJB, row = e - sr * kJB;
    const double *Zt = s.Z + (size_t)t * kJB * kJB, *xb = sr ? b2 : b1;
    double z = Zt[(kJYrhs + sr) * kJB + row];
    for (int b = 0; b < V; ++b) {
      if (t >= sp.N[b]) continue;
      const int *clb = s.cl + 16 * (d.off[b] + t);
      for (int q = 0; q < kJC; ++q) if (clb[q] >= 0) z -= Zt[jycol(b, q) * kJB + row] * xb[clb[q]];
    }
    s.Yh[(size_t)t * kJB * kJB + e] = z;
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(tt, 0, d.NI * kSI - 1) {
    const int it = tt / kSI, r = tt - it * kSI, a = veh_of_interval(d, it), t = it - d.off[a];
    const int *cl = s.cl + 16 * it;
    const double *W = s.W + (size_t)it * kSI * kJR + r, *z = s.Yh + (size_t)t * kJB * kJB;
    double y1 = W[14 * kSI], y2 = W[15 * kSI];
    for (int q = 0; q < kJC; ++q) if (cl[q] >= 0) { y1 -= W[q * kSI] * b1[cl[q]]; y2 -= W[q * kSI] * b2[cl[q]]; }
    for (int j = 0; j < 15; ++j) { y1 -= W[(16 + j) * kSI] * z[16 * a + j]; y2 -= W[(16 + j) * kSI] * z[kJB + 16 * a + j]; }
    const int p = s.bs[a] + 79 * t + 14 + r;  // (interior positions are read by nobody in this phase)
    b1[p] = y1; b2[p] = y2;
  }
  CFZP_SYNC();
  CFZJ_TICK(10);
#undef CFZJ_TICK
  { const long long t1 = tick(); ptk[2] += t1 - tp; }
  return 0;
}

}  // namespace cfzc
