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
// JWork::aug: per wavefront the 64 x 96 rows of a capacitance system between their builder and the elimination (the device; column-major,
// every lane reads back what it wrote); the CPU build stages one block (64 x 128) and a capacitance system (2 x 64 x 64) there
constexpr int kJRowBuf = (kJB + 32) * kJB, kJRows = 8 * kJRowBuf;
static_assert(kJRows >= kJB * (kJB + 64) + 2 * kJB * kJB, "the CPU build's staging fits");

constexpr int kJL = 28;   // unknowns on either side of a link between two joint separators: 7 per vehicle
struct JWork {
  double *W, *Cc, *CW, *Mt, *Z, *zt, *Ds, *Us, *Zs, *Zb, *Lc, *xs, *aug, *flag;  // Zb, Lc: the cyclic reduction's second inverse columns and link copies
  int *cl, *bs, *ordl, *meta, *cmask;
  int Nmax, NI;
};

CFZP_FN int jstruct_nmax(const CSpec &sp) { int m = 0; for (int a = 0; a < sp.V; ++a) m = sp.N[a] > m ? sp.N[a] : m; return m; }
CFZP_FN size_t jstruct_doubles(const CSpec &sp) {
  if (!jstruct_mode(sp)) return 0;
  size_t NI = 0;
  for (int a = 0; a < sp.V; ++a) NI += sp.N[a];
  const size_t Nm = jstruct_nmax(sp);
  return NI * (kSI * kJR + kSI * kJC + kJC * kJR) + Nm * (kJMt + kJB * kJB + 2 * kJB) + (Nm + 1) * (kJB * kJB + 3 * kJB * kJU + kJL * kJL + 2 * kJB) +
         (size_t)kJRows + 8 + (NI * 16 + kMaxVeh + (Nm + 1) * kJB + 64 + NI + 3) / 2 + 16;
}
// doubles the structured elimination of one joint Newton system moves between its phases (bench.py's roofline of configs[3]): the band
// written where the assembly has entries and gathered once, the pair blocks written and read; W written, read for C'W and by the back-substitution; C and C'W written and
// read; Z written, read by the Schur complements and by the back-substitution; separator blocks, right-hand sides and solutions written and
// read once; the two right-hand sides in and out
CFZP_FN size_t jstruct_alg_doubles(const CSpec &sp, size_t nk, size_t ld, size_t npp) {
  size_t NI = 0;
  for (int a = 0; a < sp.V; ++a) NI += sp.N[a];
  const size_t Nm = jstruct_nmax(sp);
  // one vehicle: no capacitance systems, 16 right-hand sides per interior, 16-row separator blocks with 10 right-hand sides
  // (the band: gathered once, nk ld; written where the assembly has entries, about an eighth of it -- it is cleared once per solve, not per
  // Newton system: cfz_colloc.inl assemble)
  if (sp.V == 1) return nk * ld + nk * ld / 8 + 3 * NI * kSI * 16 + 2 * NI * kSI * kJC + 2 * NI * kJC * 16 + 2 * (Nm + 1) * 256 + 4 * (Nm + 1) * 160 + 4 * nk;
  // (+ the capacitance systems' rows on their way from the builder to the elimination: two halves per interval index, written and read)
  return nk * ld + nk * ld / 8 + 2 * npp * 36 + 3 * NI * kSI * kJR + 2 * NI * kSI * kJC + 2 * NI * kJC * kJR + 3 * Nm * kJB * kJB + 2 * (Nm + 1) * kJB * kJB +
         2 * (Nm + 1) * kJB * 30 + 2 * (Nm + 1) * kJB * 30 + 2 * (Nm + 1) * kJB * kJL + 2 * (Nm + 1) * kJL * kJL + 4 * nk + 2 * (2 * Nm) * (size_t)kJRowBuf;
}
CFZP_FN JWork jstruct_carve(const CSpec &sp, double *p) {
  JWork s;
  size_t NI = 0;
  for (int a = 0; a < sp.V; ++a) NI += sp.N[a];
  const size_t Nm = jstruct_nmax(sp);
  s.Nmax = (int)Nm; s.NI = (int)NI;
  s.W = p; p += NI * kSI * kJR; s.Cc = p; p += NI * kSI * kJC; s.CW = p; p += NI * kJC * kJR;
  s.Mt = p; p += Nm * kJMt; s.Z = p; p += Nm * kJB * kJB; s.zt = p; p += Nm * 2 * kJB;
  s.Ds = p; p += (Nm + 1) * kJB * kJB; s.Us = p; p += (Nm + 1) * kJB * kJU; s.Zs = p; p += (Nm + 1) * kJB * kJU; s.xs = p; p += (Nm + 1) * 2 * kJB;
  s.Zb = p; p += (Nm + 1) * kJB * kJU; s.Lc = p; p += (Nm + 1) * kJL * kJL;
  s.aug = p; p += (size_t)kJRows;  // the CPU build's staging (one block with its right-hand sides, a capacitance matrix and its right-hand sides); the device's row buffers
  s.flag = p; p += 8;
  s.cl = reinterpret_cast<int *>(p); s.bs = s.cl + NI * 16; s.ordl = s.bs + kMaxVeh; s.meta = s.ordl + (Nm + 1) * kJB; s.cmask = s.meta + 64;
  return s;
}

// first position and size of separator i of vehicle a (i = 0 .. N_a)
CFZP_FN int jsep_start(const JWork &s, int a, int i) { return i == 0 ? s.bs[a] : s.bs[a] + 79 * i - 1; }
CFZP_FN int jsep_size(const CSpec &sp, int a, int i) { return i == 0 ? 14 : (i < sp.N[a] ? 15 : 5 + (sp.has_final[a] ? 1 : 0)); }

// cl[16 it + q], it = off[a] + t: q < 7 the positions of pt0 of the interval, 7 <= q < 14 the coupled positions of separator t + 1
// (-1 = none); flag[1] != 0: the ordering is not the one this file assumes (the caller falls back on nothing: status 3)
CFZP_FN void jstruct_setup(const CSpec &sp, const CDims &d, const CWork &w, const JWork &s) {
  CFZP_LANE_FOR(one, 0, 0) {
    s.flag[1] = 0.0;
    for (int a = 0; a < sp.V; ++a) { s.bs[a] = w.posc[7 * a]; if (sp.N[a] > 255) s.flag[1] = 1.0; }
    // what the device functions read instead of the specification: N, first positions, terminal headings, the pair of two vehicles, ...
    int *m = s.meta;
    for (int a = 0; a < kMaxVeh; ++a) { m[a] = a < sp.V ? sp.N[a] : 0; m[4 + a] = a < sp.V ? w.posc[7 * a] : 0; m[8 + a] = a < sp.V ? (sp.has_final[a] ? 1 : 0) : 0; }
    for (int e = 0; e < 16; ++e) m[12 + e] = -1;
    for (int e = 0; e < sp.n_pairs; ++e) { m[12 + 4 * sp.pair_a[e] + sp.pair_b[e]] = e; m[12 + 4 * sp.pair_b[e] + sp.pair_a[e]] = e; }
    for (int e = 0; e <= kMaxPairs; ++e) m[28 + e] = d.poff[e];
    m[35] = sp.V;
    for (int a = 0; a <= kMaxVeh; ++a) m[36 + a] = d.off[a];
    m[41] = d.nk; m[42] = s.Nmax;
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(it, 0, d.NI - 1) {
    const int a = veh_of_interval(d, it), t = it - d.off[a];
    int *cl = s.cl + 16 * it;
    for (int c = 0; c < 7; ++c) cl[c] = w.posx[7 * (kPts * it) + c];
    int q = 7;
    cl[q++] = w.posx[7 * (kPts * it + 5) + 6];
    // cl[7 + be] is local unknown be of separator t + 1 (or -1): the steering rate, then six continuity rows -- at the end the terminal
    // rows of v, delta, a (the steering rate's own row does not touch the interior) and the heading row
    if (t + 1 < sp.N[a]) for (int c = 0; c < 6; ++c) cl[q++] = w.posc[d.rC + 7 * (it - a) + c];
    else {
      for (int c = 0; c < 3; ++c) cl[q++] = w.posc[d.rF + 5 * a + c];
      cl[q++] = -1;
      cl[q++] = sp.has_final[a] ? w.posc[d.rF + 5 * a + 4] : -1;
    }
    while (q < 16) cl[q++] = -1;
    for (int be = 0; be < 7; ++be) if (cl[7 + be] >= 0 && cl[7 + be] != w.posx[7 * (kPts * it + 5) + 6] + be) s.flag[1] = 1.0;
    int cm = 0;
    for (int c = 0; c < kJC; ++c) if (cl[c] >= 0) cm |= 1 << c;
    s.cmask[it] = cm;
    // the layout this file computes positions from
    const int pi = w.posc[7 * a] + 79 * t + 14;
    bool ok = w.posx[7 * (kPts * it + 1)] == pi && w.posx[7 * (kPts * it + 5) + 5] == pi + 63 && cl[7] == pi + 64 && cl[0] == pi - 7;
    if (t + 1 < sp.N[a]) ok = ok && cl[8] == pi + 65;
    if (!ok) s.flag[1] = 1.0;
  }
  CFZP_SYNC();
}

#if defined(__HIP_DEVICE_COMPILE__)
// One call per wavefront and phase, the loop over the wavefront's tasks inside: an out-of-line function that uses the whole register file
// saves and restores the caller's registers in scratch at entry and exit (60-230 dwords per lane), which per TASK was 20 MB of scratch
// traffic per Newton system -- 40 % of the elimination's own bytes (profiles/r5a_extras_*: 2.05x the algorithmic bytes moved).
// an argument that is the same in every lane, moved to scalar registers: a pointer kept in vector registers across a task that wants all
// 256 of them is spilled and reloaded per task
template <class T>
__device__ __forceinline__ T *juni(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int juni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ cfzb::lds_f64 *juni_lds(cfzb::lds_f64 *p) { return (cfzb::lds_f64 *)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p); }
// interior of (vehicle, interval): rows gathered from the band, coupling columns kept in C, K^-1 [C | b1 b2 | E] to W (row = unknown)
__device__ __forceinline__ int jstruct_interior(const cfzb::glb_f64 *ab, int kb, int ld, int off, int pi, const cfzb::glb_i32 *cl,
                                                          const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *C, cfzb::glb_f64 *W, cfzb::lds_f64 *lds) {
  const int lane = lu_opaque(threadIdx.x & 63), r = pi + lane;
  return lu64_build<kJR>(lds, W, [&](auto Jc, double (&v)[16]) {  // W[q * kSI + unknown]
    constexpr int J = decltype(Jc)::value;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (J < 4) { const int c = pi + 16 * J + k, dd = r - c; v[k] = (dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0; }
      else if constexpr (J == 4) {
        if (k < kJC) {
          const int c = cl[k], dd = r - c;
          v[k] = (c >= 0 && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0;
          C[k * kSI + lane] = v[k];
        } else v[k] = k == 14 ? b1[r] : b2[r];
      } else v[k] = (k < 15 && lane == jprow(k)) ? 1.0 : 0.0;
    }
  });
}
// C'W of one interior on the matrix cores: (14 x 64) (64 x 32) as two 16 x 16 tiles of v_mfma_f64_16x16x4_f64, sixteen k-steps
// (operands: lane l holds A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15]; result register r of lane l is row (l >> 4) + 4 r,
// column l & 15).  48 loads, 32 matrix instructions: the 434 dot products of length 64 took 1.3 ms per factorisation as a loop.
typedef double jstruct_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void jstruct_cw(const cfzb::glb_f64 *C, const cfzb::glb_f64 *W, cfzb::glb_f64 *CW) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  jstruct_v4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double av[16], b0[16], b1[16];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    av[kk] = lo < kJC ? C[lo * kSI + 4 * kk + hi] : 0.0;
    b0[kk] = W[lo * kSI + 4 * kk + hi]; b1[kk] = W[(16 + lo) * kSI + 4 * kk + hi];
  }
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], b0[kk], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], b1[kk], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int al = hi + 4 * r;
    if (al < kJC) { CW[al * kJR + lo] = acc0[r]; CW[al * kJR + 16 + lo] = acc1[r]; }
  }
}
// The capacitance system of interval index t, one half of its right-hand sides: (I + M G) Z = M E'K^-1 [C | b]  (Z = M Y without forming
// Y = (I + G M)^-1 E'K^-1 [C | b]: M (I + G M)^-1 = (I + M G)^-1 M).  Row (a, i) of matrix and right-hand sides is built in the registers of
// lane 16 a + i from M's three entries per vehicle for that row and the rows jprow(.) of the interiors' solutions W; nothing but Z is stored.
// h = 0: the coupling columns of vehicles 0 and 1; h = 1: of vehicles 2 and 3, then b1, b2.
__device__ __forceinline__ int jstruct_cap(int t, int h, const cfzb::glb_i32 *mt, const cfzb::glb_f64 *Wall, const cfzb::glb_f64 *pm, cfzb::glb_f64 *Zt, cfzb::lds_f64 *lds,
                                                     cfzb::glb_f64 *S) {
  const int lane = lu_opaque(threadIdx.x & 63), va = lane >> 4, i = lane & 15, V = mt[35];
  const bool real = va < V && t < mt[va] && i < 15;
  const int kk = real ? i / 3 : 0, ii = real ? i - 3 * kk : 0;
  const int pr0 = jprow(3 * kk), pr1 = pr0 + 1, pr2 = pr0 + 2;
  // M's row of this lane, read from the pair blocks themselves: against another vehicle the cross block of their pair; against its own
  // vehicle the sum of the diagonal blocks of all its pairs
  double m[4][3], dg[3] = {0.0, 0.0, 0.0};
  const cfzb::glb_f64 *Wb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const bool part = b < V && t < mt[b];
    Wb[b] = juni(Wall + (size_t)(part ? mt[36 + b] + t : 0) * kSI * kJR);  // (scalar base + this lane's 32-bit offset: no 64-bit address per load)
    const int e = (real && part && b != va) ? mt[12 + 4 * va + b] : -1;
    const cfzb::glb_f64 *pe = pm + (size_t)(e >= 0 ? mt[28 + e] + kPts * t + kk + 1 : 0) * 36 + (va < b ? 6 * ii : 6 * (3 + ii));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      m[b][c] = e >= 0 ? pe[va < b ? 3 + c : c] : 0.0;
      dg[c] += e >= 0 ? pe[va < b ? c : 3 + c] : 0.0;
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int c = 0; c < 3; ++c) if (b == va) m[b][c] = dg[c];
  // The rows are built column by column into this wavefront's row buffer (coalesced; eight columns' loads in flight at a time) and read
  // back by the same lane a column tile at a time: builder and elimination do not compete for registers (built in registers beside the
  // tiles, two of the six tiles lived in scratch memory).
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      double x = lane == 16 * b + j ? 1.0 : 0.0;
      if (j < 15) x += m[b][0] * Wb[b][(16 + j) * kSI + pr0] + m[b][1] * Wb[b][(16 + j) * kSI + pr1] + m[b][2] * Wb[b][(16 + j) * kSI + pr2];
      S[(16 * b + j) * kJB + lane] = x;
      if ((j & 7) == 7) asm volatile("" ::: "memory");
    }
  }
#pragma unroll
  for (int lc = 0; lc < 28; ++lc) {
    const int bl = lc / kJC, q = lc % kJC;
    const cfzb::glb_f64 *Wv = h ? Wb[2 + bl] : Wb[bl];
    const double m0 = h ? m[2 + bl][0] : m[bl][0], m1 = h ? m[2 + bl][1] : m[bl][1], m2 = h ? m[2 + bl][2] : m[bl][2];
    S[(kJB + lc) * kJB + lane] = m0 * Wv[q * kSI + pr0] + m1 * Wv[q * kSI + pr1] + m2 * Wv[q * kSI + pr2];
    if (lc % 7 == 6) asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int sr = 0; sr < 2; ++sr) {
    double x = 0.0;
    if (h) {
#pragma unroll
      for (int b = 0; b < 4; ++b) x += m[b][0] * Wb[b][(14 + sr) * kSI + pr0] + m[b][1] * Wb[b][(14 + sr) * kSI + pr1] + m[b][2] * Wb[b][(14 + sr) * kSI + pr2];
    }
    S[(kJB + 28 + sr) * kJB + lane] = x;
  }
  S[(kJB + 30) * kJB + lane] = 0.0; S[(kJB + 31) * kJB + lane] = 0.0;
  asm volatile("" ::: "memory");
  const cfzb::glb_f64 *Sr = juni((const cfzb::glb_f64 *)S);
  return lu64_build<32>(lds, Zt + 32 * h * kJB, [&](auto Jc, double (&v)[16]) {  // Zt[(32 h + q) * kJB + unknown]
    constexpr int J = decltype(Jc)::value;
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = Sr[(16 * J + k) * kJB + lane];
  });
}

// meta (JWork::meta, filled by jstruct_setup): [0..3] N, [4..7] first position, [8..11] terminal heading, [12..27] pair index of two vehicles
// (-1: none), [28..34] first pair point of a pair, [35] V, [36..40] first interval of a vehicle, [41] band unknowns, [42] the longest plan
__device__ __forceinline__ int jsep_size_m(const cfzb::glb_i32 *m, int a, int i) { return i == 0 ? 14 : (i < m[a] ? 15 : 5 + m[8 + a]); }
// Separator block i before the Schur complements, a wavefront per block, lane = row 16 a + la: band entries of the vehicle's own separator,
// the pair blocks of pt0 (diagonal and off-diagonal parts), identity padding; right-hand sides: zero coupling columns, b1, b2.
__device__ __forceinline__ void jstruct_sep_base(int i, const cfzb::glb_i32 *m, const cfzb::glb_f64 *ab, int kb, int ld, int off, const cfzb::glb_f64 *pm,
                                                           const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *Di, cfzb::glb_f64 *Ui) {
  const int lane = threadIdx.x & 63, a = lane >> 4, la = lane & 15, V = m[35];
  const bool rowok = a < V && i <= m[a] && la < jsep_size_m(m, a, i);
  const int p0 = i == 0 ? 7 : 8, ps = i == 0 ? m[4 + a] : m[4 + a] + 79 * i - 1, r = ps + la;
  const bool rpose = rowok && i < m[a] && la >= p0 && la < p0 + 3;
  double own[16];  // the vehicle's own columns
#pragma unroll
  for (int lb = 0; lb < 16; ++lb) {
    const int c = ps + lb, dd = r - c;
    own[lb] = (rowok && lb < jsep_size_m(m, a, i) && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0;
  }
  double x3[4][3];  // the pair blocks' row of this lane: [other vehicle][column], and the diagonal parts summed into x3[a]
#pragma unroll
  for (int b = 0; b < 4; ++b) { x3[b][0] = 0.0; x3[b][1] = 0.0; x3[b][2] = 0.0; }
  if (rpose) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int e = b < V && b != a ? m[12 + 4 * a + b] : -1;
      if (e >= 0 && i < m[b]) {
        const cfzb::glb_f64 *pe = pm + (size_t)(m[28 + e] + kPts * i) * 36;
        const int ro = a < b ? 6 * (la - p0) : 6 * (3 + la - p0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double cross = pe[ro + (a < b ? 3 + c : c)], diag = pe[ro + (a < b ? c : 3 + c)];
          x3[b][c] = cross;
#pragma unroll
          for (int b2_ = 0; b2_ < 4; ++b2_) if (b2_ == a) x3[b2_][c] += diag;  // (static register indices)
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const bool colveh = b < V && i <= m[b];
#pragma unroll
    for (int lb = 0; lb < 16; ++lb) {
      double v = lane == 16 * b + lb ? 1.0 : 0.0;
      if (rowok) {
        v = b == a ? own[lb] : 0.0;
        if (!(colveh && lb < jsep_size_m(m, b, i))) v = 0.0;
        else if (rpose && i < m[b] && lb >= p0 && lb < p0 + 3) v += lb == p0 ? x3[b][0] : (lb == p0 + 1 ? x3[b][1] : x3[b][2]);
      }
      Di[(16 * b + lb) * kJB + lane] = v;
    }
  }
#pragma unroll
  for (int q = 0; q < 28; ++q) Ui[q * kJB + lane] = 0.0;
  Ui[28 * kJB + lane] = rowok ? b1[r] : 0.0; Ui[29 * kJB + lane] = rowok ? b2[r] : 0.0;
}
// Schur complements of the interiors of interval index t onto separator blocks t and t + 1, a wavefront per t, on the matrix cores: per
// vehicle a  T = (C_a'K_a^-1 E) Z[(a, .), :]  is (14 x 15) (15 x 62): four k-steps of four 16 x 16 tiles of v_mfma_f64_16x16x4_f64 (the
// operands' 16th row / column are zero: the spare column of W, the padding row of Z); each lane then adds its results T - [own column] C'W
// to the entries they belong to (every entry has one owner; blocks t's pt0 entries and block t + 1's right-coupled entries are disjoint).
__device__ __forceinline__ void jstruct_schur(int t, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cmask, const cfzb::glb_f64 *CWall, const cfzb::glb_f64 *Zt,
                                                        cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4, V = m[35], p0 = t == 0 ? 7 : 8;
  cfzb::glb_f64 *D0 = Ds + (size_t)t * kJB * kJB, *D1 = D0 + kJB * kJB, *U0 = Us + (size_t)t * kJB * kJU, *U1 = U0 + kJB * kJU;
  int cm[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) cm[b] = (b < V && t < m[b]) ? cmask[m[36 + b] + t] : 0;
  for (int a = 0; a < V; ++a) {
    if (t >= m[a]) continue;
    const cfzb::glb_f64 *CW = CWall + (size_t)(m[36 + a] + t) * kJC * kJR;
    const int cma = cmask[m[36 + a] + t];
    jstruct_v4 acc[4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) acc[tl] = jstruct_v4{0.0, 0.0, 0.0, 0.0};
    double av[4], bv[4][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      av[ks] = lo < kJC ? CW[lo * kJR + 16 + 4 * ks + hi] : 0.0;
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) bv[ks][tl] = Zt[(16 * tl + lo) * kJB + 16 * a + 4 * ks + hi];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) acc[tl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks][tl], acc[tl], 0, 0, 0);
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
      const int col = 16 * tl + lo, hh = col >> 5, lc = col & 31;
      const bool isrhs = col >= kJYrhs && col < kJYrhs + 2, iscol = col < kJYrhs && lc < 2 * kJC;
      const int b = iscol ? 2 * hh + lc / kJC : a, be = iscol ? lc % kJC : 0;
      const int cmb = b == 0 ? cm[0] : b == 1 ? cm[1] : b == 2 ? cm[2] : cm[3];
      const bool colok = isrhs || (iscol && ((cmb >> be) & 1));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int al = hi + 4 * r;
        if (al >= kJC || !colok || !((cma >> al) & 1)) continue;
        const bool aleft = al < 7;
        const int ra = 16 * a + (aleft ? p0 + al : al - 7);
        if (isrhs) {
          const double val = acc[tl][r] - CW[al * kJR + 14 + col - kJYrhs];
          (aleft ? U0 : U1)[(28 + col - kJYrhs) * kJB + ra] += val;
          continue;
        }
        const bool bleft = be < 7;
        if (!aleft && bleft) continue;  // (the transpose of a block that is kept)
        const double val = acc[tl][r] - (b == a ? CW[al * kJR + be] : 0.0);
        const int cb = 16 * b + (bleft ? p0 + be : be - 7);
        if (aleft && bleft) D0[cb * kJB + ra] += val;
        else if (!aleft && !bleft) D1[cb * kJB + ra] += val;
        else U0[(7 * b + be - 7) * kJB + ra] = val;
      }
    }
  }
}

// back-substitution, a wavefront per task, lane = row, every load of a task issued before the first is used:
// z = M y of interval index t:  Z[:, b] - Z[:, coupling columns] s  (s: the separators' solutions, already at their positions in b1, b2)
__device__ __forceinline__ void jstruct_zt(int t, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cl, const cfzb::glb_f64 *Zt, const cfzb::glb_f64 *b1,
                                                     const cfzb::glb_f64 *b2, cfzb::glb_f64 *zt) {
  const int lane = threadIdx.x & 63, V = m[35];
  double z1 = Zt[kJYrhs * kJB + lane], z2 = Zt[(kJYrhs + 1) * kJB + lane];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b >= V || t >= m[b]) continue;
    const cfzb::glb_i32 *clb = cl + 16 * (m[36 + b] + t);
#pragma unroll
    for (int q = 0; q < kJC; ++q) {
      const int c = clb[q];
      const double zc = Zt[jycol(b, q) * kJB + lane];
      z1 -= zc * (c >= 0 ? b1[c] : 0.0); z2 -= zc * (c >= 0 ? b2[c] : 0.0);
    }
  }
  zt[lane] = z1; zt[kJB + lane] = z2;
}
// the interior of (vehicle a, interval t):  x = W[:, b] - W[:, C] s - W[:, E] z
__device__ __forceinline__ void jstruct_back(int a, int pos0, const cfzb::glb_i32 *cl, const cfzb::glb_f64 *W, const cfzb::glb_f64 *zt, cfzb::glb_f64 *b1,
                                                       cfzb::glb_f64 *b2) {
  const int lane = threadIdx.x & 63;
  double y1 = W[14 * kSI + lane], y2 = W[15 * kSI + lane];
#pragma unroll
  for (int q = 0; q < kJC; ++q) {
    const int c = cl[q];
    const double wq = W[q * kSI + lane];
    y1 -= wq * (c >= 0 ? b1[c] : 0.0); y2 -= wq * (c >= 0 ? b2[c] : 0.0);
  }
#pragma unroll
  for (int j = 0; j < 15; ++j) { const double wq = W[(16 + j) * kSI + lane]; y1 -= wq * zt[16 * a + j]; y2 -= wq * zt[kJB + 16 * a + j]; }
  b1[pos0 + lane] = y1; b2[pos0 + lane] = y2;  // (interior positions are read by nobody in this phase)
}

__device__ __forceinline__ int jveh_of(const cfzb::glb_i32 *m, int it) { int a = 0; while (a + 1 < m[35] && it >= m[36 + a + 1]) ++a; return a; }
__device__ __attribute__((noinline)) int jstruct_interiors_all(int w0, int nw, const cfzb::glb_i32 *m, const cfzb::glb_f64 *ab, int kb, int ld, int off,
                                                               const cfzb::glb_i32 *cl, const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *Cc, cfzb::glb_f64 *W,
                                                               cfzb::lds_f64 *lds) {
  int f = 0;
  m = juni(m); ab = juni(ab); cl = juni(cl); b1 = juni(b1); b2 = juni(b2); Cc = juni(Cc); W = juni(W);
  lds = juni_lds(lds) + (threadIdx.x >> 6) * kLuLdsWave;
  w0 = juni(w0); nw = juni(nw); kb = juni(kb); ld = juni(ld); off = juni(off);
  const int NI = m[36 + m[35]];
  for (int it = w0; it < NI; it += nw) {
    const int a = jveh_of(m, it), t = it - m[36 + a];
    f |= jstruct_interior(ab, kb, ld, off, m[4 + a] + 79 * t + 14, cl + 16 * it, b1, b2, Cc + (size_t)it * kSI * kJC, W + (size_t)it * kSI * kJR, lds);
  }
  return f;
}
__device__ __attribute__((noinline)) void jstruct_cw_all(int w0, int nw, int NI, const cfzb::glb_f64 *Cc, const cfzb::glb_f64 *W, cfzb::glb_f64 *CW) {
  for (int it = w0; it < NI; it += nw) jstruct_cw(Cc + (size_t)it * kSI * kJC, W + (size_t)it * kSI * kJR, CW + (size_t)it * kJC * kJR);
}
__device__ __attribute__((noinline)) int jstruct_cap_all(int w0, int nw, int Nm, const cfzb::glb_i32 *m, const cfzb::glb_f64 *W, const cfzb::glb_f64 *pm, cfzb::glb_f64 *Z,
                                                         cfzb::lds_f64 *lds, cfzb::glb_f64 *rows) {
  int f = 0;
  m = juni(m); W = juni(W); pm = juni(pm); Z = juni(Z); w0 = juni(w0); nw = juni(nw); Nm = juni(Nm);
  lds = juni_lds(lds) + (threadIdx.x >> 6) * kLuLdsWave;
  rows = juni(rows) + (size_t)(threadIdx.x >> 6) * kJRowBuf;
  for (int k = w0; k < 2 * Nm; k += nw) f |= jstruct_cap(k >> 1, k & 1, m, W, pm, Z + (size_t)(k >> 1) * kJB * kJB, lds, rows);
  return f;
}
__device__ __attribute__((noinline)) void jstruct_sep_base_all(int w0, int nw, int Nm, const cfzb::glb_i32 *m, const cfzb::glb_f64 *ab, int kb, int ld, int off,
                                                               const cfzb::glb_f64 *pm, const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us) {
  for (int i = w0; i <= Nm; i += nw) jstruct_sep_base(i, m, ab, kb, ld, off, pm, b1, b2, Ds + (size_t)i * kJB * kJB, Us + (size_t)i * kJB * kJU);
}
__device__ __attribute__((noinline)) void jstruct_schur_all(int w0, int nw, int Nm, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cmask, const cfzb::glb_f64 *CW,
                                                            const cfzb::glb_f64 *Z, cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us) {
  for (int t = w0; t < Nm; t += nw) jstruct_schur(t, m, cmask, CW, Z + (size_t)t * kJB * kJB, Ds, Us);
}
__device__ __attribute__((noinline)) void jstruct_zt_all(int w0, int nw, int Nm, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cl, const cfzb::glb_f64 *Z,
                                                         const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *zt) {
  for (int t = w0; t < Nm; t += nw) jstruct_zt(t, m, cl, Z + (size_t)t * kJB * kJB, b1, b2, zt + (size_t)t * 2 * kJB);
}
__device__ __attribute__((noinline)) void jstruct_back_all(int w0, int nw, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cl, const cfzb::glb_f64 *W,
                                                           const cfzb::glb_f64 *zt, cfzb::glb_f64 *b1, cfzb::glb_f64 *b2) {
  const int NI = m[36 + m[35]];
  for (int it = w0; it < NI; it += nw) {
    const int a = jveh_of(m, it), t = it - m[36 + a];
    jstruct_back(a, m[4 + a] + 79 * t + 14, cl + 16 * it, W + (size_t)it * kSI * kJR, zt + (size_t)t * 2 * kJB, b1, b2);
  }
}

// The recursion over the joint separators, one wavefront per direction: side 0 eliminates blocks i0 .. i1 - 1 downwards (block i into
// i + 1), side 1 blocks i0 .. i1 + 1 upwards (block j into j - 1).  Block i: D_i (column-major), right-hand sides [U_i | b1 b2] with U_i =
// the coupling of block i's pt0 rows (row 16 a + p0 + c) with block i + 1's rows 16 b + be (column 7 b + be).  The elimination (lu64_build,
// on the matrix cores) leaves Z_i = D_i^-1 [..] in LDS, row = unknown (X, 64 x 33); every lane keeps its own row of it in global memory
// for the back-substitution (read again by the same lane), and accumulates the neighbour block's update for the row it will LOAD there
// from broadcast reads of X: the words of global memory a lane reads in the next step are words the same lane wrote (no hand-off between
// lanes through global memory, no fence; the hand-off inside the wavefront goes through LDS).  nvp: N[a] packed, 8 bits each.
constexpr int kJXld = 33;  // row stride of X (odd: the lane = row reads are conflict-free)
__device__ __attribute__((noinline)) int jstruct_chain(int side, int i0, int i1, unsigned nvp, int V, cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us, cfzb::glb_f64 *Zl,
                                                       cfzb::lds_f64 *lds, cfzb::lds_f64 *X) {
  const int lane = threadIdx.x & 63, va = lane >> 4, la = lane & 15;
  const int cu = la < 7 ? 7 * va + la : -1;  // this lane's row as a right-coupled row of its block: its column of U
  Ds = juni(Ds); Us = juni(Us); Zl = juni(Zl); lds = juni_lds(lds); X = juni_lds(X); side = juni(side); i0 = juni(i0); i1 = juni(i1); V = juni(V); nvp = (unsigned)juni((int)nvp);
  for (int i = i0; side ? i > i1 : i < i1; i += side ? -1 : 1) {
    const cfzb::glb_f64 *Di = Ds + (size_t)i * kJB * kJB, *Ui = Us + (size_t)i * kJB * kJU;
    const int ip = side ? i - 1 : i;                 // the coupling block involved: U_ip couples blocks ip and ip + 1
    const int p0 = ip == 0 ? 7 : 8;                  // pt0's first local row in block ip
    const cfzb::glb_f64 *Uc = Us + (size_t)ip * kJB * kJU;
    if (lu64_build<32, kJXld, 1, cfzb::lds_f64>(lds, X, [&](auto Jc, double (&v)[16]) {
          constexpr int J = decltype(Jc)::value;
          if constexpr (J < 4) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = Di[(16 * J + k) * kJB + lane];
          } else if (side == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int q = 16 * (J - 4) + k; v[k] = q < 30 ? Ui[q * kJB + lane] : 0.0; }
          } else {
            const int cuc = cu >= 0 ? cu : 0;  // (clamped: the loads of a tile stand in one block, their results masked)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              const int q = 16 * (J - 4) + k;
              const double x = q < 28 ? Uc[cuc * kJB + 16 * (q / 7) + p0 + (q % 7)] : (q < 30 ? Ui[q * kJB + lane] : 0.0);  // U_{i-1}' (zero where a vehicle has no interior i - 1)
              v[k] = (q < 28 && cu < 0) ? 0.0 : x;
            }
          }
        })) return 1;
    lu_wave_sync();
    cfzb::glb_f64 *Zi = Zl + (size_t)i * kJB * kJU;
#pragma unroll
    for (int q = 0; q < 30; ++q) Zi[q * kJB + lane] = X[lane * kJXld + q];
    // the neighbour block's update: acc[q] = sum over the coupled rows r of this block of U[r, this lane's row there] Z[r, q]
    double acc[30];
#pragma unroll
    for (int q = 0; q < 30; ++q) acc[q] = 0.0;
    if (side == 0) {
      for (int b = 0; b < V; ++b) {
        if (i >= (int)((nvp >> (8 * b)) & 255u)) continue;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          const int r = 16 * b + p0 + c;
          const double u = cu >= 0 ? Uc[cu * kJB + r] : 0.0;
#pragma unroll
          for (int q = 0; q < 30; ++q) acc[q] += u * X[r * kJXld + q];
        }
      }
      if (cu >= 0) {
        cfzb::glb_f64 *Dn = Ds + (size_t)(i + 1) * kJB * kJB, *Un = Us + (size_t)(i + 1) * kJB * kJU;
#pragma unroll
        for (int q = 0; q < 28; ++q) Dn[(16 * (q / 7) + (q % 7)) * kJB + lane] -= acc[q];
        Un[28 * kJB + lane] -= acc[28]; Un[29 * kJB + lane] -= acc[29];
      }
    } else {
      const bool isl = la >= p0 && la < p0 + 7 && ip < (int)((nvp >> (8 * va)) & 255u);  // this lane's row is a pt0 row of block i - 1
      for (int b = 0; b < V; ++b) {
        if (ip >= (int)((nvp >> (8 * b)) & 255u)) continue;
#pragma unroll
        for (int be = 0; be < 7; ++be) {
          const int r = 16 * b + be;
          const double u = isl ? Uc[(7 * b + be) * kJB + lane] : 0.0;
#pragma unroll
          for (int q = 0; q < 30; ++q) acc[q] += u * X[r * kJXld + q];
        }
      }
      if (isl) {
        cfzb::glb_f64 *Dn = Ds + (size_t)ip * kJB * kJB, *Un = Us + (size_t)ip * kJB * kJU;
#pragma unroll
        for (int q = 0; q < 28; ++q) Dn[(16 * (q / 7) + p0 + (q % 7)) * kJB + lane] -= acc[q];
        Un[28 * kJB + lane] -= acc[28]; Un[29 * kJB + lane] -= acc[29];
      }
    }
    lu_wave_sync();  // (X is rewritten by the next block's elimination)
  }
  return 0;
}
// the middle block (both neighbours folded in) and the back-substitution outwards: side 0 blocks mid - 1 .. 0, side 1 blocks mid + 1 .. Nm;
// xs[i][s][row] receives the solutions.  Side 1 takes x_mid from xs (written by side 0 before a workgroup barrier: `phase`)
__device__ __attribute__((noinline)) int jstruct_chain_mid(int mid, cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us, cfzb::glb_f64 *xs) {
  const int lane = threadIdx.x & 63;
  const cfzb::glb_f64 *Di = Ds + (size_t)mid * kJB * kJB, *Ui = Us + (size_t)mid * kJB * kJU;
  double a[kJB + 2];
#pragma unroll
  for (int j = 0; j < kJB; ++j) a[j] = Di[j * kJB + lane];
  a[kJB] = Ui[28 * kJB + lane]; a[kJB + 1] = Ui[29 * kJB + lane];
  int ord;
  if (wave_lu_regs<kJB, 2>(a, lane, ord)) return 1;
  xs[(size_t)mid * 2 * kJB + ord] = a[kJB]; xs[(size_t)mid * 2 * kJB + kJB + ord] = a[kJB + 1];
  return 0;
}
__device__ __attribute__((noinline)) void jstruct_chain_back(int side, int mid, int Nm, const cfzb::glb_f64 *Zl, cfzb::glb_f64 *xs) {
  const int lane = threadIdx.x & 63;
  double xp1 = xs[(size_t)mid * 2 * kJB + lane], xp2 = xs[(size_t)mid * 2 * kJB + kJB + lane];  // x of the block before, lane = unknown
  for (int i = side ? mid + 1 : mid - 1; side ? i <= Nm : i >= 0; i += side ? 1 : -1) {
    const cfzb::glb_f64 *Zi = Zl + (size_t)i * kJB * kJU;
    const int ip = side ? i - 1 : i, p0 = ip == 0 ? 7 : 8;
    double x1 = Zi[28 * kJB + lane], x2 = Zi[29 * kJB + lane];
#pragma unroll
    for (int q = 0; q < 28; ++q) {
      const int r = side ? 16 * (q / 7) + p0 + (q % 7) : 16 * (q / 7) + (q % 7);  // the unknown of the block before that column q stands for
      const double z = Zi[q * kJB + lane];
      x1 -= z * struct_lane_get(xp1, r); x2 -= z * struct_lane_get(xp2, r);
    }
    xs[(size_t)i * 2 * kJB + lane] = x1; xs[(size_t)i * 2 * kJB + kJB + lane] = x2;
    xp1 = x1; xp2 = x2;
  }
}

// ---- the recursion over the joint separators by block CYCLIC REDUCTION (round 6) ------------------------------------------
// The chain above keeps two of the workgroup's eight wavefronts busy for 26 dependent steps each.  The separator system is block
// tridiagonal with a special coupling: block i and its right neighbour meet only in 28 x 28 entries -- the pt0 rows of block i (row
// 16 a + p0 + c, p0 = 7 for block 0 and 8 otherwise) against the right-coupled rows of the neighbour (row 16 b + be) -- and eliminating a
// block leaves exactly such a coupling between ITS two neighbours:
//   D_l[P, P] -= C_l G_FF C_l',   D_r[F, F] -= C_i' G_PP C_i,   C_l <- -C_l G_FP C_i     (G = D_i^-1; F, P: the block's right-coupled / pt0 rows)
// so every other block can leave at once, level by level (strides 1, 2, 4, ..: 25 + 13 + 6 + 3 + 2 + 1 blocks for 51), each needing the
// columns F and P of its inverse: two 64-row eliminations on the matrix cores with 30 and 28 right-hand sides (unit vectors and the two
// right-hand sides), all eight wavefronts at work.  tools/joint_condense_study.py: as accurate as the chain (the blocks are as well
// conditioned without their neighbours' updates as with them: 1e12-6e15 either way, solutions equal to 5e-14 .. 2e-9).
// Per block: Za = G [E_F | r] (64 x 30), Zb = G E_P (64 x 28), Lc = the left link as it stood (28 x 28), all for the back-substitution
//   x_i = g - G[:, F] (C_l' x_l[P]) - G[:, P] (C_i x_r[F]).
__device__ __forceinline__ int jrowF(int q) { return 16 * (q / 7) + (q % 7); }              // right-coupled unknown q = 7 b + be of a block
__device__ __forceinline__ int jrowP(int q, int p0) { return 16 * (q / 7) + p0 + (q % 7); }  // pt0 unknown q = 7 a + c of a block
__device__ __forceinline__ int jbcr_count(int st, int Nm) { return (Nm / st + 1) / 2; }      // blocks st, 3 st, 5 st, .. <= Nm

// a level's eliminations: task t = (block st (2 (t >> 1) + 1), half t & 1); half 0: Za = D^-1 [E_F | r], half 1: Zb = D^-1 [E_P | 0]
__device__ __attribute__((noinline)) int jbcr_lu_all(int w0, int nw, int st, int Nm, const cfzb::glb_f64 *Ds, const cfzb::glb_f64 *Us, cfzb::glb_f64 *Za, cfzb::glb_f64 *Zb,
                                                     cfzb::lds_f64 *lds) {
  Ds = juni(Ds); Us = juni(Us); Za = juni(Za); Zb = juni(Zb); w0 = juni(w0); nw = juni(nw); st = juni(st); Nm = juni(Nm);
  lds = juni_lds(lds) + (threadIdx.x >> 6) * kLuLdsWave;
  // (st = 0: the block that remains, block 0, with its two right-hand sides: one task)
  const int cnt = st ? jbcr_count(st, Nm) : 0, ntask = st ? 2 * cnt : 1, lane0 = threadIdx.x & 63;
  int f = 0;
  for (int t = w0; t < ntask; t += nw) {
    const int i = st * (2 * (t >> 1) + 1), half = t & 1, lane = lu_opaque(lane0);
    const cfzb::glb_f64 *Di = Ds + (size_t)i * kJB * kJB, *Ui = Us + (size_t)i * kJB * kJU;
    cfzb::glb_f64 *out = (half ? Zb : Za) + (size_t)i * kJB * kJU;
    f |= lu64_build<32>(lds, out, [&](auto Jc, double (&v)[16]) {
      constexpr int J = decltype(Jc)::value;
      if constexpr (J < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Di[(16 * J + k) * kJB + lane];
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int q = 16 * (J - 4) + k;
          if (q < kJL) v[k] = lane == (half ? jrowP(q, 8) : jrowF(q)) ? 1.0 : 0.0;  // (a block that leaves is never block 0: p0 = 8)
          else if (q < 30) v[k] = half ? 0.0 : Ui[q * kJB + lane];
          else v[k] = 0.0;
        }
      }
    });
  }
  return f;
}
// X = A (G B) for operands of at most 28 x 28 given as element functions (indices run to 32: the functions return zeros beyond 28), on the
// matrix cores: T = G B as 2 x 2 tiles of v_mfma_f64_16x16x4_f64 (eight k-steps each), whose accumulators ARE the B operands of the second
// product as they stand (register r of a tile holds row (lane >> 4) + 4 r: the rows a k-step r asks of this lane); vec(i, s): two more
// columns (28, 29) put into T before the second product (X[:, 28 + s] = A vec[:, s]); out(i, j, X[i][j]) for every entry this lane holds.
typedef double jbcr_v4 __attribute__((ext_vector_type(4)));
template <bool VEC, class FA, class FG, class FB, class FV, class FO>
__device__ __forceinline__ void jbcr_triple(FA a, FG g, FB b, FV vec, FO out) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  double ga[2][8], gb[2][8], aa[2][2][4];
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int k = 0; k < 8; ++k) { ga[I][k] = g(16 * I + lo, 4 * k + hi); gb[I][k] = b(4 * k + hi, 16 * I + lo); }
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
      for (int r = 0; r < 4; ++r) aa[I][Ip][r] = a(16 * I + lo, 16 * Ip + 4 * r + hi);
  jbcr_v4 T[2][2];
#pragma unroll
  for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
    for (int J = 0; J < 2; ++J) {
      jbcr_v4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 8; ++k) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[Ip][k], gb[J][k], t, 0, 0, 0);
      T[Ip][J] = t;
    }
  if (VEC) {
    if (lo == 12 || lo == 13) {
#pragma unroll
      for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[Ip][1][r] = vec(16 * Ip + hi + 4 * r, lo - 12);
    }
  }
#pragma unroll
  for (int I = 0; I < 2; ++I)
#pragma unroll
    for (int J = 0; J < 2; ++J) {
      jbcr_v4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int Ip = 0; Ip < 2; ++Ip)
#pragma unroll
        for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[I][Ip][r], T[Ip][J][r], x, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) out(16 * I + hi + 4 * r, 16 * J + lo, x[r]);
    }
}
// a level's updates: one wavefront per leaving block, three triple products of 28 x 28 operands read where they lie (the links in Us, the
// inverse's columns in Za / Zb): no LDS.
__device__ __attribute__((noinline)) void jbcr_update_all(int w0, int nw, int st, int Nm, cfzb::glb_f64 *Ds, cfzb::glb_f64 *Us, const cfzb::glb_f64 *Za, const cfzb::glb_f64 *Zb,
                                                          cfzb::glb_f64 *Lc) {
  Ds = juni(Ds); Us = juni(Us); Za = juni(Za); Zb = juni(Zb); Lc = juni(Lc); w0 = juni(w0); nw = juni(nw); st = juni(st); Nm = juni(Nm);
  const int cnt = jbcr_count(st, Nm);
  for (int t = w0; t < cnt; t += nw) {
    const int i = st * (2 * t + 1), l = i - st, r = i + st <= Nm ? i + st : -1, p0l = l == 0 ? 7 : 8;
    cfzb::glb_f64 *Dl = Ds + (size_t)l * kJB * kJB, *Ul = Us + (size_t)l * kJB * kJU, *Lci = Lc + (size_t)i * kJL * kJL;
    const cfzb::glb_f64 *Ui = Us + (size_t)i * kJB * kJU, *Zai = Za + (size_t)i * kJB * kJU, *Zbi = Zb + (size_t)i * kJB * kJU;
    // (every element function clamps its indices to 27 for the load and masks the value: the loads of a product stand in one block)
    auto in = [](int x) { return x < kJL ? x : kJL - 1; };
    auto CL = [&](int rho, int f) { const double v = Ul[in(f) * kJB + jrowP(in(rho), p0l)]; return (rho < kJL && f < kJL) ? v : 0.0; };  // the left link
    auto CR = [&](int p, int fr) { const double v = Ui[in(fr) * kJB + jrowP(in(p), 8)]; return (p < kJL && fr < kJL) ? v : 0.0; };        // the right link
    // D_l[P, P] -= CL GFF CL',  r_l[P] -= CL gF; the link as it stands is kept for the back-substitution
    jbcr_triple<true>([&](int rho, int f) { const double v = CL(rho, f); if (rho < kJL && f < kJL) Lci[rho * kJL + f] = v; return v; },
                      [&](int f, int f2) { const double v = Zai[in(f2) * kJB + jrowF(in(f))]; return (f < kJL && f2 < kJL) ? v : 0.0; },
                      [&](int f2, int rho) { return CL(rho, f2); },
                      [&](int f, int s_) { const double v = Zai[(28 + s_) * kJB + jrowF(in(f))]; return f < kJL ? v : 0.0; },
                      [&](int rho, int j, double v) {
                        if (rho >= kJL || j >= 30) return;
                        if (j < kJL) Dl[jrowP(j, p0l) * kJB + jrowP(rho, p0l)] -= v; else Ul[j * kJB + jrowP(rho, p0l)] -= v;
                      });
    if (r < 0) continue;  // (the last block of its level may have no right neighbour: block l's link is never read again)
    cfzb::glb_f64 *Dr = Ds + (size_t)r * kJB * kJB, *Ur = Us + (size_t)r * kJB * kJU;
    // D_r[F, F] -= CR' GPP CR,  r_r[F] -= CR' gP
    jbcr_triple<true>([&](int fr, int p) { return CR(p, fr); },
                      [&](int p, int p2) { const double v = Zbi[in(p2) * kJB + jrowP(in(p), 8)]; return (p < kJL && p2 < kJL) ? v : 0.0; },
                      [&](int p2, int fr) { return CR(p2, fr); },
                      [&](int p, int s_) { const double v = Zai[(28 + s_) * kJB + jrowP(in(p), 8)]; return p < kJL ? v : 0.0; },
                      [&](int fr, int j, double v) {
                        if (fr >= kJL || j >= 30) return;
                        if (j < kJL) Dr[jrowF(j) * kJB + jrowF(fr)] -= v; else Ur[j * kJB + jrowF(fr)] -= v;
                      });
    // the new link of block l (with block r): -CL GFP CR -- last: it overwrites CL
    jbcr_triple<false>([&](int rho, int f) { return CL(rho, f); },
                       [&](int f, int p) { const double v = Zbi[in(p) * kJB + jrowF(in(f))]; return (f < kJL && p < kJL) ? v : 0.0; },
                       [&](int p, int fr) { return CR(p, fr); },
                       [&](int, int) { return 0.0; },
                       [&](int rho, int fr, double v) { if (rho < kJL && fr < kJL) Ul[fr * kJB + jrowP(rho, p0l)] = -v; });
  }
}
// a level's back-substitution (levels in reverse): x_i = g - G[:, F] (CL' x_l[P]) - G[:, P] (CR x_r[F]), lane = row of block i
__device__ __attribute__((noinline)) void jbcr_back_all(int w0, int nw, int st, int Nm, const cfzb::glb_f64 *Us, const cfzb::glb_f64 *Za, const cfzb::glb_f64 *Zb, const cfzb::glb_f64 *Lc,
                                                        cfzb::glb_f64 *xs) {
  Us = juni(Us); Za = juni(Za); Zb = juni(Zb); Lc = juni(Lc); xs = juni(xs); w0 = juni(w0); nw = juni(nw); st = juni(st); Nm = juni(Nm);
  const int cnt = jbcr_count(st, Nm), lane = threadIdx.x & 63;
  const bool on = lane < kJL;
  const int q = on ? lane : 0, rP = jrowP(q, 8);
  for (int t = w0; t < cnt; t += nw) {
    const int i = st * (2 * t + 1), l = i - st, r = i + st <= Nm ? i + st : -1, p0l = l == 0 ? 7 : 8;
    const cfzb::glb_f64 *Ui = Us + (size_t)i * kJB * kJU, *Zai = Za + (size_t)i * kJB * kJU, *Zbi = Zb + (size_t)i * kJB * kJU, *Lci = Lc + (size_t)i * kJL * kJL;
    const cfzb::glb_f64 *xl = xs + (size_t)l * 2 * kJB, *xr = xs + (size_t)(r >= 0 ? r : l) * 2 * kJB;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;  // lane f: (CL' x_l[P])[f]; lane p: (CR x_r[F])[p]
#pragma unroll
    for (int rho = 0; rho < kJL; ++rho) { const double c = Lci[rho * kJL + q]; a0 += c * xl[jrowP(rho, p0l)]; a1 += c * xl[kJB + jrowP(rho, p0l)]; }
    if (r >= 0) {
#pragma unroll
      for (int fr = 0; fr < kJL; ++fr) { const double c = Ui[fr * kJB + rP]; b0 += c * xr[jrowF(fr)]; b1 += c * xr[kJB + jrowF(fr)]; }
    }
    double x0 = Zai[28 * kJB + lane], x1 = Zai[29 * kJB + lane];
#pragma unroll
    for (int f = 0; f < kJL; ++f) { const double z = Zai[f * kJB + lane]; x0 -= z * struct_lane_get(a0, f); x1 -= z * struct_lane_get(a1, f); }
#pragma unroll
    for (int p = 0; p < kJL; ++p) { const double z = Zbi[p * kJB + lane]; x0 -= z * struct_lane_get(b0, p); x1 -= z * struct_lane_get(b1, p); }
    xs[(size_t)i * 2 * kJB + lane] = x0; xs[(size_t)i * 2 * kJB + kJB + lane] = x1;
  }
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

// n independent items, item tt: store(tt, load(tt)).  The workgroup is alone on its CU with two wavefronts per SIMD: a loop that issues
// one item's loads and waits for them is bound by the memory latency (1-2 us per item), so four items' loads are issued before the first
// is used (clamped index instead of a branch: the loads of all four stand in one block).
// ---- one vehicle (cfz_colloc, round 5): the same partition without the pair coupling ------------------------------------------------------------
// No capacitance systems and no E columns: an interior is K^-1 [C | b1 b2] (64 x 64, 16 right-hand sides), the separators are 16-row blocks
// (15 unknowns, identity-padded) with 7 coupling columns and b1, b2, and their blocks are written with the Schur complements -C'W in place
// (owner computes).  Compact storage of its own: D1[i][16 x 16], U1[i][16 x 10] (columns 0..6: the coupling with separator i + 1, 7 / 8:
// b1 / b2), Z1 likewise, lane-major.  It replaces cfz_struct.inl's path (separators of 14-31 with the tube rows inside, a recursion that
// handed rows over between the lanes of a wavefront through memory), which left the library in round 6.  (The separators of ONE vehicle stay a
// chain from both ends: their cyclic reduction -- 16-row blocks, 7 x 7 couplings, sixteen right-hand sides per leaving block -- was built in
// round 6 and measured slower, a 16-row elimination being too small for the fixed costs of a level: docs/notebook.md.)
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int kJ1U = 10;
__device__ __attribute__((noinline)) int j1_interiors_all(int w0, int nw, const cfzb::glb_i32 *m, const cfzb::glb_f64 *ab, int kb, int ld, int off,
                                                          const cfzb::glb_i32 *cl_, const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *Cc, cfzb::glb_f64 *Wc,
                                                          cfzb::lds_f64 *lds) {
  int f = 0;
  m = juni(m); ab = juni(ab); cl_ = juni(cl_); b1 = juni(b1); b2 = juni(b2); Cc = juni(Cc); Wc = juni(Wc);
  lds = juni_lds(lds) + (threadIdx.x >> 6) * kLuLdsWave;
  w0 = juni(w0); nw = juni(nw); kb = juni(kb); ld = juni(ld); off = juni(off);
  const int NI = m[0], lane0 = threadIdx.x & 63;
  for (int it = w0; it < NI; it += nw) {
    const int lane = lu_opaque(lane0), pi = m[4] + 79 * it + 14, r = pi + lane;
    const cfzb::glb_i32 *cl = cl_ + 16 * it;
    cfzb::glb_f64 *C = Cc + (size_t)it * kSI * kJC, *W = Wc + (size_t)it * kSI * kJR;
    f |= lu64_build<16>(lds, W, [&](auto Jc, double (&v)[16]) {  // W[q * kSI + unknown]
      constexpr int J = decltype(Jc)::value;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if constexpr (J < 4) { const int c = pi + 16 * J + k, dd = r - c; v[k] = (dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0; }
        else if (k < kJC) {
          const int c = cl[k], dd = r - c;
          v[k] = (c >= 0 && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0;
          C[k * kSI + lane] = v[k];
        } else v[k] = k == 14 ? b1[r] : b2[r];
      }
    });
  }
  return f;
}
// C'W (14 x 64 x 16): one tile of the matrix cores per interior
__device__ __attribute__((noinline)) void j1_cw_all(int w0, int nw, int NI, const cfzb::glb_f64 *Cc, const cfzb::glb_f64 *Wc, cfzb::glb_f64 *CWc) {
  const int lane = threadIdx.x & 63, lo = lane & 15, hi = lane >> 4;
  for (int it = w0; it < NI; it += nw) {
    const cfzb::glb_f64 *C = Cc + (size_t)it * kSI * kJC, *W = Wc + (size_t)it * kSI * kJR;
    jstruct_v4 acc = {0.0, 0.0, 0.0, 0.0};
    double av[16], bv[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { av[kk] = lo < kJC ? C[lo * kSI + 4 * kk + hi] : 0.0; bv[kk] = W[lo * kSI + 4 * kk + hi]; }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int al = hi + 4 * r; if (al < kJC) CWc[(size_t)it * kJC * kJR + al * kJR + lo] = acc[r]; }
  }
}
// separator blocks with the Schur complements in place: four separators per wavefront (lane = 16 x separator-in-group + row)
__device__ __attribute__((noinline)) void j1_sep_all(int w0, int nw, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cmask, const cfzb::glb_f64 *ab, int kb, int ld, int off,
                                                     const cfzb::glb_f64 *CWc, const cfzb::glb_f64 *b1, const cfzb::glb_f64 *b2, cfzb::glb_f64 *D1, cfzb::glb_f64 *U1) {
  const int lane = threadIdx.x & 63, la = lane & 15, N = m[0], hf = m[8];
  for (int i0 = 4 * w0; i0 <= N; i0 += 4 * nw) {
    const int i = i0 + (lane >> 4);
    if (i > N) continue;
    const int ns = i == 0 ? 14 : (i < N ? 15 : 5 + hf), ps = i == 0 ? m[4] : m[4] + 79 * i - 1, p0 = i == 0 ? 7 : 8, r = ps + la;
    const bool rowok = la < ns, left = rowok && i < N && la >= p0 && la < p0 + 7;
    const int cmp = i >= 1 ? cmask[i - 1] : 0, cmi = i < N ? cmask[i] : 0;  // (coupling masks of the interiors on either side)
    const bool right = rowok && i >= 1 && la < 7 && ((cmp >> (7 + la)) & 1);
    const cfzb::glb_f64 *CWl = CWc + (size_t)(i < N ? i : 0) * kJC * kJR + (left ? la - p0 : 0) * kJR;      // row al = la - p0 of interior i
    const cfzb::glb_f64 *CWr = CWc + (size_t)(i >= 1 ? i - 1 : 0) * kJC * kJR + (right ? 7 + la : 0) * kJR;  // row al = 7 + la of interior i - 1
    cfzb::glb_f64 *D = D1 + (size_t)i * 256, *U = U1 + (size_t)i * 16 * kJ1U;
#pragma unroll
    for (int lb = 0; lb < 16; ++lb) {
      double v = la == lb ? 1.0 : 0.0;
      if (rowok) {
        const int c = ps + lb, dd = r - c;
        v = (lb < ns && dd <= kb && -dd <= kb) ? ab[(size_t)c * ld + (off + dd)] : 0.0;
        if (left && lb >= p0 && lb < p0 + 7) v -= CWl[lb - p0];
        if (right && lb < 7 && ((cmp >> (7 + lb)) & 1)) v -= CWr[7 + lb];
      }
      D[lb * 16 + la] = v;
    }
#pragma unroll
    for (int be = 0; be < 7; ++be) U[be * 16 + la] = (left && ((cmi >> (7 + be)) & 1)) ? -CWl[7 + be] : 0.0;
#pragma unroll
    for (int sr = 0; sr < 2; ++sr) {
      double v = rowok ? (sr ? b2 : b1)[r] : 0.0;
      if (left) v -= CWl[14 + sr];
      if (right) v -= CWr[14 + sr];
      U[(7 + sr) * 16 + la] = v;
    }
    U[9 * 16 + la] = 0.0;
  }
}
// the recursion from one end (as jstruct_chain): 16-row blocks, 10 right-hand sides, lanes 16.. idle
__device__ __attribute__((noinline)) int j1_chain(int side, int i0, int i1, cfzb::glb_f64 *D1, cfzb::glb_f64 *U1, cfzb::glb_f64 *Z1, cfzb::glb_i32 *ordl) {
  const int lane = threadIdx.x & 63, la = lane & 15;
  D1 = juni(D1); U1 = juni(U1); Z1 = juni(Z1); ordl = juni(ordl); side = juni(side); i0 = juni(i0); i1 = juni(i1);
  for (int i = i0; side ? i > i1 : i < i1; i += side ? -1 : 1) {
    const cfzb::glb_f64 *Di = D1 + (size_t)i * 256, *Ui = U1 + (size_t)i * 16 * kJ1U;
    const int ip = side ? i - 1 : i, p0 = ip == 0 ? 7 : 8;
    const cfzb::glb_f64 *Uc = U1 + (size_t)ip * 16 * kJ1U;
    double a[16 + kJ1U];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = Di[j * 16 + la];
    if (side == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) a[16 + q] = Ui[q * 16 + la];
    } else {
#pragma unroll
      for (int q = 0; q < 7; ++q) a[16 + q] = la < 7 ? Uc[la * 16 + p0 + q] : 0.0;  // U_{i-1}' (row la of this block = right-coupled unknown la)
      a[16 + 7] = Ui[7 * 16 + la]; a[16 + 8] = Ui[8 * 16 + la];
    }
    a[16 + 9] = 0.0;
    int ord;
    if (wave_lu_regs<16, kJ1U>(a, lane, ord)) return 1;
    cfzb::glb_f64 *Zi = Z1 + (size_t)i * 16 * kJ1U;
    if (lane < 16) {
#pragma unroll
      for (int q = 0; q < 9; ++q) Zi[q * 16 + lane] = a[16 + q];
      ordl[i * 16 + lane] = ord;
    }
    double acc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[q] = 0.0;
    if (side == 0) {
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        const int lk = (int)__builtin_ctzll(__ballot(ord == p0 + c));
        const double u = la < 7 ? Uc[la * 16 + p0 + c] : 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] += u * struct_lane_get(a[16 + q], lk);
      }
      if (lane < 7) {
        cfzb::glb_f64 *Dn = D1 + (size_t)(i + 1) * 256, *Un = U1 + (size_t)(i + 1) * 16 * kJ1U;
#pragma unroll
        for (int q = 0; q < 7; ++q) Dn[q * 16 + lane] -= acc[q];
        Un[7 * 16 + lane] -= acc[7]; Un[8 * 16 + lane] -= acc[8];
      }
    } else {
      const bool isl = lane < 16 && la >= p0 && la < p0 + 7;
#pragma unroll
      for (int be = 0; be < 7; ++be) {
        const int lk = (int)__builtin_ctzll(__ballot(ord == be));
        const double u = isl ? Uc[be * 16 + la] : 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] += u * struct_lane_get(a[16 + q], lk);
      }
      if (isl) {
        cfzb::glb_f64 *Dn = D1 + (size_t)ip * 256, *Un = U1 + (size_t)ip * 16 * kJ1U;
#pragma unroll
        for (int q = 0; q < 7; ++q) Dn[(p0 + q) * 16 + la] -= acc[q];
        Un[7 * 16 + la] -= acc[7]; Un[8 * 16 + la] -= acc[8];
      }
    }
  }
  return 0;
}
__device__ __attribute__((noinline)) int j1_chain_mid(int mid, cfzb::glb_f64 *D1, cfzb::glb_f64 *U1, cfzb::glb_f64 *xs) {
  const int lane = threadIdx.x & 63, la = lane & 15;
  const cfzb::glb_f64 *Di = D1 + (size_t)mid * 256, *Ui = U1 + (size_t)mid * 16 * kJ1U;
  double a[16 + 2];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = Di[j * 16 + la];
  a[16] = Ui[7 * 16 + la]; a[17] = Ui[8 * 16 + la];
  int ord;
  if (wave_lu_regs<16, 2>(a, lane, ord)) return 1;
  if (lane < 16) { xs[(size_t)mid * 32 + ord] = a[16]; xs[(size_t)mid * 32 + 16 + ord] = a[17]; }
  return 0;
}
__device__ __attribute__((noinline)) void j1_chain_back(int side, int mid, int N, const cfzb::glb_f64 *Z1, const cfzb::glb_i32 *ordl, cfzb::glb_f64 *xs) {
  const int lane = threadIdx.x & 63, la = lane & 15;
  double xp1 = xs[(size_t)mid * 32 + la], xp2 = xs[(size_t)mid * 32 + 16 + la];
  int ordp = lane < 16 ? lane : -1;
  for (int i = side ? mid + 1 : mid - 1; side ? i <= N : i >= 0; i += side ? 1 : -1) {
    const cfzb::glb_f64 *Zi = Z1 + (size_t)i * 16 * kJ1U;
    const int ip = side ? i - 1 : i, p0 = ip == 0 ? 7 : 8;
    double x1 = Zi[7 * 16 + la], x2 = Zi[8 * 16 + la];
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int r = side ? p0 + q : q;
      const int lk = (int)__builtin_ctzll(__ballot(ordp == r));
      const double z = Zi[q * 16 + la];
      x1 -= z * struct_lane_get(xp1, lk); x2 -= z * struct_lane_get(xp2, lk);
    }
    const int ord = lane < 16 ? ordl[i * 16 + lane] : -1;
    if (lane < 16) { xs[(size_t)i * 32 + ord] = x1; xs[(size_t)i * 32 + 16 + ord] = x2; }
    xp1 = x1; xp2 = x2; ordp = ord;
  }
}
__device__ __attribute__((noinline)) void j1_back_all(int w0, int nw, const cfzb::glb_i32 *m, const cfzb::glb_i32 *cl_, const cfzb::glb_f64 *Wc, cfzb::glb_f64 *b1, cfzb::glb_f64 *b2) {
  const int NI = m[0], lane = threadIdx.x & 63;
  for (int it = w0; it < NI; it += nw) {
    const cfzb::glb_i32 *cl = cl_ + 16 * it;
    const cfzb::glb_f64 *W = Wc + (size_t)it * kSI * kJR;
    double y1 = W[14 * kSI + lane], y2 = W[15 * kSI + lane];
#pragma unroll
    for (int q = 0; q < kJC; ++q) {
      const int c = cl[q];
      const double wq = W[q * kSI + lane];
      y1 -= wq * (c >= 0 ? b1[c] : 0.0); y2 -= wq * (c >= 0 ? b2[c] : 0.0);
    }
    const int p = m[4] + 79 * it + 14 + lane;
    b1[p] = y1; b2[p] = y2;
  }
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// the single vehicle's solve on the device (see above); same contract as jstruct_solve
__device__ inline int jstruct_solve1(const CSpec &sp, const CDims &d, const CWork &w, const JWork &s, const Band &B, double *b1, double *b2, long long *ptk, double *lds) {
  double *flag = s.flag;
  long long tp = tick();
  const int N = sp.N[0], wv = CFZS_WAVE, nw = CFZS_NW;
  if (flag[1] != 0.0) return 1;
  if (threadIdx.x == 0) flag[0] = 0.0;
  __syncthreads();
  cfzb::glb_f64 *D1 = (cfzb::glb_f64 *)s.Ds, *U1 = (cfzb::glb_f64 *)s.Us, *Z1 = (cfzb::glb_f64 *)s.Zs, *xs = (cfzb::glb_f64 *)s.xs;
  const cfzb::glb_i32 *meta = (const cfzb::glb_i32 *)s.meta;
  if (j1_interiors_all(wv, nw, meta, (const cfzb::glb_f64 *)B.ab, B.kb, B.ld, B.off, (const cfzb::glb_i32 *)s.cl, (const cfzb::glb_f64 *)b1, (const cfzb::glb_f64 *)b2,
                       (cfzb::glb_f64 *)s.Cc, (cfzb::glb_f64 *)s.W, cfzb::opaque((cfzb::lds_f64 *)lds)) && CFZS_LANE == 0) flag[0] = 1.0;
  __syncthreads();
  { const long long t1 = tick(); ptk[0] += t1 - tp; tp = t1; }
  if (flag[0] != 0.0) return 1;
  j1_cw_all(wv, nw, N, (const cfzb::glb_f64 *)s.Cc, (const cfzb::glb_f64 *)s.W, (cfzb::glb_f64 *)s.CW);
  __syncthreads();
  j1_sep_all(wv, nw, meta, (const cfzb::glb_i32 *)s.cmask, (const cfzb::glb_f64 *)B.ab, B.kb, B.ld, B.off, (const cfzb::glb_f64 *)s.CW, (const cfzb::glb_f64 *)b1, (const cfzb::glb_f64 *)b2, D1, U1);
  __syncthreads();
  { const long long t1 = tick(); ptk[1] += t1 - tp; tp = t1; }
  const int mid = (N + 1) / 2;
  int f = 0;
  if (wv == 0) f = j1_chain(0, 0, mid, D1, U1, Z1, (cfzb::glb_i32 *)s.ordl);
  else if (wv == 1) f = j1_chain(1, N, mid, D1, U1, Z1, (cfzb::glb_i32 *)s.ordl);
  if (f && CFZS_LANE == 0) flag[0] = 1.0;
  __syncthreads();
  if (flag[0] != 0.0) return 1;
  if (wv == 0 && j1_chain_mid(mid, D1, U1, xs) && CFZS_LANE == 0) flag[0] = 1.0;
  __syncthreads();
  if (flag[0] != 0.0) return 1;
  if (wv < 2) j1_chain_back(wv, mid, N, (const cfzb::glb_f64 *)Z1, (const cfzb::glb_i32 *)s.ordl, xs);
  __syncthreads();
  for (int t = (int)threadIdx.x; t < (N + 1) * 16; t += (int)blockDim.x) {
    const int i = t >> 4, la = t & 15, ns = i == 0 ? 14 : (i < N ? 15 : 5 + (sp.has_final[0] ? 1 : 0));
    if (la < ns) { const int p = (i == 0 ? s.bs[0] : s.bs[0] + 79 * i - 1) + la; b1[p] = s.xs[(size_t)i * 32 + la]; b2[p] = s.xs[(size_t)i * 32 + 16 + la]; }
  }
  __syncthreads();
  j1_back_all(wv, nw, meta, (const cfzb::glb_i32 *)s.cl, (const cfzb::glb_f64 *)s.W, (cfzb::glb_f64 *)b1, (cfzb::glb_f64 *)b2);
  __syncthreads();
  { const long long t1 = tick(); ptk[2] += t1 - tp; }
  return 0;
}
#endif

template <class L, class S>
CFZP_FN void jstruct_map(int n, L load, S store) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int nt = (int)blockDim.x;
  for (int t0 = (int)threadIdx.x; t0 < n; t0 += 4 * nt) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int tt = t0 + u * nt; v[u] = load(tt < n ? tt : n - 1); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int tt = t0 + u * nt; if (tt < n) store(tt, v[u]); }
  }
#else
  for (int tt = 0; tt < n; ++tt) store(tt, load(tt));
#endif
}

// The whole solve: on return b1, b2 (positions of build_order_vm) hold the two solutions.  0 = ok, 1 = a block was singular.
// ptk[0..2]: interiors; capacitance systems and Schur complements; separator recursion and back-substitution (device clock)
CFZP_FN int jstruct_solve(const CSpec &sp, const CDims &d, const CWork &w, const JWork &s, const Band &B, double *b1, double *b2, long long *ptk, double *lds) {
  double *flag = s.flag;
  long long tp = tick(), ts;
#define CFZJ_TICK(k) do { const long long t1_ = tick(); ptk[k] += t1_ - ts; ts = t1_; } while (0)  // ptk[3..10]: sub-phases
  const int Nm = s.Nmax, V = sp.V;
  if (flag[1] != 0.0) return 1;
  CFZP_LANE_FOR(one, 0, 0) flag[0] = 0.0;
  CFZP_SYNC();
  // ---- phase 1: interiors ---------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int f = jstruct_interiors_all(CFZS_WAVE, CFZS_NW, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_f64 *)B.ab, B.kb, B.ld, B.off, (const cfzb::glb_i32 *)s.cl,
                                        (const cfzb::glb_f64 *)b1, (const cfzb::glb_f64 *)b2, (cfzb::glb_f64 *)s.Cc, (cfzb::glb_f64 *)s.W, cfzb::opaque((cfzb::lds_f64 *)lds));
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
  // ---- phase 2a: C'W of every interior; the pair blocks of every interval index ------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  jstruct_cw_all(CFZS_WAVE, CFZS_NW, d.NI, (const cfzb::glb_f64 *)s.Cc, (const cfzb::glb_f64 *)s.W, (cfzb::glb_f64 *)s.CW);
#else
  for (int tt = 0; tt < d.NI * kJC * 31; ++tt) {
    const int it = tt / (kJC * 31), e = tt - it * (kJC * 31), al = e / 31, q = e - al * 31;
    const double *C = s.Cc + (size_t)it * kSI * kJC + al * kSI, *W = s.W + (size_t)it * kSI * kJR + q * kSI;
    double m_ = 0.0;
    for (int r = 0; r < kSI; ++r) m_ += C[r] * W[r];
    s.CW[(size_t)it * kJC * kJR + al * kJR + q] = m_;
  }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)  // (the device reads M's rows from the pair blocks themselves: jstruct_cap)
  jstruct_map(Nm * kJMt, [&](int tt) -> double {  // Mt[t][a][b][point][i][j]: owner computes (a diagonal block sums over the pairs of its vehicle)
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
    return v;
  }, [&](int tt, double v) { s.Mt[tt] = v; });
#endif
  CFZP_SYNC();
  CFZJ_TICK(3);
  // ---- phase 2b: the capacitance systems: (I + M G) Z = M E'K^-1 [C | b] --------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
  {
    const int f = jstruct_cap_all(CFZS_WAVE, CFZS_NW, Nm, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_f64 *)s.W, (const cfzb::glb_f64 *)w.pm, (cfzb::glb_f64 *)s.Z,
                                  cfzb::opaque((cfzb::lds_f64 *)lds), (cfzb::glb_f64 *)s.aug);
    if (f && CFZS_LANE == 0) flag[0] = 1.0;
  }
#else
  for (int t = 0; t < Nm; ++t) {
    double *Cap = s.aug + (size_t)kJB * (kJB + 64), *Rh = Cap + kJB * kJB;
    for (int col = 0; col < kJB; ++col)
      for (int row = 0; row < kJB; ++row) {
        const int a = row >> 4, i = row & 15, b = col >> 4, j = col & 15;
        double cap = row == col ? 1.0 : 0.0, rh = 0.0;
        if (a < V && t < sp.N[a] && i < 15) {
          const int kk = i / 3, ii = i - 3 * kk, pr = jprow(3 * kk);
          if (b < V && t < sp.N[b] && j < 15) {
            const double *M = s.Mt + (size_t)t * kJMt + ((a * kMaxVeh + b) * 5 + kk) * 9 + 3 * ii, *Wb = s.W + (size_t)(d.off[b] + t) * kSI * kJR;
            for (int c = 0; c < 3; ++c) cap += M[c] * Wb[(16 + j) * kSI + pr + c];
          }
          const int h = col >> 5, lc = col & 31;
          if (col >= kJYrhs) {
            if (col < kJYrhs + 2)
              for (int vb = 0; vb < V; ++vb) {
                if (t >= sp.N[vb]) continue;
                const double *M = s.Mt + (size_t)t * kJMt + ((a * kMaxVeh + vb) * 5 + kk) * 9 + 3 * ii, *Wb = s.W + (size_t)(d.off[vb] + t) * kSI * kJR;
                for (int c = 0; c < 3; ++c) rh += M[c] * Wb[(14 + col - kJYrhs) * kSI + pr + c];
              }
          } else if (lc < 2 * kJC) {
            const int vb = 2 * h + lc / kJC, q = lc % kJC;
            if (vb < V && t < sp.N[vb]) {
              const double *M = s.Mt + (size_t)t * kJMt + ((a * kMaxVeh + vb) * 5 + kk) * 9 + 3 * ii, *Wb = s.W + (size_t)(d.off[vb] + t) * kSI * kJR;
              for (int c = 0; c < 3; ++c) rh += M[c] * Wb[q * kSI + pr + c];
            }
          }
        }
        Cap[col * kJB + row] = cap; Rh[col * kJB + row] = rh;
      }
    if (jstruct_block_serial(s.aug, Cap, Rh, s.Z + (size_t)t * kJB * kJB, 64)) flag[0] = 1.0;
  }
#endif
  CFZP_SYNC();
  CFZJ_TICK(5);
  if (flag[0] != 0.0) return 1;
  // ---- phase 2d: the joint separator blocks, owner computes: band entries, the pair blocks of pt0, identity padding, minus the Schur
  // complements of the interiors of interval index i (rows / columns of pt0) and i - 1 (the right-coupled unknowns):
  //   S[(a, al), (b, be)] -= [a == b] C_a'W_a[al, be] - sum_j (C_a'K_a^-1 E)[al, j] Z[(a, j), (b, be)]
#if defined(__HIP_DEVICE_COMPILE__)
  jstruct_sep_base_all(CFZS_WAVE, CFZS_NW, Nm, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_f64 *)B.ab, B.kb, B.ld, B.off, (const cfzb::glb_f64 *)w.pm, (const cfzb::glb_f64 *)b1,
                       (const cfzb::glb_f64 *)b2, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us);
  __syncthreads();
  CFZJ_TICK(7);
  jstruct_schur_all(CFZS_WAVE, CFZS_NW, Nm, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_i32 *)s.cmask, (const cfzb::glb_f64 *)s.CW, (const cfzb::glb_f64 *)s.Z, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us);
#else
  auto schur = [&](int t, int a, int al, int b, int be) -> double {  // be >= 14: right-hand side be - 14
    const double *CW = s.CW + (size_t)(d.off[a] + t) * kJC * kJR + al * kJR;
    const double *Zc = s.Z + (size_t)t * kJB * kJB + (size_t)(be >= kJC ? kJYrhs + be - kJC : jycol(b, be)) * kJB + 16 * a;
    double m_ = be >= kJC ? CW[be] : (a == b ? CW[be] : 0.0);
#pragma unroll
    for (int j = 0; j < 15; ++j) m_ -= CW[16 + j] * Zc[j];
    return m_;
  };
  jstruct_map((Nm + 1) * kJB * kJB, [&](int tt) -> double {
    const int i = tt / (kJB * kJB), e = tt - i * (kJB * kJB), col = e / kJB, row = e - col * kJB, a = row >> 4, la = row & 15, b = col >> 4, lb = col & 15;
    double v = row == col ? 1.0 : 0.0;
    if (a < V && b < V && i <= sp.N[a] && i <= sp.N[b] && la < jsep_size(sp, a, i) && lb < jsep_size(sp, b, i)) {
      const int p0 = i == 0 ? 7 : 8;  // pt0's first local index
      const bool inter = i < sp.N[a] && i < sp.N[b];
      const bool pose = inter && la >= p0 && la < p0 + 3 && lb >= p0 && lb < p0 + 3;
      v = a == b ? band_at(B, d.nk, jsep_start(s, a, i) + la, jsep_start(s, a, i) + lb) : 0.0;
      if (pose)
        for (int pe = 0; pe < sp.n_pairs; ++pe) {
          const int pa = sp.pair_a[pe], pb = sp.pair_b[pe];
          if (i >= sp.N[pa] || i >= sp.N[pb]) continue;
          const double *pm = w.pm + (size_t)(d.poff[pe] + kPts * i) * 36;
          if (a == b) { if (pa == a) v += pm[6 * (la - p0) + (lb - p0)]; else if (pb == a) v += pm[6 * (3 + la - p0) + 3 + (lb - p0)]; }
          else if (pa == a && pb == b) v += pm[6 * (la - p0) + 3 + (lb - p0)];
          else if (pa == b && pb == a) v += pm[6 * (3 + la - p0) + (lb - p0)];
        }
      if (inter && la >= p0 && la < p0 + 7 && lb >= p0 && lb < p0 + 7) v -= schur(i, a, la - p0, b, lb - p0);
      if (i >= 1 && la < 7 && lb < 7 && s.cl[16 * (d.off[a] + i - 1) + 7 + la] >= 0 && s.cl[16 * (d.off[b] + i - 1) + 7 + lb] >= 0) v -= schur(i - 1, a, 7 + la, b, 7 + lb);
    }
    return v;
  }, [&](int tt, double v) { s.Ds[tt] = v; });
  // right-hand sides of the blocks: columns 0..27 = U_i, the coupling of block i's pt0 rows with block i + 1's rows 16 b + be; 28, 29 = b1, b2
  jstruct_map((Nm + 1) * kJB * 30, [&](int tt) -> double {
    const int i = tt / (kJB * 30), e = tt - i * (kJB * 30), col = e / kJB, row = e - col * kJB, a = row >> 4, la = row & 15;
    double v = 0.0;
    if (a >= V || i > sp.N[a] || la >= jsep_size(sp, a, i)) return v;
    const int p0 = i == 0 ? 7 : 8;
    const bool left = i < sp.N[a] && la >= p0 && la < p0 + 7;
    if (col < 28) {
      const int b = col / 7, be = col - 7 * b;
      if (left && b < V && i < sp.N[b] && s.cl[16 * (d.off[b] + i) + 7 + be] >= 0) v = -schur(i, a, la - p0, b, 7 + be);
      return v;
    }
    v = (col == 28 ? b1 : b2)[jsep_start(s, a, i) + la];
    if (left) v -= schur(i, a, la - p0, a, kJC + col - 28);
    if (i >= 1 && la < 7 && s.cl[16 * (d.off[a] + i - 1) + 7 + la] >= 0) v -= schur(i - 1, a, 7 + la, a, kJC + col - 28);
    return v;
  }, [&](int tt, double v) { const int i = tt / (kJB * 30), e = tt - i * (kJB * 30); s.Us[(size_t)i * kJB * kJU + e] = v; });
#endif
  CFZP_SYNC();
  CFZJ_TICK(8);
  { const long long t1 = tick(); ptk[1] += t1 - tp; tp = t1; }
  // ---- phase 3: recursion over the joint separators -------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(CFZ_JCHAIN)
  {  // block cyclic reduction (round 6): levels of stride 1, 2, 4, ..; per level the leaving blocks' eliminations on all eight wavefronts, then
     // their neighbours' updates; block 0 is what remains; the back-substitution runs through the levels in reverse
    const int wv = CFZS_WAVE, nwv = CFZS_NW;
    cfzb::lds_f64 *lb = cfzb::opaque((cfzb::lds_f64 *)lds);
    int top = 1;
    for (int st = 1; st <= Nm; st *= 2) {
      top = st;
      if (jbcr_lu_all(wv, nwv, st, Nm, (const cfzb::glb_f64 *)s.Ds, (const cfzb::glb_f64 *)s.Us, (cfzb::glb_f64 *)s.Zs, (cfzb::glb_f64 *)s.Zb, lb) && CFZS_LANE == 0) flag[0] = 1.0;
      __syncthreads();
      if (flag[0] != 0.0) return 1;
      jbcr_update_all(wv, nwv, st, Nm, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us, (const cfzb::glb_f64 *)s.Zs, (const cfzb::glb_f64 *)s.Zb, (cfzb::glb_f64 *)s.Lc);
      __syncthreads();
    }
    // block 0 remains: the same elimination with its two right-hand sides (the unit columns ride along unused); its solution is copied to xs
    // after a workgroup barrier (the elimination stores in tile layout: another lane's words)
    if (jbcr_lu_all(wv, nwv, 0, Nm, (const cfzb::glb_f64 *)s.Ds, (const cfzb::glb_f64 *)s.Us, (cfzb::glb_f64 *)s.Zs, (cfzb::glb_f64 *)s.Zb, lb) && CFZS_LANE == 0) flag[0] = 1.0;
    __syncthreads();
    if (flag[0] != 0.0) return 1;
    if (wv < 2) s.xs[wv * kJB + CFZS_LANE] = s.Zs[(28 + wv) * kJB + CFZS_LANE];
    __syncthreads();
    for (int st = top; st >= 1; st /= 2) {
      jbcr_back_all(wv, nwv, st, Nm, (const cfzb::glb_f64 *)s.Us, (const cfzb::glb_f64 *)s.Zs, (const cfzb::glb_f64 *)s.Zb, (const cfzb::glb_f64 *)s.Lc, (cfzb::glb_f64 *)s.xs);
      __syncthreads();
    }
  }
#else
  {  // from both ends, a wavefront each; the middle block receives both; back-substitution outwards
    const int mid = (Nm + 1) / 2, wv = CFZS_WAVE;
    unsigned nvp = 0;
    for (int a = 0; a < V; ++a) nvp |= (unsigned)sp.N[a] << (8 * a);
    int f = 0;
    // (LDS: a wavefront's own kLuLdsWave doubles for the elimination; X in the areas of two of the six idle wavefronts)
    static_assert(2 * kLuLdsWave >= kJB * kJXld, "X fits two areas");
    cfzb::lds_f64 *lb = cfzb::opaque((cfzb::lds_f64 *)lds);
    if (wv == 0) f = jstruct_chain(0, 0, mid, nvp, V, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us, (cfzb::glb_f64 *)s.Zs, lb, lb + 2 * kLuLdsWave);
    else if (wv == 1) f = jstruct_chain(1, Nm, mid, nvp, V, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us, (cfzb::glb_f64 *)s.Zs, lb + kLuLdsWave, lb + 4 * kLuLdsWave);
    if (f && CFZS_LANE == 0) flag[0] = 1.0;
    __syncthreads();
    if (flag[0] != 0.0) return 1;
    if (wv == 0 && jstruct_chain_mid(mid, (cfzb::glb_f64 *)s.Ds, (cfzb::glb_f64 *)s.Us, (cfzb::glb_f64 *)s.xs) && CFZS_LANE == 0) flag[0] = 1.0;
    __syncthreads();
    if (flag[0] != 0.0) return 1;
    if (wv < 2) jstruct_chain_back(wv, mid, Nm, (const cfzb::glb_f64 *)s.Zs, (cfzb::glb_f64 *)s.xs);
    __syncthreads();
  }
#endif
#else
  for (int i = 0; i <= Nm; ++i) {
    double *Di = s.Ds + (size_t)i * kJB * kJB, *Ui = s.Us + (size_t)i * kJB * kJU, *Zi = s.Zs + (size_t)i * kJB * kJU;
    if (jstruct_block_serial(s.aug, Di, Ui, Zi, kJU)) flag[0] = 1.0;
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
#endif
  CFZJ_TICK(9);
  // ---- phase 4: the separators' and the interiors' unknowns back to their positions --------------------------------------------------------
  CFZP_LANE_FOR(tt, 0, (Nm + 1) * kJB - 1) {
    const int i = tt / kJB, row = tt - i * kJB, a = row >> 4, la = row & 15;
    if (a < V && i <= sp.N[a] && la < jsep_size(sp, a, i)) { const int p = jsep_start(s, a, i) + la; b1[p] = s.xs[(size_t)i * 2 * kJB + row]; b2[p] = s.xs[(size_t)i * 2 * kJB + kJB + row]; }
  }
  CFZP_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
  jstruct_zt_all(CFZS_WAVE, CFZS_NW, Nm, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_i32 *)s.cl, (const cfzb::glb_f64 *)s.Z, (const cfzb::glb_f64 *)b1, (const cfzb::glb_f64 *)b2, (cfzb::glb_f64 *)s.zt);
  __syncthreads();
  jstruct_back_all(CFZS_WAVE, CFZS_NW, (const cfzb::glb_i32 *)s.meta, (const cfzb::glb_i32 *)s.cl, (const cfzb::glb_f64 *)s.W, (const cfzb::glb_f64 *)s.zt, (cfzb::glb_f64 *)b1, (cfzb::glb_f64 *)b2);
#else
  jstruct_map(Nm * 2 * kJB, [&](int tt) -> double {  // z = M y of every interval index: Z[:, b] - Z[:, coupling columns] s
    const int t = tt / (2 * kJB), e = tt - t * (2 * kJB), sr = e / kJB, row = e - sr * kJB;
    const double *Zt = s.Z + (size_t)t * kJB * kJB, *xb = sr ? b2 : b1;
    double z = Zt[(kJYrhs + sr) * kJB + row];
    for (int b = 0; b < V; ++b) {
      if (t >= sp.N[b]) continue;
      const int *clb = s.cl + 16 * (d.off[b] + t);
#pragma unroll
      for (int q = 0; q < kJC; ++q) { const int c = clb[q]; z -= Zt[jycol(b, q) * kJB + row] * (c >= 0 ? xb[c] : 0.0); }
    }
    return z;
  }, [&](int tt, double z) { s.zt[tt] = z; });
  CFZP_LANE_FOR(tt, 0, d.NI * kSI - 1) {
    const int it = tt / kSI, r = tt - it * kSI, a = veh_of_interval(d, it), t = it - d.off[a];
    const int *cl = s.cl + 16 * it;
    const double *W = s.W + (size_t)it * kSI * kJR + r, *z = s.zt + (size_t)t * 2 * kJB;
    double y1 = W[14 * kSI], y2 = W[15 * kSI];
    for (int q = 0; q < kJC; ++q) if (cl[q] >= 0) { y1 -= W[q * kSI] * b1[cl[q]]; y2 -= W[q * kSI] * b2[cl[q]]; }
    for (int j = 0; j < 15; ++j) { y1 -= W[(16 + j) * kSI] * z[16 * a + j]; y2 -= W[(16 + j) * kSI] * z[kJB + 16 * a + j]; }
    const int p = s.bs[a] + 79 * t + 14 + r;  // (interior positions are read by nobody in this phase)
    b1[p] = y1; b2[p] = y2;
  }
#endif
  CFZP_SYNC();
  CFZJ_TICK(10);
#undef CFZJ_TICK
  { const long long t1 = tick(); ptk[2] += t1 - tp; }
  return 0;
}

}  // namespace cfzc
