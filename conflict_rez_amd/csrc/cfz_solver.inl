// cfz_solver.inl -- one MPC-step NLP solved by one wavefront.
//
// Included by cfz_engine.hip (gfx950 device code: one 64-lane workgroup per problem instance,
// iterate and stage data in LDS) and by tests/emu/cfz_emu.cpp (same source, lanes run as a
// loop, so the kernel logic can be checked and sanitised on a CPU).  Not a CPU fallback: the
// product library only ever contains the device build.
//
// NLP: reference confrez/control/vehicle_follower.py:146-368 with the OBCA duals eliminated
// into closed-form separation certificates; dynamics confrez/control/dynamic_model.py:5-58;
// algorithm DESIGN.md "CFZ-IPM" (interior point, filter line search, slack elimination +
// Riccati recursion).  Work split inside the wavefront:
//   blocks    (stage k, obstacle/neighbour j) tasks strided over the 64 lanes
//   dynamics  RK4 + forward sensitivities, one stage per lane
//   assembly  condensed stage Hessian / gradient, one stage per lane
//   Riccati   backward / forward / costate sweeps on lane 0 (30 dependent steps)
//   step      slack and multiplier steps, fraction-to-boundary minima, one stage per lane
//
// Conventions: code inside CFZ_LANES(lane){...}CFZ_END runs once per lane and may only write
// lane-private locals or workspace cells it owns; everything outside runs uniformly (every
// lane computes the same value from the workspace).  Cross-lane sums/maxima go through the
// `red` slots of the workspace.

#ifndef CFZ_SOLVER_INL
#define CFZ_SOLVER_INL

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CFZ_FN __host__ __device__ __forceinline__
#define CFZ_CALL __host__ __device__ __forceinline__  // measured: out-of-line helpers cost 30 % (arguments spill to scratch)
#else
#define CFZ_FN static inline
#define CFZ_CALL static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define CFZ_LANES(lane) { const int lane = (int)threadIdx.x;
#define CFZ_END } __syncthreads();
#else
#define CFZ_LANES(lane) for (int lane = 0; lane < 64; ++lane) {
#define CFZ_END }
#endif

// Diagnostic build only (-DCFZ_STAMPS): shader-clock cycles per phase of the solver, summed over the
// iterations of one instance and written to a debug buffer (never to an output).
#if defined(CFZ_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define CFZ_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += t_ - stamp_last; stamp_last = __builtin_amdgcn_s_memtime(); } while (0)
#define CFZ_STAMP_DECL unsigned long long stamp_acc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; const unsigned long long stamp_wall0 = wall_clock64(); unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#else
#define CFZ_STAMP(i) do {} while (0)
#define CFZ_STAMP_DECL
#endif

namespace cfz {

constexpr int kNP = 7;      // x y psi v delta a w
constexpr int kRed = 6;     // reduction slots (0-3 own storage, 4-5 see make_layout)
constexpr int kMaxObs = 8;  // = CFZ_MAX_OBS

// Everything the kernel needs besides per-instance data (plain old data, passed by value).
struct KSpec {
  int N, n_obs, n_nbr, rk_substeps;
  int max_iter, max_backtrack, filter_cap, stall_iters;
  int row_curvature, pad1;
  double dt, wb, dmin;
  double g[4], bounds[12], weights[6];
  double A_obs[kMaxObs][4][2], b_obs[kMaxObs][4], V_obs[kMaxObs][4][2];
  double tol, constr_viol_tol, dual_inf_tol, compl_inf_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min,
      bound_push, bound_frac, s_max, kappa_sigma, eta_phi, gamma_theta, gamma_phi, delta_sw, s_theta, s_phi,
      reg_primal, stall_kappa, warm_push;
  // static obstacles as the kernel reads them, n_obs x 20 doubles in global memory: A[4][2], b[4], V[4][2]
  // (L1/L2-resident; indexing the arrays above with a lane-varying j would copy this struct to scratch)
  const double *obs_tab;
};

// Workspace layout (offsets in doubles) for one instance.
struct Lay {
  int N, nb, nr;                            // stages, blocks per stage, rows per stage (2 per block)
  int p, sg, nuc, zs, zl, zu, pi0, pi;      // iterate
  int dp, dsg, dpi0, dpi;                   // step
  int cj, ab, d, hc, gk, kk;                // stage data
  int sel, ref, nb4, x0, cs, rP, filt, red, red2, total;
};

CFZ_FN Lay make_layout(int N, int nb, int n_nbr) {
  (void)n_nbr;
  Lay L; int o = 0;
  const int nr = 2 * nb;
  L.N = N; L.nb = nb; L.nr = nr;
  L.p = o; o += N * kNP;
  L.sg = o; o += N * nr; L.nuc = o; o += N * nr; L.zs = o; o += N * nr;
  L.zl = o; o += N * 6; L.zu = o; o += N * 6;
  L.pi0 = o; o += 5; L.pi = o; o += N * 5;
  L.dp = o; o += N * kNP; L.dpi0 = o; o += 5;
  L.cj = o; o += N * nr; L.dsg = L.cj;  // the slack step overwrites the row residual it is computed from
  L.ab = o; o += N * 15; L.d = o; o += N * 5;
  L.dpi = L.d;  // the costate sweep (last reader of the defects d_k was the forward sweep) overwrites them with d(pi)
  L.hc = o; o += N * 11; L.gk = o; o += N * kNP; L.kk = o; o += N * 12;
  L.sel = o; o += (N * nb + 7) / 8;  // working set codes (< 192), one byte each
  L.ref = 0;  // the reference stays in global memory (read-only, L2-resident)
  L.nb4 = o; o += N * n_nbr * 4; L.x0 = o; o += 5;
  L.cs = o; o += (2 * N > 32) ? 2 * N : 32;  // cos, sin of the pose heading of every stage at the point being evaluated
  L.rP = L.cs;  // value function of stage 0 (30 numbers) between the Riccati sweeps, when cos/sin are not needed
  L.filt = o; o += 32;
  // reduction slots 0-3: no reduction runs between the assembly of H_k and the costate sweep, the only phases that
  // read hc, so the slots live there (the budget: four instances per CU need 20 LDS granules of 2 KiB = 5120 doubles)
  if (N * 11 >= 256) L.red = L.hc; else { L.red = o; o += 256; }
  // reduction slots 4 and 5 are only used by the residual pass at the top of an iteration, when the step of the
  // previous iteration is dead: they live in its storage (160 KiB of LDS hold three instances only if one
  // instance stays within 26 allocation granules of 2 KiB = 6656 doubles)
  if (N * kNP >= 128) L.red2 = L.dp; else { L.red2 = o; o += 128; }
  L.total = o;
  return L;
}

CFZ_FN unsigned char *sel_ptr(double *m, const Lay &L) { return reinterpret_cast<unsigned char *>(m + L.sel); }
CFZ_FN const unsigned char *sel_ptr(const double *m, const Lay &L) { return reinterpret_cast<const unsigned char *>(m + L.sel); }

// bounded columns of p: x y v delta a w  (psi is free)
CFZ_FN int bcol(int q) { return q < 2 ? q : q + 1; }

// ------------------------------------------------------------------------------ reductions
CFZ_FN int red_at(const Lay &L, int slot) { return slot < 4 ? L.red + slot * 64 : L.red2 + (slot - 4) * 64; }
CFZ_FN double red_sum(const double *m, const Lay &L, int slot) {
  const double *r = m + red_at(L, slot);
  double s = 0.0;
  for (int i = 0; i < 64; ++i) s += r[i];
  return s;
}
CFZ_FN double red_max(const double *m, const Lay &L, int slot) {
  const double *r = m + red_at(L, slot);
  double s = r[0];
  for (int i = 1; i < 64; ++i) s = fmax(s, r[i]);
  return s;
}
CFZ_FN double red_min(const double *m, const Lay &L, int slot) {
  const double *r = m + red_at(L, slot);
  double s = r[0];
  for (int i = 1; i < 64; ++i) s = fmin(s, r[i]);
  return s;
}

// ------------------------------------------------------------------------------ dynamics
// RK4 (M sub-steps) of the kinematic bicycle.  The state rows v, delta integrate exactly
// (v+ = v + a t, delta+ = delta + w t) and x, y never feed back, so only the sensitivities of
// (x, y, psi) with respect to (psi0, v0, delta0, a, w) are propagated: S[3][5].
// sin, cos of a small angle by Taylor series (|e| <= 0.06: truncation < 1e-21), sincos otherwise
CFZ_FN void small_sincos(double e, double *s, double *c) {
  if (fabs(e) > 0.06) { sincos(e, s, c); return; }
  const double e2 = e * e;
  *s = e * (1.0 - e2 / 6.0 * (1.0 - e2 / 20.0 * (1.0 - e2 / 42.0 * (1.0 - e2 / 72.0))));
  *c = 1.0 - e2 / 2.0 * (1.0 - e2 / 12.0 * (1.0 - e2 / 30.0 * (1.0 - e2 / 56.0 * (1.0 - e2 / 90.0))));
}

// Three library sincos calls per interval instead of 32: the steering angle advances by w h/2 between RK
// stage points (one fixed rotation), the heading by small increments (|h psi'| <= 0.03 rad inside the
// actuator limits), so the stage-point sines/cosines come from rotating the previous ones.
template <bool SENS>
CFZ_CALL void rk4_step(const double z[5], double a, double w, double dt, double wb, int M, double out[5],
                     double S[3][5]) {
  const double h = dt / M;
  double x = z[0], y = z[1], psi = z[2], v = z[3], de = z[4];
  if (SENS) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 5; ++c) S[r][c] = 0.0;
    S[2][0] = 1.0;
  }
  double sp_, cp_, sd0, cd0, sh, ch;  // heading and steering angle at the sub-step start, half-step steering rotation
  sincos(psi, &sp_, &cp_);
  sincos(de, &sd0, &cd0);
  small_sincos(0.5 * h * w, &sh, &ch);
  double tsub = 0.0;  // time since the start of the interval: dv/da = ddelta/dw = tsub
  for (int m = 0; m < M; ++m) {
    double ax = 0, ay = 0, ap = 0;  // weighted stage sums
    double AS[3][5];
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 5; ++c) AS[r][c] = 0.0;
    double kp = 0;  // previous stage derivative of psi
    double KS[3][5];
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 5; ++c) KS[r][c] = 0.0;
    // steering angle at the three distinct stage times of this sub-step: +0, +h/2, +h
    const double sd1 = sd0 * ch + cd0 * sh, cd1 = cd0 * ch - sd0 * sh;
    const double sd2 = sd1 * ch + cd1 * sh, cd2 = cd1 * ch - sd1 * sh;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const double wprev = (st == 0) ? 0.0 : ((st == 3) ? h : 0.5 * h);
      const double wsum = (st == 0 || st == 3) ? 1.0 : 2.0;
      const double vs = v + wprev * a;
      double s = sp_, c = cp_;
      if (st > 0) {
        double se, ce;
        small_sincos(wprev * kp, &se, &ce);
        s = sp_ * ce + cp_ * se; c = cp_ * ce - sp_ * se;
      }
      const double sd = (st == 0) ? sd0 : ((st == 3) ? sd2 : sd1), cd = (st == 0) ? cd0 : ((st == 3) ? cd2 : cd1);
      const double t = sd / cd;
      const double fx = vs * c, fy = vs * s, fp = vs / wb * t;
      if (SENS) {
        // stage point sensitivities: psi row from S/KS, v and delta rows analytic
        const double tau = tsub + wprev;
        const double j24 = vs / wb * (1.0 + t * t), j23 = t / wb;
        double NS[3][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          const double dps = S[2][q] + wprev * KS[2][q];
          const double dvs = (q == 1) ? 1.0 : ((q == 3) ? tau : 0.0);
          const double dds = (q == 2) ? 1.0 : ((q == 4) ? tau : 0.0);
          NS[0][q] = -vs * s * dps + c * dvs;
          NS[1][q] = vs * c * dps + s * dvs;
          NS[2][q] = j23 * dvs + j24 * dds;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int q = 0; q < 5; ++q) { KS[r][q] = NS[r][q]; AS[r][q] += wsum * NS[r][q]; }
      }
      kp = fp;
      ax += wsum * fx; ay += wsum * fy; ap += wsum * fp;
    }
    x += h / 6 * ax; y += h / 6 * ay; psi += h / 6 * ap;
    v += h * a; de += h * w;
    {
      double se, ce;
      small_sincos(h / 6 * ap, &se, &ce);
      const double sn = sp_ * ce + cp_ * se, cn = cp_ * ce - sp_ * se;
      sp_ = sn; cp_ = cn; sd0 = sd2; cd0 = cd2;
    }
    if (SENS)
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 5; ++q) S[r][q] += h / 6 * AS[r][q];
    tsub += h;
  }
  out[0] = x; out[1] = y; out[2] = psi; out[3] = v; out[4] = de;
}

// ------------------------------------------------------------------------------ separation certificates
// A block (stage k, obstacle or neighbour j) is certified along a face normal: "every vertex of
// one polygon lies at least dmin outside face f of the other".  kind 1 = polygon face / body
// vertices, kind 2 = body face / polygon vertices.  The working set of a block is the face and
// the two vertices whose rows are imposed: sel = kind*64 + face*16 + vA*4 + vB, vA < vB.
constexpr double kHyst = 1e-3;  // m: a block keeps its face until another is better by this much

template <bool GRAD>
CFZ_FN void vertex_dist(const double A[4][2], const double b[4], const double V[4][2], double x, double y,
                        double c, double s, const double g[4], int kind, int f, double d[4], double gr[4][3]) {
  if (kind == 1) {
    const double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
    double ax = 0.0, ay = 0.0, bf = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i == f) { ax = A[i][0]; ay = A[i][1]; bf = b[i]; }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double dwx = -s * BV[v][0] - c * BV[v][1], dwy = c * BV[v][0] - s * BV[v][1];
      d[v] = (x + dwy) * ax + (y - dwx) * ay - bf;
      if (GRAD) { gr[v][0] = ax; gr[v][1] = ay; gr[v][2] = ax * dwx + ay * dwy; }
    }
  } else {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    double gf = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i == f) gf = g[i];
    const double nx = c * gx - s * gy, ny = s * gx + c * gy, dnx = -s * gx - c * gy, dny = c * gx - s * gy;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      d[v] = (V[v][0] - x) * nx + (V[v][1] - y) * ny - gf;
      if (GRAD) { gr[v][0] = -nx; gr[v][1] = -ny; gr[v][2] = dnx * (V[v][0] - x) + dny * (V[v][1] - y); }
    }
  }
}

CFZ_FN double pick4(const double d[4], int v) {
  double r = d[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) if (i == v) r = d[i];
  return r;
}

CFZ_CALL int select_rows(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                       double s, const double g[4], int prev) {
  const int pk = prev >> 6, pf = (prev >> 4) & 3;
  double best = 0.0, prev_val = 0.0, d[4];
  int have = 0, bk = 0, bf = 0, have_prev = 0;
  for (int kind = 1; kind <= 2; ++kind)
    for (int f = 0; f < 4; ++f) {
      vertex_dist<false>(A, b, V, x, y, c, s, g, kind, f, d, nullptr);
      const double val = fmin(fmin(d[0], d[1]), fmin(d[2], d[3]));
      if (prev && kind == pk && f == pf) { prev_val = val; have_prev = 1; }
      if (!have || val > best) { have = 1; best = val; bk = kind; bf = f; }
    }
  if (have_prev && prev_val >= best - kHyst) { bk = pk; bf = pf; }
  vertex_dist<false>(A, b, V, x, y, c, s, g, bk, bf, d, nullptr);
  int v0 = 0;
#pragma unroll
  for (int v = 1; v < 4; ++v) if (d[v] < pick4(d, v0)) v0 = v;
  const int n1 = (v0 + 1) & 3, n2 = (v0 + 3) & 3;
  const double d0 = pick4(d, v0), dn1 = pick4(d, n1), dn2 = pick4(d, n2);
  int v1 = (dn1 < dn2) ? n1 : ((dn2 < dn1) ? n2 : (n1 < n2 ? n1 : n2));
  if (prev && bk == pk && bf == pf) {
    const int oa = (prev >> 2) & 3, ob = prev & 3;
    const double da = pick4(d, oa), db = pick4(d, ob);
    if (fmin(da, db) <= d0 + 1e-12 && fmax(da, db) <= pick4(d, v1) + kHyst) { v0 = oa; v1 = ob; }
  }
  const int va = v0 < v1 ? v0 : v1, vb = v0 < v1 ? v1 : v0;
  return bk * 64 + bf * 16 + va * 4 + vb;
}

// values (and gradients wrt x,y,psi) of the two rows of working set `sel`
template <bool GRAD>
CFZ_CALL void rows_for(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double c,
                     double s, const double g[4], int sel, double sep[2], double grad[2][3]) {
  double d[4], gr[4][3];
  vertex_dist<GRAD>(A, b, V, x, y, c, s, g, sel >> 6, (sel >> 4) & 3, d, gr);
  const int va = (sel >> 2) & 3, vb = sel & 3;
  sep[0] = pick4(d, va); sep[1] = pick4(d, vb);
  if (GRAD) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v == va) { grad[0][0] = gr[v][0]; grad[0][1] = gr[v][1]; grad[0][2] = gr[v][2]; }
      if (v == vb) { grad[1][0] = gr[v][0]; grad[1][1] = gr[v][1]; grad[1][2] = gr[v][2]; }
    }
  }
}

// polygon of block j at stage k: static obstacle from the spec, neighbour from its pose
CFZ_FN void block_polygon(const KSpec &sp, const double *m, const Lay &L, int k, int j, double A[4][2], double b[4],
                          double V[4][2]) {
  if (j < sp.n_obs) {
    const double *o = sp.obs_tab + j * 20;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A[i][0] = o[2 * i]; A[i][1] = o[2 * i + 1]; b[i] = o[8 + i];
      V[i][0] = o[12 + 2 * i]; V[i][1] = o[12 + 2 * i + 1];
    }
  } else {
    const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
    const double xo = q[0], yo = q[1], co = q[2], so = q[3];
    const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
    A[0][0] = co; A[0][1] = so; A[1][0] = -so; A[1][1] = co;
    A[2][0] = -co; A[2][1] = -so; A[3][0] = so; A[3][1] = -co;
    b[0] = A[0][0] * xo + A[0][1] * yo + g0; b[1] = A[1][0] * xo + A[1][1] * yo + g1;
    b[2] = A[2][0] * xo + A[2][1] * yo + g2; b[3] = A[3][0] * xo + A[3][1] * yo + g3;
    const double BV[4][2] = {{g0, g1}, {-g2, g1}, {-g2, -g3}, {g0, -g3}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[i][0] = xo + co * BV[i][0] - so * BV[i][1];
      V[i][1] = yo + so * BV[i][0] + co * BV[i][1];
    }
  }
}

// Gradients of the two rows of block j at stage k, rebuilt from the working-set code instead of being kept in LDS
// (same expressions as vertex_dist<true>): both rows share d/dx = a0, d/dy = a1 (same face); ap[r] = d/dpsi of row r.
CFZ_FN void block_grad(const KSpec &sp, const double *m, const Lay &L, int k, int j, int sl, double x, double y, double c,
                       double s, double &a0, double &a1, double ap[2]) {
  const int f = (sl >> 4) & 3;
  const double g0 = sp.g[0], g1 = sp.g[1], g2 = sp.g[2], g3 = sp.g[3];
  if ((sl >> 6) == 1) {
    double ax, ay;
    if (j < sp.n_obs) {
      const double *o = sp.obs_tab + j * 20;
      ax = o[2 * f]; ay = o[2 * f + 1];
    } else {
      const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
      const double co = q[2], so = q[3];
      ax = f == 0 ? co : (f == 1 ? -so : (f == 2 ? -co : so));
      ay = f == 0 ? so : (f == 1 ? co : (f == 2 ? -so : -co));
    }
    a0 = ax; a1 = ay;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
      const double dwx = -s * bx - c * by, dwy = c * bx - s * by;
      ap[r] = ax * dwx + ay * dwy;
    }
  } else {
    const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
    const double nx = c * gx - s * gy, ny = s * gx + c * gy, dnx = -s * gx - c * gy, dny = c * gx - s * gy;
    a0 = -nx; a1 = -ny;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int v = r == 0 ? ((sl >> 2) & 3) : (sl & 3);
      double vx, vy;
      if (j < sp.n_obs) {
        const double *o = sp.obs_tab + j * 20;
        vx = o[12 + 2 * v]; vy = o[13 + 2 * v];
      } else {
        const double *q = m + L.nb4 + (k * sp.n_nbr + (j - sp.n_obs)) * 4;
        const double xo = q[0], yo = q[1], co = q[2], so = q[3];
        const double bx = (v == 0 || v == 3) ? g0 : -g2, by = (v < 2) ? g1 : -g3;
        vx = xo + co * bx - so * by; vy = yo + so * bx + co * by;
      }
      ap[r] = dnx * (vx - x) + dny * (vy - y);
    }
  }
}

// ------------------------------------------------------------------------------ objective pieces
CFZ_FN double stage_cost(const KSpec &sp, const double *ref, int k, const double p[kNP]) {
  const double *w = sp.weights; const int N = sp.N;
  const double ex = p[0] - ref[k], ey = p[1] - ref[N + k], ep = p[2] - ref[2 * N + k];
  return w[0] * ex * ex + w[1] * ey * ey + w[2] * ep * ep + w[3] * p[5] * p[5] + w[4] * p[3] * p[3] * p[6] * p[6] +
         w[5] * p[4] * p[4];
}
CFZ_FN void stage_grad(const KSpec &sp, const double *ref, int k, const double p[kNP], double gr[kNP]) {
  const double *w = sp.weights; const int N = sp.N;
  gr[0] = 2 * w[0] * (p[0] - ref[k]);
  gr[1] = 2 * w[1] * (p[1] - ref[N + k]);
  gr[2] = 2 * w[2] * (p[2] - ref[2 * N + k]);
  gr[3] = 2 * w[4] * p[3] * p[6] * p[6];
  gr[4] = 2 * w[5] * p[4];
  gr[5] = 2 * w[3] * p[5];
  gr[6] = 2 * w[4] * p[3] * p[3] * p[6];
}

// dense views of the compact stage storage
CFZ_FN void load_AB(const double *m, const Lay &L, int k, double dt, double A[5][5], double B[5][2]) {
  const double *s = m + L.ab + k * 15;
  for (int i = 0; i < 5; ++i) {
    for (int q = 0; q < 5; ++q) A[i][q] = (i == q) ? 1.0 : 0.0;
    B[i][0] = 0.0; B[i][1] = 0.0;
  }
  for (int r = 0; r < 3; ++r) {
    A[r][2] = s[r * 5 + 0]; A[r][3] = s[r * 5 + 1]; A[r][4] = s[r * 5 + 2];
    B[r][0] = s[r * 5 + 3]; B[r][1] = s[r * 5 + 4];
  }
  B[3][0] = dt; B[4][1] = dt;
}
CFZ_FN void load_H(const double *m, const Lay &L, int k, double H[kNP][kNP]) {
  const double *h = m + L.hc + k * 11;
  for (int i = 0; i < kNP; ++i)
    for (int q = 0; q < kNP; ++q) H[i][q] = 0.0;
  for (int i = 0; i < kNP; ++i) H[i][i] = h[i];
  H[0][1] = H[1][0] = h[7]; H[0][2] = H[2][0] = h[8]; H[1][2] = H[2][1] = h[9]; H[3][6] = H[6][3] = h[10];
}
CFZ_FN void sym2_solve6(const double M[2][2], const double rhs[2][6], double out[2][6]) {
  const double l00 = sqrt(M[0][0]), l10 = M[1][0] / l00, l11 = sqrt(M[1][1] - l10 * l10);
  for (int q = 0; q < 6; ++q) {
    const double y0 = rhs[0][q] / l00, y1 = (rhs[1][q] - l10 * y0) / l11;
    const double x1 = y1 / l11;
    out[0][q] = (y0 - l10 * x1) / l00; out[1][q] = x1;
  }
}

// ------------------------------------------------------------------------------ trial-point evaluation
// theta = |c|_1 and barrier objective at (p + alpha dp, sg + alpha dsg).  Lanes write partials
// into red slots 0 (theta), 1 (phi without the log terms), 2 (sum of logs), 3 (1 if infeasible).
CFZ_CALL void merit_partials(const KSpec &sp, const double *refg, double *m, const Lay &L, double alpha, int lane) {
  const int N = sp.N, nb = L.nb;
  // sum of logs taken as the log of a per-lane product (<= 22 factors in [1e-10, 1e2]: no over/underflow)
  double th = 0.0, ph = 0.0, lprod = 1.0, bad = 0.0;
  for (int t = lane; t < N * nb; t += 64) {
    const int k = t / nb, j = t - k * nb;
    const double x = m[L.p + k * kNP + 0] + alpha * m[L.dp + k * kNP + 0];
    const double y = m[L.p + k * kNP + 1] + alpha * m[L.dp + k * kNP + 1];
    double A[4][2], b[4], V[4][2], sep[2];
    block_polygon(sp, m, L, k, j, A, b, V);
    rows_for<false>(A, b, V, x, y, m[L.cs + 2 * k], m[L.cs + 2 * k + 1], sp.g, sel_ptr(m, L)[t], sep, nullptr);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double sg = m[L.sg + 2 * t + r] + alpha * m[L.dsg + 2 * t + r];
      th += fabs(sep[r] - sp.dmin - sg);
      if (!(sg > 0.0)) bad = 1.0; else lprod *= sg;
    }
  }
  if (lane < N) {
    const int k = lane;
    double pt[kNP];
    for (int i = 0; i < kNP; ++i) pt[i] = m[L.p + k * kNP + i] + alpha * m[L.dp + k * kNP + i];
    for (int q = 0; q < 6; ++q) {
      const double dl = pt[bcol(q)] - sp.bounds[2 * q], du = sp.bounds[2 * q + 1] - pt[bcol(q)];
      if (!(dl > 0.0) || !(du > 0.0)) bad = 1.0; else lprod *= dl * du;
    }
    ph += stage_cost(sp, refg, k, pt);
    if (k == 0) for (int i = 0; i < 5; ++i) th += fabs(pt[i] - m[L.x0 + i]);
    if (k + 1 < N) {
      double F[5];
      rk4_step<false>(pt, pt[5], pt[6], sp.dt, sp.wb, sp.rk_substeps, F, nullptr);
      for (int i = 0; i < 5; ++i)
        th += fabs(F[i] - (m[L.p + (k + 1) * kNP + i] + alpha * m[L.dp + (k + 1) * kNP + i]));
    }
  }
  m[L.red + 0 * 64 + lane] = th; m[L.red + 1 * 64 + lane] = ph;
  m[L.red + 2 * 64 + lane] = (bad == 0.0) ? log(lprod) : 0.0; m[L.red + 3 * 64 + lane] = bad;
}

// ------------------------------------------------------------------------------ the solver
// x0[5], ref[3][N], nbr[n_nbr][3][N], zu[7][N] (warm start in, solution out) in global memory;
// m = this instance's workspace (LDS on the device).  out: iters,status ; cost,err,min_sep.
// dual_out (optional): l,m [N][4 n_obs], lam_ij, lam_ji [n_nbr][N][4], s [n_nbr][N][2].
struct DualOut { double *l, *mm, *lam_ij, *lam_ji, *s; unsigned long long *stamps; };

// State a converged solve leaves in global memory for the next MPC iteration of the same vehicle
// (oracle/mpc_nlp.py carry_state), in doubles: z[N][nr] | zl[N][6] | zu[N][6] | pi0[5] | pi[N][5] | mu | valid |
// working-set codes, one byte per block.
struct CarryLay { int z, zl, zu, pi0, pi, mu, valid, sel, stride; };
CFZ_FN CarryLay carry_layout(int N, int nb) {
  CarryLay c; int o = 0;
  c.z = o; o += N * 2 * nb; c.zl = o; o += N * 6; c.zu = o; o += N * 6; c.pi0 = o; o += 5; c.pi = o; o += N * 5;
  c.mu = o; o += 1; c.valid = o; o += 1; c.sel = o; o += (N * nb + 7) / 8;
  c.stride = (o + 7) & ~7;
  return c;
}

// wst: this instance's carry record (nullptr: none kept).  carry_in != 0: start from it if it is valid
// (oracle/mpc_nlp.py warm_from_carry).  A converged solve refreshes the record, any other outcome invalidates it.
CFZ_FN void solve_instance(const KSpec &sp, const double *x0g, const double *refg, const double *nbrg, double *zu,
                           double *m, const Lay &L, int *out_i, double *out_d, const DualOut &duo, double *wst = nullptr,
                           int carry_in = 0) {
  const int N = sp.N, nb = L.nb, nr = L.nr, n_obs = sp.n_obs, n_nbr = sp.n_nbr;
  const int m_eq = 5 + 5 * (N - 1) + nr * N, n_bnd = N * (12 + nr);
  const double mu_floor = fmin(sp.tol, sp.compl_inf_tol) / (sp.kappa_eps + 1.0);
  double stall_ref = 0.0;
  int stall_cnt = 0;

  // ---- load parameters, initial point ---------------------------------------------------
  CFZ_LANES(lane)
    for (int t = lane; t < N * n_nbr; t += 64) {
      const int k = t / n_nbr, o = t - k * n_nbr;
      const double po = nbrg[(o * 3 + 2) * N + k];
      double *q = m + L.nb4 + t * 4;
      q[0] = nbrg[(o * 3 + 0) * N + k]; q[1] = nbrg[(o * 3 + 1) * N + k]; q[2] = cos(po); q[3] = sin(po);
    }
    if (lane < 5) { m[L.x0 + lane] = x0g[lane]; m[L.pi0 + lane] = 0.0; }
    for (int i = lane; i < N * kNP; i += 64) { const int k = i / kNP, c = i - k * kNP; m[L.p + i] = zu[c * N + k]; }
    for (int i = lane; i < N * 5; i += 64) m[L.pi + i] = 0.0;
  CFZ_END
  // The pose of stage 0 is pinned to the measured state: a collision row violated there by more
  // than 2*constr_viol_tol cannot be repaired (status 4; reference: IPOPT fails, step() falls back).
  CFZ_LANES(lane)
    double worst = INFINITY;
    if (lane < nb) {
      double A[4][2], b[4], V[4][2], sep[2];
      block_polygon(sp, m, L, 0, lane, A, b, V);
      double s0, c0_;
      sincos(m[L.x0 + 2], &s0, &c0_);
      const int c0 = select_rows(A, b, V, m[L.x0], m[L.x0 + 1], c0_, s0, sp.g, 0);
      rows_for<false>(A, b, V, m[L.x0], m[L.x0 + 1], c0_, s0, sp.g, c0, sep, nullptr);
      worst = fmin(sep[0], sep[1]);
    }
    m[L.red + lane] = worst;
  CFZ_END
  if (red_min(m, L, 0) < sp.dmin - 2.0 * sp.constr_viol_tol) {
    out_i[0] = 0; out_i[1] = 4; out_d[0] = 0.0; out_d[1] = INFINITY; out_d[2] = red_min(m, L, 0);
    if (wst) { CFZ_LANES(lane) if (lane == 0) wst[carry_layout(N, nb).valid] = 0.0; CFZ_END }
    return;
  }
  const CarryLay CL = carry_layout(N, nb);
  const bool warm = wst != nullptr && carry_in != 0 && wst[CL.valid] != 0.0;
  const double mu0 = warm ? fmin(fmax(wst[CL.mu], mu_floor), sp.mu_init) : sp.mu_init;
  CFZ_LANES(lane)
    // working set and slacks from the un-pushed warm start (IPOPT: s = g(x0)), then pushed inside
    for (int t = lane; t < N * nb; t += 64) {
      const int k = t / nb, j = t - k * nb;
      double A[4][2], b[4], V[4][2], sep[2];
      block_polygon(sp, m, L, k, j, A, b, V);
      const double x = m[L.p + k * kNP], y = m[L.p + k * kNP + 1];
      double sn, cn;
      sincos(m[L.p + k * kNP + 2], &sn, &cn);
      const int c0 = select_rows(A, b, V, x, y, cn, sn, sp.g, 0);
      sel_ptr(m, L)[t] = c0;
      rows_for<false>(A, b, V, x, y, cn, sn, sp.g, c0, sep, nullptr);
      if (!warm) {
        for (int r = 0; r < 2; ++r) {
          m[L.sg + 2 * t + r] = fmax(sep[r] - sp.dmin, sp.bound_push);
          m[L.zs + 2 * t + r] = 1.0; m[L.nuc + 2 * t + r] = 0.0;
        }
      } else {
        // a row whose (face, vertex) identity exists in the carried working set of stage k+1 keeps its multiplier
        const int ko = k + 1 < N ? k + 1 : N - 1;
        const int so = reinterpret_cast<const unsigned char *>(wst + CL.sel)[ko * nb + j];
        for (int r = 0; r < 2; ++r) {
          const int vn = r == 0 ? ((c0 >> 2) & 3) : (c0 & 3);
          double z = 0.0;
          if ((so >> 4) == (c0 >> 4)) {
            if (((so >> 2) & 3) == vn) z = wst[CL.z + (ko * nb + j) * 2];
            else if ((so & 3) == vn) z = wst[CL.z + (ko * nb + j) * 2 + 1];
          }
          const double gap = sep[r] - sp.dmin;
          double sg;
          if (z > 0.0) sg = fmax(fmax(gap, mu0 / z), sp.warm_push);
          else { sg = fmax(gap, sp.bound_push); z = mu0 / sg; }
          m[L.sg + 2 * t + r] = sg; m[L.zs + 2 * t + r] = z; m[L.nuc + 2 * t + r] = -z;
        }
      }
    }
  CFZ_END
  CFZ_LANES(lane)
    if (lane < N) {
      const int ko = lane + 1 < N ? lane + 1 : N - 1;
      for (int q = 0; q < 6; ++q) {
        const double lo = sp.bounds[2 * q], hi = sp.bounds[2 * q + 1];
        double v = m[L.p + lane * kNP + bcol(q)];
        if (!warm) {
          const double pl = fmin(sp.bound_push * fmax(1.0, fabs(lo)), sp.bound_frac * (hi - lo));
          const double pu = fmin(sp.bound_push * fmax(1.0, fabs(hi)), sp.bound_frac * (hi - lo));
          v = fmax(v, lo + pl); v = fmin(v, hi - pu);
          m[L.zl + lane * 6 + q] = 1.0; m[L.zu + lane * 6 + q] = 1.0;
        } else {
          v = fmin(fmax(v, lo + sp.warm_push), hi - sp.warm_push);
          m[L.zl + lane * 6 + q] = fmax(wst[CL.zl + ko * 6 + q], mu0 / (hi - lo));
          m[L.zu + lane * 6 + q] = fmax(wst[CL.zu + ko * 6 + q], mu0 / (hi - lo));
        }
        m[L.p + lane * kNP + bcol(q)] = v;
      }
      if (warm && lane + 1 < N) {
        const int kp = lane + 1 < N - 1 ? lane + 1 : N - 2;
        for (int i = 0; i < 5; ++i) m[L.pi + lane * 5 + i] = wst[CL.pi + kp * 5 + i];
      }
    }
    if (warm && lane < 5) m[L.pi0 + lane] = wst[CL.pi + lane];
  CFZ_END

  double mu = mu0, filt_mu = -1.0, theta_min = -1.0, theta_max = -1.0, err0 = INFINITY, fval_last = 0.0;
  CFZ_STAMP_DECL
  CFZ_STAMP(0);  // setup
  int nfilt = 0, status = 1, iter = 0;

  for (iter = 0; iter <= sp.max_iter; ++iter) {
    // ---- working set refresh (iter > 0), rows and dynamics at the current point --------------
    CFZ_LANES(lane)
      if (lane < N) sincos(m[L.p + lane * kNP + 2], &m[L.cs + 2 * lane + 1], &m[L.cs + 2 * lane]);
    CFZ_END
    CFZ_LANES(lane)
      double cmax = 0.0, csum = 0.0;
      for (int t = lane; t < N * nb; t += 64) {
        const int k = t / nb, j = t - k * nb;
        double A[4][2], b[4], V[4][2], sep[2], gr[2][3];
        block_polygon(sp, m, L, k, j, A, b, V);
        const double x = m[L.p + k * kNP], y = m[L.p + k * kNP + 1], cn = m[L.cs + 2 * k], sn = m[L.cs + 2 * k + 1];
        int c1 = sel_ptr(m, L)[t];
        if (iter > 0) {
          const int c0 = c1;
          c1 = select_rows(A, b, V, x, y, cn, sn, sp.g, c0);
          if (c1 != c0) sel_ptr(m, L)[t] = c1;
          rows_for<true>(A, b, V, x, y, cn, sn, sp.g, c1, sep, gr);
          if (c1 != c0) {
            // a row that keeps its (face, vertex) identity keeps slack and multipliers; a new row
            // starts at sigma = max(sep - dmin, bound_push), z = mu / sigma, nu = -z
            const int same_face = (c0 >> 4) == (c1 >> 4);
            const int ov0 = (c0 >> 2) & 3, ov1 = c0 & 3;
            const double o_sg[2] = {m[L.sg + 2 * t], m[L.sg + 2 * t + 1]};
            const double o_zs[2] = {m[L.zs + 2 * t], m[L.zs + 2 * t + 1]};
            const double o_nu[2] = {m[L.nuc + 2 * t], m[L.nuc + 2 * t + 1]};
            for (int r = 0; r < 2; ++r) {
              const int nv = r == 0 ? ((c1 >> 2) & 3) : (c1 & 3);
              const int src = same_face ? (nv == ov0 ? 0 : (nv == ov1 ? 1 : -1)) : -1;
              if (src >= 0) {
                m[L.sg + 2 * t + r] = src == 0 ? o_sg[0] : o_sg[1];
                m[L.zs + 2 * t + r] = src == 0 ? o_zs[0] : o_zs[1];
                m[L.nuc + 2 * t + r] = src == 0 ? o_nu[0] : o_nu[1];
              } else {
                const double sg = fmax(sep[r] - sp.dmin, sp.bound_push);
                m[L.sg + 2 * t + r] = sg; m[L.zs + 2 * t + r] = mu / sg; m[L.nuc + 2 * t + r] = -mu / sg;
              }
            }
          }
        } else {
          rows_for<true>(A, b, V, x, y, cn, sn, sp.g, c1, sep, gr);
        }
        for (int r = 0; r < 2; ++r) {
          const double c = sep[r] - sp.dmin - m[L.sg + 2 * t + r];
          m[L.cj + 2 * t + r] = c;
          cmax = fmax(cmax, fabs(c)); csum += fabs(c);
        }
      }
      m[L.red + 2 * 64 + lane] = cmax; m[L.red + 3 * 64 + lane] = csum;
    CFZ_END
    CFZ_STAMP(9);  // working set + rows
    CFZ_LANES(lane)
      double cmax = m[L.red + 2 * 64 + lane], csum = m[L.red + 3 * 64 + lane];
      if (lane == 0) for (int i = 0; i < 5; ++i) { const double r = m[L.p + i] - m[L.x0 + i]; cmax = fmax(cmax, fabs(r)); csum += fabs(r); }
      if (lane + 1 < N) {
        const int k = lane;
        double F[5], S[3][5];
        const double *pk = m + L.p + k * kNP;
        rk4_step<true>(pk, pk[5], pk[6], sp.dt, sp.wb, sp.rk_substeps, F, S);
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 5; ++q) m[L.ab + k * 15 + r * 5 + q] = S[r][q];
        for (int i = 0; i < 5; ++i) {
          const double d = F[i] - m[L.p + (k + 1) * kNP + i];
          m[L.d + k * 5 + i] = d; cmax = fmax(cmax, fabs(d)); csum += fabs(d);
        }
      }
      m[L.red + 0 * 64 + lane] = cmax; m[L.red + 1 * 64 + lane] = csum;
    CFZ_END
    const double cviol = red_max(m, L, 0), theta = red_sum(m, L, 1);
    CFZ_STAMP(1);  // working set, rows, dynamics
    if (theta_min < 0.0) { theta_min = 1e-4 * fmax(1.0, theta); theta_max = 1e4 * fmax(1.0, theta); }
    // ---- dual infeasibility, multiplier sums, complementarity, objective, log terms ----------
    CFZ_LANES(lane)
      double dinf = 0.0, snu = 0.0, sz = 0.0, c0 = 0.0, fv = 0.0, lprod = 1.0;
      if (lane == 0) for (int i = 0; i < 5; ++i) snu += fabs(m[L.pi0 + i]);
      if (lane < N) {
        const int k = lane;
        const double *pk = m + L.p + k * kNP;
        double r[kNP];
        stage_grad(sp, refg, k, pk, r);
        fv = stage_cost(sp, refg, k, pk);
        const double cpsi = m[L.cs + 2 * k], spsi = m[L.cs + 2 * k + 1];
        for (int jb = 0; jb < nb; ++jb) {
          double a0, a1, ap[2];
          block_grad(sp, m, L, k, jb, sel_ptr(m, L)[k * nb + jb], pk[0], pk[1], cpsi, spsi, a0, a1, ap);
          for (int r_ = 0; r_ < 2; ++r_) {
            const int t = k * nr + 2 * jb + r_;
            const double nu = m[L.nuc + t], zs = m[L.zs + t], sg = m[L.sg + t];
            r[0] += a0 * nu; r[1] += a1 * nu; r[2] += ap[r_] * nu;
            dinf = fmax(dinf, fabs(-nu - zs));
            snu += fabs(nu); sz += zs; c0 = fmax(c0, fabs(sg * zs)); lprod *= sg;
          }
        }
        if (k + 1 < N) {
          double A[5][5], B[5][2];
          load_AB(m, L, k, sp.dt, A, B);
          for (int i = 0; i < 5; ++i) {
            const double pi = m[L.pi + k * 5 + i];
            snu += fabs(pi);
            for (int q = 0; q < 5; ++q) r[q] += A[i][q] * pi;
            r[5] += B[i][0] * pi; r[6] += B[i][1] * pi;
          }
        }
        if (k == 0) for (int i = 0; i < 5; ++i) r[i] += m[L.pi0 + i];
        else for (int i = 0; i < 5; ++i) r[i] -= m[L.pi + (k - 1) * 5 + i];
        for (int q = 0; q < 6; ++q) {
          const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
          r[bcol(q)] += -zl + zu_; sz += zl + zu_;
          const double dl = pk[bcol(q)] - sp.bounds[2 * q], du = sp.bounds[2 * q + 1] - pk[bcol(q)];
          c0 = fmax(c0, fmax(fabs(dl * zl), fabs(du * zu_)));
          lprod *= dl * du;
        }
        for (int i = 0; i < kNP; ++i) dinf = fmax(dinf, fabs(r[i]));
      }
      m[L.red + 0 * 64 + lane] = dinf; m[L.red + 1 * 64 + lane] = snu; m[L.red + 2 * 64 + lane] = sz;
      m[L.red + 3 * 64 + lane] = c0; m[red_at(L, 4) + lane] = fv; m[red_at(L, 5) + lane] = log(lprod);
    CFZ_END
    const double dual_inf = red_max(m, L, 0), sum_nu = red_sum(m, L, 1), sum_z = red_sum(m, L, 2);
    const double cmp0 = red_max(m, L, 3), fval = red_sum(m, L, 4), logsum = red_sum(m, L, 5);
    const double s_d = fmax(sp.s_max, (sum_nu + sum_z) / (double)(m_eq + n_bnd)) / sp.s_max;
    const double s_c = fmax(sp.s_max, sum_z / (double)n_bnd) / sp.s_max;
    err0 = fmax(dual_inf / s_d, fmax(cviol, cmp0 / s_c));
    CFZ_STAMP(2);  // residuals
    fval_last = fval;
    if (!isfinite(err0)) { status = 3; break; }
    if (err0 <= sp.tol && dual_inf <= sp.dual_inf_tol && cviol <= sp.constr_viol_tol && cmp0 <= sp.compl_inf_tol) { status = 0; break; }
    if (iter == sp.max_iter) { status = 1; break; }
    // infeasibility stall (oracle/ipm.py): violation stuck above the tolerance -> locally infeasible, status 5
    if (iter == 0 || cviol <= sp.stall_kappa * stall_ref) { stall_ref = cviol; stall_cnt = 0; } else ++stall_cnt;
    if (sp.stall_iters > 0 && stall_cnt >= sp.stall_iters && cviol > sp.constr_viol_tol) { status = 5; break; }
    // ---- barrier update (monotone, Fiacco-McCormick) ------------------------------------------
    while (mu > mu_floor) {
      CFZ_LANES(lane)
        double cm = 0.0;
        if (lane < N) {
          const int k = lane;
          for (int j = 0; j < nr; ++j) cm = fmax(cm, fabs(m[L.sg + k * nr + j] * m[L.zs + k * nr + j] - mu));
          for (int q = 0; q < 6; ++q) {
            const double v = m[L.p + k * kNP + bcol(q)];
            cm = fmax(cm, fmax(fabs((v - sp.bounds[2 * q]) * m[L.zl + k * 6 + q] - mu),
                               fabs((sp.bounds[2 * q + 1] - v) * m[L.zu + k * 6 + q] - mu)));
          }
        }
        m[L.red + 0 * 64 + lane] = cm;
      CFZ_END
      const double emu = fmax(dual_inf / s_d, fmax(cviol, red_max(m, L, 0) / s_c));
      if (emu <= sp.kappa_eps * mu) mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu)));
      else break;
    }
    const double tau = fmax(sp.tau_min, 1.0 - mu);
    CFZ_STAMP(3);  // barrier update
    // ---- condensed stage QP: H_k (compact), g_k ---------------------------------------------------
    CFZ_LANES(lane)
      if (lane < N) {
        const int k = lane;
        const double *w = sp.weights; const double *pk = m + L.p + k * kNP;
        double g[kNP], h[11];
        stage_grad(sp, refg, k, pk, g);
        h[0] = 2 * w[0]; h[1] = 2 * w[1]; h[2] = 2 * w[2]; h[3] = 2 * w[4] * pk[6] * pk[6]; h[4] = 2 * w[5];
        h[5] = 2 * w[3]; h[6] = 2 * w[4] * pk[3] * pk[3]; h[7] = 0.0; h[8] = 0.0; h[9] = 0.0;
        h[10] = 2 * w[4] * pk[3] * pk[6];
        for (int i = 0; i < kNP; ++i) h[i] += sp.reg_primal;
        for (int q = 0; q < 6; ++q) {
          const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
          h[bcol(q)] += m[L.zl + k * 6 + q] * il + m[L.zu + k * 6 + q] * iu;
          g[bcol(q)] += mu * (iu - il);
        }
        // curvature of the separation rows weighted with their multipliers, sum_r nu_r d2 sep_r / d(x,y,psi)^2 =
        // [[0,0,ca],[0,0,cb],[ca,cb,cc]] (oracle/mpc_nlp.py row_curvature), rebuilt from the row gradients:
        //   kind 1 (polygon face A_f = (a0,a1), body vertex b_v): d2/dpsi2 = -A_f.(R b_v)
        //   kind 2 (body face normal n = -(a0,a1), polygon vertex): d2/dx dpsi = -a1, d2/dy dpsi = a0, d2/dpsi2 = -(sep + g_f)
        double ca = 0.0, cb = 0.0, cc = 0.0;
        const double cpsi = m[L.cs + 2 * k], spsi = m[L.cs + 2 * k + 1];
        double ba0 = 0.0, ba1 = 0.0, bap[2] = {0.0, 0.0};
        for (int j = 0; j < nr; ++j) {
          const int t = k * nr + j;
          const double isg = 1.0 / m[L.sg + t], S = m[L.zs + t] * isg + sp.reg_primal;
          const double coef = S * m[L.cj + t] - mu * isg;
          const int sl = sel_ptr(m, L)[t >> 1];
          if ((j & 1) == 0) block_grad(sp, m, L, k, j >> 1, sl, pk[0], pk[1], cpsi, spsi, ba0, ba1, bap);
          const double a0 = ba0, a1 = ba1, a2 = bap[j & 1];
          g[0] += a0 * coef; g[1] += a1 * coef; g[2] += a2 * coef;
          h[0] += S * a0 * a0; h[1] += S * a1 * a1; h[2] += S * a2 * a2;
          h[7] += S * a0 * a1; h[8] += S * a0 * a2; h[9] += S * a1 * a2;
          if (sp.row_curvature) {
            const int f = (sl >> 4) & 3, v = (t & 1) ? (sl & 3) : ((sl >> 2) & 3);
            const double nu = m[L.nuc + t];
            if ((sl >> 6) == 1) {
              const double bx = (v == 0 || v == 3) ? sp.g[0] : -sp.g[2], by = (v < 2) ? sp.g[1] : -sp.g[3];
              cc -= nu * (a0 * (cpsi * bx - spsi * by) + a1 * (spsi * bx + cpsi * by));
            } else {
              const double gf = f == 0 ? sp.g[0] : (f == 1 ? sp.g[1] : (f == 2 ? sp.g[2] : sp.g[3]));
              ca -= nu * a1; cb += nu * a0;
              cc -= nu * (m[L.cj + t] + sp.dmin + m[L.sg + t] + gf);
            }
          }
        }
        if (sp.row_curvature) {
          // convexity safeguard: scale by th in {1, 1/2, .., 2^-9, 0} until diag(2w) + th C keeps the margin 0.2 min(w)
          const double mg = 0.2 * fmin(w[0], fmin(w[1], w[2]));
          const double q0 = 2 * w[0] - mg, q1 = 2 * w[1] - mg, q2 = 2 * w[2] - mg;
          const double quad = ca * ca / q0 + cb * cb / q1;
          double th = 1.0;
          for (int hh = 0; hh < 11; ++hh) {
            if (hh == 10) { th = 0.0; break; }
            if (q2 + th * cc - th * th * quad >= 0.0) break;
            th *= 0.5;
          }
          h[2] += th * cc; h[8] += th * ca; h[9] += th * cb;
        }
        for (int i = 0; i < 11; ++i) m[L.hc + k * 11 + i] = h[i];
        for (int i = 0; i < kNP; ++i) m[L.gk + k * kNP + i] = g[i];
      }
    CFZ_END
    CFZ_STAMP(4);  // assembly
    // ---- Riccati backward sweep (lane 0; the 5x5 value function stays in registers) -------------------
    // Structure used: A_k = I + [0 0 s00 s01 s02; 0 0 s10 s11 s12; 0 0 0 s21 s22; 0; 0] (s20 = 1),
    // B_k = [s03 s04; s13 s14; s23 s24; dt 0; 0 dt]; H_k = diag(h0..h6) + pose off-diagonals
    // h7 (0,1), h8 (0,2), h9 (1,2) + the v-w cross term h10 (3,6).  A lane-parallel variant (matrix
    // entries spread over lanes, exchange through LDS) measured 2.2x slower: every exchange is a
    // dependent LDS round trip of a lone wavefront.
    double *const rP = m + L.rP, *const rp = rP + 25;  // value function of stage 0 handed to the forward sweep
    CFZ_LANES(lane)
      if (lane == 0) {
        const int k = N - 1;  // terminal stage: its inputs a,w are costed but drive no dynamics
        const double *h = m + L.hc + k * 11, *gk = m + L.gk + k * kNP;
        double *K = m + L.kk + k * 12;
        for (int q = 0; q < 10; ++q) K[q] = 0.0;
        K[5 + 3] = -h[10] / h[6]; K[10] = -gk[5] / h[5]; K[11] = -gk[6] / h[6];
        for (int i = 0; i < 5; ++i) { for (int q = 0; q < 5; ++q) rP[i * 5 + q] = 0.0; rP[i * 6] = h[i]; rp[i] = gk[i]; }
        rP[1] = rP[5] = h[7]; rP[2] = rP[10] = h[8]; rP[7] = rP[11] = h[9];
        rP[18] += h[10] * K[5 + 3]; rp[3] += h[10] * K[11];
      }
    CFZ_END
    CFZ_LANES(lane)
      if (lane == 0) {  // backward sweep on one lane, P in registers
        const double dt = sp.dt;
        double P[5][5], pv[5];
        for (int i = 0; i < 5; ++i) { for (int q = 0; q < 5; ++q) P[i][q] = rP[i * 5 + q]; pv[i] = rp[i]; }
        for (int k = N - 2; k >= 0; --k) {
          const double *s = m + L.ab + k * 15, *h = m + L.hc + k * 11, *gk = m + L.gk + k * kNP, *dk = m + L.d + k * 5;
          const double s00 = s[0], s01 = s[1], s02 = s[2], s03 = s[3], s04 = s[4];
          const double s10 = s[5], s11 = s[6], s12 = s[7], s13 = s[8], s14 = s[9];
          const double s21 = s[11], s22 = s[12], s23 = s[13], s24 = s[14];
          double M[5][5], PB[5][2], Pd[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            M[i][0] = P[i][0]; M[i][1] = P[i][1];
            M[i][2] = P[i][2] + s00 * P[i][0] + s10 * P[i][1];
            M[i][3] = P[i][3] + s01 * P[i][0] + s11 * P[i][1] + s21 * P[i][2];
            M[i][4] = P[i][4] + s02 * P[i][0] + s12 * P[i][1] + s22 * P[i][2];
            PB[i][0] = s03 * P[i][0] + s13 * P[i][1] + s23 * P[i][2] + dt * P[i][3];
            PB[i][1] = s04 * P[i][0] + s14 * P[i][1] + s24 * P[i][2] + dt * P[i][4];
            Pd[i] = pv[i] + P[i][0] * dk[0] + P[i][1] * dk[1] + P[i][2] * dk[2] + P[i][3] * dk[3] + P[i][4] * dk[4];
          }
          double Hxx[5][5], hx[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            Hxx[0][j] = M[0][j]; Hxx[1][j] = M[1][j];
            Hxx[2][j] = M[2][j] + s00 * M[0][j] + s10 * M[1][j];
            Hxx[3][j] = M[3][j] + s01 * M[0][j] + s11 * M[1][j] + s21 * M[2][j];
            Hxx[4][j] = M[4][j] + s02 * M[0][j] + s12 * M[1][j] + s22 * M[2][j];
          }
          Hxx[0][0] += h[0]; Hxx[1][1] += h[1]; Hxx[2][2] += h[2]; Hxx[3][3] += h[3]; Hxx[4][4] += h[4];
          Hxx[0][1] += h[7]; Hxx[1][0] += h[7]; Hxx[0][2] += h[8]; Hxx[2][0] += h[8]; Hxx[1][2] += h[9]; Hxx[2][1] += h[9];
          hx[0] = gk[0] + Pd[0]; hx[1] = gk[1] + Pd[1];
          hx[2] = gk[2] + Pd[2] + s00 * Pd[0] + s10 * Pd[1];
          hx[3] = gk[3] + Pd[3] + s01 * Pd[0] + s11 * Pd[1] + s21 * Pd[2];
          hx[4] = gk[4] + Pd[4] + s02 * Pd[0] + s12 * Pd[1] + s22 * Pd[2];
          double Hux[2][5], hu[2];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            Hux[0][j] = s03 * M[0][j] + s13 * M[1][j] + s23 * M[2][j] + dt * M[3][j];
            Hux[1][j] = s04 * M[0][j] + s14 * M[1][j] + s24 * M[2][j] + dt * M[4][j];
          }
          Hux[1][3] += h[10];
          const double a00 = h[5] + s03 * PB[0][0] + s13 * PB[1][0] + s23 * PB[2][0] + dt * PB[3][0];
          const double a01 = s03 * PB[0][1] + s13 * PB[1][1] + s23 * PB[2][1] + dt * PB[3][1];
          const double a11 = h[6] + s04 * PB[0][1] + s14 * PB[1][1] + s24 * PB[2][1] + dt * PB[4][1];
          hu[0] = gk[5] + s03 * Pd[0] + s13 * Pd[1] + s23 * Pd[2] + dt * Pd[3];
          hu[1] = gk[6] + s04 * Pd[0] + s14 * Pd[1] + s24 * Pd[2] + dt * Pd[4];
          const double idet = 1.0 / (a00 * a11 - a01 * a01);
          const double i00 = a11 * idet, i01 = -a01 * idet, i11 = a00 * idet;
          double t0[6], t1[6];  // Huu^{-1} [Hux hu]
#pragma unroll
          for (int q = 0; q < 5; ++q) { t0[q] = i00 * Hux[0][q] + i01 * Hux[1][q]; t1[q] = i01 * Hux[0][q] + i11 * Hux[1][q]; }
          t0[5] = i00 * hu[0] + i01 * hu[1]; t1[5] = i01 * hu[0] + i11 * hu[1];
          double *K = m + L.kk + k * 12;
          for (int q = 0; q < 5; ++q) { K[q] = -t0[q]; K[5 + q] = -t1[q]; }
          K[10] = -t0[5]; K[11] = -t1[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int q = i; q < 5; ++q) {
              const double v = 0.5 * (Hxx[i][q] + Hxx[q][i]) - (Hux[0][i] * t0[q] + Hux[1][i] * t1[q]);
              P[i][q] = v; P[q][i] = v;
            }
            pv[i] = hx[i] - (Hux[0][i] * t0[5] + Hux[1][i] * t1[5]);
          }
        }
        for (int i = 0; i < 5; ++i) { for (int q = 0; q < 5; ++q) rP[i * 5 + q] = P[i][q]; rp[i] = pv[i]; }
      }
    CFZ_END
    CFZ_STAMP(11);  // Riccati backward sweep
    // ---- forward step and costates (lane 0) -----------------------------------------------------------
    CFZ_LANES(lane)
      if (lane == 0) {
        const double dt = sp.dt;
        const double *pv = rp;
        double P[5][5];
        for (int i = 0; i < 5; ++i) for (int q = 0; q < 5; ++q) P[i][q] = rP[i * 5 + q];
        // forward sweep
        double *dp = m + L.dp;
        for (int i = 0; i < 5; ++i) dp[i] = m[L.x0 + i] - m[L.p + i];
        // multiplier of the initial-state row from the value function at stage 0
        for (int i = 0; i < 5; ++i) { double s_ = pv[i]; for (int q = 0; q < 5; ++q) s_ += P[i][q] * dp[q]; m[L.dpi0 + i] = -s_ - m[L.pi0 + i]; }
        double z0 = dp[0], z1 = dp[1], z2 = dp[2], z3 = dp[3], z4 = dp[4];  // current dz kept in registers
#pragma unroll 5
        for (int k = 0; k < N; ++k) {  // unrolled so that the gain/dynamics loads of later stages are in flight early
          const double *K = m + L.kk + k * 12;
          const double u0 = K[10] + K[0] * z0 + K[1] * z1 + K[2] * z2 + K[3] * z3 + K[4] * z4;
          const double u1 = K[11] + K[5] * z0 + K[6] * z1 + K[7] * z2 + K[8] * z3 + K[9] * z4;
          dp[k * kNP + 5] = u0; dp[k * kNP + 6] = u1;
          if (k + 1 < N) {
            const double *s = m + L.ab + k * 15, *dk = m + L.d + k * 5;
            const double n0 = dk[0] + z0 + s[0] * z2 + s[1] * z3 + s[2] * z4 + s[3] * u0 + s[4] * u1;
            const double n1 = dk[1] + z1 + s[5] * z2 + s[6] * z3 + s[7] * z4 + s[8] * u0 + s[9] * u1;
            const double n2 = dk[2] + z2 + s[11] * z3 + s[12] * z4 + s[13] * u0 + s[14] * u1;
            const double n3 = dk[3] + z3 + dt * u0, n4 = dk[4] + z4 + dt * u1;
            z0 = n0; z1 = n1; z2 = n2; z3 = n3; z4 = n4;
            double *zn = dp + (k + 1) * kNP;
            zn[0] = z0; zn[1] = z1; zn[2] = z2; zn[3] = z3; zn[4] = z4;
          }
        }
      }
    CFZ_END
    // costates: pi_{k-1} = (H dp + g)_z at stage k + A_k' pi_k  (new multipliers of the dynamics rows).  The stage-local
    // part (H dp + g)_z is formed by one lane per stage into the slot of d(pi_{k-1}); only the 5-vector recursion
    // through A_k' stays on lane 0.
    CFZ_LANES(lane)
      if (lane >= 1 && lane < N) {
        const int k = lane;
        const double *h = m + L.hc + k * 11, *gk = m + L.gk + k * kNP, *z = m + L.dp + k * kNP;
        double *q = m + L.dpi + (k - 1) * 5;
        q[0] = gk[0] + h[0] * z[0] + h[7] * z[1] + h[8] * z[2];
        q[1] = gk[1] + h[7] * z[0] + h[1] * z[1] + h[9] * z[2];
        q[2] = gk[2] + h[8] * z[0] + h[9] * z[1] + h[2] * z[2];
        q[3] = gk[3] + h[3] * z[3] + h[10] * z[6];
        q[4] = gk[4] + h[4] * z[4];
      }
    CFZ_END
    CFZ_LANES(lane)
      if (lane == 0) {
        double lam[5] = {0, 0, 0, 0, 0};
#pragma unroll 5
        for (int k = N - 1; k >= 1; --k) {
          double *q = m + L.dpi + (k - 1) * 5;
          double nl[5] = {q[0], q[1], q[2], q[3], q[4]};
          if (k + 1 < N) {
            const double *s = m + L.ab + k * 15;
            nl[0] += lam[0]; nl[1] += lam[1];
            nl[2] += lam[2] + s[0] * lam[0] + s[5] * lam[1];
            nl[3] += lam[3] + s[1] * lam[0] + s[6] * lam[1] + s[11] * lam[2];
            nl[4] += lam[4] + s[2] * lam[0] + s[7] * lam[1] + s[12] * lam[2];
          }
          for (int i = 0; i < 5; ++i) { lam[i] = nl[i]; q[i] = nl[i] - m[L.pi + (k - 1) * 5 + i]; }
        }
      }
    CFZ_END
    CFZ_STAMP(5);  // Riccati
    // ---- slack step, fraction to the boundary, directional derivative ------------------------------------
    // The ratio tests keep the largest -d(.)/(.) and divide once at the end; 1/distance is formed once
    // per bound and reused (a DP division is ~12 dependent instructions on this pipe).
    CFZ_LANES(lane)
      double rpri = 0.0, rdual = 0.0, dphi = 0.0;  // max of -dx/dist and -dz/z
      if (lane < N) {
        const int k = lane;
        const double *pk = m + L.p + k * kNP, *dpk = m + L.dp + k * kNP;
        double g[kNP];
        stage_grad(sp, refg, k, pk, g);
        for (int q = 0; q < 6; ++q) {
          const double dx = dpk[bcol(q)];
          const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
          const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
          g[bcol(q)] += mu * (iu - il);
          const double dzl = mu * il - zl - zl * il * dx, dzu = mu * iu - zu_ + zu_ * iu * dx;
          rpri = fmax(rpri, fmax(-dx * il, dx * iu));
          rdual = fmax(rdual, fmax(-dzl / zl, -dzu / zu_));
        }
        for (int i = 0; i < kNP; ++i) dphi += g[i] * dpk[i];
        double ba0 = 0.0, ba1 = 0.0, bap[2] = {0.0, 0.0};
        // heading of the current iterate again: its cos/sin slots carried the value function through the Riccati sweeps
        double sps, cps;
        sincos(m[L.p + k * kNP + 2], &sps, &cps);
        for (int j = 0; j < nr; ++j) {
          const int t = k * nr + j;
          const double sg = m[L.sg + t], zs = m[L.zs + t], isg = 1.0 / sg;
          if ((j & 1) == 0) block_grad(sp, m, L, k, j >> 1, sel_ptr(m, L)[t >> 1], m[L.p + k * kNP], m[L.p + k * kNP + 1], cps, sps, ba0, ba1, bap);
          const double ds = m[L.cj + t] + ba0 * dpk[0] + ba1 * dpk[1] + bap[j & 1] * dpk[2];
          m[L.dsg + t] = ds;
          const double dzs = mu * isg - zs - zs * isg * ds;
          dphi -= mu * isg * ds;
          rpri = fmax(rpri, -ds * isg);
          rdual = fmax(rdual, -dzs / zs);
        }
      }
      m[L.red + 0 * 64 + lane] = rpri; m[L.red + 1 * 64 + lane] = rdual; m[L.red + 2 * 64 + lane] = dphi;
    CFZ_END
    const double rp_max = red_max(m, L, 0), rd_max = red_max(m, L, 1);
    const double a_pri = (rp_max > tau) ? tau / rp_max : 1.0, a_dual = (rd_max > tau) ? tau / rd_max : 1.0;
    const double dphi = red_sum(m, L, 2);
    CFZ_STAMP(6);  // step
    // ---- filter line search --------------------------------------------------------------------------------
    const double phi0 = fval - mu * logsum;
    if (filt_mu != mu) { nfilt = 0; filt_mu = mu; }
    double alpha = a_pri; int accepted = 0, f_type = 0;
    for (int bt = 0; bt < sp.max_backtrack; ++bt) {
      CFZ_LANES(lane)
        if (lane < N) sincos(m[L.p + lane * kNP + 2] + alpha * m[L.dp + lane * kNP + 2], &m[L.cs + 2 * lane + 1], &m[L.cs + 2 * lane]);
      CFZ_END
      CFZ_LANES(lane)
        merit_partials(sp, refg, m, L, alpha, lane);
      CFZ_END
      const double th_t = red_sum(m, L, 0), ph_t = red_sum(m, L, 1) - mu * red_sum(m, L, 2);
      int ok = (red_max(m, L, 3) == 0.0) && isfinite(th_t) && isfinite(ph_t) && th_t <= theta_max;
      if (ok) for (int q = 0; q < nfilt; ++q) if (th_t >= m[L.filt + 2 * q] && ph_t >= m[L.filt + 2 * q + 1]) { ok = 0; break; }
      f_type = 0;
      if (ok) {
        const int sw = theta <= theta_min && dphi < 0.0 && alpha * pow(-dphi, sp.s_phi) > sp.delta_sw * pow(theta, sp.s_theta);
        if (sw) { f_type = 1; ok = ph_t <= phi0 + sp.eta_phi * alpha * dphi; }
        else ok = th_t <= (1.0 - sp.gamma_theta) * theta || ph_t <= phi0 - sp.gamma_phi * theta;
      }
      if (ok) { accepted = 1; break; }
      alpha *= 0.5;
    }
    CFZ_STAMP(7);  // line search
    if (!accepted) { status = 2; break; }
    if (!f_type) {
      CFZ_LANES(lane)
        if (lane == 0) {
          int n = nfilt;
          if (n == sp.filter_cap) { for (int q = 0; q + 1 < n; ++q) { m[L.filt + 2 * q] = m[L.filt + 2 * q + 2]; m[L.filt + 2 * q + 1] = m[L.filt + 2 * q + 3]; } n--; }
          m[L.filt + 2 * n] = (1.0 - sp.gamma_theta) * theta; m[L.filt + 2 * n + 1] = phi0 - sp.gamma_phi * theta;
        }
      CFZ_END
      nfilt = (nfilt == sp.filter_cap) ? nfilt : nfilt + 1;
    }
    // ---- update ------------------------------------------------------------------------------------------------
    CFZ_LANES(lane)
      if (lane < 5) m[L.pi0 + lane] += alpha * m[L.dpi0 + lane];
      if (lane < N) {
        const int k = lane;
        double *pk = m + L.p + k * kNP; const double *dpk = m + L.dp + k * kNP;
        const double ks = sp.kappa_sigma, iks = 1.0 / sp.kappa_sigma;
        for (int q = 0; q < 6; ++q) {
          const double dx = dpk[bcol(q)];
          const double il = 1.0 / (pk[bcol(q)] - sp.bounds[2 * q]), iu = 1.0 / (sp.bounds[2 * q + 1] - pk[bcol(q)]);
          const double zl = m[L.zl + k * 6 + q], zu_ = m[L.zu + k * 6 + q];
          const double dzl = mu * il - zl - zl * il * dx, dzu = mu * iu - zu_ + zu_ * iu * dx;
          const double xn = pk[bcol(q)] + alpha * dx;
          const double mln = mu / (xn - sp.bounds[2 * q]), mun = mu / (sp.bounds[2 * q + 1] - xn);  // mu / new distance
          m[L.zl + k * 6 + q] = fmin(fmax(zl + a_dual * dzl, mln * iks), ks * mln);
          m[L.zu + k * 6 + q] = fmin(fmax(zu_ + a_dual * dzu, mun * iks), ks * mun);
        }
        for (int j = 0; j < nr; ++j) {
          const int t = k * nr + j;
          const double sg = m[L.sg + t], zs = m[L.zs + t], ds = m[L.dsg + t], isg = 1.0 / sg;
          const double S = zs * isg + sp.reg_primal;
          const double dnu = S * ds - mu * isg - m[L.nuc + t];
          const double dzs = mu * isg - zs - zs * isg * ds;
          const double sgn = sg + alpha * ds, msn = mu / sgn;
          m[L.sg + t] = sgn; m[L.nuc + t] += alpha * dnu;
          m[L.zs + t] = fmin(fmax(zs + a_dual * dzs, msn * iks), ks * msn);
        }
        for (int i = 0; i < kNP; ++i) pk[i] += alpha * dpk[i];
        if (k + 1 < N) for (int i = 0; i < 5; ++i) m[L.pi + k * 5 + i] += alpha * m[L.dpi + k * 5 + i];
      }
    CFZ_END
    CFZ_STAMP(8);  // update
  }

  CFZ_STAMP(10);
  // ---- write back: trajectory, separations, dual certificates ---------------------------------------------------
  CFZ_LANES(lane)
    for (int i = lane; i < N * kNP; i += 64) { const int k = i / kNP, c = i - k * kNP; zu[c * N + k] = m[L.p + i]; }
    double smin = INFINITY;
    for (int t = lane; t < N * nb; t += 64) {
      const int k = t / nb, j = t - k * nb;
      double A[4][2], b[4], V[4][2], sep2[2];
      block_polygon(sp, m, L, k, j, A, b, V);
      const double psi = m[L.p + k * kNP + 2];
      double s, c;
      sincos(psi, &s, &c);
      const int c1 = select_rows(A, b, V, m[L.p + k * kNP], m[L.p + k * kNP + 1], c, s, sp.g, sel_ptr(m, L)[t]);
      rows_for<false>(A, b, V, m[L.p + k * kNP], m[L.p + k * kNP + 1], c, s, sp.g, c1, sep2, nullptr);
      const double sep = fmin(sep2[0], sep2[1]);
      const int cert = (c1 >> 6) * 16 + ((c1 >> 4) & 3) * 4 + (sep2[0] <= sep2[1] ? ((c1 >> 2) & 3) : (c1 & 3));
      smin = fmin(smin, sep);
      if (duo.l) {
        const int kind = cert >> 4, f = (cert >> 2) & 3;
        double lam[4] = {0, 0, 0, 0}, muv[4] = {0, 0, 0, 0};
        if (j < n_obs) {
          if (kind == 1) {  // n = A_f ; G' mu = -R' n
            lam[f] = 1.0;
            const double mx = -(c * A[f][0] + s * A[f][1]), my = -(-s * A[f][0] + c * A[f][1]);
            muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
          } else {  // n = -R G_f ; A' lam = n from the two obstacle faces through vertex v
            muv[f] = 1.0;
            const int v = cert & 3;
            const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
            const double nx = -(c * gx - s * gy), ny = -(s * gx + c * gy);
            // the two faces active at vertex v: those with |A_i.V_v - b_i| smallest
            int i0 = 0, i1 = 1; double r0 = INFINITY, r1 = INFINITY;
            for (int i = 0; i < 4; ++i) {
              const double r = fabs(A[i][0] * V[v][0] + A[i][1] * V[v][1] - b[i]);
              if (r < r0) { r1 = r0; i1 = i0; r0 = r; i0 = i; } else if (r < r1) { r1 = r; i1 = i; }
            }
            const int ia = i0 < i1 ? i0 : i1, ib = i0 < i1 ? i1 : i0;
            const double det = A[ia][0] * A[ib][1] - A[ib][0] * A[ia][1];
            lam[ia] = fmax((A[ib][1] * nx - A[ib][0] * ny) / det, 0.0);
            lam[ib] = fmax((-A[ia][1] * nx + A[ia][0] * ny) / det, 0.0);
          }
          for (int i = 0; i < 4; ++i) { duo.l[k * 4 * n_obs + 4 * j + i] = lam[i]; duo.mm[k * 4 * n_obs + 4 * j + i] = muv[i]; }
        } else {
          const int o = j - n_obs;
          const double *q = m + L.nb4 + (k * n_nbr + o) * 4;
          const double co = q[2], so = q[3];
          double wx, wy;  // separating direction from this vehicle to the other
          const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
          if (kind == 1) {  // a face of the OTHER vehicle: w = -Ro G_f, mu = e_f, lam = posneg(R' w)
            wx = -(co * gx - so * gy); wy = -(so * gx + co * gy);
            muv[f] = 1.0;
            const double lx = c * wx + s * wy, ly = -s * wx + c * wy;
            lam[0] = fmax(lx, 0.0); lam[1] = fmax(ly, 0.0); lam[2] = fmax(-lx, 0.0); lam[3] = fmax(-ly, 0.0);
          } else {  // a face of this vehicle: w = R G_f, lam = e_f, mu = posneg(-Ro' w)
            wx = c * gx - s * gy; wy = s * gx + c * gy;
            lam[f] = 1.0;
            const double mx = -(co * wx + so * wy), my = -(-so * wx + co * wy);
            muv[0] = fmax(mx, 0.0); muv[1] = fmax(my, 0.0); muv[2] = fmax(-mx, 0.0); muv[3] = fmax(-my, 0.0);
          }
          for (int i = 0; i < 4; ++i) { duo.lam_ij[(o * N + k) * 4 + i] = lam[i]; duo.lam_ji[(o * N + k) * 4 + i] = muv[i]; }
          // s = -A_this' lam = -R G' lam   (vehicle_follower.py:350)
          const double glx = lam[0] - lam[2], gly = lam[1] - lam[3];
          duo.s[(o * N + k) * 2 + 0] = -(c * glx - s * gly);
          duo.s[(o * N + k) * 2 + 1] = -(s * glx + c * gly);
        }
      }
    }
    m[L.red + 0 * 64 + lane] = smin;
  CFZ_END
  out_d[0] = fval_last; out_d[1] = err0; out_d[2] = red_min(m, L, 0);
  out_i[0] = iter; out_i[1] = status;
  if (wst) {  // leave the multipliers for the next MPC iteration of this vehicle, or say that there are none
    CFZ_LANES(lane)
      if (status == 0) {
        for (int i = lane; i < N * nr; i += 64) wst[CL.z + i] = m[L.zs + i];
        for (int i = lane; i < N * 6; i += 64) { wst[CL.zl + i] = m[L.zl + i]; wst[CL.zu + i] = m[L.zu + i]; }
        for (int i = lane; i < N * 5; i += 64) wst[CL.pi + i] = i < (N - 1) * 5 ? m[L.pi + i] : 0.0;
        if (lane < 5) wst[CL.pi0 + lane] = m[L.pi0 + lane];
        unsigned char *ws = reinterpret_cast<unsigned char *>(wst + CL.sel);
        for (int i = lane; i < N * nb; i += 64) ws[i] = sel_ptr(m, L)[i];
        if (lane == 0) { wst[CL.mu] = mu; wst[CL.valid] = 1.0; }
      } else if (lane == 0) {
        wst[CL.valid] = 0.0;
      }
    CFZ_END
  }
  CFZ_STAMP(10);  // output
#if defined(CFZ_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
  // slot 0 ("setup", a handful of ticks) carries the wall time of the solve in 10 ns units (constant 100 MHz counter)
  stamp_acc[0] = wall_clock64() - stamp_wall0;
  if (duo.stamps && threadIdx.x == 0) for (int i = 0; i < 12; ++i) duo.stamps[i] = stamp_acc[i];
#endif
}

}  // namespace cfz
#endif  // CFZ_SOLVER_INL
