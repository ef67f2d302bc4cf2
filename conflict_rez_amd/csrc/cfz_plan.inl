// state_ws: the warm-start NLP of the single-vehicle plan (reference confrez/control/vehicle.py:99-231).
//
// T = N (S-1) forward-Euler steps of the kinematic bicycle (:173); initial pose fixed and v0 = delta0 = a0 = w0 = 0
// (:131-138); boxes on x, y, v, delta for k < T, on a, w only if `bounded_input` (:141-167); at every k = N i the
// rear-axle point inside the back cell and the front point (x + wb cos psi, y + wb sin psi) inside the front cell of
// strategy step i, both shrunk by `shrink_tube` (:178-192); optional terminal heading (:194-195); cost sum a^2 + w^2
// (:175-176).  Hundreds of stages, a handful of instances (one per vehicle), solved once: the opposite regime of the
// MPC step, so this is not a one-wavefront-in-LDS kernel.  One instance per workgroup, workspace in global memory
// (L2-resident), the interior-point iteration of oracle/ipm.py with the EXACT Hessian of the Lagrangian and the
// curvature test  dx'(H + delta I)dx >= kappa |dx|^2  (delta: 0, 1e-4, x8 ...), on the full primal-dual system in a
// stage-interleaved ordering that makes it banded (half-bandwidth <= 40) -- solved by a banded LU with partial
// pivoting, so terminal and initial equalities, tube rows and indefinite stage Hessians need no special cases.
//
// The same source compiles for the CPU (tests/emu) and is checked iterate for iterate against oracle/plan_nlp.py.
#pragma once
#include <math.h>

#include "cfz_band.inl"

#if defined(__HIPCC__)
#define CFZP_FN __host__ __device__ inline
#else
#define CFZP_FN inline
#endif

// On the GPU all 64 lanes of the workgroup run the solver redundantly (same scalars, same addresses: one memory
// transaction per instruction, so it costs what one lane would); only the loops marked CFZP_LANE_FOR split their
// iterations over the lanes, with a workgroup barrier before anybody reads what another lane wrote.  On the CPU the
// marked loops simply run in full.
#if defined(__HIP_DEVICE_COMPILE__)
#define CFZP_LANE_FOR(q, lo, hi) for (int q = (lo) + (int)threadIdx.x; q <= (hi); q += (int)blockDim.x)
#define CFZP_SYNC() __syncthreads()
#else
#define CFZP_LANE_FOR(q, lo, hi) for (int q = (lo); q <= (hi); ++q)
#define CFZP_SYNC() do {} while (0)
#endif

namespace cfzp {

// reductions over the lanes' partial results of a CFZP_LANE_FOR loop (identity on the CPU, where the loop ran in full)
CFZP_FN double wsum(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
#endif
  return v;
}
CFZP_FN double wmax(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
#endif
  return v;
}
CFZP_FN double wmin(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
#endif
  return v;
}

// The same over a workgroup of several wavefronts (the eight-wavefront kernels: every thread runs the scalar logic, the marked
// loops are split over all threads); the per-wavefront values are combined through LDS in a fixed order, so that every thread
// gets the same bits.  With one wavefront they are wsum / wmax / wmin.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline double pcombine(double v, int op) {  // op 0 sum, 1 max, 2 min
  __shared__ double part[16];
  const int nw = (int)(blockDim.x >> 6);
  if (nw <= 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = part[0];
  for (int i = 1; i < nw; ++i) r = op == 0 ? r + part[i] : (op == 1 ? fmax(r, part[i]) : fmin(r, part[i]));
  return r;
}
CFZP_FN double psum(double v) { return pcombine(wsum(v), 0); }
CFZP_FN double pmax(double v) { return pcombine(wmax(v), 1); }
CFZP_FN double pmin(double v) { return pcombine(wmin(v), 2); }
#else
CFZP_FN double psum(double v) { return v; }
CFZP_FN double pmax(double v) { return v; }
CFZP_FN double pmin(double v) { return v; }
#endif

constexpr int kKB = 40;               // half-bandwidth of the permuted KKT matrix (asserted at set-up)
constexpr int kLd = 3 * kKB + 1;      // band storage rows (LAPACK gb layout with room for the pivoting fill-in)

struct PSpec {
  int T, N, n_chk, has_final, bounded_input;
  int max_iter, max_backtrack, filter_cap, stall_iters, pad0;
  double dt, wb, shrink, final_heading;
  double init_pose[3];
  double bounds[12];  // lo,hi for x, y, v, delta, a, w
  double tol, constr_viol_tol, dual_inf_tol, compl_inf_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min,
      bound_push, bound_frac, s_max, kappa_sigma, eta_phi, gamma_theta, gamma_phi, delta_sw, s_theta, s_phi,
      reg_primal, reg_dual, curv_kappa, stall_kappa;
};

struct PDims { int n, m, s0, r0, nk; };
CFZP_FN PDims dims(const PSpec &sp) {
  PDims d;
  d.n = 7 * sp.T + 5 + 8 * sp.n_chk; d.m = 7 + 5 * sp.T + 8 * sp.n_chk + (sp.has_final ? 1 : 0);
  d.s0 = 7 * sp.T + 5; d.r0 = 7 + 5 * sp.T; d.nk = d.n + d.m;
  return d;
}

// workspace (doubles unless noted), carved out of one slab by `carve`
struct PWork {
  double *x, *xt, *zl, *zu, *nu, *dx, *dnu, *dzl, *dzu, *g, *c, *ct, *xl, *xu, *r1, *rhs, *ab, *hd;
  int *posx, *posc, *ipiv;
};
CFZP_FN size_t work_doubles(const PSpec &sp) {
  const PDims d = dims(sp);
  return (size_t)d.n * 12 + (size_t)d.m * 4 + (size_t)d.nk * (1 + kLd) + (size_t)(d.n + d.m + d.nk + 2) / 2 + 64;
}
CFZP_FN PWork carve(const PSpec &sp, double *slab) {
  const PDims d = dims(sp);
  PWork w; double *p = slab;
  w.x = p; p += d.n; w.xt = p; p += d.n; w.zl = p; p += d.n; w.zu = p; p += d.n; w.dx = p; p += d.n; w.dzl = p; p += d.n;
  w.dzu = p; p += d.n; w.g = p; p += d.n; w.xl = p; p += d.n; w.xu = p; p += d.n; w.r1 = p; p += d.n; w.hd = p; p += d.n;
  w.nu = p; p += d.m; w.dnu = p; p += d.m; w.c = p; p += d.m; w.ct = p; p += d.m;
  w.rhs = p; p += d.nk; w.ab = p; p += (size_t)d.nk * kLd;
  w.posx = reinterpret_cast<int *>(p); w.posc = w.posx + d.n; w.ipiv = w.posc + d.m;
  return w;
}

CFZP_FN int chk_stage(const PSpec &sp, int i) { return sp.N * (i + 1); }
// tube[i][0] = back cell, tube[i][1] = front cell: A[4][2] row-major then b[4]
CFZP_FN const double *cell(const double *tube, int i, int front) { return tube + ((size_t)i * 2 + front) * 12; }

// ---- problem functions (oracle/plan_nlp.py StateWsNlp) ---------------------------------------------------------
CFZP_FN double objective(const PSpec &sp, const double *X) {
  double f = 0.0;
  CFZP_LANE_FOR(k, 0, sp.T - 1) f += X[7 * k + 5] * X[7 * k + 5] + X[7 * k + 6] * X[7 * k + 6];
  return psum(f);
}

CFZP_FN void constraints(const PSpec &sp, const double *tube, const double *X, double *c) {
  const PDims d = dims(sp);
  for (int i = 0; i < 3; ++i) c[i] = X[i] - sp.init_pose[i];
  for (int i = 3; i < 7; ++i) c[i] = X[i];
  CFZP_LANE_FOR(k, 0, sp.T - 1) {
    const double *z = X + 7 * k, *zn = X + 7 * (k + 1);
    const double cs = cos(z[2]), sn = sin(z[2]), tn = tan(z[4]);
    const double f[5] = {z[3] * cs, z[3] * sn, z[3] / sp.wb * tn, z[5], z[6]};
    for (int i = 0; i < 5; ++i) c[7 + 5 * k + i] = z[i] + sp.dt * f[i] - zn[i];
  }
  CFZP_LANE_FOR(i, 0, sp.n_chk - 1) {
    const double *z = X + 7 * chk_stage(sp, i);
    const double fx = z[0] + sp.wb * cos(z[2]), fy = z[1] + sp.wb * sin(z[2]);
    const double *cb = cell(tube, i, 0), *cf = cell(tube, i, 1);
    for (int q = 0; q < 4; ++q) {
      c[d.r0 + 8 * i + q] = cb[2 * q] * z[0] + cb[2 * q + 1] * z[1] - (cb[8 + q] - sp.shrink) + X[d.s0 + 8 * i + q];
      c[d.r0 + 8 * i + 4 + q] = cf[2 * q] * fx + cf[2 * q + 1] * fy - (cf[8 + q] - sp.shrink) + X[d.s0 + 8 * i + 4 + q];
    }
  }
  if (sp.has_final) c[d.m - 1] = X[7 * sp.T + 2] - sp.final_heading;
  CFZP_SYNC();
}

// out = J(X)' nu; every stage block is written by one lane (own Euler rows minus the previous stage's), then the tube rows
CFZP_FN void jt_nu(const PSpec &sp, const double *tube, const double *X, const double *nu, double *out) {
  const PDims d = dims(sp);
  CFZP_LANE_FOR(k, 0, sp.T) {
    double o[7] = {0, 0, 0, 0, 0, 0, 0};
    if (k < sp.T) {
      const double *z = X + 7 * k, *l = nu + 7 + 5 * k;
      const double cs = cos(z[2]), sn = sin(z[2]), tn = tan(z[4]), dt = sp.dt;
      for (int i = 0; i < 5; ++i) o[i] += l[i];
      o[2] += dt * (-z[3] * sn * l[0] + z[3] * cs * l[1]);
      o[3] += dt * (cs * l[0] + sn * l[1] + tn / sp.wb * l[2]);
      o[4] += dt * (z[3] / sp.wb * (1.0 + tn * tn) * l[2]);
      o[5] += dt * l[3]; o[6] += dt * l[4];
    }
    if (k > 0) { const double *lp = nu + 7 + 5 * (k - 1); for (int i = 0; i < 5; ++i) o[i] -= lp[i]; }
    if (k == 0) for (int i = 0; i < 7; ++i) o[i] += nu[i];
    if (k == sp.T && sp.has_final) o[2] += nu[d.m - 1];
    for (int i = 0; i < (k < sp.T ? 7 : 5); ++i) out[7 * k + i] = o[i];
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, sp.n_chk - 1) {
    const int b = 7 * chk_stage(sp, i);
    const double cs = cos(X[b + 2]), sn = sin(X[b + 2]);
    const double *cb = cell(tube, i, 0), *cf = cell(tube, i, 1);
    for (int q = 0; q < 4; ++q) {
      const double lb = nu[d.r0 + 8 * i + q], lf = nu[d.r0 + 8 * i + 4 + q];
      out[b] += cb[2 * q] * lb + cf[2 * q] * lf; out[b + 1] += cb[2 * q + 1] * lb + cf[2 * q + 1] * lf;
      out[b + 2] += sp.wb * (-cf[2 * q] * sn + cf[2 * q + 1] * cs) * lf;
      out[d.s0 + 8 * i + q] = lb; out[d.s0 + 8 * i + 4 + q] = lf;
    }
  }
  CFZP_SYNC();
}

// ---- banded storage -----------------------------------------------------------------------------------------
CFZP_FN double &band(double *ab, int i, int j) { return ab[(size_t)j * kLd + (2 * kKB + i - j)]; }

// stage-interleaved ordering: [init rows | z_0 u_0 | dyn_0 | z_1 u_1 | (slacks, tube rows at checkpoints) | dyn_1 | ...]
CFZP_FN int build_order(const PSpec &sp, int *posx, int *posc) {
  const PDims d = dims(sp);
  int p = 0;
  for (int k = 0; k <= sp.T; ++k) {
    if (k == 0) for (int i = 0; i < 7; ++i) posc[i] = p++;
    const int nv = k < sp.T ? 7 : 5;
    for (int i = 0; i < nv; ++i) posx[7 * k + i] = p++;
    if (k > 0 && k % sp.N == 0 && k / sp.N - 1 < sp.n_chk) {
      const int i = k / sp.N - 1;
      for (int q = 0; q < 8; ++q) posx[d.s0 + 8 * i + q] = p++;
      for (int q = 0; q < 8; ++q) posc[d.r0 + 8 * i + q] = p++;
    }
    if (k == sp.T && sp.has_final) posc[d.m - 1] = p++;
    if (k < sp.T) for (int i = 0; i < 5; ++i) posc[7 + 5 * k + i] = p++;
  }
  return p;
}

CFZP_FN void put(double *ab, int i, int j, double v) { band(ab, i, j) += v; if (i != j) band(ab, j, i) += v; }

// KKT matrix [[W + Sigma + (delta + reg) I, J'], [J, 0]] in band storage; hd returns the primal diagonal shift applied
CFZP_FN void assemble(const PSpec &sp, const double *tube, const PWork &w, const double *sig, double delta) {
  const PDims d = dims(sp);
  const int *px = w.posx, *pc = w.posc;
  CFZP_LANE_FOR(col, 0, d.nk - 1) for (int r = 0; r < kLd; ++r) w.ab[(size_t)col * kLd + r] = 0.0;
  CFZP_SYNC();
  const double *X = w.x, *nu = w.nu;
  CFZP_LANE_FOR(i, 0, d.n - 1) band(w.ab, px[i], px[i]) += sig[i] + delta + sp.reg_primal;
  // IPOPT's delta_c: with v = delta = 0 in the guess the heading rows lose rank once the terminal heading is fixed
  CFZP_LANE_FOR(i, 0, d.m - 1) band(w.ab, pc[i], pc[i]) -= sp.reg_dual;
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, 6) put(w.ab, pc[i], px[i], 1.0);
  CFZP_LANE_FOR(k, 0, sp.T - 1) {  // every entry written here belongs to stage k alone
    const double *z = X + 7 * k, *l = nu + 7 + 5 * k;
    const int b = 7 * k, bn = 7 * (k + 1), r = 7 + 5 * k;
    const double cs = cos(z[2]), sn = sin(z[2]), tn = tan(z[4]), dt = sp.dt, sec2 = 1.0 + tn * tn, v = z[3];
    // objective and dynamics curvature (multipliers of the x, y, psi rows)
    band(w.ab, px[b + 5], px[b + 5]) += 2.0; band(w.ab, px[b + 6], px[b + 6]) += 2.0;
    const double l0 = l[0] * dt, l1 = l[1] * dt, l2 = l[2] * dt;
    band(w.ab, px[b + 2], px[b + 2]) += l0 * (-v * cs) + l1 * (-v * sn);
    put(w.ab, px[b + 2], px[b + 3], l0 * (-sn) + l1 * cs);
    put(w.ab, px[b + 3], px[b + 4], l2 * sec2 / sp.wb);
    band(w.ab, px[b + 4], px[b + 4]) += l2 * 2.0 * v * tn * sec2 / sp.wb;
    // Jacobian of the Euler rows
    for (int i = 0; i < 5; ++i) { put(w.ab, pc[r + i], px[b + i], 1.0); put(w.ab, pc[r + i], px[bn + i], -1.0); }
    put(w.ab, pc[r + 0], px[b + 2], dt * (-v * sn)); put(w.ab, pc[r + 0], px[b + 3], dt * cs);
    put(w.ab, pc[r + 1], px[b + 2], dt * (v * cs)); put(w.ab, pc[r + 1], px[b + 3], dt * sn);
    put(w.ab, pc[r + 2], px[b + 3], dt * tn / sp.wb); put(w.ab, pc[r + 2], px[b + 4], dt * v / sp.wb * sec2);
    put(w.ab, pc[r + 3], px[b + 5], dt); put(w.ab, pc[r + 4], px[b + 6], dt);
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, sp.n_chk - 1) {
    const int b = 7 * chk_stage(sp, i), r = d.r0 + 8 * i, s = d.s0 + 8 * i;
    const double cs = cos(X[b + 2]), sn = sin(X[b + 2]);
    const double *cb = cell(tube, i, 0), *cf = cell(tube, i, 1);
    double curv = 0.0;
    for (int q = 0; q < 4; ++q) {
      put(w.ab, pc[r + q], px[b], cb[2 * q]); put(w.ab, pc[r + q], px[b + 1], cb[2 * q + 1]); put(w.ab, pc[r + q], px[s + q], 1.0);
      put(w.ab, pc[r + 4 + q], px[b], cf[2 * q]); put(w.ab, pc[r + 4 + q], px[b + 1], cf[2 * q + 1]);
      put(w.ab, pc[r + 4 + q], px[b + 2], sp.wb * (-cf[2 * q] * sn + cf[2 * q + 1] * cs));
      put(w.ab, pc[r + 4 + q], px[s + 4 + q], 1.0);
      curv += nu[r + 4 + q] * sp.wb * (-cf[2 * q] * cs - cf[2 * q + 1] * sn);
    }
    band(w.ab, px[b + 2], px[b + 2]) += curv;
  }
  CFZP_LANE_FOR(one, 0, 0) if (sp.has_final) put(w.ab, pc[d.m - 1], px[7 * sp.T + 2], 1.0);  // (one thread: += is not idempotent)
  CFZP_SYNC();
}

// LU with partial pivoting of an n x n band matrix (kl = ku = kKB) in LAPACK gb layout, then one solve; 0 = ok
// `win`: optional fast storage (LDS on the GPU) for the kv+1 = 81 columns the elimination is working on; column q lives
// in slot q mod 81 while j <= q <= j + kv, enters from `ab` when pivot j = q - kv starts and is written back after
// its own pivot step.  nullptr: work in `ab` directly.
constexpr int kWinCols = 2 * kKB + 1;
template <bool WIN>
CFZP_FN int band_solve(double *ab, int n, int *ipiv, double *b, double *win) {
  const int kl = kKB, ku = kKB, kv = kl + ku;
// WIN is a compile-time switch, and on the GPU the window is named directly (the kernel's dynamic LDS) rather than taken
// from the argument, so that every access to it is a DS instruction: through a generic pointer most of them became FLAT
// instructions, which wait on the global-memory counter as well
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ double cfzp_lds[];
#define CFZP_WIN_BASE cfzp_lds
#else
#define CFZP_WIN_BASE win
#endif
#define CFZP_COL(q) (WIN ? CFZP_WIN_BASE + (size_t)((q) % kWinCols) * kLd : ab + (size_t)(q) * kLd)
  if (WIN) {
    const int last = kv < n - 1 ? kv : n - 1;
    for (int q = 0; q <= last; ++q) CFZP_LANE_FOR(r, 0, kLd - 1) CFZP_WIN_BASE[(size_t)q * kLd + r] = ab[(size_t)q * kLd + r];
    CFZP_SYNC();
  }
  int ju = 0;
  for (int j = 0; j < n; ++j) {
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    double *cj = CFZP_COL(j);
    int jp = 0; double best;
#if defined(__HIP_DEVICE_COMPILE__)
    {  // pivot search: one candidate per lane, butterfly arg-max (first maximum wins, as in the serial loop)
      const int lane = threadIdx.x;
      best = lane <= km ? fabs(cj[kv + lane]) : -1.0; jp = lane;
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off); const int oj = __shfl_xor(jp, off);
        if (ob > best || (ob == best && oj < jp)) { best = ob; jp = oj; }
      }
    }
#else
    best = fabs(cj[kv]);
    for (int i = 1; i <= km; ++i) { const double a = fabs(cj[kv + i]); if (a > best) { best = a; jp = i; } }
#endif
    ipiv[j] = j + jp;
    if (!(best > 0.0)) return 1;
    const int reach = j + ku + jp; ju = ju > (reach < n - 1 ? reach : n - 1) ? ju : (reach < n - 1 ? reach : n - 1);
    if (jp != 0) {
      CFZP_LANE_FOR(q, j, ju) {  // swap rows j and j+jp over columns j..ju
        double *cq = CFZP_COL(q);
        double &a = cq[kv + j - q], &c = cq[kv + j + jp - q];
        const double t = a; a = c; c = t;
      }
      CFZP_SYNC();
    }
    const double inv = 1.0 / cj[kv];
    CFZP_SYNC();
    CFZP_LANE_FOR(i, 1, km) cj[kv + i] *= inv;
    CFZP_SYNC();
    CFZP_LANE_FOR(q, j + 1, ju) {  // rank-1 update of the trailing window, one column per lane
      double *cq = CFZP_COL(q);
      const double u = cq[kv + j - q];
      if (u != 0.0) for (int i = 1; i <= km; ++i) cq[kv + j + i - q] -= cj[kv + i] * u;
    }
    CFZP_SYNC();
    if (WIN) {  // column j is final: back to `ab`; its slot takes column j + kv + 1
      CFZP_LANE_FOR(r, 0, kLd - 1) ab[(size_t)j * kLd + r] = cj[r];
      CFZP_SYNC();
      const int qn = j + kv + 1;
      if (qn < n) { CFZP_LANE_FOR(r, 0, kLd - 1) cj[r] = ab[(size_t)qn * kLd + r]; }
      CFZP_SYNC();
    }
  }
#undef CFZP_COL
#undef CFZP_WIN_BASE
  // the swap and the division are not idempotent: ONE thread does them (every thread runs this scalar code; inside one wavefront
  // the lanes' redundant read-modify-writes happen to coincide, across wavefronts they would repeat -- the bug class that bit
  // `assemble` when state_ws went to eight wavefronts)
#if defined(__HIP_DEVICE_COMPILE__)
  const bool first = threadIdx.x == 0;
#else
  const bool first = true;
#endif
  for (int j = 0; j < n; ++j) {  // L y = P b
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j, p = ipiv[j];
    if (p != j) {
      if (first) { const double t = b[j]; b[j] = b[p]; b[p] = t; }
      CFZP_SYNC();
    }
    const double bj = b[j];
    CFZP_SYNC();
    if (bj != 0.0) CFZP_LANE_FOR(i, 1, km) b[j + i] -= ab[(size_t)j * kLd + kv + i] * bj;
    CFZP_SYNC();
  }
  for (int j = n - 1; j >= 0; --j) {  // U x = y
    const double bj = b[j] / ab[(size_t)j * kLd + kv];
    CFZP_SYNC();
    if (first) b[j] = bj;
    const int lo = j - kv > 0 ? j - kv : 0;
    if (bj != 0.0) CFZP_LANE_FOR(i, lo, j - 1) b[i] -= ab[(size_t)j * kLd + kv + i - j] * bj;
    CFZP_SYNC();
  }
  return 0;
}

CFZP_FN double barrier_obj(const PSpec &sp, const PWork &w, const double *X, double mu) {
  const PDims d = dims(sp);
  double s = 0.0, bad = 0.0;
  CFZP_LANE_FOR(i, 0, d.n - 1) {
    if (w.xl[i] > -1e300) { const double dl = X[i] - w.xl[i]; if (!(dl > 0.0)) bad = 1.0; else s += log(dl); }
    if (w.xu[i] < 1e300) { const double du = w.xu[i] - X[i]; if (!(du > 0.0)) bad = 1.0; else s += log(du); }
  }
  if (pmax(bad) > 0.0) return INFINITY;
  return objective(sp, X) - mu * psum(s);
}

}  // namespace cfzp
#if defined(__HIP_DEVICE_COMPILE__)
namespace cfzc {  // the eight-wavefront elimination and substitution of cfz_colloc.inl (defined there, after this file)
__device__ inline int band_factor_panel(const cfzb::Band &B, int n, int *ipiv, long long *ptk, double *lds, double *b1, double *b2);
__device__ inline void band_substitute_regs(const cfzb::Band &B, int n, const int *ipiv, double *b, double *b2, bool fwd_done);
}
#endif
namespace cfzp {

// WIDE: eight wavefronts per plan, band eliminated from global memory a panel at a time (cfz_colloc.inl) -- faster per plan than
// the one-wavefront LDS window, one plan per CU instead of two.
template <bool WIN, bool WIDE = false>
CFZP_FN void solve_state_ws(const PSpec &sp, const double *tube, double *X, double *slab, int *out_i, double *out_d,
                            double *win) {
  const PDims d = dims(sp);
  const PWork w = carve(sp, slab);
  const int n = d.n, m = d.m;
  build_order(sp, w.posx, w.posc);
  // bounds
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, n - 1) { w.xl[i] = i >= d.s0 ? 0.0 : -INFINITY; w.xu[i] = INFINITY; w.x[i] = i < d.s0 ? X[i] : 0.0; }
  CFZP_SYNC();
  CFZP_LANE_FOR(k, 0, sp.T - 1) {
    const int col[6] = {0, 1, 3, 4, 5, 6};
    for (int q = 0; q < (sp.bounded_input ? 6 : 4); ++q) { w.xl[7 * k + col[q]] = sp.bounds[2 * q]; w.xu[7 * k + col[q]] = sp.bounds[2 * q + 1]; }
  }
  CFZP_SYNC();
  // slacks from the guess (sigma = -(A p - b + shrink)), then push everything inside its bounds
  constraints(sp, tube, w.x, w.c);
  CFZP_LANE_FOR(i, 0, 8 * sp.n_chk - 1) w.x[d.s0 + i] = -w.c[d.r0 + i];
  CFZP_SYNC();
  double nbd = 0.0;
  CFZP_LANE_FOR(i, 0, n - 1) {
    const bool hl = w.xl[i] > -1e300, hu = w.xu[i] < 1e300;
    double pl = hl ? sp.bound_push * fmax(1.0, fabs(w.xl[i])) : 0.0, pu = hu ? sp.bound_push * fmax(1.0, fabs(w.xu[i])) : 0.0;
    if (hl && hu) { pl = fmin(pl, sp.bound_frac * (w.xu[i] - w.xl[i])); pu = fmin(pu, sp.bound_frac * (w.xu[i] - w.xl[i])); }
    if (hl) w.x[i] = fmax(w.x[i], w.xl[i] + pl);
    if (hu) w.x[i] = fmin(w.x[i], w.xu[i] - pu);
    w.zl[i] = hl ? 1.0 : 0.0; w.zu[i] = hu ? 1.0 : 0.0; nbd += hl + hu;
  }
  const int nb = (int)psum(nbd);
  CFZP_LANE_FOR(i, 0, m - 1) w.nu[i] = 0.0;
  CFZP_SYNC();
  double mu = sp.mu_init, filt_mu = -1.0, theta_min = -1.0, theta_max = -1.0, err0 = INFINITY;
  const double mu_floor = fmin(sp.tol, sp.compl_inf_tol) / (sp.kappa_eps + 1.0);
  double filt[64][2]; int nfilt = 0;
  double delta_last = 0.0;  // the last nonzero primal perturbation of the inertia correction (Algorithm IC)
  double delta_floor = 0.0; int delta_retry = 0;  // a failed line search is repeated with a larger perturbation (below)
  bool mu_forced = false;  // the last iteration ended without a step and lowered mu instead
  double stall_ref = 0.0; int stall_cnt = 0;
  int status = 1, iter = 0;
  for (iter = 0; iter <= sp.max_iter; ++iter) {
    constraints(sp, tube, w.x, w.c);
    CFZP_LANE_FOR(i, 0, n - 1) { const int c7 = i < d.s0 ? i % 7 : 0; w.g[i] = (i < 7 * sp.T && c7 >= 5) ? 2.0 * w.x[i] : 0.0; }
    jt_nu(sp, tube, w.x, w.nu, w.r1);  // J' nu (ends with a barrier)
    double theta = 0.0, cviol = 0.0, sum_nu = 0.0, sum_z = 0.0, dual_inf = 0.0;
    CFZP_LANE_FOR(i, 0, m - 1) { theta += fabs(w.c[i]); cviol = fmax(cviol, fabs(w.c[i])); sum_nu += fabs(w.nu[i]); }
    theta = psum(theta); cviol = pmax(cviol); sum_nu = psum(sum_nu);
    if (theta_min < 0.0) { theta_min = 1e-4 * fmax(1.0, theta); theta_max = 1e4 * fmax(1.0, theta); }
    CFZP_LANE_FOR(i, 0, n - 1) { sum_z += w.zl[i] + w.zu[i]; dual_inf = fmax(dual_inf, fabs(w.g[i] + w.r1[i] - w.zl[i] + w.zu[i])); }
    sum_z = psum(sum_z); dual_inf = pmax(dual_inf);
    const double s_d = fmax(sp.s_max, (sum_nu + sum_z) / (double)(m + nb)) / sp.s_max, s_c = fmax(sp.s_max, sum_z / (double)nb) / sp.s_max;
    double cmp0 = 0.0;
    CFZP_LANE_FOR(i, 0, n - 1) {
      if (w.xl[i] > -1e300) cmp0 = fmax(cmp0, fabs((w.x[i] - w.xl[i]) * w.zl[i]));
      if (w.xu[i] < 1e300) cmp0 = fmax(cmp0, fabs((w.xu[i] - w.x[i]) * w.zu[i]));
    }
    cmp0 = pmax(cmp0);
    err0 = fmax(dual_inf / s_d, fmax(cviol, cmp0 / s_c));
    if (!isfinite(err0)) { status = 3; break; }
    if (err0 <= sp.tol && dual_inf <= sp.dual_inf_tol && cviol <= sp.constr_viol_tol && cmp0 <= sp.compl_inf_tol) { status = 0; break; }
    if (iter == sp.max_iter) { status = 1; break; }
    if (iter == 0 || cviol <= sp.stall_kappa * stall_ref) { stall_ref = cviol; stall_cnt = 0; } else ++stall_cnt;
    if (sp.stall_iters > 0 && stall_cnt >= sp.stall_iters && cviol > sp.constr_viol_tol) { status = 5; break; }
    while (mu > mu_floor) {  // barrier update
      double cm = 0.0;
      CFZP_LANE_FOR(i, 0, n - 1) {
        if (w.xl[i] > -1e300) cm = fmax(cm, fabs((w.x[i] - w.xl[i]) * w.zl[i] - mu));
        if (w.xu[i] < 1e300) cm = fmax(cm, fabs((w.xu[i] - w.x[i]) * w.zu[i] - mu));
      }
      cm = pmax(cm);
      if (fmax(dual_inf / s_d, fmax(cviol, cm / s_c)) <= sp.kappa_eps * mu) mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu)));
      else break;
    }
    const double tau = fmax(sp.tau_min, 1.0 - mu);
    // gradient of the barrier problem's Lagrangian -> r1; Sigma -> hd
    double *sig = w.hd;
    CFZP_LANE_FOR(i, 0, n - 1) {
      double gphi = w.g[i], s = 0.0;
      if (w.xl[i] > -1e300) { const double dl = w.x[i] - w.xl[i]; gphi -= mu / dl; s += w.zl[i] / dl; }
      if (w.xu[i] < 1e300) { const double du = w.xu[i] - w.x[i]; gphi += mu / du; s += w.zu[i] / du; }
      w.g[i] = gphi; w.r1[i] = gphi + w.r1[i]; sig[i] = s;
    }
    CFZP_SYNC();
    // Newton step with the curvature test
    double delta = delta_floor; bool have = false;
    for (int tries = 0; tries < 60; ++tries) {
      assemble(sp, tube, w, sig, delta);
      CFZP_LANE_FOR(i, 0, n - 1) w.rhs[w.posx[i]] = -w.r1[i];
      CFZP_LANE_FOR(i, 0, m - 1) w.rhs[w.posc[i]] = -w.c[i];
      CFZP_SYNC();
      int fail;
#if defined(__HIP_DEVICE_COMPILE__)
      if (WIDE) {
        const cfzb::Band Bd = {w.ab, kKB, kLd};
        long long unused[3] = {0, 0, 0};
        fail = cfzc::band_factor_panel(Bd, d.nk, w.ipiv, unused, win, w.rhs, nullptr);  // the right-hand side rides along
        if (!fail) cfzc::band_substitute_regs(Bd, d.nk, w.ipiv, w.rhs, w.rhs, true);
      } else if (WIN && d.nk <= kWinCols * kLd) {  // the batched LDS elimination of cfz_band.inl; the right-hand side follows in LDS
        const cfzb::Band Bd = {w.ab, kKB, kLd};
        long long unused[3] = {0, 0, 0};
        fail = cfzb::band_factor_lds(Bd, d.nk, w.ipiv, unused);
        if (!fail) cfzb::band_substitute_lds<false>(Bd, d.nk, w.ipiv, w.rhs, nullptr);
      } else
#endif
      fail = band_solve<WIN>(w.ab, d.nk, w.ipiv, w.rhs, win);
      if (!fail) {
        double curv = 0.0, dd = 0.0, bad = 0.0;  // dx'(H) dx = -dx.r1 + c.dnu - reg_dual |dnu|^2  (from the two block rows of the system)
        CFZP_LANE_FOR(i, 0, n - 1) { const double v = w.rhs[w.posx[i]]; if (!isfinite(v)) bad = 1.0; w.dx[i] = v; curv -= v * w.r1[i]; dd += v * v; }
        CFZP_LANE_FOR(i, 0, m - 1) { const double v = w.rhs[w.posc[i]]; if (!isfinite(v)) bad = 1.0; w.dnu[i] = v; curv += w.c[i] * v - sp.reg_dual * v * v; }
        curv = psum(curv); dd = psum(dd); bad = pmax(bad);
        CFZP_SYNC();
        if (bad == 0.0 && curv >= sp.curv_kappa * dd) { have = true; break; }
      }
      // IPOPT's Algorithm IC (oracle/ipm.py next_delta_w): after delta = 0 a third of the last perturbation that worked (1e-4 if
      // there has been none), then x 8
      delta = delta == 0.0 ? (delta_last == 0.0 ? 1e-4 : fmax(1e-20, delta_last / 3.0)) : delta * 8.0;
      if (delta > 1e20) break;
    }
    if (!have) { status = 3; break; }
#if defined(CFZP_TRACE)  // CPU build only: g++ -DCFZP_TRACE -include stdio.h
    printf("it %3d mu %.2e delta %.3e (last %.3e)\n", iter, mu, delta, delta_last);
#endif
    if (delta > 0.0) delta_last = delta;
    double a_pri = 1.0, a_dual = 1.0, dphi = 0.0;
    CFZP_LANE_FOR(i, 0, n - 1) {
      const double dxi = w.dx[i];
      dphi += w.g[i] * dxi;
      w.dzl[i] = 0.0; w.dzu[i] = 0.0;
      if (w.xl[i] > -1e300) {
        const double dl = w.x[i] - w.xl[i];
        w.dzl[i] = mu / dl - w.zl[i] - w.zl[i] / dl * dxi;
        if (dxi < 0.0) a_pri = fmin(a_pri, -tau * dl / dxi);
        if (w.dzl[i] < 0.0) a_dual = fmin(a_dual, -tau * w.zl[i] / w.dzl[i]);
      }
      if (w.xu[i] < 1e300) {
        const double du = w.xu[i] - w.x[i];
        w.dzu[i] = mu / du - w.zu[i] + w.zu[i] / du * dxi;
        if (dxi > 0.0) a_pri = fmin(a_pri, tau * du / dxi);
        if (w.dzu[i] < 0.0) a_dual = fmin(a_dual, -tau * w.zu[i] / w.dzu[i]);
      }
    }
    a_pri = pmin(a_pri); a_dual = pmin(a_dual); dphi = psum(dphi);
    CFZP_SYNC();
    const double phi0 = barrier_obj(sp, w, w.x, mu);
    if (filt_mu != mu) { nfilt = 0; filt_mu = mu; }
    double alpha = a_pri; bool accepted = false, f_type = false;
    for (int bt = 0; bt < sp.max_backtrack; ++bt) {
      CFZP_LANE_FOR(i, 0, n - 1) w.xt[i] = w.x[i] + alpha * w.dx[i];
      CFZP_SYNC();
      constraints(sp, tube, w.xt, w.ct);
      double th_t = 0.0;
      CFZP_LANE_FOR(i, 0, m - 1) th_t += fabs(w.ct[i]);
      th_t = psum(th_t);
      const double ph_t = barrier_obj(sp, w, w.xt, mu);
      bool ok = isfinite(ph_t) && isfinite(th_t) && th_t <= theta_max;
      if (ok) for (int q = 0; q < nfilt; ++q) if (th_t >= filt[q][0] && ph_t >= filt[q][1]) { ok = false; break; }
      f_type = false;
      if (ok) {
        const bool sw = theta <= theta_min && dphi < 0.0 && alpha * pow(-dphi, sp.s_phi) > sp.delta_sw * pow(theta, sp.s_theta);
        if (sw) { f_type = true; ok = ph_t <= phi0 + sp.eta_phi * alpha * dphi; }
        else ok = th_t <= (1.0 - sp.gamma_theta) * theta || ph_t <= phi0 - sp.gamma_phi * theta;
      }
      if (ok) { accepted = true; break; }
      alpha *= 0.5;
    }
    if (!accepted) {
      // No step length passes the filter.  As in cfz_colloc.inl: if the barrier parameter can still fall, the barrier problem at hand is
      // given up -- mu falls, the filter starts afresh, the iterate stays -- once per failure; otherwise status 2.
      if (mu > mu_floor && !mu_forced) {
        mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu))); mu_forced = true; nfilt = 0; filt_mu = mu;
        continue;
      }
      // last resort, as in cfz_colloc.inl: the same iterate again with eight times the perturbation, up to three times
      if (delta_retry < 3 && delta < 1e8) { delta_floor = fmax(8.0 * delta, 1e-4); ++delta_retry; nfilt = 0; continue; }
      status = 2; break;
    }
    mu_forced = false; delta_floor = 0.0; delta_retry = 0;
    if (!f_type) {
      if (nfilt == sp.filter_cap) { for (int q = 1; q < nfilt; ++q) { filt[q - 1][0] = filt[q][0]; filt[q - 1][1] = filt[q][1]; } --nfilt; }
      filt[nfilt][0] = (1.0 - sp.gamma_theta) * theta; filt[nfilt][1] = phi0 - sp.gamma_phi * theta; ++nfilt;
    }
    CFZP_LANE_FOR(i, 0, m - 1) w.nu[i] += alpha * w.dnu[i];
    CFZP_LANE_FOR(i, 0, n - 1) {
      w.x[i] = w.xt[i];
      if (w.xl[i] > -1e300) { const double dl = w.x[i] - w.xl[i]; w.zl[i] = fmin(fmax(w.zl[i] + a_dual * w.dzl[i], mu / (sp.kappa_sigma * dl)), sp.kappa_sigma * mu / dl); }
      if (w.xu[i] < 1e300) { const double du = w.xu[i] - w.x[i]; w.zu[i] = fmin(fmax(w.zu[i] + a_dual * w.dzu[i], mu / (sp.kappa_sigma * du)), sp.kappa_sigma * mu / du); }
    }
    CFZP_SYNC();
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, n - 1) X[i] = w.x[i];  // trajectory and tube slacks
  out_i[0] = iter; out_i[1] = status;
  out_d[0] = objective(sp, w.x); out_d[1] = err0; out_d[2] = mu;
}

}  // namespace cfzp
