// state_ws: the warm-start NLP of the single-vehicle plan (reference confrez/control/vehicle.py:99-231).
//
// T = N (S-1) forward-Euler steps of the kinematic bicycle (:173); initial pose fixed and v0 = delta0 = a0 = w0 = 0
// (:131-138); boxes on x, y, v, delta for k < T, on a, w only if `bounded_input` (:141-167); at every k = N i the
// rear-axle point inside the back cell and the front point (x + wb cos psi, y + wb sin psi) inside the front cell of
// strategy step i, both shrunk by `shrink_tube` (:178-192); optional terminal heading (:194-195); cost sum a^2 + w^2
// (:175-176).  Hundreds of stages, a handful of instances (one per vehicle), solved once: the opposite regime of the
// MPC step.  One instance per workgroup (four wavefronts since round 5; the sweeps on one lane), workspace in global memory (L2-resident), the interior-point iteration of
// oracle/ipm.py with the EXACT Hessian of the Lagrangian and the curvature test  dx'(H + delta I)dx >= kappa |dx|^2  (delta: 0, 1e-4,
// x8 ...).  The Newton system is a stage recursion (Riccati sweep, below): rounds 1-3 solved it as a banded LU with partial pivoting
// (half-bandwidth 40, ~3,000 pivots one after the other: 10 ms per iteration); the sweep is T stages of ~350 operations on one lane.
//
// The same source compiles for the CPU (tests/emu) and is checked iterate for iterate against oracle/plan_nlp.py.
#pragma once
#include <math.h>

#include "cfz_band.inl"  // the pointer types the out-of-line sweeps take, `opaque`

#if defined(__HIPCC__)
#define CFZP_FN __host__ __device__ inline
#else
#define CFZP_FN inline
#endif

// On the GPU all lanes of the workgroup (256 since round 5: four wavefronts) run the solver redundantly (same scalars, same addresses: one memory
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

// workspace (doubles), carved out of one slab by `carve`.  st / fb / vf: per stage, what the Riccati sweep reads (kSt), the feedback it
// leaves for the forward sweep (kFb) and the value function the multipliers are read from (kVf); ce: the tube rows of a checkpoint
// condensed into its pose block; dx2: the sweep's second solution (the terminal-heading column); flag: the sweep's verdict and eta
constexpr int kSt = 33, kFb = 14, kVf = 25;
struct PWork {
  double *x, *xt, *zl, *zu, *nu, *dx, *dnu, *dzl, *dzu, *g, *c, *ct, *xl, *xu, *r1, *hd, *st, *fb, *vf, *ce, *dx2, *flag;
};
CFZP_FN size_t work_doubles(const PSpec &sp) {
  const PDims d = dims(sp);
  return (size_t)d.n * 12 + (size_t)d.m * 4 + (size_t)(sp.T + 1) * (kSt + kFb + kVf + 7) + (size_t)sp.n_chk * 6 + 72;
}
CFZP_FN PWork carve(const PSpec &sp, double *slab) {
  const PDims d = dims(sp);
  PWork w; double *p = slab;
  w.x = p; p += d.n; w.xt = p; p += d.n; w.zl = p; p += d.n; w.zu = p; p += d.n; w.dx = p; p += d.n; w.dzl = p; p += d.n;
  w.dzu = p; p += d.n; w.g = p; p += d.n; w.xl = p; p += d.n; w.xu = p; p += d.n; w.r1 = p; p += d.n; w.hd = p; p += d.n;
  w.nu = p; p += d.m; w.dnu = p; p += d.m; w.c = p; p += d.m; w.ct = p; p += d.m;
  w.st = p; p += (size_t)(sp.T + 1) * kSt; w.fb = p; p += (size_t)(sp.T + 1) * kFb; w.vf = p; p += (size_t)(sp.T + 1) * kVf;
  w.dx2 = p; p += (size_t)(sp.T + 1) * 7; w.ce = p; p += (size_t)sp.n_chk * 6; w.flag = p;
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

// ---- the Newton system as a stage recursion ------------------------------------------------------------------------------------
// [[W + Sigma + (delta + reg) I, J'], [J, -delta_c on the heading row]] (dx, dnu) = -(r1, c)  is the optimality system of a linear-
// quadratic control problem over the stages: with the tube slacks eliminated into the pose block of their checkpoint (exactly: their rows
// carry no delta_c) the unknowns of stage k are dz_k (5) and du_k (2), tied by  dz_{k+1} = A_k dz_k + B du_k + c_k,  A_k = I + dt f_z
// (six entries off the diagonal), B = dt [e_v e_delta].  Stage 0 is pinned by the initial rows.  The value function  V_k(dz) =
// dz' P_k dz / 2 + p_k' dz  runs backwards from k = T to 1 (one lane, ~200 dependent FP64 operations per stage); the terminal-heading
// row is bordered: the sweep carries a second vector p2 for the right-hand side e_psi_T, and  eta = (dpsi_T(1) + c_f) / (dpsi_T(2) +
// delta_c)  combines the two forward solutions.  Multipliers: dnu_k = P_{k+1} dz_{k+1} + p_{k+1}; the initial rows' from the stationarity
// of stage 0; the tube rows' from their slacks.  No pivoting: where the reduced Hessian  R + B' P B  of a stage is not positive the
// curvature test  dx'(H + delta I)dx >= kappa |dx|^2  would not pass either, and the sweep reports failure for a singular one.
//
// st[k] (kSt): a02 a03 a12 a13 a23 a24 | q0..q4 q23 q34 | r0 r1 | gz[5] gu[2] | c[5] | for the matrix-core sweep: q0 q1 q2 with the tube block of
// a checkpoint stage added, E01 E02 E12 (zero elsewhere);   fb[k] (kFb): K (2 x 5) kk1[2] kk2[2];
// vf[k] (kVf): P (upper triangle by rows, 15) p1[5] p2[5];   ce[i]: E00 E01 E02 E11 E12 E22 of checkpoint i
// FAST: st and fb live in the kernel's dynamic LDS ((kSt + kFb)(T + 1) doubles: 113 KB at T = 300); otherwise (plans too long for the
// LDS, batches of several plans per CU; the CPU build) in the workspace.  The inlined code names the dynamic LDS directly; the sweeps,
// functions of their own on the GPU (their registers are allocated apart from the solver's many live scalars), take LDS-typed POINTERS:
// a non-inlined function that NAMES LDS is broken on this toolchain (gfx950, ROCm 7.2; cfz_colloc.inl's header, cfz_band.inl's `opaque`).
#if defined(__HIPCC__)
#define CFZP_SWEEP __host__ __device__ __attribute__((noinline))
#else
#define CFZP_SWEEP inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define CFZP_FIRST (threadIdx.x == 0)
#define CFZP_STAGE_PTRS(w, T) extern __shared__ double cfzp_lds[]; double *const st_ = FAST ? cfzp_lds : (w).st; double *const fb_ = FAST ? cfzp_lds + (size_t)kSt * ((T) + 1) : (w).fb
typedef cfzb::lds_f64 fast_f64;
#define CFZP_OPAQUE(p) cfzb::opaque(p)  // cfz_band.inl: the sweeps must not learn WHICH LDS array they work on
#else
#define CFZP_OPAQUE(p) (p)
#define CFZP_FIRST true
#define CFZP_STAGE_PTRS(w, T) double *const st_ = (w).st; double *const fb_ = (w).fb
typedef double fast_f64;
#endif

template <bool FAST>
CFZP_FN void riccati_prepare(const PSpec &sp, const double *tube, const PWork &w, const double *sig, double delta) {
  const PDims d = dims(sp);
  CFZP_STAGE_PTRS(w, sp.T); (void)fb_;
  const double *X = w.x, *nu = w.nu;
  const double shift = delta + sp.reg_primal, dt = sp.dt;
  CFZP_LANE_FOR(k, 0, sp.T) {
    double *s = st_ + (size_t)kSt * k;
    const int b = 7 * k;
    if (k < sp.T) {
      const double *z = X + b, *l = nu + 7 + 5 * k;
      const double cs = cos(z[2]), sn = sin(z[2]), tn = tan(z[4]), sec2 = 1.0 + tn * tn, v = z[3];
      const double l0 = l[0] * dt, l1 = l[1] * dt, l2 = l[2] * dt;
      s[0] = dt * (-v * sn); s[1] = dt * cs; s[2] = dt * (v * cs); s[3] = dt * sn; s[4] = dt * tn / sp.wb; s[5] = dt * v / sp.wb * sec2;
      s[6] = sig[b] + shift; s[7] = sig[b + 1] + shift; s[9] = sig[b + 3] + shift;
      s[8] = sig[b + 2] + shift + l0 * (-v * cs) + l1 * (-v * sn); s[10] = sig[b + 4] + shift + l2 * 2.0 * v * tn * sec2 / sp.wb;
      s[11] = l0 * (-sn) + l1 * cs; s[12] = l2 * sec2 / sp.wb;
      s[13] = 2.0 + sig[b + 5] + shift; s[14] = 2.0 + sig[b + 6] + shift;
      for (int i = 0; i < 7; ++i) s[15 + i] = w.r1[b + i];
      for (int i = 0; i < 5; ++i) s[22 + i] = w.c[7 + 5 * k + i];
    } else {
      for (int i = 0; i < kSt; ++i) s[i] = 0.0;
      for (int i = 0; i < 5; ++i) { s[6 + i] = sig[b + i] + shift; s[15 + i] = w.r1[b + i]; }
    }
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(i, 0, sp.n_chk - 1) {  // tube rows of checkpoint i -> pose block E, pose gradient, heading curvature
    const int k = chk_stage(sp, i), b = 7 * k, r = d.r0 + 8 * i, sl = d.s0 + 8 * i;
    double *s = st_ + (size_t)kSt * k, *E = w.ce + 6 * i;
    const double cs = cos(X[b + 2]), sn = sin(X[b + 2]);
    const double *cb = cell(tube, i, 0), *cf = cell(tube, i, 1);
    double e[6] = {0, 0, 0, 0, 0, 0}, gp[3] = {0, 0, 0}, curv = 0.0;
    for (int q = 0; q < 8; ++q) {
      const double *cc = q < 4 ? cb : cf; const int qq = q & 3;
      const double G[3] = {cc[2 * qq], cc[2 * qq + 1], q < 4 ? 0.0 : sp.wb * (-cc[2 * qq] * sn + cc[2 * qq + 1] * cs)};
      const double S = sig[sl + q] + shift, t = S * w.c[r + q] - w.r1[sl + q];
      e[0] += S * G[0] * G[0]; e[1] += S * G[0] * G[1]; e[2] += S * G[0] * G[2]; e[3] += S * G[1] * G[1]; e[4] += S * G[1] * G[2]; e[5] += S * G[2] * G[2];
      for (int j = 0; j < 3; ++j) gp[j] += G[j] * t;
      if (q >= 4) curv += nu[r + q] * sp.wb * (-cc[2 * qq] * cs - cc[2 * qq + 1] * sn);
    }
    for (int j = 0; j < 6; ++j) E[j] = e[j];
    for (int j = 0; j < 3; ++j) s[15 + j] += gp[j];
    s[8] += curv;
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(k, 0, sp.T) {  // the pose block of the stage's Hessian as the matrix-core sweep reads it: with the tube block where there is one
    double *s = st_ + (size_t)kSt * k;
    const bool chk = k >= sp.N && k % sp.N == 0;
    const double *E = w.ce + 6 * (chk ? k / sp.N - 1 : 0);
    s[27] = s[6] + (chk ? E[0] : 0.0); s[28] = s[7] + (chk ? E[3] : 0.0); s[29] = s[8] + (chk ? E[5] : 0.0);
    s[30] = chk ? E[1] : 0.0; s[31] = chk ? E[2] : 0.0; s[32] = chk ? E[4] : 0.0;
  }
  CFZP_SYNC();
}

// backward sweep (ONE lane calls it); 0 = ok, 1 = a stage's reduced Hessian is singular or not finite.
// P in symmetric storage (P00 P01 .. P04 P11 .. P44), the products a column at a time: nothing here may spill (a spilled value costs a
// round trip to memory in a chain of dependent operations).
template <class SP>
CFZP_SWEEP int riccati_backward(const PSpec &sp, const PWork &w, SP st_, SP fb_) {
  const int T = sp.T; const double dt = sp.dt, dt2 = dt * dt;
  double P00, P01, P02, P03, P04, P11, P12, P13, P14, P22, P23, P24, P33, P34, P44;
  double p1[5], p2[5] = {0.0, 0.0, sp.has_final ? -1.0 : 0.0, 0.0, 0.0};
  {
    const auto *s = st_ + (size_t)kSt * T; const double *E = w.ce + 6 * (sp.n_chk - 1);
    P00 = s[6] + E[0]; P01 = E[1]; P02 = E[2]; P03 = 0.0; P04 = 0.0; P11 = s[7] + E[3]; P12 = E[4]; P13 = 0.0; P14 = 0.0;
    P22 = s[8] + E[5]; P23 = 0.0; P24 = 0.0; P33 = s[9]; P34 = 0.0; P44 = s[10];
    for (int i = 0; i < 5; ++i) p1[i] = s[15 + i];
  }
#define CFZP_STORE_VF(k)                                                                                                            \
  {                                                                                                                                 \
    double *v = w.vf + (size_t)kVf * (k);                                                                                           \
    v[0] = P00; v[1] = P01; v[2] = P02; v[3] = P03; v[4] = P04; v[5] = P11; v[6] = P12; v[7] = P13; v[8] = P14; v[9] = P22;         \
    v[10] = P23; v[11] = P24; v[12] = P33; v[13] = P34; v[14] = P44;                                                                \
    for (int i = 0; i < 5; ++i) { v[15 + i] = p1[i]; v[20 + i] = p2[i]; }                                                           \
  }
  CFZP_STORE_VF(T)
  for (int k = T - 1; k >= 1; --k) {
    const auto *s = st_ + (size_t)kSt * k;
    const double a02 = s[0], a03 = s[1], a12 = s[2], a13 = s[3], a23 = s[4], a24 = s[5];
    // h = p + P c;  g = A' h
    const double c0 = s[22], c1 = s[23], c2 = s[24], c3 = s[25], c4 = s[26];
    const double h0 = p1[0] + P00 * c0 + P01 * c1 + P02 * c2 + P03 * c3 + P04 * c4, h1 = p1[1] + P01 * c0 + P11 * c1 + P12 * c2 + P13 * c3 + P14 * c4,
                 h2 = p1[2] + P02 * c0 + P12 * c1 + P22 * c2 + P23 * c3 + P24 * c4, h3 = p1[3] + P03 * c0 + P13 * c1 + P23 * c2 + P33 * c3 + P34 * c4,
                 h4 = p1[4] + P04 * c0 + P14 * c1 + P24 * c2 + P34 * c3 + P44 * c4;
    // columns of P A (A = I + six entries):  col0, col1 are P's;  col2 = P2 + a02 P0 + a12 P1;  col3 = P3 + a03 P0 + a13 P1 + a23 P2;  col4 = P4 + a24 P2
    const double B02 = P02 + a02 * P00 + a12 * P01, B12 = P12 + a02 * P01 + a12 * P11, B22 = P22 + a02 * P02 + a12 * P12, B32 = P23 + a02 * P03 + a12 * P13,
                 B42 = P24 + a02 * P04 + a12 * P14;
    const double B03 = P03 + a03 * P00 + a13 * P01 + a23 * P02, B13 = P13 + a03 * P01 + a13 * P11 + a23 * P12, B23 = P23 + a03 * P02 + a13 * P12 + a23 * P22,
                 B33 = P33 + a03 * P03 + a13 * P13 + a23 * P23, B43 = P34 + a03 * P04 + a13 * P14 + a23 * P24;
    const double B04 = P04 + a24 * P02, B14 = P14 + a24 * P12, B24 = P24 + a24 * P22, B34 = P34 + a24 * P23, B44 = P44 + a24 * P24;
    // reduced Hessian of the stage and its inverse
    const double R00 = s[13] + dt2 * P33, R01 = dt2 * P34, R11 = s[14] + dt2 * P44;
    const double det = R00 * R11 - R01 * R01;
    if (!(det != 0.0) || !isfinite(det)) return 1;
    const double idet = 1.0 / det, i00 = R11 * idet, i01 = -R01 * idet, i11 = R00 * idet;
    // Rux = dt (P A)[3:5, :]  (row 3 of P A: P03 P13 B32 B33 B34; row 4: P04 P14 B42 B43 B44)
    const double x0[5] = {dt * P03, dt * P13, dt * B32, dt * B33, dt * B34}, x1[5] = {dt * P04, dt * P14, dt * B42, dt * B43, dt * B44};
    double K0[5], K1[5];
    for (int j = 0; j < 5; ++j) { K0[j] = -(i00 * x0[j] + i01 * x1[j]); K1[j] = -(i01 * x0[j] + i11 * x1[j]); }
    const double ku0 = s[20] + dt * h3, ku1 = s[21] + dt * h4, kv0 = dt * p2[3], kv1 = dt * p2[4];
    const double kk0 = -(i00 * ku0 + i01 * ku1), kk1 = -(i01 * ku0 + i11 * ku1), kl0 = -(i00 * kv0 + i01 * kv1), kl1 = -(i01 * kv0 + i11 * kv1);
    {
      auto *f = fb_ + (size_t)kFb * k;
      for (int j = 0; j < 5; ++j) { f[j] = K0[j]; f[5 + j] = K1[j]; }
      f[10] = kk0; f[11] = kk1; f[12] = kl0; f[13] = kl1;
    }
    // p = gz + A' h + Rux' kk
    const double q0 = p2[0], q1 = p2[1], q2 = p2[2], q3 = p2[3], q4 = p2[4];
    p1[0] = s[15] + h0 + x0[0] * kk0 + x1[0] * kk1;
    p1[1] = s[16] + h1 + x0[1] * kk0 + x1[1] * kk1;
    p1[2] = s[17] + h2 + a02 * h0 + a12 * h1 + x0[2] * kk0 + x1[2] * kk1;
    p1[3] = s[18] + h3 + a03 * h0 + a13 * h1 + a23 * h2 + x0[3] * kk0 + x1[3] * kk1;
    p1[4] = s[19] + h4 + a24 * h2 + x0[4] * kk0 + x1[4] * kk1;
    p2[0] = q0 + x0[0] * kl0 + x1[0] * kl1;
    p2[1] = q1 + x0[1] * kl0 + x1[1] * kl1;
    p2[2] = q2 + a02 * q0 + a12 * q1 + x0[2] * kl0 + x1[2] * kl1;
    p2[3] = q3 + a03 * q0 + a13 * q1 + a23 * q2 + x0[3] * kl0 + x1[3] * kl1;
    p2[4] = q4 + a24 * q2 + x0[4] * kl0 + x1[4] * kl1;
    // P <- Q + A' (P A) + Rux' K, upper triangle:  (A' M)_ij = M_ij + (column i of A above the diagonal) . M_:j
    const double N00 = P00 + x0[0] * K0[0] + x1[0] * K1[0], N01 = P01 + x0[0] * K0[1] + x1[0] * K1[1], N02 = B02 + x0[0] * K0[2] + x1[0] * K1[2],
                 N03 = B03 + x0[0] * K0[3] + x1[0] * K1[3], N04 = B04 + x0[0] * K0[4] + x1[0] * K1[4];
    const double N11 = P11 + x0[1] * K0[1] + x1[1] * K1[1], N12 = B12 + x0[1] * K0[2] + x1[1] * K1[2], N13 = B13 + x0[1] * K0[3] + x1[1] * K1[3],
                 N14 = B14 + x0[1] * K0[4] + x1[1] * K1[4];
    const double N22 = B22 + a02 * B02 + a12 * B12 + x0[2] * K0[2] + x1[2] * K1[2], N23 = B23 + a02 * B03 + a12 * B13 + x0[2] * K0[3] + x1[2] * K1[3],
                 N24 = B24 + a02 * B04 + a12 * B14 + x0[2] * K0[4] + x1[2] * K1[4];
    const double N33 = B33 + a03 * B03 + a13 * B13 + a23 * B23 + x0[3] * K0[3] + x1[3] * K1[3], N34 = B34 + a03 * B04 + a13 * B14 + a23 * B24 + x0[3] * K0[4] + x1[3] * K1[4];
    const double N44 = B44 + a24 * B24 + x0[4] * K0[4] + x1[4] * K1[4];
    P00 = N00 + s[6]; P01 = N01; P02 = N02; P03 = N03; P04 = N04; P11 = N11 + s[7]; P12 = N12; P13 = N13; P14 = N14;
    P22 = N22 + s[8]; P23 = N23 + s[11]; P24 = N24; P33 = N33 + s[9]; P34 = N34 + s[12]; P44 = N44 + s[10];
    if (k % sp.N == 0) {
      const double *E = w.ce + 6 * (k / sp.N - 1);
      P00 += E[0]; P01 += E[1]; P02 += E[2]; P11 += E[3]; P12 += E[4]; P22 += E[5];
    }
    CFZP_STORE_VF(k)
  }
#undef CFZP_STORE_VF
  return 0;
}

// The same backward sweep on the matrix cores (round 5; the MPC kernel's sweep, cfz_solver.inl riccati_backward_mfma, is its model: 1,850 ->
// 1,130 cycles per stage there).  One wavefront, all 64 lanes.  Homogeneous coordinates [z (5), 1, e] with e the coordinate the border vector p2
// rides on, variables [z, 1, e, -, u0, u1] (index 7 unused: the u rows stand at 8, 9 and, crossed, at 12, 13, so that the lane groups 0 and 1
// hold M's rows of u in their own registers):  [z+; 1; e] = T [z; 1; e; u],  Pt = [[P p1 p2], [p1' . .], [p2' . .]]  (the entries marked . are
// constants of the value function that feed nothing),  M = T' Pt T + Ht,  Pt <- M_kk - M_ke M_ee^-1 M_ek,  [K kk kl] = -M_ee^-1 M_ek -- five dependent
// v_mfma_f64_16x16x4_f64 per stage: Y = Pt T (Pt's accumulator is the first operand as it stands: it is symmetric), M = Tt' Y onto Ht (Y's
// accumulator is the second operand as it stands), Pt = M - U V.  Operands are read from the stage data with one load each (a constant reads
// a harmless word with mask 0).  Same recursion as riccati_backward above, the sums in another order.
struct PlEnt { int off; double c; };  // off >= 0: word off of the stage's data, else the constant c
CFZP_FN PlEnt pl_T(int r, int j, double dt) {  // T[r][j]: r = 0..6 (z+, 1, e), j a variable
  PlEnt e = {-1, 0.0};
  if (r < 0 || r > 6 || j < 0 || j == 7 || j > 9) return e;
  if (r == 5) { e.c = j == 5 ? 1.0 : 0.0; return e; }
  if (r == 6) { e.c = j == 6 ? 1.0 : 0.0; return e; }
  if (j < 5) {
    e.c = r == j ? 1.0 : 0.0;
    if (r == 0 && j == 2) e.off = 0; else if (r == 0 && j == 3) e.off = 1; else if (r == 1 && j == 2) e.off = 2; else if (r == 1 && j == 3) e.off = 3;
    else if (r == 2 && j == 3) e.off = 4; else if (r == 2 && j == 4) e.off = 5;
    return e;
  }
  if (j == 5) { e.off = 22 + r; return e; }
  if (j == 8) e.c = r == 3 ? dt : 0.0;
  if (j == 9) e.c = r == 4 ? dt : 0.0;
  return e;
}
CFZP_FN PlEnt pl_H(int a, int b) {  // Ht[a][b] over the variables (gradient in row / column 5)
  PlEnt e = {-1, 0.0};
  if (a < 0 || b < 0 || a == 7 || b == 7 || a > 9 || b > 9) return e;
  if (a > b) { const int t = a; a = b; b = t; }
  if (a == b) { if (a < 3) e.off = 27 + a; else if (a < 5) e.off = 6 + a; else if (a == 8) e.off = 13; else if (a == 9) e.off = 14; return e; }
  if (a == 0 && b == 1) e.off = 30; else if (a == 0 && b == 2) e.off = 31; else if (a == 1 && b == 2) e.off = 32;
  else if (a == 2 && b == 3) e.off = 11; else if (a == 3 && b == 4) e.off = 12;
  else if (b == 5 && a < 5) e.off = 15 + a;
  else if (a == 5 && b == 8) e.off = 20; else if (a == 5 && b == 9) e.off = 21;
  return e;
}
// the sweep in homogeneous coordinates as plain loops (the CPU build with -DCFZP_DENSE_SWEEP: the check of the tables above and of the
// recursion the matrix-core sweep runs, against riccati_backward)
template <class SP>
CFZP_FN int riccati_backward_dense(const PSpec &sp, const PWork &w, SP st_, SP fb_) {
  const int T = sp.T, var[9] = {0, 1, 2, 3, 4, 5, 6, 8, 9};
  double P[7][7];
  auto val = [&](const PlEnt &e, int k) -> double { return e.off >= 0 ? (double)st_[(size_t)kSt * k + e.off] : e.c; };
  for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) P[i][j] = val(pl_H(i, j), T);
  if (sp.has_final) { P[2][6] = -1.0; P[6][2] = -1.0; }
  auto store_vf = [&](int k) {
    double *v = w.vf + (size_t)kVf * k; int o = 0;
    for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j) v[o++] = P[i][j];
    for (int i = 0; i < 5; ++i) { v[15 + i] = P[5][i]; v[20 + i] = P[6][i]; }
  };
  store_vf(T);
  for (int k = T - 1; k >= 1; --k) {
    double Tm[7][9], H[9][9], Y[7][9], M[9][9];
    for (int r = 0; r < 7; ++r) for (int j = 0; j < 9; ++j) Tm[r][j] = val(pl_T(r, var[j], sp.dt), k);
    for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) H[a][b] = val(pl_H(var[a], var[b]), k);
#if defined(CFZP_DENSE_TRANSPOSED)  // what the matrix-core sweep does: Pt's accumulator as the first operand "as it stands" = its transpose
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 9; ++j) { double s_ = 0.0; for (int q = 0; q < 7; ++q) s_ += P[q][i] * Tm[q][j]; Y[i][j] = s_; }
#else
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 9; ++j) { double s_ = 0.0; for (int q = 0; q < 7; ++q) s_ += P[i][q] * Tm[q][j]; Y[i][j] = s_; }
#endif
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) { double s_ = H[i][j]; for (int q = 0; q < 7; ++q) s_ += Tm[q][i] * Y[q][j]; M[i][j] = s_; }
    const double det = M[7][7] * M[8][8] - M[7][8] * M[8][7];
    if (!(det != 0.0) || !isfinite(det)) return 1;
    const double idet = 1.0 / det, i00 = M[8][8] * idet, i01 = -M[7][8] * idet, i10 = -M[8][7] * idet, i11 = M[7][7] * idet;
    double V[2][7];
    for (int j = 0; j < 7; ++j) { V[0][j] = i00 * M[7][j] + i01 * M[8][j]; V[1][j] = i10 * M[7][j] + i11 * M[8][j]; }
    auto *f = fb_ + (size_t)kFb * k;
    for (int g = 0; g < 2; ++g) { for (int j = 0; j < 5; ++j) f[5 * g + j] = -V[g][j]; f[10 + g] = -V[g][5]; f[12 + g] = -V[g][6]; }
#if defined(CFZP_DENSE_TRANSPOSED)
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) P[i][j] = M[i][j] - M[7][i] * V[0][j] - M[8][i] * V[1][j];
    for (int i = 0; i < 7; ++i) for (int j = i + 1; j < 7; ++j) { const double m_ = 0.5 * (P[i][j] + P[j][i]); P[i][j] = m_; P[j][i] = m_; }  // (see riccati_backward_mfma)
#else
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) P[i][j] = M[i][j] - M[i][7] * V[0][j] - M[i][8] * V[1][j];
#endif
    store_vf(k);
  }
  return 0;
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef double pl_v4d __attribute__((ext_vector_type(4)));
struct PlFetch { int addr; double mask, c; };
__device__ __forceinline__ PlFetch pl_fetch(const PlEnt &e, int k) { PlFetch f; f.addr = (e.off >= 0 ? e.off : 6) + k * kSt; f.mask = e.off >= 0 ? 1.0 : 0.0; f.c = e.off >= 0 ? 0.0 : e.c; return f; }
template <class SP> __device__ __forceinline__ double pl_get(SP st_, const PlFetch &f) { return fma((double)st_[f.addr], f.mask, f.c); }
__device__ __forceinline__ double pl_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double pl_perm(double v, int byte_lane) {  // v of the lane whose index * 4 is byte_lane
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_lane, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_lane, __double2loint(v)));
}
template <class SP>
__device__ __attribute__((noinline)) int riccati_backward_mfma(int T, double dt, int has_final, double *vf_, SP st_, SP fb_) {
  const int lane = threadIdx.x & 63, lo = lane & 15, g = lane >> 4;
  cfzb::glb_f64 *vf = (cfzb::glb_f64 *)vf_;
  // row i of Tt' / Ht stands for variable: 0..6 themselves, 8 -> u0, 9 -> u1, 12 -> u1, 13 -> u0, the others are zero rows
  const int vlo = lo < 7 ? lo : (lo == 8 ? 8 : lo == 9 ? 9 : lo == 12 ? 9 : lo == 13 ? 8 : -1);
  int vrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int row = g + 4 * r; vrow[r] = row < 7 ? row : (row == 8 ? 8 : row == 9 ? 9 : row == 12 ? 9 : row == 13 ? 8 : -1); }
  // where this lane's entries of the value function go: register 0 = row g, register 1 = row 4 + g of Pt, column lo
  const int v0 = (lo < 5 && lo >= g) ? (g * (11 - g)) / 2 + (lo - g) : -1;
  const int v1 = lo < 5 ? (g == 0 ? (lo == 4 ? 14 : -1) : g == 1 ? 15 + lo : g == 2 ? 20 + lo : -1) : -1;
  const int kout = lo < 5 ? 5 * g + lo : (lo == 5 ? 10 + g : 12 + g);  // this lane's gain (groups 0, 1; columns 0..6)
  pl_v4d P;
  {
    PlFetch fH[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) fH[r] = pl_fetch(pl_H(vrow[r], lo), T);
    P = pl_v4d{pl_get(st_, fH[0]), pl_get(st_, fH[1]), pl_get(st_, fH[2]), pl_get(st_, fH[3])};
    if (has_final) { if (g == 2 && lo == 6) P[0] = -1.0; if (g == 2 && lo == 2) P[1] = -1.0; }  // p2 = -e_psi: Pt[2][6] = Pt[6][2]
  }
  if (v0 >= 0) vf[(size_t)kVf * T + v0] = P[0];
  if (v1 >= 0) vf[(size_t)kVf * T + v1] = P[1];
  int k = T - 1;
  PlFetch fB[2], fA[2], fH[4];
#pragma unroll
  for (int s_ = 0; s_ < 2; ++s_) {
    fB[s_] = pl_fetch(pl_T(g + 4 * s_, (lo < 7 || lo == 8 || lo == 9) ? lo : -1, dt), k);
    fA[s_] = pl_fetch(pl_T(g + 4 * s_, vlo, dt), k);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) fH[r] = pl_fetch(pl_H(vrow[r], lo), k);
  double b0 = pl_get(st_, fB[0]), b1 = pl_get(st_, fB[1]), a0 = pl_get(st_, fA[0]), a1 = pl_get(st_, fA[1]);
  pl_v4d H = {pl_get(st_, fH[0]), pl_get(st_, fH[1]), pl_get(st_, fH[2]), pl_get(st_, fH[3])};
  for (; k >= 1; --k) {
    pl_v4d Y = {0.0, 0.0, 0.0, 0.0};
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[0], b0, Y, 0, 0, 0);
    Y = __builtin_amdgcn_mfma_f64_16x16x4f64(P[1], b1, Y, 0, 0, 0);
    pl_v4d M = H;
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Y[0], M, 0, 0, 0);
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Y[1], M, 0, 0, 0);
    // the next stage's operands: requested now, used after this stage's last matrix instruction (the last pass re-reads stage 1: harmless)
    if (k > 1) {
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) { fB[s_].addr -= kSt; fA[s_].addr -= kSt; }
#pragma unroll
      for (int r = 0; r < 4; ++r) fH[r].addr -= kSt;
    }
    const double nb0 = pl_get(st_, fB[0]), nb1 = pl_get(st_, fB[1]), na0 = pl_get(st_, fA[0]), na1 = pl_get(st_, fA[1]);
    const pl_v4d nH = {pl_get(st_, fH[0]), pl_get(st_, fH[1]), pl_get(st_, fH[2]), pl_get(st_, fH[3])};
    // M_ee: rows 8, 9 = groups 0, 1, register 2; columns 8, 9
    const double m88 = pl_lane(M[2], 8), m89 = pl_lane(M[2], 9), m98 = pl_lane(M[2], 24), m99 = pl_lane(M[2], 25);
    const double det = fma(m88, m99, -(m89 * m98));
    if (!(det != 0.0) || !isfinite(det)) return 1;
    const double idet = 1.0 / det;
    const double i00 = m99 * idet, i01 = -m89 * idet, i10 = -m98 * idet, i11 = m88 * idet;
    const double m6 = g == 0 ? M[2] : M[3], m7 = g == 0 ? M[3] : M[2];  // rows of u0, u1 in this group's registers (8 / 12, 13 / 9)
    const double ia = g == 0 ? i00 : i10, ib = g == 0 ? i01 : i11;
    const double Vf = fma(ia, m6, ib * m7), V = g < 2 ? Vf : 0.0;
    const double U = g < 2 ? -M[2] : 0.0;
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(U, V, M, 0, 0, 0);
    // Pt <- (Pt + Pt') / 2 on its rows and columns 0..7.  The accumulator serves as the next stage's first operand "as it stands", i.e. as
    // its TRANSPOSE: what the two triangles differ by in the last bits is an antisymmetric part that this use does not damp -- over 300
    // stages it grows until the step is wrong in the third digit (vehicle 0's plan: 211 iterations instead of 22; the CPU build's plain
    // loops reproduce it with -DCFZP_DENSE_TRANSPOSED).  Element (lo, g + 4 r) sits in lane 16 (lo & 3) + g + 4 r, register lo >> 2.
    {
      const int l0 = (((lo & 3) << 4) + g) << 2, l1 = l0 + 16;
      const double t00 = pl_perm(M[0], l0), t01 = pl_perm(M[1], l0), t10 = pl_perm(M[0], l1), t11 = pl_perm(M[1], l1);
      const double x0 = lo < 4 ? t00 : t01, x1 = lo < 4 ? t10 : t11;
      if (lo < 8) { M[0] = 0.5 * (M[0] + x0); M[1] = 0.5 * (M[1] + x1); }
    }
    if (g < 2 && lo < 7) fb_[(size_t)kFb * k + kout] = -V;
    if (v0 >= 0) vf[(size_t)kVf * k + v0] = M[0];
    if (v1 >= 0) vf[(size_t)kVf * k + v1] = M[1];
    P = M; b0 = nb0; b1 = nb1; a0 = na0; a1 = na1; H = nH;
  }
  return 0;
}
#endif

// forward sweep (ONE lane): the two solutions' dz, du -> w.dx (right-hand side -(r1, c)) and w.dx2 (e_psi_T)
template <class SP>
CFZP_SWEEP void riccati_forward(const PSpec &sp, const PWork &w, SP st_, SP fb_) {
  const int T = sp.T; const double dt = sp.dt;
  double z1[5], z2[5] = {0, 0, 0, 0, 0}, u1[2], u2[2] = {0, 0};
  for (int i = 0; i < 5; ++i) z1[i] = -w.c[i];
  u1[0] = -w.c[5]; u1[1] = -w.c[6];
  for (int k = 0; k < T; ++k) {
    const auto *s = st_ + (size_t)kSt * k;
    if (k > 0) {
      const auto *f = fb_ + (size_t)kFb * k;
      for (int q = 0; q < 2; ++q) {
        double a = f[10 + q], b = f[12 + q];
        for (int j = 0; j < 5; ++j) { a += f[5 * q + j] * z1[j]; b += f[5 * q + j] * z2[j]; }
        u1[q] = a; u2[q] = b;
      }
    }
    for (int i = 0; i < 5; ++i) { w.dx[7 * k + i] = z1[i]; w.dx2[7 * k + i] = z2[i]; }
    w.dx[7 * k + 5] = u1[0]; w.dx[7 * k + 6] = u1[1]; w.dx2[7 * k + 5] = u2[0]; w.dx2[7 * k + 6] = u2[1];
    const double n1[5] = {z1[0] + s[0] * z1[2] + s[1] * z1[3] + s[22], z1[1] + s[2] * z1[2] + s[3] * z1[3] + s[23], z1[2] + s[4] * z1[3] + s[5] * z1[4] + s[24],
                          z1[3] + dt * u1[0] + s[25], z1[4] + dt * u1[1] + s[26]};
    const double n2[5] = {z2[0] + s[0] * z2[2] + s[1] * z2[3], z2[1] + s[2] * z2[2] + s[3] * z2[3], z2[2] + s[4] * z2[3] + s[5] * z2[4], z2[3] + dt * u2[0],
                          z2[4] + dt * u2[1]};
    for (int i = 0; i < 5; ++i) { z1[i] = n1[i]; z2[i] = n2[i]; }
  }
  for (int i = 0; i < 5; ++i) { w.dx[7 * T + i] = z1[i]; w.dx2[7 * T + i] = z2[i]; }
}

// The Newton step into w.dx, w.dnu; 0 = ok.  Ends with a barrier.
template <bool FAST>
CFZP_FN int newton_step(const PSpec &sp, const double *tube, const PWork &w, const double *sig, double delta) {
  const PDims d = dims(sp);
  CFZP_STAGE_PTRS(w, sp.T);
  const int T = sp.T; const double dt = sp.dt;
  riccati_prepare<FAST>(sp, tube, w, sig, delta);
#if defined(__HIP_DEVICE_COMPILE__)
  if (threadIdx.x < 64) {  // the backward sweep on the matrix cores: the first wavefront, all its lanes
    int fail;
    if (FAST) fail = riccati_backward_mfma(T, dt, sp.has_final, w.vf, CFZP_OPAQUE((fast_f64 *)st_), CFZP_OPAQUE((fast_f64 *)fb_));
    else fail = riccati_backward_mfma(T, dt, sp.has_final, w.vf, (cfzb::glb_f64 *)st_, (cfzb::glb_f64 *)fb_);
    if (threadIdx.x == 0) w.flag[0] = (double)fail;
  }
  __syncthreads();
#endif
  if (CFZP_FIRST) {
    int fail;
#if defined(__HIP_DEVICE_COMPILE__)
    fail = w.flag[0] != 0.0;
    if (!fail) {
      if (FAST) riccati_forward(sp, w, CFZP_OPAQUE((fast_f64 *)st_), CFZP_OPAQUE((fast_f64 *)fb_));
      else riccati_forward(sp, w, st_, fb_);
    }
#elif defined(CFZP_DENSE_SWEEP)
    fail = riccati_backward_dense(sp, w, st_, fb_);
    if (!fail) riccati_forward(sp, w, st_, fb_);
#else
    fail = riccati_backward(sp, w, st_, fb_);
    if (!fail) riccati_forward(sp, w, st_, fb_);
#endif
    double eta = 0.0;
    if (!fail && sp.has_final) eta = (w.dx[7 * T + 2] + w.c[d.m - 1]) / (w.dx2[7 * T + 2] + sp.reg_dual);
    w.flag[0] = (double)fail; w.flag[1] = eta;
  }
  CFZP_SYNC();
  if (w.flag[0] != 0.0) return 1;
  const double eta = w.flag[1];
  if (sp.has_final) {
    CFZP_LANE_FOR(i, 0, d.s0 - 1) w.dx[i] -= eta * w.dx2[i];
    CFZP_SYNC();
  }
  CFZP_LANE_FOR(k, 0, T - 1) {  // multipliers of the Euler rows; stage 0 also the initial rows'
    const double *v = w.vf + (size_t)kVf * (k + 1), *zn = w.dx + 7 * (k + 1);
    double Pm[5][5]; int o = 0;
    for (int i = 0; i < 5; ++i) for (int j = i; j < 5; ++j) { Pm[i][j] = v[o]; Pm[j][i] = v[o]; ++o; }
    double l[5];
    for (int i = 0; i < 5; ++i) { double t = v[15 + i] - eta * v[20 + i]; for (int j = 0; j < 5; ++j) t += Pm[i][j] * zn[j]; l[i] = t; w.dnu[7 + 5 * k + i] = t; }
    if (k == 0) {
      const double *s = st_, *z = w.dx;
      const double At[5] = {l[0], l[1], l[2] + s[0] * l[0] + s[2] * l[1], l[3] + s[1] * l[0] + s[3] * l[1] + s[4] * l[2], l[4] + s[5] * l[2]};
      const double Qz[5] = {s[6] * z[0], s[7] * z[1], s[8] * z[2] + s[11] * z[3], s[9] * z[3] + s[11] * z[2] + s[12] * z[4], s[10] * z[4] + s[12] * z[3]};
      for (int i = 0; i < 5; ++i) w.dnu[i] = -(Qz[i] + s[15 + i] + At[i]);
      w.dnu[5] = -(s[13] * z[5] + s[20] + dt * l[3]); w.dnu[6] = -(s[14] * z[6] + s[21] + dt * l[4]);
    }
  }
  CFZP_LANE_FOR(i, 0, sp.n_chk - 1) {  // slacks and multipliers of the tube rows
    const int b = 7 * chk_stage(sp, i), r = d.r0 + 8 * i, sl = d.s0 + 8 * i;
    const double cs = cos(w.x[b + 2]), sn = sin(w.x[b + 2]);
    const double *cb = cell(tube, i, 0), *cf = cell(tube, i, 1);
    const double shift = delta + sp.reg_primal;
    for (int q = 0; q < 8; ++q) {
      const double *cc = q < 4 ? cb : cf; const int qq = q & 3;
      const double G2 = q < 4 ? 0.0 : sp.wb * (-cc[2 * qq] * sn + cc[2 * qq + 1] * cs);
      const double ds = -w.c[r + q] - (cc[2 * qq] * w.dx[b] + cc[2 * qq + 1] * w.dx[b + 1] + G2 * w.dx[b + 2]);
      w.dx[sl + q] = ds; w.dnu[r + q] = -w.r1[sl + q] - (sig[sl + q] + shift) * ds;
    }
  }
  if (sp.has_final && CFZP_FIRST) w.dnu[d.m - 1] = eta;
  CFZP_SYNC();
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

// One plan per workgroup of ONE wavefront (the sweep is one lane's work, everything else is O(n) and shared by the lanes).
template <bool FAST>
CFZP_FN void solve_state_ws(const PSpec &sp, const double *tube, double *X, double *slab, int *out_i, double *out_d) {
  const PDims d = dims(sp);
  const PWork w = carve(sp, slab);
  const int n = d.n, m = d.m;
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
      if (!newton_step<FAST>(sp, tube, w, sig, delta)) {
        double curv = 0.0, dd = 0.0, bad = 0.0;  // dx'(H) dx = -dx.r1 + c.dnu - delta_c eta^2  (from the two block rows of the system)
        CFZP_LANE_FOR(i, 0, n - 1) { const double v = w.dx[i]; if (!isfinite(v)) bad = 1.0; curv -= v * w.r1[i]; dd += v * v; }
        CFZP_LANE_FOR(i, 0, m - 1) { const double v = w.dnu[i]; if (!isfinite(v)) bad = 1.0; curv += w.c[i] * v - ((sp.has_final && i == m - 1) ? sp.reg_dual * v * v : 0.0); }
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
