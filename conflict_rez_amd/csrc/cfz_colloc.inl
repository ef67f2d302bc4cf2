// The single-vehicle collocation plan (reference confrez/control/vehicle.py:360-661, `setup_single_final_problem` +
// `solve_single_final_problem`): N = N_per_set (S-1) intervals of free length dt, K = 5 Radau points each.
//
//   variables    p_ik = (x, y, psi, v, delta, a, w) at the 6 points of every interval (:402-409), dt (:386-390)
//   ODE          sum_j A[j,k] z_ij / dt = f(z_ik, u_ik) at ALL six points (:487-509)  [stated times dt: sum_j A z - dt f = 0]
//   continuity   sum_j D[j] p_{i-1,j} = p_{i,0}, states and inputs (:544-568); Radau: D = e_K, so p_{i,0} = p_{i-1,K}
//   tube         rear-axle and front point of p_{i,0} inside the shrunk cells at every N_per_set-th interval (:570-588)
//                and of the end state z_F = p_{N-1,K} in the last cells (:606-617)
//   terminal     v = delta = a = w = 0 at the end (:622-626), optional heading (:619-620); initial pose fixed,
//                v = delta = a = w = 0 at the start (:426-436)
//   boxes        x, y, v, delta, a, w at every point (:439-478)
//   collision    every point against every static obstacle, at least dmin apart (:523-541) -- with the OBCA duals
//                eliminated into two smooth rows per (point, obstacle) over a working set, exactly as in the MPC step
//                (DESIGN.md "Certificate elimination"); the duals l, m are rebuilt from the poses afterwards
//   cost         sum_ik B[k] (a^2 + v^2 w^2 + delta^2) dt + (N dt)^2 (:512-521, :638)
//
// The same source solves the JOINT plan of several vehicles (reference confrez/control/multi_vehicle_planner.py:343-480):
// see CSpec below.
//
// Solver: the banded primal-dual interior point of cfz_plan.inl (exact Hessian, delta_w ladder with the curvature test,
// delta_c), with these additions: dt couples to everything, so it is kept out of the band and handled by bordering
// (one factorisation, two right-hand sides); the working set of the collision rows is refreshed at every accepted
// iterate; every collision pair (slack, multiplier) is condensed into the poses it touches (assemble); a primal-dual
// merit takes over when the filter line search fails (solve_colloc).  One workgroup per plan, workspace in global
// memory; every thread runs the scalar logic, the marked loops are shared.  One vehicle: one wavefront, elimination in
// an LDS window (cfz_band.inl).  Several vehicles: eight wavefronts, band in global memory, eliminated a panel of sixteen pivots at
// a time (band_factor_panel; band_factor_wide2, one pivot at a time, is its check and fallback), both right-hand sides
// substituted in one sweep (band_substitute_regs).
#pragma once
#include "cfz_solver.inl"
#include "cfz_plan.inl"
#include "cfz_band.inl"

// Everything is inlined into the kernel on purpose.  With the big pieces as separate (noinline) device functions the
// build was broken on this toolchain (gfx950, ROCm 7.2) as soon as such a function named LDS -- the dynamic window
// (`extern __shared__`) or the static scratch of the block reductions: memory aperture violations, or silently wrong
// LDS contents (measured with tools/colloc_timing_one.sh on four build variants).  The price is ~360 spilled SGPRs.
#if defined(__HIP_DEVICE_COMPILE__)
#define CFZC_PIECE __device__ inline
#else
#define CFZC_PIECE inline
#endif

#ifndef CFZC_NO_EASE
#define CFZC_NO_EASE 0  // experiments: 1 = mu falls only when the barrier problem's own test passes
#endif
#ifndef CFZC_VV_TANGENTIAL
#define CFZC_VV_TANGENTIAL 1.0
#endif
namespace cfzc {

// Dual regularisation of proximal type: the constraint rows of the Newton system read  J dx - delta_c (nu + dnu) = -c
// instead of IPOPT's  J dx - delta_c dnu = -c.  With the latter, multiplier components that the rows do not
// determine (the reference's formulation loses LICQ wherever a vehicle stands still: six ODE rows per interval and state
// on a rank-5 derivative matrix) accumulate c_y / delta_c per iteration and wander; with the former they are SET to
// c_y / delta_c each time.  The fixed point moves to c = delta_c nu (a quadratic penalty of weight 1 / delta_c on the
// rows), which the termination test sees as constraint violation and bounds by constr_viol_tol.
// CSpec::no_prox = 1 switches it off (the exact solution at tight tolerances, tests/test_independent_solver.py).

constexpr int kPts = 6;     // points per interval (K + 1)
constexpr int kOutD = 20;    // out_d: cost, err, mu, then 100 MHz ticks spent in evaluation, assembly, factorisation, substitution,
                            // line search, and in total, three sub-phases of the elimination, eight of the joint structured elimination (zero on the CPU)
CFZP_FN long long tick() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (long long)wall_clock64();
#else
  return 0;
#endif
}
constexpr int kMaxObs = 8;

// Reductions over the partial results of a CFZP_LANE_FOR loop.  The solver may run on several wavefronts (the joint plan:
// every thread executes the scalar logic redundantly, the marked loops are split over all threads of the workgroup); the
// partial results of the wavefronts are then combined through LDS in a fixed order, so every thread gets the same bits.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline double block_combine(double v, int op) {  // op 0 sum, 1 max, 2 min of the per-wavefront values
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
#endif
CFZP_FN double bsum(double v) {
  v = cfzp::wsum(v);
#if defined(__HIP_DEVICE_COMPILE__)
  v = block_combine(v, 0);
#endif
  return v;
}
CFZP_FN double bmax(double v) {
  v = cfzp::wmax(v);
#if defined(__HIP_DEVICE_COMPILE__)
  v = block_combine(v, 1);
#endif
  return v;
}
CFZP_FN double bmin(double v) {
  v = cfzp::wmin(v);
#if defined(__HIP_DEVICE_COMPILE__)
  v = block_combine(v, 2);
#endif
  return v;
}

// A marked loop whose items each read several words.  On the GPU a plan has its CU to itself with two wavefronts per SIMD: nothing hides a
// memory round trip, and an item written as "if (bounded) { load ...; load ... }" costs two or three of them one after the other, nine
// items per thread.  Here U items' loads -- load(i) is branch-free -- are issued before the first is used, then the items are consumed in
// ascending order: the order, hence the bits, of CFZP_LANE_FOR.  On the CPU a plain loop.
struct LV { double v[12]; int k[2]; };
template <int U, class L, class C>
CFZP_FN void lane_for_loads(int n, L load, C consume) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int nt = (int)blockDim.x;
  for (int b = (int)threadIdx.x; b < n; b += U * nt) {
    LV v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int ii = b + u * nt; v[u] = load(ii < n ? ii : n - 1); }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int ii = b + u * nt; if (ii < n) consume(ii, v[u]); }
  }
#else
  for (int i = 0; i < n; ++i) consume(i, load(i));
#endif
}

constexpr int kMaxVeh = 4, kMaxPairs = 6;

// V = 1: the single-vehicle plan.  V > 1: the joint plan of the centralised planner (reference
// confrez/control/multi_vehicle_planner.py:343-480, `solve_final_problem_obca`): every vehicle's collocation problem
// with ONE shared dt (:365-366, :378-387), cost sum_a J_a, and for every pair of vehicles and every point of the shorter
// plan the two bodies at least dmin apart (:389-456) -- again as two smooth rows per (pair, point) over a working set.
struct CSpec {
  int V, Nps, n_obs, n_pairs;
  int max_iter, max_backtrack, filter_cap, no_prox;  // no_prox: bit 0 = no proximal term, bit 1 = eliminate one pivot at a time
  int vv_rows, pad0;  // vv_rows 1: vertex-vertex rows (kind 3, cfz_solver.inl) in the working sets of obstacle and pair blocks
  int N[kMaxVeh], n_chk[kMaxVeh], has_final[kMaxVeh];
  int pair_a[kMaxPairs], pair_b[kMaxPairs];
  double wb, dmin, shrink, dt0;
  double final_heading[kMaxVeh];
  double init_pose[kMaxVeh][3];
  double bounds[12];  // lo,hi of x, y, v, delta, a, w
  double g[4];        // body polytope offsets
  double A[kPts][kPts], B[kPts];  // collocation tables: A[j][k] = l_j'(tau_k), B[k] = quadrature weights
  double tol, constr_viol_tol, dual_inf_tol, compl_inf_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min,
      bound_push, bound_frac, s_max, kappa_sigma, eta_phi, gamma_theta, gamma_phi, delta_sw, s_theta, s_phi,
      reg_primal, reg_dual, curv_kappa;
  const double *obs_tab;         // n_obs x 20: A[4][2], b[4], V[4][2]
  const double *tube[kMaxVeh];   // per vehicle n_chk x 2 x 12: back cell, front cell (A[4][2], b[4]) of strategy steps 1..S-1
};

// Intervals of all vehicles are numbered vehicle-major: I = off[a] + i, points q = 6 I + k.
//   variables    7 per point | dt | static-obstacle slacks (nr per point) | tube slacks (8 per checkpoint) | pair slacks (2 per pair point)
//   constraints  init 7 per vehicle | ODE 5 per point | continuity 7 per interval but each vehicle's first | obstacle rows |
//                tube rows | terminal 5 per vehicle (v, delta, a, w, heading; the heading row is dead without one) | pair rows
struct CDims {
  int V, NI, np, nr, nchk, npp, n, m, nk;
  int iDt, sO, sT, sP;
  int rO, rC, rR, rT, rF, rP;
  int off[kMaxVeh + 1], coff[kMaxVeh + 1], poff[kMaxPairs + 1];
};
// The structured elimination of the JOINT plan (cfz_jstruct.inl; CSpec::no_prox bit 2 with V > 1) works on another statement of the same
// Newton system: positions vehicle-major (build_order), the tube slacks and rows condensed into the pose block they touch like the
// collision rows (so a vehicle's separators hold at most 15 unknowns), the condensed pair blocks kept beside the band (CWork::pm).
CFZP_FN bool jstruct_mode(const CSpec &sp) { return (sp.no_prox & 4) != 0; }  // (bit 3 chose between this scheme and round 4's for single plans until round 6: ignored)
CFZP_FN CDims cdims(const CSpec &sp) {
  CDims d;
  d.V = sp.V; d.off[0] = 0; d.coff[0] = 0;
  int nfin = 0;
  for (int a = 0; a < sp.V; ++a) { d.off[a + 1] = d.off[a] + sp.N[a]; d.coff[a + 1] = d.coff[a] + sp.n_chk[a]; nfin += sp.has_final[a] ? 1 : 0; }
  for (int a = sp.V; a < kMaxVeh; ++a) { d.off[a + 1] = d.off[sp.V]; d.coff[a + 1] = d.coff[sp.V]; }
  d.NI = d.off[sp.V]; d.nchk = d.coff[sp.V]; d.np = d.NI * kPts; d.nr = 2 * sp.n_obs;
  d.poff[0] = 0;
  for (int e = 0; e < kMaxPairs; ++e) {
    int cnt = 0;
    if (e < sp.n_pairs) { const int na = sp.N[sp.pair_a[e]], nb = sp.N[sp.pair_b[e]]; cnt = (na < nb ? na : nb) * kPts; }
    d.poff[e + 1] = d.poff[e] + cnt;
  }
  d.npp = d.poff[kMaxPairs];
  d.iDt = 7 * d.np; d.sO = d.iDt + 1; d.sT = d.sO + d.np * d.nr; d.sP = d.sT + 8 * d.nchk; d.n = d.sP + 2 * d.npp;
  d.rO = 7 * sp.V; d.rC = d.rO + 5 * d.np; d.rR = d.rC + 7 * (d.NI - sp.V); d.rT = d.rR + d.np * d.nr; d.rF = d.rT + 8 * d.nchk;
  d.rP = d.rF + 5 * sp.V; d.m = d.rP + 2 * d.npp;
  // dt is bordered, collision slacks and rows (obstacles and pairs) are condensed, dead heading rows are left out
  d.nk = 7 * d.np + 8 * d.nchk + 7 * sp.V + 5 * d.np + 7 * (d.NI - sp.V) + 8 * d.nchk + 4 * sp.V + nfin;
  if (jstruct_mode(sp)) d.nk -= 16 * d.nchk;
  return d;
}
CFZP_FN int veh_of_interval(const CDims &d, int I) { int a = 0; while (a + 1 < d.V && I >= d.off[a + 1]) ++a; return a; }
CFZP_FN int veh_of_chk(const CDims &d, int T) { int a = 0; while (a + 1 < d.V && T >= d.coff[a + 1]) ++a; return a; }
// half-bandwidth of the ordering of build_order: the 30 ODE rows of an interval sit between its third and fourth point
constexpr int kCB = 51;  // half-bandwidth of the single-vehicle plan's ordering
constexpr int kWideMaxKb = 448;  // widest half-bandwidth the eight-wavefront eliminations are compiled for (the host sizes LDS with it)
// point of tube checkpoint T (global index): start of interval (t+1) Nps of its vehicle, or the vehicle's very last point
CFZP_FN int chk_point(const CSpec &sp, const CDims &d, int T) {
  const int a = veh_of_chk(d, T), t = T - d.coff[a];
  return t + 1 < sp.n_chk[a] ? (d.off[a] + (t + 1) * sp.Nps) * kPts : d.off[a + 1] * kPts - 1;
}
CFZP_FN const double *chk_cell(const CSpec &sp, const CDims &d, int T, int front) { const int a = veh_of_chk(d, T); return cfzp::cell(sp.tube[a], T - d.coff[a], front); }

CFZP_FN void obstacle(const CSpec &sp, int j, double A[4][2], double b[4], double V[4][2]) {
  const double *o = sp.obs_tab + j * 20;
  for (int i = 0; i < 4; ++i) { A[i][0] = o[2 * i]; A[i][1] = o[2 * i + 1]; b[i] = o[8 + i]; V[i][0] = o[12 + 2 * i]; V[i][1] = o[13 + 2 * i]; }
}

CFZP_FN void f_ct(const double *p, double wb, double f[5]) {
  f[0] = p[3] * cos(p[2]); f[1] = p[3] * sin(p[2]); f[2] = p[3] / wb * tan(p[4]); f[3] = p[5]; f[4] = p[6];
}
CFZP_FN double stage_err(const double *p) { return p[5] * p[5] + p[3] * p[3] * p[6] * p[6] + p[4] * p[4]; }

// polygon of a vehicle body at pose (x, y, psi): faces R G_f, vertices t + R b_v (the layout cfz::select_rows expects)
CFZP_FN void veh_polygon(const double *q, const double g[4], double A[4][2], double b[4], double V[4][2]) {
  const double co = cos(q[2]), so = sin(q[2]);
  A[0][0] = co; A[0][1] = so; A[1][0] = -so; A[1][1] = co; A[2][0] = -co; A[2][1] = -so; A[3][0] = so; A[3][1] = -co;
  for (int i = 0; i < 4; ++i) b[i] = A[i][0] * q[0] + A[i][1] * q[1] + g[i];
  const double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
  for (int i = 0; i < 4; ++i) { V[i][0] = q[0] + co * BV[i][0] - so * BV[i][1]; V[i][1] = q[1] + so * BV[i][0] + co * BV[i][1]; }
}

// One separation row of a pair of vehicles a, b (poses pa, pb = x, y, psi), as seen from a like the rows of an obstacle:
// kind 1 = face f of b against body vertex v of a, kind 2 = face f of a against body vertex v of b.  With F the vehicle that
// owns the face and W the one that owns the vertex:  val = n.(t_W + R_W b_v - t_F) - g_f,  n = R_F G_f.
// kind 3 = body vertex f of b against body vertex v of a, each the other's closest feature: val = |w|,
// w = t_a + R_a b_v - t_b - R_b b_f (the reference's rows admit any unit direction, multi_vehicle_planner.py:419-451).
// gr, H: derivatives with respect to (a: x, y, psi | b: x, y, psi).
template <bool DER>
CFZP_FN double pair_row(const double *pa, const double *pb, const double g[4], int kind, int f, int v, double gr[6], double H[6][6]) {
  if (kind == 3) {
    const double ca = cos(pa[2]), sa = sin(pa[2]), cb = cos(pb[2]), sb = sin(pb[2]);
    const double ax = (v == 0 || v == 3) ? g[0] : -g[2], ay = (v < 2) ? g[1] : -g[3];
    const double ux = (f == 0 || f == 3) ? g[0] : -g[2], uy = (f < 2) ? g[1] : -g[3];
    const double rax = ca * ax - sa * ay, ray = sa * ax + ca * ay, rbx = cb * ux - sb * uy, rby = sb * ux + cb * uy;  // R_a b_v, R_b b_f
    const double wx = pa[0] + rax - pb[0] - rbx, wy = pa[1] + ray - pb[1] - rby;
    const double r = sqrt(wx * wx + wy * wy);
    if (DER) {
      const double ir = 1.0 / r, n0 = wx * ir, n1 = wy * ir;
      // d(R b)/dpsi = (-Rb_y, Rb_x); tangent t = (-n1, n0);  H = tau tau' / r + n.w'' with tau = (dw/dz)' t
      const double nda = -n0 * ray + n1 * rax, ndb = -n0 * rby + n1 * rbx, tda = n1 * ray + n0 * rax, tdb = n1 * rby + n0 * rbx;
      gr[0] = n0; gr[1] = n1; gr[2] = nda; gr[3] = -n0; gr[4] = -n1; gr[5] = -ndb;
      const double tau[6] = {-n1, n0, tda, n1, -n0, -tdb};
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H[i][j] = CFZC_VV_TANGENTIAL * tau[i] * tau[j] * ir;
      H[2][2] -= n0 * rax + n1 * ray;
      H[5][5] += n0 * rbx + n1 * rby;
    }
    return r;
  }
  const double *pF = kind == 1 ? pb : pa, *pW = kind == 1 ? pa : pb;
  const int oF = kind == 1 ? 3 : 0, oW = kind == 1 ? 0 : 3;
  const double gx = (f == 0) - (f == 2), gy = (f == 1) - (f == 3);
  const double gf = f == 0 ? g[0] : (f == 1 ? g[1] : (f == 2 ? g[2] : g[3]));
  const double cF = cos(pF[2]), sF = sin(pF[2]), cW = cos(pW[2]), sW = sin(pW[2]);
  const double nx = cF * gx - sF * gy, ny = sF * gx + cF * gy;
  const double bx = (v == 0 || v == 3) ? g[0] : -g[2], by = (v < 2) ? g[1] : -g[3];
  const double rx = cW * bx - sW * by, ry = sW * bx + cW * by;
  const double wx = pW[0] + rx - pF[0], wy = pW[1] + ry - pF[1];
  if (DER) {
    const double dnx = -ny, dny = nx, drx = -ry, dry = rx;
    for (int i = 0; i < 6; ++i) { gr[i] = 0.0; for (int j = 0; j < 6; ++j) H[i][j] = 0.0; }
    gr[oW] = nx; gr[oW + 1] = ny; gr[oW + 2] = nx * drx + ny * dry;
    gr[oF] = -nx; gr[oF + 1] = -ny; gr[oF + 2] = dnx * wx + dny * wy;
    H[oF + 2][oF + 2] = -(nx * wx + ny * wy);
    H[oW + 2][oW + 2] = -(nx * rx + ny * ry);
    H[oF + 2][oW + 2] = H[oW + 2][oF + 2] = dnx * drx + dny * dry;
    H[oF + 2][oW] = H[oW][oF + 2] = dnx; H[oF + 2][oW + 1] = H[oW + 1][oF + 2] = dny;
    H[oF + 2][oF] = H[oF][oF + 2] = -dnx; H[oF + 2][oF + 1] = H[oF + 1][oF + 2] = -dny;
  }
  return nx * wx + ny * wy - gf;
}
// value of row rr of a pair block with working-set code sl; the second slot of a vertex-vertex block is inert (cfz::kVvInert)
CFZP_FN double pair_value(const double *pa, const double *pb, const double g[4], int sl, int rr) {
  const double v = pair_row<false>(pa, pb, g, sl >> 6, (sl >> 4) & 3, rr == 0 ? (sl >> 2) & 3 : sl & 3, nullptr, nullptr);
  return ((sl >> 6) == 3 && rr == 1) ? v + cfz::kVvInert : v;
}
// the two points of pair point r of pair e: same interval and collocation index in both vehicles' plans
CFZP_FN void pair_points(const CSpec &sp, const CDims &d, int e, int r, int *qa, int *qb) {
  *qa = d.off[sp.pair_a[e]] * kPts + r; *qb = d.off[sp.pair_b[e]] * kPts + r;
}

CFZC_PIECE double objective(const CSpec &sp, const double *X) {
  const CDims d = cdims(sp);
  double s = 0.0;
  CFZP_LANE_FOR(q, 0, d.np - 1) s += sp.B[q % kPts] * stage_err(X + 7 * q);
  const double dt = X[d.iDt];
  double tt = 0.0;
  for (int a = 0; a < sp.V; ++a) tt += (sp.N[a] * dt) * (sp.N[a] * dt);
  return bsum(s) * dt + tt;
}

// c(X); sel[np * n_obs | npp] is the working set of the collision rows (obstacles, then pairs)
CFZC_PIECE void constraints(const CSpec &sp, const unsigned char *sel, const double *X, double *c) {
  const CDims d = cdims(sp);
  const double dt = X[d.iDt];
  for (int a = 0; a < sp.V; ++a) {
    const double *p0 = X + 7 * kPts * d.off[a], *pl = X + 7 * (kPts * d.off[a + 1] - 1);
    for (int i = 0; i < 3; ++i) c[7 * a + i] = p0[i] - sp.init_pose[a][i];
    for (int i = 3; i < 7; ++i) c[7 * a + i] = p0[i];
    for (int i = 0; i < 4; ++i) c[d.rF + 5 * a + i] = pl[3 + i];
    c[d.rF + 5 * a + 4] = sp.has_final[a] ? pl[2] - sp.final_heading[a] : 0.0;
  }
  CFZP_LANE_FOR(q, 0, d.np - 1) {
    const int i = q / kPts, k = q - i * kPts, va = veh_of_interval(d, i), il = i - d.off[va];
    const double *p = X + 7 * q;
    double f[5];
    f_ct(p, sp.wb, f);
    for (int cc = 0; cc < 5; ++cc) {
      double s = -dt * f[cc];
      for (int j = 0; j < kPts; ++j) s += sp.A[j][k] * X[7 * (i * kPts + j) + cc];
      c[d.rO + 5 * q + cc] = s;
    }
    if (k == 0 && il >= 1) for (int cc = 0; cc < 7; ++cc) c[d.rC + 7 * (i - va - 1) + cc] = p[cc] - X[7 * (q - 1) + cc];
    double sn, cs;
    sincos(p[2], &sn, &cs);
    for (int j = 0; j < sp.n_obs; ++j) {
      double A[4][2], b[4], V[4][2], sep[2];
      obstacle(sp, j, A, b, V);
      cfz::rows_for<false>(A, b, V, p[0], p[1], cs, sn, sp.g, sel[q * sp.n_obs + j], sep, nullptr);
      for (int r = 0; r < 2; ++r) c[d.rR + q * d.nr + 2 * j + r] = sep[r] - sp.dmin - X[d.sO + q * d.nr + 2 * j + r];
    }
  }
  CFZP_LANE_FOR(t, 0, d.nchk - 1) {
    const double *z = X + 7 * chk_point(sp, d, t);
    const double fx = z[0] + sp.wb * cos(z[2]), fy = z[1] + sp.wb * sin(z[2]);
    const double *cb = chk_cell(sp, d, t, 0), *cf = chk_cell(sp, d, t, 1);
    for (int r = 0; r < 4; ++r) {
      c[d.rT + 8 * t + r] = cb[2 * r] * z[0] + cb[2 * r + 1] * z[1] - (cb[8 + r] - sp.shrink) + X[d.sT + 8 * t + r];
      c[d.rT + 8 * t + 4 + r] = cf[2 * r] * fx + cf[2 * r + 1] * fy - (cf[8 + r] - sp.shrink) + X[d.sT + 8 * t + 4 + r];
    }
  }
  for (int e = 0; e < sp.n_pairs; ++e) {
    CFZP_LANE_FOR(r, 0, d.poff[e + 1] - d.poff[e] - 1) {
      int qa, qb;
      pair_points(sp, d, e, r, &qa, &qb);
      const int pp = d.poff[e] + r, sl = sel[d.np * sp.n_obs + pp];
      for (int rr = 0; rr < 2; ++rr)
        c[d.rP + 2 * pp + rr] = pair_value(X + 7 * qa, X + 7 * qb, sp.g, sl, rr) - sp.dmin - X[d.sP + 2 * pp + rr];
    }
  }
  CFZP_SYNC();
}

CFZC_PIECE void gradient(const CSpec &sp, const double *X, double *g) {
  const CDims d = cdims(sp);
  const double dt = X[d.iDt];
  double s = 0.0;
  CFZP_LANE_FOR(q, 0, d.np - 1) {
    const double *p = X + 7 * q; const double bk = sp.B[q % kPts];
    double *o = g + 7 * q;
    o[0] = o[1] = o[2] = 0.0;
    o[3] = bk * dt * 2.0 * p[3] * p[6] * p[6]; o[4] = bk * dt * 2.0 * p[4]; o[5] = bk * dt * 2.0 * p[5]; o[6] = bk * dt * 2.0 * p[3] * p[3] * p[6];
    s += bk * stage_err(p);
  }
  CFZP_LANE_FOR(i, d.sO, d.n - 1) g[i] = 0.0;
  double nn = 0.0;
  for (int a = 0; a < sp.V; ++a) nn += 2.0 * sp.N[a] * sp.N[a];
  g[d.iDt] = bsum(s) + nn * dt;
  CFZP_SYNC();
}

// out = J(X)' nu (all n entries, dt included)
CFZC_PIECE void jt_nu(const CSpec &sp, const unsigned char *sel, const double *X, const double *nu, double *out) {
  const CDims d = cdims(sp);
  const double dt = X[d.iDt];
  double sdt = 0.0;
  CFZP_LANE_FOR(q, 0, d.np - 1) {
    const int i = q / kPts, k = q - i * kPts, va = veh_of_interval(d, i), il = i - d.off[va];
    const double *p = X + 7 * q, *l = nu + d.rO + 5 * q;
    double o[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int kk = 0; kk < kPts; ++kk) {  // this point's states enter the ODE rows of all six points of the interval
      const double *lk = nu + d.rO + 5 * (i * kPts + kk);
      for (int cc = 0; cc < 5; ++cc) o[cc] += sp.A[k][kk] * lk[cc];
    }
    const double cs = cos(p[2]), sn = sin(p[2]), tn = tan(p[4]), v = p[3];
    double f[5];
    f_ct(p, sp.wb, f);
    for (int cc = 0; cc < 5; ++cc) sdt -= l[cc] * f[cc];
    o[2] -= dt * (-v * sn * l[0] + v * cs * l[1]);
    o[3] -= dt * (cs * l[0] + sn * l[1] + tn / sp.wb * l[2]);
    o[4] -= dt * (v / sp.wb * (1.0 + tn * tn) * l[2]);
    o[5] -= dt * l[3]; o[6] -= dt * l[4];
    if (k == 0 && il >= 1) for (int cc = 0; cc < 7; ++cc) o[cc] += nu[d.rC + 7 * (i - va - 1) + cc];
    if (k == kPts - 1 && il + 1 < sp.N[va]) for (int cc = 0; cc < 7; ++cc) o[cc] -= nu[d.rC + 7 * (i - va) + cc];
    if (k == 0 && il == 0) for (int cc = 0; cc < 7; ++cc) o[cc] += nu[7 * va + cc];
    if (k == kPts - 1 && il + 1 == sp.N[va]) { for (int cc = 0; cc < 4; ++cc) o[3 + cc] += nu[d.rF + 5 * va + cc]; if (sp.has_final[va]) o[2] += nu[d.rF + 5 * va + 4]; }
    for (int j = 0; j < sp.n_obs; ++j) {
      double A[4][2], b[4], V[4][2], sep[2], gr[2][3];
      obstacle(sp, j, A, b, V);
      cfz::rows_for<true>(A, b, V, p[0], p[1], cs, sn, sp.g, sel[q * sp.n_obs + j], sep, gr);
      for (int r = 0; r < 2; ++r) {
        const double nr_ = nu[d.rR + q * d.nr + 2 * j + r];
        o[0] += gr[r][0] * nr_; o[1] += gr[r][1] * nr_; o[2] += gr[r][2] * nr_;
        out[d.sO + q * d.nr + 2 * j + r] = -nr_;
      }
    }
    for (int cc = 0; cc < 7; ++cc) out[7 * q + cc] = o[cc];
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(t, 0, d.nchk - 1) {
    const int b = 7 * chk_point(sp, d, t);
    const double cs = cos(X[b + 2]), sn = sin(X[b + 2]);
    const double *cb = chk_cell(sp, d, t, 0), *cf = chk_cell(sp, d, t, 1);
    for (int r = 0; r < 4; ++r) {
      const double lb = nu[d.rT + 8 * t + r], lf = nu[d.rT + 8 * t + 4 + r];
      out[b] += cb[2 * r] * lb + cf[2 * r] * lf; out[b + 1] += cb[2 * r + 1] * lb + cf[2 * r + 1] * lf;
      out[b + 2] += sp.wb * (-cf[2 * r] * sn + cf[2 * r + 1] * cs) * lf;
      out[d.sT + 8 * t + r] = lb; out[d.sT + 8 * t + 4 + r] = lf;
    }
  }
  CFZP_SYNC();
  for (int e = 0; e < sp.n_pairs; ++e) {  // one pair after the other: within a pair every point appears once
    CFZP_LANE_FOR(r, 0, d.poff[e + 1] - d.poff[e] - 1) {
      int qa, qb;
      pair_points(sp, d, e, r, &qa, &qb);
      const int pp = d.poff[e] + r, sl = sel[d.np * sp.n_obs + pp];
      for (int rr = 0; rr < 2; ++rr) {
        double gr[6], H[6][6];
        pair_row<true>(X + 7 * qa, X + 7 * qb, sp.g, sl >> 6, (sl >> 4) & 3, rr == 0 ? (sl >> 2) & 3 : sl & 3, gr, H);
        const double nr_ = nu[d.rP + 2 * pp + rr];
        for (int a = 0; a < 3; ++a) { out[7 * qa + a] += gr[a] * nr_; out[7 * qb + a] += gr[3 + a] * nr_; }
        out[d.sP + 2 * pp + rr] = -nr_;
      }
    }
    CFZP_SYNC();
  }
  out[d.iDt] = bsum(sdt);
  CFZP_SYNC();
}

// ---- band ordering ---------------------------------------------------------------------------------------------
// per interval: [continuity mults | (tube) | pt0 | pt1 | pt2 | ODE mults of the 6 points | pt3 | pt4 | pt5 | (last tube)], pt =
// the 7 variables of the point; initial rows first, terminal rows last.  dt has no position (bordered), and neither have
// the collision slacks and rows: each pair (sigma_r, nu_r) only touches its own point's pose and is eliminated exactly
// (assemble), which leaves a half-bandwidth of 51 whatever the number of obstacles.
// With several vehicles the interval blocks are interleaved in time (interval t of vehicle 0, of vehicle 1, ...): the pair
// rows couple the poses of the same (t, k) of two vehicles, which then sit (b - a) blocks apart.
// Vehicle-major ordering of the joint plan's structured elimination: every vehicle's unknowns in one range, per interval
// [continuity (initial rows) | pt0 | pt1 pt2 | 30 ODE rows | pt3 pt4 pt5], terminal rows last; no tube positions (condensed).  With the
// steering rate of pt5 counted to the next separator: separator 0 = 14 unknowns at base, interior t = [base + 79 t + 14, + 64),
// separator t = [base + 79 t - 1, + 15), the last separator = the steering rate and the terminal rows (cfz_jstruct.inl).
CFZP_FN int build_order_vm(const CSpec &sp, int *posx, int *posc) {
  const CDims d = cdims(sp);
  int p = 0;
  for (int a = 0; a < sp.V; ++a) {
    for (int cc = 0; cc < 7; ++cc) posc[7 * a + cc] = p++;
    for (int t = 0; t < sp.N[a]; ++t) {
      const int i = d.off[a] + t;
      if (t >= 1) for (int cc = 0; cc < 7; ++cc) posc[d.rC + 7 * (i - a - 1) + cc] = p++;
      for (int k = 0; k < kPts; ++k) {
        const int q = i * kPts + k;
        if (k == 3) for (int kk = 0; kk < kPts; ++kk) for (int cc = 0; cc < 5; ++cc) posc[d.rO + 5 * (i * kPts + kk) + cc] = p++;
        for (int cc = 0; cc < 7; ++cc) posx[7 * q + cc] = p++;
        for (int r = 0; r < d.nr; ++r) { posx[d.sO + q * d.nr + r] = -1; posc[d.rR + q * d.nr + r] = -1; }
      }
    }
    for (int cc = 0; cc < 4; ++cc) posc[d.rF + 5 * a + cc] = p++;
    posc[d.rF + 5 * a + 4] = sp.has_final[a] ? p++ : -1;
  }
  for (int r = 0; r < 8 * d.nchk; ++r) { posx[d.sT + r] = -1; posc[d.rT + r] = -1; }
  for (int r = 0; r < 2 * d.npp; ++r) { posx[d.sP + r] = -1; posc[d.rP + r] = -1; }
  posx[d.iDt] = -1;
  return p;
}
CFZP_FN int build_order(const CSpec &sp, int *posx, int *posc) {
  if (jstruct_mode(sp)) return build_order_vm(sp, posx, posc);
  const CDims d = cdims(sp);
  int p = 0, nmax = 0;
  for (int a = 0; a < sp.V; ++a) nmax = sp.N[a] > nmax ? sp.N[a] : nmax;
  for (int t = 0; t < nmax; ++t)
    for (int a = 0; a < sp.V; ++a) {
      if (t >= sp.N[a]) continue;
      const int i = d.off[a] + t;
      if (t == 0) for (int cc = 0; cc < 7; ++cc) posc[7 * a + cc] = p++;
      for (int k = 0; k < kPts; ++k) {
        const int q = i * kPts + k;
        for (int pass = 0; pass < 2; ++pass) {  // tube block: before an interval's first point, after the very last point
          if (pass == 0 && k == 0 && t >= 1) for (int cc = 0; cc < 7; ++cc) posc[d.rC + 7 * (i - a - 1) + cc] = p++;
          for (int T = d.coff[a]; T < d.coff[a + 1]; ++T)
            if (chk_point(sp, d, T) == q && (pass == 0) == (k == 0)) {
              for (int r = 0; r < 8; ++r) posx[d.sT + 8 * T + r] = p++;
              for (int r = 0; r < 8; ++r) posc[d.rT + 8 * T + r] = p++;
            }
          if (pass == 1) break;
          if (k == 3) for (int kk = 0; kk < kPts; ++kk) for (int cc = 0; cc < 5; ++cc) posc[d.rO + 5 * (i * kPts + kk) + cc] = p++;
          for (int cc = 0; cc < 7; ++cc) posx[7 * q + cc] = p++;
          for (int r = 0; r < d.nr; ++r) { posx[d.sO + q * d.nr + r] = -1; posc[d.rR + q * d.nr + r] = -1; }  // condensed
        }
      }
      if (t + 1 == sp.N[a]) {
        for (int cc = 0; cc < 4; ++cc) posc[d.rF + 5 * a + cc] = p++;
        posc[d.rF + 5 * a + 4] = sp.has_final[a] ? p++ : -1;
      }
    }
  for (int r = 0; r < 2 * d.npp; ++r) { posx[d.sP + r] = -1; posc[d.rP + r] = -1; }  // condensed
  posx[d.iDt] = -1;
  return p;
}
// half-bandwidth: 51 within an interval block; with several vehicles the continuity rows reach back over the other vehicles'
// blocks to the previous interval of their own, and the pair rows couple the poses of two vehicles
CFZP_FN int half_bandwidth(const CSpec &sp, const int *posx, const int *posc) {
  if (jstruct_mode(sp)) return kCB;  // within a vehicle; the pair blocks are kept beside the band
  const CDims d = cdims(sp);
  int kb = kCB;
  for (int a = 0; a < sp.V; ++a)
    for (int t = 1; t < sp.N[a]; ++t) {
      const int i = d.off[a] + t, lo = posx[7 * (kPts * i - 1)], hi = posc[d.rC + 7 * (i - a - 1) + 6];
      if (hi - lo > kb) kb = hi - lo;
    }
  for (int e = 0; e < sp.n_pairs; ++e)
    for (int r = 0; r < d.poff[e + 1] - d.poff[e]; ++r) {
      int qa, qb;
      pair_points(sp, d, e, r, &qa, &qb);
      const int lo = posx[7 * qa] < posx[7 * qb] ? posx[7 * qa] : posx[7 * qb], hi = posx[7 * qa] < posx[7 * qb] ? posx[7 * qb] : posx[7 * qa];
      if (hi + 2 - lo > kb) kb = hi + 2 - lo;
    }
  return kb;
}

using cfzb::Band;
CFZP_FN double &bnd(const Band &B, int i, int j) { return B.ab[(size_t)j * B.ld + (B.off + i - j)]; }
CFZP_FN void put(const Band &B, int i, int j, double v) { bnd(B, i, j) += v; if (i != j) bnd(B, j, i) += v; }

struct CWork {
  double *x, *xt, *zl, *zu, *nu, *dx, *dnu, *dzl, *dzu, *g, *c, *ct, *xl, *xu, *r1, *rhs, *rhs2, *bord, *ab, *sig, *cond, *condp, *sw;
  double *pm, *condt;  // joint structured elimination: the condensed pair blocks (36 per pair point), the tube rows' gradient, D, t (5 per row)
  int *posx, *posc, *ipiv;
  unsigned char *sel;
};
CFZP_FN size_t jstruct_doubles(const CSpec &sp);  // cfz_jstruct.inl: ... of the joint plans
CFZP_FN size_t work_doubles(const CSpec &sp, int kb) {
  const CDims d = cdims(sp);
  return jstruct_doubles(sp) + (jstruct_mode(sp) ? (size_t)d.npp * 36 + (size_t)d.nchk * 40 : 0) + (size_t)d.n * 12 + (size_t)d.m * 4 + (size_t)d.nk * (3 + (3 * kb + 1)) + (size_t)d.np * d.nr * 5 + (size_t)d.npp * 16 +
         (size_t)(d.n + d.m + d.nk + 2) / 2 + (size_t)(d.np * sp.n_obs + d.npp + 7) / 8 + 64;
}
CFZP_FN CWork carve(const CSpec &sp, int kb, double *slab) {
  const CDims d = cdims(sp);
  CWork w; double *p = slab;
  w.x = p; p += d.n; w.xt = p; p += d.n; w.zl = p; p += d.n; w.zu = p; p += d.n; w.dx = p; p += d.n; w.dzl = p; p += d.n;
  w.dzu = p; p += d.n; w.g = p; p += d.n; w.xl = p; p += d.n; w.xu = p; p += d.n; w.r1 = p; p += d.n; w.sig = p; p += d.n;
  w.nu = p; p += d.m; w.dnu = p; p += d.m; w.c = p; p += d.m; w.ct = p; p += d.m;
  w.rhs = p; p += d.nk; w.rhs2 = p; p += d.nk; w.bord = p; p += d.nk; w.ab = p; p += (size_t)d.nk * (3 * kb + 1);
  w.cond = p; p += (size_t)d.np * d.nr * 5;  // per obstacle row: gradient (3), D, t (assemble)
  w.condp = p; p += (size_t)d.npp * 16;     // per pair row: gradient (6), D, t
  w.pm = nullptr; w.condt = nullptr;
  if (jstruct_mode(sp)) { w.pm = p; p += (size_t)d.npp * 36; w.condt = p; p += (size_t)d.nchk * 40; }
  w.posx = reinterpret_cast<int *>(p); w.posc = w.posx + d.n; w.ipiv = w.posc + d.m;
  p += (size_t)(d.n + d.m + d.nk + 2) / 2;
  w.sel = reinterpret_cast<unsigned char *>(p);
  p += (size_t)(d.np * sp.n_obs + d.npp + 7) / 8 + 8;
  w.sw = p;
  return w;
}

// band part of [[W + Sigma + (delta + reg) I, J'], [J, -reg_dual I]] (dt row and column left out) and the border:
// bord = column of dt restricted to the band unknowns, hdd = its diagonal entry.  The caller has filled w.rhs with -r1 and
// -c of the band unknowns; the collision pairs are condensed here.  For row r of a point with gradient g (3 pose
// entries), slack sigma, S = Sigma_sigma + delta + reg and the residuals c_r, r_sigma:
//     S dsigma - dnu = -r_sigma,   g'dp - dsigma - reg_dual dnu = -c_r
//     =>  dnu = D (g'dp + t),  dsigma = (dnu - r_sigma) / S,   D = 1 / (1/S + reg_dual),  t = c_r + r_sigma / S
// so the pose block gains D g g' and the pose right-hand side loses D t g; g, D, t are kept in w.cond for `recover`.
// In the joint scheme (jstruct_mode: the band is never factored in place) the band is NOT cleared here: solve_colloc clears it once, and
// every entry this function ever writes is written by its FIRST writer with a plain store in every assembly -- a point's Hessian entries
// are summed in registers (in the order the band path adds them: the same bits) and stored once, Jacobian entries and the rows' -reg_dual
// have one writer each; the tube rows' condensed blocks are added afterwards.  What is never written stays the zero of that one clear.
// (A third of a single plan's HBM traffic was the clear and the read half of the read-modify-writes.)
CFZC_PIECE void band_clear(const CSpec &sp, const Band &Bd) {
  const CDims d = cdims(sp);
#if defined(__HIP_DEVICE_COMPILE__)
  {  // 16-byte stores
    double *z = Bd.ab;
    const int tot = d.nk * Bd.ld, head = (int)(((size_t)z >> 3) & 1);
    if (threadIdx.x == 0 && head) z[0] = 0.0;
    double2 *z2 = reinterpret_cast<double2 *>(z + head);
    const int n2 = (tot - head) >> 1;
    for (int t = (int)threadIdx.x; t < n2; t += (int)blockDim.x) z2[t] = make_double2(0.0, 0.0);
    if (threadIdx.x == 0 && ((tot - head) & 1)) z[tot - 1] = 0.0;
  }
#else
  CFZP_LANE_FOR(t, 0, d.nk * Bd.ld - 1) Bd.ab[t] = 0.0;
#endif
  CFZP_SYNC();
}
CFZP_FN void set2(const Band &B, int i, int j, double v) { bnd(B, i, j) = v; if (i != j) bnd(B, j, i) = v; }
CFZC_PIECE double assemble(const CSpec &sp, const CWork &w, const Band &Bd, double delta) {
  const CDims d = cdims(sp);
  const double prox = (sp.no_prox & 1) ? 0.0 : 1.0;
  const int *px = w.posx, *pc = w.posc;
  const double *X = w.x, *nu = w.nu;
  const double dt = X[d.iDt];
  const bool once = jstruct_mode(sp);  // (see above)
  if (!once) {
#if defined(__HIP_DEVICE_COMPILE__)
  {  // the band is cleared with 16-byte stores (88 MB per assembly of the four-vehicle plan)
    double *z = Bd.ab;
    const int tot = d.nk * Bd.ld, head = (int)(((size_t)z >> 3) & 1);
    if (threadIdx.x == 0 && head) z[0] = 0.0;
    double2 *z2 = reinterpret_cast<double2 *>(z + head);
    const int n2 = (tot - head) >> 1;
    for (int t = (int)threadIdx.x; t < n2; t += (int)blockDim.x) z2[t] = make_double2(0.0, 0.0);
    if (threadIdx.x == 0 && ((tot - head) & 1)) z[tot - 1] = 0.0;
  }
#else
  CFZP_LANE_FOR(t, 0, d.nk * Bd.ld - 1) Bd.ab[t] = 0.0;
#endif
  }
  CFZP_LANE_FOR(col, 0, d.nk - 1) w.bord[col] = 0.0;
  CFZP_SYNC();
  if (!once) {
    CFZP_LANE_FOR(i, 0, d.n - 1) if (px[i] >= 0) bnd(Bd, px[i], px[i]) += w.sig[i] + delta + sp.reg_primal;
    CFZP_LANE_FOR(i, 0, d.m - 1) if (pc[i] >= 0) bnd(Bd, pc[i], pc[i]) -= sp.reg_dual;
    CFZP_SYNC();
    CFZP_LANE_FOR(i, 0, 7 * sp.V - 1) put(Bd, pc[i], px[7 * kPts * d.off[i / 7] + i % 7], 1.0);
  } else {
    CFZP_LANE_FOR(i, 0, d.m - 1) if (pc[i] >= 0) bnd(Bd, pc[i], pc[i]) = 0.0 - sp.reg_dual;
    CFZP_LANE_FOR(i, 0, 7 * sp.V - 1) set2(Bd, pc[i], px[7 * kPts * d.off[i / 7] + i % 7], 1.0);
  }
  if (once) CFZP_LANE_FOR(q, 0, d.np - 1) {  // the same entries as the loop below, in the same order, a point's Hessian entries in registers
    const int i = q / kPts, k = q - i * kPts, b = 7 * q, va = veh_of_interval(d, i), il = i - d.off[va];
    const double *p = X + b, *l = nu + d.rO + 5 * q;
    const double cs = cos(p[2]), sn = sin(p[2]), tn = tan(p[4]), sec2 = 1.0 + tn * tn, v = p[3], wv = p[6], bk = sp.B[k];
    double dg[7], p01 = 0.0, p02 = 0.0, p12 = 0.0;
    for (int c = 0; c < 7; ++c) dg[c] = 0.0 + (w.sig[b + c] + delta + sp.reg_primal);
    dg[3] += bk * dt * 2.0 * wv * wv; dg[6] += bk * dt * 2.0 * v * v;
    set2(Bd, px[b + 3], px[b + 6], 0.0 + bk * dt * 4.0 * v * wv);
    dg[4] += bk * dt * 2.0; dg[5] += bk * dt * 2.0;
    const double l0 = -l[0] * dt, l1 = -l[1] * dt, l2 = -l[2] * dt;
    dg[2] += l0 * (-v * cs) + l1 * (-v * sn);
    set2(Bd, px[b + 2], px[b + 3], 0.0 + (l0 * (-sn) + l1 * cs));
    set2(Bd, px[b + 3], px[b + 4], 0.0 + l2 * sec2 / sp.wb);
    dg[4] += l2 * 2.0 * v * tn * sec2 / sp.wb;
    w.bord[px[b + 2]] += -(-v * sn * l[0] + v * cs * l[1]);
    w.bord[px[b + 3]] += bk * 2.0 * v * wv * wv - (cs * l[0] + sn * l[1] + tn / sp.wb * l[2]);
    w.bord[px[b + 4]] += bk * 2.0 * p[4] - v / sp.wb * sec2 * l[2];
    w.bord[px[b + 5]] += bk * 2.0 * p[5] - l[3];
    w.bord[px[b + 6]] += bk * 2.0 * v * v * wv - l[4];
    double f[5];
    f_ct(p, sp.wb, f);
    const int r = d.rO + 5 * q;
    for (int cc = 0; cc < 5; ++cc) {
      for (int j = 0; j < kPts; ++j) set2(Bd, pc[r + cc], px[7 * (i * kPts + j) + cc], 0.0 + sp.A[j][k]);
      w.bord[pc[r + cc]] += -f[cc];
    }
    set2(Bd, pc[r + 0], px[b + 2], 0.0 + -dt * (-v * sn)); set2(Bd, pc[r + 0], px[b + 3], 0.0 + -dt * cs);
    set2(Bd, pc[r + 1], px[b + 2], 0.0 + -dt * (v * cs)); set2(Bd, pc[r + 1], px[b + 3], 0.0 + -dt * sn);
    set2(Bd, pc[r + 2], px[b + 3], 0.0 + -dt * tn / sp.wb); set2(Bd, pc[r + 2], px[b + 4], 0.0 + -dt * v / sp.wb * sec2);
    set2(Bd, pc[r + 3], px[b + 5], 0.0 + -dt); set2(Bd, pc[r + 4], px[b + 6], 0.0 + -dt);
    if (k == 0 && il >= 1) for (int cc = 0; cc < 7; ++cc) { set2(Bd, pc[d.rC + 7 * (i - va - 1) + cc], px[b + cc], 1.0); set2(Bd, pc[d.rC + 7 * (i - va - 1) + cc], px[b - 7 + cc], -1.0); }
    for (int j = 0; j < sp.n_obs; ++j) {
      double A[4][2], bb[4], V[4][2], sep[2], gr[2][3];
      obstacle(sp, j, A, bb, V);
      const int sl = w.sel[q * sp.n_obs + j], fc = (sl >> 4) & 3;
      cfz::rows_for<true>(A, bb, V, p[0], p[1], cs, sn, sp.g, sl, sep, gr);
      for (int rr = 0; rr < 2; ++rr) {
        const int row = d.rR + q * d.nr + 2 * j + rr, sk = d.sO + q * d.nr + 2 * j + rr;
        const double S = w.sig[sk] + delta + sp.reg_primal, D = 1.0 / (1.0 / S + sp.reg_dual), t = w.c[row] - prox * sp.reg_dual * nu[row] + w.r1[sk] / S;
        double *cd = w.cond + (size_t)(q * d.nr + 2 * j + rr) * 5;
        cd[0] = gr[rr][0]; cd[1] = gr[rr][1]; cd[2] = gr[rr][2]; cd[3] = D; cd[4] = t;
        for (int a = 0; a < 3; ++a) w.rhs[px[b + a]] -= D * t * gr[rr][a];
        dg[0] += D * gr[rr][0] * gr[rr][0]; p01 += D * gr[rr][0] * gr[rr][1]; p02 += D * gr[rr][0] * gr[rr][2];
        dg[1] += D * gr[rr][1] * gr[rr][1]; p12 += D * gr[rr][1] * gr[rr][2];
        dg[2] += D * gr[rr][2] * gr[rr][2];
        const double nr_ = nu[row];
        const int vtx = rr == 0 ? ((sl >> 2) & 3) : (sl & 3);
        if ((sl >> 6) == 3) {
          const double bx = (vtx == 0 || vtx == 3) ? sp.g[0] : -sp.g[2], by = (vtx < 2) ? sp.g[1] : -sp.g[3];
          const double rbx = cs * bx - sn * by, rby = sn * bx + cs * by, a0 = gr[rr][0], a1 = gr[rr][1];
          const double nq = CFZC_VV_TANGENTIAL * nr_ / sep[0], t2 = a1 * rby + a0 * rbx;
          dg[0] += nq * a1 * a1; dg[1] += nq * a0 * a0; p01 += -nq * a1 * a0;
          p02 += -nq * a1 * t2; p12 += nq * a0 * t2;
          dg[2] += nq * t2 * t2 - nr_ * (a0 * rbx + a1 * rby);
        } else if ((sl >> 6) == 1) {
          const double bx = (vtx == 0 || vtx == 3) ? sp.g[0] : -sp.g[2], by = (vtx < 2) ? sp.g[1] : -sp.g[3];
          dg[2] += nr_ * -(gr[rr][0] * (cs * bx - sn * by) + gr[rr][1] * (sn * bx + cs * by));
        } else {
          const double gf = sp.g[fc];
          p02 += nr_ * -gr[rr][1]; p12 += nr_ * gr[rr][0];
          dg[2] += nr_ * -(sep[rr] + gf);
        }
      }
    }
    for (int c = 0; c < 7; ++c) bnd(Bd, px[b + c], px[b + c]) = dg[c];
    set2(Bd, px[b], px[b + 1], p01); set2(Bd, px[b], px[b + 2], p02); set2(Bd, px[b + 1], px[b + 2], p12);
  }
  if (!once) CFZP_LANE_FOR(q, 0, d.np - 1) {  // every entry written here belongs to point q alone
    const int i = q / kPts, k = q - i * kPts, b = 7 * q, va = veh_of_interval(d, i), il = i - d.off[va];
    const double *p = X + b, *l = nu + d.rO + 5 * q;
    const double cs = cos(p[2]), sn = sin(p[2]), tn = tan(p[4]), sec2 = 1.0 + tn * tn, v = p[3], wv = p[6], bk = sp.B[k];
    // objective curvature B_k dt e''
    bnd(Bd, px[b + 3], px[b + 3]) += bk * dt * 2.0 * wv * wv; bnd(Bd, px[b + 6], px[b + 6]) += bk * dt * 2.0 * v * v;
    put(Bd, px[b + 3], px[b + 6], bk * dt * 4.0 * v * wv);
    bnd(Bd, px[b + 4], px[b + 4]) += bk * dt * 2.0; bnd(Bd, px[b + 5], px[b + 5]) += bk * dt * 2.0;
    // ODE curvature: rows are sum_j A z - dt f, multipliers l
    const double l0 = -l[0] * dt, l1 = -l[1] * dt, l2 = -l[2] * dt;
    bnd(Bd, px[b + 2], px[b + 2]) += l0 * (-v * cs) + l1 * (-v * sn);
    put(Bd, px[b + 2], px[b + 3], l0 * (-sn) + l1 * cs);
    put(Bd, px[b + 3], px[b + 4], l2 * sec2 / sp.wb);
    bnd(Bd, px[b + 4], px[b + 4]) += l2 * 2.0 * v * tn * sec2 / sp.wb;
    // border: d2L / dp ddt = B_k e' - (f_z, f_u)' l
    w.bord[px[b + 2]] += -(-v * sn * l[0] + v * cs * l[1]);
    w.bord[px[b + 3]] += bk * 2.0 * v * wv * wv - (cs * l[0] + sn * l[1] + tn / sp.wb * l[2]);
    w.bord[px[b + 4]] += bk * 2.0 * p[4] - v / sp.wb * sec2 * l[2];
    w.bord[px[b + 5]] += bk * 2.0 * p[5] - l[3];
    w.bord[px[b + 6]] += bk * 2.0 * v * v * wv - l[4];
    // Jacobian of this point's ODE rows: A[j][k] on the states of every point j of the interval, -dt f' on its own
    double f[5];
    f_ct(p, sp.wb, f);
    const int r = d.rO + 5 * q;
    for (int cc = 0; cc < 5; ++cc) {
      for (int j = 0; j < kPts; ++j) put(Bd, pc[r + cc], px[7 * (i * kPts + j) + cc], sp.A[j][k]);
      w.bord[pc[r + cc]] += -f[cc];
    }
    put(Bd, pc[r + 0], px[b + 2], -dt * (-v * sn)); put(Bd, pc[r + 0], px[b + 3], -dt * cs);
    put(Bd, pc[r + 1], px[b + 2], -dt * (v * cs)); put(Bd, pc[r + 1], px[b + 3], -dt * sn);
    put(Bd, pc[r + 2], px[b + 3], -dt * tn / sp.wb); put(Bd, pc[r + 2], px[b + 4], -dt * v / sp.wb * sec2);
    put(Bd, pc[r + 3], px[b + 5], -dt); put(Bd, pc[r + 4], px[b + 6], -dt);
    if (k == 0 && il >= 1) for (int cc = 0; cc < 7; ++cc) { put(Bd, pc[d.rC + 7 * (i - va - 1) + cc], px[b + cc], 1.0); put(Bd, pc[d.rC + 7 * (i - va - 1) + cc], px[b - 7 + cc], -1.0); }
    // collision rows: gradient, slack, curvature
    for (int j = 0; j < sp.n_obs; ++j) {
      double A[4][2], bb[4], V[4][2], sep[2], gr[2][3];
      obstacle(sp, j, A, bb, V);
      const int sl = w.sel[q * sp.n_obs + j], fc = (sl >> 4) & 3;
      cfz::rows_for<true>(A, bb, V, p[0], p[1], cs, sn, sp.g, sl, sep, gr);
      for (int rr = 0; rr < 2; ++rr) {
        const int row = d.rR + q * d.nr + 2 * j + rr, sk = d.sO + q * d.nr + 2 * j + rr;
        const double S = w.sig[sk] + delta + sp.reg_primal, D = 1.0 / (1.0 / S + sp.reg_dual), t = w.c[row] - prox * sp.reg_dual * nu[row] + w.r1[sk] / S;
        double *cd = w.cond + (size_t)(q * d.nr + 2 * j + rr) * 5;
        cd[0] = gr[rr][0]; cd[1] = gr[rr][1]; cd[2] = gr[rr][2]; cd[3] = D; cd[4] = t;
        for (int a = 0; a < 3; ++a) {
          w.rhs[px[b + a]] -= D * t * gr[rr][a];
          for (int c2 = a; c2 < 3; ++c2) put(Bd, px[b + a], px[b + c2], D * gr[rr][a] * gr[rr][c2]);
        }
        const double nr_ = nu[row];
        const int vtx = rr == 0 ? ((sl >> 2) & 3) : (sl & 3);
        if ((sl >> 6) == 3) {  // distance r of two vertices, n = (a0,a1): tau tau' / r - n.(R b_v) e_psi e_psi', tau = (-a1, a0, t.dw)
          const double bx = (vtx == 0 || vtx == 3) ? sp.g[0] : -sp.g[2], by = (vtx < 2) ? sp.g[1] : -sp.g[3];
          const double rbx = cs * bx - sn * by, rby = sn * bx + cs * by, a0 = gr[rr][0], a1 = gr[rr][1];
          const double nq = CFZC_VV_TANGENTIAL * nr_ / sep[0], t2 = a1 * rby + a0 * rbx;  // sep[0] = r (the second slot's value carries the margin)
          bnd(Bd, px[b], px[b]) += nq * a1 * a1; bnd(Bd, px[b + 1], px[b + 1]) += nq * a0 * a0; put(Bd, px[b], px[b + 1], -nq * a1 * a0);
          put(Bd, px[b], px[b + 2], -nq * a1 * t2); put(Bd, px[b + 1], px[b + 2], nq * a0 * t2);
          bnd(Bd, px[b + 2], px[b + 2]) += nq * t2 * t2 - nr_ * (a0 * rbx + a1 * rby);
        } else if ((sl >> 6) == 1) {  // polygon face (a0,a1), body vertex: d2/dpsi2 = -A_f.(R b_v)
          const double bx = (vtx == 0 || vtx == 3) ? sp.g[0] : -sp.g[2], by = (vtx < 2) ? sp.g[1] : -sp.g[3];
          bnd(Bd, px[b + 2], px[b + 2]) += nr_ * -(gr[rr][0] * (cs * bx - sn * by) + gr[rr][1] * (sn * bx + cs * by));
        } else {  // body face: d2/dx dpsi = -a1, d2/dy dpsi = a0, d2/dpsi2 = -(sep + g_f)
          const double gf = sp.g[fc];
          put(Bd, px[b], px[b + 2], nr_ * -gr[rr][1]); put(Bd, px[b + 1], px[b + 2], nr_ * gr[rr][0]);
          bnd(Bd, px[b + 2], px[b + 2]) += nr_ * -(sep[rr] + gf);
        }
      }
    }
  }
  CFZP_SYNC();
  CFZP_LANE_FOR(t, 0, d.nchk - 1) {
    const int b = 7 * chk_point(sp, d, t), r = d.rT + 8 * t, s = d.sT + 8 * t;
    const double cs = cos(X[b + 2]), sn = sin(X[b + 2]);
    const double *cb = chk_cell(sp, d, t, 0), *cf = chk_cell(sp, d, t, 1);
    double curv = 0.0;
    if (jstruct_mode(sp)) {
      // condensed like the collision rows (the slack enters its row with +1):  S ds + dnu = -r_s,  g'dp + ds - reg_dual dnu = -c'
      //   =>  dnu = D (g'dp + t),  ds = -(r_s + dnu) / S,   D = 1 / (1/S + reg_dual),  t = c' - r_s / S
      double P3[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, rp[3] = {0, 0, 0};
      for (int rr = 0; rr < 8; ++rr) {
        const double *cc_ = rr < 4 ? cb : cf; const int q_ = rr & 3;
        const double gr[3] = {cc_[2 * q_], cc_[2 * q_ + 1], rr < 4 ? 0.0 : sp.wb * (-cc_[2 * q_] * sn + cc_[2 * q_ + 1] * cs)};
        const double S = w.sig[s + rr] + delta + sp.reg_primal, D = 1.0 / (1.0 / S + sp.reg_dual), tt = w.c[r + rr] - prox * sp.reg_dual * nu[r + rr] - w.r1[s + rr] / S;
        double *cd = w.condt + (size_t)(8 * t + rr) * 5;
        cd[0] = gr[0]; cd[1] = gr[1]; cd[2] = gr[2]; cd[3] = D; cd[4] = tt;
        for (int a = 0; a < 3; ++a) { rp[a] += D * tt * gr[a]; for (int c2 = a; c2 < 3; ++c2) P3[a][c2] += D * gr[a] * gr[c2]; }
        if (rr >= 4) curv += nu[r + rr] * sp.wb * (-cc_[2 * q_] * cs - cc_[2 * q_ + 1] * sn);
      }
      for (int a = 0; a < 3; ++a) { w.rhs[px[b + a]] -= rp[a]; for (int c2 = a; c2 < 3; ++c2) put(Bd, px[b + a], px[b + c2], P3[a][c2]); }
      bnd(Bd, px[b + 2], px[b + 2]) += curv;
      continue;
    }
    for (int rr = 0; rr < 4; ++rr) {
      put(Bd, pc[r + rr], px[b], cb[2 * rr]); put(Bd, pc[r + rr], px[b + 1], cb[2 * rr + 1]); put(Bd, pc[r + rr], px[s + rr], 1.0);
      put(Bd, pc[r + 4 + rr], px[b], cf[2 * rr]); put(Bd, pc[r + 4 + rr], px[b + 1], cf[2 * rr + 1]);
      put(Bd, pc[r + 4 + rr], px[b + 2], sp.wb * (-cf[2 * rr] * sn + cf[2 * rr + 1] * cs));
      put(Bd, pc[r + 4 + rr], px[s + 4 + rr], 1.0);
      curv += nu[r + 4 + rr] * sp.wb * (-cf[2 * rr] * cs - cf[2 * rr + 1] * sn);
    }
    bnd(Bd, px[b + 2], px[b + 2]) += curv;
  }
  CFZP_LANE_FOR(a, 0, sp.V - 1) {
    const int bl = 7 * (kPts * d.off[a + 1] - 1);
    for (int i = 0; i < 4; ++i) { if (once) set2(Bd, pc[d.rF + 5 * a + i], px[bl + 3 + i], 1.0); else put(Bd, pc[d.rF + 5 * a + i], px[bl + 3 + i], 1.0); }
    if (sp.has_final[a]) { if (once) set2(Bd, pc[d.rF + 5 * a + 4], px[bl + 2], 1.0); else put(Bd, pc[d.rF + 5 * a + 4], px[bl + 2], 1.0); }
  }
  CFZP_SYNC();
  // pair rows, condensed like the obstacle rows but into the poses of both vehicles (6 x 6: D g g' + nu H); one pair after
  // the other, because two pairs may share a point
  for (int e = 0; e < sp.n_pairs; ++e) {
    CFZP_LANE_FOR(r, 0, d.poff[e + 1] - d.poff[e] - 1) {
      int qa, qb;
      pair_points(sp, d, e, r, &qa, &qb);
      const int pp = d.poff[e] + r, sl = w.sel[d.np * sp.n_obs + pp];
      int at[6];
      for (int a = 0; a < 3; ++a) { at[a] = px[7 * qa + a]; at[3 + a] = px[7 * qb + a]; }
      for (int rr = 0; rr < 2; ++rr) {
        double gr[6], H[6][6];
        pair_row<true>(X + 7 * qa, X + 7 * qb, sp.g, sl >> 6, (sl >> 4) & 3, rr == 0 ? (sl >> 2) & 3 : sl & 3, gr, H);
        const int row = d.rP + 2 * pp + rr, sk = d.sP + 2 * pp + rr;
        const double S = w.sig[sk] + delta + sp.reg_primal, D = 1.0 / (1.0 / S + sp.reg_dual), nr_ = nu[row], t = w.c[row] - prox * sp.reg_dual * nr_ + w.r1[sk] / S;
        double *cd = w.condp + (size_t)(2 * pp + rr) * 8;
        for (int a = 0; a < 6; ++a) cd[a] = gr[a];
        cd[6] = D; cd[7] = t;
        if (w.pm != nullptr) {  // joint structured elimination: the block stays beside the band (cfz_jstruct.inl), complete and symmetric
          double *pm = w.pm + (size_t)pp * 36;
          for (int a = 0; a < 6; ++a) {
            w.rhs[at[a]] -= D * t * gr[a];
            for (int c2 = 0; c2 < 6; ++c2) { const double v = D * gr[a] * gr[c2] + nr_ * H[a < c2 ? a : c2][a < c2 ? c2 : a]; pm[6 * a + c2] = rr == 0 ? v : pm[6 * a + c2] + v; }
          }
          continue;
        }
        for (int a = 0; a < 6; ++a) {
          w.rhs[at[a]] -= D * t * gr[a];
          for (int c2 = a; c2 < 6; ++c2) { const double v = D * gr[a] * gr[c2] + nr_ * H[a][c2]; if (v != 0.0) put(Bd, at[a], at[c2], v); }
        }
      }
    }
    CFZP_SYNC();
  }
  double nn = 0.0;
  for (int a = 0; a < sp.V; ++a) nn += 2.0 * sp.N[a] * sp.N[a];
  return nn + delta + sp.reg_primal;  // d2L/ddt2
}

// ---- banded LU with partial pivoting, runtime half-bandwidth, factor once / substitute many -----------------------
CFZC_PIECE int band_factor(const Band &B, int n, int *ipiv) {
  const int kl = B.kb, ku = B.kb, kv = kl + ku, ld = B.ld;
  double *ab = B.ab;
  int ju = 0;
  for (int j = 0; j < n; ++j) {
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    double *cj = ab + (size_t)j * ld;
    int jp = 0; double best = -1.0;  // first largest entry of the column, searched by all lanes together
    CFZP_LANE_FOR(i, 0, km) { const double a = fabs(cj[kv + i]); if (a > best) { best = a; jp = i; } }
#if defined(__HIP_DEVICE_COMPILE__)
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off); const int oj = __shfl_xor(jp, off);
      if (ob > best || (ob == best && oj < jp)) { best = ob; jp = oj; }
    }
    if (blockDim.x > 64) {  // combine the wavefronts' candidates (first largest wins, as in the serial loop)
      __shared__ double pb[16];
      __shared__ int pj[16];
      const int nw = (int)(blockDim.x >> 6);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) { pb[threadIdx.x >> 6] = best; pj[threadIdx.x >> 6] = jp; }
      __syncthreads();
      best = pb[0]; jp = pj[0];
      for (int i = 1; i < nw; ++i) if (pb[i] > best || (pb[i] == best && pj[i] < jp)) { best = pb[i]; jp = pj[i]; }
    }
#endif
    ipiv[j] = j + jp;
    if (!(best > 0.0)) return 1;
    const int reach = j + ku + jp; ju = ju > (reach < n - 1 ? reach : n - 1) ? ju : (reach < n - 1 ? reach : n - 1);
    if (jp != 0) {
      CFZP_LANE_FOR(q, j, ju) { double *cq = ab + (size_t)q * ld; const double t = cq[kv + j - q]; cq[kv + j - q] = cq[kv + j + jp - q]; cq[kv + j + jp - q] = t; }
      CFZP_SYNC();
    }
    const double inv = 1.0 / cj[kv];
    CFZP_SYNC();
    CFZP_LANE_FOR(i, 1, km) cj[kv + i] *= inv;
    CFZP_SYNC();
    CFZP_LANE_FOR(q, j + 1, ju) {
      double *cq = ab + (size_t)q * ld;
      const double u = cq[kv + j - q];
      if (u != 0.0) for (int i = 1; i <= km; ++i) cq[kv + j + i - q] -= cj[kv + i] * u;
    }
    CFZP_SYNC();
  }
  return 0;
}
// two right-hand sides at once (the KKT residual and the dt border)
#if defined(__HIP_DEVICE_COMPILE__)
// The elimination for a band too wide for LDS (the joint plan: half-bandwidth ~100 per vehicle), from global memory with all
// wavefronts of the workgroup.
// barrier that orders LDS traffic only: the global stores still in flight are not waited for (what is read after it was
// either not written in this phase or is forwarded through LDS)
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Built around the number of dependent global-memory round trips per pivot (the
// band does not fit the L2, a trip costs ~2.5 us): (1) the pivot column is read once, every thread keeping its own row;
// (2) one pass over the trailing columns does the row swap AND fetches the pivot-row multipliers u -- the value that
// moves to row j+jp is forwarded through LDS instead of being re-read; (3) only the rows whose multiplier l is not
// zero and the columns whose u is not zero are updated (both lists compacted in LDS; typically ~40 rows x ~80 columns
// of 298 x 596), one row per lane, sixteen columns per batch.  Needs blockDim.x > kb.
// A function of its own (registers of its own: inlined into the solver its loops reloaded spilled values from scratch at
// every pivot).  Such a function must not NAME any LDS on this toolchain (see the top of this file), so its LDS arrays
// are declared by the inlined wrapper below and come in as address-space-3 pointers.
using cfzb::lds_f64; using cfzb::lds_i32; using cfzb::glb_f64; using cfzb::glb_i32;  // address-space-qualified pointers, see cfz_band.inl
__device__ __attribute__((noinline)) int band_factor_wide2_core(glb_f64 *ab, int kb, int ld, int n, glb_i32 *ipiv, lds_f64 *ulds, lds_f64 *alds,
                                                                lds_f64 *lval, lds_i32 *cols, lds_i32 *lrow, lds_f64 *pb, lds_f64 *pbv,
                                                                lds_i32 *pj, lds_i32 *cnt, lds_f64 *ptk) {
  constexpr int NB = 16;
  const int kl = kb, kv = 2 * kb, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  int ju = 0;
  for (int j = 0; j < n; ++j) {
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    glb_f64 *cj = ab + (size_t)j * ld;
    long long tp0 = tick();
    // (1) pivot search; thread i keeps row j+i of the pivot column
    const double own = tid <= km ? cj[kv + tid] : 0.0, diag = cj[kv];
    double best = tid <= km ? fabs(own) : -1.0, bv = own;
    int jp = tid;
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off), ov = __shfl_xor(bv, off); const int oj = __shfl_xor(jp, off);
      if (ob > best || (ob == best && oj < jp)) { best = ob; jp = oj; bv = ov; }
    }
    if (lane == 0) { pb[wave] = best; pj[wave] = jp; pbv[wave] = bv; }
    if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
    __syncthreads();
    best = pb[0]; jp = pj[0]; bv = pbv[0];
    for (int i = 1; i < nw; ++i) if (pb[i] > best || (pb[i] == best && pj[i] < jp)) { best = pb[i]; jp = pj[i]; bv = pbv[i]; }
    if (tid == 0) ipiv[j] = j + jp;
    if (!(best > 0.0)) return 1;
    const int reach = j + kl + jp < n - 1 ? j + kl + jp : n - 1;
    ju = ju > reach ? ju : reach;
    const int nq = ju - j;
    { const long long t1 = tick(); if (tid == 0) ptk[0] += (double)(t1 - tp0); tp0 = t1; }
    // (2a) the pivot column: diagonal <- pivot value, multipliers l = (swapped column) / pivot; nonzero rows compacted
    {
      const double inv = 1.0 / bv;
      const double l = (tid >= 1 && tid <= km) ? (tid == jp ? diag : own) * inv : 0.0;
      if (tid == 0) cj[kv] = bv;
      if (tid >= 1 && tid <= km) cj[kv + tid] = l;
      const unsigned long long mask = __ballot(l != 0.0);
      int base = 0;
      if (lane == 0 && mask) base = __hip_atomic_fetch_add(cnt + 1, (int)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = __shfl(base, 0);
      if (l != 0.0) { const int at = base + __popcll(mask & ((1ull << lane) - 1ull)); lrow[at] = tid; lval[at] = l; }
    }
    // (2b) trailing columns: swap rows j and j+jp, u = new row j; the value now in row j+jp goes to LDS as well
    for (int t0 = 0; t0 < nq; t0 += nt) {
      const int t = t0 + tid;
      double u = 0.0;
      if (t < nq) {
        glb_f64 *cq = ab + (size_t)(j + 1 + t) * ld + (kv - 1 - t);  // row j of column j+1+t; row j+i at cq[i]
        const double a = cq[0];
        u = jp ? cq[jp] : a;
        if (jp) { cq[0] = u; cq[jp] = a; }
        ulds[t] = u; alds[t] = a;
      }
      const unsigned long long mask = __ballot(u != 0.0);
      int base = 0;
      if (lane == 0 && mask) base = __hip_atomic_fetch_add(cnt, (int)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      base = __shfl(base, 0);
      if (u != 0.0) cols[base + __popcll(mask & ((1ull << lane) - 1ull))] = t;
    }
    lds_barrier();
    { const long long t1 = tick(); if (tid == 0) ptk[1] += (double)(t1 - tp0); tp0 = t1; }
    // (3) rank-1 update of (nonzero rows) x (nonzero columns)
    const int nc = cnt[0], nr = cnt[1];
    for (int r0 = 0; r0 < nr; r0 += 64) {
      const bool mine = r0 + lane < nr;
      const int i = mine ? lrow[r0 + lane] : 0;
      const double l = mine ? lval[r0 + lane] : 0.0;
      for (int c0 = wave * NB; c0 < nc; c0 += nw * NB) {
        double x[NB], u[NB];
        int tt[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          tt[b] = cols[c0 + b < nc ? c0 + b : nc - 1];
          u[b] = ulds[tt[b]];
          const glb_f64 *cq = ab + (size_t)(j + 1 + tt[b]) * ld + (kv - 1 - tt[b]);
          x[b] = cq[i];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (c0 + b < nc && mine) {
            const double xv = (jp && i == jp) ? alds[tt[b]] : x[b];  // row j+jp was rewritten in (2b): take it from LDS
            ab[(size_t)(j + 1 + tt[b]) * ld + (kv - 1 - tt[b]) + i] = xv - l * u[b];
          }
        }
      }
    }
    __syncthreads();
    { const long long t1 = tick(); if (tid == 0) ptk[2] += (double)(t1 - tp0); }
  }
  return 0;
}

__device__ inline int band_factor_wide2(const Band &B, int n, int *ipiv, long long *ptk) {
  __shared__ double ulds[2 * kWideMaxKb], alds[2 * kWideMaxKb], lval[kWideMaxKb + 64];
  __shared__ int cols[2 * kWideMaxKb], lrow[kWideMaxKb + 64];
  __shared__ double pb[16], pbv[16], tks[3];
  __shared__ int pj[16];
  __shared__ int cnt[2];
  if (threadIdx.x == 0) { tks[0] = 0.0; tks[1] = 0.0; tks[2] = 0.0; }
  __syncthreads();
  const int fail = band_factor_wide2_core((glb_f64 *)B.ab, B.kb, B.ld, n, (glb_i32 *)ipiv, cfzb::opaque((lds_f64 *)ulds), cfzb::opaque((lds_f64 *)alds), cfzb::opaque((lds_f64 *)lval), cfzb::opaque((lds_i32 *)cols), cfzb::opaque((lds_i32 *)lrow),
                                          cfzb::opaque((lds_f64 *)pb), cfzb::opaque((lds_f64 *)pbv), cfzb::opaque((lds_i32 *)pj), cfzb::opaque((lds_i32 *)cnt), cfzb::opaque((lds_f64 *)tks));
  __syncthreads();
  for (int i = 0; i < 3; ++i) ptk[i] += (long long)tks[i];
  return fail;
}

// The same elimination P pivots at a time: same pivots, same arithmetic per entry as band_factor_wide2, so the same factor.
// One pivot at a time costs three dependent global round trips and reads row j of every column within reach (a cache line
// per column).  Here, per panel:
// (1) thread i takes row j0+i of the P panel columns into registers;
// (2) the P pivot steps run there: search by DPP + one LDS exchange between the wavefronts, the two rows of a swap pass
//     through LDS (everybody needs the new pivot row), scale and update in registers; the panel goes back to the band and
//     its multipliers to LDS (PL: P x (kb + P) doubles of the kernel's dynamic LDS);
// (3) of the trailing columns only the 2 P entries the panel's pivot rows and swaps can touch are gathered (64 / 2P columns
//     per load, all loads of a wavefront in flight together), and only as far right as the row at that position reaches
//     (ext[], below: a third of the band's reach on the joint plan);
// (4) a column with a nonzero among them is read ONCE (each lane rows lane + 64 s, only down to the last row a multiplier
//     or swap of the panel touches), taken through the P swaps and updates in registers (v_readlane; multipliers from PL)
//     and written ONCE, eight columns in flight per wavefront.
__device__ __forceinline__ double lane_get(double v, int l) {  // v of lane l (l uniform): v_readlane, no LDS round trip
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double lane_set(double v, int l, double x) {  // v with lane l replaced by the uniform x
  return (int)(threadIdx.x & 63) == l ? x : v;
}
template <int P, int SMAX>
__device__ __attribute__((noinline)) int band_factor_panel_core(glb_f64 *ab, int kb, int ld, int n, glb_i32 *ipiv, lds_f64 *PL, lds_f64 *pb,
                                                                lds_i32 *pj, lds_i32 *meta, lds_i32 *ext, lds_f64 *ptk, glb_f64 *b1, glb_f64 *b2) {
  constexpr int T = 2 * P, CG = 64 / T, NCH = 64 / CG < 16 ? 64 / CG : 16, CB = 4, RPW = (P + 7) / 8;  // RPW: needs >= 8 wavefronts
  constexpr unsigned long long TMASK = T == 64 ? ~0ull : ((1ull << (T & 63)) - 1ull);
  const int kl = kb, kv = 2 * kb, RS = kb + P;  // needs RS <= 64 SMAX and RS <= blockDim.x
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  lds_i32 *jps = meta, *juk = meta + P, *kms = meta + 2 * P;
  lds_f64 *cand = pb + 16;  // [2][9][P + 1]: per parity, the candidate row of each wavefront (+ 1 / its entry) and row k
  const int nwa = (RS + 63) >> 6;  // wavefronts that hold rows of a panel
  const int hw = nwa < nw ? nw - 1 : 0;  // the helper wavefront: the last one if it holds no rows, else the first
  lds_i32 *cint = pj, *rtop = meta + 3 * P;
  // ext[r & 1023]: no entry of the row now at position r lies right of this column.  A row enters with the extent of the
  // assembled matrix, read off its COLUMN (the matrix is symmetric: put() writes both triangles) SC columns ahead of the
  // elimination, where no update or swap has reached yet; swaps exchange extents, an updated row inherits the pivot row's.
  const int SC = 2 * kl + 2 * P;
  auto scan_rows = [&](int r0, int r1) {
    for (int r = r0 + wave; r < r1; r += nw) {
      if (r >= n) break;
      int last = 0;
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) {
        const int o = 1 + lane + 64 * s_;
        const double x = (o <= kl && r + o < n) ? ab[(size_t)r * ld + kv + o] : 0.0;
        const unsigned long long mk = __ballot(x != 0.0);
        if (mk) last = 64 * s_ + 64 - __clzll((long long)mk);
      }
      if (lane == 0) ext[r & 1023] = r + last;
    }
  };
  scan_rows(0, SC);
  __syncthreads();
  int ju = 0;
  for (int j0 = 0; j0 < n; j0 += P) {
    const int pw = P < n - j0 ? P : n - j0;
    long long tp0 = tick();
    int myext = (tid < RS && j0 + tid < n) ? ext[(j0 + tid) & 1023] : 0;
    if (tid == 0) *rtop = 0;
    // (1) the panel in registers: thread i holds row j0+i of the P panel columns
    double v[P];
#pragma unroll
    for (int k = 0; k < P; ++k) v[k] = (k < pw && tid < RS && tid <= k + kl && j0 + tid < n) ? ab[(size_t)(j0 + k) * ld + kv + tid - k] : 0.0;
    // columns of the rows that come within reach with the next panel: fetched now (after the panel's own
    // columns: these lines come from HBM, and loads return in order), looked at after the panel's pivot steps
    double sx[RPW][SMAX];
#pragma unroll
    for (int q_ = 0; q_ < RPW; ++q_) {
      const int r = j0 + SC + wave + nw * q_;
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) {
        const int o = 1 + lane + 64 * s_;
        sx[q_][s_] = (wave + nw * q_ < P && o <= kl && r + o < n) ? ab[(size_t)r * ld + kv + o] : 0.0;
      }
    }
    // (2) the panel's P pivot steps: search by DPP + one LDS exchange between the wavefronts, the two rows that change
    // places go through LDS (everybody needs the new pivot row anyway), the update stays in registers
    bool touched = tid < pw;
    // A rolled loop (unrolled, this function is larger than the instruction cache): v[] is shifted down after every step, so
    // that v[0] is always the column being eliminated, and what is final leaves the registers at once -- the pivot row (U) and
    // the multipliers to the band, the multipliers also to PL, where later swaps of the panel are applied to them (the
    // trailing columns want them in the row order after all swaps; the band keeps them unswapped for the substitution).
#pragma nounroll
    for (int k = 0; k < pw; ++k) {
      const int j = j0 + k, km = (kl < n - 1 - j) ? kl : n - 1 - j;
      // one barrier per step: every wavefront posts its candidate (largest entry, its row of the panel, the reciprocal, its
      // extent) and row k posts itself, then everybody picks the winner; the buffers alternate with the parity of k
      lds_f64 *cd = cand + (k & 1) * 9 * (P + 1), *pbk = pb + (k & 1) * 8;
      lds_i32 *ci = cint + (k & 1) * 32;
      if (wave < nwa) {  // wavefronts without rows of the panel only keep the barriers company
        const double a = (tid >= k && tid <= k + km) ? fabs(v[0]) : -1.0;
        const double wb = cfz::wave_reduce<1>(a);
        const unsigned long long hit = __ballot(a == wb && a >= 0.0);
        const int fl = hit ? __ffsll((long long)hit) - 1 : -1;
        if (lane == fl) {
#pragma unroll
          for (int c = 0; c < P; ++c) cd[wave * (P + 1) + c] = v[c];
          cd[wave * (P + 1) + P] = 1.0 / v[0];
          ci[8 + wave] = myext;
        }
        if (lane == 0) { pbk[wave] = wb; ci[wave] = hit ? wave * 64 + fl : 0x7fffffff; }
        if (tid == k) {
#pragma unroll
          for (int c = 0; c < P; ++c) cd[8 * (P + 1) + c] = v[c];
          ci[16] = myext;
        }
      }
      lds_barrier();
      double best = pbk[0];
      int p = ci[0], ws = 0;  // row of the pivot, relative to j0: the first of the largest; the wavefront it is in
      for (int i = 1; i < nwa; ++i) if (pbk[i] > best) { best = pbk[i]; p = ci[i]; ws = i; }
      const int ht = tid - hw * 64;  // the chores of a step (nothing here needs a row of the panel) go to a wavefront without rows
      if (ht == 0) ipiv[j] = j0 + p;
      if (!(best > 0.0)) return 1;
      const int jp = p - k;
      const int reach = j + kl + jp < n - 1 ? j + kl + jp : n - 1;
      ju = ju > reach ? ju : reach;
      if (ht == 0) { jps[k] = jp; juk[k] = ju; kms[k] = km; }
      // the pivot row is final: entry (j, j + t) by helper thread t
      if (ht >= 0 && ht < pw - k) ab[(size_t)(j + ht) * ld + kv - ht] = cd[ws * (P + 1) + ht];
      // multipliers of the earlier steps follow the swap (helper thread c: column c of PL)
      if (ht >= 0 && ht < k && jp) { const double t_ = PL[ht * RS + k]; PL[ht * RS + k] = PL[ht * RS + p]; PL[ht * RS + p] = t_; }
      if (wave >= nwa) continue;
      const int ek_ = ci[8 + ws];  // extent of the pivot row
      double u[P];
#pragma unroll
      for (int c = 0; c < P; ++c) u[c] = cd[ws * (P + 1) + c];
      if (tid == k) myext = ek_;
      else if (tid == p) {
#pragma unroll
        for (int c = 0; c < P; ++c) v[c] = cd[8 * (P + 1) + c];
        myext = ci[16];
      }
      double l = 0.0;
      if (tid > k && tid <= k + km) {
        l = v[0] * cd[ws * (P + 1) + P];
        ab[(size_t)j * ld + kv + tid - k] = l;
#pragma unroll
        for (int c = 1; c < P; ++c) v[c] = v[c] - l * u[c];
        if (l != 0.0) { myext = myext > ek_ ? myext : ek_; touched = true; }
      }
      if (tid < RS) PL[k * RS + tid] = l;
      if (tid == p) touched = true;
#pragma unroll
      for (int c = 0; c < P - 1; ++c) v[c] = v[c + 1];
      v[P - 1] = 0.0;
    }
    {  // rows beyond the last one a multiplier or a swap of this panel touches are left alone by the trailing columns
      const unsigned long long tm = __ballot(touched);
      if (lane == 0 && tm) __hip_atomic_fetch_max(rtop, wave * 64 + 63 - __clzll((long long)tm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_barrier();
    { const long long t1 = tick(); if (tid == 0) ptk[0] += (double)(t1 - tp0); tp0 = t1; }
    // (3), (4) trailing columns c0 .. ju
#pragma unroll
    for (int q_ = 0; q_ < RPW; ++q_) {  // extents of the rows that come within reach with the next panel
      const int r = j0 + SC + wave + nw * q_;
      int last = 0;
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) {
        const unsigned long long mk = __ballot(sx[q_][s_] != 0.0);
        if (mk) last = 64 * s_ + 64 - __clzll((long long)mk);
      }
      if (lane == 0 && wave + nw * q_ < P && r < n) ext[r & 1023] = r + last;
    }
    const int c0 = j0 + pw, q = lane / T, e = lane % T, ek = e < P ? e : e - P;
    const int ejp = ek < pw ? jps[ek] : 0;
    const bool eok = ek < pw && (e < P || ejp != 0);
    const int er = j0 + (e < P ? ek : ek + ejp);  // absolute row of this lane's test entry
    const int eext = eok ? ext[er & 1023] : -1;   // ... and where that row ended when the panel began
    const int cend = (int)cfz::wave_reduce<1>((double)eext), cmax = cend < ju ? cend : ju, ncol = cmax - c0 + 1;
    const int rt = *rtop;
    {
      const int G = ncol > 0 ? (ncol + CG - 1) / CG : 0;  // (0: no trailing columns left; the right-hand sides below still go through)
      double tv[NCH];
      auto gather = [&](int g0) {  // the test entries of NCH groups of columns, all in flight together
#pragma unroll
        for (int mm = 0; mm < NCH; ++mm) {
          const int gi = g0 + mm * nw, c = c0 + gi * CG + q;
          const bool ok = gi < G && c <= eext && er >= c - kv;
          tv[mm] = ok ? ab[(size_t)c * ld + kv + er - c] : 0.0;
        }
      };
      gather(wave);  // ... while the row permutation is built
      // where the content of each row comes from once all swaps of the panel are done (the same for every column): the
      // identity taken through the P swaps, position i = lane + 64 s
      const int mjp = lane < pw ? jps[lane] : 0;
      int src[SMAX];
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) src[s_] = lane + 64 * s_;
#pragma nounroll
      for (int k = 0; k < pw; ++k) {  // (rolled: the unrolled body of this function does not fit the instruction cache)
        const int jp = __builtin_amdgcn_readlane(mjp, k);
        if (jp) {
          const int pp = k + jp, ps = pp >> 6, pl = pp & 63, sk = __builtin_amdgcn_readlane(src[0], k);
#pragma unroll
          for (int s_ = 0; s_ < SMAX; ++s_) {
            if (ps == s_) { const int sp_ = __builtin_amdgcn_readlane(src[s_], pl); if (lane == pl) src[s_] = sk; if (lane == k) src[0] = sp_; }
          }
        }
      }
      // (the right-hand sides taken along, see below: requested here, used after the columns)
      glb_f64 *bx = wave == nw - 1 ? b1 : (wave == nw - 2 ? b2 : nullptr);  // (either may be null)
      double y[SMAX];
#pragma unroll
      for (int s_ = 0; s_ < SMAX; ++s_) {
        const int i = lane + 64 * s_, r = j0 + src[s_];
        y[s_] = (bx != nullptr && i <= rt && r < n) ? bx[r] : 0.0;
      }
      for (int g0 = wave; g0 < G; g0 += nw * NCH) {
        if (g0 != wave) gather(g0);
        unsigned long long flag = 0ull;
#pragma unroll
        for (int mm = 0; mm < NCH; ++mm) {
          const unsigned long long mask = __ballot(tv[mm] != 0.0);
#pragma unroll
          for (int qq = 0; qq < CG; ++qq) if ((mask >> (qq * T)) & TMASK) flag |= 1ull << (mm * CG + qq);
        }
        { const long long t1 = tick(); if (tid == 0) ptk[1] += (double)(t1 - tp0); tp0 = t1; }
        // flagged columns, CB at a time, the next CB in flight while these are updated
        double cur[CB][SMAX], nxt[CB][SMAX];
        int cc[CB], nc[CB];
#pragma unroll
        for (int x = 0; x < CB; ++x) {
          cc[x] = -1;
          if (flag) { const int b = __ffsll((long long)flag) - 1; flag &= flag - 1ull; cc[x] = c0 + (g0 + (b / CG) * nw) * CG + (b % CG); }
#pragma unroll
          for (int s_ = 0; s_ < SMAX; ++s_) {
            const int i = lane + 64 * s_, r = j0 + src[s_], c = cc[x];  // read through the panel's row permutation
            cur[x][s_] = (c >= 0 && i <= rt && r >= c - kv) ? ab[(size_t)c * ld + kv + r - c] : 0.0;
          }
        }
        while (cc[0] >= 0) {
#pragma unroll
          for (int x = 0; x < CB; ++x) {
            nc[x] = -1;
            if (flag) { const int b = __ffsll((long long)flag) - 1; flag &= flag - 1ull; nc[x] = c0 + (g0 + (b / CG) * nw) * CG + (b % CG); }
#pragma unroll
            for (int s_ = 0; s_ < SMAX; ++s_) {
              const int i = lane + 64 * s_, r = j0 + src[s_], c = nc[x];
              nxt[x][s_] = (c >= 0 && i <= rt && r >= c - kv) ? ab[(size_t)c * ld + kv + r - c] : 0.0;
            }
          }
#pragma nounroll
          for (int k = 0; k < pw; ++k) {
            double Lk[SMAX];  // multipliers of step k for this lane's rows, in the row order after all swaps
#pragma unroll
            for (int s_ = 0; s_ < SMAX; ++s_) { const int i = lane + 64 * s_; Lk[s_] = i < RS ? PL[k * RS + i] : 0.0; }
#pragma unroll
            for (int x = 0; x < CB; ++x) {
              const double u = lane_get(cur[x][0], k);
              if (cc[x] >= 0 && u != 0.0) {
#pragma unroll
                for (int s_ = 0; s_ < SMAX; ++s_) cur[x][s_] = cur[x][s_] - Lk[s_] * u;
              }
            }
          }
#pragma unroll
          for (int x = 0; x < CB; ++x) {
            if (cc[x] >= 0) {
              const int c = cc[x];
#pragma unroll
              for (int s_ = 0; s_ < SMAX; ++s_) {
                const int i = lane + 64 * s_, r = j0 + i;
                if (i <= rt && r >= c - kv) ab[(size_t)c * ld + kv + r - c] = cur[x][s_];
              }
            }
            cc[x] = nc[x];
#pragma unroll
            for (int s_ = 0; s_ < SMAX; ++s_) cur[x][s_] = nxt[x][s_];
          }
        }
      }
      // The two right-hand sides of the solve that follows (b1, b2; may be null) are taken along as two more columns: L^-1 P is
      // applied to them panel by panel, and the substitution is left with the backward sweep.
      if (bx != nullptr) {  // (one each for the last two wavefronts)
#pragma nounroll
        for (int k = 0; k < pw; ++k) {
          const double u = lane_get(y[0], k);
          if (u != 0.0) {
#pragma unroll
            for (int s_ = 0; s_ < SMAX; ++s_) { const int i = lane + 64 * s_; y[s_] = y[s_] - (i < RS ? PL[k * RS + i] : 0.0) * u; }
          }
        }
#pragma unroll
        for (int s_ = 0; s_ < SMAX; ++s_) {
          const int i = lane + 64 * s_, r = j0 + i;
          if (i <= rt && r < n) bx[r] = y[s_];
        }
      }
    }
    __syncthreads();
    if (tid >= pw && tid < RS && j0 + tid < n) ext[(j0 + tid) & 1023] = myext;
    lds_barrier();
    { const long long t1 = tick(); if (tid == 0) ptk[2] += (double)(t1 - tp0); }
  }
  return 0;
}

#ifndef CFZ_PANEL
#define CFZ_PANEL 16  // pivots per panel (four-vehicle plan: 8 is 7 % slower, 32 is 14 % slower: every pivot step carries 32-wide rows)
#endif
#ifndef CFZ_NO_PANEL
#define CFZ_NO_PANEL 0
#endif
#ifndef CFZ_FORCE_SBIG
#define CFZ_FORCE_SBIG 0  // diagnostic builds: the wide-register instantiation (kb > 304) whatever kb is; passes the planning GPU tests
#endif
// lds: the kernel's dynamic LDS (free during the elimination; the substitution keeps its right-hand side there)
__device__ inline int band_factor_panel(const Band &B, int n, int *ipiv, long long *ptk, double *lds, double *b1, double *b2) {
  __shared__ double pb[16 + 18 * (CFZ_PANEL + 1)], tks[3];
  __shared__ int pj[64], meta[3 * CFZ_PANEL + 4], ext[1024];
  if (threadIdx.x == 0) { tks[0] = 0.0; tks[1] = 0.0; tks[2] = 0.0; }
  __syncthreads();
  constexpr int SBIG = (kWideMaxKb + CFZ_PANEL + 63) / 64;
  const int fail = (B.kb + CFZ_PANEL <= 320 && !CFZ_FORCE_SBIG)
      ? band_factor_panel_core<CFZ_PANEL, 5>((glb_f64 *)B.ab, B.kb, B.ld, n, (glb_i32 *)ipiv, cfzb::opaque((lds_f64 *)lds), cfzb::opaque((lds_f64 *)pb), cfzb::opaque((lds_i32 *)pj), cfzb::opaque((lds_i32 *)meta), cfzb::opaque((lds_i32 *)ext), cfzb::opaque((lds_f64 *)tks), (glb_f64 *)b1, (glb_f64 *)b2)
      : band_factor_panel_core<CFZ_PANEL, SBIG>((glb_f64 *)B.ab, B.kb, B.ld, n, (glb_i32 *)ipiv, cfzb::opaque((lds_f64 *)lds), cfzb::opaque((lds_f64 *)pb), cfzb::opaque((lds_i32 *)pj), cfzb::opaque((lds_i32 *)meta), cfzb::opaque((lds_i32 *)ext), cfzb::opaque((lds_f64 *)tks), (glb_f64 *)b1, (glb_f64 *)b2);
  __syncthreads();
  for (int i = 0; i < 3; ++i) ptk[i] += (long long)tks[i];
  return fail;
}

// Substitution for the wide band with the right-hand side in LDS (the kernel's dynamic LDS, n doubles; one right-hand side
// after the other) and the factor's columns fetched eight pivots ahead, so that a pivot costs two LDS barriers instead of
// two global-memory round trips.  Needs blockDim.x > kb.
__device__ inline void band_substitute_wide(const Band &B, int n, const int *ipiv, double *b, double *b2) {
  extern __shared__ double wlds[];
  constexpr int CH = 8;
  const int kl = B.kb, kv = 2 * B.kb, ld = B.ld, tid = threadIdx.x, nt = blockDim.x;
  const double *ab = B.ab;
  for (int pass = 0; pass < 2; ++pass) {
    double *v = pass ? b2 : b;
    for (int t = tid; t < n; t += nt) wlds[t] = v[t];
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += CH) {  // L y = P b
      double Lr[CH]; int pv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j0 + c, km = j < n ? ((kl < n - 1 - j) ? kl : n - 1 - j) : 0;
        Lr[c] = tid < km ? ab[(size_t)j * ld + kv + 1 + tid] : 0.0;
        pv[c] = j < n ? ipiv[j] : j;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j0 + c;
        if (j < n) {
          const int km = (kl < n - 1 - j) ? kl : n - 1 - j, p = pv[c];
          if (p != j) {
            if (tid == 0) { const double t = wlds[j]; wlds[j] = wlds[p]; wlds[p] = t; }
            lds_barrier();
          }
          const double bj = wlds[j];
          if (bj != 0.0) {
            if (tid < km) wlds[j + 1 + tid] -= Lr[c] * bj;
            lds_barrier();
          }
        }
      }
    }
    for (int j1 = n - 1; j1 >= 0; j1 -= CH) {  // U x = y
      double Ur[CH][2], dg[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j1 - c;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int off = tid + nt * t, i = j - kv + off;
          Ur[c][t] = (j >= 0 && off < kv && i >= 0) ? ab[(size_t)j * ld + off] : 0.0;
        }
        dg[c] = j >= 0 ? ab[(size_t)j * ld + kv] : 1.0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j1 - c;
        if (j >= 0) {
          const double bj = wlds[j] / dg[c];  // row j: final since the barrier of the step before; rewritten (x_j) after this step's
          if (bj != 0.0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const int off = tid + nt * t, i = j - kv + off;
              if (off < kv && i >= 0) wlds[i] -= Ur[c][t] * bj;
            }
            lds_barrier();
            if (tid == 0) wlds[j] = bj;
          }
        }
      }
    }
    for (int t = tid; t < n; t += nt) v[t] = wlds[t];
    __syncthreads();
  }
}
// Both right-hand sides in one sweep, the rows a chunk of CH pivots works on in REGISTERS (thread i = row j0+i; the backward
// sweep's window is kv + CH rows, two per thread): a pivot step is one LDS write by the thread that owns the pivot row, one
// barrier and one broadcast read, instead of a read-modify-write of the whole window in LDS and two barriers, per right-hand
// side.  Between chunks the rows move CH threads along through a staging array; the right-hand sides themselves stay in
// global memory (CH rows enter and leave per chunk).  Same operations in the same order as band_substitute.
// Needs blockDim.x >= kb + CH and 2 blockDim.x >= 2 kb + CH.
template <int CH>
__device__ __attribute__((noinline)) void band_substitute_regs_core(const glb_f64 *ab, int kb, int ld, int n, const glb_i32 *ipiv, glb_f64 *b1,
                                                                    glb_f64 *b2, lds_f64 *slot, lds_f64 *stage, lds_f64 *tri, bool fwd_done) {
  const int kl = kb, kv = 2 * kb, tid = threadIdx.x, nt = blockDim.x;
  if (!fwd_done) {  // L y = P b
    const int RS = kl + CH;
    double y1 = (tid < RS && tid < n) ? b1[tid] : 0.0, y2 = (tid < RS && tid < n) ? b2[tid] : 0.0;
    for (int j0 = 0; j0 < n; j0 += CH) {
      double Lc[CH]; int pv[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j0 + c, km = j < n ? ((kl < n - 1 - j) ? kl : n - 1 - j) : 0, o = tid - c;  // row j0+tid = j + o
        Lc[c] = (o >= 1 && o <= km) ? ab[(size_t)j * ld + kv + o] : 0.0;
        pv[c] = j < n ? ipiv[j] : j;
      }
      const int rn = j0 + CH + tid;  // this thread's row in the next chunk; the last CH threads fetch theirs now
      double f1 = 0.0, f2 = 0.0;
      if (tid >= RS - CH && tid < RS && rn < n) { f1 = b1[rn]; f2 = b2[rn]; }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (j0 + c < n) {
          const int rel = pv[c] - j0;
          if (tid == rel) { slot[4 * c] = y1; slot[4 * c + 1] = y2; }
          if (tid == c && rel != c) { slot[4 * c + 2] = y1; slot[4 * c + 3] = y2; }
          lds_barrier();
          const double a1 = slot[4 * c], a2 = slot[4 * c + 1];
          if (tid == c) { y1 = a1; y2 = a2; }
          else if (tid == rel) { y1 = slot[4 * c + 2]; y2 = slot[4 * c + 3]; }
          y1 = y1 - Lc[c] * a1; y2 = y2 - Lc[c] * a2;  // Lc is zero outside rows j+1 .. j+km
        }
      }
      if (tid < CH && j0 + tid < n) { b1[j0 + tid] = y1; b2[j0 + tid] = y2; }
      if (tid < RS) { stage[tid] = y1; stage[RS + tid] = y2; }
      lds_barrier();
      if (tid < RS - CH) { y1 = stage[tid + CH]; y2 = stage[RS + tid + CH]; } else { y1 = f1; y2 = f2; }
    }
  }
  __syncthreads();  // y is read back from global memory by other threads
  {  // U x = y: chunk columns j1, j1-1, ..; window position i = row - lo, lo = j1 - CH + 1 - kv; positions tid and tid + nt
    const int W = kv + CH;
    double z1[2], z2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = tid + nt * t, r = n - 1 - CH + 1 - kv + i;
      z1[t] = (i < W && r >= 0 && r < n) ? b1[r] : 0.0; z2[t] = (i < W && r >= 0 && r < n) ? b2[r] : 0.0;
    }
    for (int j1 = n - 1; j1 >= 0; j1 -= CH) {
      const int lo = j1 - CH + 1 - kv;
      double Uc[CH][2], dg[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int j = j1 - c;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int i = tid + nt * t, r = lo + i;
          Uc[c][t] = (j >= 0 && i < W && r >= 0 && r >= j - kv && r < j) ? ab[(size_t)j * ld + kv + r - j] : 0.0;
        }
        dg[c] = j >= 0 ? ab[(size_t)j * ld + kv] : 1.0;
      }
      const int rf = lo - CH + tid;  // the rows that enter with the next chunk: positions 0 .. CH-1 of its window
      double g1 = 0.0, g2 = 0.0;
      if (tid < CH && rf >= 0) { g1 = b1[rf]; g2 = b2[rf]; }
      // The CH x CH triangle of the chunk's own rows is solved by one wavefront (lane c = row j1 - c: the owners hand their rows of
      // the triangle and their right-hand sides over through LDS), then every row above takes its CH updates at once: two barriers
      // per chunk instead of one per pivot.  Same operations in the same order as one pivot at a time.
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + nt * t;
        if (i >= kv && i < W) {
          const int c = W - 1 - i;
#pragma unroll
          for (int c2 = 0; c2 < CH; ++c2) tri[c * CH + c2] = Uc[c2][t];
          tri[CH * CH + c] = z1[t]; tri[CH * CH + CH + c] = z2[t];
        }
      }
      lds_barrier();
      if (tid < 64) {
        const int c = tid < CH ? tid : 0;
        double Tc[CH], y1 = tid < CH ? tri[CH * CH + c] : 0.0, y2 = tid < CH ? tri[CH * CH + CH + c] : 0.0, dd = dg[0];
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) { Tc[c2] = tid < CH ? tri[c * CH + c2] : 0.0; if (tid == c2) dd = dg[c2]; }
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
          const double x1 = lane_get(y1 / dd, c2), x2 = lane_get(y2 / dd, c2);
          y1 = y1 - Tc[c2] * x1; y2 = y2 - Tc[c2] * x2;  // Tc[c2] is zero from row c2 on
        }
        if (tid < CH) { tri[CH * CH + 2 * CH + c] = y1 / dd; tri[CH * CH + 3 * CH + c] = y2 / dd; }
      }
      lds_barrier();
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const double a1 = tri[CH * CH + 2 * CH + c], a2 = tri[CH * CH + 3 * CH + c];
#pragma unroll
        for (int t = 0; t < 2; ++t) { z1[t] = z1[t] - Uc[c][t] * a1; z2[t] = z2[t] - Uc[c][t] * a2; }  // Uc is zero from row j on
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + nt * t;
        if (i >= kv && i < W) { const int c = W - 1 - i; z1[t] = tri[CH * CH + 2 * CH + c]; z2[t] = tri[CH * CH + 3 * CH + c]; }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + nt * t, r = lo + i;
        if (i >= kv && i < W && r >= 0) { b1[r] = z1[t]; b2[r] = z2[t]; }  // the chunk's own rows: final
        if (i < W) { stage[i] = z1[t]; stage[W + i] = z2[t]; }
      }
      lds_barrier();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = tid + nt * t;
        if (i >= CH && i < W) { z1[t] = stage[i - CH]; z2[t] = stage[W + i - CH]; }
        else if (i < CH) { z1[t] = g1; z2[t] = g2; }
      }
      lds_barrier();
    }
  }
  __syncthreads();
}

__device__ inline void band_substitute_regs(const Band &B, int n, const int *ipiv, double *b, double *b2, bool fwd_done) {
  constexpr int CH = 16;
  __shared__ double slot[4 * CH], stage[2 * (2 * kWideMaxKb + CH)], tri[CH * CH + 4 * CH];
  band_substitute_regs_core<CH>((const glb_f64 *)B.ab, B.kb, B.ld, n, (const glb_i32 *)ipiv, (glb_f64 *)b, (glb_f64 *)b2, cfzb::opaque((lds_f64 *)slot), cfzb::opaque((lds_f64 *)stage), cfzb::opaque((lds_f64 *)tri), fwd_done);
}
#endif

CFZC_PIECE void band_substitute(const Band &B, int n, const int *ipiv, double *b, double *b2) {
  const int kl = B.kb, kv = 2 * B.kb, ld = B.ld;
  const double *ab = B.ab;
#if defined(__HIP_DEVICE_COMPILE__)
  const bool first = threadIdx.x == 0;  // the swaps and the division are not idempotent: one thread does them
#else
  const bool first = true;
#endif
  for (int j = 0; j < n; ++j) {
    const int km = (kl < n - 1 - j) ? kl : n - 1 - j, p = ipiv[j];
    if (p != j) {
      if (first) { const double t = b[j]; b[j] = b[p]; b[p] = t; const double t2 = b2[j]; b2[j] = b2[p]; b2[p] = t2; }
      CFZP_SYNC();
    }
    const double bj = b[j], cj = b2[j];
    if (bj != 0.0 || cj != 0.0) {
      CFZP_LANE_FOR(i, 1, km) { const double l = ab[(size_t)j * ld + kv + i]; b[j + i] -= l * bj; b2[j + i] -= l * cj; }
      CFZP_SYNC();
    }
  }
  for (int j = n - 1; j >= 0; --j) {
    const double dg = ab[(size_t)j * ld + kv];
    const double bj = b[j] / dg, cj = b2[j] / dg;
    CFZP_SYNC();
    if (first) { b[j] = bj; b2[j] = cj; }
    const int lo = j - kv > 0 ? j - kv : 0;
    if (bj != 0.0 || cj != 0.0) CFZP_LANE_FOR(i, lo, j - 1) { const double u = ab[(size_t)j * ld + kv + i - j]; b[i] -= u * bj; b2[i] -= u * cj; }
    CFZP_SYNC();
  }
}

CFZC_PIECE double barrier_obj(const CSpec &sp, const CWork &w, const double *X, double mu) {
  const CDims d = cdims(sp);
  double s = 0.0, bad = 0.0;
  lane_for_loads<3>(d.n, [&](int i) { LV q; q.v[0] = w.xl[i]; q.v[1] = w.xu[i]; q.v[2] = X[i]; return q; },
                    [&](int, const LV &q) {
                      const bool hl = q.v[0] > -1e300, hu = q.v[1] < 1e300;
                      const double dl = q.v[2] - q.v[0], du = q.v[1] - q.v[2];
                      if (hl && !(dl > 0.0)) bad = 1.0;
                      if (hl && dl > 0.0) s += log(dl);
                      if (hu && !(du > 0.0)) bad = 1.0;
                      if (hu && du > 0.0) s += log(du);
                    });
  if (bmax(bad) > 0.0) return INFINITY;
  return objective(sp, X) - mu * bsum(s);
}

// Hand-over of a block's two (slack, bound multiplier, row multiplier) triples when its working set changes from code o to
// code n (sep: the values of the new rows).  The block stands for ONE constraint of the reference, dist(body, polygon) >= dmin,
// whose value is continuous across a change of the closest features; restarting its rows at every change (slack pushed to
// bound_push, z = mu / slack) made an active contact that flips between two certificates jump by bound_push in violation and
// lose its multiplier each time: a limit cycle (vehicle_2 past a pillar corner, period 7, measured).  So:
//   a row that keeps its (face, vertex) identity keeps its triple;
//   else the tighter of the new rows takes over the triple of the tighter old row if the residual that leaves is small;
//   any other row starts afresh at slack = max(sep - dmin, min(bound_push, max(mu, 1e-8))), z = mu / slack, nu = -z.
constexpr double kHandover = 1e-2;
CFZP_FN void handover(int o, int n, const double sep[2], double dmin, double mu, double bound_push, double s[2], double z[2], double nu[2]) {
  const double so[2] = {s[0], s[1]}, zo[2] = {z[0], z[1]}, no[2] = {nu[0], nu[1]};
  const int ov[2] = {(o >> 2) & 3, o & 3}, nv[2] = {(n >> 2) & 3, n & 3};
  int src[2] = {-1, -1};
  bool used[2] = {false, false};
  if ((o >> 4) == (n >> 4) && (n >> 6) != 3)
    for (int r = 0; r < 2; ++r)
      for (int q = 0; q < 2; ++q) if (src[r] < 0 && !used[q] && nv[r] == ov[q]) { src[r] = q; used[q] = true; }
  const int io = so[0] <= so[1] ? 0 : 1, in = sep[0] <= sep[1] ? 0 : 1;
  if (src[in] < 0 && !used[io] && fabs(sep[in] - dmin - so[io]) <= kHandover) { src[in] = io; used[io] = true; }
  const double push = fmin(bound_push, fmax(mu, 1e-8));
  for (int r = 0; r < 2; ++r) {
    if (src[r] >= 0) { s[r] = so[src[r]]; z[r] = zo[src[r]]; nu[r] = no[src[r]]; }
    else { const double sg = fmax(sep[r] - dmin, push); s[r] = sg; z[r] = mu / sg; nu[r] = -mu / sg; }
  }
#if defined(CFZC_TRACE)
  if (fmax(zo[0], zo[1]) > 1e-2) printf("   handover %d -> %d  sep %.4e %.4e | old s %.3e %.3e z %.3e %.3e nu %.3e %.3e | src %d %d | new s %.3e %.3e z %.3e %.3e\n", o, n, sep[0] - dmin, sep[1] - dmin, so[0], so[1], zo[0], zo[1], no[0], no[1], src[0], src[1], s[0], s[1], z[0], z[1]);
#endif
}

// refresh the working set at the poses of X; a block whose (face, vertices) change hands its rows over (handover)
// returns true if any block changed
CFZC_PIECE bool refresh_working_set(const CSpec &sp, const CWork &w, double *X, double mu, bool first) {
  const CDims d = cdims(sp);
  double chg = 0.0;
  // In the joint plan a face block turns vertex-vertex only beyond a margin of 0.1 mm while the barrier parameter is coarse
  // (cfz::select_from, vv_enter): measured on the 254-plan launch of configs[3], two plans whose blocks flipped between the two
  // certificates of one contact at every iterate needed 160 and 174 iterations (the others at most 66) and tripled the launch.
  // From mu < 1e-4 on (the last barrier problems), and in the single plans, a block changes at once.
  const double vv_enter = (sp.V > 1 && mu >= 1e-4) ? 1e-4 : 0.0;
  CFZP_LANE_FOR(q, 0, d.np - 1) {
    const double *p = X + 7 * q;
    double sn, cs;
    sincos(p[2], &sn, &cs);
    for (int j = 0; j < sp.n_obs; ++j) {
      double A[4][2], b[4], V[4][2], sep[2];
      obstacle(sp, j, A, b, V);
      const int old = first ? 0 : w.sel[q * sp.n_obs + j];
      const int nw = cfz::select_rows(A, b, V, p[0], p[1], cs, sn, sp.g, old, sp.vv_rows, vv_enter);
      if (nw != old) {
        chg = 1.0;
        w.sel[q * sp.n_obs + j] = (unsigned char)nw;
        cfz::rows_for<false>(A, b, V, p[0], p[1], cs, sn, sp.g, nw, sep, nullptr);
        const int sk = d.sO + q * d.nr + 2 * j, rk = d.rR + q * d.nr + 2 * j;
        if (first) { X[sk] = sep[0] - sp.dmin; X[sk + 1] = sep[1] - sp.dmin; }  // pushed inside the bound afterwards
        else {
          double s_[2] = {X[sk], X[sk + 1]}, z_[2] = {w.zl[sk], w.zl[sk + 1]}, n_[2] = {w.nu[rk], w.nu[rk + 1]};
          handover(old, nw, sep, sp.dmin, mu, sp.bound_push, s_, z_, n_);
          for (int r = 0; r < 2; ++r) { X[sk + r] = s_[r]; w.zl[sk + r] = z_[r]; w.nu[rk + r] = n_[r]; }
        }
      }
    }
  }
  CFZP_LANE_FOR(pp, 0, d.npp - 1) {
    int e = 0;
    while (pp >= d.poff[e + 1]) ++e;
    int qa, qb;
    pair_points(sp, d, e, pp - d.poff[e], &qa, &qb);
    const double *pa = X + 7 * qa, *pb = X + 7 * qb;
    double A[4][2], b[4], V[4][2];
    veh_polygon(pb, sp.g, A, b, V);
    const int old = first ? 0 : w.sel[d.np * sp.n_obs + pp];
    const int nw = cfz::select_rows(A, b, V, pa[0], pa[1], cos(pa[2]), sin(pa[2]), sp.g, old, sp.vv_rows, vv_enter);
    if (nw != old) {
      chg = 1.0;
      w.sel[d.np * sp.n_obs + pp] = (unsigned char)nw;
      double sep[2];
      for (int r = 0; r < 2; ++r) sep[r] = pair_value(pa, pb, sp.g, nw, r);
      const int sk = d.sP + 2 * pp, rk = d.rP + 2 * pp;
      if (first) { X[sk] = sep[0] - sp.dmin; X[sk + 1] = sep[1] - sp.dmin; }
      else {
        double s_[2] = {X[sk], X[sk + 1]}, z_[2] = {w.zl[sk], w.zl[sk + 1]}, n_[2] = {w.nu[rk], w.nu[rk + 1]};
        handover(old, nw, sep, sp.dmin, mu, sp.bound_push, s_, z_, n_);
        for (int r = 0; r < 2; ++r) { X[sk + r] = s_[r]; w.zl[sk + r] = z_[r]; w.nu[rk + r] = n_[r]; }
      }
    }
  }
  CFZP_SYNC();
  return bmax(chg) != 0.0;
}

// X: guess for the 7 variables of every point (vehicles back to back) followed by dt; solution out (same layout).
// out_i = iterations, status; out_d = cost, err, mu, phase timers.  kb: half-bandwidth the caller sized the slab for
// (half_bandwidth() of the ordering).  MODE (GPU only; 0 = the generic elimination everywhere, the CPU build): 2 = eight
// wavefronts: the panel elimination if its multipliers fit the dynamic LDS (lds_doubles), else band_factor_wide2; lds_rhs: doubles
// of the dynamic LDS a right-hand side may occupy (band_substitute_wide, the fallback substitution), 0 = it does not fit.
// (MODE 1, one wavefront per single plan with the elimination in an LDS window, was retired in round 4: see cfz_planning.hip.)
}  // namespace cfzc
#include "cfz_struct.inl"
#include "cfz_jstruct.inl"
namespace cfzc {

template <int MODE>
CFZP_FN void solve_colloc(const CSpec &sp, double *X, double *slab, int kb, int *out_i, double *out_d, int lds_doubles, int lds_rhs = 0) {
  const CDims d = cdims(sp);
  const CWork w = carve(sp, kb, slab);
  // the structured eliminations never factor the band in place: no room for fill (a third less to clear and to stream per assembly)
  const bool compact = jstruct_mode(sp);
  const Band Bd = {w.ab, kb, compact ? 2 * kb + 1 : 3 * kb + 1, compact ? kb : 2 * kb};
  const int n = d.n, m = d.m;
  const double prox = (sp.no_prox & 1) ? 0.0 : 1.0;
  build_order(sp, w.posx, w.posc);
  CFZP_SYNC();
  // no_prox bit 2: the structured elimination of cfz_jstruct.inl (joint and single plans; a layout it does not know fails with status 3)
  const bool jstructured = jstruct_mode(sp);  // (the caller sized the slab with kb = kCB: half_bandwidth())
  JWork JW = {};
  if (jstructured) { JW = jstruct_carve(sp, w.sw); jstruct_setup(sp, d, w, JW); band_clear(sp, Bd); }  // (the band's only clear: see assemble)
  CFZP_LANE_FOR(i, 0, n - 1) { w.xl[i] = i >= d.sO ? 0.0 : -INFINITY; w.xu[i] = INFINITY; w.x[i] = i <= d.iDt ? X[i] : 0.0; }
  CFZP_SYNC();
  CFZP_LANE_FOR(q, 0, d.np - 1) {
    const int col[6] = {0, 1, 3, 4, 5, 6};
    for (int c = 0; c < 6; ++c) { w.xl[7 * q + col[c]] = sp.bounds[2 * c]; w.xu[7 * q + col[c]] = sp.bounds[2 * c + 1]; }
  }
  CFZP_SYNC();
  refresh_working_set(sp, w, w.x, sp.mu_init, true);
  constraints(sp, w.sel, w.x, w.c);
  CFZP_LANE_FOR(i, 0, 8 * d.nchk - 1) w.x[d.sT + i] = -w.c[d.rT + i];
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
  const int nb = (int)bsum(nbd);
  CFZP_LANE_FOR(i, 0, m - 1) w.nu[i] = 0.0;
  CFZP_SYNC();
  double mu = sp.mu_init, filt_mu = -1.0, theta_min = -1.0, theta_max = -1.0, err0 = INFINITY;
  const double mu_floor = fmin(sp.tol, sp.compl_inf_tol) / (sp.kappa_eps + 1.0);
  double filt[64][2]; int nfilt = 0;
  double delta_last = 0.0;  // the last nonzero primal perturbation of the inertia correction (Algorithm IC)
  double delta_floor = 0.0; int delta_retry = 0;  // a failed line search is repeated with a larger perturbation (below)
  bool mu_forced = false;  // the last iteration ended without a step and lowered mu instead
  int status = 1, iter = 0;
  long long tk[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = tick(), ta;
  for (iter = 0; iter <= sp.max_iter; ++iter) {
    // a layout the structured elimination does not know (jstruct_setup): every solve would fail and the inertia correction would climb its
    // whole ladder at every iterate before giving up -- the plan ends at once instead (the host routes what it can foresee to the band path)
    if (jstructured && JW.flag[1] != 0.0) { status = 3; break; }
    ta = tick();
    // rows of a changed working set belong to another problem: the filter starts afresh (as in the MPC step)
    if (iter > 0 && refresh_working_set(sp, w, w.x, mu, false)) nfilt = 0;
    constraints(sp, w.sel, w.x, w.c);
    gradient(sp, w.x, w.g);
    jt_nu(sp, w.sel, w.x, w.nu, w.r1);
    double theta = 0.0, cviol = 0.0, sum_nu = 0.0, sum_z = 0.0, dual_inf = 0.0;
    lane_for_loads<4>(m, [&](int i) { LV q; q.v[0] = w.c[i]; q.v[1] = w.nu[i]; return q; },
                      [&](int, const LV &q) { theta += fabs(q.v[0]); cviol = fmax(cviol, fabs(q.v[0])); sum_nu += fabs(q.v[1]); });
    theta = bsum(theta); cviol = bmax(cviol); sum_nu = bsum(sum_nu);
    if (theta_min < 0.0) { theta_min = 1e-4 * fmax(1.0, theta); theta_max = 1e4 * fmax(1.0, theta); }
    // one pass over the variables: multiplier sum, dual infeasibility, complementarity, and the barrier problem's complementarity at the
    // present mu (the first pass of the loop below)
    double cmp0 = 0.0, cm_first = 0.0;
    lane_for_loads<3>(n, [&](int i) { LV q; q.v[0] = w.zl[i]; q.v[1] = w.zu[i]; q.v[2] = w.g[i]; q.v[3] = w.r1[i]; q.v[4] = w.xl[i]; q.v[5] = w.xu[i]; q.v[6] = w.x[i]; return q; },
                      [&](int, const LV &q) {
                        sum_z += q.v[0] + q.v[1]; dual_inf = fmax(dual_inf, fabs(q.v[2] + q.v[3] - q.v[0] + q.v[1]));
                        if (q.v[4] > -1e300) { cmp0 = fmax(cmp0, fabs((q.v[6] - q.v[4]) * q.v[0])); cm_first = fmax(cm_first, fabs((q.v[6] - q.v[4]) * q.v[0] - mu)); }
                        if (q.v[5] < 1e300) { cmp0 = fmax(cmp0, fabs((q.v[5] - q.v[6]) * q.v[1])); cm_first = fmax(cm_first, fabs((q.v[5] - q.v[6]) * q.v[1] - mu)); }
                      });
    sum_z = bsum(sum_z); dual_inf = bmax(dual_inf);
    const double s_d = fmax(sp.s_max, (sum_nu + sum_z) / (double)(m + nb)) / sp.s_max, s_c = fmax(sp.s_max, sum_z / (double)nb) / sp.s_max;
    cmp0 = bmax(cmp0);
    err0 = fmax(dual_inf / s_d, fmax(cviol, cmp0 / s_c));
    if (!isfinite(err0)) { status = 3; break; }
    if (err0 <= sp.tol && dual_inf <= sp.dual_inf_tol && cviol <= sp.constr_viol_tol && cmp0 <= sp.compl_inf_tol) { status = 0; break; }
    if (iter == sp.max_iter) { status = 1; break; }  // (assigned here as well: the GPU build returned 0 for this exit without it)
    bool mu_eased = false;  // at most one such decrease per iteration
    bool cm_known = true;   // cm_first belongs to the present mu
    while (mu > mu_floor) {
      double cm = cm_first;
      if (!cm_known) {
        cm = 0.0;
        lane_for_loads<3>(n, [&](int i) { LV q; q.v[0] = w.zl[i]; q.v[1] = w.zu[i]; q.v[4] = w.xl[i]; q.v[5] = w.xu[i]; q.v[6] = w.x[i]; return q; },
                          [&](int, const LV &q) {
                            if (q.v[4] > -1e300) cm = fmax(cm, fabs((q.v[6] - q.v[4]) * q.v[0] - mu));
                            if (q.v[5] < 1e300) cm = fmax(cm, fabs((q.v[5] - q.v[6]) * q.v[1] - mu));
                          });
      }
      cm_known = false;
      cm = bmax(cm);
      if (fmax(dual_inf / s_d, fmax(cviol, cm / s_c)) <= sp.kappa_eps * mu) mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu)));
      else {
        // Everything but complementarity already meets the stopping rule of the NLP itself: only mu stands between this iterate and
        // termination, so it falls now instead of after the barrier problem's own, tighter, test (kappa_eps mu < tol at the last two
        // levels of a tol = 1e-2 solve).  Measured on the 254-plan launch of configs[3]: a plan at err 3e-3 after 28 iterations spent
        // 120 more at mu = 1.5e-4 driving the dual infeasibility from 3e-3 to 1.5e-3 through inertia corrections.
        if (!CFZC_NO_EASE && !mu_eased && fmax(dual_inf / s_d, cviol) <= sp.tol && dual_inf <= sp.dual_inf_tol && cviol <= sp.constr_viol_tol && cmp0 > sp.compl_inf_tol) {
          mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu))); mu_eased = true;
        }
        break;
      }
    }
    const double tau = fmax(sp.tau_min, 1.0 - mu);
    lane_for_loads<3>(n, [&](int i) { LV q; q.v[0] = w.zl[i]; q.v[1] = w.zu[i]; q.v[2] = w.g[i]; q.v[3] = w.r1[i]; q.v[4] = w.xl[i]; q.v[5] = w.xu[i]; q.v[6] = w.x[i]; return q; },
                      [&](int i, const LV &q) {
                        double gphi = q.v[2], s = 0.0;
                        if (q.v[4] > -1e300) { const double dl = q.v[6] - q.v[4]; gphi -= mu / dl; s += q.v[0] / dl; }
                        if (q.v[5] < 1e300) { const double du = q.v[5] - q.v[6]; gphi += mu / du; s += q.v[1] / du; }
                        w.g[i] = gphi; w.r1[i] = gphi + q.v[3]; w.sig[i] = s;
                      });
    CFZP_SYNC();
    double delta = delta_floor; bool have = false;
    tk[0] += tick() - ta;
    for (int tries = 0; tries < 60; ++tries) {
      ta = tick();
      lane_for_loads<4>(n, [&](int i) { LV q; q.k[0] = w.posx[i]; q.v[0] = w.r1[i]; return q; }, [&](int, const LV &q) { if (q.k[0] >= 0) w.rhs[q.k[0]] = -q.v[0]; });
      lane_for_loads<4>(m, [&](int i) { LV q; q.k[0] = w.posc[i]; q.v[0] = w.c[i]; q.v[1] = w.nu[i]; return q; },
                        [&](int, const LV &q) { if (q.k[0] >= 0) w.rhs[q.k[0]] = -q.v[0] + prox * sp.reg_dual * q.v[1]; });
      CFZP_SYNC();
      const double hdd = assemble(sp, w, Bd, delta);
      CFZP_LANE_FOR(i, 0, d.nk - 1) w.rhs2[i] = w.bord[i];
      CFZP_SYNC();
      tk[1] += tick() - ta; ta = tick();
      int fail;
      bool fwd_done = false;  // the elimination has already applied L^-1 P to both right-hand sides
      bool solved = false;
#if defined(__HIP_DEVICE_COMPILE__)
      if (jstructured) {  // (the dynamic LDS: 8 x kLuLdsWave doubles for the 64-row eliminations, sized by the host)
        extern __shared__ double wlds[];
        fail = sp.V == 1 ? jstruct_solve1(sp, d, w, JW, Bd, w.rhs, w.rhs2, tk + 6, wlds) : jstruct_solve(sp, d, w, JW, Bd, w.rhs, w.rhs2, tk + 6, wlds);
        solved = true;
      } else
#else
      if (jstructured) { fail = jstruct_solve(sp, d, w, JW, Bd, w.rhs, w.rhs2, tk + 6, nullptr); solved = true; } else
#endif
#if defined(__HIP_DEVICE_COMPILE__)
      if (MODE == 2 && blockDim.x >= 512 && blockDim.x >= kb + CFZ_PANEL && kb <= kWideMaxKb && CFZ_PANEL * (kb + CFZ_PANEL) <= lds_doubles && !CFZ_NO_PANEL && !(sp.no_prox & 2)) {
        extern __shared__ double wlds[];
        fail = band_factor_panel(Bd, d.nk, w.ipiv, tk + 6, wlds, w.rhs, w.rhs2);
        fwd_done = true;
      } else if (MODE == 2 && blockDim.x > kb && kb <= kWideMaxKb) fail = band_factor_wide2(Bd, d.nk, w.ipiv, tk + 6); else
#endif
      fail = band_factor(Bd, d.nk, w.ipiv);
      tk[2] += tick() - ta; ta = tick();
      if (!fail) {
        if (solved) { } else
#if defined(__HIP_DEVICE_COMPILE__)
        if (MODE == 2 && (int)blockDim.x >= kb + 16 && 2 * (int)blockDim.x >= 2 * kb + 16 && kb <= kWideMaxKb && !(sp.no_prox & 2)) band_substitute_regs(Bd, d.nk, w.ipiv, w.rhs, w.rhs2, fwd_done);
        else if (MODE == 2 && blockDim.x > kb && 2 * kb <= 2 * (int)blockDim.x && d.nk <= lds_rhs) band_substitute_wide(Bd, d.nk, w.ipiv, w.rhs, w.rhs2); else
#endif
        band_substitute(Bd, d.nk, w.ipiv, w.rhs, w.rhs2);
        tk[3] += tick() - ta;
        // bordered system: [K b; b' h] [y; s] = [r; r_dt]  ->  s = (r_dt - b'K^-1 r) / (h - b'K^-1 b)
        double bty = 0.0, btw = 0.0;
        lane_for_loads<4>(d.nk, [&](int i) { LV q; q.v[0] = w.bord[i]; q.v[1] = w.rhs[i]; q.v[2] = w.rhs2[i]; return q; },
                          [&](int, const LV &q) { bty += q.v[0] * q.v[1]; btw += q.v[0] * q.v[2]; });
        bty = bsum(bty); btw = bsum(btw);
        const double ddt = (-w.r1[d.iDt] - bty) / (hdd - btw);
        double curv = 0.0, dd = 0.0, bad = isfinite(ddt) ? 0.0 : 1.0;
        // (an item is two dependent loads -- its position, then the two right-hand sides there: the positions of U items travel together)
        lane_for_loads<4>(n, [&](int i) { LV q; const int px_ = w.posx[i]; q.k[0] = px_; q.v[0] = w.rhs[px_ >= 0 ? px_ : 0]; q.v[1] = w.rhs2[px_ >= 0 ? px_ : 0]; return q; },
                          [&](int i, const LV &q) { if (i == d.iDt || q.k[0] >= 0) w.dx[i] = i == d.iDt ? ddt : q.v[0] - q.v[1] * ddt; });
        lane_for_loads<4>(m, [&](int i) { LV q; const int pc_ = w.posc[i]; q.k[0] = pc_; q.v[0] = w.rhs[pc_ >= 0 ? pc_ : 0]; q.v[1] = w.rhs2[pc_ >= 0 ? pc_ : 0]; return q; },
                          [&](int i, const LV &q) { if (q.k[0] >= 0) w.dnu[i] = q.v[0] - q.v[1] * ddt; });
        CFZP_SYNC();
        lane_for_loads<4>(d.np * d.nr, [&](int r) {  // the condensed pairs, from the pose step of their point
          LV q; const double *cd = w.cond + (size_t)r * 5, *dp = w.dx + 7 * (r / d.nr);
          for (int c = 0; c < 5; ++c) q.v[c] = cd[c];
          q.v[5] = dp[0]; q.v[6] = dp[1]; q.v[7] = dp[2]; q.v[8] = w.sig[d.sO + r]; q.v[9] = w.r1[d.sO + r];
          return q;
        }, [&](int r, const LV &q) {
          const double S = q.v[8] + delta + sp.reg_primal;
          const double dn = q.v[3] * (q.v[0] * q.v[5] + q.v[1] * q.v[6] + q.v[2] * q.v[7] + q.v[4]);
          w.dnu[d.rR + r] = dn; w.dx[d.sO + r] = (dn - q.v[9]) / S;
        });
        if (w.condt != nullptr) CFZP_LANE_FOR(r, 0, 8 * d.nchk - 1) {  // the condensed tube rows, from the pose step of their checkpoint
          const double *cd = w.condt + (size_t)r * 5, *dp = w.dx + 7 * chk_point(sp, d, r / 8);
          const double S = w.sig[d.sT + r] + delta + sp.reg_primal;
          const double dn = cd[3] * (cd[0] * dp[0] + cd[1] * dp[1] + cd[2] * dp[2] + cd[4]);
          w.dnu[d.rT + r] = dn; w.dx[d.sT + r] = -(w.r1[d.sT + r] + dn) / S;
        }
        CFZP_LANE_FOR(r, 0, 2 * d.npp - 1) {  // pair rows: from the pose steps of both vehicles
          int e = 0;
          while (r / 2 >= d.poff[e + 1]) ++e;
          int qa, qb;
          pair_points(sp, d, e, r / 2 - d.poff[e], &qa, &qb);
          const double *cd = w.condp + (size_t)r * 8, *da = w.dx + 7 * qa, *db = w.dx + 7 * qb;
          const double S = w.sig[d.sP + r] + delta + sp.reg_primal;
          const double dn = cd[6] * (cd[0] * da[0] + cd[1] * da[1] + cd[2] * da[2] + cd[3] * db[0] + cd[4] * db[1] + cd[5] * db[2] + cd[7]);
          w.dnu[d.rP + r] = dn; w.dx[d.sP + r] = (dn - w.r1[d.sP + r]) / S;
        }
        CFZP_SYNC();
        lane_for_loads<4>(n, [&](int i) { LV q; q.v[0] = w.dx[i]; q.v[1] = w.r1[i]; return q; },
                          [&](int, const LV &q) { const double v = q.v[0]; if (!isfinite(v)) bad = 1.0; curv -= v * q.v[1]; dd += v * v; });
        lane_for_loads<4>(m, [&](int i) { LV q; q.v[0] = w.dnu[i]; q.v[1] = w.c[i]; q.v[2] = w.nu[i]; return q; },
                          [&](int, const LV &q) { const double v = q.v[0]; if (!isfinite(v)) bad = 1.0; curv += (q.v[1] - prox * sp.reg_dual * q.v[2]) * v - sp.reg_dual * v * v; });
        curv = bsum(curv); dd = bsum(dd); bad = bmax(bad);
        CFZP_SYNC();
        if (bad == 0.0 && curv >= sp.curv_kappa * dd) { have = true; break; }
      }
      // IPOPT's Algorithm IC (oracle/ipm.py next_delta_w): after delta = 0 a third of the last perturbation that worked (1e-4 if
      // there has been none), then x 8
      delta = delta == 0.0 ? (delta_last == 0.0 ? 1e-4 : fmax(1e-20, delta_last / 3.0)) : delta * 8.0;
      if (delta > 1e20) break;
    }
    if (!have) { status = 3; break; }
    if (delta > 0.0) delta_last = delta;
    ta = tick();
    double a_pri = 1.0, a_dual = 1.0, dphi = 0.0;
    lane_for_loads<3>(n, [&](int i) { LV q; q.v[0] = w.zl[i]; q.v[1] = w.zu[i]; q.v[2] = w.g[i]; q.v[3] = w.dx[i]; q.v[4] = w.xl[i]; q.v[5] = w.xu[i]; q.v[6] = w.x[i]; return q; },
                      [&](int i, const LV &q) {
                        const double dxi = q.v[3];
                        dphi += q.v[2] * dxi;
                        double dzl = 0.0, dzu = 0.0;
                        if (q.v[4] > -1e300) {
                          const double dl = q.v[6] - q.v[4];
                          dzl = mu / dl - q.v[0] - q.v[0] / dl * dxi;
                          if (dxi < 0.0) a_pri = fmin(a_pri, -tau * dl / dxi);
                          if (dzl < 0.0) a_dual = fmin(a_dual, -tau * q.v[0] / dzl);
                        }
                        if (q.v[5] < 1e300) {
                          const double du = q.v[5] - q.v[6];
                          dzu = mu / du - q.v[1] + q.v[1] / du * dxi;
                          if (dxi > 0.0) a_pri = fmin(a_pri, tau * du / dxi);
                          if (dzu < 0.0) a_dual = fmin(a_dual, -tau * q.v[1] / dzu);
                        }
                        w.dzl[i] = dzl; w.dzu[i] = dzu;
                      });
    a_pri = bmin(a_pri); a_dual = bmin(a_dual); dphi = bsum(dphi);
#if defined(CFZC_TRACE)
    {  // which bounds cut the step: the five smallest ratios (kind: p point variable 0..6, o obstacle slack, t tube slack, q pair slack)
      int bi[5] = {-1, -1, -1, -1, -1}; double bv[5] = {2, 2, 2, 2, 2};
      for (int i = 0; i < n; ++i) {
        double r_ = 2.0;
        if (w.xl[i] > -1e300 && w.dx[i] < 0.0) r_ = fmin(r_, -tau * (w.x[i] - w.xl[i]) / w.dx[i]);
        if (w.xu[i] < 1e300 && w.dx[i] > 0.0) r_ = fmin(r_, tau * (w.xu[i] - w.x[i]) / w.dx[i]);
        for (int k = 0; k < 5; ++k) if (r_ < bv[k]) { for (int k2 = 4; k2 > k; --k2) { bv[k2] = bv[k2 - 1]; bi[k2] = bi[k2 - 1]; } bv[k] = r_; bi[k] = i; break; }
      }
      printf("   blockers:");
      for (int k = 0; k < 5 && bi[k] >= 0; ++k) {
        const int i = bi[k];
        if (i < d.iDt) printf(" p[pt %d var %d] %.2e (x %.3f dx %.2e z %.1e/%.1e)", i / 7, i % 7, bv[k], w.x[i], w.dx[i], w.zl[i], w.zu[i]);
        else if (i < d.sT) printf(" o[pt %d row %d] %.2e (s %.2e dx %.2e z %.1e)", (i - d.sO) / d.nr, (i - d.sO) % d.nr, bv[k], w.x[i], w.dx[i], w.zl[i]);
        else if (i < d.sP) printf(" t[chk %d row %d] %.2e (s %.2e dx %.2e z %.1e)", (i - d.sT) / 8, (i - d.sT) % 8, bv[k], w.x[i], w.dx[i], w.zl[i]);
        else printf(" q[pp %d row %d] %.2e (s %.2e dx %.2e z %.1e)", (i - d.sP) / 2, (i - d.sP) % 2, bv[k], w.x[i], w.dx[i], w.zl[i]);
      }
      printf("\n");
    }
#endif
    CFZP_SYNC();
    const double phi0 = barrier_obj(sp, w, w.x, mu);
    if (filt_mu != mu) { nfilt = 0; filt_mu = mu; }
    double alpha = a_pri; bool accepted = false, f_type = false;
    for (int bt = 0; bt < sp.max_backtrack; ++bt) {
      lane_for_loads<4>(n, [&](int i) { LV q; q.v[0] = w.x[i]; q.v[1] = w.dx[i]; return q; }, [&](int i, const LV &q) { w.xt[i] = q.v[0] + alpha * q.v[1]; });
      CFZP_SYNC();
      constraints(sp, w.sel, w.xt, w.ct);
      double th_t = 0.0;
      lane_for_loads<4>(m, [&](int i) { LV q; q.v[0] = w.ct[i]; return q; }, [&](int, const LV &q) { th_t += fabs(q.v[0]); });
      th_t = bsum(th_t);
      const double ph_t = barrier_obj(sp, w, w.xt, mu);
      bool ok = isfinite(ph_t) && isfinite(th_t) && th_t <= theta_max && w.xt[d.iDt] > 0.0;
      if (ok) for (int q = 0; q < nfilt; ++q) if (th_t >= filt[q][0] && ph_t >= filt[q][1]) { ok = false; break; }
      f_type = false;
      if (ok) {
        const bool sw = theta <= theta_min && dphi < 0.0 && alpha * pow(-dphi, sp.s_phi) > sp.delta_sw * pow(theta, sp.s_theta);
        if (sw) { f_type = true; ok = ph_t <= phi0 + sp.eta_phi * alpha * dphi; }
        else ok = th_t <= (1.0 - sp.gamma_theta) * theta || ph_t <= phi0 - sp.gamma_phi * theta;
      }
      if (ok) { accepted = true; break; }
#if defined(CFZC_TRACE)
      if (bt < 6 || bt == sp.max_backtrack - 1) printf("   bt %d alpha %.3e th_t %.6e (theta %.6e) ph_t %.10e (phi0 %.10e) dphi %.3e nfilt %d\n", bt, alpha, th_t, theta, ph_t, phi0, dphi, nfilt);
#endif
      alpha *= 0.5;
    }
    bool nu_done = false;
    if (!accepted) {
      // The filter line search has failed: typically the iterate is feasible (theta tiny), the multipliers are still
      // off and the Newton step, which mostly corrects them, is no descent direction for the barrier function.  IPOPT
      // would enter its restoration phase here; this solver falls back on the primal-dual merit instead: the step (or a
      // fraction) is taken if it lowers max(scaled dual infeasibility, constraint violation) by a tenth, and the filter
      // starts afresh.
      const double e0 = fmax(dual_inf / s_d, cviol);
      double af = a_pri;
      for (int k = 0; k < 6 && !accepted; ++k, af *= 0.5) {
        const double az = a_dual * (af / a_pri);  // the bound multipliers move in step with the point
        CFZP_LANE_FOR(i, 0, n - 1) w.xt[i] = w.x[i] + af * w.dx[i];
        CFZP_LANE_FOR(i, 0, m - 1) w.nu[i] += af * w.dnu[i];
        CFZP_SYNC();
        constraints(sp, w.sel, w.xt, w.ct);
        gradient(sp, w.xt, w.g);
        jt_nu(sp, w.sel, w.xt, w.nu, w.r1);
        double cv = 0.0, di = 0.0;
        CFZP_LANE_FOR(i, 0, m - 1) cv = fmax(cv, fabs(w.ct[i]));
        CFZP_LANE_FOR(i, 0, n - 1) di = fmax(di, fabs(w.g[i] + w.r1[i] - (w.zl[i] + az * w.dzl[i]) + (w.zu[i] + az * w.dzu[i])));
        cv = bmax(cv); di = bmax(di);
#if defined(CFZC_TRACE)
        printf("   fallback k %d af %.3e cv %.3e di/s_d %.3e (e0 %.3e: dinf/s_d %.3e cviol %.3e) delta %.2e\n", k, af, cv, di / s_d, e0, dual_inf / s_d, cviol, delta);
#endif
        if (isfinite(cv) && isfinite(di) && w.xt[d.iDt] > 0.0 && fmax(di / s_d, cv) <= 0.9 * e0) { accepted = true; alpha = af; a_dual = az; nu_done = true; f_type = true; nfilt = 0; }
        else { CFZP_LANE_FOR(i, 0, m - 1) w.nu[i] -= af * w.dnu[i]; CFZP_SYNC(); }
      }
    }
    if (!accepted) {
      // Neither the filter nor the merit accepts any step length.  If the barrier parameter can still fall, the barrier problem
      // at hand is given up (typically it is solved to within a factor of two of its own stopping rule and the step is
      // dominated by the regularisations): mu falls, the filter starts afresh and the iterate stays.  Otherwise status 2.
      if (mu > mu_floor && !mu_forced) {
        mu = fmax(mu_floor, fmin(sp.kappa_mu * mu, pow(mu, sp.theta_mu))); mu_forced = true; nfilt = 0; filt_mu = mu;
        continue;
      }
      // Last resort before status 2: the step passed the curvature test but is no descent direction (seen at mu = mu_floor, a
      // perturbation of a third of the last one: dphi > 0, theta unchanged) -- the same iterate again with eight times the
      // perturbation, up to three times.  Only runs that would end here take this path.
      if (delta_retry < 3 && delta < 1e8) { delta_floor = fmax(8.0 * delta, 1e-4); ++delta_retry; nfilt = 0; continue; }
      status = 2; break;
    }
    mu_forced = false; delta_floor = 0.0; delta_retry = 0;
#if defined(CFZC_TRACE)  // CPU build only (tests/emu): g++ -DCFZC_TRACE -include stdio.h
    printf("it %3d mu %.2e err %.3e theta %.3e cviol %.2e dinf %.2e cmp %.2e delta %.1e alpha %.3e a_pri %.3e a_dual %.3e ftype %d dt %.5f f %.5f\n", iter, mu, err0, theta, cviol, dual_inf, cmp0, delta, alpha, a_pri, a_dual, (int)f_type, w.x[d.iDt], objective(sp, w.x));
#endif
    if (!f_type) {
      if (nfilt == sp.filter_cap) { for (int q = 1; q < nfilt; ++q) { filt[q - 1][0] = filt[q][0]; filt[q - 1][1] = filt[q][1]; } --nfilt; }
      filt[nfilt][0] = (1.0 - sp.gamma_theta) * theta; filt[nfilt][1] = phi0 - sp.gamma_phi * theta; ++nfilt;
    }
    if (!nu_done) lane_for_loads<4>(m, [&](int i) { LV q; q.v[0] = w.nu[i]; q.v[1] = w.dnu[i]; return q; }, [&](int i, const LV &q) { w.nu[i] = q.v[0] + alpha * q.v[1]; });
    lane_for_loads<3>(n, [&](int i) { LV q; q.v[0] = w.zl[i]; q.v[1] = w.zu[i]; q.v[2] = w.dzl[i]; q.v[3] = w.dzu[i]; q.v[4] = w.xl[i]; q.v[5] = w.xu[i]; q.v[6] = w.xt[i]; return q; },
                      [&](int i, const LV &q) {
                        w.x[i] = q.v[6];
                        if (q.v[4] > -1e300) { const double dl = q.v[6] - q.v[4]; w.zl[i] = fmin(fmax(q.v[0] + a_dual * q.v[2], mu / (sp.kappa_sigma * dl)), sp.kappa_sigma * mu / dl); }
                        if (q.v[5] < 1e300) { const double du = q.v[5] - q.v[6]; w.zu[i] = fmin(fmax(q.v[1] + a_dual * q.v[3], mu / (sp.kappa_sigma * du)), sp.kappa_sigma * mu / du); }
                      });
    CFZP_SYNC();
    tk[4] += tick() - ta;
  }
  CFZP_SYNC();
  tk[5] = tick() - t0;
  for (int i = 0; i < 17; ++i) out_d[3 + i] = (double)tk[i];
  CFZP_LANE_FOR(i, 0, d.iDt) X[i] = w.x[i];
  out_i[0] = iter; out_i[1] = status;
  out_d[0] = objective(sp, w.x); out_d[1] = err0; out_d[2] = mu;
}

}  // namespace cfzc
