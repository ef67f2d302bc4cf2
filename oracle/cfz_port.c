/* TEST INFRASTRUCTURE ONLY -- plain-C, single-thread port of the MPC-step solver.
 *
 * PARITY UNPINNED (see oracle/__init__.py).  This file restates, per problem instance and
 * with scalar loops, the same algorithm the HIP kernel runs (DESIGN.md "CFZ-IPM"):
 *   NLP        reference confrez/control/vehicle_follower.py:146-368 (setup_controller), with
 *              the OBCA duals eliminated into closed-form separation certificates
 *              (oracle/mpc_nlp.py: block_separation / certificate_duals)
 *   dynamics   reference confrez/control/dynamic_model.py:5-58 (RK4, M sub-steps)
 *   solver     oracle/ipm.py (interior point, filter line search) with the Newton system
 *              solved by slack elimination + Riccati recursion instead of a sparse LU
 * It is the `cpu_baseline` ("port") of bench.py and the iterate-level reference of the
 * kernel tests.  The product library never links or calls it.
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/_build/libcfz_port.so oracle/cfz_port.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 64
#define MAXB 16
#define MAXR (2 * MAXB) /* constraint rows per stage: two per block */
#define NP 7

typedef struct {
  int N, n_obs, n_nbr, rk_substeps, max_iter;
  double dt, wb, dmin;
  double g[4];
  double bounds[12];  /* lo,hi for x,y,v,delta,a,w */
  double weights[6];  /* x,y,psi,a,(v w)^2,delta */
  double A_obs[MAXB][4][2], b_obs[MAXB][4], V_obs[MAXB][4][2];
  /* solver options (oracle/ipm.py IpmOptions) */
  double tol, constr_viol_tol, dual_inf_tol, compl_inf_tol, mu_init, kappa_eps, kappa_mu, theta_mu,
      tau_min, bound_push, bound_frac, s_max, kappa_sigma, eta_phi, gamma_theta, gamma_phi, delta_sw,
      s_theta, s_phi, reg_primal, stall_kappa, warm_push;
  int filter_cap, max_backtrack, stall_iters, row_curvature;
  int vv_rows; /* 1: vertex-vertex rows (kind 3) in the working set, oracle/mpc_nlp.py MpcSpec.vv_rows */
  int shift_after; /* oracle/ipm.py IpmOptions.shift_after */
  int stag_win;    /* oracle/ipm.py IpmOptions.shift_stagnation */
  int err_stall;   /* oracle/ipm.py IpmOptions.err_stall_iters */
  int carry_shift; /* oracle/ipm.py IpmOptions.carry_shift */
  int resto;       /* oracle/ipm.py IpmOptions.restoration */
  double reg_dual_rows; /* oracle/ipm.py IpmOptions.reg_dual_rows: IPOPT's delta_c on the separation rows */
  double resto_first;   /* oracle/ipm.py IpmOptions.resto_first */
} cfz_port_spec;

/* state a converged solve hands to the next MPC iteration of the same vehicle (oracle/mpc_nlp.py carry_state) */
typedef struct {
  int valid;
  int sel[MAXN][MAXB];
  double z[MAXN][MAXR], zl[MAXN][6], zu[MAXN][6], pi0[5], pi[MAXN][5], mu;
  int shifted, pad; /* the solve that wrote the record shifted some stage's curvature (oracle/mpc_nlp.py carry_state) */
} cfz_port_carry;

typedef struct {
  double p[MAXN][NP], sg[MAXN][MAXR];              /* primal: stage vars, slacks (one per row) */
  double nuc[MAXN][MAXR], zs[MAXN][MAXR];          /* row multipliers, slack bound multipliers */
  double zl[MAXN][6], zu[MAXN][6];                 /* box multipliers */
  double pi0[5], pi[MAXN][5];                      /* initial-state and dynamics multipliers */
} iterate;

static const int BCOL[6] = {0, 1, 3, 4, 5, 6}; /* bounded columns of p */
static const double GB[4][2] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};

/* ---------------------------------------------------------------- dynamics */
static void f_ct(const double z[5], const double u[2], double wb, double out[5]) {
  out[0] = z[3] * cos(z[2]);
  out[1] = z[3] * sin(z[2]);
  out[2] = z[3] / wb * tan(z[4]);
  out[3] = u[0];
  out[4] = u[1];
}

/* f and its action on a sensitivity pair (Sz 5x5, Su 5x2) */
static void f_ct_sens(const double z[5], const double u[2], double wb, const double Sz[5][5],
                      const double Su[5][2], double f[5], double fz[5][5], double fu[5][2]) {
  double c = cos(z[2]), s = sin(z[2]), t = tan(z[4]), v = z[3];
  f[0] = v * c; f[1] = v * s; f[2] = v / wb * t; f[3] = u[0]; f[4] = u[1];
  double j02 = -v * s, j03 = c, j12 = v * c, j13 = s, j23 = t / wb, j24 = v / wb * (1.0 + t * t);
  for (int q = 0; q < 5; ++q) {
    fz[0][q] = j02 * Sz[2][q] + j03 * Sz[3][q];
    fz[1][q] = j12 * Sz[2][q] + j13 * Sz[3][q];
    fz[2][q] = j23 * Sz[3][q] + j24 * Sz[4][q];
    fz[3][q] = 0.0; fz[4][q] = 0.0;
  }
  for (int q = 0; q < 2; ++q) {
    fu[0][q] = j02 * Su[2][q] + j03 * Su[3][q];
    fu[1][q] = j12 * Su[2][q] + j13 * Su[3][q];
    fu[2][q] = j23 * Su[3][q] + j24 * Su[4][q];
    fu[3][q] = (q == 0); fu[4][q] = (q == 1);
  }
}

static void rk4(const double z[5], const double u[2], double dt, double wb, int M, double out[5]) {
  double h = dt / M, zk[5], a1[5], a2[5], a3[5], a4[5], tmp[5];
  memcpy(zk, z, sizeof zk);
  for (int m = 0; m < M; ++m) {
    f_ct(zk, u, wb, a1);
    for (int i = 0; i < 5; ++i) tmp[i] = zk[i] + h * a1[i] / 2;
    f_ct(tmp, u, wb, a2);
    for (int i = 0; i < 5; ++i) tmp[i] = zk[i] + h * a2[i] / 2;
    f_ct(tmp, u, wb, a3);
    for (int i = 0; i < 5; ++i) tmp[i] = zk[i] + h * a3[i];
    f_ct(tmp, u, wb, a4);
    for (int i = 0; i < 5; ++i) zk[i] = zk[i] + h / 6 * (a1[i] + 2 * a2[i] + 2 * a3[i] + a4[i]);
  }
  memcpy(out, zk, sizeof zk);
}

static void rk4_sens(const double z[5], const double u[2], double dt, double wb, int M, double out[5],
                     double A[5][5], double B[5][2]) {
  double h = dt / M, zk[5], Sz[5][5], Su[5][2];
  memcpy(zk, z, sizeof zk);
  memset(Sz, 0, sizeof Sz); memset(Su, 0, sizeof Su);
  for (int i = 0; i < 5; ++i) Sz[i][i] = 1.0;
  for (int m = 0; m < M; ++m) {
    double a[4][5], az[4][5][5], au[4][5][2], zt[5], Tz[5][5], Tu[5][2];
    f_ct_sens(zk, u, wb, Sz, Su, a[0], az[0], au[0]);
    for (int st = 1; st < 4; ++st) {
      double w = (st == 3) ? h : h / 2;
      for (int i = 0; i < 5; ++i) {
        zt[i] = zk[i] + w * a[st - 1][i];
        for (int q = 0; q < 5; ++q) Tz[i][q] = Sz[i][q] + w * az[st - 1][i][q];
        for (int q = 0; q < 2; ++q) Tu[i][q] = Su[i][q] + w * au[st - 1][i][q];
      }
      f_ct_sens(zt, u, wb, Tz, Tu, a[st], az[st], au[st]);
    }
    for (int i = 0; i < 5; ++i) {
      zk[i] += h / 6 * (a[0][i] + 2 * a[1][i] + 2 * a[2][i] + a[3][i]);
      for (int q = 0; q < 5; ++q) Sz[i][q] += h / 6 * (az[0][i][q] + 2 * az[1][i][q] + 2 * az[2][i][q] + az[3][i][q]);
      for (int q = 0; q < 2; ++q) Su[i][q] += h / 6 * (au[0][i][q] + 2 * au[1][i][q] + 2 * au[2][i][q] + au[3][i][q]);
    }
  }
  memcpy(out, zk, sizeof zk); memcpy(A, Sz, sizeof Sz); memcpy(B, Su, sizeof Su);
}

/* ---------------------------------------------------------------- separation certificates */
#define HYST 1e-3 /* m: a block keeps its separating face until another one is better by this much */
#define SHIFT_STAG_MIN 40 /* oracle/ipm.py SHIFT_STAG_MIN */
#define WS_STALL_DIV 4 /* iterates that change the working set count 1 / WS_STALL_DIV towards the stall test */

/* signed distances of the 4 vertices of one polygon to face f of the other (+ gradients wrt x,y,psi).
 * kind 1 = polygon face / body vertices, kind 2 = body face / polygon vertices. */
static void vertex_dist(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double psi,
                        const double g[4], int kind, int f, double d[4], double gr[4][3], double cur[4][3]) {
  double c = cos(psi), s = sin(psi);
  if (kind == 1) {
    double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
    for (int v = 0; v < 4; ++v) {
      double wx = x + c * BV[v][0] - s * BV[v][1], wy = y + s * BV[v][0] + c * BV[v][1];
      double dwx = -s * BV[v][0] - c * BV[v][1], dwy = c * BV[v][0] - s * BV[v][1];
      d[v] = wx * A[f][0] + wy * A[f][1] - b[f];
      if (gr) { gr[v][0] = A[f][0]; gr[v][1] = A[f][1]; gr[v][2] = A[f][0] * dwx + A[f][1] * dwy; }
      /* second derivatives (d2/dx dpsi, d2/dy dpsi, d2/dpsi2): only the rotation of the body vertex curves */
      if (cur) { cur[v][0] = 0.0; cur[v][1] = 0.0; cur[v][2] = -(A[f][0] * (wx - x) + A[f][1] * (wy - y)); }
    }
  } else {
    double nx = c * GB[f][0] - s * GB[f][1], ny = s * GB[f][0] + c * GB[f][1];
    double dnx = -s * GB[f][0] - c * GB[f][1], dny = c * GB[f][0] - s * GB[f][1];
    for (int v = 0; v < 4; ++v) {
      d[v] = (V[v][0] - x) * nx + (V[v][1] - y) * ny - g[f];
      if (gr) { gr[v][0] = -nx; gr[v][1] = -ny; gr[v][2] = dnx * (V[v][0] - x) + dny * (V[v][1] - y); }
      if (cur) { cur[v][0] = -dnx; cur[v][1] = -dny; cur[v][2] = -((V[v][0] - x) * nx + (V[v][1] - y) * ny); }
    }
  }
}

/* kind 3 (oracle/mpc_nlp.py closest_vertex_pair): polygon vertex u and body vertex v that lie in each other's normal cone
 * are THE closest points of the two polygons.  Body side from the signs of the kind-2 distances (V[u] outside exactly the two
 * body faces that meet at v), polygon side (W_v - V_u).e <= 0 for the two edges e leaving V[u]. */
static int closest_vertex_pair(const double V[4][2], double x, double y, double psi, const double g[4], int *uo, int *vo,
                               double *ro) {
  double c = cos(psi), s = sin(psi);
  double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
  for (int u = 0; u < 4; ++u) {
    double rx = V[u][0] - x, ry = V[u][1] - y;
    double qx = c * rx + s * ry, qy = -s * rx + c * ry;
    int fx = qx - g[0] >= 0.0 ? 0 : (-qx - g[2] >= 0.0 ? 2 : -1);
    int fy = qy - g[1] >= 0.0 ? 1 : (-qy - g[3] >= 0.0 ? 3 : -1);
    if (fx < 0 || fy < 0) continue;
    int v = fx == 0 ? (fy == 1 ? 0 : 3) : (fy == 1 ? 1 : 2);
    double wx = x + c * BV[v][0] - s * BV[v][1] - V[u][0], wy = y + s * BV[v][0] + c * BV[v][1] - V[u][1];
    int u1 = (u + 1) & 3, u3 = (u + 3) & 3;
    if (wx * (V[u1][0] - V[u][0]) + wy * (V[u1][1] - V[u][1]) <= 0.0 &&
        wx * (V[u3][0] - V[u][0]) + wy * (V[u3][1] - V[u][1]) <= 0.0) {
      *uo = u; *vo = v; *ro = sqrt(wx * wx + wy * wy);
      return 1;
    }
  }
  return 0;
}

/* working set of one block: sel = kind*64 + face*16 + vA*4 + vB (vA < vB); see oracle/mpc_nlp.py select_rows */
static int select_rows(const double A[4][2], const double b[4], const double V[4][2], double x, double y, double psi,
                       const double g[4], int prev, int vv) {
  double best = 0.0, prev_val = 0.0, d[4];
  int have = 0, bk = 0, bf = 0, have_prev = 0;
  int pk = prev >> 6, pf = (prev >> 4) & 3;
  for (int kind = 1; kind <= 2; ++kind)
    for (int f = 0; f < 4; ++f) {
      vertex_dist(A, b, V, x, y, psi, g, kind, f, d, 0, 0);
      double val = fmin(fmin(d[0], d[1]), fmin(d[2], d[3]));
      if (prev && kind == pk && f == pf) { prev_val = val; have_prev = 1; }
      if (!have || val > best) { have = 1; best = val; bk = kind; bf = f; }
    }
  if (have_prev && prev_val >= best - HYST) { bk = pk; bf = pf; }
  vertex_dist(A, b, V, x, y, psi, g, bk, bf, d, 0, 0);
  int v0 = 0;
  for (int v = 1; v < 4; ++v) if (d[v] < d[v0]) v0 = v;
  int n1 = (v0 + 1) & 3, n2 = (v0 + 3) & 3, v1;
  if (d[n1] < d[n2]) v1 = n1; else if (d[n2] < d[n1]) v1 = n2; else v1 = n1 < n2 ? n1 : n2;
  if (prev && bk == pk && bf == pf) {
    int oa = (prev >> 2) & 3, ob = prev & 3;
    if (fmin(d[oa], d[ob]) <= d[v0] + 1e-12 && fmax(d[oa], d[ob]) <= d[v1] + HYST) { v0 = oa; v1 = ob; }
  }
  int va = v0 < v1 ? v0 : v1, vb = v0 < v1 ? v1 : v0;
  const double dn = fmin(d[v0], d[v1]); /* the separation this face certifies (a kept pair is not ordered by distance) */
  if (vv && dn > 0.0) {
    int u, v; double r;
    if (closest_vertex_pair(V, x, y, psi, g, &u, &v, &r) && r > dn + 1e-9) return 192 + u * 16 + v * 4 + v;
  }
  return bk * 64 + bf * 16 + va * 4 + vb;
}

static void nbr_polygon(const double *nbr, int N, int o, int k, const double g[4], double A[4][2], double b[4],
                        double V[4][2]) {
  double xo = nbr[(o * 3 + 0) * N + k], yo = nbr[(o * 3 + 1) * N + k], po = nbr[(o * 3 + 2) * N + k];
  double c = cos(po), s = sin(po);
  double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
  for (int i = 0; i < 4; ++i) {
    A[i][0] = c * GB[i][0] - s * GB[i][1];
    A[i][1] = s * GB[i][0] + c * GB[i][1];
    b[i] = A[i][0] * xo + A[i][1] * yo + g[i];
    V[i][0] = xo + c * BV[i][0] - s * BV[i][1];
    V[i][1] = yo + s * BV[i][0] + c * BV[i][1];
  }
}

static void block_polygon(const cfz_port_spec *sp, const double *nbr, int k, int j, double A[4][2], double b[4],
                          double V[4][2]) {
  if (j < sp->n_obs) { memcpy(A, sp->A_obs[j], 64); memcpy(b, sp->b_obs[j], 32); memcpy(V, sp->V_obs[j], 64); }
  else nbr_polygon(nbr, sp->N, j - sp->n_obs, k, sp->g, A, b, V);
}

/* refresh the working set at poses p (x,y,psi of every stage); `sel` in/out */
static void select_all(const cfz_port_spec *sp, const double *nbr, const double p[][NP], int sel[][MAXB]) {
  int nb = sp->n_obs + sp->n_nbr;
  for (int k = 0; k < sp->N; ++k)
    for (int j = 0; j < nb; ++j) {
      double A[4][2], b[4], V[4][2];
      block_polygon(sp, nbr, k, j, A, b, V);
      sel[k][j] = select_rows(A, b, V, p[k][0], p[k][1], p[k][2], sp->g, sel[k][j], sp->vv_rows);
    }
}

/* rows of the current working set: sep[k][2j+r], grad[k][2j+r][3], curv[k][2j+r][6] = second derivatives
 * (x psi, y psi, psi psi, x x, y y, x y); the last three are zero except for a vertex-vertex row */
static void eval_rows(const cfz_port_spec *sp, const double *nbr, const double p[][NP], int sel[][MAXB],
                      double sep[][MAXR], double grad[][MAXR][3], double curv[][MAXR][6]) {
  int nb = sp->n_obs + sp->n_nbr;
  for (int k = 0; k < sp->N; ++k)
    for (int j = 0; j < nb; ++j) {
      double A[4][2], b[4], V[4][2], d[4], gr[4][3], cu[4][3];
      block_polygon(sp, nbr, k, j, A, b, V);
      int c = sel[k][j], kind = c >> 6, f = (c >> 4) & 3, va = (c >> 2) & 3, vb = c & 3;
      if (kind == 3) {
        /* r = |w|, w = t + R b_v - V_u; the row twice (the block keeps its two slots).  Hessian of r:
         * tau tau' / r + kappa e_psi e_psi', tau = J' t (t the unit tangent), kappa = n . (-R b_v) */
        const double *g = sp->g;
        double BV[4][2] = {{g[0], g[1]}, {-g[2], g[1]}, {-g[2], -g[3]}, {g[0], -g[3]}};
        double cs = cos(p[k][2]), sn = sin(p[k][2]);
        double rbx = cs * BV[va][0] - sn * BV[va][1], rby = sn * BV[va][0] + cs * BV[va][1];
        double dwx = -rby, dwy = rbx;
        double wx = p[k][0] + rbx - V[f][0], wy = p[k][1] + rby - V[f][1];
        double r = sqrt(wx * wx + wy * wy), ir = 1.0 / r, n0 = wx * ir, n1 = wy * ir; /* as the kernel forms them */
        double t2 = -n1 * dwx + n0 * dwy, kap = -(n0 * rbx + n1 * rby);
        for (int q = 0; q < 2; ++q) {
          sep[k][2 * j + q] = r;
          if (grad) { grad[k][2 * j + q][0] = n0; grad[k][2 * j + q][1] = n1; grad[k][2 * j + q][2] = n0 * dwx + n1 * dwy; }
          if (curv) {
            double *cq = curv[k][2 * j + q];
            cq[0] = -n1 * t2 / r; cq[1] = n0 * t2 / r; cq[2] = t2 * t2 / r + kap;
            cq[3] = n1 * n1 / r; cq[4] = n0 * n0 / r; cq[5] = -n1 * n0 / r;
          }
        }
        continue;
      }
      vertex_dist(A, b, V, p[k][0], p[k][1], p[k][2], sp->g, kind, f, d, grad ? gr : 0, curv ? cu : 0);
      sep[k][2 * j] = d[va]; sep[k][2 * j + 1] = d[vb];
      if (grad) for (int q = 0; q < 3; ++q) { grad[k][2 * j][q] = gr[va][q]; grad[k][2 * j + 1][q] = gr[vb][q]; }
      if (curv) for (int q = 0; q < 6; ++q) { curv[k][2 * j][q] = q < 3 ? cu[va][q] : 0.0; curv[k][2 * j + 1][q] = q < 3 ? cu[vb][q] : 0.0; }
    }
}

/* ---------------------------------------------------------------- objective */
static double stage_cost(const cfz_port_spec *sp, const double *ref, int k, const double p[NP]) {
  const double *w = sp->weights; int N = sp->N;
  double ex = p[0] - ref[0 * N + k], ey = p[1] - ref[1 * N + k], ep = p[2] - ref[2 * N + k];
  return w[0] * ex * ex + w[1] * ey * ey + w[2] * ep * ep + w[3] * p[5] * p[5] + w[4] * p[3] * p[3] * p[6] * p[6] +
         w[5] * p[4] * p[4];
}
static void stage_grad(const cfz_port_spec *sp, const double *ref, int k, const double p[NP], double gr[NP]) {
  const double *w = sp->weights; int N = sp->N;
  gr[0] = 2 * w[0] * (p[0] - ref[0 * N + k]);
  gr[1] = 2 * w[1] * (p[1] - ref[1 * N + k]);
  gr[2] = 2 * w[2] * (p[2] - ref[2 * N + k]);
  gr[3] = 2 * w[4] * p[3] * p[6] * p[6];
  gr[4] = 2 * w[5] * p[4];
  gr[5] = 2 * w[3] * p[5];
  gr[6] = 2 * w[4] * p[3] * p[3] * p[6];
}

/* theta = |c|_1 and barrier objective at a trial point (p, sg); returns 0 if outside bounds */
static int merit_terms(const cfz_port_spec *sp, const double *x0, const double *ref, const double *nbr,
                       const double p[][NP], const double sg[][MAXR], int sel[][MAXB], double mu, double *theta,
                       double *phi) {
  int N = sp->N, nb = 2 * (sp->n_obs + sp->n_nbr); /* rows */
  double th = 0.0, ph = 0.0, lg = 0.0;
  static double sep[MAXN][MAXR];
  for (int k = 0; k < N; ++k) {
    for (int q = 0; q < 6; ++q) {
      double dl = p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - p[k][BCOL[q]];
      if (!(dl > 0.0) || !(du > 0.0)) return 0;
      lg += log(dl) + log(du);
    }
    for (int j = 0; j < nb; ++j) {
      if (!(sg[k][j] > 0.0)) return 0;
      lg += log(sg[k][j]);
    }
    ph += stage_cost(sp, ref, k, p[k]);
  }
  for (int i = 0; i < 5; ++i) th += fabs(p[0][i] - x0[i]);
  for (int k = 0; k + 1 < N; ++k) {
    double F[5];
    rk4(p[k], p[k] + 5, sp->dt, sp->wb, sp->rk_substeps, F);
    for (int i = 0; i < 5; ++i) th += fabs(F[i] - p[k + 1][i]);
  }
  eval_rows(sp, nbr, p, sel, sep, 0, 0);
  for (int k = 1; k < N; ++k) /* the rows of stage 0 are constants: no part of the violation */
    for (int j = 0; j < nb; ++j) th += fabs(sep[k][j] - sp->dmin - sg[k][j]);
  *theta = th; *phi = ph - mu * lg;
  return isfinite(th) && isfinite(*phi);
}

/* ---------------------------------------------------------------- small dense helpers */
/* max(0, -lambda_min) of the symmetric 3 x 3 [[a00 a01 a02] [a01 a11 a12] [a02 a12 a22]] */
static double pose_shift(double a00, double a11, double a22, double a01, double a02, double a12) {
  const double d2 = a00 * a11 - a01 * a01;
  const double d3 = a22 * d2 - (a02 * a02 * a11 - 2.0 * a02 * a12 * a01 + a12 * a12 * a00);
  if (a00 > 0.0 && d2 > 0.0 && d3 >= 0.0) return 0.0;
  const double p1 = a01 * a01 + a02 * a02 + a12 * a12;
  const double qm = (a00 + a11 + a22) / 3.0;
  const double b00 = a00 - qm, b11 = a11 - qm, b22 = a22 - qm;
  const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1) / 6.0);
  const double ip = 1.0 / p;
  const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
  double r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02));
  r = fmin(1.0, fmax(-1.0, r));
  const double lam = qm + 2.0 * p * cos(acos(r) / 3.0 + 2.0943951023931953);
  return fmax(0.0, -lam);
}

static void sym2_solve(const double M[2][2], const double *rhs, int nr, double *out) {
  /* out = M^{-1} rhs for nr right-hand sides stored as rhs[2][nr] (row-major) via Cholesky */
  double l00 = sqrt(M[0][0]), l10 = M[1][0] / l00, l11 = sqrt(M[1][1] - l10 * l10);
  for (int q = 0; q < nr; ++q) {
    double y0 = rhs[0 * nr + q] / l00, y1 = (rhs[1 * nr + q] - l10 * y0) / l11;
    double x1 = y1 / l11, x0 = (y0 - l10 * x1) / l00;
    out[0 * nr + q] = x0; out[1 * nr + q] = x1;
  }
}

/* ---------------------------------------------------------------- work arrays (one solve at a time: not re-entrant) */
static iterate it, dt_; /* iterate; step stored in an `iterate` too */
static double sep[MAXN][MAXR], gra[MAXN][MAXR][3], cur[MAXN][MAXR][6], cj[MAXN][MAXR];
static int sel[MAXN][MAXB];
static double Fk[MAXN][5], Ak[MAXN][5][5], Bk[MAXN][5][2], dk[MAXN][5];
static double H[MAXN][NP][NP], gk[MAXN][NP], gphi[MAXN][NP];
static double Kk[MAXN][2][5], kf[MAXN][2];
static double pt[MAXN][NP], sgt[MAXN][MAXR];
static double Ps[MAXN][5][5], ps[MAXN][5];
static double pinew[MAXN][5], pi0new[5]; /* multipliers of the dynamics / initial-state rows after a full Newton step */

/* ---------------------------------------------------------------- the Newton system's stage recursion */
/* Riccati backward sweep over the condensed stage problems (H, gk, Ak, Bk, dk): gains Kk, kf and the value function
 * 0.5 dz'P_k dz + p_k'dz of every stage.  Returns 0 if some stage's Huu is not positive definite.
 * Since round 5 the sweep mirrors the kernel's, which runs on the matrix cores (cfz_solver.inl riccati_backward_mfma), BIT FOR BIT: the
 * stage in homogeneous coordinates, variables ordered [z (5), 1, u (2)],
 *     [z+; 1] = T [z; 1; u],  T = [[A d B], [0 1 0]],   M = T' Pt T + Ht,   Pt <- M_kk - M_ke M_ee^-1 M_ek,   K = -M_ee^-1 M_ek,
 * every sum in the order v_mfma_f64_16x16x4_f64 forms it -- k ascending, one fused multiply-add per k onto the accumulator, which starts
 * as Ht for M and as M for the update -- with the operands the instructions get (Pt read transposed, T's structural entries as the
 * constants 0, 1, dt).  tools/src/riccati_mfma_bench.hip checks this emulation against the instructions: 0 of 2,880 gains differ. */
static int riccati_backward(int N, double port_dt) {
  int ok = 1;
  double P[6][6];
  memset(P, 0, sizeof P);
  for (int k = N - 1; k >= 0; --k) {
    double T[6][8], Ht[8][8], Y[6][8], M[8][8], V[2][6];
    memset(T, 0, sizeof T); memset(Ht, 0, sizeof Ht);
    if (k < N - 1) { /* (the terminal stage's inputs a, w are costed but drive no dynamics: T = 0 against Pt = 0) */
      for (int r = 0; r < 5; ++r) { T[r][r] = 1.0; T[r][5] = dk[k][r]; }
      T[0][2] = Ak[k][0][2]; T[0][3] = Ak[k][0][3]; T[0][4] = Ak[k][0][4];
      T[1][2] = Ak[k][1][2]; T[1][3] = Ak[k][1][3]; T[1][4] = Ak[k][1][4];
      T[2][3] = Ak[k][2][3]; T[2][4] = Ak[k][2][4];
      for (int r = 0; r < 3; ++r) { T[r][6] = Bk[k][r][0]; T[r][7] = Bk[k][r][1]; }
      T[3][6] = port_dt; T[4][7] = port_dt;
      T[5][5] = 1.0;
    }
    { /* Ht: the stage's Hessian in its pattern (diagonal, the pose block, the v-w cross term), the gradient in row / column 5 */
      static const int zu[8] = {0, 1, 2, 3, 4, -1, 5, 6};
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) {
          const int za = zu[a], zb = zu[b];
          if (za < 0 && zb < 0) continue;
          if (za < 0 || zb < 0) { Ht[a][b] = gk[k][za < 0 ? zb : za]; continue; }
          const int lo = za < zb ? za : zb, hi = za < zb ? zb : za;
          if (lo == hi || (lo == 0 && hi == 1) || (lo == 0 && hi == 2) || (lo == 1 && hi == 2) || (lo == 3 && hi == 6)) Ht[a][b] = H[k][lo][hi];
        }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) { double s_ = 0.0; for (int q = 0; q < 6; ++q) s_ = fma(P[q][i], T[q][j], s_); Y[i][j] = s_; }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { double s_ = Ht[i][j]; for (int q = 0; q < 6; ++q) s_ = fma(T[q][i], Y[q][j], s_); M[i][j] = s_; }
    const double det = fma(M[6][6], M[7][7], -(M[6][7] * M[7][6]));
    if (!(M[6][6] > 0.0 && det > 0.0)) ok = 0;
    const double idet = 1.0 / det;
    const double i00 = M[7][7] * idet, i01 = -M[6][7] * idet, i10 = -M[7][6] * idet, i11 = M[6][6] * idet;
    for (int j = 0; j < 6; ++j) { V[0][j] = fma(i00, M[6][j], i01 * M[7][j]); V[1][j] = fma(i10, M[6][j], i11 * M[7][j]); }
    for (int a = 0; a < 2; ++a) { for (int q = 0; q < 5; ++q) Kk[k][a][q] = -V[a][q]; kf[k][a] = -V[a][5]; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) P[i][j] = fma(-M[7][i], V[1][j], fma(-M[6][i], V[0][j], M[i][j]));
    for (int i = 0; i < 5; ++i) { for (int q = 0; q < 5; ++q) Ps[k][i][q] = P[i][q]; ps[k][i] = P[5][i]; }
  }
  return ok;
}

/* forward sweep: the step dt_.p from dz_0 = x0 - z_0, and the multipliers a full step would leave (pi0new, pinew) */
static void riccati_forward(int N, const double *x0) {
  for (int i = 0; i < 5; ++i) dt_.p[0][i] = x0[i] - it.p[0][i];
  for (int k = 0; k < N; ++k) {
    for (int a = 0; a < 2; ++a) { double s = kf[k][a]; for (int q = 0; q < 5; ++q) s += Kk[k][a][q] * dt_.p[k][q]; dt_.p[k][5 + a] = s; }
    if (k + 1 < N)
      for (int i = 0; i < 5; ++i) {
        double s = dk[k][i];
        for (int q = 0; q < 5; ++q) s += Ak[k][i][q] * dt_.p[k][q];
        for (int q = 0; q < 2; ++q) s += Bk[k][i][q] * dt_.p[k][5 + q];
        dt_.p[k + 1][i] = s;
      }
  }
  for (int i = 0; i < 5; ++i) { double s = ps[0][i]; for (int q = 0; q < 5; ++q) s += Ps[0][i][q] * dt_.p[0][q]; pi0new[i] = -s; }
  for (int k = 0; k + 1 < N; ++k)
    for (int i = 0; i < 5; ++i) { double s = ps[k + 1][i]; for (int q = 0; q < 5; ++q) s += Ps[k + 1][i][q] * dt_.p[k + 1][q]; pinew[k][i] = s; }
}

/* ---------------------------------------------------------------- feasibility restoration (oracle/ipm.py restore) */
/* IPOPT answers a failed line search with its restoration phase (paper sec. 3.3): it minimises the constraint violation and resumes
 * from the point that reaches.  Here the violation of the separation rows and of the boxes is a sum of squared hinges -- a nonlinear
 * least-squares problem over the trajectory,
 *     min  rho / 2 sum_{k >= 1, r} max(0, dmin + eps - sep_kr(p_k))^2  +  rho_b / 2 sum (excess over the boxes shrunk by m_b)^2
 *     s.t. z_0 = x0, z_{k+1} = F(z_k, u_k)
 * solved by Levenberg-Marquardt steps on the same stage recursion as the solver's Newton system (H_k = (zeta + lambda) I + rho sum of
 * a a' over the violated rows + rho_b on the violated boxes; zeta = sqrt(mu) is IPOPT's proximity weight, paper eq. 30, taken to the
 * current iterate; lambda grows fourfold after a step the line search cut below 0.2 and falls after a full step), with an Armijo line
 * search on the l1 merit (objective + eta x |dynamics defects|_1).  No barrier and no multipliers inside: a step is never cut by the
 * fraction to the boundary (a primal barrier at the solver's small mu crawls from one saturating input to the next).
 * Returns 1 when every row of stages >= 1 is within the goal (a tenth of the violation at entry -- IPOPT asks for nine tenths -- or
 * half the margin eps), every box holds with margin m_b / 2 and the dynamics defects are no worse than at entry (the iterate is then
 * clipped m_b / 2 inside the boxes); 0 when the worst violation has not dropped by a tenth in RESTO_STALL iterations (a stationary
 * point of the violation: locally infeasible), when a line search fails, or at the iteration limit.
 * *iter counts the restoration's iterations on the solve's counter. */
#define RESTO_RHO 1000.0
#define RESTO_RHO_BOX 1e5
#define RESTO_BOX_MARGIN 2e-3
#define RESTO_MAX_ITER 40
#define RESTO_STALL 8
#define RESTO_KAPPA 0.1
#define RESTO_ARMIJO 1e-4
static double resto_objective(const cfz_port_spec *sp, const double p[][NP], const double sp_[][MAXR], double eps) {
  const int N = sp->N, nb = 2 * (sp->n_obs + sp->n_nbr);
  double phi = 0.0;
  for (int k = 0; k < N; ++k) {
    for (int q = (k == 0 ? 4 : 0); q < 6; ++q) {
      const double el = sp->bounds[2 * q] + RESTO_BOX_MARGIN - p[k][BCOL[q]], eu = p[k][BCOL[q]] - sp->bounds[2 * q + 1] + RESTO_BOX_MARGIN;
      if (el > 0.0) phi += 0.5 * RESTO_RHO_BOX * el * el;
      if (eu > 0.0) phi += 0.5 * RESTO_RHO_BOX * eu * eu;
    }
    if (k >= 1) for (int j = 0; j < nb; ++j) { const double v = sp->dmin + eps - sp_[k][j]; if (v > 0.0) phi += 0.5 * RESTO_RHO * v * v; }
  }
  return phi;
}

static int restore(const cfz_port_spec *sp, const double *x0, const double *nbr, double mu, int *iter) {
  const int N = sp->N, nblk = sp->n_obs + sp->n_nbr, nb = 2 * nblk;
  const double zeta = sqrt(mu), rho = RESTO_RHO, mb = RESTO_BOX_MARGIN;
  double eta = 0.0, eps = 0.0, lm = 0.0, vgoal = 0.0, vref = INFINITY, dgoal = 0.0; int ref_it = 0;
  for (int rit = 0;; ++rit) {
    if (rit > 0) select_all(sp, nbr, it.p, sel);
    eval_rows(sp, nbr, it.p, sel, sep, gra, 0);
    double th_dyn = 0.0, cv_dyn = 0.0, vmax = 0.0, bmax = 0.0;
    if (rit == 0) { /* the margin the rows are restored with: bound_push, but no more than the worst violation at entry */
      double v0 = 0.0;
      for (int k = 1; k < N; ++k) for (int j = 0; j < nb; ++j) v0 = fmax(v0, sp->dmin - sep[k][j]);
      eps = fmin(sp->bound_push, v0);
      vgoal = fmax(0.5 * eps, RESTO_KAPPA * (v0 + eps));
    }
    for (int i = 0; i < 5; ++i) { double r = it.p[0][i] - x0[i]; th_dyn += fabs(r); cv_dyn = fmax(cv_dyn, fabs(r)); }
    for (int k = 0; k + 1 < N; ++k) {
      rk4_sens(it.p[k], it.p[k] + 5, sp->dt, sp->wb, sp->rk_substeps, Fk[k], Ak[k], Bk[k]);
      for (int i = 0; i < 5; ++i) { dk[k][i] = Fk[k][i] - it.p[k + 1][i]; th_dyn += fabs(dk[k][i]); cv_dyn = fmax(cv_dyn, fabs(dk[k][i])); }
    }
    for (int k = 0; k < N; ++k) {
      memset(H[k], 0, sizeof H[k]);
      for (int i = 0; i < NP; ++i) { gk[k][i] = 0.0; H[k][i][i] = zeta + lm; }
      for (int q = (k == 0 ? 4 : 0); q < 6; ++q) { /* the states of stage 0 are the measurement: nothing to restore there */
        const int c = BCOL[q];
        const double el = sp->bounds[2 * q] + mb - it.p[k][c], eu = it.p[k][c] - sp->bounds[2 * q + 1] + mb;
        if (el > 0.0) { gk[k][c] -= RESTO_RHO_BOX * el; H[k][c][c] += RESTO_RHO_BOX; bmax = fmax(bmax, el); }
        if (eu > 0.0) { gk[k][c] += RESTO_RHO_BOX * eu; H[k][c][c] += RESTO_RHO_BOX; bmax = fmax(bmax, eu); }
      }
      if (k >= 1)
        for (int j = 0; j < nb; ++j) {
          const double v = sp->dmin + eps - sep[k][j];
          if (v > 0.0) {
            vmax = fmax(vmax, v);
            for (int a = 0; a < 3; ++a) {
              gk[k][a] -= rho * v * gra[k][j][a];
              for (int b = 0; b < 3; ++b) H[k][a][b] += rho * gra[k][j][a] * gra[k][j][b];
            }
          }
        }
      for (int i = 0; i < NP; ++i) gphi[k][i] = gk[k][i];
    }
    const double phi = resto_objective(sp, it.p, sep, eps);
    if (rit == 0) dgoal = fmax(sp->constr_viol_tol, cv_dyn); /* dynamics: no worse than at entry */
    if (vmax <= vgoal && bmax <= 0.5 * mb && cv_dyn <= dgoal) {
      for (int k = 0; k < N; ++k)
        for (int q = 0; q < 6; ++q) {
          const int c = BCOL[q];
          it.p[k][c] = fmin(fmax(it.p[k][c], sp->bounds[2 * q] + 0.5 * mb), sp->bounds[2 * q + 1] - 0.5 * mb);
        }
      return 1;
    }
    /* stalled: the worst violation has not dropped by a tenth in RESTO_STALL iterations -> a stationary point of the violation */
    if (vmax <= 0.9 * vref || vmax <= vgoal) { vref = vmax; ref_it = rit; }
    if (rit - ref_it >= RESTO_STALL) return 0;
    if (rit == RESTO_MAX_ITER || *iter >= sp->max_iter) return 0;
    riccati_backward(N, sp->dt);
    riccati_forward(N, x0);
    double dphi = 0.0, pim = 0.0;
    for (int i = 0; i < 5; ++i) pim = fmax(pim, fabs(pi0new[i]));
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NP; ++i) dphi += gphi[k][i] * dt_.p[k][i];
      if (k + 1 < N) for (int i = 0; i < 5; ++i) pim = fmax(pim, fabs(pinew[k][i]));
    }
    if (eta < 1.1 * pim) eta = 2.0 * pim;
    const double M0 = phi + eta * th_dyn, dM = dphi - eta * th_dyn;
    if (!(dM < -1e-10 * (1.0 + fabs(M0)))) return 0; /* stationary with rows still violated */
    double alpha = 1.0; int accepted = 0;
    for (int bt = 0; bt < sp->max_backtrack; ++bt) {
      double th_t = 0.0;
      for (int k = 0; k < N; ++k) for (int i = 0; i < NP; ++i) pt[k][i] = it.p[k][i] + alpha * dt_.p[k][i];
      for (int i = 0; i < 5; ++i) th_t += fabs(pt[0][i] - x0[i]);
      for (int k = 0; k + 1 < N; ++k) {
        double F[5];
        rk4(pt[k], pt[k] + 5, sp->dt, sp->wb, sp->rk_substeps, F);
        for (int i = 0; i < 5; ++i) th_t += fabs(F[i] - pt[k + 1][i]);
      }
      eval_rows(sp, nbr, pt, sel, sgt, 0, 0); /* the working set is held during the line search, as in the solver */
      const double M_t = resto_objective(sp, pt, sgt, eps) + eta * th_t;
      if (isfinite(M_t) && M_t <= M0 + RESTO_ARMIJO * alpha * dM) { accepted = 1; break; }
      alpha *= 0.5;
    }
    if (!accepted) return 0;
    if (alpha < 0.2) lm = fmax(4.0 * lm, 1.0); else if (alpha == 1.0) lm *= 0.25;
    for (int k = 0; k < N; ++k) for (int i = 0; i < NP; ++i) it.p[k][i] = pt[k][i];
    ++*iter;
  }
}

/* cold multipliers at the current point (after a restoration; IPOPT resets its bound multipliers there and recomputes the others):
 * slacks from the rows, z = mu / distance, the equality rows at zero */
static void cold_multipliers(const cfz_port_spec *sp, const double *nbr, double mu) {
  const int N = sp->N, nb = 2 * (sp->n_obs + sp->n_nbr);
  select_all(sp, nbr, it.p, sel);
  eval_rows(sp, nbr, it.p, sel, sep, 0, 0);
  for (int k = 0; k < N; ++k) {
    for (int j = 0; j < nb; ++j) {
      it.sg[k][j] = fmax(sep[k][j] - sp->dmin, 0.5 * sp->bound_push);
      it.zs[k][j] = mu / it.sg[k][j]; it.nuc[k][j] = -it.zs[k][j];
    }
    for (int q = 0; q < 6; ++q) {
      const double dl = it.p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - it.p[k][BCOL[q]];
      it.zl[k][q] = mu / dl; it.zu[k][q] = mu / du;
    }
    for (int i = 0; i < 5; ++i) it.pi[k][i] = 0.0;
  }
  for (int i = 0; i < 5; ++i) it.pi0[i] = 0.0;
}

/* ---------------------------------------------------------------- the solver */
/* p_io: [N][7] warm start in (x,y,psi,v,delta,a,w per stage), solution out.
 * stats: [0]=iters [1]=status(0 ok,1 maxiter,2 linesearch,3 nan,4 initial state in collision) ; fstats: [0]=f [1]=err [2]=mu
 * trace (optional): per iteration 4 doubles (mu, err0, cviol, dual_inf) then p[N][7] -> stride 4+7N */
int cfz_port_solve_carry(const cfz_port_spec *sp, const double *x0, const double *ref, const double *nbr, double *p_io,
                         double *sep_out, int *cert_out, int *stats, double *fstats, double *trace, int trace_cap,
                         const cfz_port_carry *cin, cfz_port_carry *cout) {
  if (cout) { cout->valid = 0; cout->shifted = 0; }
  const int shift_hint = (cin && cin->valid && sp->carry_shift > 0) ? cin->shifted : 0; /* oracle/ipm.py carry_shift */
  int shift_used = 0;
  const int N = sp->N, nblk = sp->n_obs + sp->n_nbr, nb = 2 * nblk; /* nb = rows per stage */
  if (N > MAXN || N < 2 || nblk > MAXB) return -1;
  double filt[64][2]; int nfilt = 0; double filt_mu = -1.0;
  double theta_min = -1.0, theta_max = -1.0;
  const double mu_floor = fmin(sp->tol, sp->compl_inf_tol) / (sp->kappa_eps + 1.0);
  double stall_ref = 0.0;
  int stall_cnt = 0, stall_ws = 0;
  int resto_calls = 0, iter0 = 0, locally_infeasible = 0;
  double mu = sp->mu_init;
  int status = 1, iter = 0;
  int stagnant = 0, best_it = 0; double best_err = INFINITY; /* oracle/ipm.py shift_stagnation */
  double err0 = INFINITY;
  const int m_eq = 5 + 5 * (N - 1) + nb * N, n_bnd = N * (12 + nb); /* nb counts rows here */

  /* ---- initial point: slacks from the un-pushed warm start, then push everything inside */
  for (int k = 0; k < N; ++k) for (int i = 0; i < NP; ++i) it.p[k][i] = p_io[k * NP + i];
  memset(sel, 0, sizeof sel);
  { /* the pose of stage 0 is fixed by z0 = x0: a collision row violated there cannot be repaired */
    double p0[1][NP]; int s0[1][MAXB]; double r0[1][MAXR];
    for (int i = 0; i < NP; ++i) p0[0][i] = (i < 5) ? x0[i] : 0.0;
    memset(s0, 0, sizeof s0);
    cfz_port_spec one = *sp; one.N = 1;
    /* neighbour arrays are indexed with stride N, so evaluate stage 0 through the full-size helpers */
    for (int j = 0; j < nblk; ++j) {
      double A[4][2], b[4], V[4][2], d[4];
      block_polygon(sp, nbr, 0, j, A, b, V);
      s0[0][j] = select_rows(A, b, V, p0[0][0], p0[0][1], p0[0][2], sp->g, 0, sp->vv_rows);
      int c = s0[0][j];
      if ((c >> 6) == 3) {
        int u_, v_;
        closest_vertex_pair(V, p0[0][0], p0[0][1], p0[0][2], sp->g, &u_, &v_, &r0[0][j]);
      } else {
        vertex_dist(A, b, V, p0[0][0], p0[0][1], p0[0][2], sp->g, c >> 6, (c >> 4) & 3, d, 0, 0);
        r0[0][j] = fmin(d[(c >> 2) & 3], d[c & 3]);
      }
      if (r0[0][j] < sp->dmin - sp->constr_viol_tol) {
        stats[0] = 0; stats[1] = 4; fstats[0] = 0.0; fstats[1] = INFINITY; fstats[2] = sp->mu_init;
        if (sep_out) for (int q = 0; q < N * nblk; ++q) sep_out[q] = 0.0;
        if (cert_out) for (int q = 0; q < N * nblk; ++q) cert_out[q] = 0;
        return 0;
      }
    }
    (void)one;
  }
  /* ... and so is a measured state outside the boxes on x, y, v, delta by more than constr_viol_tol (stage 0 is bounded like every
   * other stage, vehicle_follower.py:205-240, and pinned to the measurement, :194-199) */
  for (int q = 0; q < 4; ++q) {
    const double v = x0[BCOL[q]];
    if (v < sp->bounds[2 * q] - sp->constr_viol_tol || v > sp->bounds[2 * q + 1] + sp->constr_viol_tol) {
      stats[0] = 0; stats[1] = 4; fstats[0] = 0.0; fstats[1] = INFINITY; fstats[2] = sp->mu_init;
      if (sep_out) for (int q2 = 0; q2 < N * nblk; ++q2) sep_out[q2] = 0.0;
      if (cert_out) for (int q2 = 0; q2 < N * nblk; ++q2) cert_out[q2] = 0;
      return 0;
    }
  }
  select_all(sp, nbr, it.p, sel);
  eval_rows(sp, nbr, it.p, sel, sep, 0, 0);
  for (int k = 0; k < N; ++k) {
    for (int q = 0; q < 6; ++q) {
      double lo = sp->bounds[2 * q], hi = sp->bounds[2 * q + 1];
      double pl = fmin(sp->bound_push * fmax(1.0, fabs(lo)), sp->bound_frac * (hi - lo));
      double pu = fmin(sp->bound_push * fmax(1.0, fabs(hi)), sp->bound_frac * (hi - lo));
      double v = it.p[k][BCOL[q]];
      v = fmax(v, lo + pl); v = fmin(v, hi - pu);
      it.p[k][BCOL[q]] = v; it.zl[k][q] = 1.0; it.zu[k][q] = 1.0;
    }
    for (int j = 0; j < nb; ++j) {
      it.sg[k][j] = fmax(sep[k][j] - sp->dmin, sp->bound_push);
      it.zs[k][j] = 1.0; it.nuc[k][j] = 0.0;
    }
    for (int i = 0; i < 5; ++i) it.pi[k][i] = 0.0;
  }
  for (int i = 0; i < 5; ++i) it.pi0[i] = 0.0;
  if (cin && cin->valid) {
    /* start from the previous MPC iteration's multipliers (oracle/mpc_nlp.py warm_from_carry): the horizon has
     * moved on by one stage, new stage k takes old stage min(k+1, N-1) */
    mu = fmin(fmax(cin->mu, mu_floor), sp->mu_init);
    for (int k = 0; k < N; ++k) {
      const int ko = k + 1 < N ? k + 1 : N - 1;
      for (int q = 0; q < 6; ++q) {
        double lo = sp->bounds[2 * q], hi = sp->bounds[2 * q + 1];
        it.p[k][BCOL[q]] = fmin(fmax(p_io[k * NP + BCOL[q]], lo + sp->warm_push), hi - sp->warm_push);
        it.zl[k][q] = fmax(cin->zl[ko][q], mu / (hi - lo)); it.zu[k][q] = fmax(cin->zu[ko][q], mu / (hi - lo));
      }
      for (int j = 0; j < nblk; ++j) {
        const int so = cin->sel[ko][j], sn = sel[k][j];
        for (int r = 0; r < 2; ++r) {
          const int vn = r == 0 ? (sn >> 2) & 3 : sn & 3;
          double z = 0.0;
          if ((so >> 4) == (sn >> 4)) {
            if (((so >> 2) & 3) == vn) z = cin->z[ko][2 * j];
            else if ((so & 3) == vn) z = cin->z[ko][2 * j + 1];
          }
          const double gap = sep[k][2 * j + r] - sp->dmin;
          double sg;
          if (z > 0.0) sg = fmax(fmax(gap, mu / z), sp->warm_push);
          else { sg = fmax(gap, sp->bound_push); z = mu / sg; }
          it.sg[k][2 * j + r] = sg; it.zs[k][2 * j + r] = z; it.nuc[k][2 * j + r] = -z;
        }
      }
    }
    for (int i = 0; i < 5; ++i) it.pi0[i] = cin->pi[0][i];
    for (int k = 0; k + 1 < N; ++k) { const int ko = k + 1 < N - 1 ? k + 1 : N - 2; for (int i = 0; i < 5; ++i) it.pi[k][i] = cin->pi[ko][i]; }
  }

  if (sp->resto > 0 && sp->resto_first > 0.0) {
    /* a start whose rows are violated by more than resto_first goes through the restoration phase first (cold multipliers after it) */
    double v0 = 0.0;
    eval_rows(sp, nbr, it.p, sel, sep, 0, 0); /* at the pushed start, working set of the un-pushed one: what the first iteration sees */
    for (int k = 1; k < N; ++k) for (int j = 0; j < nb; ++j) v0 = fmax(v0, sp->dmin - sep[k][j]);
    if (v0 > sp->resto_first) {
      int rit = 0;
      if (restore(sp, x0, nbr, mu, &rit)) {
        cold_multipliers(sp, nbr, mu);
      } else {
        locally_infeasible = 1; /* IPOPT: "converged to a point of local infeasibility" */
      }
      iter0 = rit;
    }
  }
  if (locally_infeasible) { status = iter0 >= sp->max_iter ? 1 : 5; iter = iter0; } /* (out of iterations inside the restoration: the limit, not local infeasibility) */
  for (iter = iter0; iter <= sp->max_iter && !locally_infeasible; ++iter) {
    /* ---- working set: rows keep slack and multipliers while their (face, vertex) identity lasts */
    int ws_changed = 0;
    if (iter > iter0) {
      static int old[MAXN][MAXB];
      memcpy(old, sel, sizeof sel);
      select_all(sp, nbr, it.p, sel);
      eval_rows(sp, nbr, it.p, sel, sep, 0, 0);
      for (int k = 0; k < N; ++k)
        for (int j = 0; j < nblk; ++j) {
          int o = old[k][j], n_ = sel[k][j];
          if (o == n_) continue;
          ws_changed = 1;
          int same_face = (o >> 4) == (n_ >> 4);
          double vs[2][3] = {{it.sg[k][2 * j], it.zs[k][2 * j], it.nuc[k][2 * j]},
                             {it.sg[k][2 * j + 1], it.zs[k][2 * j + 1], it.nuc[k][2 * j + 1]}};
          int ov[2] = {(o >> 2) & 3, o & 3}, nv[2] = {(n_ >> 2) & 3, n_ & 3};
          for (int r = 0; r < 2; ++r) {
            int src = -1;
            if (same_face) { if (nv[r] == ov[0]) src = 0; else if (nv[r] == ov[1]) src = 1; }
            if (src >= 0) { it.sg[k][2 * j + r] = vs[src][0]; it.zs[k][2 * j + r] = vs[src][1]; it.nuc[k][2 * j + r] = vs[src][2]; }
            else {
              double sg = fmax(sep[k][2 * j + r] - sp->dmin, sp->bound_push);
              it.sg[k][2 * j + r] = sg; it.zs[k][2 * j + r] = mu / sg; it.nuc[k][2 * j + r] = -mu / sg;
            }
          }
        }
    }
    /* ---- evaluate ----------------------------------------------------------------- */
    eval_rows(sp, nbr, it.p, sel, sep, gra, cur);
    double cviol = 0.0, theta = 0.0;
    for (int i = 0; i < 5; ++i) { double r = it.p[0][i] - x0[i]; cviol = fmax(cviol, fabs(r)); theta += fabs(r); }
    for (int k = 0; k + 1 < N; ++k) {
      rk4_sens(it.p[k], it.p[k] + 5, sp->dt, sp->wb, sp->rk_substeps, Fk[k], Ak[k], Bk[k]);
      for (int i = 0; i < 5; ++i) { dk[k][i] = Fk[k][i] - it.p[k + 1][i]; cviol = fmax(cviol, fabs(dk[k][i])); theta += fabs(dk[k][i]); }
    }
    /* The pose of stage 0 is the measurement (pinned by z_0 = x0): its rows are constants -- either satisfied or violated by less than
     * constr_viol_tol (the pre-check above) -- and take no part in the iteration: zero residual, zero gradient.  (As rows of the variable
     * z_0 they fought the initial-state row whenever a parked vehicle sat a centimetre inside a clearance: both multipliers ran away and
     * every solve of that vehicle ended in a failed line search after 29 iterations.) */
    for (int j = 0; j < nb; ++j) { for (int a = 0; a < 3; ++a) gra[0][j][a] = 0.0; for (int a = 0; a < 6; ++a) cur[0][j][a] = 0.0; }
    for (int k = 0; k < N; ++k)
      for (int j = 0; j < nb; ++j) { cj[k][j] = k == 0 ? 0.0 : sep[k][j] - sp->dmin - it.sg[k][j]; cviol = fmax(cviol, fabs(cj[k][j])); theta += fabs(cj[k][j]); }
    if (theta_min < 0.0) { theta_min = 1e-4 * fmax(1.0, theta); theta_max = 1e4 * fmax(1.0, theta); }
    /* dual infeasibility, multiplier sums, complementarity */
    double dual_inf = 0.0, sum_nu = 0.0, sum_z = 0.0, fval = 0.0;
    double cmp0 = 0.0, cmpmu = 0.0; /* max |d*z - 0| and max |d*z - mu| */
    for (int i = 0; i < 5; ++i) sum_nu += fabs(it.pi0[i]);
    for (int k = 0; k < N; ++k) {
      double gr[NP], r[NP];
      stage_grad(sp, ref, k, it.p[k], gr);
      fval += stage_cost(sp, ref, k, it.p[k]);
      for (int i = 0; i < NP; ++i) r[i] = gr[i];
      for (int j = 0; j < nb; ++j) {
        for (int q = 0; q < 3; ++q) r[q] += gra[k][j][q] * it.nuc[k][j];
        dual_inf = fmax(dual_inf, fabs(-it.nuc[k][j] - it.zs[k][j]));
        sum_nu += fabs(it.nuc[k][j]); sum_z += it.zs[k][j];
        double cz = it.sg[k][j] * it.zs[k][j];
        cmp0 = fmax(cmp0, fabs(cz)); cmpmu = fmax(cmpmu, fabs(cz - mu));
      }
      if (k + 1 < N) {
        for (int i = 0; i < 5; ++i) {
          sum_nu += fabs(it.pi[k][i]);
          for (int q = 0; q < 5; ++q) r[q] += Ak[k][i][q] * it.pi[k][i];
          for (int q = 0; q < 2; ++q) r[5 + q] += Bk[k][i][q] * it.pi[k][i];
        }
      }
      if (k == 0) for (int i = 0; i < 5; ++i) r[i] += it.pi0[i];
      else for (int i = 0; i < 5; ++i) r[i] -= it.pi[k - 1][i];
      for (int q = 0; q < 6; ++q) {
        r[BCOL[q]] += -it.zl[k][q] + it.zu[k][q];
        sum_z += it.zl[k][q] + it.zu[k][q];
        double dl = it.p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - it.p[k][BCOL[q]];
        cmp0 = fmax(cmp0, fmax(fabs(dl * it.zl[k][q]), fabs(du * it.zu[k][q])));
        cmpmu = fmax(cmpmu, fmax(fabs(dl * it.zl[k][q] - mu), fabs(du * it.zu[k][q] - mu)));
      }
      for (int i = 0; i < NP; ++i) dual_inf = fmax(dual_inf, fabs(r[i]));
    }
    double s_d = fmax(sp->s_max, (sum_nu + sum_z) / (double)(m_eq + n_bnd)) / sp->s_max;
    double s_c = fmax(sp->s_max, sum_z / (double)n_bnd) / sp->s_max;
    err0 = fmax(dual_inf / s_d, fmax(cviol, cmp0 / s_c));
    if (trace && iter < trace_cap) {
      double *tr = trace + (size_t)iter * (4 + NP * N);
      tr[0] = mu; tr[1] = err0; tr[2] = cviol; tr[3] = dual_inf;
      for (int k = 0; k < N; ++k) for (int i = 0; i < NP; ++i) tr[4 + k * NP + i] = it.p[k][i];
    }
    if (!isfinite(err0)) { status = 3; break; }
    if (err0 <= sp->tol && dual_inf <= sp->dual_inf_tol && cviol <= sp->constr_viol_tol && cmp0 <= sp->compl_inf_tol) { status = 0; break; }
    if (iter == sp->max_iter) break;
    if (iter == iter0 || err0 < 0.5 * best_err) { best_err = err0; best_it = iter; }
    if (sp->stag_win > 0 && !stagnant && cviol <= sp->constr_viol_tol && iter - best_it >= sp->stag_win) stagnant = 1;
    if (sp->err_stall > 0 && iter - best_it >= sp->err_stall) { status = 5; break; }
    /* infeasibility stall (oracle/ipm.py) */
    /* an iterate that changed the working set counts a quarter (its new rows start with their own violation; but a solve that
     * changes it at EVERY iterate cycles, and must end) */
    if (iter == iter0 || cviol <= sp->stall_kappa * stall_ref) { stall_ref = cviol; stall_cnt = 0; stall_ws = 0; }
    else if (!ws_changed) ++stall_cnt;
    else if (++stall_ws >= WS_STALL_DIV) { stall_ws = 0; ++stall_cnt; }
    if (sp->stall_iters > 0 && stall_cnt >= sp->stall_iters && cviol > sp->constr_viol_tol) { status = 5; break; }
    /* ---- barrier update ------------------------------------------------------------- */
    while (mu > mu_floor) {
      /* complementarity error against the current mu has to be recomputed for each candidate mu */
      double cm = 0.0;
      for (int k = 0; k < N; ++k) {
        for (int j = 0; j < nb; ++j) cm = fmax(cm, fabs(it.sg[k][j] * it.zs[k][j] - mu));
        for (int q = 0; q < 6; ++q) {
          double dl = it.p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - it.p[k][BCOL[q]];
          cm = fmax(cm, fmax(fabs(dl * it.zl[k][q] - mu), fabs(du * it.zu[k][q] - mu)));
        }
      }
      double emu = fmax(dual_inf / s_d, fmax(cviol, cm / s_c));
      if (emu <= sp->kappa_eps * mu) mu = fmax(mu_floor, fmin(sp->kappa_mu * mu, pow(mu, sp->theta_mu)));
      else break;
    }
    (void)cmpmu;
    double tau = fmax(sp->tau_min, 1.0 - mu);
    /* ---- condensed stage QP ----------------------------------------------------------- */
    double dphi = 0.0;
    for (int k = 0; k < N; ++k) {
      const double *w = sp->weights; const double *p = it.p[k];
      memset(H[k], 0, sizeof H[k]);
      stage_grad(sp, ref, k, p, gphi[k]);
      H[k][0][0] = 2 * w[0]; H[k][1][1] = 2 * w[1]; H[k][2][2] = 2 * w[2]; H[k][4][4] = 2 * w[5]; H[k][5][5] = 2 * w[3];
      H[k][3][3] = 2 * w[4] * p[6] * p[6]; H[k][6][6] = 2 * w[4] * p[3] * p[3];
      H[k][3][6] = H[k][6][3] = 2 * w[4] * p[3] * p[6];
      for (int i = 0; i < NP; ++i) H[k][i][i] += sp->reg_primal;
      for (int q = 0; q < 6; ++q) {
        double dl = p[BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - p[BCOL[q]];
        H[k][BCOL[q]][BCOL[q]] += it.zl[k][q] / dl + it.zu[k][q] / du;
        gphi[k][BCOL[q]] += -mu / dl + mu / du;
      }
      for (int i = 0; i < NP; ++i) gk[k][i] = gphi[k][i];
      for (int j = 0; j < nb; ++j) {
        /* delta_c on the row (IPOPT's dual regularisation, eliminated with the slack): stiffness S0 / (1 + delta_c S0) */
        const double S0 = it.zs[k][j] / it.sg[k][j] + sp->reg_primal, iD = 1.0 / (1.0 + sp->reg_dual_rows * S0), S = S0 * iD;
        double coef = S * cj[k][j] - mu / it.sg[k][j] * iD + sp->reg_dual_rows * S * it.nuc[k][j];
        for (int a = 0; a < 3; ++a) {
          gk[k][a] += gra[k][j][a] * coef;
          for (int b = 0; b < 3; ++b) H[k][a][b] += S * gra[k][j][a] * gra[k][j][b];
        }
      }
      if (sp->row_curvature) {
        /* exact curvature of the separation rows, sum_r nu_r d2 sep_r: C = [[0,0,a],[0,0,b],[a,b,c]], scaled by
         * th in {1, 1/2, .., 2^-9, 0} so that diag(2 w) + th C keeps the margin 0.2 min(w) (oracle/mpc_nlp.py hess_gn) */
        double ca = 0.0, cb = 0.0, cc = 0.0, cxx = 0.0, cyy = 0.0, cxy = 0.0;
        for (int j = 0; j < nb; ++j) {
          const double nu = it.nuc[k][j];
          ca += nu * cur[k][j][0]; cb += nu * cur[k][j][1]; cc += nu * cur[k][j][2];
          cxx += nu * cur[k][j][3]; cyy += nu * cur[k][j][4]; cxy += nu * cur[k][j][5];
        }
        const double m_ = 0.2 * fmin(w[0], fmin(w[1], w[2]));
        const double q0 = 2 * w[0] - m_, q1 = 2 * w[1] - m_, q2 = 2 * w[2] - m_;
        const int full = cxx != 0.0 || cyy != 0.0 || cxy != 0.0; /* a vertex-vertex row curves x and y too */
        double th = 1.0;
        for (int h = 0; h < 11; ++h) {
          if (h == 10) { th = 0.0; break; }
          if (!full) {
            if (q2 + th * cc - th * th * (ca * ca / q0 + cb * cb / q1) >= 0.0) break;
          } else { /* diag(q) + th C positive semidefinite: leading principal minors */
            const double m00 = q0 + th * cxx, m11 = q1 + th * cyy, m22 = q2 + th * cc, m01 = th * cxy, m02 = th * ca, m12 = th * cb;
            const double d2 = m00 * m11 - m01 * m01;
            const double d3 = m22 * d2 - (m02 * m02 * m11 - 2.0 * m02 * m12 * m01 + m12 * m12 * m00);
            if (m00 > 0.0 && d2 > 0.0 && d3 >= 0.0) break;
          }
          th *= 0.5;
        }
        if (sp->shift_after > 0 && (iter >= sp->shift_after || (stagnant && iter >= SHIFT_STAG_MIN) || shift_hint) && th < 1.0) {
          shift_used = 1;
          /* late in a long solve the scaled model cycles: whole curvature + smallest identity shift (hess_gn shift=True) */
          const double dl_ = pose_shift(q0 + cxx, q1 + cyy, q2 + cc, cxy, ca, cb);
          H[k][0][0] += dl_; H[k][1][1] += dl_; H[k][2][2] += dl_;
          th = 1.0;
        }
        H[k][0][2] += th * ca; H[k][2][0] += th * ca; H[k][1][2] += th * cb; H[k][2][1] += th * cb; H[k][2][2] += th * cc;
        H[k][0][0] += th * cxx; H[k][1][1] += th * cyy; H[k][0][1] += th * cxy; H[k][1][0] += th * cxy;
      }
    }
    riccati_backward(N, sp->dt);
    /* ---- forward sweep: dp, new multipliers ------------------------------------------------ */
    riccati_forward(N, x0);
    for (int i = 0; i < 5; ++i) dt_.pi0[i] = pi0new[i] - it.pi0[i];
    for (int k = 0; k + 1 < N; ++k) for (int i = 0; i < 5; ++i) dt_.pi[k][i] = pinew[k][i] - it.pi[k][i];
    /* ---- slack / multiplier steps, fraction to the boundary ----------------------------------- */
    double a_pri = 1.0, a_dual = 1.0;
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NP; ++i) dphi += gphi[k][i] * dt_.p[k][i];
      for (int j = 0; j < nb; ++j) {
        double ds = cj[k][j]; for (int a = 0; a < 3; ++a) ds += gra[k][j][a] * dt_.p[k][a];
        double S = it.zs[k][j] / it.sg[k][j] + sp->reg_primal;
        ds = (ds + sp->reg_dual_rows * (mu / it.sg[k][j] + it.nuc[k][j])) / (1.0 + sp->reg_dual_rows * S);
        dt_.sg[k][j] = ds;
        dt_.nuc[k][j] = S * ds - mu / it.sg[k][j] - it.nuc[k][j];
        dt_.zs[k][j] = mu / it.sg[k][j] - it.zs[k][j] - it.zs[k][j] / it.sg[k][j] * ds;
        dphi += -mu / it.sg[k][j] * ds;
        if (ds < 0.0) a_pri = fmin(a_pri, -tau * it.sg[k][j] / ds);
        if (dt_.zs[k][j] < 0.0) a_dual = fmin(a_dual, -tau * it.zs[k][j] / dt_.zs[k][j]);
      }
      for (int q = 0; q < 6; ++q) {
        double dx = dt_.p[k][BCOL[q]];
        double dl = it.p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - it.p[k][BCOL[q]];
        dt_.zl[k][q] = mu / dl - it.zl[k][q] - it.zl[k][q] / dl * dx;
        dt_.zu[k][q] = mu / du - it.zu[k][q] + it.zu[k][q] / du * dx;
        if (dx < 0.0) a_pri = fmin(a_pri, -tau * dl / dx);
        if (dx > 0.0) a_pri = fmin(a_pri, tau * du / dx);
        if (dt_.zl[k][q] < 0.0) a_dual = fmin(a_dual, -tau * it.zl[k][q] / dt_.zl[k][q]);
        if (dt_.zu[k][q] < 0.0) a_dual = fmin(a_dual, -tau * it.zu[k][q] / dt_.zu[k][q]);
      }
    }
    /* ---- filter line search ---------------------------------------------------------------- */
    double phi0;
    {
      double th_dummy;
      if (!merit_terms(sp, x0, ref, nbr, it.p, it.sg, sel, mu, &th_dummy, &phi0)) { status = 3; break; }
    }
    if (filt_mu != mu) { nfilt = 0; filt_mu = mu; }
    if (ws_changed) nfilt = 0; /* the entries belong to the problem with the previous working set (oracle/ipm.py) */
    double alpha = a_pri; int accepted = 0, f_type = 0;
    for (int bt = 0; bt < sp->max_backtrack; ++bt) {
      for (int k = 0; k < N; ++k) {
        for (int i = 0; i < NP; ++i) pt[k][i] = it.p[k][i] + alpha * dt_.p[k][i];
        for (int j = 0; j < nb; ++j) sgt[k][j] = it.sg[k][j] + alpha * dt_.sg[k][j];
      }
      double th_t, ph_t;
      int ok = merit_terms(sp, x0, ref, nbr, pt, sgt, sel, mu, &th_t, &ph_t) && th_t <= theta_max;
      if (ok) for (int q = 0; q < nfilt; ++q) if (th_t >= filt[q][0] && ph_t >= filt[q][1]) { ok = 0; break; }
      f_type = 0;
      if (ok) {
        int sw = theta <= theta_min && dphi < 0.0 && alpha * pow(-dphi, sp->s_phi) > sp->delta_sw * pow(theta, sp->s_theta);
        if (sw) { f_type = 1; ok = ph_t <= phi0 + sp->eta_phi * alpha * dphi; }
        else ok = th_t <= (1.0 - sp->gamma_theta) * theta || ph_t <= phi0 - sp->gamma_phi * theta;
      }
      if (ok) { accepted = 1; break; }
      alpha *= 0.5;
    }
    if (!accepted) {
      /* IPOPT's answer to a failed line search: the restoration phase, then on with fresh multipliers and an empty filter */
      if (sp->resto > 0 && resto_calls < sp->resto && cviol > sp->constr_viol_tol) {
        if (restore(sp, x0, nbr, mu, &iter)) {
          cold_multipliers(sp, nbr, mu);
          ++resto_calls;
          nfilt = 0; stall_ref = INFINITY; stall_cnt = 0; stall_ws = 0; best_err = INFINITY; best_it = iter;
          continue;
        }
        status = iter >= sp->max_iter ? 1 : 5; break; /* the restoration failed: locally infeasible -- or simply out of iterations */
      }
      status = 2; break;
    }
    if (!f_type) {
      if (nfilt == sp->filter_cap) { memmove(filt, filt + 1, sizeof(double) * 2 * (nfilt - 1)); nfilt--; }
      filt[nfilt][0] = (1.0 - sp->gamma_theta) * theta; filt[nfilt][1] = phi0 - sp->gamma_phi * theta; nfilt++;
    }
    /* ---- update ------------------------------------------------------------------------------ */
    for (int i = 0; i < 5; ++i) it.pi0[i] += alpha * dt_.pi0[i];
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NP; ++i) it.p[k][i] = pt[k][i];
      if (k + 1 < N) for (int i = 0; i < 5; ++i) it.pi[k][i] += alpha * dt_.pi[k][i];
      for (int j = 0; j < nb; ++j) {
        it.sg[k][j] = sgt[k][j];
        it.nuc[k][j] += alpha * dt_.nuc[k][j];
        double z = it.zs[k][j] + a_dual * dt_.zs[k][j];
        it.zs[k][j] = fmin(fmax(z, mu / (sp->kappa_sigma * it.sg[k][j])), sp->kappa_sigma * mu / it.sg[k][j]);
      }
      for (int q = 0; q < 6; ++q) {
        double dl = it.p[k][BCOL[q]] - sp->bounds[2 * q], du = sp->bounds[2 * q + 1] - it.p[k][BCOL[q]];
        double z = it.zl[k][q] + a_dual * dt_.zl[k][q];
        it.zl[k][q] = fmin(fmax(z, mu / (sp->kappa_sigma * dl)), sp->kappa_sigma * mu / dl);
        z = it.zu[k][q] + a_dual * dt_.zu[k][q];
        it.zu[k][q] = fmin(fmax(z, mu / (sp->kappa_sigma * du)), sp->kappa_sigma * mu / du);
      }
    }
  }
  /* ---- output ---------------------------------------------------------------------------------- */
  select_all(sp, nbr, it.p, sel);
  eval_rows(sp, nbr, it.p, sel, sep, 0, 0);
  double fval = 0.0;
  for (int k = 0; k < N; ++k) {
    fval += stage_cost(sp, ref, k, it.p[k]);
    for (int i = 0; i < NP; ++i) p_io[k * NP + i] = it.p[k][i];
    for (int j = 0; j < nblk; ++j) {
      int c = sel[k][j], near = sep[k][2 * j] <= sep[k][2 * j + 1] ? (c >> 2) & 3 : c & 3;
      if (sep_out) sep_out[k * nblk + j] = fmin(sep[k][2 * j], sep[k][2 * j + 1]);
      if (cert_out) cert_out[k * nblk + j] = (c >> 6) * 16 + ((c >> 4) & 3) * 4 + near;
    }
  }
  stats[0] = iter; stats[1] = status;
  fstats[0] = fval; fstats[1] = err0; fstats[2] = mu;
  if (cout) cout->shifted = status == 0 ? shift_used : 0;
  if (cout && status == 0) {
    cout->valid = 1; cout->mu = mu;
    for (int k = 0; k < N; ++k) {
      for (int j = 0; j < nblk; ++j) cout->sel[k][j] = sel[k][j];
      for (int j = 0; j < nb; ++j) cout->z[k][j] = it.zs[k][j];
      for (int q = 0; q < 6; ++q) { cout->zl[k][q] = it.zl[k][q]; cout->zu[k][q] = it.zu[k][q]; }
      for (int i = 0; i < 5; ++i) cout->pi[k][i] = k + 1 < N ? it.pi[k][i] : 0.0;
    }
    for (int i = 0; i < 5; ++i) cout->pi0[i] = it.pi0[i];
  }
  return 0;
}

int cfz_port_solve(const cfz_port_spec *sp, const double *x0, const double *ref, const double *nbr, double *p_io,
                   double *sep_out, int *cert_out, int *stats, double *fstats, double *trace, int trace_cap) {
  return cfz_port_solve_carry(sp, x0, ref, nbr, p_io, sep_out, cert_out, stats, fstats, trace, trace_cap, 0, 0);
}

int cfz_port_sizeof_carry(void) { return (int)sizeof(cfz_port_carry); }

int cfz_port_sizeof_spec(void) { return (int)sizeof(cfz_port_spec); }
