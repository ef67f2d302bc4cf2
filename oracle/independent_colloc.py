"""Oracle (test infrastructure, PARITY UNPINNED like the rest of oracle/): the single-vehicle collocation plan of
`Vehicle.setup_single_final_problem` / `solve_single_final_problem` (reference confrez/control/vehicle.py:360-661) solved
INDEPENDENTLY of the planning kernel's formulation and algorithm, at the reference's size.

Statement: the reference's rows in the primal variables alone -- (x, y, psi, v, delta, a, w) at the 6 N collocation points and
the free interval length dt -- with the OBCA rows (:523-541: "there are duals certifying a separation >= dmin") replaced by
what they are equivalent to, dist(body(pose), obstacle_j) >= dmin (strong duality of the polygon distance problem; same step
as oracle/independent_mpc.py).  No duals, no working sets, no slacks, no condensation, no bordering of dt.
  equalities    initial pose and v = delta = a = w = 0 (:426-436); the ODE at all six points of every interval,
                sum_j A[j,k] z_ij - dt f(z_ik, u_ik) = 0 (:487-509, times dt); continuity of states AND inputs through D
                (:544-568); terminal v = delta = a = w = 0 (+ heading) at the end state D'Z_{N-1} (:590-626)
  inequalities  tube rows at every N_per_set-th interval start and at the end state (:570-617); obstacle distances at every
                point; boxes on x, y, v, delta, a, w (:439-478)
  cost          sum B_k (a^2 + v^2 w^2 + delta^2) dt + (N dt)^2 (:512-521, :638)
Solvers: `solve` = scipy's SLSQP with analytic Jacobians (distances: central differences in the pose they depend on) -- fine on
short tubes, but at the reference's size (1261 variables, 1128 inequality rows) its dense QP does not finish within an hour;
`solve_ipm` = oracle/ipm.py (sparse full-KKT interior point, SuperLU) on the same statement in slack form with its own exact
Hessian (`GeometricCollocIpm`), 90 s at full size: what tests/golden/make_independent_colloc.py runs.
"""
import numpy as np
from scipy.optimize import minimize

from .colloc_nlp import K_PTS, radau_tables
from .independent_mpc import _body_vertices, polygon_distance_batch
from .mpc_nlp import polytope_vertices


class GeometricColloc:
    def __init__(self, init_pose, tube, A_obs, b_obs, N_per_set=5, dmin=0.05, shrink_tube=0.5, final_heading=None, wb=2.5,
                 g=(3.3, 0.9, 0.6, 0.9), bounds=(2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0)):
        self.init_pose, self.tube, self.fh = np.asarray(init_pose, float), tube, final_heading
        self.Nps, self.n_chk = N_per_set, len(tube) - 1
        self.N = N_per_set * self.n_chk
        self.dmin, self.shrink, self.wb, self.g, self.bd = dmin, shrink_tube, wb, np.asarray(g, float), np.asarray(bounds, float)
        self.tau, self.A, self.B, self.D = radau_tables(5)
        self.obs = np.stack([polytope_vertices(A, b)[0] for A, b in zip(A_obs, b_obs)])
        self.np_ = self.N * K_PTS
        self.n = 7 * self.np_ + 1

    def split(self, z):
        return np.asarray(z[:-1], float).reshape(self.N, K_PTS, 7), float(z[-1])

    def idx(self, i, k, c):
        return ((i * K_PTS + k) * 7) + c

    # ---- cost ----------------------------------------------------------------------------------------
    def cost(self, z):
        P, dt = self.split(z)
        e = P[..., 5] ** 2 + P[..., 3] ** 2 * P[..., 6] ** 2 + P[..., 4] ** 2
        return float((e * self.B[None, :]).sum() * dt + (self.N * dt) ** 2)

    def cost_grad(self, z):
        P, dt = self.split(z)
        g = np.zeros_like(P)
        Bk = self.B[None, :] * dt
        g[..., 5] = 2 * P[..., 5] * Bk; g[..., 4] = 2 * P[..., 4] * Bk
        g[..., 3] = 2 * P[..., 3] * P[..., 6] ** 2 * Bk; g[..., 6] = 2 * P[..., 3] ** 2 * P[..., 6] * Bk
        e = P[..., 5] ** 2 + P[..., 3] ** 2 * P[..., 6] ** 2 + P[..., 4] ** 2
        return np.append(g.ravel(), (e * self.B[None, :]).sum() + 2 * self.N ** 2 * dt)

    # ---- equalities ------------------------------------------------------------------------------------
    def _f(self, P):
        psi, v, de = P[..., 2], P[..., 3], P[..., 4]
        return np.stack([v * np.cos(psi), v * np.sin(psi), v / self.wb * np.tan(de), P[..., 5], P[..., 6]], -1)

    def eq(self, z):
        P, dt = self.split(z)
        Z = P[..., :5]
        out = [P[0, 0, :3] - self.init_pose, P[0, 0, 3:]]
        ode = np.einsum("jk,ijc->ikc", self.A, Z) - dt * self._f(P)  # [N, 6, 5]
        out.append(ode.ravel())
        out.append((np.einsum("j,ijc->ic", self.D, P[:-1]) - P[1:, 0]).ravel())
        zF = self.D @ P[-1]
        out.append(zF[3:])
        if self.fh is not None:
            out.append([zF[2] - self.fh])
        return np.concatenate([np.ravel(o) for o in out])

    def eq_jac(self, z):
        P, dt = self.split(z)
        N, n = self.N, self.n
        rows = []
        J0 = np.zeros((7, n)); J0[np.arange(7), np.arange(7)] = 1.0
        rows.append(J0)
        psi, v, de = P[..., 2], P[..., 3], P[..., 4]
        c, s, t = np.cos(psi), np.sin(psi), np.tan(de)
        f = self._f(P)
        Jo = np.zeros((N, K_PTS, 5, n))
        for i in range(N):
            for k in range(K_PTS):
                for j in range(K_PTS):
                    for cc in range(5):
                        Jo[i, k, cc, self.idx(i, j, cc)] += self.A[j, k]
                b = self.idx(i, k, 0)
                Jo[i, k, 0, b + 2] -= dt * (-v[i, k] * s[i, k]); Jo[i, k, 0, b + 3] -= dt * c[i, k]
                Jo[i, k, 1, b + 2] -= dt * (v[i, k] * c[i, k]); Jo[i, k, 1, b + 3] -= dt * s[i, k]
                Jo[i, k, 2, b + 3] -= dt * t[i, k] / self.wb; Jo[i, k, 2, b + 4] -= dt * v[i, k] / self.wb * (1 + t[i, k] ** 2)
                Jo[i, k, 3, b + 5] -= dt; Jo[i, k, 4, b + 6] -= dt
                Jo[i, k, :, n - 1] = -f[i, k]
        rows.append(Jo.reshape(-1, n))
        Jc = np.zeros((N - 1, 7, n))
        for i in range(1, N):
            for cc in range(7):
                for j in range(K_PTS):
                    Jc[i - 1, cc, self.idx(i - 1, j, cc)] += self.D[j]
                Jc[i - 1, cc, self.idx(i, 0, cc)] -= 1.0
        rows.append(Jc.reshape(-1, n))
        Jt = np.zeros((4, n))
        for q, cc in enumerate((3, 4, 5, 6)):
            for j in range(K_PTS):
                Jt[q, self.idx(N - 1, j, cc)] = self.D[j]
        rows.append(Jt)
        if self.fh is not None:
            Jh = np.zeros((1, n))
            for j in range(K_PTS):
                Jh[0, self.idx(N - 1, j, 2)] = self.D[j]
            rows.append(Jh)
        return np.vstack(rows)

    # ---- inequalities (>= 0) ---------------------------------------------------------------------------
    def _tube_rows(self, p, q):
        front = p[:2] + self.wb * np.array([np.cos(p[2]), np.sin(p[2])])
        (Ab, bb), (Af, bf) = self.tube[q]["back"], self.tube[q]["front"]
        return np.concatenate([(np.asarray(bb) - self.shrink) - np.asarray(Ab) @ p[:2], (np.asarray(bf) - self.shrink) - np.asarray(Af) @ front])

    def separations(self, poses):
        W = _body_vertices(poses[:, 0], poses[:, 1], poses[:, 2], self.g)
        return polygon_distance_batch(self.obs[None], W[:, None])

    def ineq(self, z):
        P, _ = self.split(z)
        out = [self._tube_rows(P[q * self.Nps, 0], q) for q in range(1, self.n_chk)]
        out.append(self._tube_rows(self.D @ P[-1], self.n_chk))
        out.append((self.separations(P.reshape(-1, 7)[:, :3]) - self.dmin).ravel())
        return np.concatenate(out)

    def ineq_jac(self, z, h=1e-6):
        P, _ = self.split(z)
        n, rows = self.n, []

        def tube_jac(p):
            (Ab, _), (Af, _) = self.tube_q["back"], self.tube_q["front"]
            Ab, Af = np.asarray(Ab), np.asarray(Af)
            J = np.zeros((8, 3))
            J[:4, :2] = -Ab
            J[4:, :2] = -Af
            J[4:, 2] = -Af @ (self.wb * np.array([-np.sin(p[2]), np.cos(p[2])]))
            return J

        for q in range(1, self.n_chk):
            self.tube_q = self.tube[q]
            Jq = np.zeros((8, n)); b = self.idx(q * self.Nps, 0, 0)
            Jq[:, b:b + 3] = tube_jac(P[q * self.Nps, 0])
            rows.append(Jq)
        self.tube_q = self.tube[self.n_chk]
        Jp = tube_jac(self.D @ P[-1])
        Jq = np.zeros((8, n))
        for j in range(K_PTS):
            b = self.idx(self.N - 1, j, 0)
            Jq[:, b:b + 3] += self.D[j] * Jp
        rows.append(Jq)
        poses = P.reshape(-1, 7)[:, :3]
        no = len(self.obs)
        Js = np.zeros((self.np_ * no, n))
        for c in range(3):
            e = np.zeros(3); e[c] = h
            d = (self.separations(poses + e) - self.separations(poses - e)) / (2 * h)
            for q in range(self.np_):
                Js[q * no:(q + 1) * no, 7 * q + c] = d[q]
        rows.append(Js)
        return np.vstack(rows)

    def bounds(self):
        lo, hi = np.full((self.np_, 7), -np.inf), np.full((self.np_, 7), np.inf)
        for c, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
            lo[:, c], hi[:, c] = self.bd[2 * j], self.bd[2 * j + 1]
        return list(zip(lo.ravel(), hi.ravel())) + [(1e-3, None)]


def solve(nlp: GeometricColloc, guess, dt0, maxiter=300, ftol=1e-12, verbose=False):
    """SLSQP from guess [6 N, 7] and dt0.  Returns dict(traj [N, 6, 7], dt, cost, status, iters, eq, ineq)."""
    z0 = np.append(np.asarray(guess, float).ravel(), dt0)
    out = minimize(nlp.cost, z0, jac=nlp.cost_grad, method="SLSQP", bounds=nlp.bounds(),
                   constraints=[dict(type="eq", fun=nlp.eq, jac=nlp.eq_jac), dict(type="ineq", fun=nlp.ineq, jac=nlp.ineq_jac)],
                   options=dict(maxiter=maxiter, ftol=ftol, disp=verbose))
    P, dt = nlp.split(out.x)
    return dict(traj=P, dt=dt, cost=float(out.fun), status=int(out.status), iters=int(out.nit),
                eq=float(np.abs(nlp.eq(out.x)).max()), ineq=float(nlp.ineq(out.x).min()))


# ---- the same statement in the form oracle/ipm.py solves (sparse full-KKT interior point, exact Hessian) ----------------------
class GeometricCollocIpm:
    """`GeometricColloc` as  min f(X) s.t. c(X) = 0, XL <= X <= XU  with X = [z | slacks of the inequality rows]:
    c = [eq(z); ineq(z) - s], s >= 0.  Distance rows further than `prune` from their obstacle at the guess are left out (and
    checked afterwards by `solve_ipm`).  Hessian of the Lagrangian: cost, ODE and tube rows analytically, distance rows by
    second-order central differences in the three pose variables they depend on -- nothing of the kernels' working sets,
    certificates, condensation or band ordering is used, and the linear algebra is scipy's SuperLU on the full KKT matrix."""

    def __init__(self, g: GeometricColloc, z_guess, prune=3.0):
        import scipy.sparse as sp_

        self.sp_, self.g = sp_, g
        P, _ = g.split(z_guess)
        sep = g.separations(P.reshape(-1, 7)[:, :3])
        self.keep = np.argwhere(sep < prune)                  # (point, obstacle) pairs with a distance row
        self.n_tube = 8 * g.n_chk
        self.mi = self.n_tube + len(self.keep)
        self.n0 = g.n
        self.n = self.n0 + self.mi
        self.me = len(g.eq(z_guess))
        self.m = self.me + self.mi
        lo = np.array([b[0] if b[0] is not None else -np.inf for b in g.bounds()], float)
        hi = np.array([b[1] if b[1] is not None else np.inf for b in g.bounds()], float)
        self.xl = np.concatenate([lo, np.zeros(self.mi)])
        self.xu = np.concatenate([hi, np.full(self.mi, np.inf)])

    def _ineq(self, z):
        P, _ = self.g.split(z)
        full = self.g.ineq(z)
        sep = full[self.n_tube:].reshape(self.g.np_, -1)
        return np.concatenate([full[: self.n_tube], sep[self.keep[:, 0], self.keep[:, 1]]])

    def initial(self, z_guess):
        return np.concatenate([z_guess, np.maximum(self._ineq(z_guess), 1e-2)])

    def f(self, X):
        return self.g.cost(X[: self.n0])

    def grad(self, X):
        return np.concatenate([self.g.cost_grad(X[: self.n0]), np.zeros(self.mi)])

    def cons(self, X):
        z = X[: self.n0]
        return np.concatenate([self.g.eq(z), self._ineq(z) - X[self.n0:]])

    def jac(self, X):
        sp_, g, z = self.sp_, self.g, X[: self.n0]
        Ji = g.ineq_jac(z)
        no = len(g.obs)
        rows = np.concatenate([np.arange(self.n_tube), self.n_tube + self.keep[:, 0] * no + self.keep[:, 1]])
        top = sp_.hstack([sp_.csr_matrix(g.eq_jac(z)), sp_.csr_matrix((self.me, self.mi))])
        bot = sp_.hstack([sp_.csr_matrix(Ji[rows]), -sp_.eye(self.mi)])
        return sp_.vstack([top, bot]).tocsr()

    def _cons_jac(self, X, want_jac=True):
        return self.cons(X), (self.jac(X) if want_jac else None)

    def hess_exact(self, X, nu):
        sp_, g = self.sp_, self.g
        z = X[: self.n0]
        P, dt = g.split(z)
        N, n0 = g.N, self.n0
        H = np.zeros((n0, n0))
        idt = n0 - 1
        # cost
        for i in range(N):
            for k in range(K_PTS):
                b = g.idx(i, k, 0); Bk = g.B[k]
                v, de, a, w = P[i, k, 3], P[i, k, 4], P[i, k, 5], P[i, k, 6]
                H[b + 3, b + 3] += dt * Bk * 2 * w * w; H[b + 6, b + 6] += dt * Bk * 2 * v * v
                H[b + 3, b + 6] += dt * Bk * 4 * v * w; H[b + 6, b + 3] += dt * Bk * 4 * v * w
                H[b + 4, b + 4] += dt * Bk * 2; H[b + 5, b + 5] += dt * Bk * 2
                for c, val in ((3, 2 * v * w * w), (4, 2 * de), (5, 2 * a), (6, 2 * v * v * w)):
                    H[b + c, idt] += Bk * val; H[idt, b + c] += Bk * val
        H[idt, idt] += 2.0 * N * N
        # ODE rows: order of g.eq: 7 initial rows, then [N, 6, 5] ODE rows
        nu_ode = nu[7: 7 + N * K_PTS * 5].reshape(N, K_PTS, 5)
        for i in range(N):
            for k in range(K_PTS):
                b = g.idx(i, k, 0)
                psi, v, de = P[i, k, 2], P[i, k, 3], P[i, k, 4]
                c, s, t = np.cos(psi), np.sin(psi), np.tan(de)
                n0_, n1_, n2_, n3_, n4_ = nu_ode[i, k]
                # -dt * d2 f
                H[b + 2, b + 2] += -dt * (n0_ * (-v * c) + n1_ * (-v * s))
                H[b + 2, b + 3] += -dt * (n0_ * (-s) + n1_ * c); H[b + 3, b + 2] += -dt * (n0_ * (-s) + n1_ * c)
                H[b + 3, b + 4] += -dt * n2_ * (1 + t * t) / g.wb; H[b + 4, b + 3] += -dt * n2_ * (1 + t * t) / g.wb
                H[b + 4, b + 4] += -dt * n2_ * v / g.wb * 2 * t * (1 + t * t)
                # cross with dt: -grad f
                for col, val in ((2, n0_ * (-v * s) + n1_ * (v * c)), (3, n0_ * c + n1_ * s + n2_ * t / g.wb),
                                 (4, n2_ * v * (1 + t * t) / g.wb), (5, n3_), (6, n4_)):
                    H[b + col, idt] += -val; H[idt, b + col] += -val
        # tube rows (b - shrink) - A front: d2/dpsi2 = wb A.(cos, sin)
        nu_in = nu[self.me:]
        for q in range(1, g.n_chk + 1):
            rows = nu_in[8 * (q - 1) + 4: 8 * (q - 1) + 8]
            Af = np.asarray(g.tube[q]["front"][0])
            if q < g.n_chk:
                pts = [(g.idx(q * g.Nps, 0, 2), 1.0)]
                psi = P[q * g.Nps, 0, 2]
            else:
                pts = [(g.idx(N - 1, j, 2), g.D[j]) for j in range(K_PTS) if g.D[j] != 0.0]
                psi = float(g.D @ P[-1][:, 2])
            d2 = float(rows @ (Af @ (g.wb * np.array([np.cos(psi), np.sin(psi)]))))
            for (ia, wa) in pts:
                for (ib, wb_) in pts:
                    H[ia, ib] += wa * wb_ * d2
        # distance rows: central second differences
        poses = P.reshape(-1, 7)[:, :3]
        h = 1e-4
        d0 = g.separations(poses)
        nu_d = np.zeros((g.np_, len(g.obs)))
        nu_d[self.keep[:, 0], self.keep[:, 1]] = nu_in[self.n_tube:]
        E = np.eye(3) * h
        for a in range(3):
            dp, dm = g.separations(poses + E[a]), g.separations(poses - E[a])
            haa = ((dp - 2 * d0 + dm) / (h * h) * nu_d).sum(1)
            for q in range(g.np_):
                H[7 * q + a, 7 * q + a] += haa[q]
            for b_ in range(a + 1, 3):
                hab = ((g.separations(poses + E[a] + E[b_]) - g.separations(poses + E[a] - E[b_])
                        - g.separations(poses - E[a] + E[b_]) + g.separations(poses - E[a] - E[b_])) / (4 * h * h) * nu_d).sum(1)
                for q in range(g.np_):
                    H[7 * q + a, 7 * q + b_] += hab[q]; H[7 * q + b_, 7 * q + a] += hab[q]
        return sp_.block_diag([sp_.csr_matrix(H), sp_.csr_matrix((self.mi, self.mi))]).tocsr()


def solve_ipm(g: GeometricColloc, guess, dt0, opt=None, prune=3.0):
    """oracle/ipm.py on the geometric statement.  Returns dict(traj [N, 6, 7], dt, cost, status, iters, eq, ineq)."""
    from . import ipm

    z0 = np.append(np.asarray(guess, float).ravel(), dt0)
    nlp = GeometricCollocIpm(g, z0, prune)
    opt = opt or ipm.IpmOptions(max_iter=500, hessian="exact", reg_dual=1e-9, stall_iters=0, err_stall_iters=0, tol=1e-8, constr_viol_tol=1e-9,
                                compl_inf_tol=1e-9, dual_inf_tol=1e-6, lower_mu_on_failure=True)
    r = ipm.solve(nlp, nlp.initial(z0), opt)
    z = r["X"][: g.n]
    P, dt = g.split(z)
    return dict(traj=P, dt=dt, cost=g.cost(z), status=int(r["status"]), iters=int(r["iters"]), eq=float(np.abs(g.eq(z)).max()),
                ineq=float(g.ineq(z).min()), rows=len(nlp.keep))
