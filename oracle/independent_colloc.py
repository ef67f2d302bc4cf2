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
Solver: scipy's SLSQP with analytic Jacobians (distances: central differences in the pose they depend on).
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
