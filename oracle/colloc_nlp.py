"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.  Parity unpinned (CasADi / IPOPT are not in this
image): pinned instead by the reference-form residuals below and by finite differences.

The single-vehicle collocation plan (reference confrez/control/vehicle.py:360-661) restated in numpy, values only:

* `CollocNlp.f`, `CollocNlp.cons`: objective and constraint VALUES of the certificate-eliminated statement, in the layout
  of conflict_rez_amd/csrc/cfz_colloc.inl (points x 7 | dt | collision slacks | tube slacks) -- what the kernel source is
  differentiated against in tests/test_colloc.py;
* `reference_residuals`: violations of the reference's OWN rows (ODE divided by dt :487-509, OBCA rows with the duals
  l, m :523-541, continuity :544-568, tube :570-617, terminal :619-626, boxes :439-478) of a finished plan, written as
  plain loops that mirror the reference line by line.
"""
import numpy as np

from .mpc_nlp import G_BODY, body_vertices, certificate_duals, polytope_vertices, rot, rows_for, select_rows

K_PTS = 6


def radau_tables(K=5):
    """tau = [0, Radau points]; A[j,k] = l_j'(tau_k), B[j] = int l_j, D[j] = l_j(1) (vehicle.py:54-97)."""
    from scipy.special import roots_jacobi

    x, _ = roots_jacobi(K - 1, 1.0, 0.0)
    tau = np.append(0.0, np.append((x + 1.0) / 2.0, 1.0))
    A, B, D = np.zeros((K + 1, K + 1)), np.zeros(K + 1), np.zeros(K + 1)
    for j in range(K + 1):
        others = np.delete(tau, j)
        p = np.poly1d(np.poly(others) / np.prod(tau[j] - others))
        D[j], A[j, :], B[j] = p(1.0), np.polyder(p)(tau), np.polyint(p)(1.0)
    return tau, A, B, D


def f_ct(p, wb):
    return np.array([p[3] * np.cos(p[2]), p[3] * np.sin(p[2]), p[3] / wb * np.tan(p[4]), p[5], p[6]])


class CollocNlp:
    def __init__(self, init_pose, tube, A_obs, b_obs, N_per_set=5, K=5, dmin=0.05, shrink_tube=0.5, final_heading=None,
                 wb=2.5, g=(3.3, 0.9, 0.6, 0.9), bounds=None):
        """tube: list over strategy steps of dict(front=(A, b), back=(A, b)) (rl_tube); A_obs [n_obs,4,2], b_obs [n_obs,4]."""
        assert K == 5
        self.S, self.Nps = len(tube), N_per_set
        self.N = N_per_set * (self.S - 1)
        self.n_chk = self.S - 1
        self.tube, self.init_pose, self.final_heading = tube, np.asarray(init_pose, float), final_heading
        self.A_obs, self.b_obs = np.asarray(A_obs, float), np.asarray(b_obs, float)
        self.n_obs = len(self.A_obs)
        self.PV, self.adj = zip(*(polytope_vertices(A, b) for A, b in zip(self.A_obs, self.b_obs))) if self.n_obs else ((), ())
        self.dmin, self.shrink, self.wb, self.g = dmin, shrink_tube, wb, np.asarray(g, float)
        self.BV = body_vertices(self.g)
        self.bounds = np.array([2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0]) if bounds is None else np.asarray(bounds, float)
        self.tau, self.A, self.B, self.D = radau_tables(K)
        np_, nr = self.N * K_PTS, 2 * self.n_obs
        self.np, self.nr = np_, nr
        self.iDt, self.sO = 7 * np_, 7 * np_ + 1
        self.sT = self.sO + np_ * nr
        self.n = self.sT + 8 * self.n_chk
        self.rO, self.rC = 7, 7 + 5 * np_
        self.rR = self.rC + 7 * (self.N - 1)
        self.rT = self.rR + np_ * nr
        self.rF = self.rT + 8 * self.n_chk
        self.rH = self.rF + 4
        self.m = self.rH + (1 if final_heading is not None else 0)

    def chk_point(self, q):
        return (q + 1) * self.Nps * K_PTS if q + 1 < self.n_chk else self.np - 1

    def select(self, X, prev=None):
        P = X[: self.iDt].reshape(self.np, 7)
        sel = np.zeros((self.np, self.n_obs), np.uint8)
        for q in range(self.np):
            for j in range(self.n_obs):
                sel[q, j] = select_rows(self.A_obs[j], self.b_obs[j], self.PV[j], P[q, :2], P[q, 2], self.g, self.BV,
                                        0 if prev is None else int(prev[q, j]))
        return sel

    def f(self, X):
        P, dt = X[: self.iDt].reshape(self.np, 7), X[self.iDt]
        e = P[:, 5] ** 2 + P[:, 3] ** 2 * P[:, 6] ** 2 + P[:, 4] ** 2
        return float(np.sum(np.tile(self.B, self.N) * e) * dt + (self.N * dt) ** 2)

    def cons(self, X, sel):
        P, dt = X[: self.iDt].reshape(self.np, 7), X[self.iDt]
        c = np.zeros(self.m)
        c[:3], c[3:7] = P[0, :3] - self.init_pose, P[0, 3:]
        for i in range(self.N):
            Z = P[i * K_PTS : (i + 1) * K_PTS]
            for k in range(K_PTS):
                q = i * K_PTS + k
                c[self.rO + 5 * q : self.rO + 5 * q + 5] = self.A[:, k] @ Z[:, :5] - dt * f_ct(Z[k], self.wb)
                for j in range(self.n_obs):
                    sep, _ = rows_for(self.A_obs[j], self.b_obs[j], self.PV[j], Z[k, :2], Z[k, 2], self.g, self.BV, int(sel[q, j]))
                    r = self.rR + q * self.nr + 2 * j
                    c[r : r + 2] = sep - self.dmin - X[self.sO + q * self.nr + 2 * j : self.sO + q * self.nr + 2 * j + 2]
            if i >= 1:
                c[self.rC + 7 * (i - 1) : self.rC + 7 * i] = Z[0] - P[i * K_PTS - 1]
        for t in range(self.n_chk):
            z = P[self.chk_point(t)]
            front = z[:2] + self.wb * np.array([np.cos(z[2]), np.sin(z[2])])
            (Ab, bb), (Af, bf) = self.tube[t + 1]["back"], self.tube[t + 1]["front"]
            r, s = self.rT + 8 * t, self.sT + 8 * t
            c[r : r + 4] = np.asarray(Ab) @ z[:2] - (np.asarray(bb) - self.shrink) + X[s : s + 4]
            c[r + 4 : r + 8] = np.asarray(Af) @ front - (np.asarray(bf) - self.shrink) + X[s + 4 : s + 8]
        c[self.rF : self.rF + 4] = P[-1, 3:]
        if self.final_heading is not None:
            c[self.rH] = P[-1, 2] - self.final_heading
        return c

    def pack(self, zu0, dt0):
        """zu0: arrays x, y, psi, v, delta, a, w of N*(K+1) values (interp_ws_for_collocation's output), dt0."""
        X = np.zeros(self.iDt + 1)
        for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w")):
            X[c : self.iDt : 7] = np.asarray(zu0[k], float).ravel()
        X[self.iDt] = dt0
        return X

    def unpack(self, X):
        P = X[: self.iDt].reshape(self.N, K_PTS, 7)
        sol = {k: P[:, :, c].copy() for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
        sol["dt"] = float(X[self.iDt])
        sel = self.select(X)
        l, m = np.zeros((self.N, K_PTS, 4 * self.n_obs)), np.zeros((self.N, K_PTS, 4 * self.n_obs))
        for i in range(self.N):
            for k in range(K_PTS):
                p = P[i, k]
                for j in range(self.n_obs):
                    c_ = int(sel[i * K_PTS + k, j])
                    sep, _ = rows_for(self.A_obs[j], self.b_obs[j], self.PV[j], p[:2], p[2], self.g, self.BV, c_)
                    v = (c_ >> 2) & 3 if sep[0] <= sep[1] else c_ & 3
                    l[i, k, 4 * j : 4 * j + 4], m[i, k, 4 * j : 4 * j + 4] = certificate_duals(self.A_obs[j], self.adj[j], p[2], (c_ >> 6, (c_ >> 4) & 3, v))
        sol["l"], sol["m"] = l, m
        return sol


def reference_residuals(nlp: CollocNlp, sol):
    """Largest violations of the reference's own rows by `sol` (x, y, psi, v, delta, a, w [N, K+1], dt, l, m
    [N, K+1, 4 n_obs]): dict(cost, eq, ineq, bound)."""
    N, K1, A, B, D, wb = nlp.N, K_PTS, nlp.A, nlp.B, nlp.D, nlp.wb
    x, y, psi, v, de, a, w = (np.asarray(sol[k], float) for k in ("x", "y", "psi", "v", "delta", "a", "w"))
    dt, l, m = sol["dt"], np.asarray(sol["l"], float), np.asarray(sol["m"], float)
    lo, hi = nlp.bounds[0::2], nlp.bounds[1::2]
    eq = max(abs(x[0, 0] - nlp.init_pose[0]), abs(y[0, 0] - nlp.init_pose[1]), abs(psi[0, 0] - nlp.init_pose[2]),
             abs(v[0, 0]), abs(de[0, 0]), abs(a[0, 0]), abs(w[0, 0]))  # :426-436
    ineq, bnd, cost = 0.0, max((-l).max(initial=0.0), (-m).max(initial=0.0)), 0.0
    Z = np.stack([x, y, psi, v, de], -1)
    U = np.stack([a, w], -1)
    for i in range(N):
        for k in range(K1):
            for val, j in ((x[i, k], 0), (y[i, k], 1), (v[i, k], 2), (de[i, k], 3), (a[i, k], 4), (w[i, k], 5)):  # :439-478
                bnd = max(bnd, lo[j] - val, val - hi[j])
            poly_ode = sum(A[j, k] * Z[i, j] / dt for j in range(K1))  # :487-509
            eq = max(eq, np.abs(poly_ode - f_ct(np.append(Z[i, k], U[i, k]), wb)).max())
            cost += B[k] * (a[i, k] ** 2 + v[i, k] ** 2 * w[i, k] ** 2 + de[i, k] ** 2) * dt  # :512-521
            t, R = np.array([x[i, k], y[i, k]]), rot(psi[i, k])
            for j in range(nlp.n_obs):  # :523-541
                Ao, bo = nlp.A_obs[j], nlp.b_obs[j]
                lj, mj = l[i, k, 4 * j : 4 * j + 4], m[i, k, 4 * j : 4 * j + 4]
                ineq = max(ineq, nlp.dmin - (np.dot(-nlp.g, mj) + np.dot(Ao @ t - bo, lj)))
                eq = max(eq, np.abs(G_BODY.T @ mj + R.T @ Ao.T @ lj).max(), abs(np.dot(Ao.T @ lj, Ao.T @ lj) - 1.0))
        if i >= 1:  # :544-568
            eq = max(eq, np.abs(D @ Z[i - 1] - Z[i, 0]).max(), np.abs(D @ U[i - 1] - U[i, 0]).max())
            q, r = divmod(i, nlp.Nps)
            if r == 0:  # :570-588
                front = Z[i, 0, :2] + wb * np.array([np.cos(psi[i, 0]), np.sin(psi[i, 0])])
                (Ab, bb), (Af, bf) = nlp.tube[q]["back"], nlp.tube[q]["front"]
                ineq = max(ineq, (np.asarray(Ab) @ Z[i, 0, :2] - (np.asarray(bb) - nlp.shrink)).max(), (np.asarray(Af) @ front - (np.asarray(bf) - nlp.shrink)).max())
    zF, uF = D @ Z[N - 1], D @ U[N - 1]  # :590-604
    front = zF[:2] + wb * np.array([np.cos(zF[2]), np.sin(zF[2])])
    (Ab, bb), (Af, bf) = nlp.tube[-1]["back"], nlp.tube[-1]["front"]
    ineq = max(ineq, (np.asarray(Ab) @ zF[:2] - (np.asarray(bb) - nlp.shrink)).max(), (np.asarray(Af) @ front - (np.asarray(bf) - nlp.shrink)).max())
    eq = max(eq, abs(zF[3]), abs(zF[4]), abs(uF[0]), abs(uF[1]))  # :622-626
    if nlp.final_heading is not None:
        eq = max(eq, abs(zF[2] - nlp.final_heading))
    cost += (N * dt) ** 2  # :638
    return dict(cost=float(cost), eq=float(eq), ineq=float(max(ineq, 0.0)), bound=float(max(bnd, 0.0)))
