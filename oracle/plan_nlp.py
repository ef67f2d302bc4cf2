"""Oracle (test infrastructure): numpy statements of the single-vehicle planning NLPs.

`StateWsNlp` restates `Vehicle.state_ws` (reference `confrez/control/vehicle.py:99-231`):
forward-Euler kinematic bicycle over T = N*(S-1) steps (:173), initial pose fixed and
v0 = delta0 = a0 = w0 = 0 (:131-138), boxes on x,y,v,delta for k < T (:141-153; a,w only
if `bounded_input` :155-167), at k = N*i the rear-axle point inside the back cell and the
front point (x + wb cos psi, y + wb sin psi) inside the front cell, each shrunk by
`shrink_tube` (:178-192), optional terminal heading (:194-195), cost sum a^2 + w^2 (:175-176).

Same IPOPT form as `MpcNlp` (min f s.t. c = 0, bounds): each tube row `A p <= b - shrink`
gets a slack sigma >= 0.
"""
import numpy as np
import scipy.sparse as sp

from .dynamics import bicycle_ct, bicycle_ct_jac


def speed_guess(p, dt):
    """Signed speed along a guessed path p [T+1, >=3] (x, y, psi), as `cfz_state_ws` adds it to the spline guess:
    v_k = sign((p_{k+1} - p_k) . heading_k) |p_{k+1} - p_k| / dt for 0 < k < T, v_0 = v_T = 0."""
    p = np.asarray(p, float)
    d = np.diff(p[:, :2], axis=0)
    along = d[:, 0] * np.cos(p[:-1, 2]) + d[:, 1] * np.sin(p[:-1, 2])
    v = np.append(np.sign(along) * np.hypot(d[:, 0], d[:, 1]) / dt, 0.0)
    v[0] = 0.0
    return v


class StateWsNlp:
    """X layout: [z_0 u_0 | z_1 u_1 | ... | z_{T-1} u_{T-1} | z_T | tube slacks (8 per checkpoint)]."""

    def __init__(self, init_pose, tube, N=30, dt=0.1, wb=2.5, final_heading=None, shrink_tube=0.8,
                 bounded_input=False, bounds=None):
        """tube: list over strategy steps of dict(front=(A[4,2], b[4]), back=(A, b))."""
        self.N, self.dt, self.wb = N, dt, wb
        self.S = len(tube)
        self.T = T = N * (self.S - 1)
        self.init_pose = np.asarray(init_pose, float)
        self.final_heading = final_heading
        self.shrink = shrink_tube
        self.tube = tube
        self.n_chk = self.S - 1
        self.n = 7 * T + 5 + 8 * self.n_chk
        self.s0 = 7 * T + 5
        self.m = 7 + 5 * T + 8 * self.n_chk + (1 if final_heading is not None else 0)
        b = np.array([2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0]) if bounds is None else np.asarray(bounds, float)
        self.bounds = b
        xl = np.full(self.n, -np.inf)
        xu = np.full(self.n, np.inf)
        k7 = 7 * np.arange(T)
        for col, j in ((0, 0), (1, 1), (3, 2), (4, 3)) + (((5, 4), (6, 5)) if bounded_input else ()):
            xl[k7 + col], xu[k7 + col] = b[2 * j], b[2 * j + 1]
        xl[self.s0 :] = 0.0
        self.xl, self.xu = xl, xu
        self.block_mask = np.zeros(self.n)
        # IPOPT's delta_c (IpmOptions.reg_dual) on the terminal-heading row alone: the initial rows, the Euler rows (-I on z_{k+1}) and the
        # tube rows (+I on their slacks) have full row rank at any iterate; the heading row is the one that loses it (with v = delta = 0
        # the headings are pinned by the initial pose).  This keeps the Newton system a stage recursion: cfz_plan.inl solves it by a Riccati
        # sweep with that one row bordered.
        self.dual_reg_rows = np.zeros(self.m)
        if final_heading is not None:
            self.dual_reg_rows[-1] = 1.0

    def zidx(self, k):
        return 7 * k

    def pack(self, x, y, psi, v=None, delta=None, a=None, w=None):
        T = self.T
        X = np.zeros(self.n)
        for col, arr in enumerate((x, y, psi, v, delta)):
            if arr is not None:
                X[7 * np.arange(T) + col] = arr[:T]
                X[7 * T + col] = arr[T]
        for col, arr in ((5, a), (6, w)):
            if arr is not None:
                X[7 * np.arange(T) + col] = arr[:T]
        self.set_slacks(X)
        return X

    def set_slacks(self, X):
        X[self.s0 :] = 0.0
        c = self.cons(X)
        r0 = 7 + 5 * self.T
        X[self.s0 :] = -c[r0 : r0 + 8 * self.n_chk]

    def unpack(self, X):
        T = self.T
        Z = np.concatenate([X[: 7 * T].reshape(T, 7)[:, :5], X[7 * T : 7 * T + 5][None]], 0)
        U = X[: 7 * T].reshape(T, 7)[:, 5:7]
        return dict(t=np.linspace(0, T * self.dt, T + 1), x=Z[:, 0], y=Z[:, 1], psi=Z[:, 2], v=Z[:, 3], delta=Z[:, 4],
                    a=np.append(U[:, 0], U[-1, 0]), w=np.append(U[:, 1], U[-1, 1]))

    def f(self, X):
        U = X[: 7 * self.T].reshape(self.T, 7)[:, 5:7]
        return float(np.sum(U * U))

    def grad(self, X):
        g = np.zeros(self.n)
        V = g[: 7 * self.T].reshape(self.T, 7)
        V[:, 5:7] = 2 * X[: 7 * self.T].reshape(self.T, 7)[:, 5:7]
        return g

    def hess_gn(self, X):
        d = np.zeros(self.n)
        d[: 7 * self.T].reshape(self.T, 7)[:, 5:7] = 2.0
        return sp.diags(d).tocsr()

    def hess_exact(self, X, nu):
        """Hessian of the Lagrangian f + nu^T c (what CasADi hands IPOPT with `expand`)."""
        T, dt, wb = self.T, self.dt, self.wb
        ZU = X[: 7 * T].reshape(T, 7)
        psi, v, de = ZU[:, 2], ZU[:, 3], ZU[:, 4]
        lam = nu[7 : 7 + 5 * T].reshape(T, 5) * dt  # multipliers of the x,y,psi rows matter
        c, s, tn = np.cos(psi), np.sin(psi), np.tan(de)
        sec2 = 1.0 + tn * tn
        kk = 7 * np.arange(T)
        rows, cols, vals = [], [], []

        def add(i, j, val):
            rows.append(kk + i), cols.append(kk + j), vals.append(val)
            if i != j:
                rows.append(kk + j), cols.append(kk + i), vals.append(val)

        add(2, 2, lam[:, 0] * (-v * c) + lam[:, 1] * (-v * s))
        add(2, 3, lam[:, 0] * (-s) + lam[:, 1] * c)
        add(3, 4, lam[:, 2] * sec2 / wb)
        add(4, 4, lam[:, 2] * 2.0 * v * tn * sec2 / wb)
        d = np.zeros(self.n)
        d[: 7 * T].reshape(T, 7)[:, 5:7] = 2.0
        r0 = 7 + 5 * T
        for i in range(1, self.S):
            base = 7 * self.N * i
            p_ = X[base + 2]
            Af = self.tube[i]["front"][0]
            nf = nu[r0 + 8 * (i - 1) + 4 : r0 + 8 * (i - 1) + 8]
            d[base + 2] += float(nf @ (wb * (-Af[:, 0] * np.cos(p_) - Af[:, 1] * np.sin(p_))))
        H = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n, self.n))
        return H + sp.diags(d)

    def cons(self, X):
        return self._cons_jac(X, False)[0]

    def jac(self, X):
        return self._cons_jac(X, True)[1]

    def _cons_jac(self, X, want_jac=True):
        T, dt, wb = self.T, self.dt, self.wb
        ZU = X[: 7 * T].reshape(T, 7)
        Z, U = ZU[:, :5], ZU[:, 5:7]
        Zn = np.concatenate([Z[1:], X[7 * T : 7 * T + 5][None]], 0)
        c = np.zeros(self.m)
        rows, cols, vals = [], [], []

        def add(r, ci, v):
            if want_jac:
                r = np.asarray(r).ravel()
                rows.append(r), cols.append(np.asarray(ci).ravel())
                vals.append(np.broadcast_to(v, r.shape).astype(float).ravel())

        # initial conditions: pose, v, delta, a0, w0
        c[0:3] = Z[0, 0:3] - self.init_pose
        c[3:7] = ZU[0, 3:7]
        add(np.arange(7), np.arange(7), 1.0)
        # Euler dynamics z_k + dt f(z_k,u_k) - z_{k+1} = 0
        if want_jac:
            f, fz, fu = bicycle_ct_jac(Z, U, wb)
        else:
            f = bicycle_ct(Z, U, wb)
        c[7 : 7 + 5 * T] = (Z + dt * f - Zn).ravel()
        if want_jac:
            kk = np.arange(T)
            for i in range(5):
                r = 7 + 5 * kk + i
                add(r, 7 * kk + i, 1.0)
                for j in range(5):
                    nz = np.abs(fz[:, i, j]).max() > 0
                    if nz:
                        add(r, 7 * kk + j, dt * fz[:, i, j])
                for j in range(2):
                    if np.abs(fu[:, i, j]).max() > 0:
                        add(r, 7 * kk + 5 + j, dt * fu[:, i, j])
                nxt = np.where(kk + 1 < T, 7 * (kk + 1) + i, 7 * T + i)
                add(r, nxt, -1.0)
        # tube rows  A p - (b - shrink) + sigma = 0
        r0 = 7 + 5 * T
        for i in range(1, self.S):
            k = self.N * i
            base = 7 * k  # works for k == T too (z_T sits at 7T)
            x, y, psi = X[base], X[base + 1], X[base + 2]
            Ab, bb = self.tube[i]["back"]
            Af, bf = self.tube[i]["front"]
            fx, fy = x + wb * np.cos(psi), y + wb * np.sin(psi)
            rr = r0 + 8 * (i - 1)
            sl = self.s0 + 8 * (i - 1)
            c[rr : rr + 4] = Ab @ np.array([x, y]) - (bb - self.shrink) + X[sl : sl + 4]
            c[rr + 4 : rr + 8] = Af @ np.array([fx, fy]) - (bf - self.shrink) + X[sl + 4 : sl + 8]
            if want_jac:
                for q in range(4):
                    add([rr + q], [base], Ab[q, 0]), add([rr + q], [base + 1], Ab[q, 1])
                    add([rr + q], [sl + q], 1.0)
                    add([rr + 4 + q], [base], Af[q, 0]), add([rr + 4 + q], [base + 1], Af[q, 1])
                    add([rr + 4 + q], [base + 2], wb * (-Af[q, 0] * np.sin(psi) + Af[q, 1] * np.cos(psi)))
                    add([rr + 4 + q], [sl + 4 + q], 1.0)
        if self.final_heading is not None:
            c[-1] = X[7 * T + 2] - self.final_heading
            add([self.m - 1], [7 * T + 2], 1.0)
        J = None
        if want_jac:
            J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.m, self.n))
        return c, J
