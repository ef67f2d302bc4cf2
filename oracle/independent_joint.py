"""Oracle (test infrastructure, PARITY UNPINNED like the rest of oracle/): the JOINT collocation plan of
`MultiVehiclePlanner.solve_final_problem_obca` (reference confrez/control/multi_vehicle_planner.py:343-480) solved
INDEPENDENTLY of the planning kernel's formulation and algorithm, at the reference's size.

Statement (same step as oracle/independent_colloc.py): every vehicle keeps the rows of its own plan
(`setup_single_final_problem(opti=, dt=)`, vehicle.py:360-661) on ONE shared interval length dt; the vehicle-vehicle OBCA rows
(:423-456, "there are duals certifying a separation >= dmin" between the two bodies at the same collocation point, for the
first min(N_a, N_b) intervals) are replaced by what they are equivalent to, dist(body_a(pose_a), body_b(pose_b)) >= dmin.
Primal variables only: the points of all vehicles and dt.  cost = sum over vehicles of [sum B_k (a^2 + v^2 w^2 + delta^2) dt +
(N_a dt)^2] (:458-470).  No duals, no working sets, no condensation, no bordering, no band ordering; derivatives of the pair
distances by central differences in the six pose variables they depend on; oracle/ipm.py (full KKT, SuperLU) in slack form.
"""
import numpy as np

from .independent_colloc import GeometricColloc, GeometricCollocIpm
from .independent_mpc import _body_vertices, polygon_distance_batch
from .colloc_nlp import K_PTS


class GeometricJointIpm:
    """min f(X) s.t. c(X) = 0, XL <= X <= XU with X = [points of vehicle 0 | points of vehicle 1 | .. | dt | slacks];
    c = [eq_0; eq_1; ..; (ineq_a - s) for every vehicle; (pair distances - dmin - s)].  Rows of distances larger than `prune` at
    the guess are left out (and checked afterwards by `solve_joint_ipm`)."""

    def __init__(self, gs, pairs, z_guess, prune=3.0):
        import scipy.sparse as sp_

        self.sp_, self.gs, self.pairs = sp_, gs, list(pairs)
        self.V = len(gs)
        self.off = np.concatenate([[0], np.cumsum([7 * g.np_ for g in gs])]).astype(int)
        self.n0 = int(self.off[-1]) + 1  # + dt
        self.idt = self.n0 - 1
        self.dmin = gs[0].dmin
        self.sub = [GeometricCollocIpm(g, self.z_of(z_guess, a), prune) for a, g in enumerate(gs)]  # per-vehicle rows and Hessians
        # pair rows kept: (pair, point) with a distance below prune at the guess
        self.pk = []
        for e, (a, b) in enumerate(self.pairs):
            d = self.pair_dist(z_guess, a, b)
            self.pk.append(np.flatnonzero(d < prune))
        self.me_a = [s.me for s in self.sub]
        self.mi_a = [s.mi for s in self.sub]
        self.me, self.mi = sum(self.me_a), sum(self.mi_a) + sum(len(k) for k in self.pk)
        self.n, self.m = self.n0 + self.mi, self.me + self.mi
        lo, hi = [], []
        for g in gs:
            bd = g.bounds()[:-1]
            lo += [b[0] if b[0] is not None else -np.inf for b in bd]; hi += [b[1] if b[1] is not None else np.inf for b in bd]
        self.xl = np.concatenate([np.array(lo, float), [1e-3], np.zeros(self.mi)])
        self.xu = np.concatenate([np.array(hi, float), [np.inf], np.full(self.mi, np.inf)])

    # ---- layout -----------------------------------------------------------------------------------------------------
    def z_of(self, z, a):
        """vehicle a's own vector [points | dt] out of the joint one"""
        return np.append(z[self.off[a]: self.off[a + 1]], z[self.idt])

    def cols(self, a):
        """joint columns of vehicle a's own vector"""
        return np.append(np.arange(self.off[a], self.off[a + 1]), self.idt)

    def poses(self, z, a, npts):
        return z[self.off[a]: self.off[a + 1]].reshape(-1, 7)[:npts, :3]

    def pair_npts(self, a, b):
        return min(self.gs[a].N, self.gs[b].N) * K_PTS

    def pair_dist(self, z, a, b, da=None, db=None):
        n = self.pair_npts(a, b)
        pa, pb = self.poses(z, a, n), self.poses(z, b, n)
        if da is not None:
            pa = pa + da
        if db is not None:
            pb = pb + db
        g = self.gs[0].g
        Wa, Wb = _body_vertices(pa[:, 0], pa[:, 1], pa[:, 2], g), _body_vertices(pb[:, 0], pb[:, 1], pb[:, 2], g)
        return polygon_distance_batch(Wa[:, None], Wb[:, None])[:, 0]

    # ---- functions --------------------------------------------------------------------------------------------------
    def initial(self, z_guess):
        return np.concatenate([z_guess, np.maximum(self._ineq(z_guess), 1e-2)])

    def f(self, X):
        return float(sum(g.cost(self.z_of(X, a)) for a, g in enumerate(self.gs)))

    def grad(self, X):
        out = np.zeros(self.n)
        for a, g in enumerate(self.gs):
            out[self.cols(a)] += g.cost_grad(self.z_of(X, a))
        return out

    def _ineq(self, z):
        parts = [s._ineq(self.z_of(z, a)) for a, s in enumerate(self.sub)]
        for e, (a, b) in enumerate(self.pairs):
            parts.append(self.pair_dist(z, a, b)[self.pk[e]] - self.dmin)
        return np.concatenate(parts)

    def cons(self, X):
        z = X[: self.n0]
        return np.concatenate([g.eq(self.z_of(z, a)) for a, g in enumerate(self.gs)] + [self._ineq(z) - X[self.n0:]])

    def jac(self, X, h=1e-6):
        sp_, z = self.sp_, X[: self.n0]
        J = sp_.lil_matrix((self.m, self.n))
        r = 0
        for a, g in enumerate(self.gs):
            Ja = g.eq_jac(self.z_of(z, a))
            J[r: r + Ja.shape[0], self.cols(a)] = Ja
            r += Ja.shape[0]
        for a, s in enumerate(self.sub):
            g = s.g
            Ji = g.ineq_jac(self.z_of(z, a))
            no = len(g.obs)
            rows = np.concatenate([np.arange(s.n_tube), s.n_tube + s.keep[:, 0] * no + s.keep[:, 1]])
            J[r: r + len(rows), self.cols(a)] = Ji[rows]
            r += len(rows)
        for e, (a, b) in enumerate(self.pairs):
            k = self.pk[e]
            for c in range(3):
                dv = np.zeros(3); dv[c] = h
                ga = (self.pair_dist(z, a, b, da=dv) - self.pair_dist(z, a, b, da=-dv)) / (2 * h)
                gb = (self.pair_dist(z, a, b, db=dv) - self.pair_dist(z, a, b, db=-dv)) / (2 * h)
                for t, q in enumerate(k):
                    J[r + t, self.off[a] + 7 * q + c] = ga[q]
                    J[r + t, self.off[b] + 7 * q + c] = gb[q]
            r += len(k)
        J = J.tocsr()
        return sp_.hstack([J[:, : self.n0], sp_.vstack([sp_.csr_matrix((self.me, self.mi)), -sp_.eye(self.mi)])]).tocsr()

    def _cons_jac(self, X, want_jac=True):
        return self.cons(X), (self.jac(X) if want_jac else None)

    def hess_exact(self, X, nu, h=1e-4):
        sp_, z = self.sp_, X[: self.n0]
        H = np.zeros((self.n0, self.n0))
        re, ri = 0, self.me
        for a, s in enumerate(self.sub):  # the vehicle's own cost and rows, by the single-plan oracle
            nu_a = np.concatenate([nu[re: re + s.me], nu[ri: ri + s.mi]])
            Xa = np.concatenate([self.z_of(z, a), np.zeros(s.mi)])
            Ha = s.hess_exact(Xa, nu_a)[: s.n0, : s.n0].toarray()
            c = self.cols(a)
            H[np.ix_(c, c)] += Ha
            re += s.me; ri += s.mi
        for e, (a, b) in enumerate(self.pairs):  # pair distances: second central differences in (pose_a, pose_b)
            k = self.pk[e]
            nu_p = np.zeros(self.pair_npts(a, b)); nu_p[k] = nu[ri: ri + len(k)]
            ri += len(k)
            d0 = self.pair_dist(z, a, b)

            def dist(u, su, w=None, sw=0.0):
                dv = [np.zeros(3), np.zeros(3)]
                dv[u // 3][u % 3] += su * h
                if w is not None:
                    dv[w // 3][w % 3] += sw * h
                return self.pair_dist(z, a, b, da=dv[0], db=dv[1])

            base = (self.off[a], self.off[b])
            for u in range(6):
                huu = (dist(u, 1.0) - 2 * d0 + dist(u, -1.0)) / (h * h) * nu_p
                iu = base[u // 3] + 7 * np.arange(len(nu_p)) + u % 3
                H[iu, iu] += huu
                for w in range(u + 1, 6):
                    huw = (dist(u, 1.0, w, 1.0) - dist(u, 1.0, w, -1.0) - dist(u, -1.0, w, 1.0) + dist(u, -1.0, w, -1.0)) / (4 * h * h) * nu_p
                    iw = base[w // 3] + 7 * np.arange(len(nu_p)) + w % 3
                    H[iu, iw] += huw; H[iw, iu] += huw
        return sp_.block_diag([sp_.csr_matrix(H), sp_.csr_matrix((self.mi, self.mi))]).tocsr()


def solve_joint_ipm(gs, pairs, guesses, dt0, opt=None, prune=3.0):
    """gs: one GeometricColloc per vehicle; guesses: per vehicle [N_a * 6, 7]; dt0: shared.  Returns dict(trajs, dt, cost,
    status, iters, eq, ineq (smallest slack of any inequality row, pruned ones included), pair (smallest pair distance))."""
    from . import ipm

    z0 = np.concatenate([np.asarray(g_, float).ravel() for g_ in guesses] + [[dt0]])
    nlp = GeometricJointIpm(gs, pairs, z0, prune)
    opt = opt or ipm.IpmOptions(max_iter=1000, hessian="exact", reg_dual=1e-9, stall_iters=0, err_stall_iters=0, tol=1e-8, constr_viol_tol=1e-9,
                                compl_inf_tol=1e-9, dual_inf_tol=1e-6, lower_mu_on_failure=True)
    r = ipm.solve(nlp, nlp.initial(z0), opt)
    z = r["X"][: nlp.n0]
    eq = max(float(np.abs(g.eq(nlp.z_of(z, a))).max()) for a, g in enumerate(gs))
    ineq = min(float(g.ineq(nlp.z_of(z, a)).min()) for a, g in enumerate(gs))
    pair = min(float(nlp.pair_dist(z, a, b).min()) for a, b in nlp.pairs)
    return dict(trajs=[z[nlp.off[a]: nlp.off[a + 1]].reshape(g.N, K_PTS, 7) for a, g in enumerate(gs)], dt=float(z[nlp.idt]),
                cost=nlp.f(z), status=int(r["status"]), iters=int(r["iters"]), eq=eq, ineq=min(ineq, pair - nlp.dmin), pair=pair,
                rows=int(nlp.mi))
