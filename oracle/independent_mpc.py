"""Oracle (test infrastructure, PARITY UNPINNED like the rest of oracle/): the MPC-step NLP of
`VehicleFollower.setup_controller` (reference confrez/control/vehicle_follower.py:146-368) solved INDEPENDENTLY of the
engine's formulation and algorithm, at the reference's full size (N = 30, six obstacles, three neighbours).

Formulation: the OBCA rows of the reference (:280-290 per obstacle, :322-352 per neighbour) say "there are duals that certify a
separation of at least dmin"; by strong duality of the distance problem (Boyd & Vandenberghe 8.2, cited at
multi_vehicle_planner.py:446; Zhang, Liniger, Borrelli, OBCA, Prop. 1) that holds exactly when the Euclidean distance of the
two polygons is at least dmin.  So the reference NLP has the same optimal trajectories as

    min  sum_k 100 (x-xr)^2 + 100 (y-yr)^2 + 100 (psi-psir)^2 + a^2 + v^2 w^2 + delta^2        (:263-272)
    s.t. z_0 = current state (:194-199);  z_{k+1} = RK4(z_k, u_k) (:243-260);  boxes on x, y, v, delta, a, w (:204-240);
         dist(body(z_k), obstacle_j) >= dmin,  dist(body(z_k), body(neighbour_o at stage k)) >= dmin    for all k, j, o

in the 7 N primal variables alone.  No duals, no working sets, no face-normal restriction (vertex-vertex closest pairs
count with their true distance), no slacks.  Solver: scipy's SLSQP (a dense SQP; nothing in common with oracle/ipm.py or
the kernel).  Distances: closest vertex-edge pair over both polygons, vectorised; signed by the face-normal depth when the
polygons overlap.  Jacobians: dynamics by forward sensitivities (oracle/dynamics.py), distances by central differences in
the three pose variables they depend on.
"""
import numpy as np
from scipy.optimize import minimize

from .dynamics import bicycle_rk4_jac
from .mpc_nlp import MpcSpec, polytope_vertices

KEYS = ("x", "y", "psi", "v", "delta", "a", "w")


def _body_vertices(x, y, psi, g):
    """[..., 4, 2] world vertices of the body rectangle at poses (x, y, psi), counter-clockwise."""
    B = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])
    c, s = np.cos(psi)[..., None], np.sin(psi)[..., None]
    return np.stack([x[..., None] + c * B[:, 0] - s * B[:, 1], y[..., None] + s * B[:, 0] + c * B[:, 1]], -1)


def _point_segments(Q, P):
    """Squared distances of the vertices Q[..., 4, 2] to the edges of the polygons P[..., 4, 2] -> [..., 4, 4]."""
    a, b = P[..., None, :, :], np.roll(P, -1, axis=-2)[..., None, :, :]  # [..., 1, 4 edges, 2]
    q = Q[..., :, None, :]                                               # [..., 4 verts, 1, 2]
    e, w = b - a, q - a
    t = np.clip((w * e).sum(-1) / (e * e).sum(-1), 0.0, 1.0)
    r = w - t[..., None] * e
    return (r * r).sum(-1)


def polygon_distance_batch(P, Q):
    """Signed distance of convex quadrilaterals P, Q [..., 4, 2] (both counter-clockwise): the Euclidean distance when they are
    apart, minus the smallest face-normal overlap depth when they intersect."""
    d2 = np.minimum(_point_segments(Q, P).min((-1, -2)), _point_segments(P, Q).min((-1, -2)))
    dist = np.sqrt(d2)

    def depth(A, B):  # max over faces of A of (min over vertices of B of the outward distance)
        e = np.roll(A, -1, axis=-2) - A
        n = np.stack([e[..., 1], -e[..., 0]], -1)
        n = n / np.linalg.norm(n, axis=-1, keepdims=True)  # outward normals of a counter-clockwise polygon
        off = (n * A).sum(-1)
        return ((B[..., None, :, :] * n[..., :, None, :]).sum(-1) - off[..., :, None]).min(-1).max(-1)

    sep = np.maximum(depth(P, Q), depth(Q, P))
    return np.where(sep > 0.0, dist, sep)


class GeometricMpc:
    def __init__(self, spec: MpcSpec, x0, ref, nbr):
        self.spec, self.x0, self.ref, self.nbr = spec, np.asarray(x0, float), np.asarray(ref, float), np.asarray(nbr, float)
        self.N, self.no, self.nn = spec.N, spec.n_obs, spec.n_nbr
        self.n = 7 * self.N
        self.obs = np.stack([polytope_vertices(spec.A_obs[j], spec.b_obs[j])[0] for j in range(self.no)]) if self.no else np.zeros((0, 4, 2))
        if self.nn:
            self.nbv = _body_vertices(self.nbr[:, 0], self.nbr[:, 1], self.nbr[:, 2], spec.g)  # [nn, N, 4, 2]

    def split(self, X):
        return np.asarray(X, float).reshape(self.N, 7)

    def cost(self, X):
        P, w, r = self.split(X), self.spec.weights, self.ref
        return float(np.sum(w[0] * (P[:, 0] - r[0]) ** 2 + w[1] * (P[:, 1] - r[1]) ** 2 + w[2] * (P[:, 2] - r[2]) ** 2
                            + w[3] * P[:, 5] ** 2 + w[4] * P[:, 3] ** 2 * P[:, 6] ** 2 + w[5] * P[:, 4] ** 2))

    def cost_grad(self, X):
        P, w, r = self.split(X), self.spec.weights, self.ref
        g = np.zeros_like(P)
        g[:, 0] = 2 * w[0] * (P[:, 0] - r[0]); g[:, 1] = 2 * w[1] * (P[:, 1] - r[1]); g[:, 2] = 2 * w[2] * (P[:, 2] - r[2])
        g[:, 3] = 2 * w[4] * P[:, 3] * P[:, 6] ** 2; g[:, 4] = 2 * w[5] * P[:, 4]
        g[:, 5] = 2 * w[3] * P[:, 5]; g[:, 6] = 2 * w[4] * P[:, 3] ** 2 * P[:, 6]
        return g.ravel()

    def eq(self, X):
        P, sp = self.split(X), self.spec
        F, _, _ = bicycle_rk4_jac(P[:-1, :5], P[:-1, 5:], sp.dt, sp.wb, sp.rk_substeps)
        return np.concatenate([P[0, :5] - self.x0, (P[1:, :5] - F).ravel()])

    def eq_jac(self, X):
        P, sp, N = self.split(X), self.spec, self.N
        _, Fz, Fu = bicycle_rk4_jac(P[:-1, :5], P[:-1, 5:], sp.dt, sp.wb, sp.rk_substeps)
        J = np.zeros((5 * N, 7 * N))
        J[:5, :5] = np.eye(5)
        for k in range(N - 1):
            r = 5 + 5 * k
            J[r:r + 5, 7 * k:7 * k + 5] = -Fz[k]
            J[r:r + 5, 7 * k + 5:7 * k + 7] = -Fu[k]
            J[r:r + 5, 7 * (k + 1):7 * (k + 1) + 5] = np.eye(5)
        return J

    def separations(self, poses):
        """[N, n_obs + n_nbr] signed distances of the body at poses [N, 3]."""
        W = _body_vertices(poses[:, 0], poses[:, 1], poses[:, 2], self.spec.g)  # [N, 4, 2]
        out = []
        if self.no:
            out.append(polygon_distance_batch(self.obs[None, :, :, :], W[:, None, :, :]))
        if self.nn:
            out.append(polygon_distance_batch(np.moveaxis(self.nbv, 0, 1), W[:, None, :, :]))
        return np.concatenate(out, 1)

    def ineq(self, X):
        return (self.separations(self.split(X)[:, :3]) - self.spec.dmin).ravel()

    def ineq_jac(self, X, h=1e-6):
        P, N, nb = self.split(X), self.N, self.no + self.nn
        J = np.zeros((N * nb, 7 * N))
        for c in range(3):  # a stage's rows depend on that stage's pose only: one perturbation per pose variable serves all stages
            e = np.zeros(3); e[c] = h
            d = (self.separations(P[:, :3] + e) - self.separations(P[:, :3] - e)) / (2 * h)
            for k in range(N):
                J[k * nb:(k + 1) * nb, 7 * k + c] = d[k]
        return J

    def bounds(self):
        lo, hi = np.full((self.N, 7), -np.inf), np.full((self.N, 7), np.inf)
        bd = self.spec.bounds
        for c, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
            lo[:, c], hi[:, c] = bd[2 * j], bd[2 * j + 1]
        return list(zip(lo.ravel(), hi.ravel()))


def solve(spec: MpcSpec, x0, ref, nbr, zu, maxiter=400, ftol=1e-12):
    """SLSQP from the warm start zu [7, N].  Returns dict(zu [7, N], cost, status, iters, eq, ineq)."""
    nlp = GeometricMpc(spec, x0, ref, nbr)
    X0 = np.asarray(zu, float).T.ravel()
    out = minimize(nlp.cost, X0, jac=nlp.cost_grad, method="SLSQP", bounds=nlp.bounds(),
                   constraints=[dict(type="eq", fun=nlp.eq, jac=nlp.eq_jac), dict(type="ineq", fun=nlp.ineq, jac=nlp.ineq_jac)],
                   options=dict(maxiter=maxiter, ftol=ftol))
    return dict(zu=nlp.split(out.x).T.copy(), cost=float(out.fun), status=int(out.status), iters=int(out.nit),
                eq=float(np.abs(nlp.eq(out.x)).max()), ineq=float(nlp.ineq(out.x).min()))
