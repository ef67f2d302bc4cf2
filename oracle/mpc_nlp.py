"""Oracle (test infrastructure): numpy statement of the distributed-MPC step NLP.

Restates `VehicleFollower.setup_controller` (reference
`confrez/control/vehicle_follower.py:146-368`):

  variables per stage  x,y,psi,v,delta (:167-171), a,w (:173-174), l,m (:176-177),
                       per neighbour lambda_ij, lambda_ji, s (:311-313)
  parameters           current state (:179-183), reference x,y,psi (:185-187),
                       neighbours' predicted x,y,psi (:305-309)
  constraints          initial state (:194-199), l,m>=0 (:201-202), boxes (:204-240),
                       RK4 dynamics (:243-260), static-obstacle OBCA (:280-290),
                       lambda>=0 (:315-316), vehicle-vehicle OBCA (:322-352)
  cost                 (:263-272)

Two views of the same NLP are provided:

* `reference_residuals` evaluates objective and every constraint *exactly in the
  reference's own variable layout* (l[N,4*n_obs], m[N,4*n_obs], lambda_ij[o][N,4],
  lambda_ji[o][N,4], s[o][N,2]).  It is the solver-independent acceptance check.

* `MpcNlp` is the IPOPT-form problem  min f(X) s.t. c(X)=0, XL<=X<=XU  that the
  interior-point oracle (`oracle/ipm.py`) and the HIP kernel both solve.  It is the
  reference NLP with the OBCA duals eliminated by partial maximisation (DESIGN.md
  "Certificate elimination"): for a fixed pose the duals of one (stage, obstacle) or
  (stage, neighbour) block only have to *exist*, and the best they can certify is the
  separation of the two polygons, so each block collapses to one inequality
        sep_j(x_k, y_k, psi_k) - dmin - sigma = 0,   sigma >= 0,
  where sep_j is evaluated in closed form over the face normals of both polygons
  (`block_separation`), and the duals l, m, lambda_ij, lambda_ji, s that certify it are
  reconstructed on output (`certificate_duals`).  They satisfy every dual-variable
  constraint of the reference exactly (:289-290, :350-352) and the separation rows
  (:287, :348) whenever sep_j >= dmin.  Face normals only: near a corner-to-corner
  closest pair the certified separation under-estimates the Euclidean distance by at most
  a factor cos(45 deg), i.e. the solver keeps up to 0.3*dmin = 1.5 cm more clearance there
  than the reference NLP demands.
"""
from dataclasses import dataclass, field
import numpy as np
import scipy.sparse as sp

from .dynamics import bicycle_rk4, bicycle_rk4_jac

NP = 7  # x,y,psi,v,delta,a,w


@dataclass
class MpcSpec:
    """Constants of the NLP (mirrors `cfz_spec` in include/confrez_hip.h)."""

    N: int = 30
    dt: float = 0.1
    A_obs: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 2)))
    b_obs: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    n_nbr: int = 3
    # vehicle body polytope, reference vehicle_types.py:65-71
    G: np.ndarray = field(default_factory=lambda: np.array([[1.0, 0], [0, 1], [-1, 0], [0, -1]]))
    g: np.ndarray = field(default_factory=lambda: np.array([3.3, 0.9, 0.6, 0.9]))
    wb: float = 2.5
    # x,y (obstacle_types.py:10-15) then v,delta,a,w (vehicle_types.py:81-90): lo,hi pairs
    bounds: np.ndarray = field(
        default_factory=lambda: np.array([2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0])
    )
    dmin: float = 0.05
    # weights on (x-xr)^2,(y-yr)^2,(psi-psir)^2,a^2,(v w)^2,delta^2  (vehicle_follower.py:263-271)
    weights: np.ndarray = field(default_factory=lambda: np.array([100.0, 100, 100, 1, 1, 1]))
    rk_substeps: int = 4

    @property
    def n_obs(self):
        return int(self.A_obs.shape[0])

    @property
    def n_blk(self):
        return self.n_obs + self.n_nbr


def rot(psi):
    c, s = np.cos(psi), np.sin(psi)
    return np.array([[c, -s], [s, c]])


# --------------------------------------------------------------------------------------
# reference-layout evaluation (acceptance check, independent of any solver formulation)
# --------------------------------------------------------------------------------------
def reference_residuals(spec: MpcSpec, x0, ref, nbr, sol):
    """Objective and constraint violations of `sol` in the reference's own formulation.

    sol: dict with x,y,psi,v,delta,a,w [N]; l,m [N,4*n_obs]; lam_ij,lam_ji [n_nbr,N,4]; s [n_nbr,N,2].
    Returns dict(cost, eq (max |equality residual|), ineq (max inequality violation),
                 bound (max bound violation)).
    Written as plain per-stage loops on purpose (mirrors vehicle_follower.py:204-352 line by line).
    """
    N, G, g = spec.N, spec.G, spec.g
    lo, hi = spec.bounds[0::2], spec.bounds[1::2]
    wt = spec.weights
    x, y, psi, v, de, a, w = (np.asarray(sol[k], float) for k in ("x", "y", "psi", "v", "delta", "a", "w"))
    l, m = np.asarray(sol["l"], float), np.asarray(sol["m"], float)
    eq, ineq, bnd, cost = 0.0, 0.0, 0.0, 0.0
    z = np.stack([x, y, psi, v, de], -1)
    eq = max(eq, np.abs(z[0] - np.asarray(x0)).max())  # :194-199
    bnd = max(bnd, (-l).max(initial=0.0), (-m).max(initial=0.0))  # :201-202
    for i in range(N):
        for val, j in ((x[i], 0), (y[i], 1), (v[i], 2), (de[i], 3), (a[i], 4), (w[i], 5)):  # :205-240
            bnd = max(bnd, lo[j] - val, val - hi[j])
        if i < N - 1:  # :243-260
            zn = bicycle_rk4(z[i], np.array([a[i], w[i]]), spec.dt, spec.wb, spec.rk_substeps)
            eq = max(eq, np.abs(z[i + 1] - zn).max())
        cost += (
            wt[0] * (x[i] - ref[0, i]) ** 2
            + wt[1] * (y[i] - ref[1, i]) ** 2
            + wt[2] * (psi[i] - ref[2, i]) ** 2
            + wt[3] * a[i] ** 2
            + wt[4] * v[i] ** 2 * w[i] ** 2
            + wt[5] * de[i] ** 2
        )  # :263-272
        t = np.array([x[i], y[i]])
        R = rot(psi[i])
        for j in range(spec.n_obs):  # :280-290
            A, b = spec.A_obs[j], spec.b_obs[j]
            lj, mj = l[i, 4 * j : 4 * j + 4], m[i, 4 * j : 4 * j + 4]
            ineq = max(ineq, spec.dmin - (np.dot(-g, mj) + np.dot(A @ t - b, lj)))
            eq = max(eq, np.abs(G.T @ mj + R.T @ A.T @ lj).max())
            eq = max(eq, abs(np.dot(A.T @ lj, A.T @ lj) - 1.0))
        for o in range(spec.n_nbr):  # :322-352
            lik, mik, sik = sol["lam_ij"][o][i], sol["lam_ji"][o][i], sol["s"][o][i]
            bnd = max(bnd, (-np.asarray(lik)).max(), (-np.asarray(mik)).max())  # :315-316
            this_R = rot(-psi[i])
            this_A = G @ this_R
            this_b = G @ this_R @ t + g
            ot = np.array([nbr[o, 0, i], nbr[o, 1, i]])
            other_R = rot(-nbr[o, 2, i])
            other_A = G @ other_R
            other_b = G @ other_R @ ot + g
            ineq = max(ineq, spec.dmin - (-np.dot(this_b, lik) - np.dot(other_b, mik)))
            eq = max(eq, np.abs(this_A.T @ lik + sik).max())
            eq = max(eq, np.abs(other_A.T @ mik - sik).max())
            ineq = max(ineq, np.dot(sik, sik) - 1.0)
    return dict(cost=float(cost), eq=float(eq), ineq=float(max(ineq, 0.0)), bound=float(max(bnd, 0.0)))


# --------------------------------------------------------------------------------------
# closed-form separation certificates
# --------------------------------------------------------------------------------------
G_BODY = np.array([[1.0, 0], [0, 1], [-1, 0], [0, -1]])


def body_vertices(g):
    """Corners of the body rectangle {G p <= g}, counter-clockwise from front-left."""
    return np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])


def polytope_vertices(A, b):
    """Vertices of the 4-face polygon {A p <= b}; returns (V[4,2], adj[4,2]) where adj holds
    the two faces meeting at each vertex.  Faces are paired in index order (i<j)."""
    V, adj = [], []
    for i in range(4):
        for j in range(i + 1, 4):
            det = A[i, 0] * A[j, 1] - A[i, 1] * A[j, 0]
            if abs(det) < 1e-9:
                continue
            p = np.array([(b[i] * A[j, 1] - A[i, 1] * b[j]) / det, (A[i, 0] * b[j] - b[i] * A[j, 0]) / det])
            if np.all(A @ p <= b + 1e-9):
                V.append(p)
                adj.append((i, j))
    assert len(V) == 4, "obstacle must be a bounded quadrilateral"
    return np.array(V), np.array(adj)


def block_separation(A, b, PV, t, psi, g, BV):
    """Separation of polygon (A,b; vertices PV) from the body rectangle at pose (t,psi).

    sep = max over the 8 face normals of (min over the other polygon's vertices of the signed
    distance to that face).  Returns (sep, grad wrt (x,y,psi), (kind, face, vertex)) with
    kind 1 = polygon face / body vertex, kind 2 = body face / polygon vertex.  Ties keep the
    first candidate in the order polygon faces 0..3, body faces 0..3 (strict `>`).
    """
    c, s = np.cos(psi), np.sin(psi)
    R = np.array([[c, -s], [s, c]])
    dR = np.array([[-s, -c], [c, -s]])
    W = t + BV @ R.T
    best = None
    for i in range(4):
        d = W @ A[i] - b[i]
        v = int(np.argmin(d))
        if best is None or d[v] > best[0]:
            best = (d[v], np.array([A[i, 0], A[i, 1], A[i] @ (dR @ BV[v])]), (1, i, v))
    for k in range(4):
        nk = R @ G_BODY[k]
        d = (PV - t) @ nk - g[k]
        v = int(np.argmin(d))
        if d[v] > best[0]:
            best = (d[v], np.array([-nk[0], -nk[1], (dR @ G_BODY[k]) @ (PV[v] - t)]), (2, k, v))
    return best


def _posneg(m):
    return np.array([max(m[0], 0.0), max(m[1], 0.0), max(-m[0], 0.0), max(-m[1], 0.0)])


def certificate_duals(A, adj, psi, cert, psi_other=None):
    """(lam, mu) of the reference's dual constraints for the winning candidate `cert`.

    Static obstacle (psi_other None): lam multiplies the obstacle faces, mu the body faces
    (vehicle_follower.py:283-290).  Neighbour: lam multiplies this vehicle's faces, mu the
    other vehicle's (:323-352); `A` is then unused."""
    kind, f, v = cert
    c, s = np.cos(psi), np.sin(psi)
    R = np.array([[c, -s], [s, c]])
    lam, mu = np.zeros(4), np.zeros(4)
    if psi_other is None:
        if kind == 1:  # n = A_f ; G^T mu = -R^T n
            lam[f] = 1.0
            mu = _posneg(-R.T @ A[f])
        else:  # n = -R G_f ; A^T lam = n from the two faces meeting at polygon vertex v
            mu[f] = 1.0
            n = -R @ G_BODY[f]
            i, j = adj[v]
            det = A[i, 0] * A[j, 1] - A[j, 0] * A[i, 1]
            lam[i] = max((A[j, 1] * n[0] - A[j, 0] * n[1]) / det, 0.0)
            lam[j] = max((-A[i, 1] * n[0] + A[i, 0] * n[1]) / det, 0.0)
    else:
        co, so = np.cos(psi_other), np.sin(psi_other)
        Ro = np.array([[co, -so], [so, co]])
        if kind == 1:  # separating direction = a face normal of the OTHER vehicle: w = -Ro G_f
            mu[f] = 1.0
            lam = _posneg(R.T @ (-Ro @ G_BODY[f]))
        else:  # a face normal of this vehicle: w = R G_f
            lam[f] = 1.0
            mu = _posneg(-Ro.T @ (R @ G_BODY[f]))
    return lam, mu


# --------------------------------------------------------------------------------------
# IPOPT-form NLP
# --------------------------------------------------------------------------------------
class MpcNlp:
    """min f(X) s.t. c(X)=0, XL<=X<=XU for one vehicle's MPC step.

    X layout, stage-major, per stage k (stride n_stage = 7 + n_obs + n_nbr):
        [x y psi v delta a w | sigma_0 .. sigma_{n_blk-1}]      (obstacles first, then neighbours)
    c layout: [z0 - x0 (5) | F(z_k,u_k) - z_{k+1}, k<N-1 (5 each) | per stage: sep_j - dmin - sigma_j]
    """

    def __init__(self, spec: MpcSpec, x0, ref, nbr=None):
        self.spec = spec
        self.x0 = np.asarray(x0, float).reshape(5)
        self.ref = np.asarray(ref, float).reshape(3, spec.N)
        self.nbr = (
            np.asarray(nbr, float).reshape(spec.n_nbr, 3, spec.N) if spec.n_nbr else np.zeros((0, 3, spec.N))
        )
        N = spec.N
        self.nb = spec.n_obs + spec.n_nbr
        self.ns = NP + self.nb
        self.n = N * self.ns
        self.m = 5 + 5 * (N - 1) + self.nb * N
        self.c_blk0 = 5 + 5 * (N - 1)
        xl = np.full((N, self.ns), -np.inf)
        xu = np.full((N, self.ns), np.inf)
        b = spec.bounds
        for col, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
            xl[:, col], xu[:, col] = b[2 * j], b[2 * j + 1]
        xl[:, NP:] = 0.0
        self.xl, self.xu = xl.ravel(), xu.ravel()
        self.block_mask = np.zeros(self.n)
        pv = [polytope_vertices(spec.A_obs[j], spec.b_obs[j]) for j in range(spec.n_obs)]
        self.PV = [p[0] for p in pv]
        self.adj = [p[1] for p in pv]
        self.BV = body_vertices(spec.g)

    # ---- blocks --------------------------------------------------------------------
    def neighbour_polygon(self, o, k):
        """H-rep and vertices of neighbour o's body at its predicted pose of stage k."""
        to, po = self.nbr[o, 0:2, k], self.nbr[o, 2, k]
        Ro = rot(po)
        A = G_BODY @ Ro.T
        return A, A @ to + self.spec.g, to + self.BV @ Ro.T

    def blocks(self, P):
        """sep [N,nb], grad wrt (x,y,psi) [N,nb,3], certificates."""
        sp_ = self.spec
        sep = np.zeros((sp_.N, self.nb))
        gr = np.zeros((sp_.N, self.nb, 3))
        cert = {}
        for k in range(sp_.N):
            t, psi = P[k, 0:2], P[k, 2]
            for j in range(sp_.n_obs):
                r = block_separation(sp_.A_obs[j], sp_.b_obs[j], self.PV[j], t, psi, sp_.g, self.BV)
                sep[k, j], gr[k, j], cert[(k, j)] = r
            for o in range(sp_.n_nbr):
                A, b, PV = self.neighbour_polygon(o, k)
                r = block_separation(A, b, PV, t, psi, sp_.g, self.BV)
                sep[k, sp_.n_obs + o], gr[k, sp_.n_obs + o], cert[(k, sp_.n_obs + o)] = r
        return sep, gr, cert

    # ---- packing between reference layout and X ------------------------------------
    def pack(self, sol):
        """Reference-layout dict -> X.  Only the primal arrays are read; warm-start duals are
        not needed (certificates are recomputed from the poses).  Slacks follow IPOPT's
        s = g(x0) rule."""
        N = self.spec.N
        X = np.zeros((N, self.ns))
        for i, key in enumerate(("x", "y", "psi", "v", "delta", "a", "w")):
            X[:, i] = sol[key]
        X[:, NP:] = self.blocks(X)[0] - self.spec.dmin
        return X.ravel()

    def unpack(self, X):
        sp_, N = self.spec, self.spec.N
        Xs = np.asarray(X).reshape(N, self.ns)
        sol = {key: Xs[:, i].copy() for i, key in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
        sol["l"] = np.zeros((N, 4 * sp_.n_obs))
        sol["m"] = np.zeros((N, 4 * sp_.n_obs))
        sol["lam_ij"] = np.zeros((sp_.n_nbr, N, 4))
        sol["lam_ji"] = np.zeros((sp_.n_nbr, N, 4))
        sol["s"] = np.zeros((sp_.n_nbr, N, 2))
        sep, _, cert = self.blocks(Xs)
        sol["sep"] = sep
        for k in range(N):
            psi = Xs[k, 2]
            R = rot(psi)
            for j in range(sp_.n_obs):
                lam, mu = certificate_duals(sp_.A_obs[j], self.adj[j], psi, cert[(k, j)])
                sol["l"][k, 4 * j : 4 * j + 4], sol["m"][k, 4 * j : 4 * j + 4] = lam, mu
            for o in range(sp_.n_nbr):
                lam, mu = certificate_duals(None, None, psi, cert[(k, sp_.n_obs + o)], self.nbr[o, 2, k])
                sol["lam_ij"][o, k], sol["lam_ji"][o, k] = lam, mu
                sol["s"][o, k] = -R @ (G_BODY.T @ lam)  # from A_this^T lam + s = 0 (:350)
        return sol

    # ---- objective -----------------------------------------------------------------
    def f(self, X):
        P = X.reshape(self.spec.N, -1)
        wt, r = self.spec.weights, self.ref
        return float(
            np.sum(
                wt[0] * (P[:, 0] - r[0]) ** 2
                + wt[1] * (P[:, 1] - r[1]) ** 2
                + wt[2] * (P[:, 2] - r[2]) ** 2
                + wt[3] * P[:, 5] ** 2
                + wt[4] * P[:, 3] ** 2 * P[:, 6] ** 2
                + wt[5] * P[:, 4] ** 2
            )
        )

    def grad(self, X):
        P = X.reshape(self.spec.N, -1)
        wt, r = self.spec.weights, self.ref
        Gd = np.zeros_like(P)
        Gd[:, 0] = 2 * wt[0] * (P[:, 0] - r[0])
        Gd[:, 1] = 2 * wt[1] * (P[:, 1] - r[1])
        Gd[:, 2] = 2 * wt[2] * (P[:, 2] - r[2])
        Gd[:, 3] = 2 * wt[4] * P[:, 3] * P[:, 6] ** 2
        Gd[:, 4] = 2 * wt[5] * P[:, 4]
        Gd[:, 5] = 2 * wt[3] * P[:, 5]
        Gd[:, 6] = 2 * wt[4] * P[:, 3] ** 2 * P[:, 6]
        return Gd.ravel()

    def hess_gn(self, X):
        """Gauss-Newton Hessian of the Lagrangian: objective curvature only, with the
        (v w)^2 term taken as the square of the residual r = v*w (PSD)."""
        N, ns = self.spec.N, self.ns
        P = X.reshape(N, ns)
        wt = self.spec.weights
        base = np.arange(N) * ns
        rows, cols, vals = [], [], []

        def add(i, j, v):
            rows.append(base + i), cols.append(base + j), vals.append(np.broadcast_to(v, (N,)))

        add(0, 0, 2 * wt[0]), add(1, 1, 2 * wt[1]), add(2, 2, 2 * wt[2])
        add(4, 4, 2 * wt[5]), add(5, 5, 2 * wt[3])
        v, w = P[:, 3], P[:, 6]
        add(3, 3, 2 * wt[4] * w * w), add(6, 6, 2 * wt[4] * v * v)
        add(3, 6, 2 * wt[4] * v * w), add(6, 3, 2 * wt[4] * v * w)
        return sp.csr_matrix(
            (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.n, self.n)
        )

    # ---- constraints ---------------------------------------------------------------
    def cons(self, X):
        return self._cons_jac(X, want_jac=False)[0]

    def jac(self, X):
        return self._cons_jac(X, want_jac=True)[1]

    def _cons_jac(self, X, want_jac=True):
        sp_ = self.spec
        N, ns = sp_.N, self.ns
        P = X.reshape(N, ns)
        cvec = np.zeros(self.m)
        rows, cols, vals = [], [], []
        base = np.arange(N) * ns

        def add(r, cidx, v):
            if want_jac:
                r = np.asarray(r).ravel()
                rows.append(r), cols.append(np.asarray(cidx).ravel())
                vals.append(np.broadcast_to(v, r.shape).ravel().astype(float))

        cvec[0:5] = P[0, 0:5] - self.x0
        add(np.arange(5), np.arange(5), 1.0)
        if N > 1:
            z, u = P[:-1, 0:5], P[:-1, 5:7]
            if want_jac:
                F, Fz, Fu = bicycle_rk4_jac(z, u, sp_.dt, sp_.wb, sp_.rk_substeps)
            else:
                F = bicycle_rk4(z, u, sp_.dt, sp_.wb, sp_.rk_substeps)
            cvec[5 : 5 + 5 * (N - 1)] = (F - P[1:, 0:5]).ravel()
            if want_jac:
                kk = np.arange(N - 1)
                for i in range(5):
                    r = 5 + 5 * kk + i
                    for j in range(5):
                        add(r, base[:-1] + j, Fz[:, i, j])
                    for j in range(2):
                        add(r, base[:-1] + 5 + j, Fu[:, i, j])
                    add(r, base[1:] + i, -1.0)
        sep, gr, _ = self.blocks(P)
        kk = np.arange(N)
        for j in range(self.nb):
            r = self.c_blk0 + self.nb * kk + j
            cvec[r] = sep[:, j] - sp_.dmin - P[:, NP + j]
            for q in range(3):
                add(r, base + q, gr[:, j, q])
            add(r, base + NP + j, -1.0)
        J = None
        if want_jac:
            J = sp.csr_matrix(
                (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.m, self.n)
            )
        return cvec, J
