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
  separation of the two polygons along a face normal, i.e. "every vertex of one polygon is
  at least dmin outside the best face of the other".  Each block therefore collapses to
  two smooth rows (the two vertices nearest to that face; the (face, vertex, vertex) working
  set is chosen by `select_rows` at every accepted iterate and held fixed inside the line
  search, so the merit function of one iteration is smooth)
        sep_{j,r}(x_k, y_k, psi_k) - dmin - sigma_{j,r} = 0,   sigma_{j,r} >= 0,  r = 0,1
  and the duals l, m, lambda_ij, lambda_ji, s that certify it are
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
BCOLS = (0, 1, 3, 4, 5, 6)  # bounded columns of a stage: x y v delta a w
# constants of the restoration phase (MpcNlp.restore; oracle/cfz_port.c and the kernel carry the same numbers)
RESTO_RHO = 1000.0        # weight of a squared row violation
RESTO_RHO_BOX = 1e5       # weight of a squared box excess
RESTO_BOX_MARGIN = 2e-3   # the boxes are restored with this margin (half of it is kept on return)
RESTO_MAX_ITER = 40
RESTO_STALL = 8           # iterations without a drop of the worst violation by a tenth: stationary, locally infeasible
RESTO_KAPPA = 0.1         # goal: a tenth of the violation at entry (IPOPT's required_infeasibility_reduction is 0.9)
RESTO_ARMIJO = 1e-4


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
    # vertex-vertex rows (kind 3): a block whose closest features are two vertices is constrained by the Euclidean distance of
    # that pair -- with them the eliminated problem has the reference's feasible set; False = face-normal certificates only
    vv_rows: bool = True

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
    """Vertices of the 4-face polygon {A p <= b} in counter-clockwise order; returns (V[4,2], adj[4,2]) where adj holds
    the two faces meeting at each vertex."""
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
    V, adj = np.array(V), np.array(adj)
    # counter-clockwise around the polygon starting from the first one found, so that vertices v-1, v+1 (mod 4) are the
    # neighbours of v -- what select_rows assumes
    ang = np.arctan2(V[:, 1] - V[:, 1].mean(), V[:, 0] - V[:, 0].mean())
    order = np.argsort((ang - ang[0]) % (2 * np.pi), kind="stable")
    return V[order], adj[order]


HYST = 1e-3  # m: a block keeps its separating face until another one is better by this much


def vertex_distances(A, b, PV, t, psi, g, BV, kind, f):
    """Signed distances of the 4 vertices of one polygon to face f of the other, and their
    gradients wrt (x,y,psi).  kind 1 = polygon face / body vertices, 2 = body face / polygon vertices."""
    c, s = np.cos(psi), np.sin(psi)
    R = np.array([[c, -s], [s, c]])
    dR = np.array([[-s, -c], [c, -s]])
    d = np.zeros(4)
    gr = np.zeros((4, 3))
    if kind == 1:
        W = t + BV @ R.T
        d = W @ A[f] - b[f]
        for v in range(4):
            gr[v] = [A[f, 0], A[f, 1], A[f] @ (dR @ BV[v])]
    else:
        nk = R @ G_BODY[f]
        d = (PV - t) @ nk - g[f]
        for v in range(4):
            gr[v] = [-nk[0], -nk[1], (dR @ G_BODY[f]) @ (PV[v] - t)]
    return d, gr


def pose_shift(a00, a11, a22, a01, a02, a12):
    """max(0, -lambda_min) of the symmetric 3 x 3 [[a00 a01 a02] [a01 a11 a12] [a02 a12 a22]]: zero when the leading
    minors pass, else the trigonometric closed form of the smallest eigenvalue (same arithmetic as the kernel)."""
    d2 = a00 * a11 - a01 * a01
    d3 = a22 * d2 - (a02 * a02 * a11 - 2.0 * a02 * a12 * a01 + a12 * a12 * a00)
    if a00 > 0.0 and d2 > 0.0 and d3 >= 0.0:
        return 0.0
    p1 = a01 * a01 + a02 * a02 + a12 * a12
    qm = (a00 + a11 + a22) / 3.0
    b00, b11, b22 = a00 - qm, a11 - qm, a22 - qm
    p = np.sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * p1) / 6.0)
    ip = 1.0 / p
    c00, c11, c22, c01, c02, c12 = b00 * ip, b11 * ip, b22 * ip, a01 * ip, a02 * ip, a12 * ip
    r = 0.5 * (c00 * (c11 * c22 - c12 * c12) - c01 * (c01 * c22 - c12 * c02) + c02 * (c01 * c12 - c11 * c02))
    r = min(1.0, max(-1.0, r))
    lam = qm + 2.0 * p * np.cos(np.arccos(r) / 3.0 + 2.0943951023931953)
    return max(0.0, -lam)


def closest_vertex_pair(PV, t, psi, g, BV):
    """(u, v): polygon vertex PV[u] and body vertex v that are each other's closest feature -- each lies in the other's normal
    cone (beyond both edges that meet there) -- which makes them THE closest points of the two convex polygons; None when the
    closest features are not two vertices.  Body side: PV[u] is outside exactly the two body faces that meet at v (signs of
    the kind-2 distances); polygon side: (W_v - PV_u).e <= 0 for the two edges e leaving PV[u]."""
    c, s_ = np.cos(psi), np.sin(psi)
    R = np.array([[c, -s_], [s_, c]])
    for u in range(4):
        q = R.T @ (PV[u] - t)  # in the body frame
        fx = 0 if q[0] - g[0] >= 0.0 else (2 if -q[0] - g[2] >= 0.0 else -1)
        fy = 1 if q[1] - g[1] >= 0.0 else (3 if -q[1] - g[3] >= 0.0 else -1)
        if fx < 0 or fy < 0:
            continue
        v = {(0, 1): 0, (2, 1): 1, (2, 3): 2, (0, 3): 3}[(fx, fy)]
        w = t + R @ BV[v] - PV[u]
        if w @ (PV[(u + 1) % 4] - PV[u]) <= 0.0 and w @ (PV[(u + 3) % 4] - PV[u]) <= 0.0:
            return u, v, float(np.hypot(w[0], w[1]))
    return None


def select_rows(A, b, PV, t, psi, g, BV, prev=0, vv=False, vv_enter=0.0):
    """Working set of one block: the separating face and the two vertices whose rows are imposed.

    The separating direction is the face normal (8 candidates in the order polygon faces 0..3,
    body faces 0..3) with the largest  min over the other polygon's vertices of the signed
    vertex-face distance; the previous face is kept unless another beats it by more than HYST.
    `min over vertices >= dmin` is imposed as one smooth row per vertex; only the nearest vertex
    and the nearer of its two neighbours are kept (for a convex polygon the minimum sits on one
    vertex or, when an edge is parallel to the face, on two adjacent ones).  Returns the code
    sel = kind*64 + face*16 + vA*4 + vB with vA < vB (vertex indices: identity of the two rows).
    """
    best, bk, bf = None, 0, 0
    pk, pf = (prev >> 6), (prev >> 4) & 3
    prev_val = None
    for kind in (1, 2):
        for f in range(4):
            d, _ = vertex_distances(A, b, PV, t, psi, g, BV, kind, f)
            val = d.min()
            if prev and kind == pk and f == pf:
                prev_val = val
            if best is None or val > best:
                best, bk, bf = val, kind, f
    if prev_val is not None and prev_val >= best - HYST:
        bk, bf = pk, pf
    d, _ = vertex_distances(A, b, PV, t, psi, g, BV, bk, bf)
    v0 = int(np.argmin(d))
    n1, n2 = (v0 + 1) % 4, (v0 + 3) % 4
    if d[n1] < d[n2]:
        v1 = n1
    elif d[n2] < d[n1]:
        v1 = n2
    else:
        v1 = min(n1, n2)
    if prev and bk == pk and bf == pf:  # keep the old pair while it still holds the minimum
        oa, ob = (prev >> 2) & 3, prev & 3
        if min(d[oa], d[ob]) <= d[v0] + 1e-12 and max(d[oa], d[ob]) <= d[v1] + HYST:
            v0, v1 = oa, ob
    va, vb = min(v0, v1), max(v0, v1)
    dn = min(d[v0], d[v1])  # the separation this face certifies (a kept pair is not ordered by distance)
    if vv and dn > 0.0:
        # kind 3: the closest features are two vertices -> the Euclidean distance of that pair is the separation (it exceeds
        # every face-normal separation there); code 192 + u*16 + v*4 + v, both rows of the block carry that distance
        # vv_enter (the joint plan's pair blocks: 1e-4 while mu >= 1e-4): a face block turns vertex-vertex only beyond that margin
        pair = closest_vertex_pair(PV, t, psi, g, BV)
        enter = vv_enter if (prev and (prev >> 6) != 3 and vv_enter > 1e-9) else 1e-9
        if pair is not None and pair[2] > dn + enter:
            return 192 + pair[0] * 16 + pair[1] * 4 + pair[1]
    return bk * 64 + bf * 16 + va * 4 + vb


def rows_for(A, b, PV, t, psi, g, BV, sel):
    """Values and gradients of the two rows of working set `sel`: (sep[2], grad[2,3])."""
    kind, f, va, vb = sel >> 6, (sel >> 4) & 3, (sel >> 2) & 3, sel & 3
    if kind == 3:
        u, v = f, va
        c, s_ = np.cos(psi), np.sin(psi)
        R = np.array([[c, -s_], [s_, c]]); dR = np.array([[-s_, -c], [c, -s_]])
        w = t + R @ BV[v] - PV[u]
        r = float(np.hypot(w[0], w[1]))
        n = w / r
        gr = np.array([n[0], n[1], n @ (dR @ BV[v])])
        return np.array([r, r]), np.array([gr, gr])  # the row twice: the block keeps its two slots (twice the barrier weight)
    d, gr = vertex_distances(A, b, PV, t, psi, g, BV, kind, f)
    return d[[va, vb]], gr[[va, vb]]


def _posneg(m):
    return np.array([max(m[0], 0.0), max(m[1], 0.0), max(-m[0], 0.0), max(-m[1], 0.0)])


def certificate_duals(A, adj, psi, cert, psi_other=None, n=None):
    """(lam, mu) of the reference's dual constraints for the winning candidate `cert`.
    kind 3 (vertex against vertex): `n` is the unit vector from polygon vertex u = cert[1] to body vertex v = cert[2].

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
        else:  # kind 2: n = -R G_f, mu = e_f; kind 3: G^T mu = -R^T n; A^T lam = n from the two faces meeting at the polygon vertex
            if kind == 2:
                mu[f] = 1.0
                n = -R @ G_BODY[f]
            else:
                mu = _posneg(-R.T @ n)
                v = f
            i, j = adj[v]
            det = A[i, 0] * A[j, 1] - A[j, 0] * A[i, 1]
            lam[i] = max((A[j, 1] * n[0] - A[j, 0] * n[1]) / det, 0.0)
            lam[j] = max((-A[i, 1] * n[0] + A[i, 0] * n[1]) / det, 0.0)
    else:
        co, so = np.cos(psi_other), np.sin(psi_other)
        Ro = np.array([[co, -so], [so, co]])
        if kind == 3:  # w = -n from this vehicle to the other
            lam, mu = _posneg(R.T @ (-n)), _posneg(-Ro.T @ (-n))
        elif kind == 1:  # separating direction = a face normal of the OTHER vehicle: w = -Ro G_f
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

    X layout, stage-major, per stage k (stride 7 + 2 n_blk):
        [x y psi v delta a w | sigma_{j,r}, row index 2 j + r]   (obstacles first, then neighbours)
    c layout: [z0 - x0 (5) | F(z_k,u_k) - z_{k+1}, k<N-1 (5 each) | per stage: sep_{j,r} - dmin - sigma_{j,r}]
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
        self.nr = 2 * self.nb  # constraint rows per stage
        self.ns = NP + self.nr
        self.n = N * self.ns
        self.m = 5 + 5 * (N - 1) + self.nr * N
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
        self.sel = np.zeros((N, self.nb), dtype=np.int64)  # working set, 0 = none yet

    # ---- blocks --------------------------------------------------------------------
    def neighbour_polygon(self, o, k):
        """H-rep and vertices of neighbour o's body at its predicted pose of stage k."""
        to, po = self.nbr[o, 0:2, k], self.nbr[o, 2, k]
        Ro = rot(po)
        A = G_BODY @ Ro.T
        return A, A @ to + self.spec.g, to + self.BV @ Ro.T

    def polygon(self, k, j):
        if j < self.spec.n_obs:
            return self.spec.A_obs[j], self.spec.b_obs[j], self.PV[j]
        return self.neighbour_polygon(j - self.spec.n_obs, k)

    def select(self, P):
        """Recomputes the working set at poses P [N,>=3]; returns the previous one."""
        old = self.sel.copy()
        for k in range(self.spec.N):
            for j in range(self.nb):
                A, b, PV = self.polygon(k, j)
                self.sel[k, j] = select_rows(A, b, PV, P[k, 0:2], P[k, 2], self.spec.g, self.BV, int(old[k, j]), vv=self.spec.vv_rows)
        return old

    def blocks(self, P):
        """Rows of the current working set: sep [N,2 nb], grad wrt (x,y,psi) [N,2 nb,3]."""
        sp_ = self.spec
        sep = np.zeros((sp_.N, self.nr))
        gr = np.zeros((sp_.N, self.nr, 3))
        for k in range(sp_.N):
            for j in range(self.nb):
                A, b, PV = self.polygon(k, j)
                sep[k, 2 * j : 2 * j + 2], gr[k, 2 * j : 2 * j + 2] = rows_for(
                    A, b, PV, P[k, 0:2], P[k, 2], sp_.g, self.BV, int(self.sel[k, j]))
        return sep, gr

    has_row_curvature = True

    def row_curvature(self, P, NU):
        """C[k] = sum over the rows of stage k of nu_r * Hessian of sep_r wrt (x, y, psi) -> [N,3,3].
        kind 1 (polygon face A_f, body vertex b_v):  sep = A_f.(t + R b_v) - b_f       -> only d2/dpsi2 = -A_f.(R b_v)
        kind 2 (body face g_f, polygon vertex p_v):  sep = (p_v - t).(R G_f) - g_f     -> d2/dpsi2 = -(p_v - t).(R G_f),
                                                                                         d2/dt dpsi = -R' G_f"""
        sp_ = self.spec
        C = np.zeros((sp_.N, 3, 3))
        for k in range(sp_.N):
            c, s_ = np.cos(P[k, 2]), np.sin(P[k, 2])
            R = np.array([[c, -s_], [s_, c]])
            dR = np.array([[-s_, -c], [c, -s_]])
            t = P[k, 0:2]
            for j in range(self.nb):
                A, b, PV = self.polygon(k, j)
                sel = int(self.sel[k, j])
                kind, f, vs = sel >> 6, (sel >> 4) & 3, ((sel >> 2) & 3, sel & 3)
                if kind == 3:  # r = |w|, w = t + R b_v - p_u:  J'(I - n n')J / r  +  n.(-R b_v) e_psi e_psi'
                    u, v = f, vs[0]
                    w = t + R @ self.BV[v] - PV[u]
                    r_ = float(np.hypot(w[0], w[1])); n = w / r_
                    J = np.array([[1.0, 0.0, (dR @ self.BV[v])[0]], [0.0, 1.0, (dR @ self.BV[v])[1]]])
                    Hr = J.T @ (np.eye(2) - np.outer(n, n)) @ J / r_
                    Hr[2, 2] += n @ (-(R @ self.BV[v]))
                    C[k] += (NU[k, 2 * j] + NU[k, 2 * j + 1]) * Hr
                    continue
                for r, v in enumerate(vs):
                    n_ = NU[k, 2 * j + r]
                    if kind == 1:
                        C[k, 2, 2] += n_ * (-(A[f] @ (R @ self.BV[v])))
                    else:
                        C[k, 2, 2] += n_ * (-((PV[v] - t) @ (R @ G_BODY[f])))
                        m = -(dR @ G_BODY[f])
                        C[k, 0, 2] += n_ * m[0]; C[k, 2, 0] += n_ * m[0]
                        C[k, 1, 2] += n_ * m[1]; C[k, 2, 1] += n_ * m[1]
        C[0] = 0.0  # the rows of stage 0 are constants (see _cons_jac)
        return C

    def new_iterate(self, x, zl, nu, mu, bound_push):
        """Hook of oracle/ipm.py, called with every accepted iterate: refresh the working set.
        A row that keeps its (face, vertex) identity keeps slack and multipliers; a new row starts
        at sigma = max(sep - dmin, bound_push), z = mu / sigma, nu = -z.
        (Round 3 tried sigma = max(sep - dmin, min(bound_push, max(mu, 1e-8))), the rule of the planning kernels' hand-over: it ends
        a period-3 limit cycle of the closed loop on the planned table, but three instances of the independent-solver populations
        then fail; docs/notebook.md.)"""
        N = self.spec.N
        Xs = x.reshape(N, self.ns)
        old = self.select(Xs)
        self.ws_changed = not np.array_equal(old, self.sel)  # read by the stall test of oracle/ipm.py
        if not self.ws_changed:
            return x, zl, nu
        sep, _ = self.blocks(Xs)
        Z = zl.reshape(N, self.ns)
        NU = nu[self.c_blk0 :].reshape(N, self.nr)
        for k, j in zip(*np.nonzero(old != self.sel)):
            o, n_ = int(old[k, j]), int(self.sel[k, j])
            same_face = (o >> 4) == (n_ >> 4)
            keep = {}
            if same_face:
                keep = {(o >> 2) & 3: 0, o & 3: 1}
            vals = [(Xs[k, NP + 2 * j + r], Z[k, NP + 2 * j + r], NU[k, 2 * j + r]) for r in range(2)]
            for r, v in enumerate(((n_ >> 2) & 3, n_ & 3)):
                if v in keep:
                    sg, z, nn = vals[keep[v]]
                else:
                    sg = max(sep[k, 2 * j + r] - self.spec.dmin, bound_push)
                    z = mu / sg
                    nn = -z
                Xs[k, NP + 2 * j + r], Z[k, NP + 2 * j + r], NU[k, 2 * j + r] = sg, z, nn
        return x, zl, nu


    # ---- feasibility restoration (oracle/ipm.py calls it; oracle/cfz_port.c restore_run is the same iteration) ----------------
    def start_violation(self, X):
        """Worst violation dmin - sep of the rows of stages >= 1 at X (current working set)."""
        P = np.asarray(X).reshape(self.spec.N, self.ns)
        return float(max(0.0, (self.spec.dmin - self.blocks(P)[0][1:]).max()))

    def _resto_objective(self, P, sep, eps):
        b = self.spec.bounds
        phi = 0.0
        for q, c in enumerate(BCOLS):
            k0 = 1 if q < 4 else 0  # the states of stage 0 are the measurement
            el = np.maximum(b[2 * q] + RESTO_BOX_MARGIN - P[k0:, c], 0.0)
            eu = np.maximum(P[k0:, c] - b[2 * q + 1] + RESTO_BOX_MARGIN, 0.0)
            phi += 0.5 * RESTO_RHO_BOX * float((el**2).sum() + (eu**2).sum())
        v = np.maximum(self.spec.dmin + eps - sep[1:], 0.0)
        return phi + 0.5 * RESTO_RHO * float((v**2).sum())

    def restore(self, X, mu, opt, iters_done):
        """IPOPT's restoration phase (paper sec. 3.3) for this NLP: Levenberg-Marquardt on
            min rho/2 sum_{k>=1,r} max(0, dmin + eps - sep_kr)^2 + rho_b/2 sum (box excess)^2   s.t. z_0 = x0, z_{k+1} = F(z_k, u_k)
        with the damping (zeta + lambda) I (zeta = sqrt(mu): IPOPT's proximity weight, taken to the current iterate; lambda adapted
        to the step lengths) and an Armijo line search on the l1 merit with the dynamics defects.  The step comes from the full
        KKT matrix here and from the stage recursion in oracle/cfz_port.c and the kernel.
        Returns (ok, X_new, iterations): ok = every row of stages >= 1 within the goal (a tenth of the violation at entry, or
        half the margin eps), every box with margin, dynamics no worse than at entry; otherwise the caller ends with status 5
        (a stationary point of the violation: locally infeasible)."""
        import scipy.sparse.linalg as spla

        sp_, N, ns = self.spec, self.spec.N, self.ns
        b = sp_.bounds
        Xs = np.asarray(X, float).reshape(N, ns).copy()
        P = Xs[:, :NP].copy()
        zeta = float(np.sqrt(mu))
        eta = lm = 0.0
        vref, ref_it = np.inf, 0
        n, m = NP * N, 5 * N
        rit = 0
        while True:
            if rit > 0:
                self.select(P)
            sep, gr = self.blocks(P)
            F, Fz, Fu = bicycle_rk4_jac(P[:-1, 0:5], P[:-1, 5:7], sp_.dt, sp_.wb, sp_.rk_substeps)
            c = np.concatenate([P[0, 0:5] - self.x0, (F - P[1:, 0:5]).ravel()])
            th_dyn, cv_dyn = float(np.abs(c).sum()), float(np.abs(c).max())
            if rit == 0:
                v0 = float(max(0.0, (sp_.dmin - sep[1:]).max()))
                eps = min(opt.bound_push, v0)
                vgoal = max(0.5 * eps, RESTO_KAPPA * (v0 + eps))
                dgoal = max(opt.constr_viol_tol, cv_dyn)
            g = np.zeros_like(P)
            H = np.zeros((N, NP, NP))
            H[:, range(NP), range(NP)] = zeta + lm
            bmax = 0.0
            for q, col in enumerate(BCOLS):
                k0 = 1 if q < 4 else 0
                el = np.maximum(b[2 * q] + RESTO_BOX_MARGIN - P[:, col], 0.0)
                eu = np.maximum(P[:, col] - b[2 * q + 1] + RESTO_BOX_MARGIN, 0.0)
                el[:k0] = 0.0; eu[:k0] = 0.0
                g[:, col] += RESTO_RHO_BOX * (eu - el)
                H[:, col, col] += RESTO_RHO_BOX * ((el > 0.0).astype(float) + (eu > 0.0))
                bmax = max(bmax, float(el.max()), float(eu.max()))
            v = np.maximum(sp_.dmin + eps - sep, 0.0)
            v[0] = 0.0
            vmax = float(v.max())
            g[:, 0:3] -= RESTO_RHO * np.einsum("kr,kra->ka", v, gr)
            H[:, 0:3, 0:3] += RESTO_RHO * np.einsum("kr,kra,krb->kab", (v > 0.0).astype(float), gr, gr)
            phi = self._resto_objective(P, sep, eps)
            if vmax <= vgoal and bmax <= 0.5 * RESTO_BOX_MARGIN and cv_dyn <= dgoal:
                for q, col in enumerate(BCOLS):
                    P[:, col] = np.minimum(np.maximum(P[:, col], b[2 * q] + 0.5 * RESTO_BOX_MARGIN), b[2 * q + 1] - 0.5 * RESTO_BOX_MARGIN)
                Xs[:, :NP] = P
                return True, Xs.ravel(), rit
            if vmax <= 0.9 * vref or vmax <= vgoal:
                vref, ref_it = vmax, rit
            if rit - ref_it >= RESTO_STALL or rit == RESTO_MAX_ITER or iters_done + rit >= opt.max_iter:
                break
            # Newton step on [[H, J'], [J, 0]]: J = rows of the initial state and of the dynamics
            rows, cols, vals = [np.arange(5)], [np.arange(5)], [np.ones(5)]
            for k in range(N - 1):
                r0 = 5 + 5 * k
                for i in range(5):
                    rows.append(np.full(7, r0 + i)); cols.append(NP * k + np.arange(7)); vals.append(np.concatenate([Fz[k, i], Fu[k, i]]))
                    rows.append(np.array([r0 + i])); cols.append(np.array([NP * (k + 1) + i])); vals.append(np.array([-1.0]))
            J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m, n))
            Hs = sp.block_diag([H[k] for k in range(N)], format="csr")
            K = sp.bmat([[Hs, J.T], [J, None]], format="csc")
            sol = spla.splu(K).solve(-np.concatenate([g.ravel(), c]))
            d, pi = sol[:n].reshape(N, NP), sol[n:]
            dphi = float((g * d).sum())
            pim = float(np.abs(pi).max())
            if eta < 1.1 * pim:
                eta = 2.0 * pim
            M0, dM = phi + eta * th_dyn, dphi - eta * th_dyn
            if not (dM < -1e-10 * (1.0 + abs(M0))):
                break
            alpha, accepted = 1.0, False
            for _ in range(opt.max_backtrack):
                Pt = P + alpha * d
                Ft = bicycle_rk4(Pt[:-1, 0:5], Pt[:-1, 5:7], sp_.dt, sp_.wb, sp_.rk_substeps)
                th_t = float(np.abs(Pt[0, 0:5] - self.x0).sum() + np.abs(Ft - Pt[1:, 0:5]).sum())
                M_t = self._resto_objective(Pt, self.blocks(Pt)[0], eps) + eta * th_t  # working set held
                if np.isfinite(M_t) and M_t <= M0 + RESTO_ARMIJO * alpha * dM:
                    accepted = True
                    break
                alpha *= 0.5
            if not accepted:
                break
            if alpha < 0.2:
                lm = max(4.0 * lm, 1.0)
            elif alpha == 1.0:
                lm *= 0.25
            P = Pt
            rit += 1
        Xs[:, :NP] = P
        return False, Xs.ravel(), rit

    def cold_multipliers(self, X, mu, opt):
        """After a restoration: fresh working set, slacks from the rows (at least half of bound_push), z = mu / distance,
        multipliers of the equality rows zero.  Returns (X, zl, zu, nu)."""
        N, ns = self.spec.N, self.ns
        Xs = np.asarray(X, float).reshape(N, ns).copy()
        self.select(Xs)
        Xs[:, NP:] = np.maximum(self.blocks(Xs)[0] - self.spec.dmin, 0.5 * opt.bound_push)
        x = Xs.ravel()
        hasl, hasu = np.isfinite(self.xl), np.isfinite(self.xu)
        zl = np.where(hasl, mu / np.where(hasl, x - np.where(hasl, self.xl, 0.0), 1.0), 0.0)
        zu = np.where(hasu, mu / np.where(hasu, np.where(hasu, self.xu, 0.0) - x, 1.0), 0.0)
        nu = np.zeros(self.m)
        nu[self.c_blk0:] = -zl.reshape(N, ns)[:, NP:].ravel()
        return x, zl, zu, nu

    # ---- packing between reference layout and X ------------------------------------
    def pack(self, sol):
        """Reference-layout dict -> X.  Only the primal arrays are read; warm-start duals are
        not needed (certificates are recomputed from the poses).  Slacks follow IPOPT's
        s = g(x0) rule."""
        N = self.spec.N
        X = np.zeros((N, self.ns))
        for i, key in enumerate(("x", "y", "psi", "v", "delta", "a", "w")):
            X[:, i] = sol[key]
        self.sel[:] = 0
        self.select(X)
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
        self.select(Xs)
        sep = self.blocks(Xs)[0]
        sol["sep"] = np.minimum(sep[:, 0::2], sep[:, 1::2])  # separation of every block
        cert = {}
        for k in range(N):
            for j in range(self.nb):
                c_ = int(self.sel[k, j])
                va, vb = (c_ >> 2) & 3, c_ & 3
                cert[(k, j)] = (c_ >> 6, (c_ >> 4) & 3, va if sep[k, 2 * j] <= sep[k, 2 * j + 1] else vb)
        for k in range(N):
            psi = Xs[k, 2]
            R = rot(psi)
            def unit(j):  # kind 3: from polygon vertex u to body vertex v
                kind, u, v = cert[(k, j)]
                if kind != 3:
                    return None
                w = Xs[k, 0:2] + R @ self.BV[v] - self.polygon(k, j)[2][u]
                return w / np.hypot(w[0], w[1])

            for j in range(sp_.n_obs):
                lam, mu = certificate_duals(sp_.A_obs[j], self.adj[j], psi, cert[(k, j)], n=unit(j))
                sol["l"][k, 4 * j : 4 * j + 4], sol["m"][k, 4 * j : 4 * j + 4] = lam, mu
            for o in range(sp_.n_nbr):
                lam, mu = certificate_duals(None, None, psi, cert[(k, sp_.n_obs + o)], self.nbr[o, 2, k], n=unit(sp_.n_obs + o))
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

    def hess_gn(self, X, nu=None, shift=False):
        """Gauss-Newton Hessian of the Lagrangian: objective curvature with the (v w)^2 term taken as the
        square of the residual r = v*w (PSD); with `nu`, plus the exact curvature of the separation rows
        sum_r nu_r d2 sep_r / d(x,y,psi)^2 (`row_curvature`).  `shift`: a stage whose curvature would be scaled
        keeps it whole and gets the smallest multiple of the identity on (x, y, psi) that restores the margin.
        """
        N, ns = self.spec.N, self.ns
        P = X.reshape(N, ns)
        wt = self.spec.weights
        base = np.arange(N) * ns
        rows, cols, vals = [], [], []
        self.shift_applied = False  # set below if some stage of THIS evaluation is shifted (oracle/ipm.py carry_shift)

        def add(i, j, v):
            rows.append(base + i), cols.append(base + j), vals.append(np.broadcast_to(v, (N,)))

        add(0, 0, 2 * wt[0]), add(1, 1, 2 * wt[1]), add(2, 2, 2 * wt[2])
        add(4, 4, 2 * wt[5]), add(5, 5, 2 * wt[3])
        v, w = P[:, 3], P[:, 6]
        add(3, 3, 2 * wt[4] * w * w), add(6, 6, 2 * wt[4] * v * v)
        add(3, 6, 2 * wt[4] * v * w), add(6, 3, 2 * wt[4] * v * w)
        if nu is not None:
            C = self.row_curvature(P, np.asarray(nu)[self.c_blk0:].reshape(N, self.nr))
            # Convexity safeguard per stage: C = [[0,0,a],[0,0,b],[a,b,c]] is scaled by th in {1, 1/2, .., 2^-9, 0} until
            # diag(2 w_x, 2 w_y, 2 w_psi) + th C keeps a margin m = 0.2 min(w) (Schur complement on the psi entry).
            q0, q1, q2 = 2 * wt[0], 2 * wt[1], 2 * wt[2]
            m_ = 0.2 * min(wt[0], wt[1], wt[2])
            for k in range(N):
                a_, b_c, c_ = C[k, 0, 2], C[k, 1, 2], C[k, 2, 2]
                full = C[k, 0, 0] != 0.0 or C[k, 1, 1] != 0.0 or C[k, 0, 1] != 0.0  # a vertex-vertex row curves x, y too
                th = 1.0
                for h in range(11):
                    if h == 10:
                        th = 0.0
                        break
                    if not full:
                        if (q2 - m_) + th * c_ - th * th * (a_ * a_ / (q0 - m_) + b_c * b_c / (q1 - m_)) >= 0.0:
                            break
                    else:  # diag(q - m) + th C positive semidefinite: leading principal minors
                        M = np.diag([q0 - m_, q1 - m_, q2 - m_]) + th * C[k]
                        d2 = M[0, 0] * M[1, 1] - M[0, 1] ** 2
                        d3 = M[2, 2] * d2 - (M[0, 2] ** 2 * M[1, 1] - 2.0 * M[0, 2] * M[1, 2] * M[0, 1] + M[1, 2] ** 2 * M[0, 0])
                        if M[0, 0] > 0.0 and d2 > 0.0 and d3 >= 0.0:
                            break
                    th *= 0.5
                if shift and th < 1.0:
                    self.shift_applied = True
                    dl = pose_shift(q0 - m_ + C[k, 0, 0], q1 - m_ + C[k, 1, 1], q2 - m_ + C[k, 2, 2], C[k, 0, 1], a_, b_c)
                    th = 1.0
                    for a in range(3):
                        rows.append(np.array([k * ns + a])), cols.append(np.array([k * ns + a])), vals.append(np.array([dl]))
                Ck = th * C[k]
                for a in range(3):
                    for b_ in range(3):
                        if Ck[a, b_] != 0.0:
                            rows.append(np.array([k * ns + a])), cols.append(np.array([k * ns + b_])), vals.append(np.array([Ck[a, b_]]))
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
        sep, gr = self.blocks(P)
        kk = np.arange(N)
        # The pose of stage 0 is the measurement (pinned by z_0 = x0): its rows are constants -- satisfied, or violated by less than
        # constr_viol_tol (`initial_state_in_collision`) -- and take no part in the iteration: zero residual, zero pose gradient.
        gr[0] = 0.0
        for j in range(self.nr):
            r = self.c_blk0 + self.nr * kk + j
            cvec[r] = sep[:, j] - sp_.dmin - P[:, NP + j]
            cvec[r[0]] = 0.0
            for q in range(3):
                add(r, base + q, gr[:, j, q])
            add(r, base + NP + j, -1.0)
        J = None
        if want_jac:
            J = sp.csr_matrix(
                (np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(self.m, self.n)
            )
        return cvec, J


STATUS_INFEASIBLE_X0 = 4


def initial_state_in_collision(nlp: "MpcNlp", tol):
    """The pose of stage 0 is pinned to the measured state (vehicle_follower.py:194-196); if a
    collision row is violated there by more than tol the NLP has no point within constr_viol_tol (IPOPT
    would end in restoration failure and `step()` in its fallback, :501-524)."""
    sp_ = nlp.spec
    for j in range(nlp.nb):
        A, b, PV = nlp.polygon(0, j)
        sel = select_rows(A, b, PV, nlp.x0[0:2], nlp.x0[2], sp_.g, nlp.BV, 0, vv=sp_.vv_rows)
        sep, _ = rows_for(A, b, PV, nlp.x0[0:2], nlp.x0[2], sp_.g, nlp.BV, sel)
        if sep.min() < sp_.dmin - tol:
            return True
    # ... and so does a measured state outside the boxes on x, y, v, delta by more than tol (stage 0 is bounded like every other
    # stage, :205-240, and pinned to the measurement, :194-199): e.g. a speed above the limit by the measurement noise
    b = sp_.bounds
    for q, c in enumerate((0, 1, 3, 4)):
        if nlp.x0[c] < b[2 * q] - tol or nlp.x0[c] > b[2 * q + 1] + tol:
            return True
    return False


def carry_state(nlp, res):
    """What one converged solve hands to the solve of the next MPC iteration of the same vehicle (`solve_mpc(carry=)`):
    multipliers of the row slacks z [N,nr], working set sel [N,nb], box multipliers zl, zu [N,6] (columns x,y,v,delta,a,w),
    multipliers of the initial-state row pi0 [5] and of the dynamics pi [N-1,5], final barrier parameter mu."""
    N, ns = nlp.spec.N, nlp.ns
    Zl, Zu = res["zl"].reshape(N, ns), res["zu"].reshape(N, ns)
    cols = [0, 1, 3, 4, 5, 6]
    return dict(z=Zl[:, NP:].copy(), sel=nlp.sel.copy(), zl=Zl[:, cols].copy(), zu=Zu[:, cols].copy(),
                pi0=res["nu"][0:5].copy(), pi=res["nu"][5 : nlp.c_blk0].reshape(N - 1, 5).copy(), mu=float(res["mu"]),
                shifted=bool(res.get("shifted", False)))  # some stage's curvature was shifted: the next solve shifts from the start


def warm_from_carry(nlp, X, carry, opt):
    """Initial point of the interior-point iteration from the previous MPC iteration's solution (IPOPT's
    warm_start_init_point in spirit).  The horizon has moved on by one stage: new stage k takes old stage
    min(k+1, N-1).  X: packed primal warm start with fresh working set and slacks sigma = sep - dmin.
      mu0   = clip(carry mu, mu_floor, mu_init)
      boxes : p clipped to [lo + warm_push, hi - warm_push]; z = max(carried z, mu0 / (hi - lo))
      rows  : a row whose (face, vertex) identity exists in the carried working set of that stage keeps its
              multiplier z and gets sigma = max(sep - dmin, mu0 / z, warm_push); any other row starts as in a cold
              solve, sigma = max(sep - dmin, bound_push), z = mu0 / sigma; nu = -z
      pi    : initial-state row <- old first dynamics row; dynamics row k <- old row min(k+1, N-2)
    Returns (X, dict(zl, zu, nu, mu)) for ipm.solve(warm=)."""
    N, ns, nr, nb = nlp.spec.N, nlp.ns, nlp.nr, nlp.nb
    mu_floor = min(opt.tol, opt.compl_inf_tol) / (opt.kappa_eps + 1.0)
    mu0 = min(max(carry["mu"], mu_floor), opt.mu_init)
    X = X.reshape(N, ns).copy()
    Zl, Zu = np.zeros((N, ns)), np.zeros((N, ns))
    NU = np.zeros((N, nr))
    xl, xu = nlp.xl.reshape(N, ns), nlp.xu.reshape(N, ns)
    cols = [0, 1, 3, 4, 5, 6]
    for k in range(N):
        ko = min(k + 1, N - 1)
        for q, c in enumerate(cols):
            lo, hi = xl[k, c], xu[k, c]
            X[k, c] = min(max(X[k, c], lo + opt.warm_push), hi - opt.warm_push)
            Zl[k, c] = max(carry["zl"][ko, q], mu0 / (hi - lo))
            Zu[k, c] = max(carry["zu"][ko, q], mu0 / (hi - lo))
        for j in range(nb):
            so, sn = int(carry["sel"][ko, j]), int(nlp.sel[k, j])
            for r in range(2):
                vn = (sn >> 2) & 3 if r == 0 else sn & 3
                z = 0.0
                if (so >> 4) == (sn >> 4):
                    if ((so >> 2) & 3) == vn:
                        z = carry["z"][ko, 2 * j]
                    elif (so & 3) == vn:
                        z = carry["z"][ko, 2 * j + 1]
                gap = X[k, NP + 2 * j + r]
                if z > 0.0:
                    sg = max(gap, mu0 / z, opt.warm_push)
                else:
                    sg = max(gap, opt.bound_push)
                    z = mu0 / sg
                X[k, NP + 2 * j + r], Zl[k, NP + 2 * j + r], NU[k, 2 * j + r] = sg, z, -z
    nu = np.zeros(nlp.m)
    nu[0:5] = carry["pi"][0]
    for k in range(N - 1):
        nu[5 + 5 * k : 10 + 5 * k] = carry["pi"][min(k + 1, N - 2)]
    nu[nlp.c_blk0 :] = NU.ravel()
    return X.ravel(), dict(zl=Zl.ravel(), zu=Zu.ravel(), nu=nu, mu=mu0, shift_hint=bool(carry.get("shifted", False)))


def solve_mpc(spec: MpcSpec, x0, ref, nbr, zu, opt=None, trace=None, carry=None):
    """One MPC-step solve by the full-KKT oracle.  zu [7,N] warm start (rows x,y,psi,v,delta,a,w).
    carry: `carry` entry of the result of the previous MPC iteration of the same vehicle (None = cold multipliers).
    Returns dict(zu [7,N], status, iters, f, sep [N,n_blk], sol (reference-layout dict), carry (None unless converged))."""
    from . import ipm

    opt = opt or ipm.IpmOptions()
    nlp = MpcNlp(spec, x0, ref, nbr)
    zu = np.asarray(zu, float)
    if initial_state_in_collision(nlp, opt.constr_viol_tol):
        return dict(zu=zu.copy(), status=STATUS_INFEASIBLE_X0, iters=0, f=0.0, sep=None, sol=None, carry=None)
    warm = dict(zip(("x", "y", "psi", "v", "delta", "a", "w"), zu))
    X0 = nlp.pack(warm)
    if carry is None:
        res = ipm.solve(nlp, X0, opt, trace=trace)
    else:
        X0, w = warm_from_carry(nlp, X0, carry, opt)
        res = ipm.solve(nlp, X0, opt, trace=trace, warm=w)
    sol = nlp.unpack(res["X"])
    out = np.stack([sol[k] for k in ("x", "y", "psi", "v", "delta", "a", "w")])
    return dict(zu=out, status=res["status"], iters=res["iters"], f=res["f"], sep=sol["sep"], sol=sol,
                carry=carry_state(nlp, res) if res["status"] == 0 else None)
