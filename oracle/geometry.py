"""Oracle (test infrastructure): distance of two convex polygons as a tiny convex QP over their vertex weights -- no case
analysis (face-vertex, vertex-vertex), so it is independent of the closed forms the kernels use.  The reference obtains
the same number as the optimum of its dual warm-start problems (confrez/control/vehicle.py:233-296,
multi_vehicle_planner.py:208-341: maximise d over the OBCA duals = the Euclidean distance of the two sets)."""
import numpy as np
from scipy.optimize import minimize


def polygon_distance(P, Q):
    """min |p - q| over p in conv(P), q in conv(Q); P [np, 2], Q [nq, 2] vertex arrays.  Returns (dist, p, q)."""
    P, Q = np.asarray(P, float), np.asarray(Q, float)
    n, m = len(P), len(Q)

    def f(w):
        r = w[:n] @ P - w[n:] @ Q
        return float(r @ r)

    def g(w):
        r = w[:n] @ P - w[n:] @ Q
        return np.concatenate([2 * P @ r, -2 * Q @ r])

    best = None
    for i in range(n):  # a few starts: the QP is convex, but SLSQP can stall on a degenerate face
        w0 = np.zeros(n + m)
        w0[i] = 1.0
        w0[n + int(np.argmin(np.linalg.norm(Q - P[i], axis=1)))] = 1.0
        r = minimize(f, w0, jac=g, method="SLSQP", bounds=[(0, 1)] * (n + m),
                     constraints=[dict(type="eq", fun=lambda w: w[:n].sum() - 1.0, jac=lambda w: np.r_[np.ones(n), np.zeros(m)]),
                                  dict(type="eq", fun=lambda w: w[n:].sum() - 1.0, jac=lambda w: np.r_[np.zeros(n), np.ones(m)])],
                     options=dict(ftol=1e-16, maxiter=200))
        if best is None or r.fun < best.fun:
            best = r
    w = best.x
    return float(np.sqrt(max(best.fun, 0.0))), w[:n] @ P, w[n:] @ Q


def body_polygon_world(pose, g):
    """Vertices of the vehicle body {G p <= g} at pose (x, y, psi), counter-clockwise."""
    x, y, psi = pose
    c, s = np.cos(psi), np.sin(psi)
    B = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])
    return np.array([x, y]) + B @ np.array([[c, s], [-s, c]])
