"""Generates tests/golden/colloc_independent_vv.npz (run from the repo root: `python tests/golden/make_independent_colloc_vv.py`,
about 4 min): single-vehicle collocation plans at the reference's size whose optimum holds a CORNER-TO-CORNER contact -- a
collocation point where the closest features of the body and of an obstacle are two vertices and their distance equals dmin,
with a positive multiplier.  There the reference's OBCA rows (vehicle.py:523-541: any unit direction A'lambda separates) allow
more than a face-normal certificate does, so these instances tell the two feasible sets apart.

Instance = a plan of the synthetic strategy (tests/golden/colloc_independent.npz) plus a SEVENTH obstacle, a square pillar of
diagonal 1.2 m standing on one vertex: that vertex is put `intr` metres inside the clearance of body corner `v` at collocation
point `q` of the unobstructed optimum, on the bisector of the corner's normal cone (the outside front corner of a left turn
sweeps past it), the square turned by TILT against that bisector.  Both solvers start from the unobstructed optimum ("the pillar appears after the plan was made").
Solved INDEPENDENTLY of the planning kernel by oracle/independent_colloc.py (polygon distances, own derivatives, SuperLU on the
full KKT matrix) to tol 1e-8.  Stored per instance: pillar (A, b), guess, optimal trajectory, dt, cost, and `value`: the cost
corrected to first order for the residual of the equality rows, f - lambda'c with lambda from the solver-free KKT certificate
(the independent solver stops with rows at 3e-8 and multipliers up to 1e4, which alone moves the cost by 1e-5 of its value)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (agent, body corner, collocation point of the unobstructed optimum, intrusion into the clearance [m])
INSTANCES = {"vehicle_1_pillar": ("vehicle_1", 3, 130, 0.2), "vehicle_2_pillar": ("vehicle_2", 3, 60, 0.1), "vehicle_3_pillar": ("vehicle_3", 3, 210, 0.1)}
DMIN = 0.05
G_BODY = np.array([3.3, 0.9, 0.6, 0.9])


TILT = 0.15  # rad: the pillar's axis against the bisector (an exactly symmetric pillar ties two certificates at the guess, and
#              which one a solver starts with is then decided by rounding)


def pillar(pose, v, intr, half_diag=0.6, g=G_BODY, tilt=TILT):
    """(A [4, 2], b [4]) of the square whose nearest vertex sits DMIN - intr from body corner v of `pose`, outside along the
    bisector of the corner's normal cone; the square's diagonal through that vertex is turned by `tilt` against the bisector."""
    BV = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])
    c, s = np.cos(pose[2]), np.sin(pose[2])
    R = np.array([[c, -s], [s, c]])
    n = R @ (np.sign(BV[v]) / np.sqrt(2.0))
    vertex = pose[:2] + R @ BV[v] + n * (DMIN - intr)
    ct, st = np.cos(tilt), np.sin(tilt)
    n = np.array([[ct, -st], [st, ct]]) @ n
    t = np.array([-n[1], n[0]])
    m = vertex + n * half_diag
    A = np.array([n + t, n - t, -n + t, -n - t]) / np.sqrt(2.0)
    return A, A @ m + half_diag / np.sqrt(2.0)


def instance(name):
    """(agent, tube, path, final heading, A_obs [7, 4, 2], b_obs [7, 4], guess = the unobstructed optimum as points + dt)."""
    from make_independent_colloc import problem

    agent, v, q, intr = INSTANCES[name]
    f = np.load(os.path.join(HERE, "colloc_independent.npz"))
    T, dt = f[f"{agent}_traj"].reshape(-1, 7), float(f[f"{agent}_dt"])
    tube, p, fh, sp = problem(agent)
    A7, b7 = pillar(T[q, :3], v, intr)
    return agent, tube, p, fh, np.concatenate([sp.A_obs, [A7]]), np.concatenate([sp.b_obs, [b7]]), np.append(T.ravel(), dt)


def kkt_certificate(g, z):
    """Solver-free certificate: multipliers (bounded least squares: equality rows free, active inequality rows and bounds >= 0)
    that combine the active gradients into the cost gradient.  Returns (relative residual, lambda_eq, lambda_active, active rows)."""
    from scipy.optimize import lsq_linear
    from threadpoolctl import threadpool_limits

    act = np.nonzero(g.ineq(z) < 1e-6)[0]
    lo = np.array([b[0] if b[0] is not None else -np.inf for b in g.bounds()])
    hi = np.array([b[1] if b[1] is not None else np.inf for b in g.bounds()])
    at_lo, at_hi = np.nonzero(z - lo < 1e-6)[0], np.nonzero(hi - z < 1e-6)[0]
    Je, Ji = g.eq_jac(z), g.ineq_jac(z)[act]
    Eb = np.zeros((len(at_lo) + len(at_hi), g.n))
    Eb[np.arange(len(at_lo)), at_lo] = 1.0
    Eb[len(at_lo) + np.arange(len(at_hi)), at_hi] = -1.0
    A = np.vstack([Je, Ji, Eb]).T
    lb = np.concatenate([np.full(len(Je), -np.inf), np.zeros(len(Ji) + len(Eb))])
    with threadpool_limits(limits=1):  # BVLS makes thousands of small BLAS calls: threads only contend
        r = lsq_linear(A, g.cost_grad(z), bounds=(lb, np.full(A.shape[1], np.inf)), method="bvls", max_iter=800)
    res = np.abs(A @ r.x - g.cost_grad(z)).max() / np.abs(g.cost_grad(z)).max()
    return float(res), r.x[: len(Je)], r.x[len(Je): len(Je) + len(Ji)], act


def vertex_contacts(g, z, lam_act, act):
    """[(point, obstacle, multiplier)] of the ACTIVE distance rows whose closest features are two vertices."""
    from oracle.mpc_nlp import body_vertices, closest_vertex_pair

    P, _ = g.split(z)
    poses, no = P.reshape(-1, 7)[:, :3], len(g.obs)
    n_tube = 8 * g.n_chk
    out = []
    for r, lam in zip(act, lam_act):
        if r < n_tube:
            continue
        q, j = divmod(int(r) - n_tube, no)
        if closest_vertex_pair(g.obs[j], poses[q, :2], poses[q, 2], g.g, body_vertices(g.g)) is not None:
            out.append((q, j, float(lam)))
    return out


if __name__ == "__main__":
    from oracle.independent_colloc import GeometricColloc, solve_ipm

    out = {}
    for name in INSTANCES:
        agent, tube, p, fh, A_obs, b_obs, X0 = instance(name)
        g = GeometricColloc(p[0], tube, A_obs, b_obs, N_per_set=5, final_heading=fh, dmin=DMIN)
        t0 = time.time()
        from oracle import ipm

        r = solve_ipm(g, X0[:-1].reshape(-1, 7), X0[-1], opt=ipm.IpmOptions(max_iter=800, hessian="exact", reg_dual=1e-9, stall_iters=0, err_stall_iters=0, tol=1e-8,
                                                                               constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6))
        z = np.append(r["traj"].ravel(), r["dt"])
        res, lam_eq, lam_act, act = kkt_certificate(g, z)
        vc = vertex_contacts(g, z, lam_act, act)
        value = r["cost"] - float(lam_eq @ g.eq(z))
        print(name, {k: v for k, v in r.items() if k != "traj"}, "certificate %.1e" % res, "vertex contacts", vc, "value %.9f" % value,
              "%.0f s" % (time.time() - t0), flush=True)
        # status 2 = the line search ran out at the rounding floor of the merit function (as for colloc_independent.npz)
        # (1 = the iteration limit, reached by vehicle_1_pillar with the rows at 2e-9: what makes the point a fixture is the certificate)
        assert r["status"] in (0, 1, 2) and r["eq"] < 5e-8 and r["ineq"] > -1e-8 and res < 1e-8
        assert any(lam > 0.1 for _, _, lam in vc), "no active corner-to-corner contact at the optimum"
        out.update({f"{name}_A": A_obs[6], f"{name}_b": b_obs[6], f"{name}_guess": X0, f"{name}_traj": r["traj"], f"{name}_dt": r["dt"],
                    f"{name}_cost": r["cost"], f"{name}_value": value, f"{name}_iters": r["iters"], f"{name}_status": r["status"],
                    f"{name}_contacts": np.array(vc, float).reshape(-1, 3)})
    np.savez_compressed(os.path.join(HERE, "colloc_independent_vv.npz"), **out)
