"""The oracle pinned against known answers, its own two implementations and the committed goldens (CPU)."""
import numpy as np
import pytest

from oracle import ipm, port
from oracle.dynamics import bicycle_rk4, bicycle_rk4_jac, plant_step
from oracle.mpc_nlp import (MpcNlp, MpcSpec, body_vertices, polytope_vertices, reference_residuals, rows_for,
                            select_rows, solve_mpc)


# ---- dynamics: closed-form known answers (dynamic_model.py:5-58) ------------------------------------
def test_rk4_straight_line_and_arc():
    wb, dt = 2.5, 0.1
    z = bicycle_rk4(np.array([1.0, 2.0, 0.3, 1.5, 0.0]), np.array([0.5, 0.0]), dt, wb)
    s = 1.5 * dt + 0.5 * 0.5 * dt**2
    assert np.allclose(z, [1 + s * np.cos(0.3), 2 + s * np.sin(0.3), 0.3, 1.55, 0.0], atol=1e-12)
    v, de = 2.0, 0.4  # constant speed and steering: circle of radius wb / tan(delta)
    r = wb / np.tan(de)
    z = np.array([0.0, 0.0, 0.0, v, de])
    for _ in range(10):
        z = bicycle_rk4(z, np.zeros(2), dt, wb)
    th = v * 1.0 / r
    assert np.allclose(z[:3], [r * np.sin(th), r * (1 - np.cos(th)), th], atol=1e-9)
    assert np.allclose(plant_step(np.array([0, 0, 0, v, de]), np.zeros(2), 1.0, wb)[:3], z[:3], atol=1e-7)


def test_rk4_jacobian_fd():
    rng = np.random.default_rng(0)
    z, u = rng.normal(size=(6, 5)) * [1, 1, 1, 1, 0.3], rng.normal(size=(6, 2))
    F, Fz, Fu = bicycle_rk4_jac(z, u, 0.1, 2.5)
    assert np.allclose(F, bicycle_rk4(z, u, 0.1, 2.5))
    eps = 1e-6
    for i in range(5):
        e = np.zeros(5); e[i] = eps
        assert np.allclose((bicycle_rk4(z + e, u, 0.1, 2.5) - bicycle_rk4(z - e, u, 0.1, 2.5)) / (2 * eps), Fz[:, :, i], atol=1e-8)
    for i in range(2):
        e = np.zeros(2); e[i] = eps
        assert np.allclose((bicycle_rk4(z, u + e, 0.1, 2.5) - bicycle_rk4(z, u - e, 0.1, 2.5)) / (2 * eps), Fu[:, :, i], atol=1e-8)


# ---- separation certificates: geometric known answers -----------------------------------------------
def test_rows_equal_geometric_distance_for_aligned_boxes():
    """dual_ws's optimum (vehicle.py:233-296) is the rectangle-box distance; for face-to-face
    configurations the closed-form rows must reproduce it exactly."""
    g = np.array([3.3, 0.9, 0.6, 0.9])
    A = np.array([[-1.0, 0], [0, 1], [0, -1], [1, 0]]); b = np.array([-2.85, 13.75, -7.5, 14.65])
    PV, _ = polytope_vertices(A, b)
    BV = body_vertices(g)
    for (x, y, psi, want) in [(8.0, 16.25, 0.0, 16.25 - 0.9 - 13.75), (20.0, 10.0, 0.0, 20.0 - 0.6 - 14.65),
                              (8.0, 15.0, np.pi, 15.0 - 0.9 - 13.75), (18.0, 10.0, np.pi / 2, 18.0 - 0.9 - 14.65)]:
        sel = select_rows(A, b, PV, np.array([x, y]), psi, g, BV, 0)
        sep, _ = rows_for(A, b, PV, np.array([x, y]), psi, g, BV, sel)
        assert np.isclose(sep.min(), want, atol=1e-12)
    # overlapping: negative = penetration depth along the best face
    sel = select_rows(A, b, PV, np.array([8.0, 14.0]), 0.0, g, BV, 0)
    assert np.isclose(rows_for(A, b, PV, np.array([8.0, 14.0]), 0.0, g, BV, sel)[0].min(), 14.0 - 0.9 - 13.75)


def test_nlp_jacobian_fd(golden, ospec):
    b = 3
    nlp = MpcNlp(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b])
    X = nlp.pack(dict(zip(("x", "y", "psi", "v", "delta", "a", "w"), golden["zu"][b])))
    rng = np.random.default_rng(1)
    X = ipm.push_to_interior(X + 0.01 * rng.standard_normal(X.shape), nlp.xl, nlp.xu, ipm.IpmOptions())
    J, g = nlp.jac(X).toarray(), nlp.grad(X)
    eps = 1e-6
    for i in rng.choice(nlp.n, 60, replace=False):
        if 7 <= i < nlp.ns:  # a slack of stage 0: its rows are constants taken out of the iteration (zero residual, zero pose gradient),
            continue         # the -1 of the slack stays in the Newton system so that the slack's step is zero: not a derivative
        e = np.zeros(nlp.n); e[i] = eps
        assert np.allclose((nlp.cons(X + e) - nlp.cons(X - e)) / (2 * eps), J[:, i], atol=2e-6)
        assert np.isclose((nlp.f(X + e) - nlp.f(X - e)) / (2 * eps), g[i], atol=2e-5)


# ---- the two oracle implementations against each other and the committed fixtures ----------------------
def test_goldens_reproduce_and_c_port_matches(golden, ospec):
    meta = golden["meta"]
    for b in range(len(golden["x0"])):
        r = port.solve(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b].T)
        assert r["status"] == int(meta[b, 0]) and r["iters"] == int(meta[b, 1])
        if r["status"] == 0:
            assert np.abs(r["p"].T - golden["sol"][b]).max() < 1e-9
            assert np.isclose(r["f"], meta[b, 2], rtol=1e-10)
    for b in (0, 7, 11, 13):  # full-KKT numpy solver regenerates the stored vectors
        r = solve_mpc(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b])
        assert r["status"] == int(meta[b, 0]) and r["iters"] == int(meta[b, 1])
        if r["status"] == 0:
            assert np.abs(r["zu"] - golden["sol"][b]).max() < 1e-9


def test_solutions_satisfy_the_reference_formulation(golden, ospec):
    """Solver-independent acceptance: every constraint row as the reference writes it
    (vehicle_follower.py:194-352), duals rebuilt from the certificates."""
    for b in (1, 5, 7, 12):
        r = solve_mpc(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b])
        assert r["status"] == 0
        res = reference_residuals(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], r["sol"])
        assert res["eq"] < 1e-2 and res["ineq"] < 1e-2 and res["bound"] == 0.0
        assert np.isclose(res["cost"], r["f"], rtol=1e-12)
        warm = dict(zip(("x", "y", "psi", "v", "delta", "a", "w"), golden["zu"][b]))
        nlp = MpcNlp(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b])
        assert r["f"] <= nlp.f(nlp.pack(warm)) + 50.0  # not worse than the warm start by a wide margin


def test_tight_tolerance_agrees_with_reference_tolerance(golden, ospec):
    """Solving to 1e-6 moves the answer obtained at the reference's tolerance (1e-2) by < 1e-4 in the poses
    and < 1e-2 in the inputs: far inside the parity band claimed against CasADi/IPOPT (5e-2 / 1e-1)."""
    for b in (1, 5, 7, 8):
        a = solve_mpc(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b])
        t = solve_mpc(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b],
                      ipm.IpmOptions(tol=1e-6, constr_viol_tol=1e-6, compl_inf_tol=1e-6))
        assert a["status"] == 0 and t["status"] == 0
        assert np.abs(a["zu"][:3] - t["zu"][:3]).max() < 1e-4 and np.abs(a["zu"][3:] - t["zu"][3:]).max() < 1e-2


def test_infeasible_initial_state_is_reported(golden, ospec):
    b = int(np.flatnonzero(golden["meta"][:, 0] == 4)[0])
    r = solve_mpc(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b])
    assert r["status"] == 4 and r["iters"] == 0
