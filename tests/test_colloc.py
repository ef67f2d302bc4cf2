"""The single-vehicle collocation plan (reference vehicle.py:360-661): the solver source on the CPU against the numpy
statement (values), finite differences (derivatives) and the reference's own rows (solutions); the HIP build through the
C ABI and the Python surface on the GPU."""
import os
import tempfile

import numpy as np
import pytest
from scipy.interpolate import interp1d

from conflict_rez_amd import scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
from oracle import ipm
from oracle.colloc_nlp import CollocNlp, radau_tables, reference_residuals
from oracle.plan_nlp import StateWsNlp, speed_guess

COLLOC_OPT = dict(max_iter=400, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1)  # vehicle.py:650-651


@pytest.fixture(scope="module")
def plans():
    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    return {a: ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]], paths[a]) for a in sorted(hist)}


def warm_start(tube, p, fh):
    """state_ws through the planning solver source, as `plan_single_path` does before the collocation solve."""
    import plan_emu_binding as pe

    ws = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
    r = pe.solve(ws, ws.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, ws.dt)), ipm.IpmOptions(max_iter=500, hessian="exact", reg_dual=1e-9, stall_iters=0, mu_init=0.1))
    assert r["status"] == 0
    return ws.unpack(r["X"])


def colloc_guess(nlp, z):
    """interp_ws_for_collocation (vehicle.py:298-358) + dt0 = t_end / N (:388)."""
    N = nlp.N[0]
    t_i = np.concatenate([i + nlp.tau for i in range(N)]) / N * z["t"][-1]
    return nlp.pack({k: interp1d(z["t"], z[k])(t_i) for k in ("x", "y", "psi", "v", "delta", "a", "w")}, z["t"][-1] / N)


def test_collocation_tables():
    from conflict_rez_amd.control.vehicle import Vehicle, radau_points

    tau, A, B, D = radau_tables(5)
    assert np.allclose(tau[1:], radau_points(5), atol=1e-14)
    A2, B2, D2 = Vehicle.collocation_coefficients(None, 5)
    assert np.allclose(A, A2) and np.allclose(B, B2) and np.allclose(D, D2)


def _corner_pose(corner, psi, v, off, g=(3.3, 0.9, 0.6, 0.9)):
    """Pose whose body vertex v sits at corner + off: with off pointing out of the corner's quadrant and the corner inside the body
    vertex's normal cone the two vertices are each other's closest feature (a kind-3 block)."""
    bv = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])[v]
    c, s_ = np.cos(psi), np.sin(psi)
    return np.array([corner[0] + off[0] - (c * bv[0] - s_ * bv[1]), corner[1] + off[1] - (s_ * bv[0] + c * bv[1]), psi])


@pytest.mark.parametrize("vv", [False, True])
def test_colloc_source_values_and_derivatives(plans, vv):
    """Objective and constraints equal the numpy statement; gradient, J'nu, the assembled Jacobian and the assembled
    Hessian of the Lagrangian (dt border included) equal central differences; the permuted system is banded.
    vv: with vertex-vertex rows in the working set (three points posed corner to corner with an obstacle)."""
    import colloc_emu_binding as ce

    tube, p = plans["vehicle_0"]
    sp = scenarios.parking_lot_spec()
    nlp = CollocNlp(p[0], tube[:3], sp.A_obs, sp.b_obs, N_per_set=2, final_heading=0.3, vv=vv)
    opt = ipm.IpmOptions(**COLLOC_OPT)
    d = ce.dims(nlp, opt)
    assert (d["n"], d["m"], d["iDt"], d["sO"], d["sT"], d["rR"], d["rF"], d["rP"]) == (nlp.n, nlp.m, nlp.iDt, nlp.sO, nlp.sT, nlp.rR, nlp.rF, nlp.rP)
    rng = np.random.default_rng(0)
    X = np.zeros(nlp.n)
    P = X[: nlp.iDt].reshape(nlp.np, 7)
    P[:, :3] = p[np.linspace(0, 60, nlp.np).astype(int)] + 0.05 * rng.standard_normal((nlp.np, 3))
    P[:, 3:] = 0.3 * rng.standard_normal((nlp.np, 4))
    if vv:  # rear-right, rear-left and front-left body corners against corners of obstacles 0, 3 and 4
        P[3, :3] = _corner_pose((14.65, 13.75), 0.3, 2, (0.3, 0.25))
        P[7, :3] = _corner_pose((14.65, 21.25), -0.4, 1, (0.2, -0.3))
        P[10, :3] = _corner_pose((17.85, 21.25), 2.6, 0, (-0.15, -0.2))
    X[nlp.iDt] = 0.7
    X[nlp.sO :] = rng.uniform(0.1, 1.0, nlp.n - nlp.sO)
    nu = rng.standard_normal(nlp.m)
    sel = ce.select(nlp, opt, X)
    # all kinds of certificate are exercised
    assert (sel == nlp.select(X).ravel()).all() and set((sel.ravel() >> 6).tolist()) == ({1, 2, 3} if vv else {1, 2})
    assert not vv or (sel.ravel() >> 6 == 3).sum() >= 3
    f, c, g, jt = ce.evaluate(nlp, opt, sel, X, nu)
    assert abs(f - nlp.f(X)) < 1e-12 and np.abs(c - nlp.cons(X, sel)).max() < 1e-12
    h, n = 1e-6, nlp.n
    gfd, J, H = np.zeros(n), np.zeros((nlp.m, n)), np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n)
        e[i] = h
        fp, cp, gp, jp = ce.evaluate(nlp, opt, sel, X + e, nu)
        fm, cm, gm, jm = ce.evaluate(nlp, opt, sel, X - e, nu)
        gfd[i], J[:, i], H[:, i] = (fp - fm) / (2 * h), (cp - cm) / (2 * h), (gp + jp - gm - jm) / (2 * h)
    assert np.abs(g - gfd).max() < 1e-7 and np.abs(jt - J.T @ nu).max() < 1e-6
    # the assembled matrix is the Schur complement of the full KKT matrix onto everything but the collision pairs
    sig = np.zeros(n)
    sig[nlp.sO :] = rng.uniform(0.5, 50.0, n - nlp.sO)
    K, bw = ce.kkt(nlp, opt, sel, X, nu, sig=sig)
    M = np.block([[H + np.diag(sig + opt.reg_primal), J.T], [J, -opt.reg_dual * np.eye(nlp.m)]])
    E = np.r_[nlp.sO : nlp.sT, n + nlp.rR : n + nlp.rT]
    R = np.setdiff1d(np.arange(n + nlp.m), E)
    schur = M[np.ix_(R, R)] - M[np.ix_(R, E)] @ np.linalg.solve(M[np.ix_(E, E)], M[np.ix_(E, R)])
    assert np.abs(K[np.ix_(R, R)] - schur).max() < 2e-6 * max(1.0, np.abs(schur).max()) and np.abs(K - K.T).max() == 0.0
    assert np.abs(K[E]).max() == 0.0 and np.abs(K[:, E]).max() == 0.0
    free = CollocNlp(p[0], tube[:3], sp.A_obs[:0], sp.b_obs[:0], N_per_set=2)  # no obstacles, free terminal heading
    Xf, nuf = np.append(X[: free.iDt + 1], X[nlp.sT :]), rng.standard_normal(free.m)
    Kf, bwf = ce.kkt(free, opt, np.zeros(0, np.uint8), Xf, nuf)
    assert bwf <= 51 and Kf.shape[0] == free.n + free.m


@pytest.mark.parametrize("agent", ["vehicle_1", "vehicle_0"])
def test_structured_elimination_equals_the_band_elimination(plans, agent):
    """cfz_jstruct.inl's single-vehicle scheme on the CPU build: the plan with its Newton system eliminated interval by interval (CSpec::no_prox
    bit 2) takes the iterates of the band elimination -- equal status and iteration count, solution to 1e-8: another elimination order
    of the same matrix (30 and 50 intervals; the 50-interval plan is the one whose batch time the bench's configs[1] is)."""
    import colloc_emu_binding as ce

    sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tube, p = plans[agent]
    fh = float(p[-1, 2])
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
    X0 = colloc_guess(nlp, warm_start(tube, p, fh))
    band = ce.solve(nlp, X0, ipm.IpmOptions(**COLLOC_OPT))
    for bits in (4, 12):  # bit 2: cfz_jstruct.inl's scheme for one vehicle (the product's default: tube rows condensed, 16-row separators); bit 3 chose it over round 4's scheme until round 6 and is ignored now
        opt = ipm.IpmOptions(**COLLOC_OPT)
        opt.no_prox = bits
        st = ce.solve(nlp, X0, opt)
        assert (st["status"], st["iters"]) == (band["status"], band["iters"]) == (0, band["iters"]), bits
        assert np.abs(st["X"] - band["X"]).max() < 1e-8 and abs(st["f"] - band["f"]) < 1e-9 * band["f"], bits


def test_kkt_matrix_has_the_interval_structure(plans):
    """What next round's elimination relies on (tools/colloc_condense_study.py, docs/notebook.md): in the matrix the kernel assembles,
    the interior of a Radau interval (points 1..5 and the interval's 30 ODE rows: 65 unknowns) couples only to its own separator (start
    point, continuity rows, tube slacks / rows, initial rows) and to the next one; eliminating the interiors independently (dense,
    pivoted) and then the separator system with the dt border gives the solution of the whole system (1e-7 relative at condition
    numbers of 1e11) -- at the guess and at an iterate, with barrier terms on every box."""
    import importlib.util

    import colloc_emu_binding as ce

    spec_ = importlib.util.spec_from_file_location("colloc_condense_study", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "colloc_condense_study.py"))
    st = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(st)
    sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tube, p = plans["vehicle_1"]
    fh = float(p[-1, 2])
    nlp = CollocNlp(p[0], tube[:4], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=None)  # 15 intervals
    X0 = colloc_guess(nlp, warm_start(tube[:4], p[: 30 * 3 + 1], None))
    opt = ipm.IpmOptions(**COLLOC_OPT)
    rng = np.random.default_rng(1)
    N = nlp.N[0]
    for X, mu in ((X0, 0.1), (ce.solve(nlp, X0, ipm.IpmOptions(**{**COLLOC_OPT, "max_iter": 6}))["X"], 2e-2)):
        Xf = np.zeros(nlp.n)
        Xf[: len(X)] = X
        sel = ce.select(nlp, opt, Xf)
        Xf[nlp.sO :] = np.maximum(Xf[nlp.sO :], 1e-2)
        K, bw = ce.kkt(nlp, opt, sel, Xf, rng.standard_normal(nlp.m), sig=st.central_sigma(nlp, Xf, mu))
        grp, live = st.interval_groups(nlp, K)
        assert st.pattern_violations(K, grp, live) == 0 and (grp[live] != -1).all()
        assert sorted(set((grp[live][grp[live] >= 0]).tolist())) == list(range(2 * N + 1))
        sizes = [int((grp[live] == 2 * i + 1).sum()) for i in range(N)]
        assert sizes[:-1] == [65] * (N - 1) and sizes[-1] == 65 + 16  # (the last interior holds the end point's tube slacks and rows)
        rhs = rng.standard_normal(K.shape[0])
        rhs[grp == -1] = 0.0
        ref = np.zeros(K.shape[0])
        ref[live] = np.linalg.solve(K[np.ix_(live, live)], rhs[live])
        sol, info = st.structured_solve(K, grp, live, rhs, N, nlp.iDt)
        assert np.abs(sol - ref).max() < 1e-7 * np.abs(ref).max() and info["separator_unknowns"] < 25 * (N + 1)
        # ... and the separator system itself is a block recursion over the intervals with dt as a second right-hand side
        xs, _ = st.separator_recursion(info["S"], info["sepidx"], grp, N, info["rs"])
        xd = np.linalg.solve(info["S"], info["rs"])
        assert np.abs(xs - xd).max() < 1e-9 * np.abs(xd).max()
        # the partition to build on the GPU: interiors of exactly 64 unknowns (a row per lane), separators of at most 31
        g64, _ = st.interval_groups(nlp, K, lanes64=True)
        assert st.pattern_violations(K, g64, live) == 0 and all((g64[live] == 2 * i + 1).sum() == 64 for i in range(N))
        assert max((g64[live] == 2 * i).sum() for i in range(N + 1)) <= 31
        sol64, info64 = st.structured_solve(K, g64, live, rhs, N, nlp.iDt)
        xs64, _ = st.separator_recursion(info64["S"], info64["sepidx"], g64, N, info64["rs"])
        assert np.abs(sol64 - ref).max() < 1e-7 * np.abs(ref).max()
        assert np.abs(xs64 - np.linalg.solve(info64["S"], info64["rs"])).max() < 1e-8 * np.abs(xs64).max()


@pytest.mark.parametrize("agent", ["vehicle_1", "vehicle_3"])
def test_colloc_source_solves_reference_problem(plans, agent):
    """Full size (N_per_set = 5, K = 5, six obstacles), from the state_ws warm start: converges at the reference's
    tolerances, and the plan with the rebuilt OBCA duals satisfies the reference's own rows."""
    import colloc_emu_binding as ce

    tube, p = plans[agent]
    fh = float(p[-1, 2])
    sp = scenarios.parking_lot_spec()
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
    X0 = colloc_guess(nlp, warm_start(tube, p, fh))
    res = ce.solve(nlp, X0, ipm.IpmOptions(**COLLOC_OPT))
    assert res["status"] == 0 and res["iters"] < 100
    sol = nlp.unpack(res["X"])
    rr = reference_residuals(nlp, sol)
    assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9 and abs(rr["cost"] - res["f"]) < 1e-9 * max(1, res["f"])
    T_end = sol["dt"] * nlp.N[0]
    assert 2.0 < T_end < 0.1 * (len(p) - 1) * 1.5 and abs(sol["psi"][-1, -1] - fh) < 1e-2
    # faster than the warm start's fixed timetable would allow at these input costs, and it actually moved
    assert np.hypot(sol["x"][-1, -1] - p[0, 0], sol["y"][-1, -1] - p[0, 1]) > 1.0


@pytest.mark.gpu
def test_colloc_on_gpu(plans):
    """cfz_colloc, the four vehicles in one launch from their state_ws warm starts: converged, the reference's own rows
    hold for every plan (duals rebuilt by cfz_dual_ws through `Vehicle.get_solution`'s path), and the HIP build lands
    where the CPU build of the same source does."""
    import colloc_emu_binding as ce
    from conflict_rez_amd import engine

    agents = sorted(plans)
    sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    fhs = [float(plans[a][1][-1, 2]) for a in agents]
    ws = engine.state_ws([plans[a][1][0] for a in agents], tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5)
    nlps, guesses, dt0s = [], [], []
    for a, fh, w in zip(agents, fhs, ws):
        assert w["status"] == 0
        tube, p = plans[a]
        nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
        tr = w["traj"]
        z = dict(t=0.1 * np.arange(len(tr)), **{k: tr[:, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))})
        X0 = colloc_guess(nlp, z)
        nlps.append(nlp), guesses.append(X0[: nlp.iDt].reshape(-1, 7)), dt0s.append(X0[nlp.iDt])
    res = engine.colloc(sp, [plans[a][1][0] for a in agents], tubes, guesses, dt0s, fhs, max_iter=400)
    eng = engine.Engine(sp, max_batch=1)
    for a, nlp, g, d0, r in zip(agents, nlps, guesses, dt0s, res):
        assert r["status"] == 0, (a, r["status"], r["iters"])
        sol = {k: r["traj"][:, :, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
        sol["dt"] = r["dt"]
        poses = r["traj"].reshape(-1, 7)[:, :3]
        l, m, _ = eng.dual_ws(poses)
        sol["l"], sol["m"] = l.reshape(nlp.N[0], 6, -1), m.reshape(nlp.N[0], 6, -1)
        rr = reference_residuals(nlp, sol)
        assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9 and abs(rr["cost"] - r["cost"]) < 1e-8 * max(1.0, r["cost"])
        if a in ("vehicle_1", "vehicle_3"):  # the CPU build from the same guess (seconds there; the others take longer)
            re_ = ce.solve(nlp, np.append(g.ravel(), d0), ipm.IpmOptions(**COLLOC_OPT))
            assert re_["status"] == 0 and abs(re_["f"] - r["cost"]) < 2e-2 * re_["f"]
            assert np.abs(re_["X"][: nlp.iDt].reshape(-1, 7)[:, :2] - poses[:, :2]).max() < 0.1


@pytest.mark.gpu
def test_planner_single_problems_then_joint_dual_ws(tmp_path):
    """MultiVehiclePlanner.solve_single_problems (multi_vehicle_planner.py:68-109) on the GPU planning kernels, then
    joint_dual_ws (:208-341) on its results -- the reference's `main` up to the coupled solve."""
    from conflict_rez_amd.control.multi_vehicle_planner import MultiVehiclePlanner
    from conflict_rez_amd.pytypes import VehicleState

    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, strat.generate_strategy(4))
    agents = ["vehicle_%d" % i for i in range(4)]
    paths = interp_along_sets(fn, VehicleBody(), 30)
    mvp = MultiVehiclePlanner(fn, {a: True for a in agents}, {a: {"front": (1, 0, 0), "back": (0, 0, 1)} for a in agents},
                              {a: VehicleState() for a in agents}, {a: float(paths[a][-1, 2]) for a in agents})
    mvp.solve_single_problems()
    for a in agents:
        r, v = mvp.single_results[a], mvp.vehicles[a]
        assert v.final_problem_stats["status"] == 0 and (v.N, v.K) == (5 * (v.num_sets - 1), 5)
        assert len(r.x) == v.N * 6 and r.dt > 0 and np.isclose(r.t[-1], v.N * r.dt)
        assert len(r.l) == v.N and r.l[0][0].shape == (24,) and min(x.min() for row in r.l for x in row) >= 0.0
        assert abs(r.psi[-1] - mvp.final_headings[a]) < 1e-2 and abs(r.v[-1]) < 1e-2
        assert np.allclose(v.state_interpolator(0.0)[:2], paths[a][0, :2], atol=1e-6)
    mvp.joint_dual_ws(K=5)
    n01 = min(mvp.vehicles["vehicle_0"].N, mvp.vehicles["vehicle_1"].N)
    assert len(mvp.joint_l0["vehicle_0"]["vehicle_1"]) == n01 and mvp.joint_s0[("vehicle_0", "vehicle_1")][0][5].shape == (2,)
    # the four-vehicle joint solve itself takes about a minute on the GPU: tools/joint_timing.py; two vehicles below


# ---- the joint plan (multi_vehicle_planner.py:343-480): several vehicles, one shared dt, pairwise separation rows ----
def _joint_problem(plans, agents, nsets, n_obs=6, nps=2, headings=None, vv=True):
    from oracle.colloc_nlp import JointCollocNlp

    sp = scenarios.parking_lot_spec()
    vehs = []
    for i, (a, ns) in enumerate(zip(agents, nsets)):
        tube, p = plans[a]
        tube = tube[:ns] if ns else tube
        fh = (float(p[-1, 2]) if not ns else 0.3) if headings is None else headings[i]
        vehs.append(dict(init_pose=p[0], tube=tube, final_heading=fh))
    return JointCollocNlp(vehs, sp.A_obs[:n_obs], sp.b_obs[:n_obs], N_per_set=nps, vv=vv), sp


@pytest.mark.parametrize("vv", [False, True])
def test_joint_source_values_and_derivatives(plans, vv):
    """Three vehicles with plans of different lengths, one without terminal heading, all pairs: values against the numpy
    statement; gradient, J'nu and the assembled matrix (time-interleaved blocks, dt border, obstacle AND pair rows condensed,
    6 x 6 pair curvature) against central differences / the Schur complement of the full KKT matrix.
    vv: with vertex-vertex rows (body corner against body corner) in the pair working sets."""
    import colloc_emu_binding as ce

    nlp, _ = _joint_problem(plans, ["vehicle_0", "vehicle_1", "vehicle_3"], [3, 4, 3], n_obs=2, headings=[0.3, None, 0.3], vv=vv)
    opt = ipm.IpmOptions(**COLLOC_OPT)
    d = ce.dims(nlp, opt)
    assert (d["n"], d["m"], d["sP"], d["rP"], d["rF"], d["npp"]) == (nlp.n, nlp.m, nlp.sP, nlp.rP, nlp.rF, nlp.npp) and nlp.npp == 72
    rng = np.random.default_rng(1)
    X = np.zeros(nlp.n)
    P = X[: nlp.iDt].reshape(nlp.np, 7)
    for a in range(nlp.V):  # the vehicles scattered around one spot, so that pair rows of both kinds are near active
        blk = P[nlp.off[a] * 6 : nlp.off[a + 1] * 6]
        blk[:, 0] = 16 + 4.0 * a + 0.5 * rng.standard_normal(len(blk))
        blk[:, 1] = 17 + 1.5 * a + 0.5 * rng.standard_normal(len(blk))
        blk[:, 2] = rng.uniform(-3, 3, len(blk))
    P[:, 3:] = 0.3 * rng.standard_normal((nlp.np, 4))
    X[nlp.iDt] = 0.7
    X[nlp.sO :] = rng.uniform(0.1, 1.0, nlp.n - nlp.sO)
    nu = rng.standard_normal(nlp.m)
    nu[nlp.rF + 5 * 1 + 4] = 0.0  # the dead heading row of the vehicle without a terminal heading
    sel = ce.select(nlp, opt, X)
    kinds = (sel[nlp.np * nlp.n_obs :] >> 6).tolist()
    assert (sel == nlp.select(X)).all() and set(kinds) == ({1, 2, 3} if vv else {1, 2}) and (not vv or kinds.count(3) >= 3)
    f, c, g, jt = ce.evaluate(nlp, opt, sel, X, nu)
    assert abs(f - nlp.f(X)) < 1e-11 and np.abs(c - nlp.cons(X, sel)).max() < 1e-11
    h, n = 1e-6, nlp.n
    gfd, J, H = np.zeros(n), np.zeros((nlp.m, n)), np.zeros((n, n))
    for i in range(n):
        e = np.zeros(n)
        e[i] = h
        fp, cp, gp, jp = ce.evaluate(nlp, opt, sel, X + e, nu)
        fm, cm, gm, jm = ce.evaluate(nlp, opt, sel, X - e, nu)
        gfd[i], J[:, i], H[:, i] = (fp - fm) / (2 * h), (cp - cm) / (2 * h), (gp + jp - gm - jm) / (2 * h)
    assert np.abs(g - gfd).max() < 1e-7 and np.abs(jt - J.T @ nu).max() < 1e-6
    sig = np.zeros(n)
    sig[nlp.sO :] = rng.uniform(0.5, 50.0, n - nlp.sO)
    K, bw = ce.kkt(nlp, opt, sel, X, nu, sig=sig)
    M = np.block([[H + np.diag(sig + opt.reg_primal), J.T], [J, -opt.reg_dual * np.eye(nlp.m)]])
    dead = [n + nlp.rF + 5 * 1 + 4]
    E = np.r_[nlp.sO : nlp.sT, nlp.sP : n, n + nlp.rR : n + nlp.rT, n + nlp.rP : n + nlp.m]
    R = np.setdiff1d(np.arange(n + nlp.m), np.r_[E, dead])
    schur = M[np.ix_(R, R)] - M[np.ix_(R, E)] @ np.linalg.solve(M[np.ix_(E, E)], M[np.ix_(E, R)])
    assert np.abs(K[np.ix_(R, R)] - schur).max() < 1e-8 * np.abs(schur).max() and np.abs(K - K.T).max() == 0.0
    assert np.abs(K[E]).max() == 0.0 and np.abs(K[dead]).max() == 0.0 and bw <= ce.half_bandwidth(nlp, opt)


def _joint_guess(plans, agents, jn, sp, nps):
    """Every vehicle's single plan (the CPU build), packed with the mean dt (multi_vehicle_planner.py:361, :370-385)."""
    import colloc_emu_binding as ce

    singles = []
    for a in agents:
        tube, p = plans[a]
        fh = float(p[-1, 2])
        nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=nps, final_heading=fh)
        res = ce.solve(nlp, colloc_guess(nlp, warm_start(tube, p, fh)), ipm.IpmOptions(**COLLOC_OPT))
        assert res["status"] == 0
        singles.append(nlp.unpack(res["X"]))
    return jn.pack(singles, float(np.mean([s["dt"] for s in singles]))), singles


def test_joint_source_solves_pair(plans):
    """Vehicles 2 and 3 at full size (N_per_set = 5): on one clock their single plans pass each other with 0.08 m to spare in
    one separation row; the joint plan converges, every vehicle's own rows and the reference's vehicle-vehicle rows (with
    the duals rebuilt from the poses) hold, and both plans now run on the shared dt."""
    import colloc_emu_binding as ce
    from oracle.colloc_nlp import pair_residuals

    agents = ["vehicle_2", "vehicle_3"]
    jn, sp = _joint_problem(plans, agents, [0, 0], nps=5)
    X0, singles = _joint_guess(plans, agents, jn, sp, 5)
    Xf = np.zeros(jn.n)
    Xf[: len(X0)] = X0
    c0 = jn.cons(Xf, jn.select(Xf))
    assert c0[jn.rP :].min() < 0.2  # the guess is close to contact in some pair row
    res = ce.solve(jn, X0, ipm.IpmOptions(**COLLOC_OPT))
    assert res["status"] == 0 and res["iters"] < 60 and ce.half_bandwidth(jn, ipm.IpmOptions(**COLLOC_OPT)) <= 128
    sols, duals = jn.unpack(res["X"])
    for a in range(2):
        rr = reference_residuals(jn, sols[a], a)
        assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9 and sols[a]["dt"] == sols[0]["dt"]
    pr = pair_residuals(jn, sols, duals)
    assert pr["eq"] < 1e-9 and pr["ineq"] < 1e-9 and pr["bound"] == 0.0
    cost = sum(reference_residuals(jn, sols[a], a)["cost"] for a in range(2))
    assert abs(cost - res["f"]) < 1e-8 * res["f"] and cost >= sum(reference_residuals(jn, singles[a], a)["cost"] for a in range(2)) - 1e-6


@pytest.mark.parametrize("agents,nsets", [(["vehicle_2", "vehicle_3"], [0, 0]), (["vehicle_1", "vehicle_2", "vehicle_3"], [0, 0, 0]),
                                          (["vehicle_0", "vehicle_1", "vehicle_2", "vehicle_3"], [4, 3, 4, 2])])
def test_joint_structured_elimination_equals_the_band_elimination(plans, agents, nsets):
    """cfz_jstruct.inl on the CPU build (CSpec::no_prox bit 2 with V > 1): the joint plan's Newton system eliminated interval by interval
    -- vehicle-major ordering, tube rows condensed, every vehicle's 64-unknown interiors by themselves, the pair-coupled poses of an
    interval index through a capacitance system, a block recursion over the joint separators -- takes the iterates of the band elimination
    (half-bandwidth 100-298 across the vehicles): equal status and iteration count, solution to 1e-7.  Two and three vehicles at full
    length (30 / 50 intervals), four vehicles with plans of DIFFERENT lengths (the shorter vehicles drop out of the later interval indices)
    and one without a terminal heading."""
    import colloc_emu_binding as ce

    nps = 5
    if any(nsets):
        jn, sp = _joint_problem(plans, agents, nsets, nps=nps, headings=[float(plans[a][1][30 * (ns - 1), 2]) if a != "vehicle_1" else None for a, ns in zip(agents, nsets)])
        sing = []
        for a, ns in zip(agents, nsets):
            tube, p = plans[a]
            z = warm_start(tube[:ns], p[: 30 * (ns - 1) + 1], None)
            N = nps * (ns - 1)
            t_i = np.concatenate([k + jn.tau for k in range(N)]) / N * z["t"][-1]
            sing.append(({k: interp1d(z["t"], z[k])(t_i) for k in ("x", "y", "psi", "v", "delta", "a", "w")}, z["t"][-1] / N))
        X0 = jn.pack([s[0] for s in sing], float(np.mean([s[1] for s in sing])))
    else:
        jn, sp = _joint_problem(plans, agents, nsets, nps=nps)
        X0, _ = _joint_guess(plans, agents, jn, sp, nps)
    band = ce.solve(jn, X0, ipm.IpmOptions(**COLLOC_OPT))
    opt = ipm.IpmOptions(**COLLOC_OPT)
    opt.no_prox = 4
    st = ce.solve(jn, X0, opt)
    assert ce.half_bandwidth(jn, opt) == 51 and ce.half_bandwidth(jn, ipm.IpmOptions(**COLLOC_OPT)) > 90
    assert (st["status"], st["iters"]) == (band["status"], band["iters"]) == (0, band["iters"])
    assert np.abs(st["X"] - band["X"]).max() < 1e-7 and abs(st["f"] - band["f"]) < 1e-9 * band["f"]


def test_cyclic_reduction_of_the_joint_separators_equals_the_chain():
    """What the GPU's recursion over the joint separators relies on since round 6 (cfz_jstruct.inl `jbcr_*`; tools/joint_condense_study.py
    `bcr_solve`, numpy, on the separator system of the matrix the CPU build assembles; three vehicles, plans of different lengths: 10 / 15 / 10
    intervals): the
    system is block tridiagonal, every other block can be eliminated at once level by level WITHOUT its neighbours' updates -- the blocks are
    as well conditioned then as the chain finds them -- and the solution is the chain's: to 1e-11 at the guess; with half of the pair rows
    made active (condition 6e15) as close to the chain's as the chain's is to a dense solve of the whole system (8e-8 against 8e-7)."""
    import importlib.util

    spec_ = importlib.util.spec_from_file_location("joint_condense_study", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "joint_condense_study.py"))
    st = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(st)
    jn, opt, X0 = st.study_problem(3, 3)  # (three vehicles on three / four / three strategy steps: 10 and 15 intervals, 16 joint separators)
    rng = np.random.default_rng(0)
    for nu, mu, act, tol in ((np.zeros(jn.m), 0.1, 0.0, 1e-11), (rng.standard_normal(jn.m) * 0.3, 1e-4, 0.5, 1e-8)):
        K, Kown = st.matrices(jn, opt, X0, nu, mu, act, rng)
        sol, ref, conds, _ = st.structured_joint_solve(jn, K, Kown, rng.standard_normal(K.shape[0]), [])
        # (the two recursions agree at least as well as either agrees with a dense solve of the whole system: at condition 6e15 that is 8e-7)
        assert conds["bcr_vs_chain"] < max(tol, conds["chain_vs_dense"]) and conds["bcr_vs_dense"] < 10.0 * max(conds["chain_vs_dense"], 1e-12), (conds["bcr_vs_chain"], conds["bcr_vs_dense"], conds["chain_vs_dense"])
        assert max(conds["bcr_cond"]) < 100.0 * max(conds["sep"])  # eliminated without their neighbours' updates, the blocks are no worse


@pytest.mark.gpu
def test_joint_colloc_on_gpu(plans, tmp_path):
    """cfz_joint_colloc against the CPU build of the same source, and MultiVehiclePlanner.solve_single_problems ->
    solve_final_problem_obca (multi_vehicle_planner.py `main`) for two vehicles."""
    import colloc_emu_binding as ce
    from conflict_rez_amd import engine
    from conflict_rez_amd.control.multi_vehicle_planner import MultiVehiclePlanner
    from conflict_rez_amd.pytypes import VehicleState

    agents = ["vehicle_2", "vehicle_3"]
    jn, sp = _joint_problem(plans, agents, [0, 0], nps=5)
    X0, _ = _joint_guess(plans, agents, jn, sp, 5)
    re_ = ce.solve(jn, X0, ipm.IpmOptions(**COLLOC_OPT))
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    guesses = [X0[7 * 6 * jn.off[a] : 7 * 6 * jn.off[a + 1]].reshape(-1, 7) for a in range(2)]
    r = engine.joint_colloc(scenarios.parking_lot_spec(n_nbr=0, N=2), [plans[a][1][0] for a in agents], tubes, guesses, X0[jn.iDt],
                            [float(plans[a][1][-1, 2]) for a in agents], max_iter=400)
    assert (r["status"], r["iters"]) == (re_["status"], re_["iters"]) == (0, re_["iters"])
    assert abs(r["cost"] - re_["f"]) < 1e-6 * re_["f"] and abs(r["dt"] - re_["X"][jn.iDt]) < 1e-7
    assert np.abs(np.concatenate([t.reshape(-1, 7) for t in r["traj"]]) - re_["X"][: jn.iDt].reshape(-1, 7)).max() < 1e-5
    # BASELINE.json configs[3] in small: a batch of joint plans in one launch (one workgroup each) -- scenarios that differ in the
    # start pose of the first vehicle; every one equals its own call
    scen = []
    for dx in (0.0, 0.05, -0.05):
        ip = [np.array(plans[a][1][0], float) for a in agents]
        ip[0] = ip[0] + np.array([dx, 0.0, 0.0])
        scen.append(dict(init_poses=ip, tubes=tubes, guesses=guesses, dt0=X0[jn.iDt], final_headings=[float(plans[a][1][-1, 2]) for a in agents]))
    spec0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    rb = engine.joint_colloc_batch(spec0, scen, max_iter=400)
    assert [x["status"] for x in rb] == [0, 0, 0] and abs(rb[0]["cost"] - r["cost"]) < 1e-9 and rb[0]["iters"] == r["iters"]
    r2 = engine.joint_colloc(spec0, scen[2]["init_poses"], tubes, guesses, X0[jn.iDt], scen[2]["final_headings"], max_iter=400)
    assert (r2["status"], r2["iters"]) == (0, rb[2]["iters"]) and r2["cost"] == rb[2]["cost"] and abs(rb[2]["traj"][0][0, 0, 0] - (plans[agents[0]][1][0, 0] - 0.05)) < 1e-6
    # the planner surface
    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, strat.generate_strategy(4))
    mvp = MultiVehiclePlanner(fn, {a: True for a in agents}, {a: {"front": (1, 0, 0), "back": (0, 0, 1)} for a in agents},
                              {a: VehicleState() for a in agents}, {a: float(plans[a][1][-1, 2]) for a in agents})
    mvp.solve_single_problems()
    mvp.solve_final_problem_obca()
    assert mvp.final_stats["status"] == 0 and abs(mvp.final_dt - r["dt"]) < 1e-3
    n_max = max(mvp.vehicles[a].N for a in agents)
    fr = mvp.final_results
    assert all(len(fr[a].x) == n_max * 6 + 1 and np.isclose(fr[a].t[-1], n_max * mvp.final_dt) for a in agents)
    # on the common clock the two bodies never come closer than dmin: centres of the rear axles at least a body width apart
    assert np.hypot(fr[agents[0]].x - fr[agents[1]].x, fr[agents[0]].y - fr[agents[1]].y).min() > 1.8


def test_colloc_source_reproduces_fixture(plans):
    """tests/golden/colloc_golden.npz (make_fixtures.py:colloc_golden): from the stored guesses the kernel source reproduces
    status, iteration count, cost and solution of the single plan of vehicle_1 and of the joint plan of vehicles 2 and 3.
    A regression pin of the solver's behaviour, not an oracle; the GPU build is compared with the same data below."""
    import colloc_emu_binding as ce

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colloc_golden.npz"))
    sp = scenarios.parking_lot_spec()
    tube, p = plans["vehicle_1"]
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(p[-1, 2]))
    jn, _ = _joint_problem(plans, ["vehicle_2", "vehicle_3"], [0, 0], nps=5)
    for prob, key in ((nlp, "single"), (jn, "joint")):
        r = ce.solve(prob, g[key + "_guess"], ipm.IpmOptions(**COLLOC_OPT))
        st, it, f = g[key + "_meta"]
        assert (r["status"], r["iters"]) == (int(st), int(it)) and abs(r["f"] - f) < 1e-9 * f
        assert np.abs(r["X"] - g[key + "_sol"]).max() < 1e-7


@pytest.mark.gpu
def test_planning_workspace_reuse(plans):
    """`cfz_plan_ws` (the `_w` entry points): repeated calls on one workspace -- its stream, its kept device buffers, a
    smaller problem after a larger one and back -- give the results of the plain entry points, bit for bit."""
    from conflict_rez_amd import engine

    agents = sorted(plans)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    fhs = [float(plans[a][1][-1, 2]) for a in agents]
    args = ([plans[a][1][0] for a in agents], tubes, [plans[a][1] for a in agents], fhs)
    plain = engine.state_ws(*args, shrink_tube=0.5)
    ws = engine.PlanWorkspace()
    first = engine.state_ws(*args, shrink_tube=0.5, ws=ws)
    one = engine.state_ws(args[0][1:2], tubes[1:2], args[2][1:2], fhs[1:2], shrink_tube=0.5, ws=ws)
    again = engine.state_ws(*args, shrink_tube=0.5, ws=ws)
    for a, b, c in zip(plain, first, again):
        assert a["status"] == b["status"] == c["status"] == 0 and a["iters"] == b["iters"] == c["iters"]
        assert np.array_equal(a["traj"], b["traj"]) and np.array_equal(a["traj"], c["traj"])
    assert np.array_equal(one[0]["traj"], plain[1]["traj"])
    ws.close()


@pytest.mark.gpu
def test_structured_elimination_on_gpu(plans):
    """`cfz_colloc_options.structured` (1 by default): the four vehicles' plans from their state_ws warm starts with the Newton system eliminated
    interval by interval (register-resident dense eliminations of the 64-unknown interiors, eight wavefronts at a time; a block
    recursion over the separators) against the band elimination: equal status and iteration count, trajectories to 1e-7; and 300
    copies of one plan in a batch all equal the lone plan bit for bit."""
    from conflict_rez_amd import engine

    agents = sorted(plans)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    fhs = [float(plans[a][1][-1, 2]) for a in agents]
    init = [plans[a][1][0] for a in agents]
    ws = engine.state_ws(init, tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5)
    sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
    gs = []
    for w_, t_ in zip(ws, tubes):
        N = 5 * len(t_)
        t = 0.1 * np.arange(len(w_["traj"]))
        ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
        gs.append((np.stack([np.interp(ti, t, w_["traj"][:, c]) for c in range(7)], 1), t[-1] / N))
    args = (sp, init, tubes, [g[0] for g in gs], [g[1] for g in gs], fhs)
    band = engine.colloc(*args, max_iter=400, structured=0)
    st = engine.colloc(*args, max_iter=400)  # (structured = 1 is the default)
    for a, b, s_ in zip(agents, band, st):
        assert (s_["status"], s_["iters"]) == (b["status"], b["iters"]) == (0, b["iters"]), a
        assert np.abs(s_["traj"] - b["traj"]).max() < 1e-7 and abs(s_["dt"] - b["dt"]) < 1e-10 and abs(s_["cost"] - b["cost"]) < 1e-8 * b["cost"]
    B = 300
    many = engine.colloc(sp, [init[1]] * B, [tubes[1]] * B, [gs[1][0]] * B, [gs[1][1]] * B, [fhs[1]] * B, max_iter=400)
    assert all(r["iters"] == st[1]["iters"] and np.array_equal(r["traj"], st[1]["traj"]) and r["dt"] == st[1]["dt"] for r in many)


@pytest.mark.gpu
def test_block_elimination_on_the_matrix_cores_equals_the_lane_per_row_elimination(tmp_path):
    """The 64-row eliminations of the structured planning paths (cfz_struct.inl lu64_*: blocked, four pivots a panel, tiles of
    v_mfma_f64_16x16x4_f64) against the lane = row elimination they replaced (wave_lu_regs: one pivot at a time, v_readlane broadcasts), on
    2048 random blocks a third of which have a zero diagonal block with a 1e-7 regularisation like a KKT system: the same pivots and the
    same roundings, so every solution agrees BIT FOR BIT (tools/src/wave_lu_mfma_bench.hip holds both)."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wave_lu_mfma")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(root, "tools", "src", "wave_lu_mfma_bench.hip")])
    out = subprocess.run([exe, "2048", "8"], check=True, capture_output=True, text=True, timeout=300).stdout
    assert "solutions that differ in any bit: 0 of" in out and out.count("failed 0,") == 2, out


@pytest.mark.gpu
def test_structured_elimination_on_short_plans(plans):
    """Plans of 5, 10 and 15 intervals (the separator recursion from both ends meets after one to seven steps), with and without a
    terminal heading: the structured elimination ends where the band elimination ends -- status (a plan that fails its line search
    among them), iteration count, trajectory to 1e-5."""
    from conflict_rez_amd import engine

    sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
    seen = set()
    for a in ("vehicle_1", "vehicle_0"):
        for S in (2, 3, 4):
            tube = [((s_["back"][0], s_["back"][1]), (s_["front"][0], s_["front"][1])) for s_ in plans[a][0][1:S]]
            p = plans[a][1][: 30 * (S - 1) + 1]
            for fh in (None, float(p[-1, 2])):
                ws = engine.state_ws([p[0]], [tube], [p], [fh], shrink_tube=0.5)[0]
                assert ws["status"] == 0
                N = 5 * len(tube)
                t = 0.1 * np.arange(len(ws["traj"]))
                ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
                g = np.stack([np.interp(ti, t, ws["traj"][:, c]) for c in range(7)], 1)
                args = (sp, [p[0]], [tube], [g], [t[-1] / N], [fh])
                b = engine.colloc(*args, max_iter=400, structured=0)[0]
                s1 = engine.colloc(*args, max_iter=400)[0]
                assert (s1["status"], s1["iters"]) == (b["status"], b["iters"]) and np.abs(s1["traj"] - b["traj"]).max() < 1e-5, (a, S, fh)
                seen.add(b["status"])
                # ... and not only HIP against HIP (VERDICT r4 item 2 of "weak"): the CPU build of the solver source with its own, generic
                # band elimination (one pivot at a time, partial pivoting: none of the structured elimination's code) from the same guess -- status,
                # iteration count, the plan to 1e-5 where it converges
                import colloc_emu_binding as ce

                nlp = CollocNlp(p[0], plans[a][0][:S], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
                rc = ce.solve(nlp, nlp.pack({k: g[:, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}, t[-1] / N), ipm.IpmOptions(**COLLOC_OPT))
                assert (rc["status"], rc["iters"]) == (s1["status"], s1["iters"]), (a, S, fh, rc["status"], rc["iters"], s1["status"], s1["iters"])
                if rc["status"] == 0:
                    # (measured: 1.7e-6 at worst on these short, badly determined plans -- 5 intervals stop at tol = 1e-2 with flat directions)
                    dcol = np.abs(rc["X"][: nlp.iDt].reshape(-1, 7) - s1["traj"].reshape(-1, 7)).max(0)
                    assert dcol[:5].max() < 1e-5 and dcol[5:].max() < 1e-3 and abs(rc["X"][nlp.iDt] - s1["dt"]) < 1e-6, (a, S, fh, dcol, rc["X"][nlp.iDt] - s1["dt"])
    assert 0 in seen


@pytest.mark.gpu
def test_colloc_fixture_on_gpu(plans):
    """The same fixture through the C ABI: cfz_colloc and cfz_joint_colloc from the stored guesses."""
    from conflict_rez_amd import engine

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colloc_golden.npz"))
    spec = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tb = lambda a: [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]]
    fh = lambda a: float(plans[a][1][-1, 2])
    X0, st = g["single_guess"], g["single_meta"]
    for one_pivot in (0, 1):  # both eliminations of the kernel: a panel of 16 pivots at a time (default), one pivot at a time
        r = engine.colloc(spec, [plans["vehicle_1"][1][0]], [tb("vehicle_1")], [X0[:-1].reshape(-1, 7)], [X0[-1]], [fh("vehicle_1")], max_iter=400,
                          one_pivot=one_pivot)[0]
        assert (r["status"], r["iters"]) == (int(st[0]), int(st[1])) and abs(r["cost"] - st[2]) < 1e-8 * st[2]
        assert np.abs(r["traj"].reshape(-1, 7) - g["single_sol"][:-1].reshape(-1, 7)).max() < 1e-6 and abs(r["dt"] - g["single_sol"][-1]) < 1e-8
    # the iteration limit is reported as such (status 1): the GPU build once returned 0 for this exit
    r5 = engine.colloc(spec, [plans["vehicle_1"][1][0]], [tb("vehicle_1")], [X0[:-1].reshape(-1, 7)], [X0[-1]], [fh("vehicle_1")], max_iter=5)[0]
    assert (r5["status"], r5["iters"]) == (1, 5)
    J0, st = g["joint_guess"], g["joint_meta"]
    agents = ["vehicle_2", "vehicle_3"]
    n2 = 6 * 5 * (len(plans["vehicle_2"][0]) - 1)
    r = engine.joint_colloc(spec, [plans[a][1][0] for a in agents], [tb(a) for a in agents],
                            [J0[: 7 * n2].reshape(-1, 7), J0[7 * n2 : -1].reshape(-1, 7)], J0[-1], [fh(a) for a in agents], max_iter=400)
    assert (r["status"], r["iters"]) == (int(st[0]), int(st[1])) and abs(r["cost"] - st[2]) < 1e-8 * st[2]
    assert np.abs(np.concatenate([t.reshape(-1, 7) for t in r["traj"]]) - g["joint_sol"][:-1].reshape(-1, 7)).max() < 1e-6


def test_inertia_correction_remembers_and_a_failed_line_search_is_retried(plans):
    """Two pins of the inertia correction of the planning solvers (IPOPT's Algorithm IC, oracle/ipm.py next_delta_w):
    * the ladder: delta_w = 0 first; then 1e-4 if no iteration has needed a perturbation yet, else a third of the last one that
      worked; then x 8 -- not a climb from 1e-4 in every iteration (five factorisations per iteration on the slowest joint plans);
    * tests/golden/joint_retry_instance.npz (inputs of one four-vehicle joint plan of the 256-plan GPU test, start poses scattered
      by 3 cm): at mu = mu_floor a perturbation of a third of the last one passes the curvature test but the step is no descent
      direction and the line search fails -- the solver repeats the iterate with eight times the perturbation instead of ending
      with status 2 (39 iterations, status 0)."""
    import os

    import colloc_emu_binding as ce

    assert [ipm.next_delta_w(d, l) for d, l in ((0.0, 0.0), (1e-4, 0.0), (0.0, 0.3), (0.1, 0.3))] == [1e-4, 8e-4, 0.3 / 3.0, 0.8]
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "joint_retry_instance.npz"))
    sp = scenarios.parking_lot_spec()
    agents = ["vehicle_0", "vehicle_1", "vehicle_2", "vehicle_3"]
    from oracle.colloc_nlp import JointCollocNlp

    jn = JointCollocNlp([dict(init_pose=w["init"][i], tube=plans[a][0], final_heading=float(plans[a][1][-1, 2])) for i, a in enumerate(agents)],
                        sp.A_obs, sp.b_obs, N_per_set=5)
    X0 = np.concatenate([w[f"g{i}"].ravel() for i in range(4)] + [[float(w["dt0"])]])
    r = ce.solve(jn, X0, ipm.IpmOptions(**COLLOC_OPT))
    assert r["status"] == 0 and r["iters"] < 60, (r["status"], r["iters"])


def test_mirror_symmetry_of_the_collocation_plan():
    """The reflection y -> 35 - y of tube, obstacles (the pillar of tests/golden/colloc_independent_vv.npz included: a corner-to-corner
    contact), start pose, terminal heading and guess mirrors the plan of the planning source: equal status and iteration count, cost
    to 1e-9, poses to 1e-6 -- every sign of the tube rows, the face rows and the vertex-vertex rows with their second derivatives is
    on this path twice."""
    import colloc_emu_binding as ce
    from test_independent_solver import _colloc_fixture

    def mir_poly(A, b):
        A = np.asarray(A, float); b = np.asarray(b, float)
        return A * np.array([1.0, -1.0]), b - 35.0 * A[..., 1]

    for agent in ("vehicle_2_pillar", "vehicle_1"):
        d, g, (tube, p, fh, sp) = _colloc_fixture(agent)
        G = d["guess"][:-1].reshape(-1, 7)
        cols = ("x", "y", "psi", "v", "delta", "a", "w")
        nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
        r0 = ce.solve(nlp, nlp.pack({k: G[:, c] for c, k in enumerate(cols)}, float(d["guess"][-1])), ipm.IpmOptions(**COLLOC_OPT))
        tube_m = [dict(front=mir_poly(*s["front"]), back=mir_poly(*s["back"])) for s in tube]
        Am, bm = mir_poly(sp.A_obs, sp.b_obs)
        Gm = G * np.array([1.0, -1.0, -1.0, 1.0, -1.0, 1.0, -1.0]) + np.array([0.0, 35.0, 0, 0, 0, 0, 0])
        p0m = np.array([p[0][0], 35.0 - p[0][1], -p[0][2]])
        nlm = CollocNlp(p0m, tube_m, Am, bm, N_per_set=5, final_heading=-fh)
        r1 = ce.solve(nlm, nlm.pack({k: Gm[:, c] for c, k in enumerate(cols)}, float(d["guess"][-1])), ipm.IpmOptions(**COLLOC_OPT))
        assert (r0["status"], r0["iters"]) == (r1["status"], r1["iters"]) == (0, r0["iters"]), (agent, r0["status"], r0["iters"], r1["status"], r1["iters"])
        assert abs(r0["f"] - r1["f"]) < 1e-9 * abs(r0["f"])
        P0, P1 = r0["X"][: nlp.iDt].reshape(-1, 7), r1["X"][: nlm.iDt].reshape(-1, 7)
        back = P1 * np.array([1.0, -1.0, -1.0, 1.0, -1.0, 1.0, -1.0]) + np.array([0.0, 35.0, 0, 0, 0, 0, 0])
        assert np.abs(back - P0).max() < 1e-6 and abs(r0["X"][nlp.iDt] - r1["X"][nlm.iDt]) < 1e-9


def test_mirror_symmetry_of_the_joint_plan():
    """The same reflection on a JOINT plan whose optimum has a corner of one body against a corner of another
    (tests/golden/joint_independent_02_d20_s66.npz): the vehicle-vehicle rows of all three kinds, their 6 x 6 second derivatives
    and the handover between working sets mirror too -- equal status and iteration count, cost to 1e-8, poses to 1e-5."""
    import colloc_emu_binding as ce
    from test_independent_solver import VV_BODY, _joint_fixture

    from oracle.colloc_nlp import JointCollocNlp

    def mir_poly(A, b):
        A = np.asarray(A, float); b = np.asarray(b, float)
        return A * np.array([1.0, -1.0]), b - 35.0 * A[..., 1]

    d, gs, plans, sp = _joint_fixture(VV_BODY)
    agents = d["agents"]
    sgn, off = np.array([1.0, -1.0, -1.0, 1.0, -1.0, 1.0, -1.0]), np.array([0.0, 35.0, 0, 0, 0, 0, 0])
    cols = ("x", "y", "psi", "v", "delta", "a", "w")
    res = []
    for mirror in (False, True):
        vehs, singles = [], []
        for i, a in enumerate(agents):
            tube, p = plans[a]
            p0, fh, G = np.array(p[0], float), float(p[-1, 2]), d[f"guess{i}"]
            if mirror:
                tube = [dict(front=mir_poly(*s["front"]), back=mir_poly(*s["back"])) for s in tube]
                p0, fh, G = np.array([p0[0], 35.0 - p0[1], -p0[2]]), -fh, G * sgn + off
            vehs.append(dict(init_pose=p0, tube=tube, final_heading=fh))
            singles.append({k: G[:, c].reshape(gs[i].N, 6) for c, k in enumerate(cols)})
        A_obs, b_obs = mir_poly(sp.A_obs, sp.b_obs) if mirror else (sp.A_obs, sp.b_obs)
        jn = JointCollocNlp(vehs, A_obs, b_obs, N_per_set=5, dmin=d["dmin_"])
        r = ce.solve(jn, jn.pack(singles, float(d["dt0"])), ipm.IpmOptions(**COLLOC_OPT))
        res.append((r, jn))
    (r0, j0), (r1, j1) = res
    assert (r0["status"], r0["iters"]) == (r1["status"], r1["iters"]) == (0, r0["iters"]), (r0["status"], r0["iters"], r1["status"], r1["iters"])
    assert abs(r0["f"] - r1["f"]) < 1e-8 * abs(r0["f"]) and abs(r0["X"][j0.iDt] - r1["X"][j1.iDt]) < 1e-8
    P0, P1 = r0["X"][: j0.iDt].reshape(-1, 7), r1["X"][: j1.iDt].reshape(-1, 7)
    assert np.abs(P1 * sgn + off - P0).max() < 1e-5
