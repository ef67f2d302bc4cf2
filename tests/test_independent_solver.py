"""The oracle against an independent solver on the reference's own formulation.

scipy's SLSQP (a general-purpose SQP, no code in common with oracle/ipm.py) solves the MPC-step NLP in the
reference's own variables -- primal AND the OBCA duals l, m, lambda_ij, lambda_ji, s, as IPOPT sees it
(oracle/reference_nlp.py, vehicle_follower.py:146-368) -- on reduced instances (short horizon, two obstacles, one
neighbour, so that finite-difference Jacobians stay cheap).  The engine's formulation restricts the duals to
face-normal certificates, so its feasible set is contained in the reference's:
    cost(reference optimum) <= cost(oracle) <= cost(reference optimum) + what <= 0.3 dmin of extra clearance costs.
Both solvers are run to tight tolerances here (the production tolerance 1e-2 would hide any difference)."""
import numpy as np
import pytest
from scipy.optimize import minimize

from conflict_rez_amd import scenarios
from oracle import ipm
from oracle.mpc_nlp import MpcNlp, MpcSpec, reference_residuals, solve_mpc
from oracle.reference_nlp import ReferenceNlp


def _instance(k0, v, n_obs_keep, N):
    table, _ = scenarios.load_reference_table()
    sp = scenarios.parking_lot_spec(n_nbr=1, N=N, n_obs=6)
    A, b = sp.A_obs[list(n_obs_keep)], sp.b_obs[list(n_obs_keep)]
    spec = MpcSpec(N=N, dt=sp.dt, A_obs=A, b_obs=b, n_nbr=1)
    T = table.shape[1]
    idx = np.minimum(k0 + np.arange(N), T - 1)
    adv = np.minimum(idx + 1, T - 1)
    u = (v + 1) % 4
    x0 = table[v, k0, :5] + np.array([0.03, -0.02, 0.01, 0.02, 0.0])
    return spec, x0, table[v, idx, :3].T.copy(), table[u, adv, :3].T.copy()[None], table[v, adv, :].T.copy()


def _slsqp(nlp, X0):
    lo, hi = nlp.bounds()
    cons = [dict(type="eq", fun=lambda X: nlp.constraints(X)[0]), dict(type="ineq", fun=lambda X: nlp.constraints(X)[1])]
    out = minimize(nlp.cost, X0, method="SLSQP", bounds=list(zip(lo, hi)), constraints=cons, options=dict(maxiter=300, ftol=1e-12))
    eq, ineq = nlp.constraints(out.x)
    assert out.status == 0 and np.abs(eq).max() < 1e-6 and ineq.min() > -1e-6
    return out


TIGHT = dict(tol=1e-7, constr_viol_tol=1e-8, compl_inf_tol=1e-8, dual_inf_tol=1e-5)


@pytest.mark.parametrize("k0,v,obs", [(60, 1, (0, 1)), (150, 3, (3, 4)), (30, 0, (1, 4))])
def test_slsqp_cannot_improve_the_oracle_solution(k0, v, obs):
    """Started AT the oracle's solution (with its certificate duals), the independent solver stays there."""
    spec, x0, ref, nbr, zu = _instance(k0, v, obs, 6)
    r = solve_mpc(spec, x0, ref, nbr, zu, ipm.IpmOptions(**TIGHT))
    assert r["status"] == 0
    res = reference_residuals(spec, x0, ref, nbr, r["sol"])
    assert res["eq"] < 1e-7 and res["ineq"] < 1e-7 and res["bound"] < 1e-9  # feasible for the reference NLP, duals included
    nlp = ReferenceNlp(spec, x0, ref, nbr)
    X0 = nlp.pack(r["sol"])
    assert abs(nlp.cost(X0) - res["cost"]) < 1e-9
    out = _slsqp(nlp, X0)
    assert abs(out.fun - res["cost"]) <= 1e-6 * max(1.0, res["cost"])
    s2 = nlp.unpack(out.x)
    assert max(np.abs(s2[k] - r["sol"][k]).max() for k in ("x", "y", "psi")) < 1e-5


@pytest.mark.parametrize("k0,v,obs,dmin,active_block", [(120, 1, (0, 1), 0.12, 1), (50, 2, (1, 4), 0.09, 2)])
def test_slsqp_from_the_warm_start_reaches_the_oracle_solution(k0, v, obs, dmin, active_block):
    """Independent convergence with an ACTIVE collision constraint (dmin raised until a static obstacle, resp. the
    neighbour, is touched): from the MPC warm start, with the duals of the warm-start poses, SLSQP ends at the oracle's
    cost and trajectory -- the certificate-eliminated problem and the reference's NLP have the same optimum here."""
    spec, x0, ref, nbr, zu = _instance(k0, v, obs, 6)
    spec.dmin = dmin
    r = solve_mpc(spec, x0, ref, nbr, zu, ipm.IpmOptions(**TIGHT))
    assert r["status"] == 0 and abs(r["sep"].min() - dmin) < 1e-6
    assert np.unravel_index(np.argmin(r["sep"]), r["sep"].shape)[1] == active_block
    res = reference_residuals(spec, x0, ref, nbr, r["sol"])
    m = MpcNlp(spec, x0, ref, nbr)
    ws = m.unpack(m.pack(dict(zip(("x", "y", "psi", "v", "delta", "a", "w"), zu))))  # warm-start poses and their certificates
    nlp = ReferenceNlp(spec, x0, ref, nbr)
    out = _slsqp(nlp, nlp.pack(ws))
    assert out.fun <= res["cost"] + 1e-7  # the reference's feasible set contains the engine's
    assert res["cost"] - out.fun <= 1e-6 * max(1.0, res["cost"])
    s2 = nlp.unpack(out.x)
    assert max(np.abs(s2[k] - r["sol"][k]).max() for k in ("x", "y", "psi")) < 1e-5


def test_state_ws_oracle_matches_slsqp():
    """The planning oracle (oracle/plan_nlp.py, vehicle.py:99-231) against scipy's SLSQP on a short tube (4 strategy
    steps, 6 Euler steps each): same optimal cost and trajectory, started from the spline guess."""
    import os
    import tempfile

    from conflict_rez_amd import strategy as strat
    from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
    from conflict_rez_amd.vehicle_types import VehicleBody
    from oracle.plan_nlp import StateWsNlp

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 6)
    a = "vehicle_1"
    tube = [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]][:4]
    p = paths[a][: 6 * 3 + 1]
    nlp = StateWsNlp(p[0], tube, N=6, shrink_tube=0.5)
    X0 = nlp.pack(p[:, 0], p[:, 1], p[:, 2])
    ro = ipm.solve(nlp, X0, ipm.IpmOptions(max_iter=300, hessian="exact", reg_dual=1e-9, stall_iters=0, tol=1e-8,
                                           constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6))
    assert ro["status"] == 0
    nz = 7 * nlp.T + 5  # SLSQP works on the trajectory variables; the tube rows become inequalities
    r0 = 7 + 5 * nlp.T

    def eq(z):
        return nlp.cons(np.concatenate([z, np.zeros(nlp.n - nz)]))[:r0]

    def ineq(z):  # A p - (b - shrink) <= 0
        return -nlp.cons(np.concatenate([z, np.zeros(nlp.n - nz)]))[r0 : r0 + 8 * nlp.n_chk]

    lo, hi = nlp.xl[:nz], nlp.xu[:nz]
    out = minimize(lambda z: nlp.f(np.concatenate([z, np.zeros(nlp.n - nz)])), X0[:nz], method="SLSQP",
                   bounds=[(None if not np.isfinite(l) else l, None if not np.isfinite(u) else u) for l, u in zip(lo, hi)],
                   constraints=[dict(type="eq", fun=eq), dict(type="ineq", fun=ineq)], options=dict(maxiter=500, ftol=1e-13))
    assert out.status == 0 and np.abs(eq(out.x)).max() < 1e-7 and ineq(out.x).min() > -1e-7
    assert abs(out.fun - ro["f"]) < 1e-5 * max(1.0, ro["f"]), (out.fun, ro["f"])
    so, ss = nlp.unpack(ro["X"]), nlp.unpack(np.concatenate([out.x, np.zeros(nlp.n - nz)]))
    assert max(np.abs(so[k] - ss[k]).max() for k in ("x", "y", "psi", "v")) < 1e-3


def test_collocation_plan_is_stationary_for_slsqp():
    """The collocation plan of the kernel source (CPU build, tight tolerances) on a short tube with the two nearest
    obstacles, handed to scipy's SLSQP in the trajectory variables and dt (numpy statement of oracle/colloc_nlp.py; the
    collision rows of the final working set and the tube rows as inequalities): five SLSQP iterations started there do
    not move it (1e-5) and do not lower the cost.  Run to convergence (175 iterations, 200 s) it ends 9e-7 away at the
    same cost to 1e-10 -- too slow to keep in the suite."""
    import os
    import tempfile

    import colloc_emu_binding as ce
    from conflict_rez_amd import scenarios, strategy as strat
    from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
    from conflict_rez_amd.vehicle_types import VehicleBody
    from oracle.colloc_nlp import CollocNlp

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    a, ns = "vehicle_1", 4
    tube = [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]][:ns]
    p = paths[a][: 30 * (ns - 1) + 1]
    sp = scenarios.parking_lot_spec()
    keep = np.argsort([np.min([np.max(sp.A_obs[j] @ q[:2] - sp.b_obs[j]) for q in p]) for j in range(6)])[:2]
    nlp = CollocNlp(p[0], tube, sp.A_obs[keep], sp.b_obs[keep], N_per_set=2, final_heading=float(p[-1, 2]))
    N = nlp.N[0]
    t = 0.1 * np.arange(len(p))
    t_i = np.concatenate([i + nlp.tau for i in range(N)]) / N * t[-1]
    zu0 = {k: np.interp(t_i, t, p[:, c]) for c, k in enumerate(("x", "y", "psi"))}
    zu0.update({k: np.zeros(len(t_i)) for k in ("delta", "a", "w")})
    vp = np.hypot(np.gradient(p[:, 0], 0.1), np.gradient(p[:, 1], 0.1))  # speed along the spline, zero at both ends
    vp[0] = vp[-1] = 0.0
    zu0["v"] = np.interp(t_i, t, vp)
    X0 = nlp.pack(zu0, t[-1] / N)
    opt = ipm.IpmOptions(max_iter=600, reg_dual=1e-9, tol=1e-8, constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6)
    opt.no_prox = 1  # the unregularised rows: the proximal form stops at c = delta_c nu
    res = ce.solve(nlp, X0, opt)
    assert res["status"] == 0
    nz = nlp.iDt + 1
    Xs = np.zeros(nlp.n)
    Xs[:nz] = res["X"]
    sel = nlp.select(Xs)

    def full(z):
        X = np.zeros(nlp.n)
        X[:nz] = z
        return X

    def eq(z):
        c = nlp.cons(full(z), sel)
        return np.concatenate([c[: nlp.rR], c[nlp.rF :]])

    def ineq(z):  # separation - dmin >= 0 ; tube rows <= 0
        c = nlp.cons(full(z), sel)
        return np.concatenate([c[nlp.rR : nlp.rT], -c[nlp.rT : nlp.rF]])

    lo, hi = np.full(nz, -np.inf), np.full(nz, np.inf)
    for c, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
        lo[c : nlp.iDt : 7], hi[c : nlp.iDt : 7] = nlp.bounds[2 * j], nlp.bounds[2 * j + 1]
    bounds = [(None if not np.isfinite(l) else l, None if not np.isfinite(u) else u) for l, u in zip(lo, hi)]
    assert np.abs(eq(res["X"])).max() < 1e-8 and ineq(res["X"]).min() > -1e-8
    out = minimize(lambda z: nlp.f(full(z)), res["X"][:nz], method="SLSQP", bounds=bounds,
                   constraints=[dict(type="eq", fun=eq), dict(type="ineq", fun=ineq)], options=dict(maxiter=5, ftol=1e-14))
    assert np.abs(out.x - res["X"][:nz]).max() < 1e-5 and out.fun > res["f"] - 1e-7 and np.abs(eq(out.x)).max() < 1e-7


# ---- full size: N = 30, six obstacles, three neighbours (tests/golden/mpc_independent.npz, make_independent.py) --------------
def _independent_fixture():
    return _fixture_file("mpc_independent.npz")


def _fixture_file(name):
    import os

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    return {k: d[k] for k in d.files}, MpcSpec(N=30, dt=0.1, A_obs=d["A_obs"], b_obs=d["b_obs"], n_nbr=3)


def check_against_independent(solve, tight_tol, prod, fixture="mpc_independent.npz", better=(), fails=(), stuck=()):
    """Shared by the CPU tests (C port) and the GPU tests (HIP engine): `solve(b, x0, ref, nbr, zu) -> (status, zu [7, N])`.
    Asserted for every instance of the fixture:
      * the engine's trajectory satisfies the GEOMETRIC statement of the reference's constraints (polygon distance >= dmin,
        dynamics, initial state) -- i.e. it is feasible for the reference's NLP;
      * it is the independent solver's optimum: cost to 1e-6 (tight) / 2e-4 (production tolerances), poses to 1e-4 m / rad
        (tight) and inside the claimed band 5e-2 m, 5e-2 rad at the production tolerance 1e-2 -- with vertex-vertex contacts
        (kind 3 rows), with nine active rows (instance 7 of the first fixture), with an intruder's corner in the path;
      * `better`: instances where the engine ends at a DIFFERENT local optimum that is cheaper than the independent solver's
        (the problems are not convex; SLSQP started at the engine's point stays there): only feasibility is asserted;
      * `fails`: instances the engine does NOT solve (status != 0, so the caller takes the reference's fallback).  Asserted as
        failures so that a silent wrong answer cannot hide there.  (Round 3 had two: warm starts 0.3-0.5 m inside an obstacle's
        clearance.  With the dual regularisation of the rows and the restoration phase of round 4 there are none.)
      * `stuck`: instances where the engine ends at a point that is feasible only WITHIN the production tolerance and dearer than
        the independent optimum: it converges there (status 0) at constr_viol_tol = 1e-2 and reports local infeasibility (status 5,
        after a failed restoration) at tight tolerances.  mpc_independent_turn.npz 13: the body's corner passes an obstacle 1.6 mm
        inside the clearance at stage 4, with the steering rate of the stages before it at its bound -- SLSQP started from that point
        cannot remove the 1.6 mm either (oracle/independent_mpc.py: status 8, violation unchanged), the independent optimum turns
        in two stages earlier.  Asserted: status, feasibility within 1e-2, cost within 30 % of the independent optimum.
    Tight mode accepts status 2 (line search exhausted at the rounding floor of the merit function) next to 0: what says
    "optimal" here are the comparisons, not the engine's own verdict."""
    from oracle import independent_mpc as im

    d, ospec = _fixture_file(fixture)
    gaps = []
    for b in range(len(d["x0"])):
        status, z = solve(b, d["x0"][b], d["ref"][b], d["nbr"][b], d["zu"][b])
        if b in fails:
            assert status != 0, (b, status)
            continue
        if b in stuck:
            assert status == (0 if prod else 5), (b, status)
            nlp = im.GeometricMpc(ospec, d["x0"][b], d["ref"][b], d["nbr"][b])
            X = z.T.ravel()
            assert np.abs(nlp.eq(X)).max() < 1e-2 and -1e-2 < nlp.ineq(X).min() < -1e-4, b
            assert 0.0 < (nlp.cost(X) - d["cost"][b]) / d["cost"][b] < 0.3, b
            continue
        assert status == 0 or (status == 2 and not prod), (b, status)
        nlp = im.GeometricMpc(ospec, d["x0"][b], d["ref"][b], d["nbr"][b])
        X = z.T.ravel()
        assert np.abs(nlp.eq(X)).max() < (1e-6 if not prod else 1e-2), b
        assert nlp.ineq(X).min() > -(1e-6 if not prod else 1e-2), b
        cost, ref_cost = nlp.cost(X), d["cost"][b]
        gap = (cost - ref_cost) / ref_cost
        gaps.append(gap)
        if b in better:
            assert gap < 0.0, (b, gap)
            continue
        assert gap > -(1e-6 if not prod else 1e-3), (b, gap)  # feasible for the reference's NLP: cannot be cheaper
        dpose = np.abs(z[:3] - d["sol"][b][:3]).max()
        assert gap < (1e-6 if not prod else 2e-4), (b, gap)
        assert dpose < (tight_tol if not prod else 5e-2), (b, dpose)
    return gaps


TIGHT_FULL = dict(tol=1e-7, constr_viol_tol=1e-8, compl_inf_tol=1e-8, dual_inf_tol=1e-5)


@pytest.mark.parametrize("prod", [False, True])
def test_full_size_instances_against_the_independent_solver(prod):
    """The engine's algorithm (C port of the kernel's formulation) against the independent optimum of the reference's NLP at
    the reference's full size, at tight tolerances and at the production tolerance 1e-2."""
    from oracle import port

    _, ospec = _independent_fixture()
    opt = ipm.IpmOptions() if prod else ipm.IpmOptions(**TIGHT_FULL, stall_iters=0)

    def solve(b, x0, ref, nbr, zu):
        r = port.solve(ospec, x0, ref, nbr, zu.T.copy(), opt)
        return r["status"], r["p"].T

    check_against_independent(solve, 1e-4, prod)


# fixture -> (instances where the engine's local optimum is CHEAPER than the independent solver's (0.44 %, 2.7 %; 70 % on the turning
#             instance whose start is 0.5 m inside a clearance: the restoration phase finds a way round the box that SLSQP does not --
#             SLSQP started from the engine's optimum stays there),
#             instances the engine fails on (none since round 4),
#             instances where it stops at a point feasible only within the production tolerance (see check_against_independent))
POPULATIONS = {"mpc_independent_more.npz": ((18, 25), (), ()), "mpc_independent_obs.npz": ((), (), ()), "mpc_independent_turn.npz": ((0,), (), (13,))}


@pytest.mark.parametrize("fixture", sorted(POPULATIONS))
@pytest.mark.parametrize("prod", [False, True])
def test_population_against_the_independent_solver(prod, fixture):
    """Three populations of full-size instances with the independent solver's optimum each:
      mpc_independent_more.npz (make_independent_more.py): 16 from the bench's scenario sampler with active collision rows, 16 with
        a parked intruder's corner in the ego's path (vertex-vertex contacts);
      mpc_independent_obs.npz (make_independent_obs.py): 24 with reference, warm start and state pushed 0.2-0.9 m sideways into the
        parking-lot furniture (static obstacles active);
      mpc_independent_turn.npz (make_independent_turn.py): 24 turning references pushed 0.1-1.6 m sideways (corners swung past
        static boxes, six vertex-vertex contacts with them).
    76 of 80 are solved to the independent optimum; on three the engine ends at a cheaper local optimum; on one at a point that is
    feasible only within the production tolerance (`stuck`).  None fails (round 3: two, warm starts half a metre inside a clearance).  (The first population is
    what found two defects of the first vertex-vertex implementation: the kept vertex pair of a face was compared by its first
    entry instead of its minimum, and the filter kept entries of the previous working set.)"""
    from oracle import port

    _, ospec = _fixture_file(fixture)
    opt = ipm.IpmOptions() if prod else ipm.IpmOptions(**TIGHT_FULL, stall_iters=0)

    def solve(b, x0, ref, nbr, zu):
        r = port.solve(ospec, x0, ref, nbr, zu.T.copy(), opt)
        return r["status"], r["p"].T

    check_against_independent(solve, 1e-4, prod, fixture=fixture, better=POPULATIONS[fixture][0], fails=POPULATIONS[fixture][1], stuck=POPULATIONS[fixture][2])


def test_face_normal_certificates_alone_are_a_restriction():
    """vv_rows = 0 (face-normal certificates only, round 1's formulation): with a vertex-vertex pair active the feasible set
    is strictly smaller than the reference's -- cost 0.2-0.8 % above the independent optimum on instances 8, 9, 11, and on
    instance 7 another stationary point at twice the cost.  Pins what the vertex-vertex rows are for."""
    from oracle import independent_mpc as im
    from oracle import port

    d, ospec = _independent_fixture()
    ospec.vv_rows = False
    for b, lo, hi in ((7, 0.9, 1.1), (8, 1e-4, 1e-2), (9, 1e-4, 1e-2), (11, 1e-4, 1e-2)):
        r = port.solve(ospec, d["x0"][b], d["ref"][b], d["nbr"][b], d["zu"][b].T.copy(), ipm.IpmOptions())
        nlp = im.GeometricMpc(ospec, d["x0"][b], d["ref"][b], d["nbr"][b])
        gap = (nlp.cost(r["p"].ravel()) - d["cost"][b]) / d["cost"][b]
        assert r["status"] == 0 and lo < gap < hi, (b, r["status"], gap)


def test_independent_fixture_is_reproducible():
    """The committed optimum of one instance is what the independent solver returns today (scipy SLSQP on the geometric
    statement, oracle/independent_mpc.py), and its polygon distance agrees with the QP oracle of oracle/geometry.py."""
    from oracle import independent_mpc as im
    from oracle.geometry import body_polygon_world, polygon_distance

    d, ospec = _independent_fixture()
    r = im.solve(ospec, d["x0"][1], d["ref"][1], d["nbr"][1], d["zu"][1])
    assert abs(r["cost"] - d["cost"][1]) < 1e-8 * d["cost"][1] and np.abs(r["zu"][:3] - d["sol"][1][:3]).max() < 1e-6
    assert d["n_active"][1] > 0 and (d["n_vv"][8:] > 0).all() and (d["n_vv"][:8] == 0).all()
    nlp = im.GeometricMpc(ospec, d["x0"][8], d["ref"][8], d["nbr"][8])
    sep = nlp.separations(d["sol"][8][:3].T)
    k, j = np.unravel_index(np.argmin(sep), sep.shape)
    W = body_polygon_world(d["sol"][8][:3, k], ospec.g)
    Q = nlp.obs[j] if j < ospec.n_obs else body_polygon_world(d["nbr"][8][j - ospec.n_obs, :, k], ospec.g)
    assert abs(polygon_distance(Q, W)[0] - sep[k, j]) < 1e-7 and abs(sep[k, j] - ospec.dmin) < 1e-6


# ---- full size collocation plans (tests/golden/colloc_independent.npz, make_independent_colloc.py) ----------------------------
COLLOC_AGENTS = ("vehicle_1", "vehicle_2", "vehicle_3")
# vehicle 0 on its first six strategy steps (25 intervals; tests/golden/colloc_independent_trunc.npz, `make_independent_colloc.py --truncated
# vehicle_0_s6`): at full length (50 intervals, it waits for 20 of them) the plan has two local solutions and neither solver a tight optimum
TRUNC_AGENTS = ("vehicle_0_s6",)


def _colloc_fixture(agent="vehicle_1"):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_independent_colloc import problem
    from oracle.independent_colloc import GeometricColloc

    if agent.endswith("_pillar"):  # a plan with a corner-to-corner contact: seventh obstacle, make_independent_colloc_vv.py
        import dataclasses

        f = np.load(os.path.join(here, "golden", "colloc_independent_vv.npz"))
        d = {k: f[f"{agent}_{k}"] for k in ("guess", "traj", "dt", "cost", "value", "contacts")}
        tube, p, fh, sp = problem(agent[: -len("_pillar")])
        sp = dataclasses.replace(sp, A_obs=np.concatenate([sp.A_obs, [f[f"{agent}_A"]]]), b_obs=np.concatenate([sp.b_obs, [f[f"{agent}_b"]]]))
    else:
        f = np.load(os.path.join(here, "golden", "colloc_independent_trunc.npz" if agent in TRUNC_AGENTS else "colloc_independent.npz"))
        d = {k: f[f"{agent}_{k}"] for k in ("guess", "traj", "dt", "cost")}
        d["value"] = d["cost"]
        tube, p, fh, sp = problem(agent)
    return d, GeometricColloc(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh), (tube, p, fh, sp)


def check_plan_against_independent(traj, dt, tight, agent="vehicle_1"):
    """Shared by the CPU test (kernel source compiled for the host) and the GPU test (`cfz_colloc`): the plan [N, 6, 7] + dt
    satisfies the GEOMETRIC statement of the reference's rows (ODE at all points, continuity, tube, polygon distances >= dmin,
    boxes; oracle/independent_colloc.py) and is the independent optimum: cost to 1e-6 / poses to 1e-5 m at tight tolerances
    (vehicle_1: 1e-8 / 1e-6); at the reference's tolerance 1e-2 the rows hold to 1e-2, the cost is within 3e-3 (below: the
    rows are relaxed by the tolerance) and the poses within 5 mm.  The `_pillar` plans (corner-to-corner contact active at the
    optimum) are compared with the fixture's `value` (the independent cost corrected to first order for its own row residuals):
    1e-6; poses of vehicle_1_pillar to 1e-4 m (measured 1.4e-5)."""
    d, g, _ = _colloc_fixture(agent)
    z = np.append(np.asarray(traj, float).ravel(), float(dt))
    eq, ineq = np.abs(g.eq(z)).max(), g.ineq(z).min()
    gap = (g.cost(z) - float(d["value"])) / float(d["value"])
    dpose, ddt = np.abs(np.asarray(traj)[..., :3] - d["traj"][..., :3]).max(), abs(float(dt) - float(d["dt"]))
    if tight:
        lim = (1e-8, 1e-6, 1e-8) if agent == "vehicle_1" else ((1e-6, 1e-4, 1e-6) if agent == "vehicle_1_pillar" else (1e-6, 1e-5, 1e-7))
        assert eq < 1e-7 and ineq > -1e-7 and abs(gap) < lim[0] and dpose < lim[1] and ddt < lim[2], (agent, eq, ineq, gap, dpose, ddt)
    elif agent in TRUNC_AGENTS:  # (vehicle 0's truncated plan stops earlier on the central path: measured -0.6 %, 1.1 cm, 1.7e-3 s)
        assert eq < 1e-2 and ineq > -1e-2 and -1e-2 < gap < 1e-4 and dpose < 2e-2 and ddt < 3e-3, (agent, eq, ineq, gap, dpose, ddt)
    else:
        assert eq < 1e-2 and ineq > -1e-2 and -3e-3 < gap < 1e-4 and dpose < 5e-3 and ddt < 1e-3, (agent, eq, ineq, gap, dpose, ddt)
    return gap, dpose


VV_PLANS = ("vehicle_1_pillar", "vehicle_2_pillar", "vehicle_3_pillar")  # tests/golden/make_independent_colloc_vv.py


def _solve_plan_on_cpu(agent, tight, vv=True):
    import colloc_emu_binding as ce
    import test_colloc as tc
    from oracle.colloc_nlp import CollocNlp

    d, g, (tube, p, fh, sp) = _colloc_fixture(agent)
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh, vv=vv)
    X0 = nlp.pack({k: d["guess"][:-1].reshape(-1, 7)[:, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}, float(d["guess"][-1]))
    if tight:
        opt = ipm.IpmOptions(max_iter=800, reg_dual=1e-9, tol=1e-8, constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6)
        opt.no_prox = 1
    else:
        opt = ipm.IpmOptions(**tc.COLLOC_OPT)
    r = ce.solve(nlp, X0, opt)
    P, dt = g.split(r["X"][: nlp.iDt + 1])
    return r, P, dt, d, g


@pytest.mark.parametrize("agent", VV_PLANS)
@pytest.mark.parametrize("tight", [True, False])
def test_corner_to_corner_plan_against_the_independent_solver(tight, agent):
    """Plans whose optimum has an ACTIVE vertex-vertex contact (a pillar's corner against the outer front corner of the turning
    body): the planning kernel's source (CPU build) with vertex-vertex rows, from the fixture's guess, ends at the optimum the
    independent solver found on the geometric statement of the reference's rows (vehicle.py:523-541) -- the kernel's feasible set
    is the reference's there, not a face-normal restriction of it."""
    r, P, dt, d, _ = _solve_plan_on_cpu(agent, tight)
    assert r["status"] == 0 or (tight and agent != "vehicle_1_pillar" and r["status"] in (2, 3)), r["status"]
    assert not tight or r["status"] == 0 or agent == "vehicle_2_pillar"  # vehicle_2 waits at the start: rank loss, as without the pillar
    assert (d["contacts"][:, 2] > 0.1).any()  # the fixture holds such a contact with a multiplier well above zero
    check_plan_against_independent(P, dt, tight, agent)


@pytest.mark.parametrize("agent,least", [("vehicle_1_pillar", 5e-6), ("vehicle_2_pillar", 2e-5), ("vehicle_3_pillar", 1e-3)])
def test_face_normal_rows_alone_restrict_the_collocation_plan(agent, least):
    """The gap `vv_rows = 0` leaves: with face-normal certificates only (round 2's planning kernels) the same instances end feasible
    for the reference's rows but dearer than the optimum -- by 0.0014 %, 0.004 % and 5.2 % (another local optimum) -- because at the corner-to-corner contact
    the best face normal certifies less than the distance of the two corners."""
    r, P, dt, d, g = _solve_plan_on_cpu(agent, True, vv=False)
    z = np.append(P.ravel(), dt)
    assert np.abs(g.eq(z)).max() < 1e-7 and g.ineq(z).min() > -1e-7
    gap = (g.cost(z) - float(d["value"])) / float(d["value"])
    assert gap > least, gap


@pytest.mark.parametrize("agent", COLLOC_AGENTS + TRUNC_AGENTS)
@pytest.mark.parametrize("tight", [True, False])
def test_full_size_collocation_plan_against_the_independent_solver(tight, agent):
    """The planning kernel's source (CPU build) from the fixture's guess against the independent optimum of the geometric
    statement, at tight tolerances (unregularised rows) and at the reference's tolerance, for three vehicles of the synthetic
    strategy.  At tight tolerances the verdict is the comparison: the kernel source ends with status 2 (vehicle_2, 561
    iterations) or 3 (vehicle_3, at iteration 57) AT the optimum -- the unregularised rows lose rank where a vehicle waits."""
    import colloc_emu_binding as ce
    import test_colloc as tc
    from oracle.colloc_nlp import CollocNlp

    d, g, (tube, p, fh, sp) = _colloc_fixture(agent)
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh, vv=True)
    X0 = nlp.pack({k: d["guess"][:-1].reshape(-1, 7)[:, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}, float(d["guess"][-1]))
    if tight:
        opt = ipm.IpmOptions(max_iter=800, reg_dual=1e-9, tol=1e-8, constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6)
        opt.no_prox = 1
    else:
        opt = ipm.IpmOptions(**tc.COLLOC_OPT)
    r = ce.solve(nlp, X0, opt)
    assert r["status"] == 0 or (tight and agent != "vehicle_1" and r["status"] in (2, 3)), r["status"]
    P, dt = g.split(r["X"][: nlp.iDt + 1])
    check_plan_against_independent(P, dt, tight, agent)


@pytest.mark.parametrize("agent", COLLOC_AGENTS + VV_PLANS + TRUNC_AGENTS)
def test_independent_collocation_fixture_is_a_kkt_point(agent):
    """Certificate of the fixtures that needs no solver: at the stored plan the gradient of the cost is a combination of the
    gradients of the equality rows and of the ACTIVE inequality rows and bounds of the geometric statement with multipliers of
    the right sign (bounded least squares: residual 1e-5 of the gradient's size), and every row holds to 2e-8."""
    from scipy.optimize import lsq_linear

    d, g, _ = _colloc_fixture(agent)
    z = np.append(d["traj"].ravel(), float(d["dt"]))
    assert np.abs(g.eq(z)).max() < (2e-8 if agent != "vehicle_1_pillar" else 5e-8) and g.ineq(z).min() > -1e-8 and abs(g.cost(z) - float(d["cost"])) < 1e-9
    act = np.nonzero(g.ineq(z) < 1e-6)[0]
    lo = np.array([b[0] if b[0] is not None else -np.inf for b in g.bounds()])
    hi = np.array([b[1] if b[1] is not None else np.inf for b in g.bounds()])
    at_lo, at_hi = np.nonzero(z - lo < 1e-6)[0], np.nonzero(hi - z < 1e-6)[0]
    Je, Ji = g.eq_jac(z), g.ineq_jac(z)[act]
    Eb = np.zeros((len(at_lo) + len(at_hi), g.n))
    Eb[np.arange(len(at_lo)), at_lo] = 1.0
    Eb[len(at_lo) + np.arange(len(at_hi)), at_hi] = -1.0
    # grad f = Je' a + Ji' b + Eb' c with b, c >= 0 (inequalities are >= 0 rows, so their multipliers pull the cost up)
    A = np.vstack([Je, Ji, Eb]).T
    lb = np.concatenate([np.full(len(Je), -np.inf), np.zeros(len(Ji) + len(Eb))])
    from threadpoolctl import threadpool_limits

    with threadpool_limits(limits=1):  # BVLS makes thousands of small BLAS calls: threads only contend (minutes under pytest-xdist)
        r = lsq_linear(A, g.cost_grad(z), bounds=(lb, np.full(A.shape[1], np.inf)), method="bvls", max_iter=800)
    assert len(act) >= 3 and np.abs(A @ r.x - g.cost_grad(z)).max() < 1e-5 * np.abs(g.cost_grad(z)).max(), (len(act), np.abs(A @ r.x - g.cost_grad(z)).max())


# ---- the joint plan of two vehicles (tests/golden/joint_independent.npz, make_independent_joint.py) ---------------------------
JOINT_AGENTS = ("vehicle_2", "vehicle_3")
# name -> (file, agents, dmin).  "23": the bodies stay 0.12 m apart (pair rows inactive); "23_d20" and "123_d20" (three vehicles, three
# pairs: the shape of the reference's `main`, multi_vehicle_planner.py:605-642) at dmin = 0.2: the bodies of vehicles 2 and 3 are in
# CONTACT at the optimum (active pair rows, multipliers 0.35 / 0.28; tests/golden/make_independent_joint.py --dmin 0.2)
JOINT_FIXTURES = {"23": ("joint_independent.npz", JOINT_AGENTS, 0.05), "23_d20": ("joint_independent_23_d20.npz", JOINT_AGENTS, 0.2),
                  "123_d20": ("joint_independent_123_d20.npz", ("vehicle_1", "vehicle_2", "vehicle_3"), 0.2),
                  # vehicles 0 and 2 on their first six strategy steps (25 intervals each), dmin 0.2: at the optimum a corner of one BODY
                  # touches a corner of the other (vertex-vertex pair rows active at two collocation points, multipliers 0.34 / 0.024)
                  "02_d20_s66": ("joint_independent_02_d20_s66.npz", ("vehicle_0", "vehicle_2"), 0.2, (6, 6)),
                  # ALL FOUR vehicles (six pairs, one shared dt: the shape of BASELINE configs[3]) on their first five strategy steps, dmin 0.2
                  # (round 4; `make_independent_joint.py vehicle_0 .. vehicle_3 --dmin 0.2 --sets 5,5,5,5 --start-near --certify-kernel`):
                  # two pairs of bodies in contact at the optimum, 18 active pair rows, two of them corner against corner.  BOTH solvers
                  # converge tightly (independent: 413 iterations, certificate 2.3e-11; planning source: 250 iterations, 4.0e-11) to costs
                  # within 1.1e-6 -- at points up to 0.14 m apart in one vehicle's poses: the minimiser is not unique there (a vehicle may
                  # wait earlier or later at no cost), so this fixture pins cost and certificate, not poses
                  "0123_d20_s5555": ("joint_independent_0123_d20_s5555.npz", ("vehicle_0", "vehicle_1", "vehicle_2", "vehicle_3"), 0.2, (5, 5, 5, 5))}
VV_BODY = "02_d20_s66"
FOUR = "0123_d20_s5555"


def _joint_fixture(name="23"):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_independent_joint import plans_of_strategy, truncated
    from oracle.independent_colloc import GeometricColloc

    fn, agents, dmin = JOINT_FIXTURES[name][:3]
    f = np.load(os.path.join(here, "golden", fn))
    d = {k: f[k] for k in f.files}
    d.setdefault("value", d["cost"])
    d["agents"], d["dmin_"] = agents, dmin
    plans, sp = plans_of_strategy(), scenarios.parking_lot_spec(dmin=dmin)
    if len(JOINT_FIXTURES[name]) > 3:
        plans = truncated(plans, agents, JOINT_FIXTURES[name][3])
    gs = [GeometricColloc(plans[a][1][0], plans[a][0], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(plans[a][1][-1, 2]), dmin=dmin) for a in agents]
    return d, gs, plans, sp


def check_joint_against_independent(trajs, dt, tight, name="23"):
    """Shared by the CPU test (kernel source compiled for the host) and the GPU test (`cfz_joint_colloc`): both vehicles' plans
    [N_a, 6, 7] on the shared dt satisfy the GEOMETRIC statement of the reference's rows (each vehicle's own rows as in
    check_plan_against_independent, plus polygon distance >= dmin between the two bodies at every common collocation point;
    oracle/independent_joint.py) and are the independent optimum: cost to 1e-6, poses to 5e-5 m at tight tolerances (measured
    2e-8 / 8e-6 on the CPU, 3e-8 / 2.3e-5 on the GPU); at the reference's tolerance 1e-2: rows to 1e-2, cost within 1 % (measured -0.77 %: below, the rows are relaxed
    by the tolerance), poses within 1 cm (6 mm), dt within 2e-3 s (1.3e-3)."""
    from oracle.independent_joint import GeometricJointIpm

    d, gs, _, _ = _joint_fixture(name)
    V = len(gs)
    pairs = [(a, b) for a in range(V) for b in range(a + 1, V)]
    z = np.concatenate([np.asarray(t, float).ravel() for t in trajs] + [[float(dt)]])
    nlp = GeometricJointIpm(gs, pairs, z)
    eq = max(np.abs(g.eq(nlp.z_of(z, a))).max() for a, g in enumerate(gs))
    ineq = min(min(g.ineq(nlp.z_of(z, a)).min() for a, g in enumerate(gs)), min(nlp.pair_dist(z, a, b).min() for a, b in pairs) - nlp.dmin)
    gap = (nlp.f(z) - float(d["value"])) / float(d["value"])
    dpose = max(np.abs(np.asarray(trajs[a])[..., :3] - d[f"traj{a}"][..., :3]).max() for a in range(V))
    ddt = abs(float(dt) - float(d["dt"]))
    if tight and name == VV_BODY:
        # The independent solver stops short of its tolerance on this instance (line search at the rounding floor of its
        # finite-difference distance gradients: certificate 8.7e-5, rows 2e-8), so the comparison with it is looser (measured: cost
        # 7e-8, poses 1.1e-4 m, dt 1.5e-6 s) -- and the verdict on the plan handed in is the SOLVER-FREE certificate on the independent
        # statement: its rows hold and the cost gradient is a combination of the gradients of the equality rows and of the active
        # inequality rows with multipliers of the right sign (1.8e-12 measured), two of them vertex-vertex contacts of the two bodies.
        from make_independent_joint import joint_kkt_certificate, vertex_pair_contacts

        assert eq < 1e-7 and ineq > -1e-7 and abs(gap) < 1e-6 and dpose < 3e-4 and ddt < 1e-5, (eq, ineq, gap, dpose, ddt)
        res, _, _, active = joint_kkt_certificate(nlp, z)
        vv = vertex_pair_contacts(nlp, z, active)
        assert res < 1e-8 and len(vv) >= 1 and max(lam for _, _, lam in vv) > 0.1, (res, active, vv)
    elif tight and name == FOUR:
        # cost to 3e-6 of the independent optimum (measured 1.1e-6), dt to 1e-6 s; poses are not compared (non-unique minimiser, see
        # JOINT_FIXTURES); the verdict on the plan handed in is the solver-free certificate on the independent statement
        from make_independent_joint import joint_kkt_certificate, vertex_pair_contacts

        assert eq < 1e-7 and ineq > -1e-7 and abs(gap) < 3e-6 and ddt < 1e-6 and dpose < 0.3, (eq, ineq, gap, dpose, ddt)
        # the generator certified the planning source's own tight plan (`ktraj*`, residual 4.0e-11, 15 active pair rows in two pairs, two
        # of them corner against corner); a plan within 2e-5 m of it shares that certificate, any other gets its own (2.5 minutes of
        # bounded least squares)
        dk = max(np.abs(np.asarray(trajs[a])[..., :3] - d[f"ktraj{a}"][..., :3]).max() for a in range(V))
        assert float(d["kcertificate"]) < 1e-8 and len(d["kactive"]) >= 8 and len(set(d["kactive"][:, 0])) >= 2 and len(d["kcontacts"]) >= 1
        if not (dk < 2e-5 and abs(nlp.f(z) - float(d["kcost"])) < 1e-8 * float(d["kcost"])):
            res, _, _, active = joint_kkt_certificate(nlp, z)
            vv = vertex_pair_contacts(nlp, z, active)
            # (`cfz_joint_colloc` on the GPU ends next to the OTHER minimiser, the independent solver's -- 0.138 m from `ktraj*`, 18 active
            # pair rows -- at its iteration limit: certificate 1.6e-6 there, with rows to 1e-7 and the cost to 3e-6 asserted above)
            assert res < 1e-5 and len(active) >= 8 and len({e_ for e_, _, _ in active}) >= 2 and len(vv) >= 1, (dk, res, active, vv)
    elif tight:
        assert eq < 1e-7 and ineq > -1e-7 and abs(gap) < 1e-6 and dpose < 5e-5 and ddt < 1e-7, (eq, ineq, gap, dpose, ddt)
    elif name == FOUR:  # measured at the reference's tolerance: -0.31 % (the rows of four plans relaxed by the tolerance)
        assert eq < 1e-2 and ineq > -1e-2 and -1.5e-2 < gap < 1e-4 and ddt < 5e-3 and dpose < 0.3, (eq, ineq, gap, dpose, ddt)
    else:
        # (the contact fixtures at dmin = 0.2: three vehicles -1.03 %, 1.35 cm measured -- the tolerance relaxes three plans' rows)
        # (the corner-to-corner fixture: -1.16 %, 3.2 cm, 3.6e-3 s measured: the relaxed plan cuts the corner closer)
        lim = (1e-2, 1e-2, 2e-3) if name == "23" else ((1.5e-2, 5e-2, 5e-3) if name == VV_BODY else (1.5e-2, 2e-2, 2e-3))
        assert eq < 1e-2 and ineq > -1e-2 and -lim[0] < gap < 1e-4 and dpose < lim[1] and ddt < lim[2], (eq, ineq, gap, dpose, ddt)
    return gap, dpose


@pytest.mark.parametrize("name", sorted(JOINT_FIXTURES))
@pytest.mark.parametrize("tight", [True, False])
def test_joint_plan_against_the_independent_solver(tight, name):
    """The planning kernel's source (CPU build) on the JOINT plan of vehicles 2 and 3 (multi_vehicle_planner.py:343-480: shared dt,
    vehicle-vehicle rows) from the fixture's guess against the optimum the independent solver found on the geometric
    statement: at the reference's tolerance, and at tight tolerances with unregularised rows -- there the verdict is the
    comparison: the kernel source runs into its iteration limit AT the optimum (the rows lose rank where a vehicle waits, as
    for the single plans of these two vehicles)."""
    import colloc_emu_binding as ce
    import test_colloc as tc

    from oracle.colloc_nlp import JointCollocNlp

    d, gs, plans, sp = _joint_fixture(name)
    agents, V = d["agents"], len(gs)
    jn = JointCollocNlp([dict(init_pose=plans[a][1][0], tube=plans[a][0], final_heading=float(plans[a][1][-1, 2])) for a in agents],
                        sp.A_obs, sp.b_obs, N_per_set=5, dmin=d["dmin_"])
    if name != "23":  # the contact fixtures: the pair rows are active at the stored optimum
        assert len(d["active"]) >= 1 and d["active"][:, 2].max() > 0.2
    singles = [{k: d[f"guess{a}"][:, c].reshape(gs[a].N, 6) for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))} for a in range(V)]
    X0 = jn.pack(singles, float(d["dt0"]))
    if tight:
        opt = ipm.IpmOptions(max_iter=800, reg_dual=1e-9, tol=1e-8, constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6)
        opt.no_prox = 1
    else:
        opt = ipm.IpmOptions(**tc.COLLOC_OPT)
    r = ce.solve(jn, X0, opt)
    assert r["status"] == 0 or (tight and r["status"] in (1, 2, 3)), r["status"]
    P = r["X"][: jn.iDt].reshape(-1, 7)
    trajs = [P[6 * jn.off[a]: 6 * jn.off[a + 1]].reshape(-1, 6, 7) for a in range(V)]
    check_joint_against_independent(trajs, r["X"][jn.iDt], tight, name)


def test_joint_face_normal_rows_alone_pay_at_a_body_corner_contact():
    """The corner-to-corner fixture with `vv_rows = 0` (face-normal certificates only: a restriction of the reference's feasible set
    where two vertices are the closest features of the two bodies): the planning source still returns a feasible plan at the
    reference's tolerance, but a dearer one -- above the independent optimum (+0.41 % measured; +1.7 % at tight tolerances), where the
    plan with the vertex-vertex rows lies below it by the relaxation of the tolerance (-1.16 %)."""
    import colloc_emu_binding as ce
    import test_colloc as tc

    from oracle.colloc_nlp import JointCollocNlp
    from oracle.independent_joint import GeometricJointIpm

    d, gs, plans, sp = _joint_fixture(VV_BODY)
    gaps = {}
    for vv in (True, False):
        jn = JointCollocNlp([dict(init_pose=plans[a][1][0], tube=plans[a][0], final_heading=float(plans[a][1][-1, 2])) for a in d["agents"]],
                            sp.A_obs, sp.b_obs, N_per_set=5, dmin=d["dmin_"], vv=vv)
        singles = [{k: d[f"guess{a}"][:, c].reshape(gs[a].N, 6) for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))} for a in range(2)]
        r = ce.solve(jn, jn.pack(singles, float(d["dt0"])), ipm.IpmOptions(**tc.COLLOC_OPT))
        assert r["status"] == 0
        z = r["X"][: jn.iDt + 1]
        nlp = GeometricJointIpm(gs, [(0, 1)], z)
        assert nlp.pair_dist(z, 0, 1).min() - nlp.dmin > -1e-2
        gaps[vv] = nlp.f(z) / float(d["value"]) - 1.0
    assert gaps[True] < 0.0 and gaps[False] > 1e-3, gaps


@pytest.mark.parametrize("name", sorted(JOINT_FIXTURES))
def test_independent_joint_fixture_is_a_kkt_point(name):
    """Certificate of the joint fixture that needs no solver: at the stored plans the gradient of the cost is a combination of the
    gradients of the equality rows and of the ACTIVE inequality rows and bounds of the geometric statement (both vehicles' rows
    and the pair distances) with multipliers of the right sign, and every row holds to 2e-8."""
    from scipy.optimize import lsq_linear
    from threadpoolctl import threadpool_limits

    from oracle.independent_joint import GeometricJointIpm

    d, gs, _, _ = _joint_fixture(name)
    V = len(gs)
    if name == VV_BODY:  # the independent solver's best point, short of its tolerance; the certificate that counts is taken at the
        assert float(d["certificate"]) < 2e-4 and len(d["contacts"]) >= 1  # kernel's plan (check_joint_against_independent)
        return
    if name == FOUR:  # both solvers' points carry the generator's certificate; the kernel source's is recomputed live in
        assert float(d["certificate"]) < 1e-8 and float(d["kcertificate"]) < 1e-8 and int(d["status"]) == 0 and int(d["kstatus"]) == 0  # check_joint_against_independent
        assert abs(float(d["kcost"]) - float(d["value"])) < 3e-6 * float(d["value"]) and len(d["active"]) >= 8 and len(d["contacts"]) >= 1
        return
    if V > 2 or name == "23_d20":  # (three vehicles: 2.5 minutes of bounded least squares; 23_d20: 50 s) the generator ran the same
        assert float(d["certificate"]) < 1e-8 and int(d["status"]) in (0, 1, 2)  # certificate (make_independent_joint.py
        return  # joint_kkt_certificate) and stored its residual; the live computation below runs on the first fixture
    z = np.concatenate([d[f"traj{a}"].ravel() for a in range(V)] + [[float(d["dt"])]])
    nlp = GeometricJointIpm(gs, [(a, b) for a in range(V) for b in range(a + 1, V)], z, prune=0.5)
    X = nlp.initial(z)
    c = nlp.cons(np.concatenate([z, np.zeros(nlp.mi)]))  # equality rows, then inequality rows as values (slack 0)
    assert np.abs(c[: nlp.me]).max() < 2e-8 and c[nlp.me:].min() > -1e-8 and abs(nlp.f(z) - float(d["cost"])) < 1e-9
    act = np.nonzero(c[nlp.me:] < 1e-6)[0]
    J = nlp.jac(X)[:, : nlp.n0].toarray()
    at_lo, at_hi = np.nonzero(z - nlp.xl[: nlp.n0] < 1e-6)[0], np.nonzero(nlp.xu[: nlp.n0] - z < 1e-6)[0]
    Eb = np.zeros((len(at_lo) + len(at_hi), nlp.n0))
    Eb[np.arange(len(at_lo)), at_lo] = 1.0
    Eb[len(at_lo) + np.arange(len(at_hi)), at_hi] = -1.0
    A = np.vstack([J[: nlp.me], J[nlp.me + act], Eb]).T
    lb = np.concatenate([np.full(nlp.me, -np.inf), np.zeros(len(act) + len(Eb))])
    g0 = nlp.grad(X)[: nlp.n0]
    with threadpool_limits(limits=1):
        r = lsq_linear(A, g0, bounds=(lb, np.full(A.shape[1], np.inf)), method="bvls", max_iter=1500)
    assert len(act) >= 3 and np.abs(A @ r.x - g0).max() < 1e-5 * np.abs(g0).max(), (len(act), np.abs(A @ r.x - g0).max())
