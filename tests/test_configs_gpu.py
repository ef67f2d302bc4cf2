"""BASELINE.json configs[1] and configs[3] on the GPU planning kernels, at their named sizes, through the C ABI.

configs[1]: B = 256 independent single-vehicle OBCA plans (`Vehicle.state_ws` -> `dual_ws` -> collocation plan,
            reference vehicle.py:99-661) in one launch each of `cfz_state_ws` and `cfz_colloc`.
configs[3]: the centralised plan of the reference itself -- FOUR vehicles, all six pairs, one shared dt
            (`MultiVehiclePlanner.solve_final_problem_obca`, multi_vehicle_planner.py:343-480, pairs :56-58) -- and a batch of
            them in one launch of `cfz_joint_colloc`.

The acceptance check is solver-independent: the reference's own rows, restated as plain loops in
oracle/colloc_nlp.py (`reference_residuals` per vehicle, vehicle.py:426-638; `pair_residuals` for the vehicle-vehicle rows,
multi_vehicle_planner.py:423-456), evaluated on the returned plans with the OBCA duals the product path rebuilds
(`cfz_dual_ws`, `cfz_joint_dual_ws`).  Tolerances are the reference's (tol = constr_viol_tol = 1e-2, vehicle.py:650-651).
"""
import os
import tempfile

import numpy as np
import pytest

from conflict_rez_amd import scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
from oracle.colloc_nlp import CollocNlp, JointCollocNlp, pair_residuals, reference_residuals

pytestmark = pytest.mark.gpu

TAU = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
KEYS = ("x", "y", "psi", "v", "delta", "a", "w")


@pytest.fixture(scope="module")
def lot():
    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    agents = sorted(hist)
    tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
    otubes = {a: [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in sets[a]] for a in agents}
    return dict(agents=agents, tubes=tubes, otubes=otubes, paths=paths, fh={a: float(paths[a][-1, 2]) for a in agents})


def _guess_of(ws, n_sets, nps=5):
    """interp_ws_for_collocation (vehicle.py:298-358) + dt0 = t_end / N (:388) of a state_ws result [T+1, 7]."""
    N = nps * (n_sets - 1)
    t = 0.1 * np.arange(len(ws))
    ti = (np.arange(N)[:, None] + TAU[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N


def _sol_of(traj, dt, eng):
    """Plan [N, 6, 7] + dt -> the dict `reference_residuals` reads, duals rebuilt on the GPU (`Vehicle.get_solution`'s path)."""
    sol = {k: traj[:, :, c] for c, k in enumerate(KEYS)}
    sol["dt"] = dt
    l, m, _ = eng.dual_ws(traj.reshape(-1, 7)[:, :3])
    sol["l"], sol["m"] = l.reshape(traj.shape[0], 6, -1), m.reshape(traj.shape[0], 6, -1)
    return sol


def _single_plans(lot, who, init, n_obs=6):
    from conflict_rez_amd import engine

    sp = scenarios.parking_lot_spec(n_nbr=0, N=2, n_obs=n_obs)
    tubes, paths, fh = lot["tubes"], lot["paths"], lot["fh"]
    ws = engine.state_ws(init, [tubes[a] for a in who], [paths[a] for a in who], [fh[a] for a in who], shrink_tube=0.5)
    good = [i for i, w in enumerate(ws) if w["status"] == 0]
    gs = {i: _guess_of(ws[i]["traj"], len(tubes[who[i]]) + 1) for i in good}
    rg = engine.colloc(sp, [init[i] for i in good], [tubes[who[i]] for i in good], [gs[i][0] for i in good], [gs[i][1] for i in good],
                       [fh[who[i]] for i in good], max_iter=400)
    return ws, good, dict(zip(good, rg))


def test_config1_256_single_vehicle_plans(lot):
    """configs[1] at B = 256: the four vehicles of the synthetic strategy in turn, start poses scattered by +-3 cm.  Every
    warm start that converges is refined; every refined plan meets the cheap vectorised rows (initial pose, terminal
    v = delta = a = w = 0 and heading, boxes, dt > 0); a sample of 16 plans is put through the reference's full row list."""
    from conflict_rez_amd import engine

    B = 256
    agents = lot["agents"]
    rng = np.random.default_rng(0)
    who = [agents[i % 4] for i in range(B)]
    init = [lot["paths"][a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
    ws, good, plans = _single_plans(lot, who, init)
    # every warm start converges (round 2: 1-4 of 256 ended in the line search; since round 3 a failed line search lowers mu once
    # before giving up, cfz_plan.inl) and every plan is refined
    assert len(good) == B, len(good)
    assert sum(r["status"] == 0 for r in plans.values()) == B
    sp = scenarios.parking_lot_spec()
    bd = sp.bounds
    D = None
    for i, r in plans.items():
        if r["status"] != 0:
            continue
        tr, a = r["traj"], who[i]
        assert tr.shape == (5 * len(lot["tubes"][a]), 6, 7) and 0.01 < r["dt"] < 2.0
        assert np.abs(tr[0, 0, :3] - init[i]).max() < 1e-2 and np.abs(tr[0, 0, 3:]).max() < 1e-2
        if D is None:
            D = CollocNlp(init[i], lot["otubes"][a], sp.A_obs, sp.b_obs).D
        zF = D @ tr[-1]
        assert np.abs(zF[3:]).max() < 1e-2 and abs(zF[2] - lot["fh"][a]) < 1e-2
        for col, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
            assert tr[:, :, col].min() >= bd[2 * j] - 1e-9 and tr[:, :, col].max() <= bd[2 * j + 1] + 1e-9
    eng = engine.Engine(scenarios.parking_lot_spec(n_nbr=0, N=2), max_batch=1)
    checked = 0
    for i in [i for i in sorted(plans) if plans[i]["status"] == 0][:: max(1, len(plans) // 16)][:16]:
        a, r = who[i], plans[i]
        nlp = CollocNlp(init[i], lot["otubes"][a], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=lot["fh"][a])
        rr = reference_residuals(nlp, _sol_of(r["traj"], r["dt"], eng))
        assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9, (i, rr)
        assert abs(rr["cost"] - r["cost"]) < 1e-8 * max(1.0, r["cost"])
        checked += 1
    assert checked == 16
    eng.close()


def test_config1_as_baseline_words_it_four_obstacles(lot):
    """BASELINE.json configs[1] to the letter: B = 256 independent single-vehicle OBCA plans with FOUR polytope obstacles
    (obstacles 0, 1, 3, 4 of the reference's six) on the planning kernels.  Every plan converges; a sample of 16 goes
    through the reference's full row list on the four-obstacle map; and since the two dropped boxes are not in any vehicle's way on
    this strategy, every plan is close to its six-obstacle twin (the rows of the remaining boxes are the same rows)."""
    from conflict_rez_amd import engine

    B = 256
    agents = lot["agents"]
    rng = np.random.default_rng(0)
    who = [agents[i % 4] for i in range(B)]
    init = [lot["paths"][a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
    ws, good, plans = _single_plans(lot, who, init, n_obs=4)
    assert len(good) == B and sum(r["status"] == 0 for r in plans.values()) == B
    sp4 = scenarios.parking_lot_spec(n_obs=4)
    assert sp4.n_obs == 4
    eng = engine.Engine(scenarios.parking_lot_spec(n_nbr=0, N=2, n_obs=4), max_batch=1)
    for i in sorted(plans)[:: B // 16][:16]:
        a, r = who[i], plans[i]
        nlp = CollocNlp(init[i], lot["otubes"][a], sp4.A_obs, sp4.b_obs, N_per_set=5, final_heading=lot["fh"][a])
        rr = reference_residuals(nlp, _sol_of(r["traj"], r["dt"], eng))
        assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9, (i, rr)
        assert abs(rr["cost"] - r["cost"]) < 1e-8 * max(1.0, r["cost"])
    eng.close()
    _, _, plans6 = _single_plans(lot, who[:8], init[:8])
    for i in range(8):
        assert abs(plans[i]["cost"] - plans6[i]["cost"]) < 2e-2 * plans6[i]["cost"] and np.abs(plans[i]["traj"][..., :2] - plans6[i]["traj"][..., :2]).max() < 0.1, i


def test_config3_four_vehicle_joint_plans(lot):
    """configs[3] in the reference's own shape: four vehicles, six pairs, one shared dt.  One plan from the nominal start
    poses plus seven with the start poses scattered by +-3 cm, all in ONE launch (one workgroup per plan).  Every plan
    converges; every vehicle's rows and all six pairs' rows hold in the reference's layout; all vehicles of a plan run on
    the shared dt; the joint cost is not below the sum of the single-vehicle optima (those ignore the other vehicles)."""
    from conflict_rez_amd import engine

    agents = lot["agents"]
    B = 8
    rng = np.random.default_rng(1)
    who = [a for _ in range(B) for a in agents]
    init = [lot["paths"][a][0] + (np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] if i >= 4 else 0.0) for i, a in enumerate(who)]
    ws, good, plans = _single_plans(lot, who, init)
    assert len(good) == len(who) and all(r["status"] == 0 for r in plans.values())
    scen = []
    for b in range(B):
        sing = [plans[4 * b + i] for i in range(4)]
        scen.append(dict(init_poses=[init[4 * b + i] for i in range(4)], tubes=[lot["tubes"][a] for a in agents],
                         guesses=[s["traj"].reshape(-1, 7) for s in sing], dt0=float(np.mean([s["dt"] for s in sing])),  # :360
                         final_headings=[lot["fh"][a] for a in agents]))
    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    rj = engine.joint_colloc_batch(sp0, scen, max_iter=300)
    assert [r["status"] for r in rj] == [0] * B, [(r["status"], r["iters"]) for r in rj]
    sp = scenarios.parking_lot_spec()
    eng = engine.Engine(sp0, max_batch=1)
    for b, r in enumerate(rj):
        jn = JointCollocNlp([dict(init_pose=init[4 * b + i], tube=lot["otubes"][a], final_heading=lot["fh"][a]) for i, a in enumerate(agents)],
                            sp.A_obs, sp.b_obs, N_per_set=5)
        assert len(jn.pairs) == 6
        sols = [_sol_of(r["traj"][i], r["dt"], eng) for i in range(4)]
        cost = 0.0
        for i in range(4):
            rr = reference_residuals(jn, sols[i], i)
            assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9, (b, i, rr)
            cost += rr["cost"]
        assert abs(cost - r["cost"]) < 1e-8 * r["cost"]
        singles_cost = sum(plans[4 * b + i]["cost"] for i in range(4))
        assert cost >= singles_cost * (1 - 1e-2), (cost, singles_cost)
        duals = []
        for (ia, ib) in jn.pairs:  # pair duals from the product path (MultiVehiclePlanner.joint_dual_ws's kernel)
            nmin = min(jn.N[ia], jn.N[ib])
            pa, pb = r["traj"][ia][:nmin].reshape(-1, 7)[:, :3], r["traj"][ib][:nmin].reshape(-1, 7)[:, :3]
            lam, mu, s, d = eng.joint_dual_ws(pa, pb)
            assert d.min() > sp.dmin - 1e-2, (b, ia, ib, d.min())
            duals.append(dict(lam=lam.reshape(nmin, 6, 4), mu=mu.reshape(nmin, 6, 4), s=s.reshape(nmin, 6, 2)))
        pr = pair_residuals(jn, sols, duals)
        assert pr["eq"] < 1e-9 and pr["ineq"] < 1e-2 and pr["bound"] == 0.0, (b, pr)
    # the batch entry equals its own single call (instances are independent)
    r1 = engine.joint_colloc(sp0, scen[3]["init_poses"], scen[3]["tubes"], scen[3]["guesses"], scen[3]["dt0"], scen[3]["final_headings"], max_iter=300)
    assert (r1["status"], r1["iters"]) == (rj[3]["status"], rj[3]["iters"]) and r1["cost"] == rj[3]["cost"]
    eng.close()


def test_config3_at_batch_256(lot):
    """configs[3] at its named batch: 256 four-vehicle scenarios (start poses scattered by +-3 cm), their 1024 single plans in
    one launch each of `cfz_state_ws` / `cfz_colloc`, then all joint plans in ONE launch of `cfz_joint_colloc`.  Every joint plan
    whose four single plans converged converges; the reference's rows (per vehicle, and the six pairs through the product
    path's duals) are checked on a sample of four plans."""
    from conflict_rez_amd import engine

    agents, B = lot["agents"], 256
    rng = np.random.default_rng(1)
    who = [a for _ in range(B) for a in agents]
    init = [lot["paths"][a][0] + (np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] if i >= 4 else 0.0) for i, a in enumerate(who)]
    ws, good, plans = _single_plans(lot, who, init)
    ok = [b for b in range(B) if all(4 * b + i in plans and plans[4 * b + i]["status"] == 0 for i in range(4))]
    assert len(ok) >= B - 2
    scen = []
    for b in ok:
        sing = [plans[4 * b + i] for i in range(4)]
        scen.append(dict(init_poses=[init[4 * b + i] for i in range(4)], tubes=[lot["tubes"][a] for a in agents],
                         guesses=[s["traj"].reshape(-1, 7) for s in sing], dt0=float(np.mean([s["dt"] for s in sing])),
                         final_headings=[lot["fh"][a] for a in agents]))
    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    rj = engine.joint_colloc_batch(sp0, scen, max_iter=300)
    assert [r["status"] for r in rj] == [0] * len(scen), [(i, r["status"], r["iters"]) for i, r in enumerate(rj) if r["status"]]
    sp = scenarios.parking_lot_spec()
    eng = engine.Engine(sp0, max_batch=1)
    for q in range(0, len(scen), max(1, len(scen) // 4))[:4]:
        b, r = ok[q], rj[q]
        jn = JointCollocNlp([dict(init_pose=init[4 * b + i], tube=lot["otubes"][a], final_heading=lot["fh"][a]) for i, a in enumerate(agents)],
                            sp.A_obs, sp.b_obs, N_per_set=5)
        sols = [_sol_of(r["traj"][i], r["dt"], eng) for i in range(4)]
        for i in range(4):
            rr = reference_residuals(jn, sols[i], i)
            assert rr["eq"] < 1e-2 and rr["ineq"] < 1e-2 and rr["bound"] <= 1e-9, (b, i, rr)
        duals = []
        for (ia, ib) in jn.pairs:
            nmin = min(jn.N[ia], jn.N[ib])
            pa, pb = r["traj"][ia][:nmin].reshape(-1, 7)[:, :3], r["traj"][ib][:nmin].reshape(-1, 7)[:, :3]
            lam, mu, s_, d = eng.joint_dual_ws(pa, pb)
            assert d.min() > sp.dmin - 1e-2
            duals.append(dict(lam=lam.reshape(nmin, 6, 4), mu=mu.reshape(nmin, 6, 4), s=s_.reshape(nmin, 6, 2)))
        pr = pair_residuals(jn, sols, duals)
        assert pr["eq"] < 1e-9 and pr["ineq"] < 1e-2 and pr["bound"] == 0.0, (b, pr)
    eng.close()


def test_planner_surface_four_vehicles(tmp_path):
    """`MultiVehiclePlanner` as the reference's `main` drives it (multi_vehicle_planner.py:609-668) with all four agents:
    solve_single_problems -> joint_dual_ws -> solve_final_problem_obca; results on the common clock never overlap."""
    from conflict_rez_amd.control.multi_vehicle_planner import MultiVehiclePlanner
    from conflict_rez_amd.pytypes import VehicleState

    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, strat.generate_strategy(4))
    agents = ["vehicle_%d" % i for i in range(4)]
    paths = interp_along_sets(fn, VehicleBody(), 30)
    # ws_config as in the reference's main (multi_vehicle_planner.py:610-615): no spline guess for vehicle_0
    mvp = MultiVehiclePlanner(fn, {a: a != "vehicle_0" for a in agents}, {a: {"front": (1, 0, 0), "back": (0, 0, 1)} for a in agents},
                              {a: VehicleState() for a in agents}, {a: float(paths[a][-1, 2]) for a in agents})
    mvp.solve_single_problems()
    mvp.joint_dual_ws(K=5)
    mvp.solve_final_problem_obca()
    assert mvp.final_stats["status"] == 0 and 0.01 < mvp.final_dt < 2.0
    fr = mvp.final_results
    n_max = max(mvp.vehicles[a].N for a in agents)
    assert all(len(fr[a].x) == n_max * 6 + 1 for a in agents)
    g = np.array([3.3, 0.9, 0.6, 0.9])
    corners = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])

    def poly(p, i):
        c, s = np.cos(p.psi[i]), np.sin(p.psi[i])
        return np.array([p.x[i], p.y[i]]) + corners @ np.array([[c, s], [-s, c]])

    def separated(P, Q):
        for poly_ in (P, Q):
            for a_, b_ in zip(poly_, np.roll(poly_, -1, 0)):
                n = np.array([b_[1] - a_[1], a_[0] - b_[0]])
                if (P @ n).max() < (Q @ n).min() or (Q @ n).max() < (P @ n).min():
                    return True
        return False

    for i in range(0, n_max * 6 + 1, 2):
        for ia in range(4):
            for ib in range(ia + 1, 4):
                assert separated(poly(fr[agents[ia]], i), poly(fr[agents[ib]], i)), (i, ia, ib)


@pytest.mark.parametrize("agent", ["vehicle_1", "vehicle_2", "vehicle_3", "vehicle_1_pillar", "vehicle_2_pillar", "vehicle_3_pillar", "vehicle_0_s6"])
def test_collocation_plan_against_the_independent_solver_on_gpu(agent):
    """`cfz_colloc` (HIP, through the C ABI) from the fixture's guess against the optimum an INDEPENDENT solver found on an
    independent statement of the reference's single-vehicle plan (tests/golden/colloc_independent.npz: polygon distances, no
    working sets, no condensation, SuperLU on the full KKT system; vehicle.py:360-661, N = 30 / 30 / 40 intervals, free dt) --
    not against the CPU compile of the kernel's own source.  At the reference's tolerance 1e-2: rows of the geometric statement to
    1e-2, cost within 3e-3 of the optimum, poses within 5 mm, dt within 1e-3 s; at tight tolerances (`exact_rows`) the optimum
    itself.  Assertions shared with the CPU test (tests/test_independent_solver.py:check_plan_against_independent).
    The `_pillar` plans (tests/golden/colloc_independent_vv.npz) carry a seventh obstacle whose corner is in contact with a corner of
    the body at the optimum, multiplier 0.5 ... 23: there the kernel's vertex-vertex rows (`vv_rows`, the default) are what makes
    its feasible set the reference's (vehicle.py:523-541); with `vv_rows = 0` the same launch ends feasible but dearer.
    `vehicle_0_s6` (tests/golden/colloc_independent_trunc.npz): vehicle 0 on its first six strategy steps -- the longest plan's vehicle
    (at full length, 50 intervals of which it waits for 20, neither solver has a tight optimum)."""
    import dataclasses

    from conflict_rez_amd import engine
    from test_independent_solver import _colloc_fixture, check_plan_against_independent

    d, g, (tube, p, fh, sp) = _colloc_fixture(agent)
    tb = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in tube[1:]]
    guess = d["guess"][:-1].reshape(-1, 7)
    spec = dataclasses.replace(scenarios.parking_lot_spec(n_nbr=0, N=2), A_obs=sp.A_obs, b_obs=sp.b_obs)
    r = engine.colloc(spec, [p[0]], [tb], [guess], [float(d["guess"][-1])], [fh], max_iter=400)[0]
    assert r["status"] == 0 and r["iters"] < 60
    check_plan_against_independent(r["traj"], r["dt"], False, agent)
    if agent.endswith("_pillar"):  # the gap face-normal certificates leave (CPU twin: test_face_normal_rows_alone_restrict_...)
        r0 = engine.colloc(spec, [p[0]], [tb], [guess], [float(d["guess"][-1])], [fh], max_iter=800, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1, vv_rows=0)[0]
        z0 = np.append(r0["traj"].ravel(), r0["dt"])
        assert np.abs(g.eq(z0)).max() < 1e-7 and g.ineq(z0).min() > -1e-7 and (g.cost(z0) - float(d["value"])) / float(d["value"]) > 5e-6
    # the same kernel at tight tolerances with IPOPT's form of the dual regularisation (`exact_rows`): the independent optimum
    # itself (vehicle_1: status 0, cost to 1e-8, poses to 1e-6 m; vehicles 2, 3: the unregularised rows lose rank where the vehicle
    # waits and the solve ends with status 2 / 3 at the optimum, cost to 1e-6, poses to 1e-5 m)
    # (iteration limit 1200: at these tolerances the plans with a waiting vehicle converge linearly -- vehicle_2_pillar takes 780 iterations
    # with the band elimination, 801 with the structured one, 560 / 654 on the CPU build: tools/pillar_tight.py)
    r2 = engine.colloc(spec, [p[0]], [tb], [guess], [float(d["guess"][-1])], [fh], max_iter=1200, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)[0]
    assert r2["status"] == 0 or (agent not in ("vehicle_1", "vehicle_1_pillar") and r2["status"] in (2, 3)), (r2["status"], r2["iters"])
    check_plan_against_independent(r2["traj"], r2["dt"], True, agent)


@pytest.mark.parametrize("name", ["23", "23_d20", "123_d20", "02_d20_s66", "0123_d20_s5555"])
def test_joint_plan_against_the_independent_solver_on_gpu(name):
    """`cfz_joint_colloc` (HIP, through the C ABI) on the joint plan of vehicles 2 and 3 from the fixture's guess against the optimum
    an INDEPENDENT solver found on an independent statement of `solve_final_problem_obca` (tests/golden/joint_independent.npz:
    polygon distances between bodies and obstacles and between the two bodies, no working sets, no condensation, no band, SuperLU on
    the full KKT system; multi_vehicle_planner.py:343-480, 30 + 40 intervals, one shared free dt) -- not against the CPU compile of
    the kernel's own source.  At the reference's tolerance: rows of the geometric statement to 1e-2, cost within 1 %, poses within
    1 cm; at tight tolerances (`exact_rows`) the optimum itself, cost to 1e-6, poses to 5e-5 m.  Assertions shared with the CPU
    test (tests/test_independent_solver.py:check_joint_against_independent).
    `23_d20`, `123_d20` (dmin = 0.2; the latter three vehicles and three pairs, the shape of the reference's `main`,
    multi_vehicle_planner.py:605-642): the bodies of vehicles 2 and 3 are in CONTACT at the optimum -- pair rows active with
    multipliers 0.35 / 0.28 -- so the vehicle-vehicle rows decide the plan there.
    `02_d20_s66` (vehicles 0 and 2 on their first six strategy steps): a CORNER of one body touches a CORNER of the other at the
    optimum (vertex-vertex pair rows, multipliers 0.34 / 0.024); there the plan also carries a solver-free KKT certificate on the
    independent statement (1.8e-12).
    `0123_d20_s5555` (round 4): ALL FOUR vehicles, six pairs, on their first five strategy steps -- BASELINE configs[3]'s shape -- with
    two pairs of bodies in contact; both solvers converge tightly to the same cost (1.1e-6) at points that differ in one vehicle's
    poses (non-unique minimiser), so cost and the solver-free certificate at the kernel's own plan are what is asserted."""
    from conflict_rez_amd import engine
    from test_independent_solver import _joint_fixture, check_joint_against_independent

    d, gs, plans, sp = _joint_fixture(name)
    agents = d["agents"]
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    guesses = [d[f"guess{a}"] for a in range(len(agents))]
    args = (scenarios.parking_lot_spec(n_nbr=0, N=2, dmin=d["dmin_"]), [plans[a][1][0] for a in agents], tubes, guesses, float(d["dt0"]),
            [float(plans[a][1][-1, 2]) for a in agents])
    r = engine.joint_colloc(*args, max_iter=400)
    # (52 iterations on the CPU build for the corner contact; the four-vehicle plan with two pairs of bodies in contact: 80 there, 95 here)
    assert r["status"] == 0
    # iteration counts are asserted where they are a property of the problem (equal on every build so far).  The four-vehicle fixture is a
    # plan that wanders between two minimisers -- 80 to 285 iterations depending on the build's roundings (docs/notebook.md) -- : there the
    # count is a draw and NOT asserted (round 5 wrote "< 400" beside max_iter = 400, which asserted nothing: ADVICE r5); what is asserted
    # for it is what the others get too: convergence within the limit, cost and rows against the independent optimum, the certificate
    if name != "0123_d20_s5555":
        assert r["iters"] < {"02_d20_s66": 80}.get(name, 60), r["iters"]
    check_joint_against_independent(r["traj"], r["dt"], False, name)
    r2 = engine.joint_colloc(*args, max_iter=800, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)
    assert r2["status"] in (0, 1, 2, 3), r2["status"]  # ends AT the optimum with the iteration limit or the line search exhausted
    check_joint_against_independent(r2["traj"], r2["dt"], True, name)


def test_full_length_four_vehicle_plan_ends_at_a_certified_optimum():
    """BASELINE configs[3]'s problem at FULL length (50 / 30 / 30 / 40 intervals, six pairs, dmin = 0.2: two pairs of bodies in contact) --
    the one problem of this path no independent solver converges on.  The fixture
    (tests/golden/make_joint_full_certificate.py from a dump of tools/joint_full_tight.py) holds the kernel's plan at TIGHT tolerances and
    what a solver-free check on the INDEPENDENT statement (oracle/independent_joint.py: polygon distances, no working sets) says about it:
    every row holds (equalities 2.4e-7, inequalities 4e-12) and the cost gradient is a combination of the gradients of the equality rows
    and of the ACTIVE inequality rows with multipliers of the right sign to 1.8e-7 (relative) -- eight active pair rows in three pairs, two
    of them corner against corner.  Here: the kernel, from the fixture's guess, ends at that cost again (1e-7 relative; rows to 1e-6 on the
    independent statement; the line search exhausted or the iteration limit AT the optimum, as in the other tight runs), and at the
    reference's tolerance at a plan whose rows hold to 1e-2 and whose cost lies within 1.5 % below (the rows relaxed by the tolerance)."""
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from conflict_rez_amd import engine
    from make_independent_joint import plans_of_strategy
    from make_joint_full_certificate import rows_of, statement
    from oracle.independent_joint import GeometricJointIpm

    d = np.load(os.path.join(here, "golden", "joint_kernel_0123_d20_full.npz"))
    assert float(d["kcertificate"]) < 1e-6 and d["keq"] < 1e-6 and d["kineq"] > -1e-7 and len(d["kactive"]) >= 6 and len(set(d["kactive"][:, 0])) >= 2 and int(d["kcontacts"]) >= 1
    assert (d["kactive"][:, 2] >= 0.0).all() and (d["kactive"][:, 2] > 0.1).sum() >= 6  # multipliers of the right sign, seven of the eight rows really pushing
    dmin = float(d["dmin"])
    plans = plans_of_strategy()
    agents = sorted(plans)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    args = (scenarios.parking_lot_spec(n_nbr=0, N=2, dmin=dmin), [plans[a][1][0] for a in agents], tubes, [d[f"guess{a}"] for a in range(4)], float(d["dt0"]),
            [float(plans[a][1][-1, 2]) for a in agents])
    gs, pairs = statement(dmin)

    def on_the_independent_statement(r):
        z = np.concatenate([np.asarray(t, float).ravel() for t in r["traj"]] + [[float(r["dt"])]])
        nlp = GeometricJointIpm(gs, pairs, z)
        return float(nlp.f(z)), rows_of(nlp, gs, pairs, z)

    r1 = engine.joint_colloc(*args, max_iter=400)
    f1, (eq1, ineq1) = on_the_independent_statement(r1)
    assert r1["status"] == 0 and eq1 < 1e-2 and ineq1 > -1e-2 and -1.5e-2 < (f1 - float(d["kcost"])) / float(d["kcost"]) < 1e-4, (r1["status"], eq1, ineq1, f1)
    r2 = engine.joint_colloc(*args, max_iter=3000, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)
    f2, (eq2, ineq2) = on_the_independent_statement(r2)
    assert r2["status"] in (0, 1, 2) and eq2 < 1e-6 and ineq2 > -1e-6 and abs(f2 - float(d["kcost"])) < 1e-7 * float(d["kcost"]), (r2["status"], r2["iters"], eq2, ineq2, f2, float(d["kcost"]))


def test_vehicle_0_at_full_length_against_its_certificate():
    """Vehicle 0's single plan at FULL length (50 intervals: the longest plan; vehicle.py:360-661) -- the single plan no solver reaches a
    1e-8 optimum on.  The fixture (tests/golden/make_single_full_certificate.py from a dump of tools/single_full_tight.py) holds what the
    kernel reaches at tight tolerances (2,208 iterations, then a singular Newton system: status 3) and what the solver-free check on the
    INDEPENDENT statement says there: rows to 1.5e-6, the cost gradient a combination of the active gradients with multipliers of the
    right sign to 1.3e-5 (13 active rows) -- a weaker certificate than the other fixtures' (1e-8 ... 1e-12), asserted as what it is.  Here:
    from the fixture's guess the kernel ends there again (cost to 1e-7, rows to 1e-5), and at the reference's tolerance converges in ~32
    iterations to a plan with rows to 1e-2 whose cost lies within 1 % below (0.58 % measured: the rows relaxed by the tolerance)."""
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from conflict_rez_amd import engine
    from make_single_full_certificate import statement

    d = np.load(os.path.join(here, "golden", "colloc_kernel_v0_full.npz"))
    assert float(d["kcertificate"]) < 1e-4 and d["keq"] < 1e-5 and d["kineq"] > -1e-6 and len(d["kactive"]) >= 8 and (d["kactive"][:, 1] >= 0.0).all()
    agent = str(d["agent"])
    g, plans = statement(agent)
    tube = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[agent][0][1:]]
    p = plans[agent][1]
    args = (scenarios.parking_lot_spec(n_nbr=0, N=2), [p[0]], [tube], [d["guess"]], [float(d["dt0"])], [float(p[-1, 2])])

    def on_the_independent_statement(r):
        z = np.append(np.asarray(r["traj"], float).ravel(), float(r["dt"]))
        return float(g.cost(z)), float(np.abs(g.eq(z)).max()), float(g.ineq(z).min())

    r1 = engine.colloc(*args, max_iter=400)[0]
    f1, eq1, ineq1 = on_the_independent_statement(r1)
    assert r1["status"] == 0 and r1["iters"] < 60 and eq1 < 1e-2 and ineq1 > -1e-2 and -1e-2 < (f1 - float(d["kcost"])) / float(d["kcost"]) < 1e-4, (r1["status"], r1["iters"], eq1, ineq1, f1)
    r2 = engine.colloc(*args, max_iter=3000, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)[0]
    f2, eq2, ineq2 = on_the_independent_statement(r2)
    assert eq2 < 1e-5 and ineq2 > -1e-6 and abs(f2 - float(d["kcost"])) < 1e-7 * float(d["kcost"]), (r2["status"], r2["iters"], eq2, ineq2, f2, float(d["kcost"]))


def test_joint_plan_does_not_depend_on_the_order_of_its_vehicles():
    """A property the reference's joint NLP has by construction (multi_vehicle_planner.py:343-480 loops over the agents and over
    the pairs; no vehicle is special): listing the vehicles in another order gives the same plans and the same shared dt.  The
    kernel interleaves the vehicles' unknowns in the band by that order, so the elimination runs differently: equal to the solver's
    tolerance, not to rounding."""
    from conflict_rez_amd import engine
    from test_independent_solver import _joint_fixture

    d, gs, plans, sp = _joint_fixture("123_d20")
    agents = list(d["agents"])
    tubes = {a: [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents}
    guess = {a: d[f"guess{i}"] for i, a in enumerate(agents)}
    spec = scenarios.parking_lot_spec(n_nbr=0, N=2, dmin=d["dmin_"])
    res = {}
    for order in (agents, agents[::-1]):
        r = engine.joint_colloc(spec, [plans[a][1][0] for a in order], [tubes[a] for a in order], [guess[a] for a in order], float(d["dt0"]),
                                [float(plans[a][1][-1, 2]) for a in order], max_iter=400)
        assert r["status"] == 0
        res[tuple(order)] = (r, {a: r["traj"][i] for i, a in enumerate(order)})
    (ra, ta), (rb, tb_) = res[tuple(agents)], res[tuple(agents[::-1])]
    assert abs(ra["cost"] - rb["cost"]) < 2e-3 * ra["cost"] and abs(ra["dt"] - rb["dt"]) < 1e-3
    assert max(np.abs(ta[a][..., :3] - tb_[a][..., :3]).max() for a in agents) < 2e-2


def test_panel_elimination_equals_one_pivot_at_a_time(lot):
    """The joint plan's band is eliminated a panel of sixteen pivots at a time and both right-hand sides are substituted in one
    sweep (cfz_colloc.inl: band_factor_panel, band_substitute_regs); `one_pivot = 1` takes the one-pivot elimination and the
    LDS-resident substitution they replaced.  Same pivots and the same operations per entry in the same order: the four-vehicle
    plan comes out with the same iteration count and the same numbers."""
    from conflict_rez_amd import engine

    agents = lot["agents"]
    ws, good, plans = _single_plans(lot, agents, [lot["paths"][a][0] for a in agents])
    assert good == list(range(4))
    args = (scenarios.parking_lot_spec(n_nbr=0, N=2), [lot["paths"][a][0] for a in agents], [lot["tubes"][a] for a in agents],
            [plans[i]["traj"].reshape(-1, 7) for i in range(4)], float(np.mean([plans[i]["dt"] for i in range(4)])), [lot["fh"][a] for a in agents])
    r = engine.joint_colloc(*args, max_iter=300, structured=0)
    r1 = engine.joint_colloc(*args, max_iter=300, one_pivot=1)
    assert r["status"] == r1["status"] == 0 and r["iters"] == r1["iters"]
    assert r["cost"] == r1["cost"] and r["dt"] == r1["dt"] and all(np.array_equal(a, b) for a, b in zip(r["traj"], r1["traj"]))
    # ... and the structured elimination (cfz_jstruct.inl, the default: interval by interval, no band across the vehicles) is another
    # elimination order of the same Newton systems: same status and iteration count, the plan to rounding
    rs = engine.joint_colloc(*args, max_iter=300)
    assert (rs["status"], rs["iters"]) == (r["status"], r["iters"]) and abs(rs["cost"] - r["cost"]) < 1e-9 * r["cost"] and abs(rs["dt"] - r["dt"]) < 1e-9
    assert max(np.abs(a - b).max() for a, b in zip(rs["traj"], r["traj"])) < 1e-7
