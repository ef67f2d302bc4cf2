"""Generates tests/golden/joint_independent.npz (run from the repo root: `python tests/golden/make_independent_joint.py`): the
JOINT collocation plan of vehicles 2 and 3 of the synthetic strategy at the reference's size (30 and 40 intervals of six points,
six obstacles, one shared free dt, vehicle-vehicle rows at the first 30 intervals; multi_vehicle_planner.py:343-480) solved
INDEPENDENTLY of the planning kernel -- oracle/independent_joint.py: polygon distances instead of OBCA duals or working sets, no
condensation, no bordering of dt, derivatives of its own (distance rows by finite differences), scipy's SuperLU on the full KKT
matrix (oracle/ipm.py) -- to tol 1e-8.  Stored: the guess both solvers start from (points per vehicle + dt: the single plans
of the kernel source at the reference's tolerance on their mean dt, as `solve_final_problem_obca` does), the optimal
trajectories, dt and cost."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))
AGENTS = ("vehicle_2", "vehicle_3")  # default: joint_independent.npz; `... vehicle_0 vehicle_2` -> joint_independent_02.npz (a corner of
#   vehicle 0 against a corner of vehicle 2 at dmin: the vertex-vertex pair rows); `... vehicle_0 vehicle_2 vehicle_3` ->
#   joint_independent_023.npz (three vehicles, three pairs: the shape of the reference's `main`, multi_vehicle_planner.py:605-642);
#   all four -> joint_independent_0123.npz (BASELINE configs[3]: four vehicles, six pairs)


def plans_of_strategy():
    from conflict_rez_amd import strategy as strat
    from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
    from conflict_rez_amd.vehicle_types import VehicleBody

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    return {a: ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]], paths[a]) for a in sorted(hist)}


def truncated(plans, agents, sets):
    """The plans of `agents` cut to their first sets[i] strategy steps (tube and path; the final heading is then the path's heading
    there): short joint plans on which the independent solver converges also with vehicle 0."""
    return {a: (plans[a][0][:ns], plans[a][1][: 30 * (ns - 1) + 1]) if ns else plans[a] for a, ns in zip(agents, sets)}


def joint_kkt_certificate(nlp, z):
    """Solver-free certificate of a joint plan z = [points of every vehicle | dt] on oracle/independent_joint.py's statement: bounded
    least squares for multipliers (equality rows free; active inequality rows -- every vehicle's own and the pair distances -- and
    active bounds >= 0) that combine the active gradients into the cost gradient.
    Returns (relative residual, lambda_eq, c_eq, [(pair, point, multiplier)] of the active pair rows)."""
    from scipy.optimize import lsq_linear
    from threadpoolctl import threadpool_limits

    gs, h = nlp.gs, 1e-6
    Je, ce, Ji, Eb = [], [], [], []
    for a, g in enumerate(gs):
        za, cols = nlp.z_of(z, a), nlp.cols(a)
        J = np.zeros((len(g.eq(za)), nlp.n0))
        J[:, cols] = g.eq_jac(za)
        Je.append(J); ce.append(g.eq(za))
        act = np.nonzero(g.ineq(za) < 1e-6)[0]
        if len(act):
            J = np.zeros((len(act), nlp.n0))
            J[:, cols] = g.ineq_jac(za)[act]
            Ji.append(J)
        bd = g.bounds()
        lo = np.array([b[0] if b[0] is not None else -np.inf for b in bd]); hi = np.array([b[1] if b[1] is not None else np.inf for b in bd])
        for i in np.nonzero(za - lo < 1e-6)[0]:
            e = np.zeros(nlp.n0); e[cols[i]] = 1.0; Eb.append(e)
        for i in np.nonzero(hi - za < 1e-6)[0]:
            e = np.zeros(nlp.n0); e[cols[i]] = -1.0; Eb.append(e)
    pair_rows = []
    for e_, (a, b) in enumerate(nlp.pairs):  # active pair distances: gradient by central differences in the six pose variables
        d0 = nlp.pair_dist(z, a, b)
        for r in np.nonzero(d0 - nlp.dmin < 1e-6)[0]:
            row = np.zeros(nlp.n0)
            for veh, which in ((a, 0), (b, 1)):
                for c in range(3):
                    dlt = np.zeros((len(d0), 3)); dlt[r, c] = h
                    dp = nlp.pair_dist(z, a, b, da=dlt if which == 0 else None, db=dlt if which == 1 else None)[r]
                    dm = nlp.pair_dist(z, a, b, da=-dlt if which == 0 else None, db=-dlt if which == 1 else None)[r]
                    row[nlp.off[veh] + 7 * r + c] = (dp - dm) / (2 * h)
            Ji.append(row[None]); pair_rows.append((e_, int(r)))
    Je, ce = np.vstack(Je), np.concatenate(ce)
    Ji = np.vstack(Ji) if Ji else np.zeros((0, nlp.n0))
    Eb = np.array(Eb).reshape(-1, nlp.n0)
    gf = nlp.grad(np.concatenate([z, np.zeros(nlp.mi)]))[: nlp.n0]
    A = np.vstack([Je, Ji, Eb]).T
    lb = np.concatenate([np.full(len(Je), -np.inf), np.zeros(len(Ji) + len(Eb))])
    with threadpool_limits(limits=1):
        r = lsq_linear(A, gf, bounds=(lb, np.full(A.shape[1], np.inf)), method="bvls", max_iter=1500)
    res = float(np.abs(A @ r.x - gf).max() / np.abs(gf).max())
    lam_pairs = r.x[len(Je) + len(Ji) - len(pair_rows): len(Je) + len(Ji)] if pair_rows else np.zeros(0)
    return res, r.x[: len(Je)], ce, [(e_, q, float(l)) for (e_, q), l in zip(pair_rows, lam_pairs)]


def vertex_pair_contacts(nlp, z, active):
    """Of the active pair rows [(pair, point, multiplier)] those whose closest features are two vertices."""
    from oracle.independent_mpc import _body_vertices
    from oracle.mpc_nlp import body_vertices, closest_vertex_pair

    g = nlp.gs[0].g
    out = []
    for e_, q, lam in active:
        a, b = nlp.pairs[e_]
        pa, pb = nlp.poses(z, a, q + 1)[q], nlp.poses(z, b, q + 1)[q]
        Wb = _body_vertices(pb[None, 0], pb[None, 1], pb[None, 2], g)[0]
        if closest_vertex_pair(Wb, pa[:2], pa[2], g, body_vertices(g)) is not None:
            out.append((e_, q, lam))
    return out


if __name__ == "__main__":
    import test_colloc as tc
    from oracle.independent_colloc import GeometricColloc
    from oracle.independent_joint import GeometricJointIpm, solve_joint_ipm

    DMIN = 0.05  # `--dmin 0.2`: a larger clearance (vehicle.py:369 / multi_vehicle_planner.py:343 take it as an argument); with 0.2 the
    #              bodies of vehicles 2 and 3 are in contact (five ACTIVE pair rows) at the optimum -> joint_independent_23_d20.npz
    args = sys.argv[1:]
    if "--dmin" in args:
        DMIN = float(args[args.index("--dmin") + 1])
        del args[args.index("--dmin"): args.index("--dmin") + 2]
    SETS = None  # `--sets 5,5`: every vehicle's first five strategy steps only (20 intervals): `vehicle_0 vehicle_2 --dmin 0.2 --sets 5,5`
    #              -> joint_independent_02_d20_s55.npz, a corner of vehicle 0 against a corner of vehicle 2 at the optimum
    if "--sets" in args:
        SETS = [int(x) for x in args[args.index("--sets") + 1].split(",")]
        del args[args.index("--sets"): args.index("--sets") + 2]
    NEAR = "--start-near" in args  # the independent solver starts from the planning source's plan at the reference's tolerance instead of
    #   from the single plans (which violate a clearance of 0.2 m by 8 cm; its line search fails from there).  What makes the stored
    #   point a fixture is the solver-free certificate below, not the path to it; the tests still start the kernel from `guess*`.
    if NEAR:
        args.remove("--start-near")
    CERTIFY_KERNEL = "--certify-kernel" in args  # also store the planning source's tight plan and its solver-free certificate
    if CERTIFY_KERNEL:
        args.remove("--certify-kernel")
    if len(args) >= 2:
        AGENTS = tuple(args)
    V = len(AGENTS)
    out_name = "joint_independent.npz" if (AGENTS == ("vehicle_2", "vehicle_3") and DMIN == 0.05) else "joint_independent_%s%s%s.npz" % (
        "".join(a[-1] for a in AGENTS), "" if DMIN == 0.05 else "_d%02d" % round(100 * DMIN), "" if SETS is None else "_s" + "".join(map(str, SETS)))
    plans = plans_of_strategy()
    if SETS is not None:
        plans = truncated(plans, AGENTS, SETS)
    jn, sp = tc._joint_problem(plans, list(AGENTS), [0] * V, nps=5)
    X0, _ = tc._joint_guess(plans, list(AGENTS), jn, sp, 5)  # the single plans (at the reference's dmin 0.05) on their mean dt
    guesses = [X0[7 * 6 * jn.off[a]: 7 * 6 * jn.off[a + 1]].reshape(-1, 7) for a in range(V)]
    gs = [GeometricColloc(plans[a][1][0], plans[a][0], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(plans[a][1][-1, 2]), dmin=DMIN) for a in AGENTS]
    pairs = [(a, b) for a in range(V) for b in range(a + 1, V)]
    t0 = time.time()
    starts, dt_start = guesses, float(X0[jn.iDt])
    if NEAR:
        import colloc_emu_binding as ce
        from oracle import ipm
        from oracle.colloc_nlp import JointCollocNlp

        jd = JointCollocNlp([dict(init_pose=plans[a][1][0], tube=plans[a][0], final_heading=float(plans[a][1][-1, 2])) for a in AGENTS],
                            sp.A_obs, sp.b_obs, N_per_set=5, dmin=DMIN)
        rk = ce.solve(jd, X0, ipm.IpmOptions(**tc.COLLOC_OPT))
        assert rk["status"] == 0
        starts, dt_start = [rk["X"][7 * 6 * jd.off[a]: 7 * 6 * jd.off[a + 1]].reshape(-1, 7) for a in range(V)], float(rk["X"][jd.iDt])
    r = solve_joint_ipm(gs, pairs, starts, dt_start)
    print({k: v for k, v in r.items() if k != "trajs"}, "%.0f s" % (time.time() - t0), flush=True)
    z = np.concatenate([t.ravel() for t in r["trajs"]] + [[r["dt"]]])
    res, lam_eq, c_eq, active = joint_kkt_certificate(GeometricJointIpm(gs, pairs, z), z)
    vc = vertex_pair_contacts(GeometricJointIpm(gs, pairs, z), z, active)
    value = r["cost"] - float(lam_eq @ c_eq)  # first-order correction for the residual of the equality rows (make_independent_colloc_vv.py)
    print("certificate %.1e" % res, "active pair rows", active, "vertex-vertex", vc, "value %.9f" % value, flush=True)
    out = dict(dmin=DMIN, dt0=float(X0[jn.iDt]), dt=r["dt"], cost=r["cost"], value=value, iters=r["iters"], status=r["status"], pair=r["pair"],
               contacts=np.array(vc, float).reshape(-1, 3), active=np.array(active, float).reshape(-1, 3), certificate=res)
    if CERTIFY_KERNEL:
        # the planning source's own plan at tight tolerances (CPU build) and ITS solver-free certificate on the independent statement:
        # what the fixture rests on where the independent solver stops short of its tolerance (four vehicles)
        import colloc_emu_binding as ce
        from oracle import ipm
        from oracle.colloc_nlp import JointCollocNlp

        jd = JointCollocNlp([dict(init_pose=plans[a][1][0], tube=plans[a][0], final_heading=float(plans[a][1][-1, 2])) for a in AGENTS],
                            sp.A_obs, sp.b_obs, N_per_set=5, dmin=DMIN)
        opt = ipm.IpmOptions(max_iter=800, reg_dual=1e-9, tol=1e-8, constr_viol_tol=1e-9, compl_inf_tol=1e-9, dual_inf_tol=1e-6)
        opt.no_prox = 1
        rk = ce.solve(jd, X0, opt)
        zk = rk["X"][: jd.iDt + 1]
        nk_ = GeometricJointIpm(gs, pairs, zk)
        kres, _, _, kact = joint_kkt_certificate(nk_, zk)
        kvv = vertex_pair_contacts(nk_, zk, kact)
        print("kernel source: status", rk["status"], "iters", rk["iters"], "cost %.9f" % nk_.f(zk), "certificate %.1e" % kres, "active", len(kact), "vertex-vertex", kvv, flush=True)
        out.update(kdt=float(zk[-1]), kcost=nk_.f(zk), kcertificate=kres, kactive=np.array(kact, float).reshape(-1, 3), kstatus=rk["status"], kiters=rk["iters"],
                   kcontacts=np.array(kvv, float).reshape(-1, 3))
        for a in range(V):
            out[f"ktraj{a}"] = nk_.z_of(zk, a)[:-1].reshape(-1, 6, 7)
    for a in range(V):
        out[f"guess{a}"], out[f"traj{a}"] = guesses[a], r["trajs"][a]
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    assert (r["status"] in (0, 1, 2) and r["eq"] < 5e-8 and r["ineq"] > -1e-8 and res < 1e-7) or (CERTIFY_KERNEL and out["kcertificate"] < 1e-8)
