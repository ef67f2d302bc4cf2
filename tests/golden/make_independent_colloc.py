"""Generates tests/golden/colloc_independent.npz (run from the repo root: `python tests/golden/make_independent_colloc.py`,
about 3 min): the single-vehicle collocation plans of vehicles 1, 2, 3 of the synthetic strategy at the reference's size (N = 30,
30, 40 intervals of six collocation points, six obstacles, free dt; vehicle.py:360-661) solved INDEPENDENTLY of the planning kernel --
oracle/independent_colloc.py: polygon distances instead of OBCA duals or working sets, no condensation, no bordering of dt,
derivatives of its own (distance rows by finite differences), scipy's SuperLU on the full KKT matrix (oracle/ipm.py) -- to
tol 1e-8.  Stored per vehicle: the guess both solvers start from (points + dt), the optimal trajectory, dt and cost."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))


def problem(agent="vehicle_1"):
    """(tube, path, final heading, obstacle spec) of one vehicle of the synthetic strategy.  `vehicle_0_s6`: the vehicle's first six
    strategy steps only (tube and path cut there; the final heading is the path's heading at the cut) -- vehicle 0 at full length
    (N = 50, it waits for 20 intervals) has two local solutions and no tight optimum, docs/notebook.md."""
    n_sets = None
    if "_s" in agent:
        agent, n_sets = agent.rsplit("_s", 1)[0], int(agent.rsplit("_s", 1)[1])
    from conflict_rez_amd import scenarios, strategy as strat
    from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
    from conflict_rez_amd.vehicle_types import VehicleBody

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    tube = [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[agent]]
    p = paths[agent]
    if n_sets is not None:
        tube, p = tube[:n_sets], p[: 30 * (n_sets - 1) + 1]
    return tube, p, float(p[-1, 2]), scenarios.parking_lot_spec()


AGENTS = ("vehicle_1", "vehicle_2", "vehicle_3")  # vehicle_0 (N = 50): neither solver reaches 1e-8 (the vehicle waits: rank loss)


if __name__ == "__main__":
    import test_colloc as tc
    from oracle.colloc_nlp import CollocNlp
    from oracle.independent_colloc import GeometricColloc, solve_ipm

    out = {}
    name = "colloc_independent.npz"
    if "--truncated" in sys.argv:  # `--truncated vehicle_0_s6` -> colloc_independent_trunc.npz
        AGENTS = tuple(sys.argv[sys.argv.index("--truncated") + 1:])
        name = "colloc_independent_trunc.npz"
    for agent in AGENTS:
        tube, p, fh, sp = problem(agent)
        nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
        X0 = tc.colloc_guess(nlp, tc.warm_start(tube, p, fh))  # state_ws -> interpolation: what plan_single_path hands over
        g = GeometricColloc(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
        t0 = time.time()
        r = solve_ipm(g, X0[: nlp.iDt].reshape(-1, 7), X0[nlp.iDt])
        print(agent, {k: v for k, v in r.items() if k != "traj"}, "%.0f s" % (time.time() - t0), flush=True)
        # status 2 = the line search ran out at the rounding floor of the merit function; what makes the point a fixture is that
        # the rows hold to 2e-8 and the KKT certificate of tests/test_independent_solver.py passes
        assert r["status"] in (0, 2) and r["eq"] < 2e-8 and r["ineq"] > -1e-8
        out.update({f"{agent}_guess": X0[: nlp.iDt + 1], f"{agent}_traj": r["traj"], f"{agent}_dt": r["dt"], f"{agent}_cost": r["cost"],
                    f"{agent}_iters": r["iters"], f"{agent}_status": r["status"]})
    np.savez_compressed(os.path.join(HERE, name), **out)
