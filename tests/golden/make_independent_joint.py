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
AGENTS = ("vehicle_2", "vehicle_3")  # default: joint_independent.npz; `... vehicle_0 vehicle_2` -> joint_independent_02.npz


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


if __name__ == "__main__":
    import test_colloc as tc
    from oracle.independent_colloc import GeometricColloc
    from oracle.independent_joint import solve_joint_ipm

    if len(sys.argv) == 3:
        AGENTS = (sys.argv[1], sys.argv[2])
    out_name = "joint_independent.npz" if AGENTS == ("vehicle_2", "vehicle_3") else "joint_independent_%s%s.npz" % (AGENTS[0][-1], AGENTS[1][-1])
    plans = plans_of_strategy()
    jn, sp = tc._joint_problem(plans, list(AGENTS), [0, 0], nps=5)
    X0, _ = tc._joint_guess(plans, list(AGENTS), jn, sp, 5)
    guesses = [X0[7 * 6 * jn.off[a]: 7 * 6 * jn.off[a + 1]].reshape(-1, 7) for a in range(2)]
    gs = [GeometricColloc(plans[a][1][0], plans[a][0], sp.A_obs, sp.b_obs, N_per_set=5, final_heading=float(plans[a][1][-1, 2])) for a in AGENTS]
    t0 = time.time()
    r = solve_joint_ipm(gs, [(0, 1)], guesses, float(X0[jn.iDt]))
    print({k: v for k, v in r.items() if k != "trajs"}, "%.0f s" % (time.time() - t0), flush=True)
    assert r["status"] in (0, 2) and r["eq"] < 2e-8 and r["ineq"] > -1e-8
    np.savez_compressed(os.path.join(HERE, out_name), guess0=guesses[0], guess1=guesses[1], dt0=float(X0[jn.iDt]),
                        traj0=r["trajs"][0], traj1=r["trajs"][1], dt=r["dt"], cost=r["cost"], iters=r["iters"], status=r["status"], pair=r["pair"])
