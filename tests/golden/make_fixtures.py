"""Generates the committed fixtures of tests/golden/ (run from the repo root:
`python tests/golden/make_fixtures.py`).

  pytypes_fields.json   field names / defaults of the reference's importable struct modules
                        (needs /root/reference; skipped when absent)
  refs_4v.npz           synthetic 4-vehicle strategy (conflict_rez_amd.strategy) and, per vehicle,
                        the reference trajectory of the reference's `state_ws` NLP
                        (vehicle.py:99-231, tube constraints, dt = 0.1) solved by the oracle
                        (oracle/plan_nlp.py + oracle/ipm.py, exact Hessian), sampled every dt
  mpc_golden.npz        MPC-step instances (inputs) and their solutions by the full-KKT numpy
                        oracle (oracle/ipm.py on oracle/mpc_nlp.py)
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from conflict_rez_amd import strategy as strat  # noqa: E402
from conflict_rez_amd.control.compute_sets import (  # noqa: E402
    compute_obstacles, compute_sets, interp_along_sets)
from conflict_rez_amd.vehicle_types import VehicleBody  # noqa: E402
from oracle import ipm  # noqa: E402
from oracle.mpc_nlp import MpcSpec, solve_mpc  # noqa: E402
from oracle.plan_nlp import StateWsNlp  # noqa: E402


def pytypes_fields():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference not present: pytypes_fields.json left as is")
        return
    sys.path.insert(0, ref)
    import dataclasses
    import importlib

    out = {}
    for mod, names in (("confrez.pytypes", ["Position", "VehicleActuation", "BodyLinearVelocity", "BodyAngularVelocity",
                                             "BodyLinearAcceleration", "BodyAngularAcceleration", "OrientationEuler",
                                             "OrientationQuaternion", "ParametricPose", "ParametricVelocity", "VehicleState",
                                             "VehiclePrediction"]),
                       ("confrez.vehicle_types", ["VehicleConfig", "VehicleBody"]),
                       ("confrez.obstacle_types", ["GeofenceRegion"])):
        m = importlib.import_module(mod)
        for n in names:
            inst = getattr(m, n)()
            out[n] = {f.name: (getattr(inst, f.name) if isinstance(getattr(inst, f.name), (int, float, type(None))) else
                               type(getattr(inst, f.name)).__name__) for f in dataclasses.fields(inst)}
    vb = importlib.import_module("confrez.vehicle_types").VehicleBody()
    out["VehicleBody.A"] = np.asarray(vb.A).tolist()
    out["VehicleBody.b"] = np.asarray(vb.b).tolist()
    out["VehicleBody.V"] = np.asarray(vb.V).tolist()
    with open(os.path.join(HERE, "pytypes_fields.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    sys.path.remove(ref)


def refs_4v():
    hist = strat.generate_strategy(4)
    vb = VehicleBody()
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, vb, 30)
    agents = sorted(hist)
    trajs = []
    for a in agents:
        tube = [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]]
        p = paths[a]
        nlp = StateWsNlp(p[0], tube, final_heading=None, shrink_tube=0.5)
        res = ipm.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2]), ipm.IpmOptions(max_iter=300, hessian="exact"))
        assert res["status"] == 0, (a, res["status"])
        s = nlp.unpack(res["X"])
        trajs.append(np.stack([s["x"], s["y"], s["psi"], s["v"], s["delta"], s["a"], s["w"]], 1))
        print(a, "state_ws iters", res["iters"], "f %.4f" % res["f"], "T", len(s["x"]))
    T = max(len(t) for t in trajs)
    table = np.stack([np.concatenate([t, np.repeat(t[-1:], T - len(t), 0)], 0) for t in trajs])  # [V,T,7]
    table[:, :, 3:] = np.where(np.arange(T)[None, :, None] < np.array([len(t) for t in trajs])[:, None, None], table[:, :, 3:], 0.0)
    cells = {a: np.array([[*s["front"], *s["back"]] for s in hist[a]], dtype=np.int32) for a in agents}
    np.savez_compressed(os.path.join(HERE, "refs_4v.npz"), table=table, lengths=np.array([len(t) for t in trajs]),
                        **{"cells_" + a: cells[a] for a in agents})
    return table


def mpc_golden(table):
    obs = compute_obstacles()
    spec = MpcSpec(A_obs=np.stack([o.A for o in obs]), b_obs=np.stack([o.b for o in obs]), n_nbr=3)
    N, V, T = spec.N, table.shape[0], table.shape[1]
    rng = np.random.default_rng(7)
    X0, REF, NBR, ZU, SOL, META = [], [], [], [], [], []
    for case in range(16):
        v = case % V
        k0 = int(rng.integers(0, T - 40))
        idx = np.minimum(k0 + np.arange(N), T - 1)
        adv = np.minimum(idx + 1, T - 1)
        ref = table[v, idx, :3].T.copy()
        x0 = table[v, k0, :5] + rng.normal(0, [0.05, 0.05, 0.02, 0.05, 0.0])
        others = [u for u in range(V) if u != v]
        nbr = np.stack([table[u, adv, :3].T for u in others])
        zu = table[v, adv, :].T.copy()
        if case >= 12:  # cold guess: poses only, as after a fallback
            zu[3:] = 0.0
        res = solve_mpc(spec, x0, ref, nbr, zu)
        X0.append(x0), REF.append(ref), NBR.append(nbr), ZU.append(zu)
        SOL.append(res["zu"])
        META.append([res["status"], res["iters"], res["f"], res["sep"].min() if res["sep"] is not None else np.nan])
        print("case", case, "vehicle", v, "k0", k0, "status", res["status"], "iters", res["iters"], "f %.5f" % res["f"],
              "min sep", META[-1][3])
    # cases 16, 17: locally infeasible starts out of the benchmark's scenario sampler (the perturbed state sits
    # inside the dmin margin of a neighbour but within the 2 constr_viol_tol band of the status-4 pre-check):
    # the constraint violation stalls and the solver stops with status 5
    from conflict_rez_amd import scenarios

    k0s, noise = scenarios.sample_scenarios(1024, table, seed=2024)
    bx0, bref, bnbr, bzu = scenarios.mpc_batch_from_table(scenarios.parking_lot_spec(), table, k0s, noise)
    for case, b in ((16, 716), (17, 764)):
        res = solve_mpc(spec, bx0[b], bref[b], bnbr[b], bzu[b])
        X0.append(bx0[b]), REF.append(bref[b]), NBR.append(bnbr[b]), ZU.append(bzu[b])
        SOL.append(res["zu"])
        META.append([res["status"], res["iters"], res["f"], res["sep"].min() if res["sep"] is not None else np.nan])
        print("case", case, "scenario instance", b, "status", res["status"], "iters", res["iters"], "f %.5f" % res["f"])
        assert res["status"] in (4, 5)
    # cases 18, 19: a vehicle pushed hard against a separation row (inputs recorded from a closed-loop run of the
    # engine, tools/capture_hard.py -> contact_inputs.npz).  Without the curvature of the separation rows in the
    # Hessian these need 450-600 iterations (period-2 oscillation of the Gauss-Newton iteration); with it about 10.
    ci = np.load(os.path.join(HERE, "contact_inputs.npz"))
    for case, i in ((18, 0), (19, 1)):
        res = solve_mpc(spec, ci["x0"][i], ci["ref"][i], ci["nbr"][i], ci["zu"][i])
        X0.append(ci["x0"][i]), REF.append(ci["ref"][i]), NBR.append(ci["nbr"][i]), ZU.append(ci["zu"][i])
        SOL.append(res["zu"])
        META.append([res["status"], res["iters"], res["f"], res["sep"].min()])
        print("case", case, "contact input", i, "status", res["status"], "iters", res["iters"], "f %.5f" % res["f"])
        assert res["status"] == 0 and res["iters"] < 20
    np.savez_compressed(os.path.join(HERE, "mpc_golden.npz"), x0=np.array(X0), ref=np.array(REF), nbr=np.array(NBR),
                        zu=np.array(ZU), sol=np.array(SOL), meta=np.array(META), A_obs=spec.A_obs, b_obs=spec.b_obs)


def carry_golden():
    """Two vehicles over three consecutive MPC iterations (inputs recorded from a closed-loop run of the engine,
    tools/capture_hard.py -> carry_inputs.npz): every solve after the first starts from the multipliers of the one
    before (`solve_mpc(carry=)`).  Stored: status, iterations, solution of every solve, with and without carrying."""
    obs = compute_obstacles()
    spec = MpcSpec(A_obs=np.stack([o.A for o in obs]), b_obs=np.stack([o.b for o in obs]), n_nbr=3)
    ci = np.load(os.path.join(HERE, "carry_inputs.npz"))
    sol, meta = [], []
    for seq in range(len(ci["x0"]) // 3):
        carry = None
        for t in range(3):
            i = 3 * seq + t
            cold = solve_mpc(spec, ci["x0"][i], ci["ref"][i], ci["nbr"][i], ci["zu"][i])
            res = solve_mpc(spec, ci["x0"][i], ci["ref"][i], ci["nbr"][i], ci["zu"][i], carry=carry)
            carry = res["carry"]
            sol.append(res["zu"]); meta.append([res["status"], res["iters"], res["f"], cold["status"], cold["iters"], cold["f"]])
            print("sequence", seq, "step", t, "carried: status", res["status"], "iters", res["iters"], "| cold: iters", cold["iters"],
                  "max |dz|", np.abs(res["zu"] - cold["zu"]).max())
    np.savez_compressed(os.path.join(HERE, "carry_golden.npz"), sol=np.array(sol), meta=np.array(meta))


def colloc_golden():
    """Regression pins of the planning solvers (NOT an oracle: produced by the CPU build of the kernel source,
    tests/emu/cfz_colloc_emu.cpp; what checks those solutions independently is tests/test_colloc.py and
    tests/test_independent_solver.py): the single plan of vehicle_1 and the joint plan of vehicles 2 and 3 of the synthetic
    strategy at the reference's sizes -- guess (points + dt) in, status, iterations, cost and solution out."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import colloc_emu_binding as ce
    import test_colloc as tc
    from conflict_rez_amd import scenarios
    from oracle.colloc_nlp import CollocNlp

    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    plans = {a: ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]], paths[a]) for a in sorted(hist)}
    opt = ipm.IpmOptions(**tc.COLLOC_OPT)
    sp = scenarios.parking_lot_spec()
    tube, p = plans["vehicle_1"]
    fh = float(p[-1, 2])
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
    X0 = tc.colloc_guess(nlp, tc.warm_start(tube, p, fh))
    r1 = ce.solve(nlp, X0, opt)
    jn, _ = tc._joint_problem(plans, ["vehicle_2", "vehicle_3"], [0, 0], nps=5)
    J0, _ = tc._joint_guess(plans, ["vehicle_2", "vehicle_3"], jn, sp, 5)
    r2 = ce.solve(jn, J0, opt)
    print("single", r1["status"], r1["iters"], r1["f"], "joint", r2["status"], r2["iters"], r2["f"])
    np.savez_compressed(os.path.join(HERE, "colloc_golden.npz"), single_guess=X0[: nlp.iDt + 1], single_sol=r1["X"],
                        single_meta=np.array([r1["status"], r1["iters"], r1["f"]]), joint_guess=J0[: jn.iDt + 1], joint_sol=r2["X"],
                        joint_meta=np.array([r2["status"], r2["iters"], r2["f"]]))


if __name__ == "__main__":
    if "--mpc-only" in sys.argv:  # the MPC fixtures again on the committed table (after a change of the MPC algorithm)
        mpc_golden(np.load(os.path.join(HERE, "refs_4v.npz"))["table"])
        carry_golden()
    else:
        pytypes_fields()
        mpc_golden(refs_4v())
        carry_golden()
    colloc_golden()
