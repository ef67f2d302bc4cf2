"""Generates tests/golden/joint_retry_instance.npz ON A GPU BOX (`python tests/golden/make_joint_retry_instance.py`): the inputs of one
four-vehicle joint plan of tests/test_configs_gpu.py::test_config3_at_batch_256 (256 scenarios, start poses scattered by 3 cm, sampler
seed 1; single plans by `cfz_state_ws` / `cfz_colloc` on the GPU) -- scenario 89, the plan that ended with status 2 at mu = mu_floor
before a failed line search was repeated with a larger inertia perturbation (tests/test_colloc.py pins it on the CPU build).
Stored: the four start poses, the four single plans (the joint plan's guess) and their mean dt."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_configs_gpu as tcg
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
lot = dict(agents=agents, tubes={a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}, paths=paths, fh={a: float(paths[a][-1, 2]) for a in agents})
B = 256
rng = np.random.default_rng(1)
who = [a for _ in range(B) for a in agents]
init = [lot["paths"][a][0] + (np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] if i >= 4 else 0.0) for i, a in enumerate(who)]
ws, good, plans = tcg._single_plans(lot, who, init)
ok = [b for b in range(B) if all(4 * b + i in plans and plans[4 * b + i]["status"] == 0 for i in range(4))]
scen = []
for b in ok:
    sing = [plans[4 * b + i] for i in range(4)]
    scen.append(dict(init_poses=[init[4 * b + i] for i in range(4)], tubes=[lot["tubes"][a] for a in agents], guesses=[s["traj"].reshape(-1, 7) for s in sing], dt0=float(np.mean([s["dt"] for s in sing])), final_headings=[lot["fh"][a] for a in agents]))
sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
rj = engine.joint_colloc_batch(sp0, scen, max_iter=300)
i = 89
np.savez(os.path.join(ROOT, "tests", "golden", "joint_retry_instance.npz"), init=np.array(scen[i]["init_poses"]), dt0=scen[i]["dt0"],
         **{f"g{k}": scen[i]["guesses"][k] for k in range(4)})
print("scenario", i, "on this build:", rj[i]["status"], rj[i]["iters"])
