"""BASELINE configs[3] at its named batch: B four-vehicle joint plans (six pairs, one shared dt) in ONE launch of
cfz_joint_colloc, start poses scattered by +-3 cm around the synthetic strategy's.  Prints status counts, iterations and time.
usage (GPU box): python tools/joint_batch_timing.py [B=256]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
import test_configs_gpu as tcg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
lot = dict(agents=agents, tubes={a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents},
           paths=paths, fh={a: float(paths[a][-1, 2]) for a in agents})
rng = np.random.default_rng(1)
who = [a for _ in range(B) for a in agents]
init = [lot["paths"][a][0] + (np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] if i >= 4 else 0.0) for i, a in enumerate(who)]
t0 = time.time(); ws, good, plans = tcg._single_plans(lot, who, init); t1 = time.time()
print(f"{len(who)} single plans (state_ws + collocation): {t1 - t0:.2f} s, converged {sum(r['status'] == 0 for r in plans.values())}", flush=True)
scen = []
for b in range(B):
    if not all(4 * b + i in plans and plans[4 * b + i]["status"] == 0 for i in range(4)):
        continue
    sing = [plans[4 * b + i] for i in range(4)]
    scen.append(dict(init_poses=[init[4 * b + i] for i in range(4)], tubes=[lot["tubes"][a] for a in agents],
                     guesses=[s["traj"].reshape(-1, 7) for s in sing], dt0=float(np.mean([s["dt"] for s in sing])),
                     final_headings=[lot["fh"][a] for a in agents]))
sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
t0 = time.time(); rj = engine.joint_colloc_batch(sp0, scen, max_iter=300); t1 = time.time()
st = np.array([r["status"] for r in rj]); it = np.array([r["iters"] for r in rj])
print(f"{len(scen)} four-vehicle joint plans in one launch: {t1 - t0:.2f} s wall; status counts {dict(zip(*np.unique(st, return_counts=True)))}; "
      f"iterations mean {it.mean():.1f} max {it.max()}", flush=True)
