"""cfz_joint_colloc with the structured elimination (cfz_colloc_options.structured = 1, cfz_jstruct.inl) against the band elimination on
the GPU: B four-vehicle joint plans of configs[3] (status, iterations, trajectories), timed.
    CFZ_COLLOC_PROFILE=1 python tools/jstruct_gpu_check.py [B=16] [band=1]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
import test_configs_gpu as tcg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with_band = int(sys.argv[2]) if len(sys.argv) > 2 else 1
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
print(f"{len(who)} single plans: {t1 - t0:.2f} s, converged {sum(r['status'] == 0 for r in plans.values())}", flush=True)
scen = []
for b in range(B):
    if not all(4 * b + i in plans and plans[4 * b + i]["status"] == 0 for i in range(4)):
        continue
    sing = [plans[4 * b + i] for i in range(4)]
    scen.append(dict(init_poses=[init[4 * b + i] for i in range(4)], tubes=[lot["tubes"][a] for a in agents],
                     guesses=[s["traj"].reshape(-1, 7) for s in sing], dt0=float(np.mean([s["dt"] for s in sing])),
                     final_headings=[lot["fh"][a] for a in agents]))
sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
res = {}
for name, kw in ((("band", dict(structured=0)),) if with_band else ()) + (("structured", dict(structured=1)), ("structured", dict(structured=1))):
    t0 = time.time(); r = engine.joint_colloc_batch(sp0, scen, max_iter=300, **kw); t1 = time.time()
    res[name] = r
    st = np.array([x["status"] for x in r]); it = np.array([x["iters"] for x in r])
    print(f"{name}: {len(scen)} four-vehicle joint plans {t1 - t0:.3f} s; status counts {dict(zip(*np.unique(st, return_counts=True)))}; iterations {it.min()}-{it.max()} mean {it.mean():.1f}", flush=True)
if with_band:
    a, b = res["band"], res["structured"]
    same = sum(x["iters"] == y["iters"] and x["status"] == y["status"] for x, y in zip(a, b))
    dd = [max(float(np.abs(u - v).max()) for u, v in zip(x["traj"], y["traj"])) for x, y in zip(a, b) if x["iters"] == y["iters"]]
    print("plans with other iteration counts:", [(i, x["iters"], y["iters"]) for i, (x, y) in enumerate(zip(a, b)) if x["iters"] != y["iters"]][:16])
    print(f"same status and iteration count: {same} of {len(scen)}; largest trajectory difference among those {max(dd) if dd else float('nan'):.2e}; "
          f"cost difference {max(abs(x['cost'] - y['cost']) / x['cost'] for x, y in zip(a, b)):.2e}")
