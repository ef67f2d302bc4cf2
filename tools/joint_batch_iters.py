"""Iteration counts of the 256-plan four-vehicle joint launch (bench.py's configs[3] inputs) with and without vertex-vertex rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from conflict_rez_amd import engine, scenarios

import tempfile
from conflict_rez_amd import strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
fh = {a: float(paths[a][-1, 2]) for a in agents}
sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
rng = np.random.default_rng(0)
who = [agents[i % 4] for i in range(4 * B)]
init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
def guess_of(ws, n_sets, nps=5):
    N = nps * (n_sets - 1); t = 0.1 * np.arange(len(ws))
    ti = (np.arange(N)[:, None] + bench.TAU5[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N
idx = list(range(4 * B))
ws = engine.state_ws([init[i] for i in idx], [tubes[who[i]] for i in idx], [paths[who[i]] for i in idx], [fh[who[i]] for i in idx], shrink_tube=0.5)
good = [k for k, w in enumerate(ws) if w["status"] == 0]
gs = {k: guess_of(ws[k]["traj"], len(tubes[who[k]]) + 1) for k in good}
for vv in (1, 0):
    rg = dict(zip(good, engine.colloc(sp0, [init[k] for k in good], [tubes[who[k]] for k in good], [gs[k][0] for k in good], [gs[k][1] for k in good], [fh[who[k]] for k in good], max_iter=400, vv_rows=vv)))
    scen, sid = [], []
    for b in range(B):
        ks = [4 * b + i for i in range(4)]
        if all(k in rg and rg[k]["status"] == 0 for k in ks):
            scen.append(dict(init_poses=[init[k] for k in ks], tubes=[tubes[a] for a in agents], guesses=[rg[k]["traj"].reshape(-1, 7) for k in ks],
                             dt0=float(np.mean([rg[k]["dt"] for k in ks])), final_headings=[fh[a] for a in agents])); sid.append(b)
    t0 = time.perf_counter()
    rj = engine.joint_colloc_batch(sp0, scen, max_iter=300, vv_rows=vv)
    t = time.perf_counter() - t0
    it = np.array([r["iters"] for r in rj]); st = np.array([r["status"] for r in rj])
    print("vv_rows", vv, "plans", len(scen), "time %.2f s" % t, "iters mean %.1f max %d" % (it.mean(), it.max()), "status", np.bincount(st, minlength=4).tolist(),
          "slowest", [(sid[i], int(it[i])) for i in np.argsort(it)[-5:]], "single iters mean %.1f max %d" % (np.mean([r["iters"] for r in rg.values()]), max(r["iters"] for r in rg.values())), flush=True)
    if vv == 1:
        worst = int(np.argmax(it))
        np.savez("gpurun_out/joint_worst.npz", init=np.array(scen[worst]["init_poses"]), dt0=scen[worst]["dt0"], **{f"g{i}": scen[worst]["guesses"][i] for i in range(4)})
