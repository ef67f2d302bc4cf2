"""configs[1] with BASELINE.json's four obstacles: which of the 256 plans of the bench's `extra` takes the most iterations, and its inputs
(saved to gpurun_out/four_worst.npz for a replay on the CPU build).   python tools/four_obstacle_tail.py"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
import bench

hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
fh = {a: float(paths[a][-1, 2]) for a in agents}
B = 256
rng = np.random.default_rng(0)
who = [agents[i % 4] for i in range(4 * B)]
init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
idx = list(range(B))
ws = engine.state_ws([init[i] for i in idx], [tubes[who[i]] for i in idx], [paths[who[i]] for i in idx], [fh[who[i]] for i in idx], shrink_tube=0.5)
def guess_of(w, n_sets, nps=5):
    N = nps * (n_sets - 1); t = 0.1 * np.arange(len(w)); ti = (np.arange(N)[:, None] + bench.TAU5[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, w[:, c]) for c in range(7)], 1), t[-1] / N
gs = [guess_of(ws[k]["traj"], len(tubes[who[k]]) + 1) for k in idx]
for n_obs in (4, 6):
    sp = scenarios.parking_lot_spec(n_nbr=0, N=2, n_obs=n_obs)
    r = engine.colloc(sp, [init[k] for k in idx], [tubes[who[k]] for k in idx], [g[0] for g in gs], [g[1] for g in gs], [fh[who[k]] for k in idx], max_iter=400)
    it = np.array([x["iters"] for x in r])
    top = np.argsort(-it)[:5]
    print(n_obs, "obstacles: iterations max", it.max(), "mean", it.mean(), "top", [(int(k), who[k], int(it[k])) for k in top])
    if n_obs == 4:
        k = int(top[0])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "four_worst.npz"), k=k, agent=who[k], init=init[k], guess=gs[k][0], dt0=gs[k][1], iters=it[k], traj=r[k]["traj"], dt=r[k]["dt"])
