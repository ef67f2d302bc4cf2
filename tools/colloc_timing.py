import sys, os, time, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_colloc import colloc_guess
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
from oracle.colloc_nlp import CollocNlp
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    tubes_, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
plans = {a: ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes_[a]], paths[a]) for a in agents}
sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
fhs = [float(plans[a][1][-1, 2]) for a in agents]
ws = engine.state_ws([plans[a][1][0] for a in agents], tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5)
guesses, dt0s = [], []
for a, fh, w in zip(agents, fhs, ws):
    tube, p = plans[a]
    nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
    tr = w["traj"]
    z = dict(t=0.1 * np.arange(len(tr)), **{k: tr[:, c] for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))})
    X0 = colloc_guess(nlp, z)
    guesses.append(X0[: nlp.iDt].reshape(-1, 7)); dt0s.append(X0[nlp.iDt])
for sel in ([1], [0, 1, 2, 3]):
    t0 = time.time()
    res = engine.colloc(sp, [plans[agents[i]][1][0] for i in sel], [tubes[i] for i in sel], [guesses[i] for i in sel], [dt0s[i] for i in sel], [fhs[i] for i in sel], max_iter=int(os.environ.get("MAXIT", 400)))
    print(sel, "time", time.time() - t0, [(r["status"], r["iters"], round(r["cost"], 4), round(r["dt"], 5)) for r in res], flush=True)
