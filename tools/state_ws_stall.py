import os, sys, time, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conflict_rez_amd import engine, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
fh = {a: float(paths[a][-1, 2]) for a in agents}
rng = np.random.default_rng(0); B = 256
who = [agents[i % 4] for i in range(B)]
init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
for st in (0, 10, 20, 40):
    t0 = time.time(); ws = engine.state_ws(init, [tubes[a] for a in who], [paths[a] for a in who], [fh[a] for a in who], shrink_tube=0.5, stall_iters=st); t1 = time.time()
    s = np.array([w["status"] for w in ws]); it = np.array([w["iters"] for w in ws])
    print("stall_iters", st, "%.2f s" % (t1 - t0), "status counts", dict(zip(*np.unique(s, return_counts=True))), "max iters", it.max(), "max iters of converged", it[s == 0].max(), flush=True)
