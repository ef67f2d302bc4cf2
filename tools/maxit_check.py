import os, sys, tempfile
sys.path.insert(0, "/root/repo")
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
a = "vehicle_1"
tubes = [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]]
r = engine.state_ws([paths[a][0]], [tubes], [paths[a]], [float(paths[a][-1, 2])], shrink_tube=0.5, max_iter=3)[0]
print("state_ws max_iter=3:", r["status"], r["iters"])
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(2, table, seed=1)
x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
eng = engine.Engine(spec, max_batch=len(x0), max_iter=2)
out = eng.solve(x0, ref, nbr, zu)
print("mpc max_iter=2:", out["status"].tolist(), out["iters"].tolist())
