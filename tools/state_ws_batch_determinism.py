"""The same state_ws plan alone and 520 times in one batch (LDS and workspace variants), several times over: every plan of the batch must
equal the lone one bit for bit.  Run on the GPU box: python tools/state_ws_batch_determinism.py [repetitions]"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from conflict_rez_amd import engine, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for a in sorted(hist):
        tube = [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]]
        fh = float(paths[a][-1, 2]); B = 520
        one = engine.state_ws([paths[a][0]], [tube], [paths[a]], [fh], shrink_tube=0.5)[0]
        for kern in (engine.KERNEL_WIDE, engine.KERNEL_NARROW):
            many = engine.state_ws([paths[a][0]] * B, [tube] * B, [paths[a]] * B, [fh] * B, shrink_tube=0.5, kernel=kern)
            bad = sum(not (np.array_equal(r["traj"], one["traj"]) and r["iters"] == one["iters"]) for r in many)
            print(rep, a, "LDS" if kern == 1 else "workspace", "iters", one["iters"], "differing plans", bad, flush=True)
