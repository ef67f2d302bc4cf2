"""The same collocation plan alone and 520 times in one batch, several times over: every plan of the batch must equal the lone one bit for
bit (the check that caught the retired one-wavefront kernel, docs/notebook.md round 4).  Run on the GPU box:
python tools/colloc_batch_determinism.py [repetitions]"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
a = "vehicle_1"
tube = [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]]
fh = float(paths[a][-1, 2]); B = 520
sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
ws1 = engine.state_ws([paths[a][0]], [tube], [paths[a]], [fh], shrink_tube=0.5, kernel=engine.KERNEL_WIDE)[0]["traj"]
N = 5 * len(tube); t = 0.1 * np.arange(len(ws1))
tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
guess = np.stack([np.interp(ti, t, ws1[:, c]) for c in range(7)], 1)
cargs = lambda B_: (sp0, [paths[a][0]] * B_, [tube] * B_, [guess] * B_, [t[-1] / N] * B_, [fh] * B_)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for kern, kw in (("structured", dict(structured=1)), ("band", dict(structured=0))):
        one = engine.colloc(*cargs(1), max_iter=400, **kw)[0]
        many = engine.colloc(*cargs(B), max_iter=400, **kw)
        bad = [(i, float(np.abs(r["traj"] - one["traj"]).max()), r["iters"]) for i, r in enumerate(many) if not np.array_equal(r["traj"], one["traj"])]
        print(rep, kern, "one iters", one["iters"], "differing plans", len(bad), bad[:6], flush=True)
