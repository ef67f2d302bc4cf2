"""cfz_state_ws on B plans (the four vehicles of the synthetic strategy in turn, start poses shifted by centimetres): wall time of the
second call (the first pays the library's and the arena's start-up).  Run on the GPU box: python tools/state_ws_timing.py [B]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, strategy as strat
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
rng = np.random.default_rng(0)
who = [agents[i % 4] for i in range(B)]
init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
args = (init, [tubes[a] for a in who], [paths[a] for a in who], [fh[a] for a in who])
for rep in range(3):
    t0 = time.time()
    ws = engine.state_ws(*args, shrink_tube=0.5)
    t1 = time.time()
    its = np.array([w["iters"] for w in ws])
    print(f"call {rep}: {B} plans {t1 - t0:.4f} s, converged {sum(w['status'] == 0 for w in ws)}, iterations {its.min()}-{its.max()} mean {its.mean():.1f}", flush=True)
for a in agents:  # one plan alone: wall time of the call / its iterations
    arg1 = ([paths[a][0]], [tubes[a]], [paths[a]], [fh[a]])
    engine.state_ws(*arg1, shrink_tube=0.5)
    t0 = time.time(); r = engine.state_ws(*arg1, shrink_tube=0.5)[0]; t1 = time.time()
    print(f"{a} alone: T {len(r['traj']) - 1}, {r['iters']} iterations, {1e3 * (t1 - t0):.2f} ms -> {1e3 * (t1 - t0) / max(r['iters'], 1):.3f} ms per iteration", flush=True)
worst = int(np.argmax(its))
arg1 = ([init[worst]], [tubes[who[worst]]], [paths[who[worst]]], [fh[who[worst]]])
t0 = time.time(); r = engine.state_ws(*arg1, shrink_tube=0.5)[0]; t1 = time.time()
print(f"slowest of the batch ({who[worst]}, #{worst}) alone: {r['iters']} iterations, {1e3 * (t1 - t0):.2f} ms -> {1e3 * (t1 - t0) / max(r['iters'], 1):.3f} ms per iteration")
