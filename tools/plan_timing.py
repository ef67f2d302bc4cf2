"""Times cfz_state_ws on the synthetic 4-vehicle strategy (one launch for all four plans)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conflict_rez_amd import engine, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody

hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj")
    strat.write_strategy(fn, hist)
    tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
tb = [[((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in tubes[a][1:]] for a in agents]
for rep in range(2):
    t = time.perf_counter()
    res = engine.state_ws([paths[a][0] for a in agents], tb, [paths[a] for a in agents], [None] * 4, shrink_tube=0.5)
    dt = time.perf_counter() - t
    print(f"4 plans in {dt:.2f} s:", [(len(r["traj"]) - 1, r["status"], r["iters"]) for r in res], "(T, status, iterations)")
for i, a in enumerate(agents):
    t = time.perf_counter()
    r = engine.state_ws([paths[a][0]], [tb[i]], [paths[a]], [None], shrink_tube=0.5)[0]
    print(a, f"alone {time.perf_counter() - t:.2f} s, T {len(r['traj']) - 1}, {r['iters']} iterations -> {(time.perf_counter() - t) / max(r['iters'], 1) * 1e3:.1f} ms per iteration")
