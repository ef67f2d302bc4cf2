"""Joint collocation plan on the GPU box: single plans of the chosen vehicles, then cfz_joint_colloc.
usage: python tools/joint_timing.py vehicle_0,vehicle_3 [N_per_set]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import strategy as strat
from conflict_rez_amd.control.multi_vehicle_planner import MultiVehiclePlanner
from conflict_rez_amd.control.compute_sets import interp_along_sets
from conflict_rez_amd.pytypes import VehicleState
from conflict_rez_amd.vehicle_types import VehicleBody

agents = sys.argv[1].split(",") if len(sys.argv) > 1 else ["vehicle_%d" % i for i in range(4)]
nps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fn = os.path.join(tempfile.mkdtemp(), "4v_rl_traj")
strat.write_strategy(fn, strat.generate_strategy(4))
paths = interp_along_sets(fn, VehicleBody(), 30)
mvp = MultiVehiclePlanner(fn, {a: True for a in agents}, {a: {"front": (1, 0, 0), "back": (0, 0, 1)} for a in agents},
                          {a: VehicleState() for a in agents}, {a: float(paths[a][-1, 2]) for a in agents})
t0 = time.time(); mvp.solve_single_problems(N_per_set=nps); t1 = time.time()
print("single plans %.2f s" % (t1 - t0), {a: (mvp.vehicles[a].final_problem_stats["iters"], round(mvp.single_results[a].dt, 4)) for a in agents}, flush=True)
t0 = time.time(); mvp.solve_final_problem_obca(N_per_set=nps); t1 = time.time()
print("joint plan %.2f s" % (t1 - t0), mvp.final_stats, "dt", mvp.final_dt, flush=True)
P = {a: mvp.final_results[a] for a in agents}
dmin = min(np.hypot(P[a].x - P[b].x, P[a].y - P[b].y).min() for a in agents for b in agents if a < b)
print("closest rear axles on the common clock: %.2f m" % dmin)
