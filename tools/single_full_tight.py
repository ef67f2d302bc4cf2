"""Vehicle 0's single-vehicle collocation plan at FULL length (50 Radau intervals: the longest plan, the one no independent solver reaches a
tight optimum on) at tight tolerances on the GPU, from its `state_ws` guess as `plan_single_path` starts it (vehicle.py:99-231, :360-661);
the plan is dumped for the solver-free certificate (tests/golden/make_single_full_certificate.py, on the CPU).  VERDICT r5 item 7a.
    python tools/single_full_tight.py <out.npz> [agent=vehicle_0] [max_iter=3000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from conflict_rez_amd import engine, scenarios
from make_independent_joint import plans_of_strategy

out = sys.argv[1]
agent = sys.argv[2] if len(sys.argv) > 2 else "vehicle_0"
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
plans = plans_of_strategy()
sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
tube = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[agent][0][1:]]
p = plans[agent][1]
fh = float(p[-1, 2])
ws = engine.state_ws([p[0]], [tube], [p], [fh], shrink_tube=0.5)[0]
tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
N = 5 * len(tube)
t = 0.1 * np.arange(len(ws["traj"]))
ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
g = np.stack([np.interp(ti, t, ws["traj"][:, c]) for c in range(7)], 1)
dt0 = t[-1] / N
args = (sp, [p[0]], [tube], [g], [dt0], [fh])
r1 = engine.colloc(*args, max_iter=400)[0]
print(f"{agent}: state_ws status {ws['status']} iterations {ws['iters']}; reference tolerance: status {r1['status']} iterations {r1['iters']} cost {r1['cost']:.6f} dt {r1['dt']:.6f}", flush=True)
t0 = time.time()
r2 = engine.colloc(*args, max_iter=max_iter, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)[0]
print(f"tight: status {r2['status']} iterations {r2['iters']} cost {r2['cost']:.9f} dt {r2['dt']:.9f} ({time.time() - t0:.1f} s)", flush=True)
np.savez(out, agent=agent, guess=g, dt0=dt0, traj=np.asarray(r2["traj"]), dt=r2["dt"], cost=r2["cost"], status=r2["status"], iters=r2["iters"],
         ptraj=np.asarray(r1["traj"]), pdt=r1["dt"], pcost=r1["cost"], pstatus=r1["status"], piters=r1["iters"])
