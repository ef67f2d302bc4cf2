"""The FULL-LENGTH four-vehicle joint plan (BASELINE configs[3]'s problem: 50 / 30 / 30 / 40 intervals, six pairs) at tight tolerances on the
GPU, from the single plans' guess as `solve_final_problem_obca` starts it (multi_vehicle_planner.py:343-480): does the kernel reach an
optimum there, and what does the solver-free certificate on the independent statement say about it (tests/golden/make_independent_joint.py
joint_kkt_certificate, run on the CPU afterwards)?  VERDICT r5 item 7a.
    python tools/joint_full_tight.py <out.npz> [dmin=0.2] [max_iter=3000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
from conflict_rez_amd import engine, scenarios
from make_independent_joint import plans_of_strategy

out = sys.argv[1]
dmin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
max_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
plans = plans_of_strategy()
agents = sorted(plans)
sp = scenarios.parking_lot_spec(n_nbr=0, N=2, dmin=dmin)
tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
fhs = [float(plans[a][1][-1, 2]) for a in agents]
init = [plans[a][1][0] for a in agents]
ws = engine.state_ws(init, tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5)
tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
gs = []
for w_, t_ in zip(ws, tubes):
    N = 5 * len(t_)
    t = 0.1 * np.arange(len(w_["traj"]))
    ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
    gs.append((np.stack([np.interp(ti, t, w_["traj"][:, c]) for c in range(7)], 1), t[-1] / N))
sing = engine.colloc(sp, init, tubes, [g[0] for g in gs], [g[1] for g in gs], fhs, max_iter=400)
print("single plans:", [(s["status"], s["iters"]) for s in sing], flush=True)
guesses = [s["traj"].reshape(-1, 7) for s in sing]
dt0 = float(np.mean([s["dt"] for s in sing]))
t0 = time.time()
r1 = engine.joint_colloc(sp, init, tubes, guesses, dt0, fhs, max_iter=400)
print(f"production tolerance: status {r1['status']} iterations {r1['iters']} cost {r1['cost']:.6f} dt {r1['dt']:.6f} ({time.time() - t0:.1f} s)", flush=True)
t0 = time.time()
r2 = engine.joint_colloc(sp, init, tubes, guesses, dt0, fhs, max_iter=max_iter, tol=1e-8, constr_viol_tol=1e-9, exact_rows=1)
print(f"tight: status {r2['status']} iterations {r2['iters']} cost {r2['cost']:.9f} dt {r2['dt']:.9f} ({time.time() - t0:.1f} s)", flush=True)
np.savez(out, dmin=dmin, dt0=dt0, **{f"guess{a}": guesses[a] for a in range(4)}, **{f"traj{a}": np.asarray(r2["traj"][a]) for a in range(4)}, dt=r2["dt"], cost=r2["cost"],
         status=r2["status"], iters=r2["iters"], **{f"ptraj{a}": np.asarray(r1["traj"][a]) for a in range(4)}, pdt=r1["dt"], pcost=r1["cost"], pstatus=r1["status"], piters=r1["iters"])
