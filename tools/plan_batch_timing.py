"""BASELINE.json configs[1] and configs[3] on the planning kernels: 256 single-vehicle collocation plans in one launch, and
256 two-vehicle joint plans in one launch (one workgroup per plan).  Run on the GPU box: python tools/plan_batch_timing.py [B]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
MU = dict(mu_init=float(os.environ["MU0"])) if "MU0" in os.environ else {}  # experiments: initial barrier parameter
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
fh = {a: float(paths[a][-1, 2]) for a in agents}
rng = np.random.default_rng(0)
tau = np.append(0.0, [0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
def guess_of(ws, n_sets):
    N = 5 * (n_sets - 1)
    t = 0.1 * np.arange(len(ws)); ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N
# B single plans: the four vehicles in turn, start poses shifted by a few centimetres
who = [agents[i % 4] for i in range(B)]
init = [paths[a][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
t0 = time.time()
ws = engine.state_ws(init, [tubes[a] for a in who], [paths[a] for a in who], [fh[a] for a in who], shrink_tube=0.5)
t1 = time.time()
gs = [guess_of(w["traj"], len(tubes[a]) + 1) for w, a in zip(ws, who)]
good = [i for i, w in enumerate(ws) if w["status"] == 0]  # a vehicle whose warm start failed is not refined (plan_single_path raises there)
rg = engine.colloc(sp, [init[i] for i in good], [tubes[who[i]] for i in good], [gs[i][0] for i in good], [gs[i][1] for i in good], [fh[who[i]] for i in good], max_iter=int(os.environ.get("MAXIT", 150)), **MU)
t2 = time.time()
res = [None] * B
for i, r in zip(good, rg):
    res[i] = r
for i in range(B):
    if res[i] is None:
        res[i] = res[i - 4]
ok = sum(r["status"] == 0 for r in rg)
its = np.sort([r['iters'] for r in rg])
print('slow or failed:', sorted((r['iters'], r['status']) for r in rg if r['iters'] > 60 or r['status'] != 0))
print('collocation iterations: median', int(np.median(its)), '90%', int(its[int(0.9 * len(its))]), '99%', int(its[int(0.99 * len(its))]), 'top', its[-4:].tolist())
print(f"{B} single plans: state_ws {t1 - t0:.2f} s ({sum(w['status'] == 0 for w in ws)} converged), collocation {t2 - t1:.2f} s ({ok} converged, "
      f"iterations {min(r['iters'] for r in rg)}-{max(r['iters'] for r in rg)}) -> {len(good) / (t2 - t0):.0f} plans/s", flush=True)
# B joint plans of vehicles 2 and 3 from their single plans
pair = ["vehicle_2", "vehicle_3"]
scen = []
for b in range(B):
    idx = [next(i for i in range(b % 4 * 0, B) if who[i] == a and i >= (b // 4) * 4) if False else None for a in pair]
    sing = [res[i] for i in [ (b // 4) * 4 % B + agents.index(a) for a in pair ]]
    ip = [init[(b // 4) * 4 % B + agents.index(a)] for a in pair]
    scen.append(dict(init_poses=ip, tubes=[tubes[a] for a in pair], guesses=[s["traj"].reshape(-1, 7) for s in sing],
                     dt0=float(np.mean([s["dt"] for s in sing])), final_headings=[fh[a] for a in pair]))
t0 = time.time()
rj = engine.joint_colloc_batch(sp, scen, max_iter=400, **MU)
t1 = time.time()
print(f"{B} two-vehicle joint plans: {t1 - t0:.2f} s ({sum(r['status'] == 0 for r in rj)} converged, iterations {min(r['iters'] for r in rj)}-{max(r['iters'] for r in rj)})"
      f" -> {B / (t1 - t0):.1f} plans/s", flush=True)
# configs[3] proper: B4 four-vehicle joint plans (default 32; argv[2])
B4 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
scen = []
for b in range(B4):
    base = (b % (B // 4)) * 4
    sing = [res[base + i] for i in range(4)]
    scen.append(dict(init_poses=[init[base + i] for i in range(4)], tubes=[tubes[a] for a in agents], guesses=[s["traj"].reshape(-1, 7) for s in sing],
                     dt0=float(np.mean([s["dt"] for s in sing])), final_headings=[fh[a] for a in agents]))
t0 = time.time()
rj = engine.joint_colloc_batch(sp, scen, max_iter=300, **MU)
t1 = time.time()
print(f"{B4} four-vehicle joint plans: {t1 - t0:.2f} s ({sum(r['status'] == 0 for r in rj)} converged, iterations {sorted(r['iters'] for r in rj)})"
      f" -> {B4 / (t1 - t0):.2f} plans/s", flush=True)
