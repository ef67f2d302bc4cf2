"""cfz_colloc's eliminations against each other on the GPU: band (structured = 0) and cfz_jstruct.inl's scheme (1, the default) on the
256-plan launch of configs[1]: status, iterations, trajectories, time.  python tools/struct_gpu_check.py [B]
(Round 4's cfz_struct.inl scheme, structured = 2, was a third leg until round 6 removed it from the library.)"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
agents = sorted(hist)
sp = scenarios.parking_lot_spec(n_nbr=0, N=2, n_obs=int(os.environ.get("NOBS", 6)))
tubes = {a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents}
fh = {a: float(paths[a][-1, 2]) for a in agents}
tau = np.append(0.0, [0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
def guess_of(ws, n_sets):
    N = 5 * (n_sets - 1)
    t = 0.1 * np.arange(len(ws)); ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N
rng = np.random.default_rng(0)
who = [agents[i % 4] for i in range(B)]
init = [paths[a][0] + (np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] if i >= 4 else 0.0) for i, a in enumerate(who)]
ws = engine.state_ws(init, [tubes[a] for a in who], [paths[a] for a in who], [fh[a] for a in who], shrink_tube=0.5)
gs = [guess_of(w["traj"], len(tubes[a]) + 1) for w, a in zip(ws, who)]
args = (sp, init, [tubes[a] for a in who], [g[0] for g in gs], [g[1] for g in gs], [fh[a] for a in who])
res = {}
for name, kw in (("band", dict(structured=0)), ("jstruct", dict(structured=1)), ("jstruct", dict(structured=1))):
    t0 = time.time(); r = engine.colloc(*args, max_iter=400, **kw); t1 = time.time()
    res[name] = r
    its = np.array([x["iters"] for x in r])
    print(f"{name}: {B} plans {t1 - t0:.3f} s, converged {sum(x['status'] == 0 for x in r)}, iterations {its.min()}-{its.max()} mean {its.mean():.1f}", flush=True)
for nm in ("jstruct",):
    a, b = res["band"], res[nm]
    same = sum(x["iters"] == y["iters"] and x["status"] == y["status"] for x, y in zip(a, b))
    dd = [float(np.abs(x["traj"] - y["traj"]).max()) for x, y in zip(a, b) if x["iters"] == y["iters"]]
    print(nm, "against band: other iteration counts:", [(i, who[i], x["iters"], y["iters"]) for i, (x, y) in enumerate(zip(a, b)) if x["iters"] != y["iters"]][:10])
    print(f"   same status and iteration count: {same} of {B}; largest trajectory difference among those {max(dd):.2e}; dt difference {max(abs(x['dt'] - y['dt']) for x, y in zip(a, b)):.2e}")
