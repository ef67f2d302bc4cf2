"""HIP engine against the C port on the independent-solver populations and on a sample of the bench's own cold starts:
status, iteration count and trajectory of every instance (a quick GPU sanity run; the tests assert the same)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios
from oracle import ipm, port
from oracle.mpc_nlp import MpcSpec

gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
bad = tot = 0
for fx in ("mpc_independent.npz", "mpc_independent_more.npz", "mpc_independent_obs.npz", "mpc_independent_turn.npz"):
    d = np.load(os.path.join(gold, fx))
    B = len(d["x0"])
    spec = engine.ProblemSpec(N=30, dt=0.1, n_nbr=3, A_obs=d["A_obs"], b_obs=d["b_obs"])
    ospec = MpcSpec(N=30, dt=0.1, A_obs=d["A_obs"], b_obs=d["b_obs"], n_nbr=3)
    eng = engine.Engine(spec, max_batch=B)
    r = eng.solve(d["x0"], d["ref"], d["nbr"], d["zu"])
    for b in range(B):
        q = port.solve(ospec, d["x0"][b], d["ref"][b], d["nbr"][b], d["zu"][b].T.copy())
        ok = (q["status"], q["iters"]) == (int(r["status"][b]), int(r["iters"][b])) and (q["status"] != 0 or np.abs(r["zu"][b] - q["p"].T).max() < 1e-7)
        tot += 1; bad += not ok
        if not ok:
            print(fx, b, "port", q["status"], q["iters"], "hip", int(r["status"][b]), int(r["iters"][b]), "dz", np.abs(r["zu"][b] - q["p"].T).max())
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table(kind="planned")
ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)
k0, noise = scenarios.sample_scenarios(256, table, seed=7)  # raw starts: status 4 of both kinds, deep violations
x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
eng = engine.Engine(spec, max_batch=len(x0))
r = eng.solve(x0, ref, nbr, zu)
from collections import Counter
cnt = Counter()
for b in range(len(x0)):
    q = port.solve(ospec, x0[b], ref[b], nbr[b], zu[b].T.copy())
    cnt[q["status"]] += 1
    ok = (q["status"], q["iters"]) == (int(r["status"][b]), int(r["iters"][b])) and (q["status"] != 0 or np.abs(r["zu"][b] - q["p"].T).max() < 1e-7)
    tot += 1; bad += not ok
    if not ok:
        print("cold", b, "port", q["status"], q["iters"], "hip", int(r["status"][b]), int(r["iters"][b]), "dz", np.abs(r["zu"][b] - q["p"].T).max())
print("compared", tot, "mismatches", bad, "cold-start statuses", dict(cnt))
