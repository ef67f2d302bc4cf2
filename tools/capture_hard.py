"""Runs the closed loop stepwise and saves the inputs of solves that needed many IPM iterations
(gpurun_out/hard_cases.npz) so that they can be replayed with the oracle on the CPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios

S, K = int(sys.argv[1]), int(sys.argv[2])
# third argument: iteration threshold, or inst:b1,b2,..@t1,t2,.. to record given instances at given iterations
want = None
if sys.argv[3].startswith("inst:"):
    a, b = sys.argv[3][5:].split("@")
    want = ([int(x) for x in a.split(",")], [int(x) for x in b.split(",")])
    thresh = 10 ** 9
else:
    thresh = int(sys.argv[3])
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
V, T, N = table.shape[0], table.shape[1], spec.N
k0, noise = scenarios.sample_scenarios(S, table, seed=2024)
eng = engine.Engine(spec, max_batch=S * 4)
eng.loop_init(table, k0, noise)
cases = []
for t in range(K):
    g0 = eng.loop_get()
    eng.loop_step()
    g1 = eng.loop_get()
    it, st = g1["iters"].reshape(S, V), g1["status"].reshape(S, V)
    picks = list(zip(*np.nonzero(it >= thresh)))
    if want is not None and t in want[1]:
        picks = [(b // V, b % V) for b in want[0]]
    for s, v in picks:
        if len(cases) >= 16:
            break
        adv = np.minimum(np.arange(N) + 1, N - 1)
        pred = g0["pred"].reshape(S, V, 7, N)
        kr = np.minimum(k0[s] + t + np.arange(N), T - 1)
        cases.append(dict(x0=g0["state"].reshape(S, V, 5)[s, v], ref=table[v, kr, :3].T.copy(),
                          nbr=np.stack([pred[s, u][:3][:, adv] for u in range(V) if u != v]), zu=pred[s, v][:, adv],
                          iters=it[s, v], status=st[s, v], t=t, s=s, v=v))
        print("captured t", t, "scenario", s, "vehicle", v, "iters", it[s, v], "status", st[s, v], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/hard_cases.npz", **{k: np.array([c[k] for c in cases]) for k in cases[0]})
