"""One-off wide parity sweep on the GPU box: 4096 cold instances and a carried closed loop (64 scenarios x 8
iterations) through the C ABI against the oracle's C port (status, iteration count, trajectory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios
from oracle import port
from oracle.dynamics import plant_step
from oracle.mpc_nlp import MpcSpec

spec = scenarios.parking_lot_spec()
ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
table, _ = scenarios.load_reference_table()
V, T, N = table.shape[0], table.shape[1], spec.N
k0, noise = scenarios.sample_scenarios(1024, table, seed=77)
x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
eng = engine.Engine(spec, max_batch=4096)
out = eng.solve(x0, ref, nbr, zu, want_duals=False)
bad, worst = 0, 0.0
t0 = time.time()
for b in range(len(x0)):
    r = port.solve(ospec, x0[b], ref[b], nbr[b], zu[b].T.copy())
    if (r["status"], r["iters"]) != (out["status"][b], out["iters"][b]):
        bad += 1
    elif r["status"] == 0:
        worst = max(worst, np.abs(r["p"].T[:5] - out["zu"][b][:5]).max())
print(f"cold batch: {len(x0)} instances, status/iteration mismatches {bad}, max |dz| {worst:.2e}, port {time.time() - t0:.1f} s", flush=True)

S, K = 64, 8
k0, noise = scenarios.sample_scenarios(S, table, seed=78)
eng.loop_init(table, k0, noise)
state = np.zeros((S, V, 5)); pred = np.zeros((S, V, 7, N))
for s in range(S):
    for v in range(V):
        pred[s, v] = table[v, np.minimum(k0[s] + np.arange(N), T - 1), :].T
        state[s, v] = table[v, k0[s], :5] + noise[s, v]
carry = [[None] * V for _ in range(S)]
adv = np.minimum(np.arange(N) + 1, N - 1)
for t in range(K):
    eng.loop_step()
    g = eng.loop_get()
    newp = pred.copy(); mism = 0
    for s in range(S):
        for v in range(V):
            kr = np.minimum(k0[s] + t + np.arange(N), T - 1)
            nb = np.stack([pred[s, u][:3][:, adv] for u in range(V) if u != v])
            w = pred[s, v][:, adv]
            r = port.solve(ospec, state[s, v], table[v, kr, :3].T, nb, w.T.copy(), carry=carry[s][v])
            carry[s][v] = r["carry"]
            newp[s, v] = r["p"].T if r["status"] == 0 else w
            state[s, v] = plant_step(state[s, v], newp[s, v][5:7, 0], spec.dt, spec.wb)
            mism += (r["status"], r["iters"]) != (g["status"][s, v], g["iters"][s, v])
    pred = newp
    print(f"closed loop t={t}: status/iteration mismatches {mism} of {S * V}, max |state diff| {np.abs(state - g['state']).max():.2e}", flush=True)
