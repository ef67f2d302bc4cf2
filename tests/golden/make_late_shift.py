"""Instances on which the scaled row curvature cycles: vehicle 3 of scenario 477 (bench seed 2024) is carried far off its
reference and pressed into a corner of the lot.  Replays that closed loop with the C port (carried multipliers, as the
loop kernel does) and keeps the cold-start inputs of the steps whose solve needs >= 300 iterations without the late shift
(IpmOptions.shift_after = 0), with the solution of the full-KKT numpy oracle at the default shift_after.

    python tests/golden/make_late_shift.py      -> tests/golden/mpc_late_shift.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from conflict_rez_amd import scenarios  # noqa: E402
from oracle import ipm, port  # noqa: E402
from oracle.closed_loop import seed  # noqa: E402
from oracle.dynamics import plant_step  # noqa: E402
from oracle.mpc_nlp import MpcSpec, solve_mpc  # noqa: E402


def main(scenario=477, vehicle=3, keep=6):
    spec = scenarios.parking_lot_spec()
    table, _ = scenarios.load_reference_table()
    k0a, noisea = scenarios.sample_scenarios(1024, table, seed=2024)
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)
    k0, noise = k0a[scenario:scenario + 1], noisea[scenario:scenario + 1]
    V, T, N = 4, table.shape[1], ospec.N
    state, pred = seed(table, k0, noise, N)
    carry = [None] * V
    adv = np.minimum(np.arange(N) + 1, N - 1)
    noshift = ipm.IpmOptions(shift_after=0, err_stall_iters=0)  # neither the late shift nor the error-stall stop: the cycle runs to max_iter
    out = {k: [] for k in ("x0", "ref", "nbr", "zu", "iters_noshift", "iters_shift", "status_shift", "sol", "step")}
    for t in range(200):
        newp = pred.copy()
        kr = np.minimum(k0[0] + t + np.arange(N), T - 1)
        for v in range(V):
            nb = np.stack([pred[0, u][:3][:, adv] for u in range(V) if u != v])
            w = pred[0, v][:, adv]
            ref = table[v, kr, :3].T
            r = port.solve(ospec, state[0, v], ref, nb, w.T.copy(), carry=carry[v])
            if v == vehicle and len(out["x0"]) < keep:
                cold0 = port.solve(ospec, state[0, v], ref, nb, w.T.copy(), opt=noshift)
                cold = port.solve(ospec, state[0, v], ref, nb, w.T.copy())
                if cold0["iters"] >= 300 and cold["status"] == 0:
                    full = solve_mpc(ospec, state[0, v], ref, nb, w.copy())  # the stored answer: full-KKT numpy oracle
                    assert (full["status"], full["iters"]) == (cold["status"], cold["iters"]), (t, full["iters"], cold["iters"])
                    for k, a in zip(out, (state[0, v].copy(), ref, nb, w.copy(), cold0["iters"], full["iters"], full["status"], full["zu"], t)):
                        out[k].append(a)
            carry[v] = r["carry"]
            newp[0, v] = r["p"].T if r["status"] == 0 else w
            state[0, v] = plant_step(state[0, v], newp[0, v][5:7, 0], spec.dt, spec.wb)
        pred = newp
    out = {k: np.array(a) for k, a in out.items()}
    print({k: out[k].tolist() for k in ("step", "iters_noshift", "iters_shift", "status_shift")})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mpc_late_shift.npz"), A_obs=spec.A_obs, b_obs=spec.b_obs, **out)


def resolve():
    """Round 4: the stored inputs solved again after a change of the MPC algorithm (the closed loop they were recorded from has
    moved on): `iters_noshift` = iterations without the late shift AND without the dual regularisation of the rows (the cycle:
    600), `iters_nodc` = default late shift but no dual regularisation (round 3's algorithm), `iters_shift`, `status_shift`, `sol`
    = the full-KKT oracle at the defaults (delta_c = 1e-8)."""
    path = os.path.join(ROOT, "tests", "golden", "mpc_late_shift.npz")
    g = dict(np.load(path))
    ospec = MpcSpec(N=30, dt=0.1, A_obs=g["A_obs"], b_obs=g["b_obs"], n_nbr=3)
    cyc = ipm.IpmOptions(shift_after=0, err_stall_iters=0, reg_dual_rows=0.0)
    nodc = ipm.IpmOptions(reg_dual_rows=0.0)
    g["iters_nodc"] = np.zeros(len(g["x0"]), int)
    for b in range(len(g["x0"])):
        a = (g["x0"][b], g["ref"][b], g["nbr"][b], g["zu"][b])
        g["iters_noshift"][b] = port.solve(ospec, a[0], a[1], a[2], a[3].T.copy(), opt=cyc)["iters"]
        g["iters_nodc"][b] = port.solve(ospec, a[0], a[1], a[2], a[3].T.copy(), opt=nodc)["iters"]
        full = solve_mpc(ospec, *a)
        cold = port.solve(ospec, a[0], a[1], a[2], a[3].T.copy())
        assert (full["status"], full["iters"]) == (cold["status"], cold["iters"]), (b, full["iters"], cold["iters"])
        g["iters_shift"][b], g["status_shift"][b], g["sol"][b] = full["iters"], full["status"], full["zu"]
    print({k: g[k].tolist() for k in ("iters_noshift", "iters_nodc", "iters_shift", "status_shift")})
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    resolve() if "--resolve" in sys.argv else main()
