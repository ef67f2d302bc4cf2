"""Fixture generator (offline, ~6 min): tests/golden/mpc_independent_turn.npz -- 24 full-size MPC-step instances whose reference
TURNS inside the horizon (|delta psi| > 0.4 rad), pushed 0.1-1.6 m sideways towards the parking-lot furniture so that the ego
swings its corners past static boxes (six instances with a vertex-vertex contact with a static obstacle at the optimum).  Some
warm starts are deep inside an obstacle's clearance (up to 0.5 m): those need what IPOPT's restoration phase does.

    python tests/golden/make_independent_turn.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from conflict_rez_amd import scenarios  # noqa: E402
from make_independent import face_separation  # noqa: E402
from oracle import independent_mpc as im  # noqa: E402
from oracle.mpc_nlp import MpcSpec  # noqa: E402


def main(n_want=24):
    spec = scenarios.parking_lot_spec()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
    table, _ = scenarios.load_reference_table()
    k0, noise = scenarios.sample_scenarios(512, table, seed=1357)
    x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
    turn = np.nonzero(np.abs(ref[:, 2, -1] - ref[:, 2, 0]) > 0.4)[0]
    rng = np.random.default_rng(9)
    rows, tries = [], 0
    while len(rows) < n_want and tries < 3000:
        tries += 1
        b = int(rng.choice(turn))
        side, off = rng.choice([-1.0, 1.0]), rng.uniform(0.1, 1.6)
        psi = ref[b][2]
        sx, sy = -np.sin(psi) * side * off, np.cos(psi) * side * off
        r2, z2, s2 = ref[b].copy(), zu[b].copy(), x0[b].copy()
        r2[0] += sx; r2[1] += sy
        z2[0] += sx; z2[1] += sy
        s2[0] += sx[0]; s2[1] += sy[0]
        nlp = im.GeometricMpc(ospec, s2, r2, nbr[b])
        first = np.vstack([s2[:3], z2[:3].T[1:]])
        if nlp.separations(first)[0].min() < ospec.dmin + 0.02 or nlp.separations(z2[:3].T)[:, :6].min() > 0.25:
            continue
        r = im.solve(ospec, s2, r2, nbr[b], z2)
        if r["status"] not in (0, 8) or r["eq"] > 1e-8 or r["ineq"] < -1e-8:
            continue
        poses = r["zu"][:3].T
        dist, face = nlp.separations(poses), face_separation(nlp, poses)
        active = dist < ospec.dmin + 1e-6
        if not active[:, :6].any():
            continue
        vv = active & (face < dist - 1e-4)
        rows.append(dict(x0=s2, ref=r2, nbr=nbr[b], zu=z2, sol=r["zu"], cost=r["cost"], n_active=int(active.sum()), n_vv=int(vv.sum())))
        print(f"instance {len(rows) - 1}: cost {r['cost']:.6f}, SLSQP {r['iters']} iterations, active rows {int(active.sum())} "
              f"(static {int(active[:, :6].sum())}), vertex-vertex active {int(vv.sum())}", flush=True)
    out = {k: np.stack([r[k] for r in rows]) for k in ("x0", "ref", "nbr", "zu", "sol")}
    out["cost"] = np.array([r["cost"] for r in rows])
    out["n_active"] = np.array([r["n_active"] for r in rows]); out["n_vv"] = np.array([r["n_vv"] for r in rows])
    out["A_obs"], out["b_obs"] = spec.A_obs, spec.b_obs
    np.savez_compressed(os.path.join(HERE, "mpc_independent_turn.npz"), **out)
    print("wrote", len(rows), "instances, vertex-vertex", int((out["n_vv"] > 0).sum()), "tries", tries)


if __name__ == "__main__":
    main()
