"""Fixture generator (offline, ~20 min): tests/golden/mpc_independent_more.npz -- 32 more full-size MPC-step instances with
their optimum from the independent solver (see make_independent.py): 16 from the bench's scenario sampler (another seed) with at
least one ACTIVE collision row at the optimum, 16 with a vertex-vertex closest pair active (a parked intruder's corner in the
ego's path).  Used as a population (how many does the engine solve to the same optimum), not instance by instance.

    python tests/golden/make_independent_more.py
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


def main(n_act=16, n_vv=16):
    spec = scenarios.parking_lot_spec()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
    table, _ = scenarios.load_reference_table()
    k0, noise = scenarios.sample_scenarios(256, table, seed=97531)
    x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
    rows = []

    def consider(x0_, ref_, nbr_, zu_, want_vv):
        r = im.solve(ospec, x0_, ref_, nbr_, zu_)
        if r["status"] not in (0, 8) or r["eq"] > 1e-8 or r["ineq"] < -1e-8:
            return False
        nlp = im.GeometricMpc(ospec, x0_, ref_, nbr_)
        poses = r["zu"][:3].T
        dist, face = nlp.separations(poses), face_separation(nlp, poses)
        active = dist < ospec.dmin + 1e-6
        vv = active & (face < dist - 1e-4)
        if not active.any() or want_vv != bool(vv.any()):
            return False
        rows.append(dict(x0=x0_, ref=ref_, nbr=nbr_, zu=zu_, sol=r["zu"], cost=r["cost"], n_active=int(active.sum()), n_vv=int(vv.sum())))
        print(f"instance {len(rows) - 1}: cost {r['cost']:.6f}, SLSQP {r['iters']} iterations, active rows {int(active.sum())}, vertex-vertex active {int(vv.sum())}", flush=True)
        return True

    # pre-filter with the cheap geometric check of the warm start: only instances that start within 0.3 m of something
    for b in range(len(x0)):
        if sum(r["n_vv"] == 0 for r in rows) >= n_act:
            break
        nlp = im.GeometricMpc(ospec, x0[b], ref[b], nbr[b])
        if nlp.separations(zu[b][:3].T).min() > 0.3:
            continue
        consider(x0[b], ref[b], nbr[b], zu[b], want_vv=False)
    rng = np.random.default_rng(11)
    tries = 0
    while sum(r["n_vv"] > 0 for r in rows) < n_vv and tries < 400:
        tries += 1
        b = int(rng.integers(0, len(x0)))
        k = int(rng.integers(8, 20))
        px, py, ps = ref[b][0, k], ref[b][1, k], ref[b][2, k]
        side = rng.choice([-1.0, 1.0])
        theta = ps + side * rng.uniform(0.5, 1.1)
        off = 0.9 + rng.uniform(-0.05, 0.25)
        cx, cy = px - np.sin(ps) * side * off + np.cos(ps) * 3.3, py + np.cos(ps) * side * off + np.sin(ps) * 3.3
        c, s = np.cos(theta), np.sin(theta)
        corner = np.array([-0.6, -side * 0.9])
        tx, ty = cx - (c * corner[0] - s * corner[1]), cy - (s * corner[0] + c * corner[1])
        nb2 = nbr[b].copy()
        nb2[0, 0], nb2[0, 1], nb2[0, 2] = tx, ty, theta
        consider(x0[b], ref[b], nb2, zu[b], want_vv=True)
    out = {k: np.stack([r[k] for r in rows]) for k in ("x0", "ref", "nbr", "zu", "sol")}
    out["cost"] = np.array([r["cost"] for r in rows])
    out["n_active"] = np.array([r["n_active"] for r in rows]); out["n_vv"] = np.array([r["n_vv"] for r in rows])
    out["A_obs"], out["b_obs"] = spec.A_obs, spec.b_obs
    np.savez_compressed(os.path.join(HERE, "mpc_independent_more.npz"), **out)
    print("wrote", len(rows), "instances", "vv", int((out["n_vv"] > 0).sum()))


if __name__ == "__main__":
    main()
