"""Fixture generator (run offline, ~2 min): tests/golden/mpc_independent.npz -- full-size MPC-step instances (N = 30, six
obstacles, three neighbours) with their optimum from an INDEPENDENT solver on an independent statement of the reference's
NLP (oracle/independent_mpc.py: polygon distances instead of OBCA duals, scipy SLSQP).  Eight instances of the bench's
scenario sampler (several with an active collision row) and instances built so that a VERTEX-VERTEX closest pair is active
at the optimum -- the one geometry where the engine's face-normal certificates are a strict restriction of the
reference's constraint: a parked intruder replaces one neighbour, its corner reaching into the ego's path.

    python tests/golden/make_independent.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from conflict_rez_amd import scenarios  # noqa: E402
from oracle import independent_mpc as im  # noqa: E402
from oracle.mpc_nlp import MpcSpec  # noqa: E402


def face_separation(nlp, poses):
    """[N, nb] best FACE-NORMAL separation (what the engine certifies), for telling vertex-vertex pairs apart."""
    W = im._body_vertices(poses[:, 0], poses[:, 1], poses[:, 2], nlp.spec.g)

    def depth(A, B):
        e = np.roll(A, -1, axis=-2) - A
        n = np.stack([e[..., 1], -e[..., 0]], -1)
        n = n / np.linalg.norm(n, axis=-1, keepdims=True)
        off = (n * A).sum(-1)
        return ((B[..., None, :, :] * n[..., :, None, :]).sum(-1) - off[..., :, None]).min(-1).max(-1)

    P = np.concatenate([np.broadcast_to(nlp.obs[None], (nlp.N,) + nlp.obs.shape), np.moveaxis(nlp.nbv, 0, 1)], 1)
    Wb = np.broadcast_to(W[:, None], P.shape)
    return np.maximum(depth(P, Wb), depth(Wb, P))


def main():
    spec = scenarios.parking_lot_spec()
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=3)
    table, _ = scenarios.load_reference_table()
    k0, noise = scenarios.sample_scenarios(64, table, seed=4242)
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
        if want_vv != bool(vv.any()):
            return False
        rows.append(dict(x0=x0_, ref=ref_, nbr=nbr_, zu=zu_, sol=r["zu"], cost=r["cost"], n_active=int(active.sum()), n_vv=int(vv.sum())))
        print(f"instance {len(rows) - 1}: cost {r['cost']:.6f}, SLSQP {r['iters']} iterations, active rows {int(active.sum())}, vertex-vertex active {int(vv.sum())}")
        return True

    n_plain = n_act = 0
    for b in range(len(x0)):  # sampler instances: four without and four with an active collision row
        if n_plain + n_act == 8:
            break
        before = len(rows)
        if consider(x0[b], ref[b], nbr[b], zu[b], want_vv=False):
            if rows[-1]["n_active"] > 0 and n_act < 4:
                n_act += 1
            elif rows[-1]["n_active"] == 0 and n_plain < 4:
                n_plain += 1
            else:
                rows.pop()
        assert len(rows) <= before + 1
    # vertex-vertex instances: an intruder parked beside the reference path, turned by theta, its nearest corner
    # `gap` inside the corridor the ego's near corner sweeps
    rng = np.random.default_rng(7)
    tries = 0
    while sum(r["n_vv"] > 0 for r in rows) < 4 and tries < 200:
        tries += 1
        b = int(rng.integers(0, len(x0)))
        k = int(rng.integers(8, 20))
        px, py, ps = ref[b][0, k], ref[b][1, k], ref[b][2, k]
        side = rng.choice([-1.0, 1.0])
        theta = ps + side * rng.uniform(0.5, 1.1)
        # intruder's reference point so that one of its corners lies `off` beside the path point, ahead of the ego's corner
        off = 0.9 + rng.uniform(-0.05, 0.25)
        cx, cy = px - np.sin(ps) * side * off + np.cos(ps) * 3.3, py + np.cos(ps) * side * off + np.sin(ps) * 3.3
        c, s = np.cos(theta), np.sin(theta)
        corner = np.array([-0.6, -side * 0.9])  # rear corner on the path side, in the intruder's body frame
        tx, ty = cx - (c * corner[0] - s * corner[1]), cy - (s * corner[0] + c * corner[1])
        nb2 = nbr[b].copy()
        nb2[0, 0], nb2[0, 1], nb2[0, 2] = tx, ty, theta
        consider(x0[b], ref[b], nb2, zu[b], want_vv=True)
    assert sum(r["n_vv"] > 0 for r in rows) >= 2, "no vertex-vertex instance found"
    out = {k: np.stack([r[k] for r in rows]) for k in ("x0", "ref", "nbr", "zu", "sol")}
    out["cost"] = np.array([r["cost"] for r in rows])
    out["n_active"] = np.array([r["n_active"] for r in rows]); out["n_vv"] = np.array([r["n_vv"] for r in rows])
    out["A_obs"], out["b_obs"] = spec.A_obs, spec.b_obs
    np.savez_compressed(os.path.join(HERE, "mpc_independent.npz"), **out)
    print("wrote", len(rows), "instances")


if __name__ == "__main__":
    main()
