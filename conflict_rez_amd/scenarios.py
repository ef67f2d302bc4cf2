"""Synthetic inputs for the batched MPC engine.

The 4-vehicle parking-lot scenario: the six static obstacle boxes of the reference
(`compute_obstacles`), one planned reference trajectory per vehicle (a table sampled every
dt = 0.1 s) and, per scenario, a random start time on those references plus state noise.
The reference tables come from `tests/golden/refs_4v.npz` (tube-constrained `state_ws`
plans of the synthetic strategy, produced offline -- see tests/golden/make_fixtures.py);
the MPC's own NLP never sees how they were made.
"""
import os

import numpy as np

from .control.compute_sets import compute_obstacles
from .engine import ProblemSpec
from .obstacle_types import GeofenceRegion
from .vehicle_types import VehicleBody, VehicleConfig

_REFS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "refs_4v.npz")


def parking_lot_spec(n_nbr=3, N=30, dt=0.1, n_obs=6, dmin=0.05):
    """`ProblemSpec` of the reference's MPC (vehicle_follower.py:146) on the reference's map.
    n_obs = 4 keeps obstacles 0,1,3,4 (BASELINE.json config 2 wording)."""
    obs = compute_obstacles()
    if n_obs == 4:
        obs = [obs[i] for i in (0, 1, 3, 4)]
    elif n_obs != 6:
        obs = obs[:n_obs]
    return ProblemSpec.from_objects(obs, VehicleBody(), VehicleConfig(), GeofenceRegion(), n_nbr=n_nbr, N=N, dt=dt, dmin=dmin)


def load_reference_table(path=None):
    """[V, T, 7] planned trajectories (x,y,psi,v,delta,a,w) sampled every dt; held at the goal."""
    d = np.load(path or _REFS)
    return d["table"].copy(), d["lengths"].copy()


def sample_scenarios(S, table, seed=2024, horizon_margin=30, noise=(0.05, 0.05, 0.02, 0.05, 0.0)):
    """Start sample k0[S] ~ U[0, T - margin) and state noise [S, V, 5] (sigma_xy 0.05 m,
    sigma_psi 0.02 rad, sigma_v 0.05 m/s)."""
    rng = np.random.default_rng(seed)
    V, T = table.shape[0], table.shape[1]
    k0 = rng.integers(0, max(T - horizon_margin, 1), size=S).astype(np.int32)
    nz = rng.normal(0.0, 1.0, size=(S, V, 5)) * np.asarray(noise)
    return k0, nz


def mpc_batch_from_table(spec: ProblemSpec, table, k0, noise):
    """Host arrays of one cold MPC step for S scenarios x V vehicles, instance order [s][v]:
    x0 [B,5], ref [B,3,N], nbr [B,V-1,3,N], zu [B,7,N] (the first `step()` of every vehicle:
    prediction = the planned trajectory incl. v, delta, a, w as `get_current_ref` seeds it
    (vehicle_follower.py:397-400), neighbours' predictions = theirs, all advanced by one)."""
    V, T, N = table.shape[0], table.shape[1], spec.N
    S = len(k0)
    B = S * V
    x0 = np.zeros((B, 5)); ref = np.zeros((B, 3, N)); nbr = np.zeros((B, V - 1, 3, N)); zu = np.zeros((B, 7, N))
    for s in range(S):
        idx = np.minimum(k0[s] + np.arange(N), T - 1)
        adv = np.minimum(np.arange(N) + 1, N - 1)
        preds = table[:, idx, :]  # [V,N,7]
        for v in range(V):
            b = s * V + v
            x0[b] = table[v, k0[s], :5] + noise[s, v]
            ref[b] = preds[v][:, :3].T
            zu[b] = preds[v][adv].T
            others = [u for u in range(V) if u != v]
            for o, u in enumerate(others):
                nbr[b, o] = preds[u][adv][:, :3].T
    return x0, ref, nbr, zu
