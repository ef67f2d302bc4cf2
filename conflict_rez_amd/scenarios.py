"""Synthetic inputs for the batched MPC engine.

The 4-vehicle parking-lot scenario: the six static obstacle boxes of the reference
(`compute_obstacles`), one planned reference trajectory per vehicle (a table sampled every
dt = 0.1 s) and, per scenario, a random start time on those references plus state noise.
The default reference table is package data, `conflict_rez_amd/data/refs_4v.npz`: the four vehicles' `Vehicle.state_ws`
plans (vehicle.py:99-231, tube-constrained, 18-30 s long) of the synthetic strategy (`strategy.generate_strategy(4)`),
sampled every dt.  `planned_reference_table` builds the table the way `VehicleFollower.plan_single_path` does
(vehicle_follower.py:91-138): state_ws -> dual_ws -> collocation plan with free dt on the GPU planning kernels, then
sampled every dt (8-16 s plans).  The MPC's own NLP never sees how a table was made.
"""
import os

import numpy as np

from .control.compute_sets import compute_obstacles
from .engine import ProblemSpec
from .obstacle_types import GeofenceRegion
from .vehicle_types import VehicleBody, VehicleConfig

_REFS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "refs_4v.npz")


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


def planned_reference_table(dt=0.1, pad=30, device=0):
    """[V, T, 7] reference table from the build's own planning chain on the GPU ("run the build's
    own single-vehicle plan for the 4 agents of the synthetic strategy once"): per vehicle `cfz_state_ws`, then the
    collocation plan `cfz_colloc` from it, evaluated every `dt` with `Vehicle.interpolate_states` (degree-5 Lagrange
    interpolant per interval, vehicle.py:722-829) and held at the goal for `pad` further samples.  Returns
    (table, lengths, info); raises RuntimeError if a plan does not converge (as the reference's IPOPT call would)."""
    import tempfile

    from . import strategy as strat
    from .control.vehicle_follower import VehicleFollower
    from .pytypes import VehicleState

    hist = strat.generate_strategy(4)
    trajs, info = [], {}
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        from .control.compute_sets import interp_along_sets
        from .vehicle_types import VehicleBody

        paths = interp_along_sets(fn, VehicleBody(), 30)
        for a in sorted(hist):
            v = VehicleFollower(rl_file_name=fn, agent=a, color={"front": (1, 0, 0), "back": (0, 1, 0)}, init_offset=VehicleState(),
                                final_heading=float(paths[a][-1, 2]))
            v.plan_single_path(spline_ws=True)
            if not getattr(v, "plan_refined", False):
                raise RuntimeError(f"collocation plan of {a} did not converge")
            t_end = float(v.reference_traj.t[-1])
            tt = np.arange(0.0, t_end + 0.5 * dt, dt)
            r = v.interpolate_states(tt)
            trajs.append(np.stack([r.x, r.y, r.psi, r.v, r.u_steer, r.u_a, r.u_steer_dot], 1))
            info[a] = dict(t_end=t_end, samples=len(tt))
    T = max(len(t) for t in trajs) + pad
    table = np.zeros((len(trajs), T, 7))
    for i, tr in enumerate(trajs):
        table[i, : len(tr)] = tr
        table[i, len(tr):, :3] = tr[-1, :3]  # goal pose held, at rest
    return table, np.array([len(t) for t in trajs]), info


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
