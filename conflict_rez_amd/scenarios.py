"""Synthetic inputs for the batched MPC engine.

The 4-vehicle parking-lot scenario: the six static obstacle boxes of the reference
(`compute_obstacles`), one planned reference trajectory per vehicle (a table sampled every
dt = 0.1 s) and, per scenario, a random start time on those references plus state noise.
Two reference tables are package data (`load_reference_table(kind)`):
  "planned"   `conflict_rez_amd/data/refs_4v_planned.npz`: what `VehicleFollower.plan_single_path` produces (vehicle_follower.py:91-138:
              state_ws -> dual_ws -> collocation plan with free dt) for the four vehicles of the synthetic strategy
              (`strategy.generate_strategy(4)`) on the GPU planning kernels, sampled every dt (7.8-15.7 s plans); written by
              `planned_reference_table` on an MI355X and reproduced by it in the GPU tests.  SURVEY.md 8d config 3 names this table.
  "state_ws"  `conflict_rez_amd/data/refs_4v.npz`: the four vehicles' `Vehicle.state_ws` warm-start plans (vehicle.py:99-231,
              tube-constrained, 18-30 s long); the table of rounds 1-2, and the one the MPC goldens were generated on.
The MPC's own NLP never sees how a table was made.
"""
import os

import numpy as np

from .control.compute_sets import compute_obstacles
from .engine import ProblemSpec
from .obstacle_types import GeofenceRegion
from .vehicle_types import VehicleBody, VehicleConfig

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_REFS = os.path.join(_DATA, "refs_4v.npz")
_TABLES = {"state_ws": _REFS, "planned": os.path.join(_DATA, "refs_4v_planned.npz")}


def parking_lot_spec(n_nbr=3, N=30, dt=0.1, n_obs=6, dmin=0.05):
    """`ProblemSpec` of the reference's MPC (vehicle_follower.py:146) on the reference's map.
    n_obs = 4 keeps obstacles 0,1,3,4 (BASELINE.json config 2 wording)."""
    obs = compute_obstacles()
    if n_obs == 4:
        obs = [obs[i] for i in (0, 1, 3, 4)]
    elif n_obs != 6:
        obs = obs[:n_obs]
    return ProblemSpec.from_objects(obs, VehicleBody(), VehicleConfig(), GeofenceRegion(), n_nbr=n_nbr, N=N, dt=dt, dmin=dmin)


def load_reference_table(path=None, kind="state_ws"):
    """[V, T, 7] planned trajectories (x,y,psi,v,delta,a,w) sampled every dt; held at the goal.  kind: "state_ws" or "planned"
    (module docstring); `path` overrides."""
    d = np.load(path or _TABLES[kind])
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


def _quad_distance(P, Q):
    """Distance of convex quadrilaterals P, Q [..., 4, 2] (vertex arrays), 0 when they touch or overlap: the smallest
    vertex-to-edge distance of the two boundaries, or 0 if a vertex of one lies inside the other / two edges cross."""
    def point_edges(A, B):  # squared distance of every vertex of A to every edge of B -> [..., 4, 4]
        b0, e = B, np.roll(B, -1, axis=-2) - B
        w = A[..., :, None, :] - b0[..., None, :, :]
        t = np.clip((w * e[..., None, :, :]).sum(-1) / np.maximum((e * e).sum(-1)[..., None, :], 1e-300), 0.0, 1.0)
        d = w - t[..., None] * e[..., None, :, :]
        return (d * d).sum(-1)

    def inside(A, B):  # a vertex of A inside B (either orientation)
        e = np.roll(B, -1, axis=-2) - B
        w = A[..., :, None, :] - B[..., None, :, :]
        cr = e[..., None, :, 0] * w[..., 1] - e[..., None, :, 1] * w[..., 0]
        return ((cr >= 0).all(-1) | (cr <= 0).all(-1)).any(-1)

    def edges_cross(A, B):
        a0, a1, b0, b1 = A[..., :, None, :], np.roll(A, -1, axis=-2)[..., :, None, :], B[..., None, :, :], np.roll(B, -1, axis=-2)[..., None, :, :]
        def orient(p, q, r):
            return (q[..., 0] - p[..., 0]) * (r[..., 1] - p[..., 1]) - (q[..., 1] - p[..., 1]) * (r[..., 0] - p[..., 0])
        return ((orient(a0, a1, b0) * orient(a0, a1, b1) < 0) & (orient(b0, b1, a0) * orient(b0, b1, a1) < 0)).any((-1, -2))

    d = np.sqrt(np.minimum(point_edges(P, Q).min((-1, -2)), point_edges(Q, P).min((-1, -2))))
    return np.where(inside(P, Q) | inside(Q, P) | edges_cross(P, Q), 0.0, d)


def start_clearances(spec: ProblemSpec, table, k0, noise):
    """[S, V]: distance of every vehicle's body at its measured start state (table pose at k0 + noise) from the nearest static
    obstacle or other vehicle (0 = touching or overlapping).  A vehicle that starts closer than dmin - constr_viol_tol has an
    infeasible first NLP (status 4: the pose of stage 0 is pinned to the measured state, vehicle_follower.py:194-199, :280-290)."""
    S, V = len(k0), table.shape[0]
    x = table[np.arange(V)[None, :], np.asarray(k0)[:, None], :3] + noise[..., :3]  # [S, V, 3]
    g = np.asarray(spec.g, float)
    BV = np.array([[g[0], g[1]], [-g[2], g[1]], [-g[2], -g[3]], [g[0], -g[3]]])
    c, s_ = np.cos(x[..., 2]), np.sin(x[..., 2])
    W = x[..., None, :2] + np.stack([c[..., None] * BV[:, 0] - s_[..., None] * BV[:, 1], s_[..., None] * BV[:, 0] + c[..., None] * BV[:, 1]], -1)  # [S, V, 4, 2]
    out = np.full((S, V), np.inf)
    for j in range(spec.n_obs):  # static obstacles: vertices from the H-representation (4 half-planes, adjacent rows meet)
        A, b = np.asarray(spec.A_obs[j], float), np.asarray(spec.b_obs[j], float)
        ang = np.argsort(np.arctan2(A[:, 1], A[:, 0]))
        PV = np.array([np.linalg.solve(A[[ang[i], ang[(i + 1) % 4]]], b[[ang[i], ang[(i + 1) % 4]]]) for i in range(4)])
        out = np.minimum(out, _quad_distance(W, PV[None, None]))
    for a in range(V):
        for b_ in range(a + 1, V):
            d = _quad_distance(W[:, a], W[:, b_])
            out[:, a] = np.minimum(out[:, a], d); out[:, b_] = np.minimum(out[:, b_], d)
    return out


def start_box_excess(spec: ProblemSpec, table, k0, noise):
    """[S, V]: how far every vehicle's measured start state (table state at k0 + noise) lies OUTSIDE the boxes of the MPC's NLP on
    x, y, v, delta (vehicle_follower.py:205-240; 0 = inside).  The state of stage 0 is pinned to the measurement (:194-199) and
    bounded like every other stage, so an excess above constr_viol_tol makes the first NLP infeasible (status 4) -- e.g. a plan that
    drives at the speed limit, measured with +0.05 m/s of noise."""
    V = table.shape[0]
    x = table[np.arange(V)[None, :], np.asarray(k0)[:, None], :5] + noise  # [S, V, 5]
    b = np.asarray(spec.bounds, float).reshape(6, 2)
    out = np.zeros(x.shape[:2])
    for q, c in enumerate((0, 1, 3, 4)):
        out = np.maximum(out, np.maximum(b[q, 0] - x[..., c], x[..., c] - b[q, 1]))
    return out


def sample_scenarios(S, table, seed=2024, horizon_margin=30, noise=(0.05, 0.05, 0.02, 0.05, 0.0), spec=None, margin=0.01, box_tol=1e-2):
    """Start sample k0[S] ~ U[0, T - margin) and state noise [S, V, 5] (sigma_xy 0.05 m,
    sigma_psi 0.02 rad, sigma_v 0.05 m/s).
    spec (optional): only FEASIBLE starts -- a scenario in which some vehicle's measured state is closer than dmin - margin to an
    obstacle or to another vehicle, or outside the NLP's state boxes by more than box_tol (`start_box_excess`: round 4 found the
    planned table's vehicle 1 at the speed limit, so that half of its noisy starts were), is drawn again (start time and noise,
    same generator, up to 50 times): such a state is not one `VehicleFollower` can be in (its first NLP is infeasible, status 4).  Without spec: the draws of rounds 1-2 as they come
    (the MPC goldens and the closed-loop tests were generated on those)."""
    rng = np.random.default_rng(seed)
    V, T = table.shape[0], table.shape[1]
    k0 = rng.integers(0, max(T - horizon_margin, 1), size=S).astype(np.int32)
    nz = rng.normal(0.0, 1.0, size=(S, V, 5)) * np.asarray(noise)
    if spec is not None:
        for _ in range(50):
            bad = np.flatnonzero((start_clearances(spec, table, k0, nz).min(1) < spec.dmin - margin)
                                 | (start_box_excess(spec, table, k0, nz).max(1) > box_tol))
            if not len(bad):
                break
            k0[bad] = rng.integers(0, max(T - horizon_margin, 1), size=len(bad)).astype(np.int32)
            nz[bad] = rng.normal(0.0, 1.0, size=(len(bad), V, 5)) * np.asarray(noise)
    return k0, nz


def lane_sampler(spec: ProblemSpec, B=256, seed=1234, speed=1.0, radius=12.0):
    """BASELINE.md section 4, config 2 (SURVEY.md 8d "Config 2"): B independent single-vehicle MPC-form problems (N stages, no
    neighbours) in the lane between the two rows of parking spots: x0 ~ U[5, 30], y0 ~ U[15, 20], psi0 in {0, pi} + N(0, 0.05),
    v0 ~ U[-1, 1], delta0 = 0, `default_rng(seed)`; the reference is a straight segment (even instances) or an arc of `radius` m
    bending back towards the lane's centre line (odd instances), driven at `speed` m/s along the lane direction from the start
    position.  Returns x0 [B, 5], ref [B, 3, N], zu [B, 7, N] (warm start = the reference at the measured speed, inputs zero)."""
    rng = np.random.default_rng(seed)
    N, dt = spec.N, spec.dt
    x = rng.uniform(5.0, 30.0, B); y = rng.uniform(15.0, 20.0, B)
    lane = rng.integers(0, 2, B) * np.pi
    psi = lane + rng.normal(0.0, 0.05, B)
    v = rng.uniform(-1.0, 1.0, B)
    x0 = np.stack([x, y, psi, v, np.zeros(B)], 1)
    s_ = speed * dt * (np.arange(N) + 1.0)
    ref = np.zeros((B, 3, N)); zu = np.zeros((B, 7, N))
    for b in range(B):
        if b % 2 == 0:
            ref[b, 0] = x[b] + s_ * np.cos(lane[b]); ref[b, 1] = y[b]; ref[b, 2] = lane[b]
        else:  # an arc towards the centre line y = 17.5
            k = (1.0 if y[b] < 17.5 else -1.0) * (1.0 if lane[b] == 0.0 else -1.0) / radius
            th = lane[b] + k * s_
            ref[b, 0] = x[b] + (np.sin(th) - np.sin(lane[b])) / k; ref[b, 1] = y[b] - (np.cos(th) - np.cos(lane[b])) / k; ref[b, 2] = th
        zu[b, :3] = ref[b]; zu[b, 3] = v[b]
    return x0, ref, zu


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
