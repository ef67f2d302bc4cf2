"""Oracle (test infrastructure): the closed loop of `MultiDistributedFollower.solve` (reference
confrez/control/vehicle_follower.py:630-663) replayed on the host with the plain-C port of the solver.

Per MPC iteration and scenario: every vehicle's prediction is copied first (Jacobi exchange, :636-637), then every
vehicle steps (:642-647): `_adv_onestep` shift of its own and its neighbours' predictions (:413-426, :444-476), solve
(:479) started from the multipliers its previous solve left (`cfz_port_solve_carry`; the reference hands the old duals
to `opti.set_initial`, :458-464), read-back (:484-500) or shift fallback (:501-524), plant over dt (:528-543).

Used by tests/test_gpu_parity.py (the device loop must reproduce it solve by solve) and by bench.py's `cpu_baseline`
leg (the same workload timed on the host cores).  Never imported by the product package.
"""
import numpy as np

from . import port
from .dynamics import plant_step
from .mpc_nlp import MpcSpec


def seed(table, k0, noise, N):
    """State and first prediction of every (scenario, vehicle) as `get_current_ref` seeds them (:397-400)."""
    S, V, T = len(k0), table.shape[0], table.shape[1]
    state = np.zeros((S, V, 5)); pred = np.zeros((S, V, 7, N))
    for s in range(S):
        for v in range(V):
            pred[s, v] = table[v, np.minimum(k0[s] + np.arange(N), T - 1), :].T
            state[s, v] = table[v, k0[s], :5] + noise[s, v]
    return state, pred


def replay(ospec: MpcSpec, table, k0, noise, steps, dt=0.1, wb=2.5, carry_duals=True, opt=None):
    """Generator: after every iteration yields (state [S,V,5], pred [S,V,7,N], status [S,V], iters [S,V])."""
    S, V, T, N = len(k0), table.shape[0], table.shape[1], ospec.N
    state, pred = seed(table, k0, noise, N)
    carry = [[None] * V for _ in range(S)]
    adv = np.minimum(np.arange(N) + 1, N - 1)
    for t in range(steps):
        newp = pred.copy()
        status = np.zeros((S, V), int); iters = np.zeros((S, V), int)
        for s in range(S):
            kr = np.minimum(k0[s] + t + np.arange(N), T - 1)
            for v in range(V):
                nb = np.stack([pred[s, u][:3][:, adv] for u in range(V) if u != v]) if V > 1 else np.zeros((0, 3, N))
                w = pred[s, v][:, adv]
                r = port.solve(ospec, state[s, v], table[v, kr, :3].T, nb, w.T.copy(), **({} if opt is None else {"opt": opt}), carry=carry[s][v] if carry_duals else None)
                carry[s][v] = r["carry"]
                newp[s, v] = r["p"].T if r["status"] == 0 else w
                state[s, v] = plant_step(state[s, v], newp[s, v][5:7, 0], dt, wb)
                status[s, v], iters[s, v] = r["status"], r["iters"]
        pred = newp
        yield state.copy(), pred.copy(), status, iters
