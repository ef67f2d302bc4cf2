"""Synthetic recorded-strategy files.

The reference's planners start from `<name>.pkl`, the (front, back) grid-cell history that its
DQN policy produced in the PettingZoo parking-lot environment
(`confrez/rl/record_states_history.py:10-31`).  Neither that file nor the policy weights ship
with the reference (`.gitignore:12`, `README.md:20`), so this module produces files of the
same format from a scripted planner that obeys the environment's own rules:

  * 14 x 14 grid of 2.5 m cells, walls / free lanes / free slots as `pklot_env.py:226-282`
  * the four start and goal configurations of `pklot_env.py:141-158`
  * the 7 actions of `pklot_env.py:131-139` with the kinematics of `move` :300-356
    (forward: back <- front, front <- front + round(cos,sin)(heading + turn);
     backward symmetric), a move into a wall keeps the vehicle in place.

Vehicles are planned one after another by time-expanded breadth-first search against the
space-time reservations of the vehicles planned before them (a cell may not be entered
within +-1 step of another vehicle using it), so the recorded strategy is conflict-free on
the grid, like a successful rollout of the learned policy.
"""
import pickle
from collections import deque
from itertools import product

import numpy as np

N_GRID, N_EDGE, N_CENTER = 14, 3, 8
_UPPER_WALL_X = (3, 4, 5, 7, 8, 10)
_LOWER_WALL_X = (3, 4, 5, 7, 9, 10)
ACTIONS = ((0, 0.0), (1, -np.pi / 4), (1, 0.0), (1, np.pi / 4), (-1, -np.pi / 4), (-1, 0.0), (-1, np.pi / 4))

# default priority and initial waits: the combination for which the four tube plans stay clear of
# each other in continuous space (see tests/golden/make_fixtures.py)
DEFAULT_ORDER = [1, 0, 2, 3]
DEFAULT_DELAYS = [2, 0, 0, 0]

AGENT_CONFIGS = (
    {"init_state": {"front": (6, 8), "back": (6, 7)}, "goal": {"front": (12, 6), "back": (11, 6)}},
    {"init_state": {"front": (8, 7), "back": (9, 7)}, "goal": {"front": (6, 3), "back": (6, 4)}},
    {"init_state": {"front": (6, 5), "back": (6, 4)}, "goal": {"front": (1, 7), "back": (2, 7)}},
    {"init_state": {"front": (5, 6), "back": (4, 6)}, "goal": {"front": (6, 10), "back": (6, 9)}},
)


def wall_cells():
    walls = set()
    hi = N_EDGE + N_CENTER
    walls |= set(product(range(N_GRID), range(hi, N_GRID)))  # top
    walls |= set(product(range(N_GRID), range(N_EDGE)))  # bottom
    walls |= set(product(range(N_EDGE), range(N_EDGE, hi)))  # left
    walls -= set(product(range(1, N_EDGE), range(N_EDGE + 3, N_EDGE + 5)))
    walls |= set(product(range(hi, N_GRID), range(N_EDGE, hi)))  # right
    walls -= set(product(range(hi, hi + 2), range(N_EDGE + 3, N_EDGE + 5)))
    for i in _UPPER_WALL_X:
        walls |= {(i, hi - 1), (i, hi - 2), (i, hi - 3)}
    for i in _LOWER_WALL_X:
        walls |= {(i, N_EDGE), (i, N_EDGE + 1), (i, N_EDGE + 2)}
    return walls


def move(front, back, action, walls):
    """One environment step for one vehicle; returns (front, back)."""
    d, a = ACTIONS[action]
    if d == 0:
        return front, back
    ang = np.arctan2(front[1] - back[1], front[0] - back[0]) + a
    dx, dy = int(d * np.rint(np.cos(ang))), int(d * np.rint(np.sin(ang)))
    if d > 0:
        nf, nb = (front[0] + dx, front[1] + dy), front
    else:
        nf, nb = back, (back[0] + dx, back[1] + dy)
    if nf in walls or nb in walls:
        return front, back
    return nf, nb


def _plan_one(init, goal, walls, reserved, horizon, start_delay=0):
    """Time-expanded BFS; `reserved[t]` = cells other vehicles use at step t."""

    def blocked(cells, t):
        for tt in (t - 1, t, t + 1):
            if 0 <= tt and cells & reserved[min(tt, len(reserved) - 1)]:
                return True
        return False

    start = (init["front"], init["back"])
    target = (goal["front"], goal["back"])
    queue = deque([(start, 0)])
    parent = {(start, 0): None}
    while queue:
        (f, b), t = queue.popleft()
        if (f, b) == target:
            # must be able to stay parked for the rest of the horizon
            if not any(blocked({f, b}, tt) for tt in range(t, horizon)):
                path, key = [], ((f, b), t)
                while key is not None:
                    path.append(key[0])
                    key = parent[key]
                return [{"front": p[0], "back": p[1]} for p in reversed(path)]
        if t + 1 >= horizon:
            continue
        acts = (0,) if t < start_delay else range(len(ACTIONS))
        for act in acts:
            nf, nb = move(f, b, act, walls)
            if act != 0 and (nf, nb) == (f, b):
                continue
            key = ((nf, nb), t + 1)
            if key in parent or blocked({nf, nb}, t + 1):
                continue
            parent[key] = ((f, b), t)
            queue.append(key)
    raise RuntimeError("no conflict-free strategy found within the horizon")


def generate_strategy(n_vehicles=4, order=None, start_delays=None, horizon=48):
    """Dict[agent -> list of {"front","back"}] in the recorded-strategy format.

    order        planning priority (default 0..n-1); different orders give different strategies
    start_delays per-agent number of initial waiting steps
    """
    if order is None:
        # vehicles whose start blocks another's goal slot go first; fall back to any feasible priority
        from itertools import permutations

        if start_delays is None:
            start_delays = DEFAULT_DELAYS[:n_vehicles]
        first = [i for i in DEFAULT_ORDER if i < n_vehicles]
        for cand in [first] + [list(p) for p in permutations(range(n_vehicles))]:
            try:
                return generate_strategy(n_vehicles, cand, start_delays, horizon)
            except RuntimeError:
                continue
        raise RuntimeError("no feasible planning priority")
    walls = wall_cells()
    order = list(order)
    start_delays = [0] * n_vehicles if start_delays is None else list(start_delays)
    reserved = [set() for _ in range(horizon + 2)]
    # vehicles not yet planned still block their start cells at t=0
    hist = {}
    for idx in order:
        cfg = AGENT_CONFIGS[idx]
        others0 = set()
        for j in range(n_vehicles):
            if j != idx and j not in [o for o in order[: order.index(idx)]]:
                others0 |= {AGENT_CONFIGS[j]["init_state"]["front"], AGENT_CONFIGS[j]["init_state"]["back"]}
        res = [set(r) for r in reserved]
        for t in range(0, 3):
            res[t] |= others0
        path = _plan_one(cfg["init_state"], cfg["goal"], walls, res, horizon, start_delays[idx])
        for t in range(horizon + 2):
            st = path[min(t, len(path) - 1)]
            reserved[t] |= {st["front"], st["back"]}
        hist["vehicle_%d" % idx] = path
    return {k: hist[k] for k in sorted(hist)}


def write_strategy(file_name, hist):
    """Writes `<file_name>.pkl` (what `compute_sets.load_strategy` / the reference read)."""
    with open(file_name + ".pkl", "wb") as f:
        pickle.dump(hist, f)
    return file_name + ".pkl"
