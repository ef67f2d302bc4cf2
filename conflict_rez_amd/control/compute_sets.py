"""Inputs of every NLP on the path: strategy file -> tube sets, initial states, obstacles,
spline initial guess.

Reference: `confrez/control/compute_sets.py` -- `compute_sets` :27-139, `convert_rl_states`
:142-164, `interp_along_sets` :167-240, `compute_initial_states` :243-256, `compute_obstacles`
:259-330.  The strategy file `<name>.pkl` is a pickle of
Dict[agent -> List[{"front": (ix,iy), "back": (ix,iy)}]] (grid cells of 2.5 m; reference
`rl/record_states_history.py:10-31`).
"""
import pickle
from typing import Dict, List

import numpy as np

from ..obstacle_types import Polytope
from ..pytypes import VehicleState
from ..vehicle_types import VehicleBody
from .bezier import BezierPlanner
from .utils import pi_2_pi


def load_strategy(file_name: str):
    with open(file_name + ".pkl", "rb") as f:
        return pickle.load(f)


def compute_sets(file_name: str, L=2.5) -> Dict[str, List[Dict[str, Polytope]]]:
    """Per agent and strategy step: the L x L cell squares of the front and back body halves."""
    cell = Polytope([[0, 0], [0, L], [L, 0], [L, L]])
    hist = load_strategy(file_name)
    return {
        agent: [{part: cell + np.array(st[part]) * L for part in ("front", "back")} for st in steps]
        for agent, steps in hist.items()
    }


def convert_rl_states(states, vehicle_body: VehicleBody, L: float = 2.5) -> VehicleState:
    """Grid cells of (front, back) -> pose of the rear-axle reference point."""
    fx, fy = states["front"]
    bx, by = states["back"]
    dx, dy = fx - bx, fy - by
    psi = np.arctan2(dy, dx)
    if dy == 0:
        cx, cy = max(fx, bx) * L, (fy + 0.5) * L
    elif dx == 0:
        cx, cy = (fx + 0.5) * L, max(fy, by) * L
    else:
        cx, cy = max(fx, bx) * L, max(fy, by) * L
    out = VehicleState()
    out.e.psi = psi
    out.x.x = cx - vehicle_body.wb / 2 * np.cos(psi)
    out.x.y = cy - vehicle_body.wb / 2 * np.sin(psi)
    return out


def compute_initial_states(file_name: str, vehicle_body: VehicleBody, L=2.5) -> Dict[str, VehicleState]:
    return {a: convert_rl_states(s[0], vehicle_body, L) for a, s in load_strategy(file_name).items()}


def interp_along_sets(file_name: str, vehicle_body: VehicleBody, N: int):
    """Pose guess [N*(S-1)+1, 3] per agent: hold / straight line / cubic Bezier per strategy
    transition (N samples each, end excluded), final pose appended, heading unwrapped."""
    planner = BezierPlanner(offset=2.5)
    out = {}
    for agent, steps in load_strategy(file_name).items():
        segs = []
        for cur, nxt in zip(steps[:-1], steps[1:]):
            a, b = convert_rl_states(cur, vehicle_body), convert_rl_states(nxt, vehicle_body)
            seg = np.tile([a.x.x, a.x.y, a.e.psi], (N, 1)).astype(float)
            if nxt == cur:
                pass  # waiting
            elif a.e.psi == b.e.psi:
                seg[:, 0] = np.linspace(a.x.x, b.x.x, N, endpoint=False)
                seg[:, 1] = np.linspace(a.x.y, b.x.y, N, endpoint=False)
            else:
                flip = np.pi if nxt["front"] == cur["back"] else 0.0  # turning while reversing
                a.e.psi, b.e.psi = pi_2_pi(a.e.psi + flip), pi_2_pi(b.e.psi + flip)
                seg = planner.interpolate(a, b, N)
                seg[:, 2] -= flip
            segs.append(seg)
        last = convert_rl_states(steps[-1], vehicle_body)
        segs.append(np.array([[last.x.x, last.x.y, last.e.psi]]))
        path = np.vstack(segs)
        path[:, 2] = np.unwrap(path[:, 2])
        out[agent] = path
    return out


# [xmin, xmax, ymin, ymax] in cell units (L) plus +-w/2 in x; reference compute_sets.py:259-330
_OBSTACLE_CELLS = ((1.5, 5.5, 3, 5.5), (7.5, 7.5, 3, 5.5), (9.5, 12.5, 3, 5.5),
                   (1.5, 5.5, 8.5, 11), (7.5, 8.5, 8.5, 11), (10.5, 12.5, 8.5, 11))


def compute_obstacles(L: float = 2.5, vb: VehicleBody = None) -> List[Polytope]:
    """The six static boxes of the parking lot (rows of parked cars)."""
    hw = (vb or VehicleBody()).w / 2
    return [Polytope.from_box(x0 * L - hw, x1 * L + hw, y0 * L, y1 * L) for x0, x1, y0, y1 in _OBSTACLE_CELLS]


# parked cars drawn by the visualiser: (slot index, randomly set back?) per lane; reference compute_sets.py:349-433
_PARKED_LOWER = ((1, True), (2, True), (3, True), (4, True), (5, False), (7, False), (9, True), (10, True), (11, True), (12, True))
_PARKED_UPPER = ((1, True), (2, True), (3, True), (4, True), (5, False), (7, False), (8, False), (10, True), (11, True), (12, True))


def compute_static_vehicles(L: float = 2.5, vb: VehicleBody = None) -> List[Polytope]:
    """The parked cars the reference's visualiser draws into the slots (cosmetics: no NLP sees them).  What matters to the
    path: the reference builds them at import of `vehicle_follower` right after `np.random.seed(0)` and so consumes
    FIFTEEN `np.random.sample()` draws (one per randomly set-back car, lower lane first, in slot order) before any
    vehicle draws its first dual guesses (vehicle_follower.py:29-31, :401-402).  Same draws, same order."""
    vb = vb or VehicleBody()
    out = []
    for lane, edge, sign in ((_PARKED_LOWER, 5.5 * L, -1.0), (_PARKED_UPPER, 8.5 * L, 1.0)):
        for slot, randomised in lane:
            near = edge + sign * (np.random.sample() * 0.7 * L if randomised else 0.0)  # the car's end at the lane side
            far = near + sign * vb.l
            xc = (slot + 0.5) * L
            out.append(Polytope.from_box(xc - vb.w / 2, xc + vb.w / 2, min(near, far), max(near, far)))
    return out
