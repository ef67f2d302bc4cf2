"""Synthetic strategy files obey the reference environment's rules (rl/pklot_env.py:131-158,226-356) and feed the
geometry helpers the way the reference's `.pkl` does (compute_sets.py:27-256)."""
import pickle

import numpy as np

from conflict_rez_amd import strategy as strat
from conflict_rez_amd.control.bezier import BezierPlanner
from conflict_rez_amd.control.compute_sets import (compute_initial_states, compute_sets, convert_rl_states,
                                                   interp_along_sets)
from conflict_rez_amd.control.utils import pi_2_pi
from conflict_rez_amd.pytypes import VehicleState
from conflict_rez_amd.vehicle_types import VehicleBody


def test_strategy_is_legal_and_conflict_free():
    hist, walls = strat.generate_strategy(4), strat.wall_cells()
    assert sorted(hist) == ["vehicle_%d" % i for i in range(4)]
    for i, (a, steps) in enumerate(sorted(hist.items())):
        cfg = strat.AGENT_CONFIGS[i]
        assert steps[0] == cfg["init_state"] and steps[-1] == cfg["goal"]
        for cur, nxt in zip(steps[:-1], steps[1:]):
            assert nxt["front"] not in walls and nxt["back"] not in walls
            legal = [strat.move(cur["front"], cur["back"], act, walls) for act in range(7)]
            assert (nxt["front"], nxt["back"]) in legal
    T = max(len(s) for s in hist.values())
    for t in range(T):
        cells = []
        for s in hist.values():
            st = s[min(t, len(s) - 1)]
            cells += [st["front"], st["back"]]
        assert len(set(cells)) == len(cells)


def test_pkl_format_and_geometry(tmp_path):
    hist = strat.generate_strategy(4)
    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, hist)
    with open(fn + ".pkl", "rb") as f:
        assert pickle.load(f) == hist
    vb = VehicleBody()
    tubes, inits, paths = compute_sets(fn), compute_initial_states(fn, vb), interp_along_sets(fn, vb, 30)
    for a, steps in hist.items():
        assert len(tubes[a]) == len(steps)
        for st, sets in zip(steps, tubes[a]):
            for part in ("front", "back"):
                centre = (np.array(st[part]) + 0.5) * 2.5
                assert sets[part].contains(centre) and not sets[part].contains(centre + [2.6, 0])
        p = paths[a]
        assert p.shape == (30 * (len(steps) - 1) + 1, 3)
        assert np.allclose(p[0, :2], [inits[a].x.x, inits[a].x.y])
        assert np.isclose(pi_2_pi(p[0, 2] - inits[a].e.psi), 0.0, atol=1e-12)
        assert np.abs(np.diff(p[:, 2])).max() < 0.5  # unwrapped heading
        last = convert_rl_states(steps[-1], vb)
        assert np.allclose(p[-1, :2], [last.x.x, last.x.y])


def test_initial_state_known_answer():
    vb = VehicleBody()
    s = convert_rl_states({"front": (6, 8), "back": (6, 7)}, vb)  # heading +y, centre (16.25, 20)
    assert np.isclose(s.e.psi, np.pi / 2) and np.allclose([s.x.x, s.x.y], [16.25, 20 - 1.25])
    s = convert_rl_states({"front": (5, 6), "back": (4, 6)}, vb)  # heading +x, centre (12.5, 16.25)
    assert np.isclose(s.e.psi, 0.0) and np.allclose([s.x.x, s.x.y], [12.5 - 1.25, 16.25])
    s = convert_rl_states({"front": (7, 7), "back": (6, 6)}, vb)  # diagonal
    assert np.isclose(s.e.psi, np.pi / 4) and np.allclose([s.x.x, s.x.y], 17.5 - 1.25 * np.sqrt(0.5))


def test_bezier_endpoints_and_tangents():
    a, b = VehicleState(), VehicleState()
    a.x.x, a.x.y, a.e.psi = 1.0, 2.0, 0.0
    b.x.x, b.x.y, b.e.psi = 4.0, 5.0, np.pi / 2
    path = BezierPlanner(offset=2.5).interpolate(a, b, 30)
    assert path.shape == (30, 3) and np.allclose(path[0], [1.0, 2.0, 0.0])
    assert np.hypot(*(path[-1, :2] - [4.0, 5.0])) < 0.5  # end point itself excluded
    assert np.all(np.diff(path[:, 2]) > -1e-9) and path[-1, 2] < np.pi / 2 + 1e-9
