"""The benchmark's scenario sampler (conflict_rez_amd/scenarios.py): both package reference tables load, the polygon distance behind
the start-feasibility check equals the oracle's, and drawing with a spec only yields starts whose first NLP is feasible."""
import numpy as np

from conflict_rez_amd import scenarios


def test_reference_tables():
    ws, ws_len = scenarios.load_reference_table(kind="state_ws")
    pl, pl_len = scenarios.load_reference_table(kind="planned")
    assert ws.shape[0] == pl.shape[0] == 4 and ws.shape[2] == pl.shape[2] == 7 and (pl_len < ws_len).all()
    for t, n in ((ws, ws_len), (pl, pl_len)):
        for v in range(4):
            if n[v] < t.shape[1]:  # goal pose held, at rest
                assert np.abs(t[v, n[v]:, :3] - t[v, n[v] - 1, :3]).max() == 0.0 and np.abs(t[v, n[v]:, 3:]).max() == 0.0
    # default of the loader stays the table the MPC goldens were generated on
    assert np.array_equal(scenarios.load_reference_table()[0], ws)


def test_quad_distance_equals_the_oracle():
    from oracle.independent_mpc import polygon_distance_batch

    rng = np.random.default_rng(0)

    def quad(c, a, s):
        B = np.array([[1, .5], [-1, .5], [-1, -.5], [1, -.5]]) * s
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        return c + B @ R.T

    P = np.stack([quad(rng.uniform(-3, 3, 2), rng.uniform(-3, 3), 1.0) for _ in range(500)])
    Q = np.stack([quad(rng.uniform(-3, 3, 2), rng.uniform(-3, 3), 1.3) for _ in range(500)])
    d1, d2 = scenarios._quad_distance(P, Q), np.maximum(polygon_distance_batch(P[:, None], Q[:, None])[:, 0], 0.0)
    assert np.abs(d1 - d2).max() < 1e-12 and 50 < (d1 == 0.0).sum() < 450


def test_feasible_starts():
    spec = scenarios.parking_lot_spec()
    for kind in ("planned", "state_ws"):
        table, _ = scenarios.load_reference_table(kind=kind)
        k0, nz = scenarios.sample_scenarios(256, table, seed=3)
        raw = scenarios.start_clearances(spec, table, k0, nz)
        assert (raw < spec.dmin - 0.01).any()  # the raw draws contain starts inside a clearance (status 4 at the first solve) ...
        k1, n1 = scenarios.sample_scenarios(256, table, seed=3, spec=spec)
        box = scenarios.start_box_excess(spec, table, k0, nz)  # ... and starts outside the NLP's state boxes (a noisy speed above the limit)
        ok = (raw.min(1) >= spec.dmin - 0.01) & (box.max(1) <= 1e-2)
        assert (kind == "state_ws") or (box.max(1) > 1e-2).any()  # the planned table's vehicle 1 drives at the speed limit
        assert np.array_equal(k1[ok], k0[ok]) and np.array_equal(n1[ok], nz[ok])  # ... which alone are drawn again
        assert (scenarios.start_clearances(spec, table, k1, n1) >= spec.dmin - 0.01).all()
        assert (scenarios.start_box_excess(spec, table, k1, n1) <= 1e-2).all()
