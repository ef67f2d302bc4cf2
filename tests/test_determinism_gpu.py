"""Determinism of the kernels, repeated: identical inputs in one launch and across launches give identical bits, 20 times over
(VERDICT r4 item 6).  Round 4 retired a one-wavefront collocation kernel that returned two or three different results for identical plans
of one batch after a recompile that left its source untouched (docs/notebook.md); its wavefront-local fence + wave-barrier idiom for
hand-offs between the lanes of ONE wavefront through global memory is still used by the single-vehicle separator recursion
(`CFZS_WFENCE`, cfz_struct.inl) -- within the rules of the AMDGPU memory model (workgroup-scope release = s_waitcnt vmcnt(0); the
wavefronts of a workgroup share their CU's vector L1 outside tgsplit mode), but a rule is not a measurement.  The joint plan's recursion
(cfz_jstruct.inl, round 5) hands nothing over between lanes through memory: what a lane reads back it wrote itself."""
import os
import tempfile

import numpy as np
import pytest

from conflict_rez_amd import scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody

pytestmark = pytest.mark.gpu
TAU = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
REPS = 20


@pytest.fixture(scope="module")
def lot():
    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        sets, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    agents = sorted(hist)
    return dict(agents=agents, paths=paths, fh={a: float(paths[a][-1, 2]) for a in agents},
                tubes={a: [((s["back"].A, s["back"].b), (s["front"].A, s["front"].b)) for s in sets[a][1:]] for a in agents})


def _guess(ws, n_sets):
    N = 5 * (n_sets - 1)
    t = 0.1 * np.arange(len(ws))
    ti = (np.arange(N)[:, None] + TAU[None, :]).ravel() / N * t[-1]
    return np.stack([np.interp(ti, t, ws[:, c]) for c in range(7)], 1), t[-1] / N


@pytest.mark.parametrize("structured", [1, 0])
def test_520_identical_single_plans_twenty_times(lot, structured):
    """520 copies of vehicle 1's collocation plan in one launch (more than two plans per CU), 20 launches: every plan of every launch equals
    the lone plan bit for bit -- the structured elimination (cfz_jstruct.inl's single-vehicle scheme `jstruct_solve1`: the interiors on the
    matrix cores, the separator recursion from both ends with every hand-off in registers; the only structured path since round 6 removed
    round 4's, whose recursion handed blocks over through global memory behind wavefront fences) and the band elimination a panel at a time
    (eight wavefronts, barriers between panels)."""
    from conflict_rez_amd import engine

    a, B = "vehicle_1", 520
    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    tube, p, fh = lot["tubes"][a], lot["paths"][a], lot["fh"][a]
    ws = engine.state_ws([p[0]], [tube], [p], [fh], shrink_tube=0.5)[0]
    g, dt0 = _guess(ws["traj"], len(tube) + 1)
    one = engine.colloc(sp0, [p[0]], [tube], [g], [dt0], [fh], max_iter=400, structured=structured)[0]
    assert one["status"] == 0
    for rep in range(REPS):
        many = engine.colloc(sp0, [p[0]] * B, [tube] * B, [g] * B, [dt0] * B, [fh] * B, max_iter=400, structured=structured)
        bad = [i for i, r in enumerate(many) if not (r["iters"] == one["iters"] and r["dt"] == one["dt"] and np.array_equal(r["traj"], one["traj"]))]
        assert not bad, (rep, len(bad), bad[:8])


def test_identical_joint_plans_repeated(lot):
    """64 copies of the four-vehicle joint plan in one launch, five launches, and the lone plan: bit for bit the same plan (cfz_jstruct.inl:
    per-vehicle interiors, capacitance systems, Schur complements on the matrix cores with every entry owned by one lane, the two-sided
    recursion in registers)."""
    from conflict_rez_amd import engine

    agents = lot["agents"]
    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    init = [lot["paths"][a][0] for a in agents]
    ws = engine.state_ws(init, [lot["tubes"][a] for a in agents], [lot["paths"][a] for a in agents], [lot["fh"][a] for a in agents], shrink_tube=0.5)
    gs = [_guess(w["traj"], len(lot["tubes"][a]) + 1) for w, a in zip(ws, agents)]
    sing = engine.colloc(sp0, init, [lot["tubes"][a] for a in agents], [g[0] for g in gs], [g[1] for g in gs], [lot["fh"][a] for a in agents], max_iter=400)
    assert all(r["status"] == 0 for r in sing)
    scen = dict(init_poses=init, tubes=[lot["tubes"][a] for a in agents], guesses=[r["traj"].reshape(-1, 7) for r in sing],
                dt0=float(np.mean([r["dt"] for r in sing])), final_headings=[lot["fh"][a] for a in agents])
    one = engine.joint_colloc_batch(sp0, [scen], max_iter=300)[0]
    assert one["status"] == 0
    for rep in range(5):
        many = engine.joint_colloc_batch(sp0, [scen] * 64, max_iter=300)
        bad = [i for i, r in enumerate(many) if not (r["iters"] == one["iters"] and r["dt"] == one["dt"] and all(np.array_equal(x, y) for x, y in zip(r["traj"], one["traj"])))]
        assert not bad, (rep, len(bad), bad[:8])


def test_mpc_batch_twenty_times_and_alone():
    """1024 cold MPC-step instances of the planned table (256 scenarios x 4 vehicles) in one launch, 20 launches: every launch returns the
    first one's bits; and instances solved alone return what they return inside the batch (two wavefronts per instance, four instances per
    CU: no instance sees its neighbours)."""
    from conflict_rez_amd import engine

    spec = scenarios.parking_lot_spec()
    table, _ = scenarios.load_reference_table(kind="planned")
    k0, noise = scenarios.sample_scenarios(256, table, seed=7, spec=spec)
    x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
    eng = engine.Engine(spec, max_batch=len(x0))
    first = eng.solve(x0, ref, nbr, zu, want_duals=False)
    assert (first["status"] == 0).mean() > 0.9
    for rep in range(REPS):
        r = eng.solve(x0, ref, nbr, zu, want_duals=False)
        assert np.array_equal(r["status"], first["status"]) and np.array_equal(r["iters"], first["iters"]) and np.array_equal(r["zu"], first["zu"]), rep
    for b in (0, 1, 517, 1023, int(np.argmax(first["iters"]))):
        for rep in range(4):
            r1 = eng.solve(x0[b : b + 1], ref[b : b + 1], nbr[b : b + 1], zu[b : b + 1], want_duals=False)
            assert r1["status"][0] == first["status"][b] and r1["iters"][0] == first["iters"][b] and np.array_equal(r1["zu"][0], first["zu"][b]), (b, rep)
    eng.close()


def test_bench_workload_keeps_the_bodies_apart_where_both_solves_converge():
    """The separating-axis invariant at bench scale (until now only the nominal four-vehicle run had it): 256 scenarios
    of the bench's sampler x 25 stepwise MPC iterations on the planned table, after every iteration the body polygons of the driven
    states.  The Jacobi iteration lets bodies overlap (each vehicle plans against the others' LAST predictions) -- measured 227 of 38,400
    pairs -- but never where both vehicles' solves of that iteration converged: there the polygons are at least dmin - constr_viol_tol
    apart (measured 0.041 m).  The overlaps are what status 4 reports at the next iteration (the reference's IPOPT fails there too)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec_ = importlib.util.spec_from_file_location("closed_loop_separation", os.path.join(root, "tools", "closed_loop_separation.py"))
    cls = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(cls)
    rows = cls.run(256, 25, 2024)
    sep = np.concatenate([r[3] for r in rows])
    both = np.concatenate([r[4] for r in rows])
    assert len(sep) == 256 * 25 * 6 and both.mean() > 0.85
    assert sep[both].min() > 0.05 - 1.5e-2, float(sep[both].min())
    assert 0 < (sep < 0).sum() < 0.02 * len(sep) and not ((sep < 0) & both).any()
    # the polygon test itself: two unit squares a known distance apart / overlapping
    sq = np.array([[[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]]])
    assert abs(cls.separation(sq, sq + np.array([1.5, 0.0]))[0] - 0.5) < 1e-12 and abs(cls.separation(sq, sq + np.array([0.75, 0.25]))[0] + 0.25) < 1e-12
