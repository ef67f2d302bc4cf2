"""state_ws (reference vehicle.py:99-231): the planning solver source against the numpy oracle (CPU), and the HIP
build through the C ABI and the Python surface (GPU)."""
import os
import tempfile

import numpy as np
import pytest

from conflict_rez_amd import strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
from oracle import ipm
from oracle.plan_nlp import StateWsNlp, speed_guess

PLAN_OPT = dict(max_iter=500, hessian="exact", reg_dual=1e-9, stall_iters=0, mu_init=0.1)  # cfz_default_plan_options: IPOPT's mu_init


@pytest.fixture(scope="module")
def plans():
    """The synthetic 4-vehicle strategy: per agent (tube for the oracle, spline guess [T+1,3])."""
    hist = strat.generate_strategy(4)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "4v_rl_traj")
        strat.write_strategy(fn, hist)
        tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
    out = {}
    for a in sorted(hist):
        out[a] = ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]], paths[a])
    return out


def test_plan_kernel_source_matches_oracle(plans):
    """The kernel source (Newton system as a Riccati sweep over the stages) against the full-KKT sparse-LU oracle: same iteration counts,
    solutions to 1e-9, with bounded inputs and with a terminal heading when the guess carries the speed along the path (what
    `cfz_state_ws` does); from the guess that stands still (v = 0: no control authority over the headings in the first linearisations,
    an ill-conditioned start) the two linear solvers' rounding shows in the iterates, 8e-7 on the longest plan, at equal iteration
    counts.  With a terminal heading AND the standing guess the first linearisation is rank deficient (the headings cannot move, the
    terminal row contradicts them; delta_c sits on that row alone, oracle/plan_nlp.py): both give up with status 2 after 2-20
    iterations, which is what is compared -- `cfz_state_ws` seeds the speed, the documented way out (include/confrez_hip.h)."""
    import plan_emu_binding as pe

    opt = ipm.IpmOptions(**PLAN_OPT)
    for a, (tube, p) in plans.items():
        for fh, bounded, exact, seeded in ((None, False, True, False), (None, True, True, True), (float(p[-1, 2]), False, True, True),
                                           (float(p[-1, 2]), False, False, False)):
            nlp = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5, bounded_input=bounded)
            X0 = nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt) if seeded else None)
            ro, re_ = ipm.solve(nlp, X0, opt), pe.solve(nlp, X0, opt)
            assert ro["status"] == re_["status"], (a, fh, bounded)
            if exact:
                assert ro["status"] == 0 and ro["iters"] == re_["iters"]
                assert np.abs(ro["X"][: nlp.s0] - re_["X"][: nlp.s0]).max() < (1e-9 if seeded else 2e-6), (a, fh, bounded)
            else:
                assert ro["status"] == 2  # (after 2-45 iterations, not the same count: the two linear solvers part ways on the singular system)
            if ro["status"] == 0:  # the tube is respected
                s = nlp.unpack(re_["X"])
                c = nlp.cons(re_["X"])
                assert np.abs(c).max() < 2e-2 and re_["X"][nlp.s0 :].min() >= 0.0
                assert np.isclose(s["x"][0], p[0, 0]) and np.isclose(s["v"][0], 0.0, atol=1e-9)


def test_homogeneous_sweep_of_the_matrix_cores_in_plain_loops(plans, tmp_path):
    """The recursion `riccati_backward_mfma` runs on the GPU (cfz_plan.inl: homogeneous coordinates [z, 1, e], five matrix products per
    stage, the value function used as its own transpose and symmetrised at every stage) as plain loops in the CPU build
    (-DCFZP_DENSE_SWEEP -DCFZP_DENSE_TRANSPOSED), against the hand-written one-lane sweep of the default build: the four plans of the
    strategy take the same iterations to the same cost.  (Without the symmetrisation the transposed use lets an antisymmetric part of P
    grow over vehicle 0's 300 stages: 220 iterations, status 2 -- docs/notebook.md.)"""
    import ctypes
    import subprocess

    import plan_emu_binding as pe

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libplan_dense.so")
    subprocess.check_call(["g++", "-O2", "-DCFZP_DENSE_SWEEP", "-DCFZP_DENSE_TRANSPOSED", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", lib,
                           os.path.join(root, "tests", "emu", "cfz_plan_emu.cpp")])
    opt = ipm.IpmOptions(**PLAN_OPT)
    want = {}
    for a, (tube, p) in plans.items():
        nlp = StateWsNlp(p[0], tube, final_heading=float(p[-1, 2]), shrink_tube=0.5)
        want[a] = (nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)))
        want[a] += (pe.solve(want[a][0], want[a][1], opt),)
    default_lib = pe._lib
    try:
        pe._lib = ctypes.CDLL(lib)
        for a, (nlp, X0, r0) in want.items():
            r = pe.solve(nlp, X0, opt)
            assert (r["status"], r["iters"]) == (r0["status"], r0["iters"]) == (0, r0["iters"]) and abs(r["f"] - r0["f"]) < 1e-9 * r0["f"], a
    finally:
        pe._lib = default_lib


def test_default_guess_through_the_tube(plans):
    """`cfz_state_ws_default_guess` (host arithmetic of the library, no GPU): what `cfz_state_ws` starts from when the caller has no guess
    -- `spline_ws = False`, which the reference's own scripts configure for vehicle_0 (vehicle.py:894-899, vehicle_follower.py:871-876;
    IPOPT then starts from zeros).  The path goes from the initial pose through the centres of the back cells, heading towards the
    front cell's centre, and ends on the terminal heading; from it the kernel source converges on every vehicle of the strategy to
    the optimum it reaches from the reference's spline guess (cost to 1e-3 at the solver's tolerance of 1e-2), where the standing start (every stage at the initial
    pose: rank deficient under a terminal heading) ends with status 2 on all four."""
    import plan_emu_binding as pe
    from conflict_rez_amd import engine

    opt = ipm.IpmOptions(**PLAN_OPT)
    for a, (tube, p) in plans.items():
        fh = float(p[-1, 2])
        ctube = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in tube[1:]]
        g = engine.state_ws_default_guess(p[0], ctube, fh)
        nlp = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
        assert g.shape == (nlp.T + 1, 3) and np.allclose(g[0], p[0]) and abs(g[-1, 2] - fh) < 1e-12
        for i in range(1, nlp.S):  # the checkpoints sit inside their (unshrunk) back cells, the headings are continuous
            A, b = tube[i]["back"]
            assert (A @ g[30 * i, :2] <= b + 1e-9).all()
        assert np.abs(np.diff(g[:, 2])).max() < 0.2 and np.hypot(*np.diff(g[:, :2], axis=0).T).max() < 0.2
        r = pe.solve(nlp, nlp.pack(g[:, 0], g[:, 1], g[:, 2], v=speed_guess(g, nlp.dt)), opt)
        rs = pe.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)), opt)
        assert r["status"] == rs["status"] == 0 and r["iters"] <= 30 and abs(r["f"] - rs["f"]) < 1e-3 * rs["f"], a  # (both stop at tol = 1e-2)
        standing = pe.solve(nlp, nlp.pack(np.full(nlp.T + 1, p[0, 0]), np.full(nlp.T + 1, p[0, 1]), np.full(nlp.T + 1, p[0, 2])), opt)
        assert standing["status"] == 2
    lib = engine.load_library()  # a plan needs two strategy steps
    assert lib.cfz_state_ws_default_guess(1, 30, None, 0.0, None, None) != 0 and b"bad argument" in lib.cfz_last_error()


@pytest.mark.gpu
def test_state_ws_without_a_guess(plans, tmp_path):
    """`cfz_state_ws` with guess = NULL: all four vehicles converge from the default guess, to the plans they reach from the spline
    guess (cost to 1e-3, poses to 2 cm: both stop at tol = 1e-2); and `Vehicle.state_ws(spline_ws=False)` of the host mirror -- the reference's configuration for vehicle_0 --
    returns that plan instead of raising."""
    from conflict_rez_amd import engine
    from conflict_rez_amd.control.vehicle import Vehicle

    agents = sorted(plans)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    fhs = [float(plans[a][1][-1, 2]) for a in agents]
    init = [plans[a][1][0] for a in agents]
    none = engine.state_ws(init, tubes, None, fhs, shrink_tube=0.5)
    spl = engine.state_ws(init, tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5)
    for a, r, s_ in zip(agents, none, spl):
        assert r["status"] == s_["status"] == 0 and r["iters"] <= 30 and abs(r["cost"] - s_["cost"]) < 1e-3 * s_["cost"], a
        assert np.abs(r["traj"][:, :3] - s_["traj"][:, :3]).max() < 2e-2
    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, strat.generate_strategy(4))
    v = Vehicle(rl_file_name=fn, agent="vehicle_0", color={"front": (1, 0, 0), "back": (0, 1, 0)})
    z = v.state_ws(N=30, dt=0.1, final_heading=fhs[0], shrink_tube=0.5, spline_ws=False)
    assert v.state_ws_stats["status"] == 0 and abs(v.state_ws_stats["cost"] - spl[0]["cost"]) < 1e-3 * spl[0]["cost"] and len(z.x) == len(spl[0]["traj"])


def _corridor(S, h=1.25, x0=5.0, y0=17.5):
    """A straight corridor of S strategy steps (2.5 m cells along +x): tube for the oracle, tube for the C ABI, straight-line guess."""
    A = np.array([[0.0, -1.0], [-1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    cell = lambda i: (A, np.array([-(y0 - h), -(x0 + 2.5 * i - h), x0 + 2.5 * i + h, y0 + h]))
    tube = [dict(back=cell(i), front=cell(i + 1)) for i in range(S)]
    T = 30 * (S - 1)
    p = np.stack([np.linspace(x0, x0 + 2.5 * (S - 1), T + 1), np.full(T + 1, y0), np.zeros(T + 1)], 1)
    return tube, [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in tube[1:]], p


LONG_BOUNDS = [0.0, 80.0, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0]


def test_a_plan_of_510_stages(plans):
    """T = 30 x 17 = 510 Euler stages (the strategy's longest has 300): the Riccati sweep of the kernel source against the sparse-LU
    oracle on a plan too long for the LDS of a CU (41 doubles per stage: T <= 498) -- same iteration count, trajectory to 1e-9."""
    import plan_emu_binding as pe

    tube, _, p = _corridor(18)
    nlp = StateWsNlp(p[0], tube, final_heading=0.0, shrink_tube=0.5, bounds=LONG_BOUNDS)
    assert nlp.T == 510
    X0 = nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt))
    opt = ipm.IpmOptions(**PLAN_OPT)
    ro, re_ = ipm.solve(nlp, X0, opt), pe.solve(nlp, X0, opt)
    assert (ro["status"], ro["iters"]) == (re_["status"], re_["iters"]) == (0, ro["iters"]) and ro["iters"] < 30
    assert np.abs(ro["X"][: nlp.s0] - re_["X"][: nlp.s0]).max() < 1e-9


@pytest.mark.gpu
def test_scattered_starts_on_gpu_against_the_cpu_build(plans):
    """32 plans from start poses scattered by +-3 cm (the bench's configs[1] sample, eight per vehicle) in one launch against the CPU
    build of the same source, plan by plan: equal status and iteration count, trajectories to 1e-6 -- iteration counts 9 ... 28, not the
    four nominal plans again."""
    import plan_emu_binding as pe
    from conflict_rez_amd import engine

    agents = sorted(plans)
    rng = np.random.default_rng(0)
    who = [agents[i % 4] for i in range(32)]
    init = [plans[a][1][0] + np.r_[rng.uniform(-0.03, 0.03, 2), 0.0] for a in who]
    tubes = {a: [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents}
    fh = {a: float(plans[a][1][-1, 2]) for a in agents}
    res = engine.state_ws(init, [tubes[a] for a in who], [plans[a][1] for a in who], [fh[a] for a in who], shrink_tube=0.5)
    opt = ipm.IpmOptions(**PLAN_OPT)
    its = []
    for i, (a, r) in enumerate(zip(who, res)):
        tube, p = plans[a]
        nlp = StateWsNlp(init[i], tube, final_heading=fh[a], shrink_tube=0.5)
        re_ = pe.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)), opt)
        se = nlp.unpack(re_["X"])
        want = np.stack([se["x"], se["y"], se["psi"], se["v"], se["delta"], se["a"], se["w"]], 1)
        assert (r["status"], r["iters"]) == (re_["status"], re_["iters"]) == (0, re_["iters"]), (i, a)
        assert np.abs(r["traj"] - want).max() < 1e-6, (i, a)
        its.append(r["iters"])
    assert min(its) <= 10 and max(its) >= 20


@pytest.mark.gpu
def test_a_plan_too_long_for_the_lds_runs_from_the_workspace():
    """The same 510-stage plan through `cfz_state_ws`: asked for the LDS kernel (`kernel = CFZ_KERNEL_WIDE`) it runs from the workspace
    (the sweep's per-stage data would take 167 KB), bit for bit what `CFZ_KERNEL_NARROW` returns, and equals the oracle; in one batch
    with a short plan (the batch's longest plan decides) the short plan is the one it is alone."""
    from conflict_rez_amd import engine

    tube, ctube, p = _corridor(18)
    tube6, ctube6, p6 = _corridor(6)
    kw = dict(shrink_tube=0.5, bounds=LONG_BOUNDS)
    wide = engine.state_ws([p[0]], [ctube], [p], [0.0], kernel=engine.KERNEL_WIDE, **kw)[0]
    narrow = engine.state_ws([p[0]], [ctube], [p], [0.0], kernel=engine.KERNEL_NARROW, **kw)[0]
    assert wide["status"] == 0 and np.array_equal(wide["traj"], narrow["traj"]) and wide["iters"] == narrow["iters"]
    nlp = StateWsNlp(p[0], tube, final_heading=0.0, shrink_tube=0.5, bounds=LONG_BOUNDS)
    ro = ipm.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)), ipm.IpmOptions(**PLAN_OPT))
    so = nlp.unpack(ro["X"])
    want = np.stack([so["x"], so["y"], so["psi"], so["v"], so["delta"], so["a"], so["w"]], 1)
    assert wide["iters"] == ro["iters"] and np.abs(wide["traj"] - want).max() < 1e-6
    short = engine.state_ws([p6[0]], [ctube6], [p6], [0.0], **kw)[0]  # alone: LDS
    both = engine.state_ws([p[0], p6[0]], [ctube, ctube6], [p, p6], [0.0, 0.0], **kw)
    assert np.array_equal(both[1]["traj"], short["traj"]) and np.array_equal(both[0]["traj"], wide["traj"])


@pytest.mark.gpu
@pytest.mark.parametrize("narrow", [False, True])
def test_state_ws_on_gpu_matches_oracle(plans, narrow):
    """cfz_state_ws, all four vehicles in one launch, against the oracle: status, iteration count, trajectory -- with the sweep's
    per-stage data in LDS (what a batch this small gets by default) and in the workspace (what batches of more than one plan per CU
    and plans too long for the LDS get; here asked for through `cfz_plan_options.kernel`)."""
    from conflict_rez_amd import engine

    kern = dict(kernel=engine.KERNEL_NARROW) if narrow else {}

    agents = sorted(plans)
    tubes = [[((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]] for a in agents]
    opt = ipm.IpmOptions(**PLAN_OPT)
    for with_heading in (False, True):  # the reference's callers fix the terminal heading (vehicle.py:901-912)
        fhs = [float(plans[a][1][-1, 2]) if with_heading else None for a in agents]
        res = engine.state_ws([plans[a][1][0] for a in agents], tubes, [plans[a][1] for a in agents], fhs, shrink_tube=0.5, **kern)
        for a, fh, r in zip(agents, fhs, res):
            tube, p = plans[a]
            nlp = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
            ro = ipm.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)), opt)
            so = nlp.unpack(ro["X"])
            assert (r["status"], r["iters"]) == (ro["status"], ro["iters"]) == (0, ro["iters"]), (a, fh)
            want = np.stack([so["x"], so["y"], so["psi"], so["v"], so["delta"], so["a"], so["w"]], 1)
            assert np.abs(r["traj"] - want).max() < 1e-6 and abs(r["cost"] - ro["f"]) < 1e-7
            if fh is not None:
                assert abs(r["traj"][-1, 2] - fh) < 1e-6


@pytest.mark.gpu
def test_a_plan_does_not_depend_on_its_batch_when_the_kernel_is_pinned(plans):
    """A plan's iterates do not depend on the batch it is solved in.  `cfz_plan_options.kernel` moves state_ws' sweep data between LDS
    (default up to one plan per CU) and the workspace (larger batches): same arithmetic, same bits -- the same plan alone and inside a
    batch of more than two plans per CU returns the same status, iteration count and trajectory, pinned or not.  The collocation plan
    has one kernel since round 4 (`cfz_colloc_options.kernel` other than 0 is an error since round 5): alone or in a batch of 520, bit for bit
    the same plan -- the check that caught the one-wavefront kernel returning two results for identical plans (docs/notebook.md)."""
    from conflict_rez_amd import engine

    cus = 256  # MI355X; only "more than two plans per CU" matters
    a = "vehicle_1"
    tube = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in plans[a][0][1:]]
    fh = float(plans[a][1][-1, 2])
    B = 2 * cus + 8
    args1 = ([plans[a][1][0]], [tube], [plans[a][1]], [fh])
    argsB = ([plans[a][1][0]] * B, [tube] * B, [plans[a][1]] * B, [fh] * B)
    for kern in (engine.KERNEL_WIDE, engine.KERNEL_NARROW):
        one = engine.state_ws(*args1, shrink_tube=0.5, kernel=kern)[0]
        many = engine.state_ws(*argsB, shrink_tube=0.5, kernel=kern)
        for r in (many[0], many[B // 2], many[-1]):
            assert (r["status"], r["iters"]) == (one["status"], one["iters"]) and np.array_equal(r["traj"], one["traj"])
    auto1, autoB = engine.state_ws(*args1, shrink_tube=0.5)[0], engine.state_ws(*argsB, shrink_tube=0.5)[0]
    assert np.array_equal(auto1["traj"], engine.state_ws(*args1, shrink_tube=0.5, kernel=engine.KERNEL_WIDE)[0]["traj"])
    assert np.array_equal(autoB["traj"], engine.state_ws(*args1, shrink_tube=0.5, kernel=engine.KERNEL_NARROW)[0]["traj"])
    assert np.array_equal(autoB["traj"], auto1["traj"])  # state_ws: LDS or workspace, the same bits
    with pytest.raises(RuntimeError, match="kernel"):
        engine.state_ws(*args1, shrink_tube=0.5, kernel=7)
    # the same for the collocation refinement
    from conflict_rez_amd import scenarios

    sp0 = scenarios.parking_lot_spec(n_nbr=0, N=2)
    ws1 = engine.state_ws(*args1, shrink_tube=0.5, kernel=engine.KERNEL_WIDE)[0]["traj"]
    N = 5 * len(tube)
    t = 0.1 * np.arange(len(ws1))
    tau = np.array([0.0, 0.05710419611451768, 0.2768430136381238, 0.5835904323689168, 0.8602401356562195, 1.0])
    ti = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * t[-1]
    guess = np.stack([np.interp(ti, t, ws1[:, c]) for c in range(7)], 1)
    cargs = lambda B_: (sp0, [plans[a][1][0]] * B_, [tube] * B_, [guess] * B_, [t[-1] / N] * B_, [fh] * B_)
    one = engine.colloc(*cargs(1), max_iter=400)[0]
    many = engine.colloc(*cargs(B), max_iter=400)
    assert one["status"] == 0
    for r in many:  # every plan of the batch, not a sample: the retired kernel differed on about half of them
        assert (r["status"], r["iters"]) == (one["status"], one["iters"]) and np.array_equal(r["traj"], one["traj"]) and r["dt"] == one["dt"]
    with pytest.raises(RuntimeError, match="retired"):  # the field that named the retired kernel is an error now, not a silent no-op
        engine.colloc(*cargs(1), max_iter=400, kernel=engine.KERNEL_NARROW)
    # cfz_plan_ws_trim: the memory the batch left in the thread's workspace goes back; the next call allocates again
    engine.trim_default_workspaces()
    assert np.array_equal(engine.state_ws(*args1, shrink_tube=0.5)[0]["traj"], auto1["traj"])
    ws = engine.PlanWorkspace()
    r1 = engine.state_ws(*args1, shrink_tube=0.5, ws=ws)[0]
    ws.trim()
    assert np.array_equal(engine.state_ws(*args1, shrink_tube=0.5, ws=ws)[0]["traj"], r1["traj"])
    ws.close()


@pytest.mark.gpu
def test_plan_single_path_then_follow_on_gpu(tmp_path):
    """The reference's flow from a strategy file alone, with the configuration of its `main` (vehicle_0 without spline guess):
    MultiDistributedFollower plans every vehicle (state_ws, dual_ws and the collocation refinement on the GPU) and runs the
    distributed MPC on the result."""
    from conflict_rez_amd.control.vehicle_follower import MultiDistributedFollower

    fn = str(tmp_path / "4v_rl_traj")
    strat.write_strategy(fn, strat.generate_strategy(4))
    agents = ["vehicle_%d" % i for i in range(4)]
    colors = {a: {"front": (1.0, 0.0, 0.0), "back": (0.0, 0.0, 1.0)} for a in agents}
    paths = interp_along_sets(fn, VehicleBody(), 30)
    # the reference's own configuration (vehicle_follower.py:871-890): no spline guess for vehicle_0, terminal headings 0, 3 pi / 2, pi, pi / 2
    ws_config = {"vehicle_0": False, "vehicle_1": True, "vehicle_2": True, "vehicle_3": True}
    heads = {"vehicle_0": 0.0, "vehicle_1": 3 * np.pi / 2, "vehicle_2": np.pi, "vehicle_3": np.pi / 2}
    assert all(abs(np.angle(np.exp(1j * (heads[a] - paths[a][-1, 2])))) < 1e-9 for a in agents)
    mdf = MultiDistributedFollower(fn, ws_config, colors, {a: None for a in agents}, heads)
    mdf.setup_multi_vehicles()
    for v in mdf.vehicles:
        assert v.plan_refined is True and v.state_ws_stats["status"] == 0 and v.final_problem_stats["status"] == 0
        assert abs(np.angle(np.exp(1j * (v.reference_traj.psi[-1] - v.final_heading)))) < 1e-2  # the terminal heading of the reference's callers
        # the collocation plan frees dt: it is faster than the warm start's fixed 0.1 s x 30 steps per strategy step
        assert 2.0 < v.reference_traj.t[-1] < 0.1 * 30 * (v.num_sets - 1) and v.K == 5
        assert v.reference_traj.x.shape == v.reference_traj.psi.shape and np.isfinite(v.reference_xy).all()
    mdf.solve(num_iter=5, dump=False)
    assert sum(v.status == 0 for v in mdf.vehicles) >= 3
    for v in mdf.vehicles:
        ref = v.interpolate_states([v.state.t])
        assert np.hypot(v.state.x.x - ref.x[0], v.state.x.y - ref.y[0]) < 0.5


@pytest.mark.gpu
def test_planned_reference_table_reproduces_the_package_data():
    """`scenarios.planned_reference_table` (the build's own single-vehicle plans: `cfz_state_ws` -> `cfz_colloc` through
    `VehicleFollower.plan_single_path`, sampled every 0.1 s) against the table bench.py follows by default,
    `conflict_rez_amd/data/refs_4v_planned.npz` (written by this function on an MI355X in round 3): vehicles 1-3 to the sample -- same
    lengths, poses within 1 mm, inputs within 1e-2.  Vehicle 0's collocation NLP has neighbouring local solutions: from the round-4 warm
    start (state_ws at IPOPT's mu_init = 0.1 instead of 1e-3) the refinement ends on one that takes 15.9 s instead of 15.7 s (160
    samples instead of 158); the package table is kept so that the MPC workload stays the one of rounds 3-4, and vehicle 0 is compared
    for what matters to it: a converged plan of the same route, within 0.3 s and 0.5 m of the stored one at equal fractions of the way.
    The plans are the fast ones (7.8-15.9 s against the 18-30 s of the state_ws warm starts)."""
    from conflict_rez_amd import scenarios

    table, lengths, info = scenarios.planned_reference_table()
    ref, ref_len = scenarios.load_reference_table(kind="planned")
    assert lengths[1:].tolist() == ref_len[1:].tolist() and abs(int(lengths[0]) - int(ref_len[0])) <= 3
    T = min(table.shape[1], ref.shape[1])
    assert np.abs(table[1:, :T, :3] - ref[1:, :T, :3]).max() < 1e-3 and np.abs(table[1:, :T, 3:] - ref[1:, :T, 3:]).max() < 1e-2
    s_new, s_old = np.linspace(0, lengths[0] - 1, 50).round().astype(int), np.linspace(0, ref_len[0] - 1, 50).round().astype(int)
    assert np.hypot(*(table[0, s_new, :2] - ref[0, s_old, :2]).T).max() < 0.5 and np.abs(table[0, lengths[0] - 1, :3] - ref[0, ref_len[0] - 1, :3]).max() < 1e-2
    assert 7.0 < min(i["t_end"] for i in info.values()) and max(i["t_end"] for i in info.values()) < 16.5
    lengths = ref_len
    ws, ws_len = scenarios.load_reference_table(kind="state_ws")
    assert (ws_len > 1.7 * lengths).all()  # the warm-start plans take about twice as long
    # every start the bench draws is feasible for the first NLP of every vehicle
    spec = scenarios.parking_lot_spec()
    k0, noise = scenarios.sample_scenarios(256, ref, seed=5, spec=spec)
    assert (scenarios.start_clearances(spec, ref, k0, noise) >= spec.dmin - 0.01).all()


def test_mirror_symmetry_of_state_ws(plans):
    """The reflection y -> 35 - y of tube, start pose, terminal heading and path guess mirrors the warm start `state_ws` computes
    (planning source, CPU build): equal status and iteration count, cost to 1e-9, trajectory to 1e-6."""
    import plan_emu_binding as pe

    def mir_poly(A, b):
        A = np.asarray(A, float); b = np.asarray(b, float)
        return A * np.array([1.0, -1.0]), b - 35.0 * A[..., 1]

    opt = ipm.IpmOptions(**PLAN_OPT)
    for a in ("vehicle_1", "vehicle_3"):
        tube, p = plans[a]
        fh = float(p[-1, 2])
        nlp = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
        r0 = pe.solve(nlp, nlp.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, nlp.dt)), opt)
        tube_m = [dict(front=mir_poly(*s["front"]), back=mir_poly(*s["back"])) for s in tube]
        pm = np.stack([p[:, 0], 35.0 - p[:, 1], -p[:, 2]], 1)
        nlm = StateWsNlp(pm[0], tube_m, final_heading=-fh, shrink_tube=0.5)
        r1 = pe.solve(nlm, nlm.pack(pm[:, 0], pm[:, 1], pm[:, 2], v=speed_guess(pm, nlm.dt)), opt)
        assert (r0["status"], r0["iters"]) == (r1["status"], r1["iters"]) == (0, r0["iters"]), (a, r0["status"], r0["iters"], r1["status"], r1["iters"])
        s0, s1 = nlp.unpack(r0["X"]), nlm.unpack(r1["X"])
        assert abs(r0["f"] - r1["f"]) < 1e-9 * max(1.0, abs(r0["f"]))
        assert np.abs(s0["x"] - s1["x"]).max() < 1e-6 and np.abs(s0["y"] - (35.0 - s1["y"])).max() < 1e-6 and np.abs(s0["psi"] + s1["psi"]).max() < 1e-6
        assert np.abs(s0["delta"] + s1["delta"]).max() < 1e-6 and np.abs(s0["v"] - s1["v"]).max() < 1e-6
