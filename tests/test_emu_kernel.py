"""The kernel source itself (conflict_rez_amd/csrc/cfz_solver.inl), compiled for the CPU with the 64 lanes
run as a loop (tests/emu/cfz_emu.cpp), against the oracle.  Checks the kernel's logic without a GPU; the GPU
build of the same source is checked by tests/test_gpu_parity.py."""
import ctypes as C
import subprocess
import sys

import numpy as np

import emu_binding as emu
from oracle import ipm, port
from oracle.mpc_nlp import reference_residuals


def _sol(r):
    z = r["zu"]
    return dict(x=z[0], y=z[1], psi=z[2], v=z[3], delta=z[4], a=z[5], w=z[6], l=r["l"], m=r["m"],
                lam_ij=r["lam_ij"], lam_ji=r["lam_ji"], s=r["s"])


def test_kernel_source_matches_oracle_on_goldens(golden, ospec):
    opt = ipm.IpmOptions()
    meta = golden["meta"]
    for b in range(len(golden["x0"])):
        r = emu.solve(ospec, opt, golden["x0"][b], golden["ref"][b], golden["nbr"][b], golden["zu"][b])
        assert r["status"] == int(meta[b, 0]) and r["iters"] == int(meta[b, 1]), b
        if r["status"] == 0:
            assert np.abs(r["zu"] - golden["sol"][b]).max() < 1e-7
            assert np.isclose(r["cost"], meta[b, 2], rtol=1e-9) and np.isclose(r["min_sep"], meta[b, 3], atol=1e-8)
            res = reference_residuals(ospec, golden["x0"][b], golden["ref"][b], golden["nbr"][b], _sol(r))
            assert res["eq"] < 1e-2 and res["ineq"] < 1e-2 and res["bound"] == 0.0


def test_carried_multipliers_port_and_kernel_source_match_oracle(ospec):
    """Three consecutive MPC iterations of three vehicles, each solve started from the multipliers of the one before
    (tests/golden/carry_golden.npz from the full-KKT oracle): the C port and the kernel source reproduce status,
    iteration count and solution (tests/golden/make_carry_inputs.py: a vehicle working against active bounds and a neighbour, and
    one that merely tracks its reference -- for the latter carrying takes a third of the cold iterations) and land in the same
    local solution to within the solver tolerance."""
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ci, cg = np.load(os.path.join(here, "carry_inputs.npz")), np.load(os.path.join(here, "carry_golden.npz"))
    opt = ipm.IpmOptions()
    for seq in range(len(ci["x0"]) // 3):
        ce, cp = None, None
        for t in range(3):
            i = 3 * seq + t
            st, it, cold_it = int(cg["meta"][i, 0]), int(cg["meta"][i, 1]), int(cg["meta"][i, 4])
            re_ = emu.solve(ospec, opt, ci["x0"][i], ci["ref"][i], ci["nbr"][i], ci["zu"][i], want_duals=False, carry=ce)
            rp = port.solve(ospec, ci["x0"][i], ci["ref"][i], ci["nbr"][i], ci["zu"][i].T.copy(), opt, carry=cp)
            ce, cp = re_["carry"], rp["carry"]
            assert (re_["status"], re_["iters"]) == (st, it) == (rp["status"], rp["iters"]), (seq, t)
            assert np.abs(re_["zu"] - cg["sol"][i]).max() < 1e-6 and np.abs(rp["p"].T - cg["sol"][i]).max() < 1e-6
            if t > 0 and seq == 1:  # the vehicle that merely tracks: a third of the cold iterations; the one working against
                assert it <= cold_it // 2  # active bounds and a neighbour needs about as many as from cold multipliers
            if seq == 2:  # a cornered vehicle: the first solve waits for the late curvature shift (50 iterations), its successors
                assert (it > 40) if t == 0 else (it < cold_it)  # start shifted (IpmOptions.carry_shift): 13, 18 against 20, 24 from cold


def test_kernel_source_other_shapes(ospec):
    """n_nbr = 0 / n_obs = 4 (BASELINE.json config 2) and a short horizon: same answers as the C port."""
    from conflict_rez_amd import scenarios
    from oracle.mpc_nlp import MpcSpec

    table, _ = scenarios.load_reference_table()
    opt = ipm.IpmOptions()
    for n_obs, n_nbr, N in ((4, 0, 30), (6, 1, 12), (0, 2, 8)):
        sp = scenarios.parking_lot_spec(n_nbr=n_nbr, N=N, n_obs=n_obs)
        osp = MpcSpec(N=N, dt=sp.dt, A_obs=sp.A_obs, b_obs=sp.b_obs, n_nbr=n_nbr)
        k0, noise = scenarios.sample_scenarios(3, table, seed=4)
        x0, ref, nbr, zu = scenarios.mpc_batch_from_table(sp, table[: n_nbr + 1], k0, noise[:, : n_nbr + 1])
        for b in range(len(x0)):
            r1 = port.solve(osp, x0[b], ref[b], nbr[b], zu[b].T, opt)
            r2 = emu.solve(osp, opt, x0[b], ref[b], nbr[b], zu[b])
            assert (r1["status"], r1["iters"]) == (r2["status"], r2["iters"])
            if r1["status"] == 0:
                assert np.abs(r1["p"].T - r2["zu"]).max() < 1e-7


def test_restoration_in_other_shapes():
    """The restoration phase outside the 4-vehicle shape: a single vehicle (four obstacles, no neighbour: BASELINE.json config 2) and a
    two-vehicle problem on a short horizon, reference and warm start pushed 0.6-0.9 m sideways into a box's clearance (restoration
    first) -- equal status and iteration count in the C port and the kernel source, and the phase does its work: the converged
    plans keep the clearance that the starts violated."""
    from conflict_rez_amd import scenarios
    from oracle.mpc_nlp import MpcSpec

    table, _ = scenarios.load_reference_table()
    opt, off = ipm.IpmOptions(), ipm.IpmOptions(restoration=0)
    seen = 0
    for n_obs, n_nbr, N, push in ((4, 0, 30, 0.9), (6, 1, 12, 0.6)):
        sp = scenarios.parking_lot_spec(n_nbr=n_nbr, N=N, n_obs=n_obs)
        osp = MpcSpec(N=N, dt=sp.dt, A_obs=sp.A_obs, b_obs=sp.b_obs, n_nbr=n_nbr)
        k0, noise = scenarios.sample_scenarios(6, table, seed=11)
        x0, ref, nbr, zu = scenarios.mpc_batch_from_table(sp, table[: n_nbr + 1], k0, noise[:, : n_nbr + 1])
        for b in range(0, len(x0), n_nbr + 1):
            # sideways (to the left of the heading) by `push` from stage 3 on: the start state itself stays where it is
            shift = np.zeros((2, N)); shift[:, 3:] = push * np.array([[-np.sin(ref[b][2, 0])], [np.cos(ref[b][2, 0])]])
            rf, z0 = ref[b].copy(), zu[b].copy()
            rf[:2] += shift; z0[:2] += shift
            r1 = port.solve(osp, x0[b], rf, nbr[b], z0.T, opt)
            r2 = emu.solve(osp, opt, x0[b], rf, nbr[b], z0)
            assert (r1["status"], r1["iters"]) == (r2["status"], r2["iters"]), (n_obs, b)
            if r1["status"] == 0:
                assert np.abs(r1["p"].T - r2["zu"]).max() < 1e-7 and r1["sep"].min() > osp.dmin - 1e-2
            r0 = port.solve(osp, x0[b], rf, nbr[b], z0.T, off)
            seen += (r1["status"] == 0 and r1["iters"] > r0["iters"] + 5) or (r1["status"] == 0 and r0["status"] != 0)
    assert seen >= 1  # some start went through the phase (more iterations than without it, or a solve that fails without it)


def test_kernel_source_under_sanitizers(golden, ospec, tmp_path):
    """AddressSanitizer + UBSan on the CPU build of the kernel source (no GPU sanitizers on this pool)."""
    lib = emu.build(sanitize=True)
    code = f"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {str(emu.ROOT)!r}); sys.path.insert(0, {str(emu.ROOT + '/tests')!r})
import emu_binding as emu
from oracle import ipm
from oracle.mpc_nlp import MpcSpec
d = np.load({str(emu.ROOT + '/tests/golden/mpc_golden.npz')!r})
emu._lib = C.CDLL({lib!r})
sp = MpcSpec(A_obs=d['A_obs'], b_obs=d['b_obs'], n_nbr=3)
for b in (0, 7, 11):
    r = emu.solve(sp, ipm.IpmOptions(), d['x0'][b], d['ref'][b], d['nbr'][b], d['zu'][b])
    assert r['status'] == int(d['meta'][b, 0])
print('sanitized ok')
"""
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(**__import__("os").environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "sanitized ok" in out.stdout, out.stderr[-2000:]


def test_dual_regularisation_and_late_shift_end_the_cycle_of_the_scaled_curvature(ospec):
    """tests/golden/mpc_late_shift.npz (a vehicle pressed into a corner of the lot): with neither the dual regularisation of the
    separation rows (IpmOptions.reg_dual_rows, IPOPT's delta_c) nor the late curvature shift the scaled row curvature cycles to
    max_iter; round 3's answer, the late shift alone (from iteration 60, or 40 once the error stagnates), ends them in 51-80
    iterations; with delta_c (round 4, default) they end in 40-74 -- on three of the six the late shift is not even reached.  The C
    port and the kernel source reproduce the full-KKT oracle's iteration counts and solutions at the defaults, and (first
    instance) the full-KKT oracle itself regenerates the stored vector."""
    import os

    from oracle.mpc_nlp import solve_mpc

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_late_shift.npz"))
    opt = ipm.IpmOptions()
    cyc, nodc = ipm.IpmOptions(shift_after=0, err_stall_iters=0, reg_dual_rows=0.0), ipm.IpmOptions(reg_dual_rows=0.0)
    assert opt.shift_after == 60 and opt.reg_dual_rows == 1e-8
    for b in range(len(g["x0"])):
        args = (g["x0"][b], g["ref"][b], g["nbr"][b], g["zu"][b])
        re_ = emu.solve(ospec, opt, *args, want_duals=False)
        rp = port.solve(ospec, args[0], args[1], args[2], args[3].T.copy(), opt)
        assert (re_["status"], re_["iters"]) == (0, int(g["iters_shift"][b])) == (rp["status"], rp["iters"]), b
        assert 35 < re_["iters"] < 100
        assert np.abs(re_["zu"] - g["sol"][b]).max() < 1e-6 and np.abs(rp["p"].T - g["sol"][b]).max() < 1e-6
        rn = emu.solve(ospec, nodc, *args, want_duals=False)
        assert (rn["status"], rn["iters"]) == (0, int(g["iters_nodc"][b])) and rn["iters"] >= re_["iters"] - 2
        if b < 2:
            r0 = emu.solve(ospec, cyc, *args, want_duals=False)
            assert r0["iters"] == int(g["iters_noshift"][b]) == 600 and r0["status"] == 1
    full = solve_mpc(ospec, *[g[k][0] for k in ("x0", "ref", "nbr", "zu")])
    assert (full["status"], full["iters"]) == (0, int(g["iters_shift"][0])) and np.abs(full["zu"] - g["sol"][0]).max() < 1e-9


def test_error_stall_ends_a_cycle_below_the_violation_tolerance(ospec):
    """`err_stall_iters` (150): with the late shift and the dual regularisation off the solves of tests/golden/mpc_late_shift.npz
    cycle at a violation below constr_viol_tol, where the violation-based stall test never fires; they end with status 5 once the
    scaled optimality error has not halved for 150 iterations instead of running to the iteration limit -- in the port and in the
    kernel source alike."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_late_shift.npz"))
    opt = ipm.IpmOptions(shift_after=0, reg_dual_rows=0.0)
    args = (g["x0"][0], g["ref"][0], g["nbr"][0], g["zu"][0])
    re_ = emu.solve(ospec, opt, *args, want_duals=False)
    rp = port.solve(ospec, args[0], args[1], args[2], args[3].T.copy(), opt)
    assert (re_["status"], re_["iters"]) == (rp["status"], rp["iters"]) and re_["status"] == 5 and 150 <= re_["iters"] < 400


def test_restoration_phase_agrees_across_the_three_implementations():
    """The restoration phase (IpmOptions.restoration / resto_first; oracle/mpc_nlp.py MpcNlp.restore): the full-KKT oracle, the C port
    and the kernel source take the same Levenberg-Marquardt steps -- equal status and iteration counts, solutions to 1e-7 -- on
    instances of tests/golden/mpc_independent_turn.npz: 6 (start 0.5 m inside a clearance: restoration first, then 20 interior-point
    iterations to the independent optimum), 13 (start 0.13 m inside: no restoration) and 3 (untouched); and the port and the kernel
    source on instance 0 (150 iterations, 40 of them in the restoration) and, with `restoration = 0`, on the failure that round 3
    asserted there (status 2)."""
    import os

    from oracle.mpc_nlp import MpcSpec, solve_mpc

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_independent_turn.npz"))
    sp = MpcSpec(N=30, dt=0.1, A_obs=d["A_obs"], b_obs=d["b_obs"], n_nbr=3)
    opt = ipm.IpmOptions()
    for b, full in ((6, True), (13, True), (3, True), (0, False)):
        args = (d["x0"][b], d["ref"][b], d["nbr"][b], d["zu"][b])
        re_ = emu.solve(sp, opt, *args, want_duals=False)
        rp = port.solve(sp, args[0], args[1], args[2], args[3].T.copy(), opt)
        assert (re_["status"], re_["iters"]) == (rp["status"], rp["iters"]) and re_["status"] == 0, b
        assert np.abs(re_["zu"] - rp["p"].T).max() < 1e-7
        if full:
            rn = solve_mpc(sp, *args, opt)
            assert (rn["status"], rn["iters"]) == (rp["status"], rp["iters"]) and np.abs(rn["zu"] - rp["p"].T).max() < 1e-7, b
    off = ipm.IpmOptions(restoration=0, reg_dual_rows=0.0)
    args = (d["x0"][0], d["ref"][0], d["nbr"][0], d["zu"][0])
    re_ = emu.solve(sp, off, *args, want_duals=False)
    rp = port.solve(sp, args[0], args[1], args[2], args[3].T.copy(), off)
    assert (re_["status"], re_["iters"]) == (rp["status"], rp["iters"]) and re_["status"] == 2 and 50 < re_["iters"] < 70


def test_mirror_symmetry_on_the_cpu(golden, ospec):
    """The reflection y -> 35 - y of obstacles, state, reference, neighbours and warm start (psi, delta, w change sign) mirrors the
    solution: C port and kernel source on goldens with face contacts, a vertex-vertex contact and none (the GPU twin runs the full
    batch, tests/test_gpu_parity.py::test_mirror_symmetry)."""
    from oracle.mpc_nlp import MpcSpec

    A, b = np.asarray(ospec.A_obs), np.asarray(ospec.b_obs)
    mspec = MpcSpec(N=ospec.N, dt=ospec.dt, A_obs=A * np.array([1.0, -1.0]), b_obs=b - 35.0 * A[:, :, 1], n_nbr=ospec.n_nbr)

    def mir(a, axis):
        a = np.array(a, float)
        idx = [slice(None)] * a.ndim
        idx[axis] = 1; a[tuple(idx)] = 35.0 - a[tuple(idx)]
        idx[axis] = 2; a[tuple(idx)] = -a[tuple(idx)]
        return a

    opt = ipm.IpmOptions()
    for i in (0, 7, 18, 19):
        x0, ref, nbr, zu = golden["x0"][i], golden["ref"][i], golden["nbr"][i], golden["zu"][i]
        x0m = mir(x0, 0); x0m[4] = -x0m[4]
        zum = mir(zu, 0); zum[4] = -zum[4]; zum[6] = -zum[6]
        for solve in (lambda sp, *a: (lambda r: (r["status"], r["iters"], r["p"].T))(port.solve(sp, a[0], a[1], a[2], a[3].T.copy(), opt)),
                      lambda sp, *a: (lambda r: (r["status"], r["iters"], r["zu"]))(emu.solve(sp, opt, *a, want_duals=False))):
            s0, it0, z0 = solve(ospec, x0, ref, nbr, zu)
            s1, it1, z1 = solve(mspec, x0m, mir(ref, 0), mir(nbr, 1), zum)
            back = mir(z1, 0); back[4] = -back[4]; back[6] = -back[6]
            assert (s0, it0) == (s1, it1) and np.abs(back - z0).max() < 1e-7, (i, s0, it0, s1, it1)


def test_eight_lanes_per_stage_build_of_the_kernel_source():
    """`-DCFZ_LPS=8` (round 6's go / no-go: 256-lane instances, four wavefronts, the stage sum a `row_half_mirror` longer, the reductions over
    four wavefronts) is a diagnostic build, not the product -- but the switch must not rot: the CPU build of the same source with eight
    lanes per stage solves the goldens with the oracle's status and iteration counts (other lane partials, the same algorithm).  Run in a
    child process: the binding caches one library per process."""
    import os

    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import emu_binding as emu\n"
            "from oracle import ipm\n"
            "from oracle.mpc_nlp import MpcSpec\n"
            "from conflict_rez_amd import scenarios\n"
            "spec = scenarios.parking_lot_spec()\n"
            "ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)\n"
            "g = np.load(%r)\n"
            "assert emu.build().endswith('_lps8.so')\n"
            "for b in (0, 5, 9, 17, 19):\n"
            "    r = emu.solve(ospec, ipm.IpmOptions(), g['x0'][b], g['ref'][b], g['nbr'][b], g['zu'][b], want_duals=False)\n"
            "    assert (r['status'], r['iters']) == (int(g['meta'][b, 0]), int(g['meta'][b, 1])), (b, r['status'], r['iters'])\n"
            "    assert r['status'] != 0 or np.abs(r['zu'] - g['sol'][b]).max() < 1e-7\n"
            "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)),
                                os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_golden.npz"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CFZ_EMU_LPS="8"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-1500:]


def test_horizons_at_the_edges_of_the_lane_map():
    """Horizons where round 6's layout tricks switch over, CPU build of the kernel source against the C port (status, iteration count,
    solution): N = 32 leaves no spare stage slot and N = 31 exactly one stage's worth of lanes for the stage-0 feasibility check (it runs
    in the lanes beyond the horizon when there are at least as many as a stage has, and as a pass of its own otherwise); N = 4 is too short
    for the value function of stage 0 to stand in the step's slots of stages 1-5 (it shares the cos / sin slots then, and the step phase
    forms the headings' sine and cosine again)."""
    from conflict_rez_amd import scenarios
    from oracle.mpc_nlp import MpcSpec

    table, _ = scenarios.load_reference_table()
    seen = set()
    for N, n_nbr, seed in ((32, 3, 4), (31, 3, 5), (4, 1, 6)):
        sp = scenarios.parking_lot_spec(n_nbr=n_nbr, N=N)
        osp = MpcSpec(N=N, dt=sp.dt, A_obs=sp.A_obs, b_obs=sp.b_obs, n_nbr=n_nbr)
        k0, noise = scenarios.sample_scenarios(2, table, seed=seed)  # raw draws: an infeasible measured state may be among them
        x0, ref, nbr, zu = scenarios.mpc_batch_from_table(sp, table[: n_nbr + 1], k0, noise[:, : n_nbr + 1])
        for b in range(min(len(x0), 4)):
            r = emu.solve(osp, ipm.IpmOptions(), x0[b], ref[b], nbr[b], zu[b], want_duals=False)
            q = port.solve(osp, x0[b], ref[b], nbr[b], zu[b].T.copy())
            assert (r["status"], r["iters"]) == (q["status"], q["iters"]), (N, b, r["status"], r["iters"], q["status"], q["iters"])
            seen.add(r["status"])
            if r["status"] == 0:
                assert np.abs(r["zu"][:5] - q["p"].T[:5]).max() < 1e-6, (N, b)
    assert 0 in seen
