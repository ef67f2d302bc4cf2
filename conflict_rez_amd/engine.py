"""ctypes binding of libconfrez_hip.so (C ABI: include/confrez_hip.h).

This is the only place the Python package touches the solver.  There is no CPU fallback:
if the HIP library is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libconfrez_hip.so")
MAX_OBS, MAX_NBR, MAX_N = 8, 7, 32

STATUS_NAMES = {0: "converged", 1: "iteration limit", 2: "line search failed", 3: "non-finite iterate",
                4: "measured state in collision (infeasible)", 5: "constraint violation stalled (locally infeasible)"}


class _CSpec(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("n_obs", C.c_int32), ("n_nbr", C.c_int32), ("rk_substeps", C.c_int32),
        ("dt", C.c_double), ("wb", C.c_double), ("dmin", C.c_double),
        ("g", C.c_double * 4), ("bounds", C.c_double * 12), ("weights", C.c_double * 6),
        ("A_obs", C.c_double * (MAX_OBS * 8)), ("b_obs", C.c_double * (MAX_OBS * 4)),
    ]


_OPT_INTS = ("max_iter", "max_backtrack", "filter_cap", "stall_iters", "row_curvature", "carry_duals", "vv_rows", "shift_after", "restoration", "shift_stagnation", "err_stall_iters", "carry_shift")
_OPT_DBLS = ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
             "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal stall_kappa warm_push reg_dual_rows resto_first").split()


KERNEL_AUTO, KERNEL_WIDE, KERNEL_NARROW = 0, 1, 2  # cfz_plan_options.kernel / cfz_colloc_options.kernel (include/confrez_hip.h)


class _CPlanOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("N", "max_iter", "bounded_input", "stall_iters", "kernel", "reserved0")] + [
        ("dt", C.c_double), ("wb", C.c_double), ("shrink_tube", C.c_double), ("bounds", C.c_double * 12),
        ("tol", C.c_double), ("constr_viol_tol", C.c_double), ("mu_init", C.c_double), ("curv_kappa", C.c_double)]


class _COptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in _OPT_INTS] + [(k, C.c_double) for k in _OPT_DBLS]


@dataclass
class ProblemSpec:
    """Constants of the MPC-step NLP (`cfz_spec`); defaults are the reference's
    (`VehicleFollower.setup_controller`, `VehicleBody`, `VehicleConfig`, `GeofenceRegion`)."""

    N: int = 30
    dt: float = 0.1
    n_nbr: int = 3
    A_obs: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 2)))
    b_obs: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    g: np.ndarray = field(default_factory=lambda: np.array([3.3, 0.9, 0.6, 0.9]))
    wb: float = 2.5
    bounds: np.ndarray = field(
        default_factory=lambda: np.array([2.5, 32.5, 7.5, 27.5, -2.5, 2.5, -0.85, 0.85, -1.5, 1.5, -1.0, 1.0])
    )
    dmin: float = 0.05
    weights: np.ndarray = field(default_factory=lambda: np.array([100.0, 100, 100, 1, 1, 1]))
    rk_substeps: int = 4

    @property
    def n_obs(self):
        return int(np.asarray(self.A_obs).shape[0])

    @classmethod
    def from_objects(cls, obstacles, vehicle_body, vehicle_config, region, n_nbr, N=30, dt=0.1, dmin=0.05):
        """From the reference-style objects (`Polytope` list, `VehicleBody`, `VehicleConfig`, `GeofenceRegion`)."""
        vc, r = vehicle_config, region
        return cls(
            N=N, dt=dt, n_nbr=n_nbr, dmin=dmin, wb=vehicle_body.wb, g=np.asarray(vehicle_body.b, float),
            A_obs=np.stack([o.A for o in obstacles]) if obstacles else np.zeros((0, 4, 2)),
            b_obs=np.stack([o.b for o in obstacles]) if obstacles else np.zeros((0, 4)),
            bounds=np.array([r.x_min, r.x_max, r.y_min, r.y_max, vc.v_min, vc.v_max, vc.delta_min, vc.delta_max,
                             vc.a_min, vc.a_max, vc.w_delta_min, vc.w_delta_max], float),
        )

    def to_c(self):
        if self.n_obs > MAX_OBS or self.n_nbr > MAX_NBR or self.N > MAX_N:
            raise ValueError("problem exceeds the engine's compiled limits")
        s = _CSpec()
        s.N, s.n_obs, s.n_nbr, s.rk_substeps = self.N, self.n_obs, self.n_nbr, self.rk_substeps
        s.dt, s.wb, s.dmin = self.dt, self.wb, self.dmin
        s.g[:] = [float(v) for v in self.g]
        s.bounds[:] = [float(v) for v in self.bounds]
        s.weights[:] = [float(v) for v in self.weights]
        A = np.zeros((MAX_OBS, 4, 2)); b = np.zeros((MAX_OBS, 4))
        if self.n_obs:
            A[: self.n_obs], b[: self.n_obs] = self.A_obs, self.b_obs
        s.A_obs[:] = list(A.ravel()); s.b_obs[:] = list(b.ravel())
        return s


_lib = None


ABI_VERSION = 6  # CFZ_ABI_VERSION of include/confrez_hip.h


def load_library(path=None):
    """Loads libconfrez_hip.so and declares the prototypes of include/confrez_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("CFZ_LIBRARY") or LIB_PATH  # CFZ_LIBRARY: diagnostic builds (tools/)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). conflict_rez_amd has no CPU solver."
        )
    lib = C.CDLL(path)
    vp, i32p = C.c_void_p, C.c_void_p
    # the structs below mirror include/confrez_hip.h at this layout version (a caller compiled against another round's header
    # would hand over shorter structs)
    if not hasattr(lib, "cfz_abi_version") or lib.cfz_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: struct layout version {lib.cfz_abi_version() if hasattr(lib, 'cfz_abi_version') else 'none'}, this binding is written against {ABI_VERSION}: rebuild the library")
    lib.cfz_last_error.restype = C.c_char_p
    lib.cfz_default_spec.argtypes = [C.POINTER(_CSpec)]
    lib.cfz_default_options.argtypes = [C.POINTER(_COptions)]
    lib.cfz_create.argtypes = [C.POINTER(_CSpec), C.POINTER(_COptions), C.c_int, C.c_int, C.POINTER(vp)]
    lib.cfz_destroy.argtypes = [vp]
    lib.cfz_max_batch.argtypes = [vp]
    lib.cfz_kernel_info.argtypes = [vp, vp, vp]
    lib.cfz_mpc_set_params.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.cfz_mpc_set_warm.argtypes = [vp, C.c_int, vp]
    lib.cfz_joint_dual_ws.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.cfz_default_plan_options.argtypes = [C.POINTER(_CPlanOptions)]
    lib.cfz_state_ws.argtypes = [C.c_int, C.c_int, C.POINTER(_CPlanOptions)] + [vp] * 9
    lib.cfz_state_ws_w.argtypes = [vp, C.c_int, C.POINTER(_CPlanOptions)] + [vp] * 9
    lib.cfz_state_ws_default_guess.argtypes = [C.c_int32, C.c_int32, vp, C.c_double, vp, vp]
    lib.cfz_plan_ws_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.cfz_plan_ws_destroy.argtypes = [vp]
    lib.cfz_plan_ws_trim.argtypes = [vp]
    lib.cfz_default_colloc_options.argtypes = [C.POINTER(_CCollocOptions)]
    lib.cfz_colloc.argtypes = [C.c_int, C.c_int, C.POINTER(_CSpec), C.POINTER(_CCollocOptions)] + [vp] * 11
    lib.cfz_colloc_w.argtypes = [vp, C.c_int, C.POINTER(_CSpec), C.POINTER(_CCollocOptions)] + [vp] * 11
    lib.cfz_joint_colloc.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(_CSpec), C.POINTER(_CCollocOptions)] + [vp] * 6 + [C.c_int] + [vp] * 6
    lib.cfz_joint_colloc_w.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_CSpec), C.POINTER(_CCollocOptions)] + [vp] * 6 + [C.c_int] + [vp] * 6
    lib.cfz_mpc_set_carry.argtypes = [vp, C.c_int, vp]
    lib.cfz_mpc_set_carry_device.argtypes = [vp, C.c_int, vp]
    lib.cfz_mpc_set_slots.argtypes = [vp, C.c_int, vp]
    lib.cfz_mpc_solve.argtypes = [vp, C.c_int]
    lib.cfz_mpc_get.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.cfz_mpc_stats.argtypes = [vp, C.c_int, i32p, i32p, vp, vp, vp]
    lib.cfz_last_solve_ms.argtypes = [vp]
    lib.cfz_last_solve_ms.restype = C.c_double
    lib.cfz_mpc_solve_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cfz_dual_ws.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.cfz_loop_init.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    lib.cfz_loop_step.argtypes = [vp]
    lib.cfz_loop_run.argtypes = [vp, C.c_int]
    lib.cfz_loop_last_iterations.argtypes = [vp]
    lib.cfz_loop_last_iterations.restype = C.c_long
    lib.cfz_vsl_step.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cfz_loop_last_converged.argtypes = [vp]
    lib.cfz_loop_last_converged.restype = C.c_long
    lib.cfz_loop_last_status_counts.argtypes = [vp, vp]
    lib.cfz_loop_get.argtypes = [vp, vp, vp, vp, vp]
    _lib = lib
    return lib


EXPORTS = (
    "cfz_default_spec cfz_default_options cfz_create cfz_destroy cfz_max_batch cfz_kernel_info cfz_mpc_set_params cfz_mpc_set_warm "
    "cfz_source_hash cfz_abi_version cfz_colloc_elimination_info cfz_colloc_band_info cfz_joint_dual_ws cfz_default_plan_options cfz_state_ws cfz_state_ws_default_guess cfz_default_colloc_options cfz_colloc cfz_joint_colloc cfz_plan_ws_create cfz_plan_ws_destroy cfz_plan_ws_trim cfz_state_ws_w cfz_colloc_w cfz_joint_colloc_w cfz_mpc_set_carry cfz_mpc_set_carry_device cfz_mpc_set_slots cfz_mpc_solve cfz_mpc_get cfz_mpc_stats cfz_last_solve_ms cfz_mpc_solve_device cfz_dual_ws cfz_loop_init cfz_loop_step cfz_loop_run cfz_loop_last_iterations cfz_loop_last_converged cfz_loop_last_status_counts cfz_vsl_step "
    "cfz_loop_get cfz_last_error"
).split()


def default_options(**overrides):
    lib = load_library()
    o = _COptions()
    lib.cfz_default_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError(f"expected array of shape {tuple(shape)}, got {a.shape}")
    return a


def trim_default_workspaces():
    """`cfz_plan_ws_trim(NULL)`: release the device memory of the calling thread's own workspaces behind `state_ws`, `colloc`,
    `joint_colloc` called without `ws=` (a 256-plan joint launch leaves 25 GB there)."""
    lib = load_library()
    if lib.cfz_plan_ws_trim(None) != 0:
        raise RuntimeError("cfz_plan_ws_trim: " + lib.cfz_last_error().decode())


class PlanWorkspace:
    """`cfz_plan_ws`: a stream and the device buffers of the planning calls, kept between calls.  Pass as `ws=` to
    `state_ws`, `colloc`, `joint_colloc_batch`; without it those use a per-thread workspace inside the library."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self._w = C.c_void_p()
        if self.lib.cfz_plan_ws_create(int(device), C.byref(self._w)) != 0:
            raise RuntimeError("cfz_plan_ws_create: " + self.lib.cfz_last_error().decode())

    def trim(self):
        """`cfz_plan_ws_trim`: give the device memory held between calls back now (the next call allocates again)."""
        if self.lib.cfz_plan_ws_trim(self._w) != 0:
            raise RuntimeError("cfz_plan_ws_trim: " + self.lib.cfz_last_error().decode())

    def close(self):
        if getattr(self, "_w", None):
            self.lib.cfz_plan_ws_destroy(self._w)
            self._w = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def state_ws_default_guess(init_pose, tube, final_heading=None, N=30):
    """`cfz_state_ws_default_guess`: the path through the tube's cells that `cfz_state_ws` starts from when it is given no guess
    (host arithmetic, no GPU).  tube as for `state_ws` (one vehicle); returns [N (n_sets - 1) + 1, 3]."""
    lib = load_library()
    n_sets = len(tube) + 1
    t = np.ascontiguousarray(np.concatenate([np.concatenate([np.asarray(A, float).ravel(), np.asarray(b, float).ravel()]) for cellpair in tube for (A, b) in cellpair]))
    g = np.zeros((N * (n_sets - 1) + 1, 3))
    p0 = _f64(np.asarray(init_pose, float)[:3], (3,))
    rc = lib.cfz_state_ws_default_guess(n_sets, N, _ptr(p0), float("nan") if final_heading is None else float(final_heading), _ptr(t), _ptr(g))
    if rc != 0:
        raise RuntimeError("cfz_state_ws_default_guess: " + lib.cfz_last_error().decode())
    return g


def state_ws(init_poses, tubes, guesses=None, final_headings=None, device=0, ws=None, **options):
    """`cfz_state_ws`: the warm-start plans of several vehicles in one launch.
    init_poses [B][3]; tubes: per vehicle a list over strategy steps 1.. of ((A_back, b_back), (A_front, b_front));
    guesses: per vehicle an array [T+1, 3] of x, y, psi, or None (that vehicle then starts from `state_ws_default_guess`);
    final_headings: per vehicle a float or None.
    options: fields of `cfz_plan_options` (N, dt, wb, shrink_tube, bounded_input, max_iter, tol, ...).
    Returns a list of dict(traj [T+1,7], status, iters, cost)."""
    lib = load_library()
    po = _CPlanOptions()
    lib.cfz_default_plan_options(C.byref(po))
    for k, v in options.items():
        if k == "bounds":
            po.bounds[:] = [float(x) for x in v]
        else:
            setattr(po, k, v)
    B = len(tubes)
    n_sets = np.array([len(t) + 1 for t in tubes], dtype=np.int32)
    T = po.N * (n_sets - 1)
    tube = np.concatenate([np.concatenate([np.concatenate([np.asarray(A, float).ravel(), np.asarray(b, float).ravel()])
                                           for cellpair in t for (A, b) in cellpair]) for t in tubes])
    init = _f64(np.asarray(init_poses, float), (B, 3))
    fh = np.array([np.nan if (final_headings is None or final_headings[b] is None) else float(final_headings[b]) for b in range(B)])
    guess = None
    if guesses is not None and any(g is not None for g in guesses):
        # a mixed batch (the reference's own `spline_ws_config`: vehicle_0 without a spline guess, the others with one): the vehicles
        # without a guess get the one the library would build for them, the others keep theirs (they used to lose it)
        gs = [np.asarray(g, float)[: T[b] + 1, :3] if g is not None else
              state_ws_default_guess(init[b], tubes[b], None if np.isnan(fh[b]) else float(fh[b]), N=po.N) for b, g in enumerate(guesses)]
        guess = np.ascontiguousarray(np.concatenate(gs))
        assert guess.shape[0] == int((T + 1).sum())
    traj = np.zeros((int((T + 1).sum()), 7))
    status, iters, cost = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    args = (B, C.byref(po), _ptr(n_sets), _ptr(init), _ptr(fh), _ptr(np.ascontiguousarray(tube)), _ptr(guess),
            _ptr(traj), _ptr(status), _ptr(iters), _ptr(cost))
    rc = lib.cfz_state_ws(int(device), *args) if ws is None else lib.cfz_state_ws_w(ws._w, *args)
    if rc != 0:
        raise RuntimeError("cfz_state_ws: " + lib.cfz_last_error().decode())
    out, o = [], 0
    for b in range(B):
        out.append(dict(traj=traj[o : o + T[b] + 1].copy(), status=int(status[b]), iters=int(iters[b]), cost=float(cost[b])))
        o += T[b] + 1
    return out


class _CCollocOptions(C.Structure):
    _fields_ = [("N_per_set", C.c_int32), ("max_iter", C.c_int32), ("exact_rows", C.c_int32), ("one_pivot", C.c_int32), ("vv_rows", C.c_int32), ("kernel", C.c_int32),
                ("shrink_tube", C.c_double),
                ("tol", C.c_double), ("constr_viol_tol", C.c_double), ("mu_init", C.c_double), ("curv_kappa", C.c_double),
                ("structured", C.c_int32), ("reserved1", C.c_int32)]


def colloc(spec, init_poses, tubes, guesses, dt0s, final_headings=None, device=0, ws=None, **options):
    """`cfz_colloc`: the collocation plans (vehicle.py:360-661) of several vehicles in one launch.
    spec: ProblemSpec (wb, dmin, body, bounds, static obstacles); init_poses [B][3]; tubes as in `state_ws`;
    guesses: per vehicle an array [6 N, 7] of x, y, psi, v, delta, a, w at the collocation points; dt0s [B];
    options: fields of `cfz_colloc_options` (N_per_set, max_iter, shrink_tube, tol, constr_viol_tol, ...).
    Returns a list of dict(traj [N, 6, 7], dt, status, iters, cost)."""
    lib = load_library()
    co = _CCollocOptions()
    lib.cfz_default_colloc_options(C.byref(co))
    for k, v in options.items():
        if not hasattr(co, k):
            raise TypeError(f"unknown collocation option {k!r}")
        setattr(co, k, v)
    B = len(tubes)
    n_sets = np.array([len(t) + 1 for t in tubes], dtype=np.int32)
    Np = co.N_per_set * (n_sets - 1) * 6
    tube = np.ascontiguousarray(np.concatenate([np.concatenate([np.concatenate([np.asarray(A, float).ravel(), np.asarray(b, float).ravel()])
                                                                for cellpair in t for (A, b) in cellpair]) for t in tubes]))
    init = _f64(np.asarray(init_poses, float), (B, 3))
    fh = np.array([np.nan if (final_headings is None or final_headings[b] is None) else float(final_headings[b]) for b in range(B)])
    guess = np.ascontiguousarray(np.concatenate([_f64(np.asarray(g, float), (int(Np[b]), 7)) for b, g in enumerate(guesses)]))
    dt0 = _f64(np.asarray(dt0s, float), (B,))
    traj, dt = np.zeros((int(Np.sum()), 7)), np.zeros(B)
    status, iters, cost = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    cs = spec.to_c()
    args = (B, C.byref(cs), C.byref(co), _ptr(n_sets), _ptr(init), _ptr(fh), _ptr(tube), _ptr(guess), _ptr(dt0),
            _ptr(traj), _ptr(dt), _ptr(status), _ptr(iters), _ptr(cost))
    rc = lib.cfz_colloc(int(device), *args) if ws is None else lib.cfz_colloc_w(ws._w, *args)
    if rc != 0:
        raise RuntimeError("cfz_colloc: " + lib.cfz_last_error().decode())
    out, o = [], 0
    for b in range(B):
        out.append(dict(traj=traj[o : o + Np[b]].reshape(-1, 6, 7).copy(), dt=float(dt[b]), status=int(status[b]), iters=int(iters[b]), cost=float(cost[b])))
        o += Np[b]
    return out


def source_hash():
    """`cfz_source_hash`: which kernel sources the loaded library was built from (16 hex digits, or "unknown")."""
    lib = load_library()
    lib.cfz_source_hash.restype = C.c_char_p
    return lib.cfz_source_hash().decode()


def colloc_band_info(n_sets, N_per_set=5, n_obs=6, pairs=None, has_final=None):
    """`cfz_colloc_band_info`: (unknowns nk, half-bandwidth kb, bytes of the band) of the banded system of one (joint) collocation
    plan of len(n_sets) vehicles with n_sets[a] strategy steps each (host arithmetic, needs no GPU)."""
    lib = load_library()
    ns = np.ascontiguousarray(np.asarray(n_sets, np.int32))
    hf = None if has_final is None else np.ascontiguousarray(np.asarray(has_final, np.int32))
    pr = None if pairs is None else np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    nk, kb, bb = C.c_int32(), C.c_int32(), C.c_int64()
    lib.cfz_colloc_band_info.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.cfz_colloc_band_info(len(ns), _ptr(ns), _ptr(hf), int(N_per_set), int(n_obs), 0 if pr is None else len(pr), _ptr(pr),
                                  C.addressof(nk), C.addressof(kb), C.addressof(bb))
    if rc != 0:
        raise RuntimeError("cfz_colloc_band_info: " + lib.cfz_last_error().decode())
    return int(nk.value), int(kb.value), int(bb.value)


def colloc_elimination_info(n_sets, N_per_set=5, n_obs=6, pairs=None, has_final=None, structured=1):
    """`cfz_colloc_elimination_info`: dict(nk, kb, band_bytes, alg_bytes, workspace_bytes) of the elimination one (joint) collocation plan
    goes through (host arithmetic, needs no GPU); alg_bytes: bytes per Newton system, bench.py's roofline of the planning kernels."""
    lib = load_library()
    ns = np.ascontiguousarray(np.asarray(n_sets, np.int32))
    hf = None if has_final is None else np.ascontiguousarray(np.asarray(has_final, np.int32))
    pr = None if pairs is None else np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    nk, kb, bb, ab, wb = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
    lib.cfz_colloc_elimination_info.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5
    rc = lib.cfz_colloc_elimination_info(len(ns), _ptr(ns), _ptr(hf), int(N_per_set), int(n_obs), 0 if pr is None else len(pr), _ptr(pr), int(structured),
                                         C.addressof(nk), C.addressof(kb), C.addressof(bb), C.addressof(ab), C.addressof(wb))
    if rc != 0:
        raise RuntimeError("cfz_colloc_elimination_info: " + lib.cfz_last_error().decode())
    return dict(nk=int(nk.value), kb=int(kb.value), band_bytes=int(bb.value), alg_bytes=int(ab.value), workspace_bytes=int(wb.value))


def joint_colloc_batch(spec, scenarios, pairs=None, device=0, ws=None, **options):
    """`cfz_joint_colloc`: B joint collocation plans (one workgroup each) of V vehicles with one shared dt per plan and pairwise
    separation rows (multi_vehicle_planner.py:343-480).  scenarios: list of dict(init_poses [V][3], tubes, guesses, dt0,
    final_headings) with per-vehicle entries as in `colloc`, the same V in every scenario; pairs: list of (a, b) vehicle
    index pairs, default all.  Returns a list of dict(traj: per vehicle [N_a, 6, 7], dt, status, iters, cost)."""
    lib = load_library()
    co = _CCollocOptions()
    lib.cfz_default_colloc_options(C.byref(co))
    for k, v in options.items():
        if not hasattr(co, k):
            raise TypeError(f"unknown collocation option {k!r}")
        setattr(co, k, v)
    B, V = len(scenarios), len(scenarios[0]["tubes"])
    assert all(len(sc["tubes"]) == V for sc in scenarios)
    tubes = [t for sc in scenarios for t in sc["tubes"]]
    n_sets = np.array([len(t) + 1 for t in tubes], dtype=np.int32)
    Np = co.N_per_set * (n_sets - 1) * 6
    tube = np.ascontiguousarray(np.concatenate([np.concatenate([np.concatenate([np.asarray(A, float).ravel(), np.asarray(b, float).ravel()])
                                                                for cellpair in t for (A, b) in cellpair]) for t in tubes]))
    init = _f64(np.concatenate([np.asarray(sc["init_poses"], float).reshape(V, 3) for sc in scenarios]), (B * V, 3))
    fhs = [fh for sc in scenarios for fh in (sc.get("final_headings") or [None] * V)]
    fh = np.array([np.nan if f is None else float(f) for f in fhs])
    guesses = [g for sc in scenarios for g in sc["guesses"]]
    guess = np.ascontiguousarray(np.concatenate([_f64(np.asarray(g, float), (int(Np[b]), 7)) for b, g in enumerate(guesses)]))
    dt0 = _f64(np.array([float(sc["dt0"]) for sc in scenarios]), (B,))
    pr = None if pairs is None else np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    traj, dt = np.zeros((int(Np.sum()), 7)), np.zeros(B)
    status, iters, cost = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
    cs = spec.to_c()
    args = (B, V, C.byref(cs), C.byref(co), _ptr(n_sets), _ptr(init), _ptr(fh), _ptr(tube), _ptr(guess), _ptr(dt0),
            0 if pr is None else len(pr), _ptr(pr), _ptr(traj), _ptr(dt), _ptr(status), _ptr(iters), _ptr(cost))
    rc = lib.cfz_joint_colloc(int(device), *args) if ws is None else lib.cfz_joint_colloc_w(ws._w, *args)
    if rc != 0:
        raise RuntimeError("cfz_joint_colloc: " + lib.cfz_last_error().decode())
    out, o = [], 0
    for b in range(B):
        tr = []
        for a in range(V):
            n = int(Np[b * V + a])
            tr.append(traj[o : o + n].reshape(-1, 6, 7).copy())
            o += n
        out.append(dict(traj=tr, dt=float(dt[b]), status=int(status[b]), iters=int(iters[b]), cost=float(cost[b])))
    return out


def joint_colloc(spec, init_poses, tubes, guesses, dt0, final_headings=None, pairs=None, device=0, **options):
    """One joint plan: `joint_colloc_batch` with a single scenario.  Returns dict(traj: per vehicle [N_a, 6, 7], dt, status, iters, cost)."""
    return joint_colloc_batch(spec, [dict(init_poses=init_poses, tubes=tubes, guesses=guesses, dt0=dt0, final_headings=final_headings)],
                              pairs=pairs, device=device, **options)[0]


class Engine:
    """One batched solver instance on one GPU (a `cfz_handle`)."""

    def __init__(self, spec: ProblemSpec, max_batch: int, device: int = 0, **options):
        self.lib = load_library()
        self.spec = spec
        self.max_batch = int(max_batch)
        self._h = C.c_void_p()
        cs, co = spec.to_c(), default_options(**options)
        if self.lib.cfz_create(C.byref(cs), C.byref(co), int(device), self.max_batch, C.byref(self._h)) != 0:
            raise RuntimeError("cfz_create: " + self.lib.cfz_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.cfz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: " + self.lib.cfz_last_error().decode())

    # ---- host-buffer path ------------------------------------------------------------------
    def solve(self, x0, ref, nbr, zu, want_duals=True, carry=None, slots=None):
        """x0 [B,5], ref [B,3,N], nbr [B,n_nbr,3,N], zu [B,7,N] (warm start) ->
        dict(zu, status, iters, cost, kkt_err, min_sep[, l, m, lam_ij, lam_ji, s], solve_ms).
        carry: int/bool [B]; carry[b] says that this solve of slot b is the MPC iteration following the one last
        solved in slot b, so the interior point starts from its multipliers (`cfz_mpc_set_carry`).
        slots: int [B]; the carry record instance b reads and refreshes (`cfz_mpc_set_slots`; default b)."""
        sp = self.spec
        N, no, nn = sp.N, sp.n_obs, sp.n_nbr
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B = x0.shape[0]
        x0 = _f64(x0, (B, 5)); ref = _f64(ref, (B, 3, N)); zu = _f64(zu, (B, 7, N))
        nbr = _f64(nbr, (B, nn, 3, N)) if nn else None
        self._ck(self.lib.cfz_mpc_set_params(self._h, B, _ptr(x0), _ptr(ref), _ptr(nbr)), "cfz_mpc_set_params")
        self._ck(self.lib.cfz_mpc_set_warm(self._h, B, _ptr(zu)), "cfz_mpc_set_warm")
        if carry is not None:
            cflags = np.ascontiguousarray(np.broadcast_to(np.asarray(carry), (B,)), dtype=np.int32)
            self._ck(self.lib.cfz_mpc_set_carry(self._h, B, _ptr(cflags)), "cfz_mpc_set_carry")
        if slots is not None:
            sl = np.ascontiguousarray(np.broadcast_to(np.asarray(slots), (B,)), dtype=np.int32)
            self._ck(self.lib.cfz_mpc_set_slots(self._h, B, _ptr(sl)), "cfz_mpc_set_slots")
        self._ck(self.lib.cfz_mpc_solve(self._h, B), "cfz_mpc_solve")
        out = dict(zu=np.empty((B, 7, N)), status=np.empty(B, np.int32), iters=np.empty(B, np.int32),
                   cost=np.empty(B), kkt_err=np.empty(B), min_sep=np.empty(B))
        if want_duals:
            out.update(l=np.zeros((B, N, 4 * no)), m=np.zeros((B, N, 4 * no)), lam_ij=np.zeros((B, nn, N, 4)),
                       lam_ji=np.zeros((B, nn, N, 4)), s=np.zeros((B, nn, N, 2)))
        self._ck(self.lib.cfz_mpc_get(self._h, B, _ptr(out["zu"]), _ptr(out.get("l")), _ptr(out.get("m")),
                                      _ptr(out.get("lam_ij")), _ptr(out.get("lam_ji")), _ptr(out.get("s"))), "cfz_mpc_get")
        self._ck(self.lib.cfz_mpc_stats(self._h, B, _ptr(out["status"]), _ptr(out["iters"]), _ptr(out["cost"]),
                                        _ptr(out["kkt_err"]), _ptr(out["min_sep"])), "cfz_mpc_stats")
        out["solve_ms"] = self.last_solve_ms()
        return out

    def kernel_info(self):
        """(LDS bytes per instance, resident instances per CU)."""
        a, b = C.c_int32(), C.c_int32()
        self._ck(self.lib.cfz_kernel_info(self._h, C.byref(a), C.byref(b)), "cfz_kernel_info")
        return a.value, b.value

    def last_solve_ms(self):
        return float(self.lib.cfz_last_solve_ms(self._h))

    def set_carry(self, flags):
        """`cfz_mpc_set_carry` for the next solve (host or device path): flags [B] int."""
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        self._ck(self.lib.cfz_mpc_set_carry(self._h, len(flags), _ptr(flags)), "cfz_mpc_set_carry")

    def set_carry_device(self, B, d_flags):
        """`cfz_mpc_set_carry_device`: flags as a device int32 array (torch tensor or pointer), no copy, no sync."""
        p = C.c_void_p(d_flags.data_ptr() if hasattr(d_flags, "data_ptr") else int(d_flags))
        self._ck(self.lib.cfz_mpc_set_carry_device(self._h, int(B), p), "cfz_mpc_set_carry_device")

    # ---- device-pointer path (torch tensors or any object with data_ptr()) --------------------
    def solve_device(self, B, d_x0, d_ref, d_nbr, d_zu, d_status, d_iters, d_stats, stream=None):
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))
        self._ck(self.lib.cfz_mpc_solve_device(self._h, int(B), ptr(d_x0), ptr(d_ref), ptr(d_nbr), ptr(d_zu),
                                               ptr(d_status), ptr(d_iters), ptr(d_stats),
                                               None if stream is None else C.c_void_p(int(stream))), "cfz_mpc_solve_device")

    def vsl_step(self, S, V, d_own, T, d_table, d_k0, t, d_allpred, d_pred, d_state, d_status, d_iters, d_stats, d_carry, stream=None):
        """`cfz_vsl_step`: one iteration of the vehicle-sharded closed loop on device tensors, no host synchronisation."""
        ptr = lambda x: C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._ck(self.lib.cfz_vsl_step(self._h, int(S), int(V), int(d_own.numel()), ptr(d_own), int(T), ptr(d_table), ptr(d_k0), int(t),
                                       ptr(d_allpred), ptr(d_pred), ptr(d_state), ptr(d_status), ptr(d_iters), ptr(d_stats), ptr(d_carry),
                                       None if stream is None else C.c_void_p(int(stream))), "cfz_vsl_step")

    # ---- dual warm start (Vehicle.dual_ws) ------------------------------------------------------------
    def dual_ws(self, poses):
        """poses [n,3] (x,y,psi) -> (l [n,4 n_obs], m [n,4 n_obs], d [n,n_obs])."""
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        n, no = poses.shape[0], self.spec.n_obs
        poses = _f64(poses, (n, 3))
        l, m, d = np.zeros((n, 4 * no)), np.zeros((n, 4 * no)), np.zeros((n, no))
        self._ck(self.lib.cfz_dual_ws(self._h, n, _ptr(poses), _ptr(l), _ptr(m), _ptr(d)), "cfz_dual_ws")
        return l, m, d

    def joint_dual_ws(self, poses_this, poses_other):
        """`cfz_joint_dual_ws`: poses [n,3] of two vehicles -> (lam [n,4], mu [n,4], s [n,2], d [n])."""
        pa = _f64(np.asarray(poses_this, float), (len(poses_this), 3))
        pb = _f64(np.asarray(poses_other, float), (len(pa), 3))
        n = len(pa)
        lam, mu, s, d = np.zeros((n, 4)), np.zeros((n, 4)), np.zeros((n, 2)), np.zeros(n)
        self._ck(self.lib.cfz_joint_dual_ws(self._h, n, _ptr(pa), _ptr(pb), _ptr(lam), _ptr(mu), _ptr(s), _ptr(d)), "cfz_joint_dual_ws")
        return lam, mu, s, d

    # ---- batched closed loop ------------------------------------------------------------------------
    def loop_init(self, ref_table, k0, noise=None):
        """ref_table [V,T,7] (x,y,psi,v,delta,a,w), k0 int32 [S], noise [S,V,5] or None."""
        V = self.spec.n_nbr + 1
        ref_table = np.ascontiguousarray(ref_table, dtype=np.float64)
        T = ref_table.shape[1]
        ref_table = _f64(ref_table, (V, T, 7))
        k0 = np.ascontiguousarray(k0, dtype=np.int32)
        S = k0.shape[0]
        if noise is not None:
            noise = _f64(noise, (S, V, 5))
        self._S, self._V = S, V
        self._ck(self.lib.cfz_loop_init(self._h, S, T, _ptr(ref_table), _ptr(k0), _ptr(noise)), "cfz_loop_init")

    def loop_step(self):
        self._ck(self.lib.cfz_loop_step(self._h), "cfz_loop_step")

    def loop_run(self, K):
        """K closed-loop iterations in one persistent launch (same results as K x loop_step)."""
        self._ck(self.lib.cfz_loop_run(self._h, int(K)), "cfz_loop_run")
        return int(self.lib.cfz_loop_last_iterations(self._h))

    def loop_last_converged(self):
        """Solves of the last `loop_run` that converged (status 0)."""
        return int(self.lib.cfz_loop_last_converged(self._h))

    def loop_last_status_counts(self):
        """How the solves of the last `loop_run` ended: [6] counts of status 0..5."""
        c = np.zeros(6, dtype=np.int64)
        self._ck(self.lib.cfz_loop_last_status_counts(self._h, c.ctypes.data_as(C.c_void_p)), "cfz_loop_last_status_counts")
        return c

    def loop_get(self):
        S, V, N = self._S, self._V, self.spec.N
        out = dict(state=np.empty((S, V, 5)), pred=np.empty((S, V, 7, N)), status=np.empty((S, V), np.int32),
                   iters=np.empty((S, V), np.int32))
        self._ck(self.lib.cfz_loop_get(self._h, _ptr(out["state"]), _ptr(out["pred"]), _ptr(out["status"]),
                                       _ptr(out["iters"])), "cfz_loop_get")
        return out
