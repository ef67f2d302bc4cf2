"""ctypes binding of the test-only CPU build of the kernel source (tests/emu/cfz_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "tests", "_build", "libcfz_emu.so")
_OPTS = ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
         "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal stall_kappa warm_push reg_dual_rows resto_first").split()


class KSpec(C.Structure):
    _fields_ = [(k, C.c_int) for k in "N n_obs n_nbr rk_substeps max_iter max_backtrack filter_cap stall_iters row_curvature vv_rows stag_win err_stall carry_shift pad_ks shift_after resto".split()] + [
        ("dt", C.c_double), ("wb", C.c_double), ("dmin", C.c_double),
        ("g", C.c_double * 4), ("bounds", C.c_double * 12), ("weights", C.c_double * 6),
        ("A_obs", C.c_double * 64), ("b_obs", C.c_double * 32), ("V_obs", C.c_double * 64),
    ] + [(k, C.c_double) for k in _OPTS] + [("obs_tab", C.c_void_p)]


def build(force=False, sanitize=False):
    srcs = [os.path.join(ROOT, "tests", "emu", "cfz_emu.cpp"),
            os.path.join(ROOT, "conflict_rez_amd", "csrc", "cfz_solver.inl")]
    lib = _LIB.replace(".so", "_asan.so") if sanitize else _LIB
    lps = os.environ.get("CFZ_EMU_LPS", "")  # lanes per stage of the emulated kernel (default: the source's)
    if lps:
        lib = lib.replace(".so", f"_lps{lps}.so")
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        flags = ["-O1", "-g", "-fsanitize=address,undefined"] if sanitize else ["-O2"]
        tmp = lib + ".%d.tmp" % os.getpid()  # built aside and renamed: parallel test workers never see a half-written library
        fma_ = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []  # (the Riccati sweep's fma() per term: hardware where there is one; as oracle/port.py)
        subprocess.check_call(["g++", *flags, *([f"-DCFZ_LPS={lps}"] if lps else []), "-ffp-contract=off", *fma_, "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, lib)
    return lib


def make_kspec(spec, opt):
    """spec: oracle.mpc_nlp.MpcSpec, opt: oracle.ipm.IpmOptions"""
    from oracle.mpc_nlp import polytope_vertices

    s = KSpec()
    s.N, s.n_obs, s.n_nbr, s.rk_substeps = spec.N, spec.n_obs, spec.n_nbr, spec.rk_substeps
    s.max_iter, s.max_backtrack, s.filter_cap, s.stall_iters = opt.max_iter, opt.max_backtrack, opt.filter_cap, opt.stall_iters
    s.row_curvature = int(opt.row_curvature)
    s.vv_rows = int(getattr(spec, "vv_rows", False))
    s.shift_after = int(opt.shift_after)
    s.resto = int(opt.restoration)
    s.stag_win = int(opt.shift_stagnation)
    s.err_stall = int(opt.err_stall_iters)
    s.carry_shift = int(opt.carry_shift)
    s.dt, s.wb, s.dmin = spec.dt, spec.wb, spec.dmin
    s.g[:] = list(spec.g); s.bounds[:] = list(spec.bounds); s.weights[:] = list(spec.weights)
    A = np.zeros((8, 4, 2)); b = np.zeros((8, 4)); V = np.zeros((8, 4, 2))
    for j in range(spec.n_obs):
        A[j], b[j] = spec.A_obs[j], spec.b_obs[j]
        V[j] = polytope_vertices(spec.A_obs[j], spec.b_obs[j])[0]
    s.A_obs[:] = list(A.ravel()); s.b_obs[:] = list(b.ravel()); s.V_obs[:] = list(V.ravel())
    for k in _OPTS:
        setattr(s, k, getattr(opt, k))
    # the table the kernel reads: per obstacle A[4][2], b[4], V[4][2]; kept alive by the returned struct
    tab = np.ascontiguousarray(np.concatenate([A.reshape(8, 8), b, V.reshape(8, 8)], axis=1)[: max(spec.n_obs, 1)])
    s._obs_tab = tab
    s.obs_tab = tab.ctypes.data
    return s


_lib = None


def solve(spec, opt, x0, ref, nbr, zu, want_duals=True, carry=None):
    """zu [7,N] warm start -> dict(zu, iters, status, cost, err, min_sep, l, m, lam_ij, lam_ji, s, carry).
    carry: the `carry` record (numpy array) of the previous MPC iteration's result; None = cold multipliers."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.cfz_emu_sizeof_kspec() == C.sizeof(KSpec)
    N, no, nn = spec.N, spec.n_obs, spec.n_nbr
    ks = make_kspec(spec, opt)
    zu = np.ascontiguousarray(zu, dtype=np.float64).copy()
    x0 = np.ascontiguousarray(x0, dtype=np.float64); ref = np.ascontiguousarray(ref, dtype=np.float64)
    nbr = np.ascontiguousarray(nbr if nn else np.zeros(1), dtype=np.float64)
    oi = np.zeros(2, dtype=np.int32); od = np.zeros(3)
    l = np.zeros((N, 4 * no)); m = np.zeros((N, 4 * no))
    lij = np.zeros((nn, N, 4)); lji = np.zeros((nn, N, 4)); s = np.zeros((nn, N, 2))
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    dd = (lambda a: dp(a)) if want_duals else (lambda a: None)
    wst = np.zeros(_lib.cfz_emu_carry_doubles(N, no + nn)) if carry is None else np.array(carry, dtype=np.float64)
    rc = _lib.cfz_emu_solve(C.byref(ks), dp(x0), dp(ref), dp(nbr), dp(zu), dp(oi), dp(od), dd(l), dd(m), dd(lij), dd(lji), dd(s),
                            dp(wst), C.c_int(0 if carry is None else 1))
    assert rc > 0
    return dict(zu=zu, iters=int(oi[0]), status=int(oi[1]), cost=od[0], err=od[1], min_sep=od[2], l=l, m=m, lam_ij=lij,
                lam_ji=lji, s=s, lds_doubles=rc, carry=wst)
