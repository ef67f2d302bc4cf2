"""Oracle (test infrastructure): ctypes binding of the plain-C port `oracle/cfz_port.c`."""
import ctypes as C
import os
import subprocess

import numpy as np

from .ipm import IpmOptions
from .mpc_nlp import MpcSpec, polytope_vertices

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libcfz_port.so")
MAXB = 16


def build(force=False):
    src = os.path.join(_HERE, "cfz_port.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        tmp = _LIB + ".%d.tmp" % os.getpid()  # built aside and renamed: parallel test workers never see a half-written library
        # -mfma where the host has it: the Riccati sweep mirrors the kernel's matrix instructions with fma() per term (libm's software fma is
        # exact too, and fifty times slower); -ffp-contract=off: no other product-sum may be fused behind the source's back
        has_fma = False
        try:
            has_fma = " fma " in open("/proc/cpuinfo").read()
        except OSError:
            pass
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + (["-mfma"] if has_fma else []) + ["-fPIC", "-shared", "-o", tmp, src, "-lm"])
        os.replace(tmp, _LIB)
    return _LIB


class _Spec(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("n_obs", C.c_int), ("n_nbr", C.c_int), ("rk_substeps", C.c_int), ("max_iter", C.c_int),
        ("dt", C.c_double), ("wb", C.c_double), ("dmin", C.c_double),
        ("g", C.c_double * 4), ("bounds", C.c_double * 12), ("weights", C.c_double * 6),
        ("A_obs", C.c_double * (MAXB * 8)), ("b_obs", C.c_double * (MAXB * 4)), ("V_obs", C.c_double * (MAXB * 8)),
    ] + [(k, C.c_double) for k in (
        "tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
        "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal stall_kappa warm_push").split()
    ] + [("filter_cap", C.c_int), ("max_backtrack", C.c_int), ("stall_iters", C.c_int), ("row_curvature", C.c_int), ("vv_rows", C.c_int), ("shift_after", C.c_int), ("stag_win", C.c_int), ("err_stall", C.c_int), ("carry_shift", C.c_int), ("resto", C.c_int), ("reg_dual_rows", C.c_double), ("resto_first", C.c_double)]


def make_spec(spec: MpcSpec, opt: IpmOptions = IpmOptions()):
    s = _Spec()
    s.N, s.n_obs, s.n_nbr, s.rk_substeps, s.max_iter = spec.N, spec.n_obs, spec.n_nbr, spec.rk_substeps, opt.max_iter
    s.dt, s.wb, s.dmin = spec.dt, spec.wb, spec.dmin
    s.g[:] = list(spec.g)
    s.bounds[:] = list(spec.bounds)
    s.weights[:] = list(spec.weights)
    A = np.zeros((MAXB, 4, 2)); b = np.zeros((MAXB, 4)); V = np.zeros((MAXB, 4, 2))
    for j in range(spec.n_obs):
        A[j], b[j] = spec.A_obs[j], spec.b_obs[j]
        V[j] = polytope_vertices(spec.A_obs[j], spec.b_obs[j])[0]
    s.A_obs[:] = list(A.ravel()); s.b_obs[:] = list(b.ravel()); s.V_obs[:] = list(V.ravel())
    for k in ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
              "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal stall_kappa warm_push").split():
        setattr(s, k, getattr(opt, k))
    s.filter_cap, s.max_backtrack, s.stall_iters = opt.filter_cap, opt.max_backtrack, opt.stall_iters
    s.row_curvature = int(opt.row_curvature)
    s.vv_rows = int(spec.vv_rows)
    s.shift_after = int(opt.shift_after)
    s.stag_win = int(opt.shift_stagnation)
    s.err_stall = int(opt.err_stall_iters)
    s.carry_shift = int(opt.carry_shift)
    s.resto = int(opt.restoration)
    s.reg_dual_rows = float(opt.reg_dual_rows)
    s.resto_first = float(opt.resto_first)
    return s


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.cfz_port_sizeof_spec() == C.sizeof(_Spec), "struct layout mismatch"
    return _lib


MAXN, MAXR = 64, 32


class Carry(C.Structure):
    """cfz_port_carry: the state a converged solve hands to the next MPC iteration of the same vehicle."""
    _fields_ = [("valid", C.c_int), ("sel", C.c_int * (MAXN * MAXB)), ("z", C.c_double * (MAXN * MAXR)),
                ("zl", C.c_double * (MAXN * 6)), ("zu", C.c_double * (MAXN * 6)), ("pi0", C.c_double * 5),
                ("pi", C.c_double * (MAXN * 5)), ("mu", C.c_double), ("shifted", C.c_int), ("pad", C.c_int)]


def solve(spec: MpcSpec, x0, ref, nbr, warm_p, opt: IpmOptions = IpmOptions(), trace_cap=0, carry=None):
    """warm_p [N,7] -> dict(p [N,7], sep [N,nb], cert [N,nb], iters, status, f, err, mu, trace, carry).
    carry: the `carry` of the previous MPC iteration's result (None = cold multipliers)."""
    lib = _load()
    assert lib.cfz_port_sizeof_carry() == C.sizeof(Carry)
    N, nb = spec.N, spec.n_blk
    cs = make_spec(spec, opt)
    p = np.ascontiguousarray(warm_p, dtype=np.float64).copy()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    nbr = np.ascontiguousarray(nbr if spec.n_nbr else np.zeros(1), dtype=np.float64)
    sep = np.zeros((N, nb)); cert = np.zeros((N, nb), dtype=np.int32)
    stats = np.zeros(2, dtype=np.int32); fst = np.zeros(3)
    trace = np.zeros((max(trace_cap, 1), 4 + 7 * N))
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    cout = Carry()
    rc = lib.cfz_port_solve_carry(C.byref(cs), dp(x0), dp(ref), dp(nbr), dp(p), dp(sep), dp(cert), dp(stats), dp(fst),
                                  dp(trace) if trace_cap else None, C.c_int(trace_cap),
                                  C.byref(carry) if carry is not None else None, C.byref(cout))
    assert rc == 0
    return dict(p=p, sep=sep, cert=cert, iters=int(stats[0]), status=int(stats[1]), f=fst[0], err=fst[1], mu=fst[2],
                trace=trace[: min(trace_cap, stats[0] + 1)], carry=cout if cout.valid else None)
