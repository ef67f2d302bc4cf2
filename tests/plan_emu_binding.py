"""ctypes binding of the test-only CPU build of the planning solver source (tests/emu/cfz_plan_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "tests", "_build", "libcfz_plan_emu.so")
_INTS = "T N n_chk has_final bounded_input max_iter max_backtrack filter_cap stall_iters pad0".split()
_OPTS = ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
         "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal reg_dual curv_kappa stall_kappa").split()


class PSpec(C.Structure):
    _fields_ = [(k, C.c_int) for k in _INTS] + [("dt", C.c_double), ("wb", C.c_double), ("shrink", C.c_double),
                                                ("final_heading", C.c_double), ("init_pose", C.c_double * 3),
                                                ("bounds", C.c_double * 12)] + [(k, C.c_double) for k in _OPTS]


def build(force=False):
    srcs = [os.path.join(ROOT, "tests", "emu", "cfz_plan_emu.cpp"), os.path.join(ROOT, "conflict_rez_amd", "csrc", "cfz_plan.inl")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        tmp = _LIB + ".%d.tmp" % os.getpid()  # built aside and renamed: parallel test workers never see a half-written library
        subprocess.check_call(["g++", "-O2", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, _LIB)
    return _LIB


def make_spec(nlp, opt):
    """nlp: oracle.plan_nlp.StateWsNlp, opt: oracle.ipm.IpmOptions -> (PSpec, tube array [n_chk,2,12])."""
    s = PSpec()
    s.T, s.N, s.n_chk, s.has_final = nlp.T, nlp.N, nlp.n_chk, int(nlp.final_heading is not None)
    s.bounded_input = int(np.isfinite(nlp.xl[5]))
    s.max_iter, s.max_backtrack, s.filter_cap, s.stall_iters = opt.max_iter, opt.max_backtrack, opt.filter_cap, opt.stall_iters
    s.dt, s.wb, s.shrink = nlp.dt, nlp.wb, nlp.shrink
    s.final_heading = float(nlp.final_heading) if nlp.final_heading is not None else 0.0
    s.init_pose[:] = list(nlp.init_pose)
    s.bounds[:] = [float(v) for v in nlp.bounds]
    for k in _OPTS:
        setattr(s, k, getattr(opt, k))
    tube = np.zeros((nlp.n_chk, 2, 12))
    for i in range(1, nlp.S):
        for f, key in enumerate(("back", "front")):
            A, b = nlp.tube[i][key]
            tube[i - 1, f, :8], tube[i - 1, f, 8:] = np.asarray(A, float).ravel(), b
    return s, tube


_lib = None


def solve(nlp, X0, opt):
    """X0: packed guess (oracle layout) -> dict(X, iters, status, f, err, mu)."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.cfzp_emu_sizeof_spec() == C.sizeof(PSpec)
    s, tube = make_spec(nlp, opt)
    assert _lib.cfzp_emu_n(C.byref(s)) == nlp.n
    X = np.array(X0, dtype=np.float64)
    oi = np.zeros(2, np.int32); od = np.zeros(3)
    dp = lambda a: a.ctypes.data_as(C.c_void_p)
    assert _lib.cfzp_emu_state_ws(C.byref(s), dp(tube), dp(X), dp(oi), dp(od)) == 0
    return dict(X=X, iters=int(oi[0]), status=int(oi[1]), f=od[0], err=od[1], mu=od[2])
