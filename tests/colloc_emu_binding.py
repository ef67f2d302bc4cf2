"""ctypes binding of the test-only CPU build of the collocation solver source (tests/emu/cfz_colloc_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "tests", "_build", "libcfz_colloc_emu.so")
_INTS = "N Nps n_chk n_obs has_final max_iter max_backtrack filter_cap pad0 pad1".split()
_OPTS = ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
         "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal reg_dual curv_kappa").split()


class CSpec(C.Structure):
    _fields_ = ([(k, C.c_int) for k in _INTS] +
                [(k, C.c_double) for k in "wb dmin shrink final_heading dt0".split()] +
                [("init_pose", C.c_double * 3), ("bounds", C.c_double * 12), ("g", C.c_double * 4),
                 ("A", C.c_double * 36), ("B", C.c_double * 6)] + [(k, C.c_double) for k in _OPTS] +
                [("obs_tab", C.c_void_p), ("tube", C.c_void_p)])


def build(force=False):
    srcs = [os.path.join(ROOT, "tests", "emu", "cfz_colloc_emu.cpp")] + [os.path.join(ROOT, "conflict_rez_amd", "csrc", f)
                                                                         for f in ("cfz_colloc.inl", "cfz_plan.inl", "cfz_solver.inl")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-Wno-unknown-pragmas", "-Wno-maybe-uninitialized", "-fPIC", "-shared", "-o", _LIB, srcs[0]])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.cfzc_emu_sizeof_spec() == C.sizeof(CSpec)
    return _lib


def make_spec(nlp, opt):
    """nlp: oracle.colloc_nlp.CollocNlp, opt: oracle.ipm.IpmOptions -> (CSpec, keep-alive arrays)."""
    s = CSpec()
    s.N, s.Nps, s.n_chk, s.n_obs, s.has_final = nlp.N, nlp.Nps, nlp.n_chk, nlp.n_obs, int(nlp.final_heading is not None)
    s.max_iter, s.max_backtrack, s.filter_cap = opt.max_iter, opt.max_backtrack, opt.filter_cap
    s.wb, s.dmin, s.shrink = nlp.wb, nlp.dmin, nlp.shrink
    s.final_heading = float(nlp.final_heading) if nlp.final_heading is not None else 0.0
    s.init_pose[:] = list(nlp.init_pose)
    s.bounds[:] = list(nlp.bounds)
    s.g[:] = list(nlp.g)
    s.A[:] = list(nlp.A.ravel())
    s.B[:] = list(nlp.B)
    for k in _OPTS:
        setattr(s, k, getattr(opt, k))
    tab = np.zeros((max(nlp.n_obs, 1), 20))
    for j in range(nlp.n_obs):
        tab[j, :8], tab[j, 8:12], tab[j, 12:] = nlp.A_obs[j].ravel(), nlp.b_obs[j], np.asarray(nlp.PV[j]).ravel()
    tube = np.zeros((nlp.n_chk, 2, 12))
    for i in range(1, nlp.S):
        for f, key in enumerate(("back", "front")):
            A, b = nlp.tube[i][key]
            tube[i - 1, f, :8], tube[i - 1, f, 8:] = np.asarray(A, float).ravel(), b
    s.obs_tab, s.tube = tab.ctypes.data, tube.ctypes.data
    return s, (tab, tube)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def dims(nlp, opt):
    s, keep = make_spec(nlp, opt)
    out = np.zeros(14, np.int32)
    lib().cfzc_emu_dims(C.byref(s), _p(out))
    return dict(zip("np nr n m nk iDt sO sT rO rC rR rT rF rH".split(), map(int, out)))


def select(nlp, opt, X, prev=None):
    s, keep = make_spec(nlp, opt)
    sel = np.zeros((nlp.np, nlp.n_obs), np.uint8) if prev is None else np.array(prev, np.uint8)
    lib().cfzc_emu_select(C.byref(s), _p(np.ascontiguousarray(X)), _p(sel))
    return sel


def evaluate(nlp, opt, sel, X, nu):
    s, keep = make_spec(nlp, opt)
    f, c, g, jt = C.c_double(), np.zeros(nlp.m), np.zeros(nlp.n), np.zeros(nlp.n)
    X, nu, sel = np.ascontiguousarray(X, float), np.ascontiguousarray(nu, float), np.ascontiguousarray(sel, np.uint8)
    lib().cfzc_emu_eval(C.byref(s), _p(sel), _p(X), _p(nu), C.byref(f), _p(c), _p(g), _p(jt))
    return f.value, c, g, jt


def kkt(nlp, opt, sel, X, nu, sig=None, delta=0.0):
    s, keep = make_spec(nlp, opt)
    nt = nlp.n + nlp.m
    K = np.zeros((nt, nt))
    sig = np.zeros(nlp.n) if sig is None else np.ascontiguousarray(sig, float)
    fn = lib().cfzc_emu_kkt
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    bw = fn(C.addressof(s), _p(np.ascontiguousarray(sel, np.uint8)), _p(np.ascontiguousarray(X, float)), _p(np.ascontiguousarray(nu, float)), _p(sig), delta, _p(K))
    return K, bw


def half_bandwidth(nlp, opt):
    s, keep = make_spec(nlp, opt)
    return lib().cfzc_emu_half_bandwidth(C.byref(s))


def solve(nlp, X0, opt):
    """X0: points and dt (7 np + 1) -> dict(X, iters, status, f, err, mu)."""
    s, keep = make_spec(nlp, opt)
    X = np.array(X0[: nlp.iDt + 1], dtype=np.float64)
    oi, od = np.zeros(2, np.int32), np.zeros(12)
    assert lib().cfzc_emu_solve(C.byref(s), _p(X), _p(oi), _p(od)) == 0
    return dict(X=X, iters=int(oi[0]), status=int(oi[1]), f=od[0], err=od[1], mu=od[2])
