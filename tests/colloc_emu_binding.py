"""ctypes binding of the test-only CPU build of the collocation solver source (tests/emu/cfz_colloc_emu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "tests", "_build", "libcfz_colloc_emu.so")
_OPTS = ("tol constr_viol_tol dual_inf_tol compl_inf_tol mu_init kappa_eps kappa_mu theta_mu tau_min bound_push "
         "bound_frac s_max kappa_sigma eta_phi gamma_theta gamma_phi delta_sw s_theta s_phi reg_primal reg_dual curv_kappa").split()
MAX_VEH, MAX_PAIRS = 4, 6


class CSpec(C.Structure):
    _fields_ = ([(k, C.c_int) for k in "V Nps n_obs n_pairs max_iter max_backtrack filter_cap no_prox vv_rows pad0".split()] +
                [("N", C.c_int * MAX_VEH), ("n_chk", C.c_int * MAX_VEH), ("has_final", C.c_int * MAX_VEH),
                 ("pair_a", C.c_int * MAX_PAIRS), ("pair_b", C.c_int * MAX_PAIRS)] +
                [(k, C.c_double) for k in "wb dmin shrink dt0".split()] +
                [("final_heading", C.c_double * MAX_VEH), ("init_pose", C.c_double * (3 * MAX_VEH)), ("bounds", C.c_double * 12),
                 ("g", C.c_double * 4), ("A", C.c_double * 36), ("B", C.c_double * 6)] + [(k, C.c_double) for k in _OPTS] +
                [("obs_tab", C.c_void_p), ("tube", C.c_void_p * MAX_VEH)])


def build(force=False):
    srcs = [os.path.join(ROOT, "tests", "emu", "cfz_colloc_emu.cpp")] + [os.path.join(ROOT, "conflict_rez_amd", "csrc", f)
                                                                         for f in ("cfz_colloc.inl", "cfz_struct.inl", "cfz_jstruct.inl", "cfz_plan.inl", "cfz_solver.inl")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        tmp = _LIB + ".%d.tmp" % os.getpid()  # built aside and renamed: parallel test workers never see a half-written library
        subprocess.check_call(["g++", "-O2", "-Wno-unknown-pragmas", "-Wno-maybe-uninitialized", "-fPIC", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, _LIB)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        assert _lib.cfzc_emu_sizeof_spec() == C.sizeof(CSpec)
    return _lib


def make_spec(nlp, opt):
    """nlp: oracle.colloc_nlp.JointCollocNlp (or CollocNlp), opt: oracle.ipm.IpmOptions -> (CSpec, keep-alive arrays)."""
    s = CSpec()
    s.V, s.Nps, s.n_obs, s.n_pairs = nlp.V, nlp.Nps, nlp.n_obs, len(nlp.pairs)
    s.max_iter, s.max_backtrack, s.filter_cap = opt.max_iter, opt.max_backtrack, opt.filter_cap
    s.no_prox = int(getattr(opt, "no_prox", 0))  # 1: IPOPT's form of the dual regularisation (exact solutions at tight tolerances)
    s.vv_rows = int(getattr(nlp, "vv", 0))  # vertex-vertex rows in the working sets, as the numpy statement selects them
    s.wb, s.dmin, s.shrink = nlp.wb, nlp.dmin, nlp.shrink
    keep = []
    for a, v in enumerate(nlp.veh):
        s.N[a], s.n_chk[a] = nlp.N[a], nlp.n_chk[a]
        fh = v.get("final_heading")
        s.has_final[a], s.final_heading[a] = int(fh is not None), float(fh) if fh is not None else 0.0
        for i in range(3):
            s.init_pose[3 * a + i] = float(v["init_pose"][i])
        tube = np.zeros((nlp.n_chk[a], 2, 12))
        for i in range(1, len(v["tube"])):
            for f, key in enumerate(("back", "front")):
                A, b = v["tube"][i][key]
                tube[i - 1, f, :8], tube[i - 1, f, 8:] = np.asarray(A, float).ravel(), b
        keep.append(tube)
        s.tube[a] = tube.ctypes.data
    for e, (a, b) in enumerate(nlp.pairs):
        s.pair_a[e], s.pair_b[e] = a, b
    s.bounds[:] = list(nlp.bounds)
    s.g[:] = list(nlp.g)
    s.A[:] = list(nlp.A.ravel())
    s.B[:] = list(nlp.B)
    for k in _OPTS:
        setattr(s, k, getattr(opt, k))
    tab = np.zeros((max(nlp.n_obs, 1), 20))
    for j in range(nlp.n_obs):
        tab[j, :8], tab[j, 8:12], tab[j, 12:] = nlp.A_obs[j].ravel(), nlp.b_obs[j], np.asarray(nlp.PV[j]).ravel()
    s.obs_tab = tab.ctypes.data
    keep.append(tab)
    return s, keep


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def dims(nlp, opt):
    s, keep = make_spec(nlp, opt)
    out = np.zeros(16, np.int32)
    lib().cfzc_emu_dims(C.byref(s), _p(out))
    return dict(zip("np nr n m nk iDt sO sT sP rO rC rR rT rF rP npp".split(), map(int, out)))


def select(nlp, opt, X, prev=None):
    """Working-set codes (obstacles [np, n_obs] then pairs [npp], flat); prev: codes to keep with hysteresis."""
    s, keep = make_spec(nlp, opt)
    sel = np.zeros(nlp.np * nlp.n_obs + nlp.npp, np.uint8) if prev is None else np.array(prev, np.uint8).ravel()
    Xf = np.zeros(nlp.n)
    Xf[: len(X)] = X
    lib().cfzc_emu_select(C.byref(s), _p(Xf), _p(sel), int(prev is None))
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
    oi, od = np.zeros(2, np.int32), np.zeros(20)
    assert lib().cfzc_emu_solve(C.byref(s), _p(X), _p(oi), _p(od)) == 0
    return dict(X=X, iters=int(oi[0]), status=int(oi[1]), f=od[0], err=od[1], mu=od[2])
