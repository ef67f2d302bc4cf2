"""Feasibility study for the next round (test infrastructure: uses oracle/ and the CPU build of the kernel source under tests/; nothing here
is product code).  The collocation plan's assembled KKT matrix -- exactly what `colloc_kernel` factors today as one band, ~4,100 pivots
one after the other -- eliminated interval by interval instead: the interiors of the N Radau intervals (points 1..5 of an interval and
its 30 ODE rows: 65 unknowns) independently, by dense pivoted LU, then the system over the separators (the start point of every interval
with its continuity rows, tube slacks / rows and the initial / terminal rows: 14-22 unknowns each) and the dt border.  Printed: that the
matrix has this coupling pattern, the interiors' condition numbers, and the difference to a dense solve of the whole system.

    python tools/colloc_condense_study.py [vehicle_0 .. vehicle_3]
"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scipy.linalg as sla
import colloc_emu_binding as ce
import plan_emu_binding as pe
from scipy.interpolate import interp1d
from conflict_rez_amd import scenarios, strategy as strat
from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
from conflict_rez_amd.vehicle_types import VehicleBody
from oracle import ipm
from oracle.colloc_nlp import CollocNlp
from oracle.plan_nlp import StateWsNlp, speed_guess
hist = strat.generate_strategy(4)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
    tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
sp = scenarios.parking_lot_spec(n_nbr=0, N=2)
a = sys.argv[1] if len(sys.argv) > 1 else "vehicle_1"
p = paths[a]; fh = float(p[-1, 2])
tube = [dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]]
ws = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
r = pe.solve(ws, ws.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, ws.dt)), ipm.IpmOptions(max_iter=500, hessian="exact", reg_dual=1e-9, stall_iters=0, mu_init=0.1))
z = ws.unpack(r["X"])
nlp = CollocNlp(p[0], tube, sp.A_obs, sp.b_obs, N_per_set=5, final_heading=fh)
N = nlp.N[0]
t_i = np.concatenate([k + nlp.tau for k in range(N)]) / N * z["t"][-1]
X0 = nlp.pack({k: interp1d(z["t"], z[k])(t_i) for k in ("x", "y", "psi", "v", "delta", "a", "w")}, z["t"][-1] / N)
opt = ipm.IpmOptions(max_iter=400, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1)
d = ce.dims(nlp, opt)
n, m = nlp.n, nlp.m
rng = np.random.default_rng(0)
def study(X, nu, mu, label):
    Xf = np.zeros(n); Xf[: len(X)] = X
    sel = ce.select(nlp, opt, Xf)
    f, c, g, jt = ce.evaluate(nlp, opt, sel, Xf, nu)
    Xf[nlp.sO:] = np.maximum(1e-2, -c[nlp.rR: nlp.rR + n - nlp.sO] + Xf[nlp.sO:]) if False else np.maximum(Xf[nlp.sO:], 1e-2)
    # Sigma = mu / d^2 for bounded variables (primal-dual on the central path)
    sig = np.zeros(n)
    b = nlp.bounds
    P = Xf[: nlp.iDt].reshape(nlp.np, 7)
    for col, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
        dl, du = np.maximum(P[:, col] - b[2 * j], 1e-3), np.maximum(b[2 * j + 1] - P[:, col], 1e-3)
        sig[7 * np.arange(nlp.np) + col] = mu / dl ** 2 + mu / du ** 2
    sig[nlp.sO:] = mu / np.maximum(Xf[nlp.sO:], 1e-3) ** 2
    K, bw = ce.kkt(nlp, opt, sel, Xf, nu, sig=sig)
    nt = n + m
    live = np.where(np.abs(K).sum(1) > 0)[0]
    # groups
    grp = -np.ones(nt, int)  # 2 i = separator i, 2 i + 1 = interior i; separator N = 2 N
    for i in range(N):
        for k in range(6):
            pt = 6 * i + k
            grp[7 * pt: 7 * pt + 7] = 2 * i if k == 0 else 2 * i + 1
            grp[n + nlp.rO + 5 * pt: n + nlp.rO + 5 * pt + 5] = 2 * i + 1
        if i > 0: grp[n + nlp.rC + 7 * (i - 1): n + nlp.rC + 7 * i] = 2 * i
    grp[n: n + 7] = 0  # initial rows
    grp[n + nlp.rF: n + nlp.rF + 5] = 2 * N  # terminal rows
    # tube slacks and rows: checkpoint T sits at a point; find it through the coupling in K
    for q in list(range(nlp.sT, nlp.sP)) + list(range(n + nlp.rT, n + nlp.rF)):
        if q in live:
            nb = [j for j in np.nonzero(K[q])[0] if grp[j] >= 0]
            grp[q] = max(grp[j] for j in nb) if nb else -1
    for _ in range(2):
        for q in list(range(nlp.sT, nlp.sP)):
            if q in live and grp[q] < 0:
                nb = [j for j in np.nonzero(K[q])[0] if grp[j] >= 0]
                grp[q] = max(grp[j] for j in nb) if nb else -1
    dtc = nlp.iDt
    grp[dtc] = -2  # border
    und = [q for q in live if grp[q] == -1]
    assert not und, und[:10]
    # check coupling pattern
    bad = 0
    for q in live:
        for j in np.nonzero(K[q])[0]:
            gq, gj = grp[q], grp[j]
            if gq < 0 or gj < 0: continue
            if abs(gq - gj) > 2 or (gq % 2 == 1 and gj % 2 == 1 and gq != gj): bad += 1
    rhs = rng.standard_normal(nt); rhs[[q for q in range(nt) if q not in set(live)]] = 0.0
    ref = np.zeros(nt); ref[live] = np.linalg.solve(K[np.ix_(live, live)], rhs[live])
    # structured: eliminate interiors (odd groups) -> Schur on separators + dt
    sepidx = [q for q in live if grp[q] % 2 == 0 and grp[q] >= 0] + [dtc]
    S = K[np.ix_(sepidx, sepidx)].copy(); rs = rhs[sepidx].copy()
    pos = {q: k for k, q in enumerate(sepidx)}
    conds, facs = [], []
    for i in range(N):
        I = [q for q in live if grp[q] == 2 * i + 1]
        KII = K[np.ix_(I, I)]
        conds.append(np.linalg.cond(KII))
        nbr = [q for q in sepidx if np.abs(K[np.ix_(I, [q])]).sum() > 0]
        KIS = K[np.ix_(I, nbr)]
        lu = sla.lu_factor(KII)
        W = sla.lu_solve(lu, np.c_[KIS, rhs[I]])
        ix = [pos[q] for q in nbr]
        S[np.ix_(ix, ix)] -= KIS.T @ W[:, :-1]
        rs[ix] -= KIS.T @ W[:, -1]
        facs.append((I, nbr, lu, KIS))
    ys = np.linalg.solve(S, rs)
    sol = np.zeros(nt); sol[sepidx] = ys
    for I, nbr, lu, KIS in facs:
        sol[I] = sla.lu_solve(lu, rhs[I] - KIS @ ys[[pos[q] for q in nbr]])
    err = np.abs(sol - ref).max() / np.abs(ref).max()
    res_s = np.abs(K[np.ix_(live, live)] @ sol[live] - rhs[live]).max(); res_r = np.abs(K[np.ix_(live, live)] @ ref[live] - rhs[live]).max()
    # bandwidth of the separator system in separator order
    print(f"{label}: N {N} live {len(live)} bw {bw} pattern violations {bad}; interior blocks {len(facs[0][0])} unknowns, cond max {max(conds):.2e} median {np.median(conds):.2e}; "
          f"separator system {len(sepidx)} (cond {np.linalg.cond(S):.2e}); rel. difference to the dense solve {err:.2e}; residuals structured {res_s:.2e} dense {res_r:.2e}; cond K {np.linalg.cond(K[np.ix_(live, live)]):.2e}")
study(X0, np.zeros(m), 0.1, "guess, mu 0.1")
rc = ce.solve(nlp, X0, opt)
study(rc["X"], rng.standard_normal(m) * 0.1, 1e-4, "solution, mu 1e-4")
for k in (3, 8, 14):
    rk = ce.solve(nlp, X0, ipm.IpmOptions(max_iter=k, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1))
    study(rk["X"], rng.standard_normal(m), 2e-2, f"iterate {k}, mu 2e-2")
