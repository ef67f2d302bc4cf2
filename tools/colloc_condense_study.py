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


def interval_groups(nlp, K, lanes64=False):
    """Group index of every unknown of the assembled matrix K ([x | c] of a single-vehicle CollocNlp): 2 i = separator i (start point of
    interval i, its continuity rows, tube slacks / rows of a checkpoint, initial rows; 2 N: the terminal rows), 2 i + 1 = interior i
    (points 1..5 of interval i and its 30 ODE rows), -2 = dt (border), -1 = eliminated before assembly (zero row).
    lanes64: interiors of exactly 64 unknowns and separators of 14 / 15 / 22 / 31 (see the code) -- in the band ordering of
    cfz_colloc.inl both are still contiguous position ranges."""
    n, N = nlp.n, nlp.N[0]
    nt = n + nlp.m
    live = np.where(np.abs(K).sum(1) > 0)[0]
    liveset = set(live.tolist())
    grp = -np.ones(nt, int)
    for i in range(N):
        for k in range(6):
            pt = 6 * i + k
            grp[7 * pt: 7 * pt + 7] = 2 * i if k == 0 else 2 * i + 1
            grp[n + nlp.rO + 5 * pt: n + nlp.rO + 5 * pt + 5] = 2 * i + 1
        if i > 0:
            grp[n + nlp.rC + 7 * (i - 1): n + nlp.rC + 7 * i] = 2 * i
    grp[n: n + 7] = 0
    grp[n + nlp.rF: n + nlp.rF + 5] = 2 * N
    for _ in range(2):  # tube rows take the group of the point they constrain, their slacks that of their row
        for q in list(range(n + nlp.rT, n + nlp.rF)) + list(range(nlp.sT, nlp.sP)):
            if q in liveset and grp[q] < 0:
                nb = [j for j in np.nonzero(K[q])[0] if grp[j] >= 0]
                grp[q] = max(grp[j] for j in nb) if nb else -1
    if lanes64:  # the partition to build: interiors of exactly 64 unknowns (one row per lane of a wavefront)
        for i in range(N):  # the steering rate of an interval's last point joins the next separator ...
            grp[7 * (6 * i + 5) + 6] = 2 * (i + 1)
        for q in live:  # ... and so do the end point's tube slacks and rows
            if grp[q] == 2 * N - 1 and (nlp.sT <= q < nlp.sP or n + nlp.rT <= q < n + nlp.rF):
                grp[q] = 2 * N
    grp[nlp.iDt] = -2
    grp[[q for q in range(nt) if q not in liveset]] = -1
    return grp, live


def pattern_violations(K, grp, live):
    """Couplings the structured elimination does not allow: between groups more than two apart, or between two different interiors."""
    bad = 0
    for q in live:
        for j in np.nonzero(K[q])[0]:
            gq, gj = grp[q], grp[j]
            if gq < 0 or gj < 0:
                continue
            if abs(gq - gj) > 2 or (gq % 2 == 1 and gj % 2 == 1 and gq != gj):
                bad += 1
    return bad


def structured_solve(K, grp, live, rhs, N, dtc, want_cond=False):
    """Interiors by dense pivoted LU, independently; then the separator system with the dt border (dense here); back-substitution."""
    sepidx = [q for q in live if grp[q] % 2 == 0 and grp[q] >= 0] + [dtc]
    S = K[np.ix_(sepidx, sepidx)].copy(); rs = rhs[sepidx].copy()
    pos = {q: k for k, q in enumerate(sepidx)}
    conds, facs = [], []
    for i in range(N):
        I = [q for q in live if grp[q] == 2 * i + 1]
        KII = K[np.ix_(I, I)]
        if want_cond:
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
    sol = np.zeros(K.shape[0]); sol[sepidx] = ys
    for I, nbr, lu, KIS in facs:
        sol[I] = sla.lu_solve(lu, rhs[I] - KIS @ ys[[pos[q] for q in nbr]])
    return sol, dict(separator_unknowns=len(sepidx), interior_unknowns=len(facs[0][0]), conds=conds, S=S, sepidx=sepidx, rs=rs)


def separator_recursion(S, sepidx, grp, N, rs):
    """The separator system [Sb b; b' h] (dt last) by a block recursion over the N + 1 separators (block tridiagonal Sb, dense pivoted LU
    of the diagonal blocks only), dt as a second right-hand side: eta = (r_dt - b' y1) / (h - b' y2).  Returns (x, largest condition
    number of a diagonal block met on the way)."""
    blocks = [[k for k, q in enumerate(sepidx[:-1]) if grp[q] == 2 * i] for i in range(N + 1)]
    Sb, bcol, h = S[:-1, :-1], S[:-1, -1], S[-1, -1]
    D = [Sb[np.ix_(blocks[i], blocks[i])].copy() for i in range(N + 1)]
    R = [np.c_[rs[:-1][blocks[i]], bcol[blocks[i]]].copy() for i in range(N + 1)]
    cmax = 0.0
    for i in range(N):
        U = Sb[np.ix_(blocks[i], blocks[i + 1])]
        cmax = max(cmax, np.linalg.cond(D[i]))
        W = sla.lu_solve(sla.lu_factor(D[i]), np.c_[U, R[i]])
        D[i + 1] -= U.T @ W[:, : len(blocks[i + 1])]
        R[i + 1] -= U.T @ W[:, len(blocks[i + 1]):]
    Y = [None] * (N + 1)
    Y[N] = np.linalg.solve(D[N], R[N])
    for i in range(N - 1, -1, -1):
        Y[i] = np.linalg.solve(D[i], R[i] - Sb[np.ix_(blocks[i], blocks[i + 1])] @ Y[i + 1])
    order = np.concatenate(blocks)
    y1, y2 = np.concatenate([Y[i][:, 0] for i in range(N + 1)]), np.concatenate([Y[i][:, 1] for i in range(N + 1)])
    eta = (rs[-1] - bcol[order] @ y1) / (h - bcol[order] @ y2)
    x = np.zeros(len(sepidx)); x[order] = y1 - eta * y2; x[-1] = eta
    return x, cmax


def central_sigma(nlp, Xf, mu):
    """Sigma = mu / d^2 of the boxes and of the slacks' lower bound: barrier terms of a point on the central path."""
    sig = np.zeros(nlp.n)
    b = nlp.bounds
    P = Xf[: nlp.iDt].reshape(nlp.np, 7)
    for col, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
        dl, du = np.maximum(P[:, col] - b[2 * j], 1e-3), np.maximum(b[2 * j + 1] - P[:, col], 1e-3)
        sig[7 * np.arange(nlp.np) + col] = mu / dl ** 2 + mu / du ** 2
    sig[nlp.sO:] = mu / np.maximum(Xf[nlp.sO:], 1e-3) ** 2
    return sig


def main():
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
    rng = np.random.default_rng(0)

    def study(X, nu, mu, label):
        Xf = np.zeros(nlp.n); Xf[: len(X)] = X
        sel = ce.select(nlp, opt, Xf)
        Xf[nlp.sO:] = np.maximum(Xf[nlp.sO:], 1e-2)
        K, bw = ce.kkt(nlp, opt, sel, Xf, nu, sig=central_sigma(nlp, Xf, mu))
        grp, live = interval_groups(nlp, K, lanes64=True)
        rhs = rng.standard_normal(K.shape[0]); rhs[grp == -1] = 0.0
        ref = np.zeros(K.shape[0]); ref[live] = np.linalg.solve(K[np.ix_(live, live)], rhs[live])
        sol, info = structured_solve(K, grp, live, rhs, N, nlp.iDt, want_cond=True)
        KL = K[np.ix_(live, live)]
        print(f"{label}: N {N} live {len(live)} bw {bw} pattern violations {pattern_violations(K, grp, live)}; interior blocks {info['interior_unknowns']} unknowns, "
              f"cond max {max(info['conds']):.2e} median {np.median(info['conds']):.2e}; separator system {info['separator_unknowns']} (cond {np.linalg.cond(info['S']):.2e}); "
              f"rel. difference to the dense solve {np.abs(sol - ref).max() / np.abs(ref).max():.2e}; residuals structured {np.abs(KL @ sol[live] - rhs[live]).max():.2e} "
              f"dense {np.abs(KL @ ref[live] - rhs[live]).max():.2e}; cond K {np.linalg.cond(KL):.2e}", flush=True)
        xs, cmax = separator_recursion(info["S"], info["sepidx"], grp, N, info["rs"])
        xd = np.linalg.solve(info["S"], info["rs"])
        print(f"    separator system by block recursion: difference to its dense solve {np.abs(xs - xd).max() / np.abs(xd).max():.2e}, diagonal blocks' cond up to {cmax:.2e}", flush=True)

    study(X0, np.zeros(nlp.m), 0.1, "guess, mu 0.1")
    rc = ce.solve(nlp, X0, opt)
    study(rc["X"], rng.standard_normal(nlp.m) * 0.1, 1e-4, "solution, mu 1e-4")
    for k in (3, 8, 14):
        rk = ce.solve(nlp, X0, ipm.IpmOptions(max_iter=k, reg_dual=1e-7, tol=1e-2, constr_viol_tol=1e-2, mu_init=0.1))
        study(rk["X"], rng.standard_normal(nlp.m), 2e-2, f"iterate {k}, mu 2e-2")


if __name__ == "__main__":
    main()
