"""Feasibility study for the structured elimination of the JOINT plan's Newton system (test infrastructure: uses oracle/ and the CPU build
of the kernel source under tests/; nothing here is product code).  docs/notebook.md, round 5.

The matrix `assemble` fills for V vehicles is, per vehicle, the single-vehicle plan's matrix (interiors of 64 unknowns per Radau interval,
separators between them) plus the condensed pair blocks: 6 x 6 blocks on the poses (x, y, psi) of two vehicles at the same (interval,
point).  Eliminated here as the kernel is meant to:
  0. the tube slacks and rows of a checkpoint condensed into the pose block they touch (exactly as the obstacle rows are);
  1. every vehicle's interior by itself, dense with partial pivoting, right-hand sides = its coupling columns, the right-hand side and the
     15 unit vectors E of its pair-coupled poses (points 1..5);
  2. per interval index the capacitance system  (I + G M) y = E'K^-1(...)  over the <= 60 pair-coupled pose unknowns of the <= 4 vehicles
     (G = blockdiag E'K_a^-1 E, M = the complete condensed pair blocks: diagonal AND off-diagonal parts, so that nothing cancels);
  3. the Schur complements onto the joint separators (<= 4 x 15 unknowns) and a block recursion over them;
  4. back-substitution.
Printed: the difference to a dense solve of the whole system, condition numbers of interiors, capacitance matrices and separator blocks.

    python tools/joint_condense_study.py [n_vehicles=3] [n_sets=4]
"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, scipy.linalg as sla
import colloc_emu_binding as ce
from oracle import ipm


def joint_groups(nlp):
    """(vehicle, kind, index) of every unknown of K ([x | c]): kind 0 = separator i of the vehicle, 1 = interior t, 2 = tube (condensed
    first), -1 = dt / condensed before assembly."""
    n = nlp.n
    nt = n + nlp.m
    veh = -np.ones(nt, int); kind = -np.ones(nt, int); idx = -np.ones(nt, int)
    for a in range(nlp.V):
        for t in range(nlp.N[a]):
            I = int(nlp.off[a]) + t
            for k in range(6):
                pt = 6 * I + k
                sl = slice(7 * pt, 7 * pt + 7)
                veh[sl] = a; kind[sl] = 0 if k == 0 else 1; idx[sl] = t
                r = slice(n + nlp.rO + 5 * pt, n + nlp.rO + 5 * pt + 5)
                veh[r] = a; kind[r] = 1; idx[r] = t
            w5 = 7 * (6 * I + 5) + 6  # the steering rate of the last point joins the next separator
            kind[w5] = 0; idx[w5] = t + 1
            if t > 0:
                r = slice(n + nlp.rC + 7 * (I - a - 1), n + nlp.rC + 7 * (I - a))
                veh[r] = a; kind[r] = 0; idx[r] = t
        r = slice(n + 7 * a, n + 7 * a + 7); veh[r] = a; kind[r] = 0; idx[r] = 0
        r = slice(n + nlp.rF + 5 * a, n + nlp.rF + 5 * a + 5); veh[r] = a; kind[r] = 0; idx[r] = nlp.N[a]
    kind[nlp.sT: nlp.sP] = 2; kind[n + nlp.rT: n + nlp.rF] = 2
    return veh, kind, idx


def structured_joint_solve(nlp, K, Kown, rhs, report=None):
    """K: the assembled matrix with the pair blocks, Kown: without them (same pattern otherwise); both [x | c] dense."""
    n = nlp.n
    veh, kind, idx = joint_groups(nlp)
    live = np.where(np.abs(K).sum(1) > 0)[0]
    live = live[live != nlp.iDt]
    lv = np.zeros(K.shape[0], bool); lv[live] = True
    # 0. tube slacks and rows condensed (an exact Schur complement; in the kernel a closed form per row)
    T = np.where(lv & (kind == 2))[0]
    R = np.where(lv & (kind != 2))[0]
    def condense(Kx, b=None):
        X = np.linalg.solve(Kx[np.ix_(T, T)], np.c_[Kx[np.ix_(T, R)], b[T]] if b is not None else Kx[np.ix_(T, R)])
        Kc = Kx[np.ix_(R, R)] - Kx[np.ix_(R, T)] @ X[:, : len(R)]
        return (Kc, b[R] - Kx[np.ix_(R, T)] @ X[:, -1]) if b is not None else Kc
    K1, r1 = condense(K, rhs)
    K1own = condense(Kown)
    Mp = K1 - K1own  # the condensed pair blocks, complete
    pos = {q: k for k, q in enumerate(R)}
    vehR, kindR, idxR = veh[R], kind[R], idx[R]
    Nmax = max(nlp.N)
    sep = [np.where((kindR == 0) & (idxR == i))[0] for i in range(Nmax + 1)]
    S = {}
    sol = np.zeros(len(R))
    allsep = np.concatenate(sep)
    spos = {q: k for k, q in enumerate(allsep)}
    Sfull = K1[np.ix_(allsep, allsep)].copy(); rs = r1[allsep].copy()
    keep = []
    conds = dict(interior=[], cap=[], capdim=[])
    for t in range(Nmax):
        vs = [a for a in range(nlp.V) if nlp.N[a] > t]
        Ia = [np.where((kindR == 1) & (idxR == t) & (vehR == a))[0] for a in vs]
        assert all(len(I) == 64 for I in Ia)
        # pair-coupled poses of the interiors: x, y, psi of points 1..5
        Ea = []
        for a, I in zip(vs, Ia):
            base = int(nlp.off[a]) + t
            e = [pos[7 * (6 * base + k) + c] for k in range(1, 6) for c in range(3)]
            Ea.append(np.array([np.where(I == q)[0][0] for q in e]))
        E_glob = np.concatenate([I[e] for I, e in zip(Ia, Ea)])
        M = Mp[np.ix_(E_glob, E_glob)]
        # nothing of the pair blocks may lie outside (poses of interior points x poses of interior points of the same interval index)
        Iall = np.concatenate(Ia)
        rest = Mp[Iall].copy(); rest[:, E_glob] = 0.0
        assert np.abs(rest).max() == 0.0
        ny = len(E_glob)
        G = np.zeros((ny, ny)); Yh = []; Ws = []; Cs = []; nbrs = []
        o = 0
        for a, I, e in zip(vs, Ia, Ea):
            KII = K1own[np.ix_(I, I)]
            conds["interior"].append(np.linalg.cond(KII))
            nbr = np.array([q for q in allsep if np.abs(K1own[np.ix_(I, [q])]).sum() > 0])
            assert len(nbr) <= 15 and all(vehR[q] == a for q in nbr)
            C = K1own[np.ix_(I, nbr)]
            Eu = np.zeros((64, 15)); Eu[e, np.arange(15)] = 1.0
            W = sla.lu_solve(sla.lu_factor(KII), np.c_[C, r1[I], Eu])
            nc = len(nbr)
            G[o: o + 15, o: o + 15] = W[e, nc + 1:]
            Yh.append(W[e, : nc + 1]); Ws.append(W); Cs.append(C); nbrs.append(nbr)
            o += 15
        ncs = [len(x) for x in nbrs]
        Yhat = np.zeros((ny, sum(ncs) + 1)); o = 0; oc = 0
        for j, yh in enumerate(Yh):
            Yhat[o: o + 15, oc: oc + ncs[j]] = yh[:, :-1]; Yhat[o: o + 15, -1] = yh[:, -1]
            o += 15; oc += ncs[j]
        Cap = np.eye(ny) + G @ M
        conds["cap"].append(np.linalg.cond(Cap)); conds["capdim"].append(ny)
        Y = sla.lu_solve(sla.lu_factor(Cap), Yhat)
        Z = M @ Y
        allnbr = np.concatenate(nbrs)
        ix_all = [spos[q] for q in allnbr]
        o = 0; oc = 0
        for j, (a, I, e) in enumerate(zip(vs, Ia, Ea)):
            W, C, nbr = Ws[j], Cs[j], nbrs[j]
            nc = ncs[j]
            ix = [spos[q] for q in nbr]
            Sfull[np.ix_(ix, ix)] -= C.T @ W[:, :nc]
            rs[ix] -= C.T @ W[:, nc]
            CWE = C.T @ W[:, nc + 1:]  # = (E'K^-1 C)' for a symmetric K
            Sfull[np.ix_(ix, ix_all)] += CWE @ Z[o: o + 15, :-1]
            rs[ix] += CWE @ Z[o: o + 15, -1]
            if report is not None:
                report.append(np.abs(CWE - W[e, :nc].T).max() / max(np.abs(CWE).max(), 1e-300))
            o += 15; oc += nc
        keep.append((vs, Ia, Ea, Ws, nbrs, ncs, Z, allnbr))
    # pattern of the separator system: block tridiagonal over the joint separators
    for i in range(Nmax + 1):
        for j in range(i + 2, Nmax + 1):
            a_, b_ = [spos[q] for q in sep[i]], [spos[q] for q in sep[j]]
            assert np.abs(Sfull[np.ix_(a_, b_)]).max() == 0.0
    conds["sepdim"] = [len(s_) for s_ in sep]
    # 3. block recursion over the joint separators
    blocks = [[spos[q] for q in s_] for s_ in sep]
    D = [Sfull[np.ix_(b_, b_)].copy() for b_ in blocks]
    Rr = [rs[b_].copy() for b_ in blocks]
    # 3'. (round 6 study) the same block tridiagonal system by block CYCLIC REDUCTION: every other block eliminated at once, level by level --
    # what a parallel recursion over all eight wavefronts would do instead of a chain from both ends.  The blocks are eliminated WITHOUT the
    # neighbours' updates the chain gives them first: their conditioning is what decides whether that is safe.
    if report is not None and isinstance(report, list):
        Ul = [Sfull[np.ix_(blocks[i], blocks[i + 1])].copy() for i in range(Nmax)]
        xb, cb = bcr_solve([d_.copy() for d_ in D], Ul, [r_.copy() for r_ in Rr])
        conds["bcr_cond"] = cb
        conds["bcr_x"] = xb
    conds["sep"] = []
    Zs = []
    for i in range(Nmax):
        U = Sfull[np.ix_(blocks[i], blocks[i + 1])]
        conds["sep"].append(np.linalg.cond(D[i]))
        Wb = sla.lu_solve(sla.lu_factor(D[i]), np.c_[U, Rr[i]])
        D[i + 1] -= U.T @ Wb[:, :-1]
        Rr[i + 1] -= U.T @ Wb[:, -1]
        Zs.append(Wb)
    conds["sep"].append(np.linalg.cond(D[Nmax]))
    ys = [None] * (Nmax + 1)
    ys[Nmax] = np.linalg.solve(D[Nmax], Rr[Nmax])
    for i in range(Nmax - 1, -1, -1):
        ys[i] = Zs[i][:, -1] - Zs[i][:, :-1] @ ys[i + 1]
    ysep = np.zeros(len(allsep))
    for i in range(Nmax + 1):
        ysep[blocks[i]] = ys[i]
    sol[allsep] = ysep
    # 4. interiors back
    for (vs, Ia, Ea, Ws, nbrs, ncs, Z, allnbr) in keep:
        s_all = ysep[[spos[q] for q in allnbr]]
        z = Z[:, -1] - Z[:, :-1] @ s_all
        o = 0
        for j, I in enumerate(Ia):
            nc = ncs[j]
            s_a = ysep[[spos[q] for q in nbrs[j]]]
            sol[I] = Ws[j][:, nc] - Ws[j][:, :nc] @ s_a - Ws[j][:, nc + 1:] @ z[o: o + 15]
            o += 15
    ref = np.linalg.solve(K1, r1)
    if "bcr_x" in conds:  # the separators' solution by cyclic reduction against the chain's and the dense one
        xb = np.concatenate([conds["bcr_x"][i] for i in range(Nmax + 1)])
        order = np.concatenate(blocks)
        conds["bcr_vs_chain"] = float(np.abs(xb - ysep[order]).max() / np.abs(ysep).max())
        conds["bcr_vs_dense"] = float(np.abs(xb - ref[allsep][order]).max() / np.abs(ref[allsep]).max())
        conds["chain_vs_dense"] = float(np.abs(ysep - ref[allsep]).max() / np.abs(ref[allsep]).max())
    return sol, ref, conds, (K1, r1)


def bcr_solve(D, U, r):
    """Block tridiagonal [D_i, U_i; U_i', D_{i+1}] x = r by block cyclic reduction (dense LU with partial pivoting inside a block, no
    pivoting across blocks -- like the chain).  -> (x per block, condition numbers of every block at the moment it is eliminated)."""
    n = len(D)
    alive = list(range(n))
    links = {i: U[i] for i in range(n - 1)}  # links[l] couples block l with the next ALIVE block to its right
    elim = []  # (j, left, right, G C_l', G C_j, g): what the back-substitution needs
    conds = []
    while len(alive) > 1:
        keep, gone = alive[0::2], alive[1::2]
        pos = {b: k for k, b in enumerate(alive)}
        newlinks = {}
        for j in gone:
            k = pos[j]
            l = alive[k - 1]
            rr = alive[k + 1] if k + 1 < len(alive) else None
            conds.append(np.linalg.cond(D[j]))
            lu = sla.lu_factor(D[j])
            Cl = links[l]                      # rows: block l, columns: block j
            cols = [Cl.T, r[j][:, None]] + ([links[j]] if rr is not None else [])
            X = sla.lu_solve(lu, np.hstack(cols))
            nl = Cl.shape[0]
            GCl, g = X[:, :nl], X[:, nl]
            D[l] -= Cl @ GCl
            r[l] -= Cl @ g
            if rr is not None:
                Cj = links[j]
                GCj = X[:, nl + 1:]
                D[rr] -= Cj.T @ GCj
                r[rr] -= Cj.T @ g
                newlinks[l] = -Cl @ GCj      # block l against block rr
                # (the lower coupling is its transpose: G is symmetric up to rounding)
            else:
                GCj = None
            elim.append((j, l, rr, GCl, GCj, g))
        for kk in range(len(keep) - 1):
            if keep[kk] not in newlinks:
                raise AssertionError("link lost")
        links = newlinks
        alive = keep
    x = {alive[0]: np.linalg.solve(D[alive[0]], r[alive[0]])}
    conds.append(np.linalg.cond(D[alive[0]]))
    for j, l, rr, GCl, GCj, g in reversed(elim):
        xj = g - GCl @ x[l]
        if rr is not None:
            xj = xj - GCj @ x[rr]
        x[j] = xj
    return x, conds


def central_sigma(nlp, Xf, mu):
    sig = np.zeros(nlp.n)
    b = nlp.bounds
    P = Xf[: nlp.iDt].reshape(nlp.np, 7)
    for col, j in ((0, 0), (1, 1), (3, 2), (4, 3), (5, 4), (6, 5)):
        dl, du = np.maximum(P[:, col] - b[2 * j], 1e-3), np.maximum(b[2 * j + 1] - P[:, col], 1e-3)
        sig[7 * np.arange(nlp.np) + col] = mu / dl ** 2 + mu / du ** 2
    sig[nlp.sO:] = mu / np.maximum(Xf[nlp.sO:], 1e-3) ** 2
    return sig


def matrices(nlp, opt, X, nu, mu, active_pairs=0.0, rng=None):
    """(K, Kown): the assembled matrix at X, nu with barrier terms of mu, and the same without the pair blocks (pair slacks' Sigma = 0
    and pair multipliers = 0: D = 1e-8, i.e. nothing).  active_pairs: fraction of pair rows given a tiny slack (D ~ 1 / delta_c)."""
    Xf = np.zeros(nlp.n); Xf[: len(X)] = X
    sel = ce.select(nlp, opt, Xf)
    c0 = nlp.cons(Xf, sel)
    Xf[nlp.sO:] = np.maximum(Xf[nlp.sO:], 1e-2)
    sig = central_sigma(nlp, Xf, mu)
    if active_pairs > 0.0:
        pick = rng.random(2 * nlp.npp) < active_pairs
        sig[nlp.sP:][pick] = 1e9
    K, bw = ce.kkt(nlp, opt, sel, Xf, nu, sig=sig)
    sig0 = sig.copy(); sig0[nlp.sP:] = 0.0
    nu0 = nu.copy(); nu0[nlp.rP:] = 0.0
    Kown, _ = ce.kkt(nlp, opt, sel, Xf, nu0, sig=sig0)
    return K, Kown


def study_problem(nv=3, ns=4):
    """(joint NLP, options, guess along the vehicles' spline paths) of nv vehicles on their first ns (+ 1 for every other vehicle) strategy
    steps: plans of different lengths, as tests/test_colloc.py builds them."""
    import test_colloc as tc
    plans = tc.plans.__wrapped__() if hasattr(tc.plans, "__wrapped__") else None
    if plans is None:
        from conflict_rez_amd import strategy as strat
        from conflict_rez_amd.control.compute_sets import compute_sets, interp_along_sets
        from conflict_rez_amd.vehicle_types import VehicleBody
        hist = strat.generate_strategy(4)
        with tempfile.TemporaryDirectory() as d:
            fn = os.path.join(d, "4v_rl_traj"); strat.write_strategy(fn, hist)
            tubes, paths = compute_sets(fn), interp_along_sets(fn, VehicleBody(), 30)
        plans = {a: ([dict(front=(s["front"].A, s["front"].b), back=(s["back"].A, s["back"].b)) for s in tubes[a]], paths[a]) for a in sorted(hist)}
    agents = ["vehicle_%d" % i for i in range(4)][4 - nv:]
    nsets = [ns + (i % 2) for i in range(nv)]  # plans of different lengths
    jn, sp = tc._joint_problem(plans, agents, nsets, nps=5)
    opt = ipm.IpmOptions(**tc.COLLOC_OPT)
    # a guess: every vehicle along its spline path
    from scipy.interpolate import interp1d
    zs = []
    for a, ns_a in zip(agents, nsets):
        tube, p = plans[a]
        z = tc.warm_start(tube[:ns_a], p[: 30 * (ns_a - 1) + 1], None)
        zs.append(z)
    sing = []
    for j, (a, ns_a) in enumerate(zip(agents, nsets)):
        N = jn.N[j]
        t_i = np.concatenate([k + jn.tau for k in range(N)]) / N * zs[j]["t"][-1]
        sing.append({k: interp1d(zs[j]["t"], zs[j][k])(t_i) for k in ("x", "y", "psi", "v", "delta", "a", "w")})
    X0 = jn.pack(sing, float(np.mean([z["t"][-1] / N for z, N in zip(zs, jn.N)])))
    return jn, opt, X0


def main():
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    jn, opt, X0 = study_problem(nv, ns)
    rng = np.random.default_rng(0)
    for label, X, nu, mu, act in (("guess, mu 0.1", X0, np.zeros(jn.m), 0.1, 0.0),
                                  ("guess, random multipliers, 10 % of the pair rows active", X0, rng.standard_normal(jn.m) * 0.3, 1e-3, 0.1),
                                  ("guess, random multipliers, 50 % of the pair rows active", X0, rng.standard_normal(jn.m) * 0.3, 1e-4, 0.5)):
        K, Kown = matrices(jn, opt, X, nu, mu, act, rng)
        rhs = rng.standard_normal(K.shape[0])
        rep = []
        sol, ref, conds, (K1, r1) = structured_joint_solve(jn, K, Kown, rhs, rep)
        if "bcr_vs_dense" in conds:
            print(f"   separators by cyclic reduction: blocks' condition at elimination max {max(conds['bcr_cond']):.1e} (chain: {max(conds['sep']):.1e}); "
                  f"solution against the chain's {conds['bcr_vs_chain']:.1e}, against the dense solve {conds['bcr_vs_dense']:.1e} (chain against dense {conds['chain_vs_dense']:.1e})")
        print(f"{label}: unknowns {len(ref)}; interiors cond max {max(conds['interior']):.1e}; capacitance dims {sorted(set(conds['capdim']))} cond max {max(conds['cap']):.1e}; "
              f"separator dims {sorted(set(conds['sepdim']))} cond max {max(conds['sep']):.1e}; cond K {np.linalg.cond(K1):.1e}; "
              f"difference to the dense solve {np.abs(sol - ref).max() / np.abs(ref).max():.1e}; residual structured {np.abs(K1 @ sol - r1).max():.1e} dense {np.abs(K1 @ ref - r1).max():.1e}; "
              f"C'K^-1E against (E'K^-1C)' {max(rep):.1e}", flush=True)


if __name__ == "__main__":
    main()
