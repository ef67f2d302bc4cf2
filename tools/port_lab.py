"""Laboratory for the MPC algorithm on the CPU (oracle's C port; test infrastructure, nothing here is product code).

    python tools/port_lab.py pop  [key=value ...]         the four independent-solver populations, production and tight tolerances
    python tools/port_lab.py loop [S=1024] [K=25] [W=5] [seeds=2024,2025,2026] [key=value ...]
                                                          the bench's closed loop (planned table, feasible starts) on the port

key=value pairs override fields of oracle.ipm.IpmOptions.  `loop` prints, per seed: converged solves of the timed iterations, status
counts, mean interior-point iterations per solve, and the chain statistics that bound the persistent GPU launch (per scenario the sum over
the timed MPC iterations of the maximum over its vehicles: max, 99th percentile, mean).
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def parse(argv):
    kv = {}
    for a in argv:
        k, v = a.split("=", 1)
        kv[k] = v
    return kv


def options(kv):
    from oracle import ipm

    opt = ipm.IpmOptions()
    for k, v in kv.items():
        if hasattr(opt, k):
            cur = getattr(opt, k)
            setattr(opt, k, type(cur)(float(v)) if not isinstance(cur, (bool, str)) else (v if isinstance(cur, str) else bool(int(v))))
    return opt


def _loop_worker(args):
    seed, lo, hi, S, W, K, kv = args
    from conflict_rez_amd import scenarios
    from oracle.closed_loop import replay
    from oracle.mpc_nlp import MpcSpec

    opt = options(kv)
    spec = scenarios.parking_lot_spec()
    table, _ = scenarios.load_reference_table(kind="planned")
    ospec = MpcSpec(N=spec.N, dt=spec.dt, A_obs=spec.A_obs, b_obs=spec.b_obs, n_nbr=spec.n_nbr)
    k0, noise = scenarios.sample_scenarios(S, table, seed=seed, spec=spec)
    k0, noise = k0[lo:hi], noise[lo:hi]
    st_all, it_all = [], []
    for t, (_, _, status, iters) in enumerate(replay(ospec, table, k0, noise, W + K, dt=spec.dt, wb=spec.wb, opt=opt)):
        if t >= W:
            st_all.append(status.copy()); it_all.append(iters.copy())
    return np.array(st_all), np.array(it_all)  # [K, s, V]


def loop(kv):
    import concurrent.futures as cf

    S, K, W = int(kv.pop("S", 1024)), int(kv.pop("K", 20)), int(kv.pop("W", 5))
    seeds = [int(s) for s in kv.pop("seeds", "2024,2025,2026").split(",")]
    P = int(kv.pop("procs", 8))
    for seed in seeds:
        t0 = time.time()
        cuts = np.linspace(0, S, P + 1).astype(int)
        with cf.ProcessPoolExecutor(P) as pool:
            res = list(pool.map(_loop_worker, [(seed, cuts[i], cuts[i + 1], S, W, K, kv) for i in range(P)]))
        st = np.concatenate([r[0] for r in res], 1); it = np.concatenate([r[1] for r in res], 1)
        chain = it.max(2).sum(0)
        cnt = {int(s): int((st == s).sum()) for s in np.unique(st)}
        print(f"seed {seed}: solves {st.size} converged {cnt.get(0, 0)} ({cnt.get(0, 0) / st.size:.4f}) status {cnt} mean its {it.mean():.3f} "
              f"chain max {chain.max()} p99 {np.percentile(chain, 99):.0f} p95 {np.percentile(chain, 95):.0f} mean {chain.mean():.1f} "
              f"its(status0) {it[st == 0].mean():.2f} max it {it.max()}  [{time.time() - t0:.0f} s]", flush=True)


FIXTURES = ("mpc_independent.npz", "mpc_independent_more.npz", "mpc_independent_obs.npz", "mpc_independent_turn.npz")
TIGHT_FULL = dict(tol=1e-7, constr_viol_tol=1e-8, compl_inf_tol=1e-8, dual_inf_tol=1e-5)


def pop(kv):
    from oracle import independent_mpc as im
    from oracle import port
    from oracle.mpc_nlp import MpcSpec

    only = kv.pop("only", None)
    modes = [m == "prod" for m in kv.pop("modes", "prod,tight").split(",")]
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for prod in modes:
        opt = options(kv)
        if not prod:
            for k, v in TIGHT_FULL.items():
                setattr(opt, k, v)
            opt.stall_iters = 0
        for fx in FIXTURES:
            if only and only not in fx:
                continue
            d = np.load(os.path.join(gold, fx))
            ospec = MpcSpec(N=30, dt=0.1, A_obs=d["A_obs"], b_obs=d["b_obs"], n_nbr=3)
            rows = []
            for b in range(len(d["x0"])):
                r = port.solve(ospec, d["x0"][b], d["ref"][b], d["nbr"][b], d["zu"][b].T.copy(), opt)
                nlp = im.GeometricMpc(ospec, d["x0"][b], d["ref"][b], d["nbr"][b])
                X = r["p"].ravel()
                gap = (nlp.cost(X) - d["cost"][b]) / d["cost"][b]
                dpose = np.abs(r["p"].T[:3] - d["sol"][b][:3]).max()
                feas = max(np.abs(nlp.eq(X)).max(), -nlp.ineq(X).min())
                rows.append((b, r["status"], r["iters"], gap, dpose, feas))
            gt, pt, ft = (2e-4, 5e-2, 1e-2) if prod else (1e-6, 1e-4, 1e-6)
            ok = [x for x in rows if (x[1] == 0 or (x[1] == 2 and not prod)) and abs(x[3]) < gt and x[4] < pt and x[5] < ft]
            odd = [x for x in rows if x not in ok]
            print(f"{'prod ' if prod else 'tight'} {fx:28s} same optimum {len(ok):2d}/{len(rows)}  iterations {sum(x[2] for x in rows)}  others: "
                  + "; ".join(f"#{b} st {s} it {i} gap {g:+.2e} dpose {dp:.1e} feas {f:.1e}" for b, s, i, g, dp, f in odd), flush=True)


if __name__ == "__main__":
    cmd, kv = sys.argv[1], parse(sys.argv[2:])
    {"pop": pop, "loop": loop}[cmd](kv)
