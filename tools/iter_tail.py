"""Closed-loop iteration-count tail: per MPC iteration, the largest IPM iteration counts and how those solves ended."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios

S, K = int(sys.argv[1]), int(sys.argv[2])
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(S, table, seed=2024)
eng = engine.Engine(spec, max_batch=S * 4)
eng.loop_init(table, k0, noise)
hist = np.zeros(700, int)
tot_it = np.zeros(S * 4, int)
fails = np.zeros(S * 4, int)
quiet = len(sys.argv) > 3
by_status = {}
prev_hard = None
for t in range(K):
    eng.loop_step()
    g = eng.loop_get()
    it, st = g["iters"].ravel(), g["status"].ravel()
    hist += np.bincount(it, minlength=700)[:700]
    top = np.argsort(-it)[:6]
    hard = set(np.nonzero(it > 25)[0] // 4)
    rep = len(hard & prev_hard) if prev_hard is not None else 0
    prev_hard = hard
    if t >= 25:
        tot_it += it
        fails += st != 0
    if not quiet:
      print(f"t={t:3d} ms={eng.last_solve_ms():6.2f} mean={it.mean():5.2f} top iters={it[top]} status={st[top]} inst={top} n>25={int((it > 25).sum())} repeat-scen={rep}", flush=True)
    for s_ in range(5):
        by_status.setdefault(s_, []).extend(it[st == s_])
top = np.argsort(-tot_it)[:10]
print('largest iteration totals over t>=25:', [(int(b), int(tot_it[b]), int(fails[b])) for b in top], 'mean total', tot_it.mean())
per_scen = tot_it.reshape(S, 4).max(1)
print('scenario critical path (sum over t of max over vehicles is >= max vehicle total): top', np.sort(per_scen)[-8:], 'median', np.median(per_scen))
c = np.cumsum(hist[::-1])[::-1]
print("P(iters>=k):", {k: round(c[k] / c[0], 5) for k in (5, 10, 15, 20, 30, 40, 60, 100)})
for s_, v in by_status.items():
    if len(v):
        v = np.array(v)
        print("status", s_, "n", len(v), "mean it", v.mean().round(2), "max", v.max(), "sum share", (v.sum() / max(1, sum(np.sum(x) for x in by_status.values()))).round(3))
