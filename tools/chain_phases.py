#!/usr/bin/env python3
"""Where the solves ON THE CRITICAL PATH of the bench's launch spend their cycles.  The bench's workload (planned table, feasible starts,
seed 2024 by default, 1024 scenarios, 5 warm-up + 20 timed MPC iterations) run stepwise on a -DCFZ_STAMPS build: after every
`cfz_loop_step` the phase cycles of every instance are read (`cfz_debug_stamps`); per scenario and iteration the vehicle with the most
interior-point iterations is on that scenario's chain; the scenario with the longest chain is the launch's critical path.  Cycles are
those of a loaded GPU (4096 instances per step), the SHARES are what matters.
    python tools/chain_phases.py <stamps-lib.so> [seed=2024] [scenarios=1024]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios
lib = engine.load_library(sys.argv[1]); engine._lib = lib
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table(kind="planned")
k0, noise = scenarios.sample_scenarios(S, table, seed=seed, spec=spec)
V, B = 4, S * 4
e = engine.Engine(spec, max_batch=B)
e.loop_init(table, k0, noise)
lib.cfz_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
names = ["wall(10ns)", "rows:reduce", "residuals", "barrier", "assembly", "costates", "step", "linesearch", "update", "ric_fwd", "output", "ric_bwd", "rows:lastpass", "rows:rk4",
         "resid:rows", "asm:rows", "step:rows", "upd:rows", "iter:head", "-", "ws:pass1", "ws:pass2", "-", "-"]
W, K = 5, 20
chain_it = np.zeros(S, int); chain_ph = np.zeros((S, 24)); chain_log = [[] for _ in range(S)]
for t in range(W + K):
    e.loop_step()
    g = e.loop_get()
    it, stt = g["iters"].reshape(S, V), g["status"].reshape(S, V)
    st = np.zeros((B, 24), dtype=np.uint64)
    assert lib.cfz_debug_stamps(e._h, B, st.ctypes.data_as(C.c_void_p)) == 0
    st = st.reshape(S, V, 24).astype(float)
    if t < W:
        continue
    v = it.argmax(1)
    ar = np.arange(S)
    chain_it += it[ar, v]; chain_ph += st[ar, v]
    for s in range(S):
        chain_log[s].append((int(it[s, v[s]]), int(stt[s, v[s]])))
order = np.argsort(-chain_it)
print(f"seed {seed}: longest chains {chain_it[order[:5]].tolist()} (scenarios {order[:5].tolist()}), 99th percentile {int(np.percentile(chain_it, 99))}, mean {chain_it.mean():.1f}")
for s in order[:3]:
    ph = chain_ph[s]; tot = ph[1:].sum()
    print(f"scenario {s}: chain {chain_it[s]} iterations; (iterations, status) of its critical solves: {chain_log[s]}")
    print("   share of the chain's cycles: " + ", ".join(f"{n} {100 * ph[i] / tot:.1f}%" for i, n in enumerate(names) if i and n != "-" and ph[i] > 0.004 * tot))
    print(f"   line-search cycles per iteration {ph[7] / chain_it[s]:.0f}, all phases per iteration {tot / chain_it[s]:.0f}")
allp = chain_ph.sum(0); tot = allp[1:].sum()
print("all chains together: " + ", ".join(f"{n} {100 * allp[i] / tot:.1f}%" for i, n in enumerate(names) if i and n != "-" and allp[i] > 0.004 * tot))
