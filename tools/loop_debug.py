"""Diagnostic driver for cfz_loop_run: small cases first, watchdog on (CFZ_LOOP_WATCHDOG=seconds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios

S, K = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    engine.load_library(os.path.abspath(sys.argv[3]))
    engine._lib = engine.load_library(os.path.abspath(sys.argv[3]))
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(S, table, seed=11)
eng = engine.Engine(spec, max_batch=S * 4)
eng.loop_init(table, k0, noise)
t0 = time.time()
n = eng.loop_run(K)
print("S", S, "K", K, "iterations", n, "wall", time.time() - t0, "kernel ms", eng.last_solve_ms(), flush=True)
a = eng.loop_get()
eng.loop_init(table, k0, noise)
for _ in range(K):
    eng.loop_step()
b = eng.loop_get()
for key in ("state", "pred", "status", "iters"):
    print(key, "equal" if np.array_equal(a[key], b[key]) else "DIFFERENT", flush=True)
print("status run ", np.bincount(a["status"].ravel(), minlength=5), "step", np.bincount(b["status"].ravel(), minlength=5))
print("iters run", a["iters"].ravel()[:8], "step", b["iters"].ravel()[:8])
print("max |dstate|", np.nanmax(np.abs(a["state"] - b["state"])), "max |dpred|", np.nanmax(np.abs(a["pred"] - b["pred"])))
print("state run", a["state"].reshape(-1, 5)[:2], "step", b["state"].reshape(-1, 5)[:2])
