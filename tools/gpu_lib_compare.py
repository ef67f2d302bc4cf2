#!/usr/bin/env python3
"""Two builds of the library on the same seeded MPC batch (cold solves and a short closed loop): are the results equal bit for bit?
usage: python tools/gpu_lib_compare.py <libA.so> <libB.so> [scenarios]     (each library runs in a process of its own)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from conflict_rez_amd import engine, scenarios
    lib, S, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    engine._lib = engine.load_library(lib)
    spec = scenarios.parking_lot_spec()
    table, _ = scenarios.load_reference_table(kind="planned")
    k0, noise = scenarios.sample_scenarios(S, table, seed=11)  # raw draws: restorations, status 4 / 5 among them
    x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
    e = engine.Engine(spec, max_batch=len(x0))
    r = e.solve(x0, ref, nbr, zu, want_duals=False)
    e.loop_init(table, k0, noise)
    its = e.loop_run(6)
    g = e.loop_get()
    np.savez(out, status=r["status"], iters=r["iters"], zu=r["zu"], ms=r["solve_ms"], loop_status=g["status"], loop_iters=g["iters"], loop_pred=g["pred"], loop_state=g["state"], its=its)
    sys.exit(0)
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
res = []
with tempfile.TemporaryDirectory() as d:
    for i, lib in enumerate(sys.argv[1:3]):
        out = os.path.join(d, f"r{i}.npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib), str(S), out])
        res.append(dict(np.load(out)))
a, b = res
print(f"{4 * S} cold solves: iterations {int(a['iters'].sum())} / {int(b['iters'].sum())}, kernel {float(a['ms']):.2f} / {float(b['ms']):.2f} ms")
for k in ("status", "iters", "zu", "loop_status", "loop_iters", "loop_pred", "loop_state"):
    same = np.array_equal(a[k], b[k])
    print(f"  {k}: {'equal bit for bit' if same else 'DIFFERENT: %d entries, max %.3e' % (int((a[k] != b[k]).sum()), float(np.abs(a[k].astype(float) - b[k].astype(float)).max()))}")
print("closed loop iterations", int(a["its"]), int(b["its"]))
