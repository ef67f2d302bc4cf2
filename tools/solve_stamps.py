#!/usr/bin/env python3
"""Diagnostic (needs tools/_libcfz_stamps.so, -DCFZ_STAMPS): phase cycles per SOLVE in the stepwise closed loop, i.e.
with carried multipliers and the real iteration-count mix, to see what the fixed part of a solve costs."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios  # noqa: E402

lib = engine.load_library(os.path.join(ROOT, "tools", "_libcfz_stamps.so"))
engine._lib = lib
S, K = 1024, 12
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(S, table, seed=2024)
eng = engine.Engine(spec, max_batch=S * 4)
eng.loop_init(table, k0, noise)
names = ["wall(10ns)", "rows+dyn", "residuals", "barrier", "assembly", "costates", "step", "linesearch", "update", "ric_fwd", "output", "ric_bwd"]
for t in range(K):
    eng.loop_step()
B = S * 4
st = np.zeros(B * 12, dtype=np.uint64)
lib.cfz_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
assert lib.cfz_debug_stamps(eng._h, B, st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(B, 12).astype(float)
g = eng.loop_get()
it = g["iters"].ravel()
ok = g["status"].ravel() != 4
print(f"iteration {K}: mean IPM iterations {it[ok].mean():.2f}; per solve (mean over {ok.sum()} solves that ran), shader cycles:")
tot = st[ok][:, 1:].sum(1).mean()
for i, n in enumerate(names):
    if i == 0:
        print(f"   wall time per solve {st[ok][:, 0].mean() / 100:.1f} us")
    else:
        print(f"   {n:12s} {st[ok][:, i].mean():10.0f}  ({100 * st[ok][:, i].mean() / tot:4.1f} %)")
print(f"   total {tot:.0f} cycles")
for k in (0, 1, 2, 3):
    sel = ok & (it == k)
    if sel.sum():
        print(f"   solves with {k} iterations: {sel.sum():5d}, wall {st[sel][:, 0].mean() / 100:.1f} us, cycles {st[sel][:, 1:].sum(1).mean():.0f}")
