#!/usr/bin/env python3
"""Diagnostic: phase cycles of ONE instance alone on the GPU (no SIMD sharing): the latency that bounds a tail-bound launch.
usage: python tools/lone_stamps.py <stamps-lib.so> [golden index]
A stamp accumulates the cycles since the stamp before it: "rows:ws" is the LAST pass of the working-set loop ("ws:pass1", "ws:pass2" the ones
before it, "iter:head" what precedes the loop), "rows+dyn" what is left of that phase after "rows:rk4" (its reduction); "resid:rows",
"asm:rows", "step:rows", "upd:rows" the block loops of their phases, the phase's own name the rest of it."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios
lib = engine.load_library(sys.argv[1]); engine._lib = lib
b = int(sys.argv[2]) if len(sys.argv) > 2 else 17
d = np.load(os.path.join(ROOT, "tests", "golden", "mpc_golden.npz"))
e = engine.Engine(scenarios.parking_lot_spec(), max_batch=4)
for rep in range(3):
    out = e.solve(d["x0"][b:b+1], d["ref"][b:b+1], d["nbr"][b:b+1], d["zu"][b:b+1], want_duals=False)
st = np.zeros(24, dtype=np.uint64)
lib.cfz_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
assert lib.cfz_debug_stamps(e._h, 1, st.ctypes.data_as(C.c_void_p)) == 0
names = ["wall(10ns)", "rows+dyn", "residuals", "barrier", "assembly", "costates", "step", "linesearch", "update", "ric_fwd", "output", "ric_bwd", "rows:ws", "rows:rk4", "resid:rows", "asm:rows", "step:rows", "upd:rows", "iter:head", "-", "ws:pass1", "ws:pass2"] + ["-"] * 2
it = int(out["iters"][0])
print(f"{sys.argv[1]}: golden {b}: status {out['status'][0]} iterations {it}, kernel {out['solve_ms']*1e3:.0f} us = {out['solve_ms']*1e3/max(it,1):.1f} us per iteration")
print("   cycles per iteration: " + ", ".join(f"{n} {st[i]/max(it,1):.0f}" for i, n in enumerate(names) if i and n != "-"))
