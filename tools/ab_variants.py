#!/usr/bin/env python3
"""A/B of library builds in one process on one GPU (interleaved rounds): uniform batch and closed-loop step."""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ctypes as C
from conflict_rez_amd import engine, scenarios
spec = scenarios.parking_lot_spec(); table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(1024, table, seed=2024)
libs = sorted(glob.glob(os.path.join(ROOT, "tools", "_lib_*.so")))
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        engine._lib = None; engine.load_library(l)
        eng = engine.Engine(spec, max_batch=4096)
        eng.loop_init(table, k0, noise)
        for _ in range(4): eng.loop_step()
        ms = []
        for _ in range(8): eng.loop_step(); ms.append(eng.last_solve_ms())
        res[l].append(np.mean(ms)); info = eng.kernel_info(); eng.close()
for l in libs: print(os.path.basename(l), "kernel ms/step: median %.2f min %.2f" % (np.median(res[l]), np.min(res[l])), "info", info)
