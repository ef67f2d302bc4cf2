#!/usr/bin/env python3
"""Kernel time per interior-point iteration of ONE instance alone on the GPU (no SIMD sharing), product build (no stamps): the latency
that bounds a tail-bound launch.  usage: python tools/lone_latency.py <lib.so> [golden indices...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios
engine._lib = engine.load_library(sys.argv[1])
idx = [int(a) for a in sys.argv[2:]] or [9, 17, 18, 19]
d = np.load(os.path.join(ROOT, "tests", "golden", "mpc_golden.npz"))
e = engine.Engine(scenarios.parking_lot_spec(), max_batch=4)
for b in idx:
    ms = []
    for rep in range(5):
        out = e.solve(d["x0"][b:b+1], d["ref"][b:b+1], d["nbr"][b:b+1], d["zu"][b:b+1], want_duals=False)
        ms.append(out["solve_ms"])
    it = int(out["iters"][0])
    print(f"{os.path.basename(sys.argv[1])}: golden {b}: status {out['status'][0]} iterations {it}, kernel {min(ms)*1e3:.0f} us = {min(ms)*1e3/max(it,1):.1f} us per iteration (setup and output included)")
