#!/usr/bin/env python3
"""Diagnostic: where one solver iteration spends its shader cycles.  Needs the -DCFZ_STAMPS build of the
library (tools/_libcfz_stamps.so: hipcc ... -DCFZ_STAMPS -o tools/_libcfz_stamps.so conflict_rez_amd/csrc/cfz_engine.hip).
Reads the 12 per-instance phase counters the diagnostic kernel leaves behind the stats array."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios  # noqa: E402

engine.load_library(os.path.join(ROOT, "tools", "_libcfz_stamps.so"))
spec = scenarios.parking_lot_spec()
table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(4, table, seed=2024)
x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
names = ["setup", "dynamics", "residuals", "barrier", "assembly", "ric_fwd+co", "step", "linesearch", "update", "ws+rows", "output", "ric_bwd"]
for B in (1, 256, 768):
    eng = engine.Engine(spec, max_batch=max(B, 16))
    rep = lambda a: np.repeat(a[0:1], B, 0)
    out = eng.solve(rep(x0), rep(ref), rep(nbr), rep(zu), want_duals=False)
    # the stamps sit after the B*3 stats doubles in the same device buffer: read them with a raw copy
    hip = C.CDLL("libamdhip64.so")
    buf = np.zeros(B * 12, dtype=np.uint64)
    h = C.cast(eng._h, C.POINTER(C.c_void_p))
    # struct layout is private; use the public stats call to locate nothing -- instead rely on cfz_mpc_stats copy size:
    # simpler: re-run via solve_device with our own buffers
    import torch
    d = lambda a, dt=torch.float64: torch.tensor(a, dtype=dt, device="cuda")
    dx0, dref, dnbr, dzu = d(rep(x0)), d(rep(ref)), d(rep(nbr)), d(rep(zu))
    dst, dit = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    dstats = torch.zeros(B * 15, dtype=torch.float64, device="cuda")
    eng.solve_device(B, dx0, dref, dnbr, dzu, dst, dit, dstats)
    torch.cuda.synchronize()
    import time; time.sleep(0.2)
    st = dstats[B * 3:].cpu().numpy().view(np.uint64).reshape(B, 12)
    it = int(dit[0])
    med = np.median(st, 0)
    print(f"B={B}: iters {it}, cycles per iteration by phase (median over instances):")
    wall_us = med[0] / 100.0
    med = med.copy(); med[0] = 0
    tot = med.sum()
    for n, c in zip(names, med):
        if n != "-":
            per = c / (it if n not in ("setup", "output") else 1)
            print(f"   {n:10s} {per:12.0f}  ({100 * c / tot:4.1f} % of the solve)")
    print(f"   solve wall time {wall_us:.1f} us -> shader clock {tot / wall_us / 1e3:.3f} GHz, {wall_us / it:.1f} us per iteration")
    print(f"   total cycles {tot:.0f}; kernel {eng.last_solve_ms():.3f} ms -> {tot / eng.last_solve_ms() / 1e6:.3f} GHz if the stamps tick at the shader clock")
    eng.close()
