#!/usr/bin/env python3
"""Diagnostic (needs tools/_libcfz_stamps.so): distribution of per-instance solve time in the headline batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from conflict_rez_amd import engine, scenarios
import torch
engine.load_library(os.path.join(ROOT, "tools", "_libcfz_stamps.so"))
spec = scenarios.parking_lot_spec(); table, _ = scenarios.load_reference_table()
k0, noise = scenarios.sample_scenarios(1024, table, seed=2024)
x0, ref, nbr, zu = scenarios.mpc_batch_from_table(spec, table, k0, noise)
B = len(x0)
eng = engine.Engine(spec, max_batch=B)
d = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")
dx0, dref, dnbr, dzu = d(x0), d(ref), d(nbr), d(zu)
dst = torch.zeros(B, dtype=torch.int32, device="cuda"); dit = torch.zeros(B, dtype=torch.int32, device="cuda")
dstats = torch.zeros(B * 15, dtype=torch.float64, device="cuda")
for _ in range(2):
    dzu2 = dzu.clone(); eng.solve_device(B, dx0, dref, dnbr, dzu2, dst, dit, dstats); torch.cuda.synchronize()
print("kernel ms", eng.last_solve_ms() if False else "n/a (device call)")
st = dstats[B * 3:].cpu().numpy().view(np.uint64).reshape(B, 12).astype(np.float64)
tot = st.sum(1); it = dit.cpu().numpy(); status = dst.cpu().numpy()
print("instances", B, "sum cycles %.3e" % tot.sum(), "mean %.3e" % tot.mean(), "p50 %.3e p90 %.3e p99 %.3e max %.3e" % (*np.percentile(tot, [50, 90, 99]), tot.max()))
print("ideal time at 512 slots, 2.4 GHz: %.2f ms; longest single instance: %.2f ms" % (tot.sum() / 512 / 2.4e6, tot.max() / 2.4e6))
w = np.argsort(-tot)[:8]
for b in w: print("  instance", b, "status", status[b], "iters", it[b], "cycles %.2e" % tot[b], "linesearch share %.0f %%" % (100 * st[b, 7] / tot[b]))
print("line-search share overall %.1f %%" % (100 * st[:, 7].sum() / tot.sum()))
