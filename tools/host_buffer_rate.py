"""`cfz_mpc_solve` with HOST buffers (parameters in, trajectories out over PCIe) against its own kernel time: the PCIe-inclusive rate of the
stepwise boundary.  4096 instances = the 20 MPC goldens tiled.  Run on the GPU box: python tools/host_buffer_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conflict_rez_amd import engine, scenarios
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mpc_golden.npz"))
B = 4096
idx = np.arange(B) % len(d["x0"])
x0, ref, nbr, zu = (np.ascontiguousarray(d[k][idx]) for k in ("x0", "ref", "nbr", "zu"))
e = engine.Engine(scenarios.parking_lot_spec(), max_batch=B)
for rep in range(4):
    t0 = time.perf_counter(); out = e.solve(x0, ref, nbr, zu, want_duals=False); t1 = time.perf_counter()
    nbytes = x0.nbytes + ref.nbytes + nbr.nbytes + zu.nbytes + out["zu"].nbytes
    print(f"call {rep}: {B} solves, wall {1e3 * (t1 - t0):.2f} ms = {B / (t1 - t0):.0f} solves/s; kernel {out['solve_ms']:.2f} ms = {B / out['solve_ms'] * 1e3:.0f} solves/s; "
          f"{nbytes / 1e6:.1f} MB over PCIe; mean iterations {out['iters'].mean():.1f}", flush=True)
