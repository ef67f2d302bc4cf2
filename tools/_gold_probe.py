import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from conflict_rez_amd import engine, scenarios
if len(sys.argv) > 1:
    engine._lib = engine.load_library(sys.argv[1])
d = np.load('tests/golden/mpc_golden.npz')
e = engine.Engine(scenarios.parking_lot_spec(), max_batch=64)
out = e.solve(d['x0'], d['ref'], d['nbr'], d['zu'], want_duals=False)
print(sys.argv[1:], 'status', out['status'].tolist(), 'iters', out['iters'].tolist(), 'expected', d['meta'][:, :2].astype(int).T.tolist(), flush=True)
