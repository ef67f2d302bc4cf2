"""cfz_state_ws against the CPU build of the same source after k interior-point iterations from the same guess (diagnostic for the
matrix-core sweep of cfz_plan.inl): largest difference of the iterates.   python tools/state_ws_step_check.py [k ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_colloc as tc
import plan_emu_binding as pe
from conflict_rez_amd import engine
from oracle import ipm
from oracle.plan_nlp import StateWsNlp, speed_guess

plans = tc.plans.__wrapped__()
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 5, 8]:
    for a in sorted(plans):
        tube, p = plans[a]; fh = float(p[-1, 2])
        ws = StateWsNlp(p[0], tube, final_heading=fh, shrink_tube=0.5)
        rc = pe.solve(ws, ws.pack(p[:, 0], p[:, 1], p[:, 2], v=speed_guess(p, ws.dt)), ipm.IpmOptions(max_iter=k, hessian="exact", reg_dual=1e-9, stall_iters=0, mu_init=0.1))
        zc = ws.unpack(rc["X"])
        tb = [((s["back"][0], s["back"][1]), (s["front"][0], s["front"][1])) for s in tube[1:]]
        rg = engine.state_ws([p[0]], [tb], [p], [fh], shrink_tube=0.5, max_iter=k)[0]
        want = np.stack([zc[c] for c in ("x", "y", "psi", "v", "delta", "a", "w")], 1)
        print(f"{k} iterations, {a}: status gpu {rg['status']} cpu {rc['status']}, iterations {rg['iters']} / {rc['iters']}, largest difference {np.abs(rg['traj'] - want).max():.3e}", flush=True)
