"""Closed loop of the four vehicles from their own plans: per vehicle the solver status of every MPC iteration
(run on the GPU box: python tools/follow_diag.py [num_iter])."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conflict_rez_amd import strategy  # noqa: E402
from conflict_rez_amd.control.vehicle_follower import MultiDistributedFollower  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
stem = os.path.join(tempfile.mkdtemp(), "4v_rl_traj")
strategy.write_strategy(stem, strategy.generate_strategy(4))
agents = ["vehicle_%d" % i for i in range(4)]
mdf = MultiDistributedFollower(rl_file_name=stem, spline_ws_config={a: True for a in agents}, colors={a: {"front": (1, 0, 0), "back": (0, 0, 1)} for a in agents},
                               init_offsets={a: None for a in agents}, final_headings={a: None for a in agents})
mdf.setup_multi_vehicles()
hist = {a: [] for a in agents}
dist = []
for it in range(n_it):
    mdf.solve(num_iter=1, dump=False)
    for v in mdf.vehicles:
        hist[v.agent].append(v.status)
    P = np.array([[v.state.x.x, v.state.x.y] for v in mdf.vehicles])
    dist.append(min(np.hypot(*(P[i] - P[j])) for i in range(4) for j in range(i)))
for v in mdf.vehicles:
    h = np.array(hist[v.agent])
    bad = np.nonzero(h)[0]
    ref_end = v.reference_traj.t[-1]
    print(v.agent, "plan end %.1f s" % ref_end, "statuses", {int(s): int((h == s).sum()) for s in np.unique(h)}, "first/last bad iteration", (bad[:1], bad[-1:]),
          "tracking error now %.3f m" % np.hypot(v.state.x.x - v.interpolate_states([v.state.t]).x[0], v.state.x.y - v.interpolate_states([v.state.t]).y[0]))
    print("   bad iterations:", bad.tolist()[:60])
print("closest pair of rear axles over the run: %.2f m" % min(dist))
