#!/usr/bin/env python3
"""The reference's `confrez/control/vehicle_follower.py:main` with the import switched (needs an MI355X):

    strategy .pkl  ->  MultiDistributedFollower: plan every vehicle (state_ws + dual_ws on the GPU), then the distributed
    MPC, four solves per iteration in one launch  ->  <name>_follower_final.pkl, <name>_follower_iter_time.pkl

Without arguments a conflict-free 4-vehicle strategy is generated with `conflict_rez_amd.strategy` (the reference ships
none); pass the path stem of a recorded strategy (the reference's `4v_rl_traj`) to use that instead."""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conflict_rez_amd import strategy  # noqa: E402
from conflict_rez_amd.control.vehicle_follower import MultiDistributedFollower  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rl_file_name", nargs="?", help="path stem of the strategy pickle (without .pkl)")
    ap.add_argument("--num-iter", type=int, default=100)
    args = ap.parse_args()
    stem = args.rl_file_name
    if stem is None:
        stem = os.path.join(tempfile.mkdtemp(), "4v_rl_traj")
        strategy.write_strategy(stem, strategy.generate_strategy(4))
    agents = ["vehicle_%d" % i for i in range(4)]
    colors = {"vehicle_0": {"front": (1.0, 0.47, 0.42), "back": (0.6, 0.2, 0.2)},
              "vehicle_1": {"front": (0.0, 0.62, 0.45), "back": (0.0, 0.3, 0.2)},
              "vehicle_2": {"front": (0.0, 0.45, 0.7), "back": (0.0, 0.2, 0.4)},
              "vehicle_3": {"front": (0.8, 0.47, 0.65), "back": (0.4, 0.2, 0.3)}}
    mdf = MultiDistributedFollower(rl_file_name=stem, spline_ws_config={a: True for a in agents}, colors=colors,
                                   init_offsets={a: None for a in agents}, final_headings={a: None for a in agents})
    mdf.setup_multi_vehicles()
    for v in mdf.vehicles:
        print(v.agent, "plan:", v.state_ws_stats, "horizon", round(float(v.reference_traj.t[-1]), 1), "s")
    mdf.solve(num_iter=args.num_iter)
    for v in mdf.vehicles:
        it = np.array(v.iter_time)
        print(v.agent, "final pose", np.round([v.state.x.x, v.state.x.y, v.state.e.psi], 3),
              f"mean solve {it[it < 0.5].mean() * 1e3:.2f} ms, fallbacks {(it == 0.5).sum()}")
    print("results in", stem + "_follower_final.pkl")


if __name__ == "__main__":
    main()
