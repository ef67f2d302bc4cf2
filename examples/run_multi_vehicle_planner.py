#!/usr/bin/env python3
"""The reference's `confrez/control/multi_vehicle_planner.py:main` with the import switched (needs an MI355X):

    strategy .pkl  ->  MultiVehiclePlanner: every vehicle's own plan (state_ws, dual_ws, collocation with free dt), then the
    joint plan with one shared dt and vehicle-vehicle separation  ->  <name>_opt.pkl (final_results, as :622-623)

Without arguments a 4-vehicle strategy is generated with `conflict_rez_amd.strategy` (the reference ships none); pass the
path stem of a recorded strategy (the reference's `4v_rl_traj`) to use that instead.  `--agents` plans a subset (the
four-vehicle joint solve takes about a minute, two vehicles a few seconds).  The matplotlib animation of the reference
(`plot_results`) is not reproduced."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conflict_rez_amd import strategy  # noqa: E402
from conflict_rez_amd.control.compute_sets import interp_along_sets  # noqa: E402
from conflict_rez_amd.control.multi_vehicle_planner import MultiVehiclePlanner  # noqa: E402
from conflict_rez_amd.pytypes import VehicleState  # noqa: E402
from conflict_rez_amd.vehicle_types import VehicleBody  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rl_file_name", nargs="?", help="path stem of the strategy pickle (without .pkl)")
    ap.add_argument("--agents", default="vehicle_0,vehicle_1,vehicle_2,vehicle_3")
    ap.add_argument("--interp-dt", type=float, default=None)
    args = ap.parse_args()
    stem = args.rl_file_name
    if stem is None:
        stem = os.path.join(tempfile.mkdtemp(), "4v_rl_traj")
        strategy.write_strategy(stem, strategy.generate_strategy(4))
    agents = args.agents.split(",")
    colors = {a: {"front": (255, 119, 0), "back": (128, 128, 128)} for a in agents}
    # the reference's `main` fixes the terminal headings by hand (:617-618); here: the heading of the strategy's spline
    paths = interp_along_sets(stem, VehicleBody(), 30)
    planner = MultiVehiclePlanner(rl_file_name=stem, ws_config={a: True for a in agents}, colors=colors,
                                  init_offsets={a: VehicleState() for a in agents},
                                  final_headings={a: float(paths[a][-1, 2]) for a in agents})
    t0 = time.time()
    planner.solve_single_problems()
    t1 = time.time()
    for a in agents:
        r = planner.single_results[a]
        print(f"{a}: single plan {planner.vehicles[a].final_problem_stats}, dt {r.dt:.4f} s, horizon {r.t[-1]:.2f} s")
    planner.solve_final_problem_obca(interp_dt=args.interp_dt)
    t2 = time.time()
    print(f"joint plan {planner.final_stats}, shared dt {planner.final_dt:.4f} s; single plans {t1 - t0:.1f} s, joint {t2 - t1:.1f} s")
    fr = planner.final_results
    for i, a in enumerate(agents):
        for b in agents[i + 1:]:
            print(f"  {a} - {b}: closest rear axles {np.hypot(fr[a].x - fr[b].x, fr[a].y - fr[b].y).min():.2f} m")
    print("results in", planner.dump_results(stem))


if __name__ == "__main__":
    main()
