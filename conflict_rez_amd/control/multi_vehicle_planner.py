"""`MultiVehiclePlanner`: the centralised planner's surface (mirror of the reference's
`confrez/control/multi_vehicle_planner.py`).

Built: the constructor (:31-66), `joint_dual_ws` (:208-341) on the GPU in closed form (`cfz_joint_dual_ws`), with the
reference's result layout `joint_l0[agent][other][i][k]` (4,), `joint_s0[(agent, other)][i][k]` (2,).  The solves it
sits between -- `solve_single_problems` (:68-109, needs the collocation NLP) and `solve_final_problem_obca` (:343-480,
the coupled collocation NLP) -- are rows of the coverage table that have no kernel yet (DESIGN.md "Next") and raise
`NotImplementedError`; `single_results` can be supplied to exercise `joint_dual_ws`.
"""
from itertools import combinations, product
from typing import Dict

import numpy as np

from ..pytypes import VehiclePrediction
from ..vehicle_types import VehicleBody, VehicleConfig
from .vehicle import Vehicle


class MultiVehiclePlanner:
    def __init__(self, rl_file_name: str, ws_config: Dict[str, bool], colors: Dict[str, dict], init_offsets: Dict,
                 final_headings: Dict[str, float], vehicle_body: VehicleBody = None, vehicle_config: VehicleConfig = None,
                 region=None):
        self.rl_file_name = rl_file_name
        self.ws_config, self.colors = ws_config, colors
        self.init_offsets, self.final_headings = init_offsets, final_headings
        self.vehicle_body = vehicle_body or VehicleBody()
        self.vehicle_config = vehicle_config or VehicleConfig()
        self.agents = sorted(ws_config.keys())
        self.vehicles: Dict[str, Vehicle] = {
            a: Vehicle(rl_file_name=rl_file_name, agent=a, color=colors[a], vehicle_config=self.vehicle_config,
                       vehicle_body=self.vehicle_body, region=region) for a in self.agents}
        self.agent_pairs = list(combinations(self.agents, 2))  # (:56-58)
        self.single_results: Dict[str, VehiclePrediction] = {}
        self.joint_l0, self.joint_s0 = {}, {}
        self._engine = None

    def solve_single_problems(self, *args, **kwargs):
        raise NotImplementedError("needs the collocation NLP (vehicle.py:360-661), which has no HIP kernel yet")

    def solve_final_problem_obca(self, *args, **kwargs):
        raise NotImplementedError("the coupled collocation NLP (multi_vehicle_planner.py:343-480) has no HIP kernel yet")

    def joint_dual_ws(self, K: int = 5, verbose: int = 0):
        """warm starting the dual multipliers for joint collision avoidance (:208-341): for every pair of vehicles and
        every collocation point of the shorter plan, the duals that certify the largest separation of the two bodies.
        `single_results[agent]` must hold x, y, psi flattened from (N, K+1); `vehicles[agent].N` the interval counts."""
        from ..engine import Engine, ProblemSpec

        print("warm starting dual variables for joint OBCA...")
        if self._engine is None:
            first = self.vehicles[self.agents[0]]
            spec = ProblemSpec.from_objects(first.obstacles, self.vehicle_body, self.vehicle_config, first.region, n_nbr=1, N=2)
            self._engine = Engine(spec, max_batch=1)
        self.joint_l0, self.joint_s0 = {}, {}
        for agent, other in self.agent_pairs:
            self.joint_l0.setdefault(agent, {})
            self.joint_l0.setdefault(other, {})
            Na, Nb = self.vehicles[agent].N, self.vehicles[other].N
            N_min = min(Na, Nb)
            pose = lambda a, N: np.stack([np.reshape(getattr(self.single_results[a], f), (N, K + 1))[:N_min].ravel()
                                          for f in ("x", "y", "psi")], 1)
            lam, mu, s, _ = self._engine.joint_dual_ws(pose(agent, Na), pose(other, Nb))
            shape = lambda arr: [[arr[i * (K + 1) + k].copy() for k in range(K + 1)] for i in range(N_min)]
            self.joint_l0[agent][other], self.joint_l0[other][agent] = shape(lam), shape(mu)
            self.joint_s0[(agent, other)] = shape(s)
