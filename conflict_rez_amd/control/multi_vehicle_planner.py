"""`MultiVehiclePlanner`: the centralised planner's surface (mirror of the reference's
`confrez/control/multi_vehicle_planner.py`).

Built: the constructor (:31-66), `joint_dual_ws` (:208-341) on the GPU in closed form (`cfz_joint_dual_ws`), with the
reference's result layout `joint_l0[agent][other][i][k]` (4,), `joint_s0[(agent, other)][i][k]` (2,), and
`solve_single_problems` (:68-109) and `solve_final_problem_obca` (:343-480: the collocation problems of all vehicles coupled
by one shared dt and pairwise separation rows, `cfz_joint_colloc`) on the GPU planning kernels.  Not reproduced:
`solve_final_problem_circles` (:111-206, unused by the reference's `main`) and the matplotlib animation (:482-603).
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
        self.final_results: Dict[str, VehiclePrediction] = {}
        self._engine = None

    def solve_single_problems(self, N: int = 30, K: int = 5, N_per_set: int = 5, dt: float = 0.1, shrink_tube: float = 0.5,
                              dmin: float = 0.05):
        """solve single vehicle control problems (:68-109): per agent state_ws -> dual_ws -> interp_ws_for_collocation ->
        setup_single_final_problem -> solve_single_final_problem -> get_solution, all on the GPU (`cfz_state_ws`,
        `cfz_dual_ws`, `cfz_colloc`).  Fills `single_results[agent]` (with `.dt`, nested `l`, `m`) and leaves every
        vehicle's interpolators and N, K set, which is what `joint_dual_ws` reads."""
        self.single_results = {agent: VehiclePrediction() for agent in self.agents}
        for agent in self.agents:
            print(f"==== Solving single vehicle problem for {agent} ====")
            vehicle = self.vehicles[agent]
            zu0 = vehicle.state_ws(N=N, dt=dt, init_offset=self.init_offsets[agent], final_heading=self.final_headings[agent],
                                   shrink_tube=shrink_tube, spline_ws=self.ws_config[agent])
            zu0 = vehicle.dual_ws(zu0=zu0)
            zu0 = vehicle.interp_ws_for_collocation(zu0=zu0, K=K, N_per_set=N_per_set)
            vehicle.setup_single_final_problem(zu0=zu0, init_offset=self.init_offsets[agent], final_heading=self.final_headings[agent],
                                               K=K, N_per_set=N_per_set, shrink_tube=shrink_tube, dmin=dmin)
            sol = vehicle.solve_single_final_problem()
            self.single_results[agent] = vehicle.get_solution(sol=sol)

    def solve_final_problem_obca(self, K: int = 5, N_per_set: int = 5, shrink_tube: float = 0.5, dmin: float = 0.05,
                                 interp_dt: float = None):
        """solve joint collision avoidance problem with OBCA (:343-480): every vehicle's collocation problem (from its single
        result) with ONE shared dt, cost sum_a J_a, and every pair of vehicles at least `dmin` apart at every collocation
        point of the shorter plan -- on the GPU (`cfz_joint_colloc`).  `joint_dual_ws` runs first as in the reference; its
        duals are not needed by the kernel (the vehicle-vehicle duals are eliminated like the obstacle duals).  Fills
        `final_results[agent]` = the plan interpolated on the common time grid (:466-480), `final_dt`, `final_stats`;
        raises RuntimeError when the solver does not converge, as `opti.solve()` does."""
        from .joint_problem import JointOpti

        self.joint_dual_ws(K=K)
        print("Solving joint final problem with obca...")
        dt0 = float(np.mean([self.single_results[agent].dt for agent in self.agents]))  # (:360)
        opti = JointOpti()      # (:365)
        dt = opti.variable()    # (:366)
        opti.set_initial(dt, dt0)
        for agent in self.agents:  # (:371-386) every vehicle adds its collocation problem, on the shared dt
            self.vehicles[agent].setup_single_final_problem(zu0=self.single_results[agent], init_offset=self.init_offsets[agent],
                                                            final_heading=self.final_headings[agent], opti=opti, dt=dt, K=K,
                                                            N_per_set=N_per_set, dmin=dmin, shrink_tube=shrink_tube)
        index = {a: i for i, a in enumerate(self.agents)}
        res = opti.solve(pairs=[(index[a], index[b]) for a, b in self.agent_pairs])  # (:389-466)
        self.final_stats = dict(status=res["status"], iters=res["iters"], cost=res["cost"])
        if res["status"] != 0:
            raise RuntimeError(f"joint final problem did not converge (status {res['status']} after {res['iters']} iterations)")
        print("Solve_Succeeded")
        self.final_dt = res["dt"]
        N_max = max(self.vehicles[agent].N for agent in self.agents)
        if interp_dt is None:
            final_t = np.linspace(0, N_max * res["dt"], N_max * (K + 1) + 1, endpoint=True)
        else:
            final_t = np.arange(0, N_max * res["dt"], interp_dt)
        self.final_results = {agent: VehiclePrediction() for agent in self.agents}
        for agent in self.agents:
            tr = res["traj"][index[agent]]
            sol = {k: tr[:, :, c].copy() for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
            sol["dt"] = res["dt"]
            self.vehicles[agent].get_solution(sol=sol)
            self.final_results[agent] = self.vehicles[agent].interpolate_states(final_t)

    def dump_results(self, rl_file_name: str = None):
        """`dill.dump(planner.final_results, open(f"{rl_file_name}_opt.pkl", "wb"))` of the reference's `main` (:668)."""
        from ..results import dump

        return dump(self.final_results, f"{rl_file_name or self.rl_file_name}_opt.pkl")

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
