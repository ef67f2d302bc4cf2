"""`VehicleFollower` / `MultiDistributedFollower`: the distributed-MPC call surface.

Same constructor arguments, method names, defaults and attribute side effects as the reference's
`confrez/control/vehicle_follower.py` (`VehicleFollower` :36-563, `MultiDistributedFollower`
:566-670); `opti.solve()` (:479) is replaced by one call into the batched HIP engine
(`conflict_rez_amd.engine.Engine`, C ABI include/confrez_hip.h).  A solver outcome other than
"converged" plays the role of the exception the reference catches: the prediction is shifted one
step and `back_up_steps` decremented (:501-524).

Differences a caller can see (all listed in INTEGRATION.md): no CasADi attributes (`opti`, `x`
as MX...); the OBCA duals `pred.l`, `pred.m`, `opt_lambda_ij/ji`, `opt_s` are outputs only
(rebuilt from the poses every solve); `iter_time` holds the device time of the solve kernel;
the visualiser is optional.
"""
import time
from typing import Dict, List, Tuple

import numpy as np

from ..engine import Engine, ProblemSpec
from ..obstacle_types import GeofenceRegion
from ..pytypes import VehiclePrediction, VehicleState
from ..vehicle_types import VehicleBody, VehicleConfig
from .compute_sets import compute_obstacles, compute_sets, compute_static_vehicles
from .dynamic_model import simulator
from .vehicle import Vehicle

np.random.seed(0)  # reference vehicle_follower.py:29 (seeds the first dual guesses :401-402)
static_vehicles = compute_static_vehicles()  # :31 -- cosmetics, but its 15 draws come BEFORE the vehicles' first dual guesses

_PRIMAL = ("x", "y", "psi", "v", "u_steer", "u_a", "u_steer_dot")  # zu rows: x y psi v delta a w


class VehicleFollower(Vehicle):
    """Vehicle that avoids obstacles and the other vehicles while following its planned path."""

    def __init__(
        self,
        rl_file_name: str,
        agent: str,
        color: Dict[str, Tuple[float, float, float]],
        init_offset: VehicleState,
        final_heading: float,
        vehicle_config: VehicleConfig = None,
        vehicle_body: VehicleBody = None,
        region: GeofenceRegion = None,
        printer: callable = None,
        engine: Engine = None,
    ) -> None:
        super().__init__(rl_file_name, agent, color, vehicle_config, vehicle_body, region)
        self.init_offset = init_offset
        self.final_heading = final_heading
        self.state: VehicleState = self.init_state  # alias, as in the reference (:59)
        self.state.t = 0
        self.pred: VehiclePrediction = None
        self.back_up_steps: int = 0
        self.others: List[str] = []
        self.others_pred: Dict[str, VehiclePrediction] = {}
        self.reference_traj: VehiclePrediction = None
        self.reference_xy = None
        self.ref_idx_lb, self.ref_idx_ub = 0, -1
        self.ref_pair: List[np.ndarray] = []
        ft = self.final_traj = VehiclePrediction()
        ft.t, ft.x, ft.y, ft.psi = [self.state.t], [self.state.x.x], [self.state.x.y], [self.state.e.psi]
        ft.v, ft.u_steer = [self.state.v.v], [self.state.u.u_steer]
        ft.u_a, ft.u_steer_dot = [self.state.u.u_a], [self.state.u.u_steer_dot]
        self.iter_time = []
        self.print = printer or print
        self.engine = engine
        self.status = None

    # ---- reference path ---------------------------------------------------------------------
    def plan_single_path(self, N_ws=30, dt_ws=0.1, K=5, N_per_set=5, shrink_tube=0.5, dmin=0.05, spline_ws=True,
                         interp_dt=0.01, strict=False):
        """plan a single vehicle reference path (:91-138): state_ws -> dual_ws -> resampling onto the collocation grid ->
        collocation plan -> `reference_traj` sampled every `interp_dt` from the collocation interpolant.
        `self.plan_refined` is True when the reference is the collocation plan.  Two fallbacks the reference does not
        have (it would raise): a final heading the warm start cannot meet is retried without it, and if the collocation
        solve fails the warm start itself becomes the reference (`plan_refined` False).  `strict=True` switches both off:
        any solver failure raises RuntimeError, as `opti.solve()` does in the reference (vehicle.py:216, :658)."""
        try:
            zu0 = self.state_ws(N=N_ws, dt=dt_ws, init_offset=self.init_offset, final_heading=self.final_heading,
                                shrink_tube=shrink_tube, spline_ws=spline_ws)
        except RuntimeError:
            if self.final_heading is None or strict:
                raise
            zu0 = self.state_ws(N=N_ws, dt=dt_ws, init_offset=self.init_offset, final_heading=None,
                                shrink_tube=shrink_tube, spline_ws=spline_ws)
        zu0 = self.dual_ws(zu0=zu0)
        zuc = self.interp_ws_for_collocation(zu0=zu0, K=K, N_per_set=N_per_set)
        self.setup_single_final_problem(zu0=zuc, init_offset=self.init_offset, final_heading=self.final_heading, K=K,
                                        N_per_set=N_per_set, shrink_tube=shrink_tube, dmin=dmin)
        try:
            sol = self.solve_single_final_problem()
        except RuntimeError as e:
            if strict:
                raise
            self.print(f"{self.agent}: {e}; following the warm start")
            self.plan_refined = False
            self.set_reference(zu0, interp_dt=interp_dt)
            return
        result = self.get_solution(sol=sol)
        self.plan_refined = True
        interp_time = np.linspace(start=result.t[0], stop=result.t[-1], num=int((result.t[-1] - result.t[0]) / interp_dt), endpoint=True)
        self.reference_traj = self.interpolate_states(interp_time)
        self.reference_xy = np.vstack([self.reference_traj.x, self.reference_traj.y]).T

    def set_reference(self, traj: VehiclePrediction, interp_dt: float = 0.01):
        """Installs a planned trajectory: builds the interpolators and `reference_traj` sampled every
        `interp_dt`, exactly what `plan_single_path` leaves behind (:130-138)."""
        self.set_reference_trajectory(traj)
        t0, t1 = traj.t[0], traj.t[-1]
        self.reference_traj = self.interpolate_states(np.linspace(t0, t1, int((t1 - t0) / interp_dt), endpoint=True))
        self.reference_xy = np.vstack([self.reference_traj.x, self.reference_traj.y]).T

    def get_others(self, vehicles):
        self.others = [v.agent for v in vehicles if v.agent != self.agent]

    # ---- controller ------------------------------------------------------------------------------
    def setup_controller(self, dt: float = 0.1, N: int = 30, dmin=0.05):
        """Fixes the NLP structure (:146-368) and, unless one was passed in, creates the engine."""
        self.print(f"setting up controller for {self.agent}...")
        self.N, self.dt = N, dt
        self.horizon_interp_ahead = np.linspace(0, N * dt, N, endpoint=False)
        self.spec = ProblemSpec.from_objects(self.obstacles, self.vehicle_body, self.vehicle_config, self.region,
                                             n_nbr=len(self.others), N=N, dt=dt, dmin=dmin)
        self.simulator = simulator(dt=dt, vehicle_body=self.vehicle_body)
        n_l = 4 * len(self.obstacles)
        self.l_shape, self.m_shape = (N, n_l), (N, n_l)
        self.opt_lambda_ij = {o: np.zeros((N, 4)) for o in self.others}
        self.opt_lambda_ji = {o: np.zeros((N, 4)) for o in self.others}
        self.opt_s = {o: np.zeros((N, 2)) for o in self.others}
        if self.engine is None:
            self.engine = Engine(self.spec, max_batch=1)

    def get_current_ref(self):
        """Closest reference sample in time, then N samples dt apart, final pose held (:370-404)."""
        min_idx = np.abs(self.reference_traj.t - self.state.t).argmin()
        t_span = self.reference_traj.t[min_idx] + self.horizon_interp_ahead
        self.ref_idx_ub = np.abs(self.reference_traj.t - t_span[-1]).argmin()
        self.ref_pair.append(np.array([[self.reference_traj.x[min_idx], self.reference_traj.y[min_idx]],
                                       [self.state.x.x, self.state.x.y]]))
        result = self.interpolate_states(time=t_span)
        if self.pred is None:
            self.pred = result.copy()
            self.pred.l = 0.1 * np.random.rand(*self.l_shape)  # kept for RNG-stream compatibility (:401-402)
            self.pred.m = 0.1 * np.random.rand(*self.m_shape)
        return result

    def get_others_pred(self, vehicles):
        for v in vehicles:
            self.others_pred[v.agent] = v.pred.copy()

    def _adv_onestep(self, array: np.ndarray):
        array = np.asarray(array)
        if array.ndim == 1:
            return np.append(array[1:], array[-1])
        if array.ndim == 2:
            return np.vstack([array[1:, :], array[-1, :]])
        raise ValueError("unexpected shape when advancing the array to one step ahead.")

    # ---- one MPC step, split so that several vehicles can share one kernel launch ----------------------
    def prepare_step(self):
        """Parameters and warm start of this step (:432-476): (x0[5], ref[3,N], nbr[n_nbr,3,N], zu[7,N])."""
        s = self.state
        x0 = np.array([s.x.x, s.x.y, s.e.psi, s.v.v, s.u.u_steer], float)
        cur = self.get_current_ref()
        ref = np.stack([cur.x, cur.y, cur.psi])
        nbr = np.zeros((len(self.others), 3, self.N))
        for o, other in enumerate(self.others):
            p = self.others_pred[other]
            nbr[o] = np.stack([self._adv_onestep(p.x), self._adv_onestep(p.y), self._adv_onestep(p.psi)])
        zu = np.stack([self._adv_onestep(getattr(self.pred, n)) for n in _PRIMAL])
        return x0, ref, nbr, zu

    def finish_step(self, out, b: int = 0, solve_time: float = None):
        """Read-back (:484-500) or shift fallback (:501-524), clock and plant (:526-563)."""
        self.status = int(out["status"][b])
        if self.status == 0:
            self.iter_time.append(out["solve_ms"] / 1e3 if solve_time is None else solve_time)
            self.back_up_steps = self.N - 1
            for r, n in enumerate(_PRIMAL):
                setattr(self.pred, n, out["zu"][b, r].copy())
            self.pred.l, self.pred.m = out["l"][b].copy(), out["m"][b].copy()
            for o, other in enumerate(self.others):
                self.opt_lambda_ij[other] = out["lam_ij"][b, o].copy()
                self.opt_lambda_ji[other] = out["lam_ji"][b, o].copy()
                self.opt_s[other] = out["s"][b, o].copy()
        else:
            self.iter_time.append(0.5)
            self.back_up_steps -= 1
            for n in _PRIMAL + ("l", "m"):
                setattr(self.pred, n, self._adv_onestep(getattr(self.pred, n)))
            for other in self.others:
                self.opt_lambda_ij[other] = self._adv_onestep(self.opt_lambda_ij[other])
                self.opt_lambda_ji[other] = self._adv_onestep(self.opt_lambda_ji[other])
                self.opt_s[other] = self._adv_onestep(self.opt_s[other])
        s = self.state
        s.t += self.dt
        z = self.simulator([s.x.x, s.x.y, s.e.psi, s.v.v, s.u.u_steer], [self.pred.u_a[0], self.pred.u_steer_dot[0]])
        s.x.x, s.x.y, s.e.psi, s.v.v, s.u.u_steer = (float(v) for v in z)
        s.u.u_a, s.u.u_steer_dot = float(self.pred.u_a[0]), float(self.pred.u_steer_dot[0])
        ft = self.final_traj
        for lst, val in ((ft.t, s.t), (ft.x, s.x.x), (ft.y, s.x.y), (ft.psi, s.e.psi), (ft.v, s.v.v),
                         (ft.u_steer, s.u.u_steer), (ft.u_a, s.u.u_a), (ft.u_steer_dot, s.u.u_steer_dot)):
            lst.append(val)

    def step(self):
        """step the controller (:428-563)"""
        x0, ref, nbr, zu = self.prepare_step()
        # a batch of one in THIS vehicle's carry slot of the (possibly shared) engine: after a converged step the next
        # one starts from its own multipliers (the reference hands the previous duals to opti.set_initial, :458-464,
        # :475-476), never from another vehicle's
        out = self.engine.solve(x0[None], ref[None], nbr[None], zu[None], carry=[int(getattr(self, "status", 1) == 0)],
                                slots=[getattr(self, "slot", 0)])
        self.finish_step(out, 0)


class MultiDistributedFollower:
    """Several vehicles as distributed path followers (:566-670); the four `step()` solves of one
    iteration are independent (Jacobi exchange, :636-641) and go to the GPU as one batch."""

    def __init__(self, rl_file_name, spline_ws_config, colors, init_offsets, final_headings, visualizer=None):
        self.rl_file_name = rl_file_name
        self.spline_ws_config = spline_ws_config
        self.colors, self.init_offsets, self.final_headings = colors, init_offsets, final_headings
        self.agents = sorted(spline_ws_config.keys())
        self.vehicles: List[VehicleFollower] = [
            VehicleFollower(rl_file_name=rl_file_name, agent=a, color=colors[a], init_offset=init_offsets[a],
                            final_heading=final_headings[a]) for a in self.agents]
        self.rl_tubes = compute_sets(rl_file_name)
        self.obstacles = compute_obstacles()
        self.iter_time = {a: [] for a in self.agents}
        self.single_results: Dict[str, VehiclePrediction] = {}
        self.final_results: Dict[str, VehiclePrediction] = {}
        self.vis = visualizer  # the reference opens a pygame window here (:609); optional in this build
        self.engine = None

    def setup_multi_vehicles(self, references: Dict[str, VehiclePrediction] = None):
        """plan -> get_others -> setup_controller -> get_current_ref for every vehicle (:614-624).
        `references` supplies planned trajectories from elsewhere instead of planning them here."""
        for v in self.vehicles:
            if references is not None:
                v.set_reference(references[v.agent])
            else:
                v.plan_single_path(spline_ws=self.spline_ws_config[v.agent])
            v.get_others(self.vehicles)
        first = self.vehicles[0]
        spec = ProblemSpec.from_objects(first.obstacles, first.vehicle_body, first.vehicle_config, first.region,
                                        n_nbr=len(self.vehicles) - 1)
        self.engine = Engine(spec, max_batch=len(self.vehicles))
        for b, v in enumerate(self.vehicles):
            v.engine, v.slot = self.engine, b  # slot b of the shared engine is vehicle b, in `solve()` and in `v.step()`
            v.setup_controller()
            v.get_current_ref()
            self.single_results[v.agent] = v.reference_traj

    def solve(self, num_iter: int = 500, dump: bool = True):
        for _ in range(num_iter):
            for v in self.vehicles:
                v.get_others_pred(self.vehicles)
            batch = [v.prepare_step() for v in self.vehicles]
            t0 = time.perf_counter()
            # slot b is always vehicle b: a vehicle whose last step converged starts from its multipliers
            carry = [int(getattr(v, "status", 1) == 0) for v in self.vehicles]
            out = self.engine.solve(*(np.stack([b[i] for b in batch]) for i in range(4)), carry=carry)
            wall = time.perf_counter() - t0
            for b, v in enumerate(self.vehicles):
                v.finish_step(out, b, solve_time=wall / len(self.vehicles))
            if self.vis is not None:
                self.vis.draw_background(), self.vis.draw_obstacles()
                for v in self.vehicles:
                    self.vis.draw_traj(v.final_traj, 255 * np.array(v.color["front"]))
                    self.vis.draw_car(v.state, 255 * np.array(v.color["front"]))
                self.vis.render()
        for v in self.vehicles:
            self.iter_time[v.agent] = v.iter_time
            self.final_results[v.agent] = v.final_traj
        print(f"Mean iteration time = {[np.mean(self.iter_time[a]) for a in self.agents]}")
        print(f"Max iteration time = {[np.amax(self.iter_time[a]) for a in self.agents]}")
        if dump:  # same file names as the reference (:665-670)
            from ..results import dump as dump_file

            dump_file(self.final_results, f"{self.rl_file_name}_follower_final.pkl")
            dump_file(self.iter_time, f"{self.rl_file_name}_follower_iter_time.pkl")
