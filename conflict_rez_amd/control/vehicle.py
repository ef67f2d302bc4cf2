"""`Vehicle`: single-vehicle planning surface (mirror of the reference's `confrez/control/vehicle.py`).

Built in this round: the constructor (:29-52), the collocation tables (`collocation_coefficients`
:54-97), the warm-start resampling (`interp_ws_for_collocation` :298-358), the Lagrange
interpolant of a collocation solution (`get_interpolator` :722-786, `interpolate_states`
:788-829) -- all numpy, no CasADi.  The three NLPs run on the GPU: `state_ws` (:99-231, `cfz_state_ws`: banded interior
point), `dual_ws` (:233-296, `cfz_dual_ws`: closed form) and the Radau collocation plan with free dt
(`setup_single_final_problem` / `solve_single_final_problem` :360-661, `cfz_colloc`).
"""
from typing import Dict, Tuple

import numpy as np

from ..obstacle_types import GeofenceRegion
from ..pytypes import VehiclePrediction, VehicleState
from ..vehicle_types import VehicleBody, VehicleConfig
from .compute_sets import compute_initial_states, compute_obstacles, compute_sets


def radau_points(K: int) -> np.ndarray:
    """Radau IIA collocation points on (0, 1]: what `ca.collocation_points(K, "radau")` returns.
    They are the roots of P_{K-1}(x) - P_K(x) (Legendre, x in [-1,1]) mapped to [0,1]; the root
    x = 1 gives tau = 1."""
    from numpy.polynomial import legendre as L

    c = np.zeros(K + 1)
    c[K - 1], c[K] = 1.0, -1.0
    roots = np.sort(np.real(L.legroots(c)))
    return (roots + 1.0) / 2.0


class Vehicle:
    def __init__(
        self,
        rl_file_name: str,
        agent: str,
        color: Dict[str, Tuple[float, float, float]],
        vehicle_config: VehicleConfig = None,
        vehicle_body: VehicleBody = None,
        region: GeofenceRegion = None,
    ) -> None:
        self.rl_file_name = rl_file_name
        self.agent = agent
        self.color = color
        self.vehicle_config = vehicle_config or VehicleConfig()
        self.vehicle_body = vehicle_body or VehicleBody()
        self.region = region or GeofenceRegion()
        self.init_state: VehicleState = compute_initial_states(rl_file_name, self.vehicle_body)[agent]
        self.obstacles = compute_obstacles()
        self.rl_tube = compute_sets(rl_file_name)[agent]
        self.num_sets = len(self.rl_tube)
        self.state_interpolator = None
        self.input_interpolator = None

    # ---- collocation tables -------------------------------------------------------------
    def collocation_coefficients(self, K: int):
        """A[j,k] = l_j'(tau_k), B[j] = int_0^1 l_j, D[j] = l_j(1) for the Lagrange basis on
        tau = [0, radau_points(K)]."""
        tau = np.append(0.0, radau_points(K))
        A, B, D = np.zeros((K + 1, K + 1)), np.zeros(K + 1), np.zeros(K + 1)
        for j in range(K + 1):
            others = np.delete(tau, j)
            p = np.poly1d(np.poly(others) / np.prod(tau[j] - others))
            D[j] = p(1.0)
            A[j, :] = np.polyder(p)(tau)
            B[j] = np.polyint(p)(1.0)
        return A, B, D

    # ---- planning NLPs (next rows of the coverage table) ------------------------------------
    def state_ws(self, N: int = 30, dt: float = 0.1, init_offset: VehicleState = None, final_heading: float = None,
                 bounded_input: bool = False, shrink_tube: float = 0.8, spline_ws: bool = False,
                 verbose: int = 0) -> VehiclePrediction:
        """Warm-start plan through the strategy's tube (vehicle.py:99-231) on the GPU (`cfz_state_ws`): T = N (S-1)
        Euler steps, rear-axle and front point inside the shrunk back/front cells at every strategy step, cost
        sum a^2 + w^2.  Returns t, x, y, psi, v, u_steer (= delta), u_a, u_steer_dot with the last input repeated
        (:219-231); raises RuntimeError when the solver does not converge, as `opti.solve()` does."""
        from ..engine import state_ws as cfz_state_ws
        from .compute_sets import interp_along_sets

        print("Solving state ws...")
        off = init_offset if init_offset is not None else VehicleState()
        s0 = self.init_state
        init_pose = [s0.x.x + off.x.x, s0.x.y + off.x.y, s0.e.psi + off.e.psi]
        tube = [((st["back"].A, st["back"].b), (st["front"].A, st["front"].b)) for st in self.rl_tube[1:]]
        guess = None
        if spline_ws:
            guess = interp_along_sets(file_name=self.rl_file_name, vehicle_body=self.vehicle_body, N=N)[self.agent]
        vc, r = self.vehicle_config, self.region
        bounds = [r.x_min, r.x_max, r.y_min, r.y_max, vc.v_min, vc.v_max, vc.delta_min, vc.delta_max,
                  vc.a_min, vc.a_max, vc.w_delta_min, vc.w_delta_max]
        res = cfz_state_ws([init_pose], [tube], None if guess is None else [guess], [final_heading], N=N, dt=dt,
                           wb=self.vehicle_body.wb, shrink_tube=shrink_tube, bounded_input=int(bounded_input), bounds=bounds)[0]
        self.state_ws_stats = dict(status=res["status"], iters=res["iters"], cost=res["cost"])
        if res["status"] != 0:
            raise RuntimeError(f"state_ws did not converge (status {res['status']} after {res['iters']} iterations)")
        T = N * (self.num_sets - 1)
        tr = res["traj"]
        out = VehiclePrediction()
        out.t = np.linspace(0, T * dt, T + 1, endpoint=True)
        out.x, out.y, out.psi, out.v, out.u_steer = (tr[:, c].copy() for c in range(5))
        out.u_a, out.u_steer_dot = tr[:, 5].copy(), tr[:, 6].copy()
        return out

    def dual_ws(self, zu0: VehiclePrediction, verbose: int = 0) -> VehiclePrediction:
        """Warm start of the OBCA duals for the fixed poses of `zu0` (vehicle.py:233-296): fills
        `zu0.l`, `zu0.m` as [4 n_obs, T+1] arrays, like `sol.value(l)`, `sol.value(m)` (:293-294).
        One GPU thread per (pose, obstacle): closed-form separation certificates (`cfz_dual_ws`)."""
        from ..engine import Engine, ProblemSpec

        if getattr(self, "_dual_engine", None) is None:
            spec = ProblemSpec.from_objects(self.obstacles, self.vehicle_body, self.vehicle_config, self.region,
                                            n_nbr=0, N=2)
            self._dual_engine = Engine(spec, max_batch=1)
        poses = np.stack([np.asarray(zu0.x, float), np.asarray(zu0.y, float), np.asarray(zu0.psi, float)], 1)
        l, m, _ = self._dual_engine.dual_ws(poses)
        zu0.l, zu0.m = l.T.copy(), m.T.copy()
        return zu0

    def setup_single_final_problem(self, zu0: VehiclePrediction, init_offset: VehicleState = None, final_heading: float = None,
                                   opti=None, dt=None, K: int = 5, N_per_set: int = 5, dmin: float = 0.05,
                                   shrink_tube: float = 0.8):
        """setup the final problem of a single vehicle (vehicle.py:360-640): N = N_per_set (S-1) intervals of free length
        dt with K = 5 Radau points, ODE at every point, continuity, tube rows at every N_per_set-th interval and at the
        end, terminal v = delta = a = w = 0 (+ heading), boxes, at least `dmin` from every static obstacle at every point,
        cost sum B_k (a^2 + v^2 w^2 + delta^2) dt + (N dt)^2.  `zu0`: guess at the collocation points
        (`interp_ws_for_collocation`); its l, m are not used: the OBCA duals are eliminated in the kernel
        (csrc/cfz_colloc.inl) and rebuilt from the poses by `get_solution`.  Returns the problem description that
        `solve_single_final_problem` hands to `cfz_colloc` (the reference returns its `ca.Opti`).
        `opti`, `dt` (:364-366, :386-389): a shared problem (`joint_problem.JointOpti`) and its shared interval length
        (`opti.variable()`), as `MultiVehiclePlanner.solve_final_problem_obca` passes them: the vehicle's problem is then
        added to the shared one, to be solved with the other vehicles' (`opti.solve`), and `self.opti` is that object."""
        if (opti is None) != (dt is None):
            raise TypeError("opti and dt come together: dt is the shared variable of opti (JointOpti.variable())")
        if K != 5:
            raise NotImplementedError("cfz_colloc is built for K = 5 (CFZ_COLLOC_K), the reference's only caller value")
        from ..engine import ProblemSpec

        off = init_offset if init_offset is not None else VehicleState()
        s0 = self.init_state
        N = N_per_set * (self.num_sets - 1)
        self.N, self.K = N, K
        guess = np.stack([np.asarray(getattr(zu0, n), float).reshape(N * (K + 1)) for n in
                          ("x", "y", "psi", "v", "u_steer", "u_a", "u_steer_dot")], 1)
        self.final_problem = dict(
            spec=ProblemSpec.from_objects(self.obstacles, self.vehicle_body, self.vehicle_config, self.region, n_nbr=0, N=2, dmin=dmin),
            init_pose=[s0.x.x + off.x.x, s0.x.y + off.x.y, s0.e.psi + off.e.psi], final_heading=final_heading,
            tube=[((st["back"].A, st["back"].b), (st["front"].A, st["front"].b)) for st in self.rl_tube[1:]],
            guess=guess, dt0=float(zu0.t[-1]) / N, N_per_set=N_per_set, shrink_tube=shrink_tube)
        if opti is not None:
            opti.add(self.final_problem, dt)
            self.opti = opti
            return opti
        return self.final_problem

    def solve_single_final_problem(self, verbose: int = 0):
        """solve the trajectory of single vehicle problem (vehicle.py:642-661) on the GPU (`cfz_colloc`, tol =
        constr_viol_tol = 1e-2 as :650-651).  Returns the mapping `get_solution` reads (x, y, psi, v, delta, a, w of shape
        (N, K+1), dt); raises RuntimeError when the solver does not converge, as `opti.solve()` does."""
        from ..engine import colloc as cfz_colloc

        print("Solving single vehicle final trajectory...")
        fp = self.final_problem
        res = cfz_colloc(fp["spec"], [fp["init_pose"]], [fp["tube"]], [fp["guess"]], [fp["dt0"]], [fp["final_heading"]],
                         N_per_set=fp["N_per_set"], shrink_tube=fp["shrink_tube"])[0]
        self.final_problem_stats = dict(status=res["status"], iters=res["iters"], cost=res["cost"])
        if verbose:
            print(self.final_problem_stats)
        if res["status"] != 0:
            raise RuntimeError(f"single final problem did not converge (status {res['status']} after {res['iters']} iterations)")
        print("Solve_Succeeded")
        tr = res["traj"]
        sol = {k: tr[:, :, c].copy() for c, k in enumerate(("x", "y", "psi", "v", "delta", "a", "w"))}
        sol["dt"] = res["dt"]
        return sol

    def get_solution(self, sol) -> VehiclePrediction:
        """get solution of this vehicle (vehicle.py:663-720): the collocation arrays (N, K+1) flattened row-major,
        t = (i + tau) dt, l, m as nested lists [N][K+1] of (24,) arrays, and the interpolators set.  `sol` is whatever
        holds the solved arrays: a mapping with x, y, psi, v, delta, a, w of shape (N, K+1), dt, and optionally l, m
        of shape (N, K+1, 24) -- the reference reads the same quantities off a CasADi `OptiSol`; when l, m are absent
        they are the closed-form certificates of the poses (`dual_ws`)."""
        X = {k: np.asarray(sol[k], float) for k in ("x", "y", "psi", "v", "delta", "a", "w")}
        N, K1 = X["x"].shape
        self.N, self.K = N, K1 - 1
        dt = float(sol["dt"])
        result = VehiclePrediction()
        result.dt = dt
        tau_root = np.append(0.0, radau_points(self.K))
        result.t = (np.arange(N)[:, None] + tau_root[None, :]).ravel() * dt
        result.x, result.y, result.psi = X["x"].ravel(), X["y"].ravel(), X["psi"].ravel()
        result.v, result.u_steer = X["v"].ravel(), X["delta"].ravel()
        result.u_a, result.u_steer_dot = X["a"].ravel(), X["w"].ravel()
        if "l" in sol and "m" in sol:
            L, M = np.asarray(sol["l"], float), np.asarray(sol["m"], float)
        else:
            tmp = VehiclePrediction()
            tmp.x, tmp.y, tmp.psi = result.x, result.y, result.psi
            self.dual_ws(tmp)
            L, M = tmp.l.T.reshape(N, K1, -1), tmp.m.T.reshape(N, K1, -1)
        result.l = [[np.array(L[i, k]) for k in range(K1)] for i in range(N)]
        result.m = [[np.array(M[i, k]) for k in range(K1)] for i in range(N)]
        self.get_interpolator(K=self.K, N=N, dt=dt, opt=result)
        return result

    def dump_plan(self, zu0: VehiclePrediction, result: VehiclePrediction, rl_file_name: str = None):
        """The two files the reference's `main` leaves behind (vehicle.py:927-928): the warm start on the collocation grid
        and the collocation plan, as `<rl_file_name>_<agent>_zu0.pkl` / `_zufinal.pkl`.  Returns the two paths."""
        from ..results import dump

        stem = f"{rl_file_name or self.rl_file_name}_{self.agent}"
        return dump(zu0, stem + "_zu0.pkl"), dump(result, stem + "_zufinal.pkl")

    # ---- resampling of a warm start onto the collocation grid ---------------------------------
    def interp_ws_for_collocation(self, zu0: VehiclePrediction, K: int = 5, N_per_set: int = 5):
        """Linear interpolation of every array of `zu0` at t = (i + tau) / N * T_end; l, m come in as
        [n, T+1] (after dual_ws) and go out as nested lists [N][K+1] of (n,) arrays."""
        N = N_per_set * (self.num_sets - 1)
        tau = np.append(0.0, radau_points(K))
        t = (np.arange(N)[:, None] + tau[None, :]).ravel() / N * zu0.t[-1]
        out = VehiclePrediction()
        out.t = t
        for name in ("x", "y", "psi", "v", "u_steer", "u_a", "u_steer_dot"):
            setattr(out, name, np.interp(t, zu0.t, getattr(zu0, name)))
        for name in ("l", "m"):
            arr = getattr(zu0, name)
            if arr is None:
                continue
            cols = np.stack([np.interp(t, zu0.t, row) for row in np.asarray(arr)], 0)  # [n, N*(K+1)]
            setattr(out, name, [[cols[:, i * (K + 1) + k] for k in range(K + 1)] for i in range(N)])
        return out

    # ---- interpolant of a collocation solution ---------------------------------------------------
    def get_interpolator(self, K: int, N: int, dt: float, opt: VehiclePrediction):
        """state: per interval the degree-K Lagrange polynomial through its K+1 collocation values,
        interval chosen right-continuously (t >= t_i), final value sum_k D[k] X[-1,k] held for
        t >= N dt; inputs: piecewise constant per collocation point."""
        tau = np.append(0.0, radau_points(K))
        _, _, D = self.collocation_coefficients(K)
        X = np.stack([np.reshape(getattr(opt, n), (N, K + 1)) for n in ("x", "y", "psi", "v", "u_steer")], -1)
        x_final = np.tensordot(D, X[-1], axes=(0, 0))
        denom = np.array([np.prod(np.delete(tau[j] - tau, j)) for j in range(K + 1)])
        t_in = np.asarray(opt.t, float)
        ua, uw = np.asarray(opt.u_a, float), np.asarray(opt.u_steer_dot, float)

        def state_interpolator(t):
            t = float(t)
            if t >= N * dt:
                return x_final.copy()
            i = min(max(int(np.floor(t / dt + 1e-12)), 0), N - 1)
            rel = (t - i * dt) / dt
            basis = np.array([np.prod(np.delete(rel - tau, j)) for j in range(K + 1)]) / denom
            return basis @ X[i]

        def input_interpolator(t):
            idx = min(int(np.searchsorted(t_in[1:], float(t), side="right")), len(ua) - 1)
            return np.array([ua[idx], uw[idx]])

        self.state_interpolator, self.input_interpolator = state_interpolator, input_interpolator

    def set_reference_trajectory(self, traj: VehiclePrediction):
        """Interpolators from a sampled trajectory (t, x, y, psi, v, u_steer, u_a, u_steer_dot): linear in
        the states, piecewise constant in the inputs, final sample held -- stands in for the
        collocation interpolant when a trajectory comes from elsewhere (a `state_ws` result, a recorded plan)."""
        t = np.asarray(traj.t, float)
        S = np.stack([np.asarray(getattr(traj, n), float) for n in ("x", "y", "psi", "v", "u_steer")], 1)
        ua, uw = np.asarray(traj.u_a, float), np.asarray(traj.u_steer_dot, float)

        def state_interpolator(tt):
            return np.array([np.interp(float(tt), t, S[:, c]) for c in range(5)])

        def input_interpolator(tt):
            idx = min(int(np.searchsorted(t[1:], float(tt), side="right")), len(ua) - 1)
            return np.array([ua[idx], uw[idx]])

        self.state_interpolator, self.input_interpolator = state_interpolator, input_interpolator

    def interpolate_states(self, time: np.ndarray) -> VehiclePrediction:
        if self.state_interpolator is None:
            raise RuntimeError("no interpolator: call get_solution/get_interpolator or set_reference_trajectory first")
        time = np.asarray(time, float)
        S = np.array([self.state_interpolator(t) for t in time]).reshape(len(time), 5)
        U = np.array([self.input_interpolator(t) for t in time]).reshape(len(time), 2)
        out = VehiclePrediction()
        out.t = time.copy()
        out.x, out.y, out.psi, out.v, out.u_steer = (S[:, c].copy() for c in range(5))
        out.u_a, out.u_steer_dot = U[:, 0].copy(), U[:, 1].copy()
        return out
