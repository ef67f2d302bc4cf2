"""The object `MultiVehiclePlanner.solve_final_problem_obca` hands to every vehicle's `setup_single_final_problem(opti=, dt=)`
(reference confrez/control/multi_vehicle_planner.py:365-386: one `ca.Opti()` and one `opti.variable()` for the shared interval
length dt, every vehicle adds its collocation problem to it, :387 sums the costs, :389-456 adds the vehicle-vehicle rows,
:458-466 solves).  Here the shared problem is a list of the vehicles' problem descriptions; `solve` assembles and solves
it on the GPU (`cfz_joint_colloc`: one banded interior-point problem, csrc/cfz_colloc.inl)."""
from typing import List, Optional, Sequence, Tuple


class SharedDt:
    """The shared decision variable `dt = opti.variable()` (:366): identity only, plus its initial value."""

    def __init__(self, owner):
        self.owner, self.initial = owner, None


class JointOpti:
    def __init__(self):
        self.problems: List[dict] = []
        self._dt: Optional[SharedDt] = None

    def variable(self) -> SharedDt:
        """`dt = opti.variable()` (:366).  One scalar variable: the interval length every vehicle's plan runs on."""
        if self._dt is not None:
            raise RuntimeError("the joint problem has one shared variable (dt)")
        self._dt = SharedDt(self)
        return self._dt

    def set_initial(self, var: SharedDt, value: float):
        """`opti.set_initial(dt, dt0)` (:367)."""
        if var is not self._dt:
            raise ValueError("not this problem's variable")
        var.initial = float(value)

    def add(self, problem: dict, dt: SharedDt):
        """A vehicle's collocation problem on the shared dt (what `setup_single_final_problem(opti=opti, dt=dt)` does)."""
        if dt is not self._dt:
            raise ValueError("dt must be the variable created by this problem's variable()")
        self.problems.append(problem)

    def solve(self, pairs: Sequence[Tuple[int, int]] = None, **options):
        """`opti.solve()` (:466) of the joint problem: cost sum_a J_a (:387), vehicle-vehicle separation for `pairs` (indices
        into the order in which the vehicles were added; None = all pairs, :56-58).  Returns the result of
        `engine.joint_colloc`: dict(traj per vehicle [N_a, 6, 7], dt, status, iters, cost)."""
        from ..engine import joint_colloc

        if not self.problems:
            raise RuntimeError("no vehicle has been added to the joint problem")
        if self._dt is None or self._dt.initial is None:
            raise RuntimeError("set_initial(dt, dt0) has not been called (:367)")
        p = self.problems
        return joint_colloc(p[0]["spec"], [q["init_pose"] for q in p], [q["tube"] for q in p], [q["guess"] for q in p], self._dt.initial,
                            [q["final_heading"] for q in p], pairs=pairs, N_per_set=p[0]["N_per_set"], shrink_tube=p[0]["shrink_tube"], **options)
