"""Cubic Bezier pose interpolation used for the spline initial guess.

Behaviour of the reference's `BezierPlanner.interpolate` (`confrez/control/bezier.py:22-56`):
control points P0=start, P1=start+d*heading(start), P2=end-d*heading(end), P3=end with
d = |end-start| / offset; N samples at t = linspace(0,1,N,endpoint=False); yaw = direction of
the curve tangent.  Written in closed form (Bernstein basis as a matrix product).
"""
import numpy as np


class BezierPlanner:
    def __init__(self, offset: float):
        self.offset = offset

    @staticmethod
    def _basis(t):
        t = np.asarray(t, float)[:, None]
        return np.hstack([(1 - t) ** 3, 3 * t * (1 - t) ** 2, 3 * t**2 * (1 - t), t**3])

    @staticmethod
    def _dbasis(t):
        t = np.asarray(t, float)[:, None]
        return np.hstack([(1 - t) ** 2, 2 * t * (1 - t), t**2])

    def control_points(self, start_state, end_state):
        s = np.array([start_state.x.x, start_state.x.y], float)
        e = np.array([end_state.x.x, end_state.x.y], float)
        d = np.hypot(*(s - e)) / self.offset
        hs = np.array([np.cos(start_state.e.psi), np.sin(start_state.e.psi)])
        he = np.array([np.cos(end_state.e.psi), np.sin(end_state.e.psi)])
        return np.array([s, s + d * hs, e - d * he, e])

    def interpolate(self, start_state, end_state, N):
        """[N,3] rows (x, y, yaw); the end state itself is not included."""
        cp = self.control_points(start_state, end_state)
        t = np.linspace(0, 1, N, endpoint=False)
        xy = self._basis(t) @ cp
        tangent = self._dbasis(t) @ (3 * np.diff(cp, axis=0))
        return np.column_stack([xy, np.arctan2(tangent[:, 1], tangent[:, 0])])
