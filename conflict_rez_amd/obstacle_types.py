"""Region and polytope types used by the OBCA constraints.

`GeofenceRegion` mirrors the reference's `confrez/obstacle_types.py:10-25`.
`Polytope` stands in for `pytope.Polytope` as the reference uses it
(`compute_sets.py:7,38-39,136,265-328`): built from 2-D vertices, exposes the H-rep
`A p <= b` with unit-norm rows in qhull facet order, and `P + offset` translates.
"""
from dataclasses import dataclass, field

import numpy as np
from scipy.spatial import ConvexHull


@dataclass
class GeofenceRegion:
    x_max: float = field(default=13 * 2.5)
    x_min: float = field(default=2.5)
    y_max: float = field(default=11 * 2.5)
    y_min: float = field(default=3 * 2.5)

    def xy(self):
        c = [(self.x_max, self.y_max), (self.x_max, self.y_min), (self.x_min, self.y_min), (self.x_min, self.y_max)]
        return np.array(c + c[:1])


class Polytope:
    """Bounded convex polygon {p : A p <= b}."""

    def __init__(self, V=None, A=None, b=None):
        if V is not None:
            self.V = np.asarray(V, dtype=float)
            eq = ConvexHull(self.V).equations  # rows [n_x, n_y, offset], n unit, n.p + offset <= 0 inside
            self.A = eq[:, :2].copy()
            self.b = -eq[:, 2].copy()
        else:
            self.A, self.b = np.asarray(A, float), np.asarray(b, float)
            self.V = None

    def __add__(self, offset):
        offset = np.asarray(offset, dtype=float).reshape(2)
        out = Polytope(A=self.A.copy(), b=self.b + self.A @ offset)
        out.V = None if self.V is None else self.V + offset
        return out

    def contains(self, p, tol=1e-9):
        return bool(np.all(self.A @ np.asarray(p, float) <= self.b + tol))

    @staticmethod
    def from_box(xmin, xmax, ymin, ymax):
        return Polytope([[xmin, ymin], [xmin, ymax], [xmax, ymax], [xmax, ymin]])
