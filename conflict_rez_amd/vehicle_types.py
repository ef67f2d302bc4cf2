"""Vehicle body rectangle and actuator limits.

Same numbers as the reference's `confrez/vehicle_types.py`: `VehicleBody` :9-71 (rear-axle
origin, wb 2.5, front/rear overhang 0.8/0.6, width 1.8, body-frame H-rep rows +x,+y,-x,-y),
`VehicleConfig` :75-90.
"""
from dataclasses import dataclass, field

import numpy as np

from .pytypes import PythonMsg


@dataclass
class VehicleBody:
    hf: float = 0.8  # front overhang
    wb: float = 2.5  # wheelbase
    hr: float = 0.6  # rear overhang
    w: float = 1.8  # width
    offset: float = 0.0  # rear axle -> body centre
    lf: float = 0.0  # rear axle -> front bumper
    lr: float = 0.0  # rear axle -> rear bumper
    l: float = 0.0  # total length
    cr: float = -0.2  # circle approximation (unused on the OBCA path)
    cf: float = 2.45
    num_circles: int = 4

    def __post_init__(self):
        self.offset = self.wb / 2
        self.lf = self.wb + self.hf
        self.lr = self.hr
        self.l = self.lf + self.lr
        hw = self.w / 2
        self.V = np.array([[self.lf, hw], [-self.lr, hw], [-self.lr, -hw], [self.lf, -hw]])
        self.xy = np.vstack([self.V, self.V[:1]])
        self.A = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]])
        self.b = np.array([self.lf, hw, self.lr, hw])


@dataclass
class VehicleConfig(PythonMsg):
    v_max: float = field(default=2.5)
    v_min: float = field(default=-2.5)
    a_max: float = field(default=1.5)
    a_min: float = field(default=-1.5)
    delta_max: float = field(default=0.85)
    delta_min: float = field(default=-0.85)
    w_delta_max: float = field(default=1)
    w_delta_min: float = field(default=-1)
