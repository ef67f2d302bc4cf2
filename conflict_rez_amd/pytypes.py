"""State structs exchanged across the planning / MPC call surface.

Field-compatible with the reference's `confrez/pytypes.py` (`PythonMsg` :13-90,
`VehicleState` :355-451, `VehiclePrediction` :456-553 and the nested messages :152-352):
same class names, field names, defaults, the frozen-field `TypeError` on unknown
attributes (:24-38) and `copy()` == deep copy (:89).  The classes are generated from the
field tables below; `tests/golden/pytypes_fields.json` pins them against the reference.
"""
import copy as _copy
import dataclasses as _dc

import numpy as np


class PythonMsg:
    """Base message: attributes that were not declared as fields cannot be created by
    assignment (guards against typos such as `state.xk = 1`)."""

    def __setattr__(self, key, value):
        if not hasattr(self, key):
            raise TypeError('Cannot add new field "%s" to frozen class %s' % (key, self))
        object.__setattr__(self, key, value)

    def copy(self):
        return _copy.deepcopy(self)

    def print(self, depth=0, name=None):
        pad = "  " * depth
        head = "%s%s (%s):\n" % (pad, name, type(self).__name__) if name else "%s%s:\n" % (pad, type(self).__name__)
        body = ""
        for key, val in vars(self).items():
            if isinstance(val, PythonMsg):
                body += val.print(depth=depth + 1, name=key)
            else:
                body += "%s  %s=%s\n" % (pad, key, val)
        if depth == 0:
            print(head + body)
            return None
        return head + body


def _msg(name, fields, doc, namespace=None):
    cls = _dc.make_dataclass(
        name,
        [(k, object, _dc.field(default=v)) for k, v in fields],
        bases=(PythonMsg,),
        namespace=namespace or {},
    )
    cls.__doc__ = doc
    cls.__module__ = __name__
    return cls


Position = _msg("Position", [("x", 0), ("y", 0), ("z", 0)], "global position")
VehicleActuation = _msg(
    "VehicleActuation",
    [("t", 0), ("u_a", 0), ("u_steer", 0), ("u_steer_dot", 0)],
    "acceleration, steering angle and steering rate commands",
)
BodyLinearVelocity = _msg(
    "BodyLinearVelocity",
    [("v_long", 0), ("v_tran", 0), ("v_n", 0), ("v", 0)],
    "body-frame velocity; `v` is the bicycle-model speed",
    {"mag": lambda self: float(np.sqrt(self.v_long**2 + self.v_tran**2 + self.v_n**2))},
)
BodyAngularVelocity = _msg("BodyAngularVelocity", [("w_phi", 0), ("w_theta", 0), ("w_psi", 0)], "body rates")
BodyLinearAcceleration = _msg("BodyLinearAcceleration", [("a_long", 0), ("a_tran", 0), ("a_n", 0)], "body accel")
BodyAngularAcceleration = _msg("BodyAngularAcceleration", [("a_phi", 0), ("a_theta", 0), ("a_psi", 0)], "body ang. accel")
OrientationEuler = _msg("OrientationEuler", [("phi", 0), ("theta", 0), ("psi", 0)], "roll, pitch, yaw")


def _q_from_yaw(self, yaw):
    self.qr, self.qi, self.qj, self.qk = float(np.cos(yaw / 2)), 0.0, 0.0, float(np.sin(yaw / 2))


def _q_to_yaw(self):
    return float(np.arctan2(2 * (self.qr * self.qk + self.qi * self.qj), 1 - 2 * (self.qj**2 + self.qk**2)))


OrientationQuaternion = _msg(
    "OrientationQuaternion",
    [("qr", 1), ("qi", 0), ("qj", 0), ("qk", 0)],
    "global orientation quaternion (real part first)",
    {"from_yaw": _q_from_yaw, "to_yaw": _q_to_yaw,
     "norm": lambda self: float(np.sqrt(self.qr**2 + self.qi**2 + self.qj**2 + self.qk**2))},
)
ParametricPose = _msg("ParametricPose", [("s", 0), ("x_tran", 0), ("n", 0), ("e_psi", 0)], "path-relative pose")
ParametricVelocity = _msg(
    "ParametricVelocity", [("ds", 0), ("dx_tran", 0), ("dn", 0), ("de_psi", 0)], "path-relative velocity"
)

_NESTED = (
    ("x", Position),
    ("v", BodyLinearVelocity),
    ("w", BodyAngularVelocity),
    ("a", BodyLinearAcceleration),
    ("aa", BodyAngularAcceleration),
    ("q", OrientationQuaternion),
    ("e", OrientationEuler),
    ("p", ParametricPose),
    ("pt", ParametricVelocity),
    ("u", VehicleActuation),
)


def _state_post_init(self):
    for key, cls in _NESTED:
        if getattr(self, key) is None:
            object.__setattr__(self, key, cls())


def _state_get_R(self, reverse=False):
    psi = -self.e.psi if reverse else self.e.psi
    c, s = np.cos(psi), np.sin(psi)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 0]])


VehicleState = _msg(
    "VehicleState",
    [("vehicle_id", 1), ("t", None)] + [(k, None) for k, _ in _NESTED],
    "complete vehicle state; pose in x.x,x.y,e.psi, speed in v.v, steering angle in u.u_steer",
    {"__post_init__": _state_post_init, "get_R": _state_get_R},
)

_PRED_FIELDS = (
    "t dt x y v v_x v_y a_x a_y l m psi psidot v_long v_tran a_long a_tran e_psi s x_tran "
    "u_a u_steer u_steer_dot lap_num local_state_covariance global_state_covariance"
).split()

VehiclePrediction = _msg(
    "VehiclePrediction",
    [(k, None) for k in _PRED_FIELDS],
    "trajectory arrays: t,x,y,psi,v and inputs u_a,u_steer(=delta),u_steer_dot(=w); l,m are the OBCA duals",
)
