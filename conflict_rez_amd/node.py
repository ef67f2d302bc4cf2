"""One vehicle of the distributed MPC as a message-driven node (mirror of the reference's ROS2 deployment,
`ros2_ws/src/confrez_ros/src/vehicle_node.py:101-190`, message `msg/VehiclePredictionMsg.msg`, field transfer
`base_node.py:209-282`), without ROS: the transport is whatever `bus` object is passed in.

Protocol of the reference, kept here:
  * every node publishes its own prediction (x, y, psi only, :151-157, :176-186) on `/<agent>/pred` and a Bool on
    `/<agent>/info` at every timer tick (:167-169);
  * a node steps only when it has heard `info == True` from every other vehicle (:171);
  * a received prediction replaces `vehicle.others_pred[other]` as it arrives (:154-163), so within one tick later
    nodes already see the new predictions of earlier ones (unlike the Jacobi exchange of
    `MultiDistributedFollower.solve`, which `VehicleShardedExchange` / `cfz_loop_*` implement).

`InProcessBus` delivers synchronously in publish order and is what the tests and single-process deployments use; a
ROS2 binding only has to provide `publish(topic, msg)` / `subscribe(topic, callback)` with the same two message types.
"""
from array import array
from dataclasses import dataclass, field
from typing import Callable, Dict, List

import numpy as np

from .pytypes import PythonMsg, VehiclePrediction

_ARRAYS = ("t", "x", "y", "v", "v_x", "v_y", "a_y", "a_x", "psi", "psidot", "s", "x_tran", "v_long", "v_tran",
           "a_long", "a_tran", "e_psi", "u_a", "u_steer", "u_steer_dot")


@dataclass
class Header:
    stamp: float = 0.0
    frame_id: str = ""


@dataclass
class VehiclePredictionMsg:
    """Field for field `VehiclePredictionMsg.msg`: a header, 20 float64 arrays, dt and lap_num."""
    header: Header = field(default_factory=Header)
    dt: float = 0.0
    lap_num: float = 0.0
    t: array = field(default_factory=lambda: array("d"))
    x: array = field(default_factory=lambda: array("d"))
    y: array = field(default_factory=lambda: array("d"))
    v: array = field(default_factory=lambda: array("d"))
    v_x: array = field(default_factory=lambda: array("d"))
    v_y: array = field(default_factory=lambda: array("d"))
    a_y: array = field(default_factory=lambda: array("d"))
    a_x: array = field(default_factory=lambda: array("d"))
    psi: array = field(default_factory=lambda: array("d"))
    psidot: array = field(default_factory=lambda: array("d"))
    s: array = field(default_factory=lambda: array("d"))
    x_tran: array = field(default_factory=lambda: array("d"))
    v_long: array = field(default_factory=lambda: array("d"))
    v_tran: array = field(default_factory=lambda: array("d"))
    a_long: array = field(default_factory=lambda: array("d"))
    a_tran: array = field(default_factory=lambda: array("d"))
    e_psi: array = field(default_factory=lambda: array("d"))
    u_a: array = field(default_factory=lambda: array("d"))
    u_steer: array = field(default_factory=lambda: array("d"))
    u_steer_dot: array = field(default_factory=lambda: array("d"))

    def get_fields_and_field_types(self):
        """As the generated ROS2 class: field name -> type string, declaration order of the .msg file."""
        out = {"header": "std_msgs/Header", "t": "sequence<double>", "dt": "double"}
        out.update({k: "sequence<double>" for k in _ARRAYS[1:]})
        out["lap_num"] = "double"
        return out


@dataclass
class Bool:
    data: bool = False


def populate_msg(msg, data):
    """`MPClabNode.populate_msg` (base_node.py:209-256): every attribute of `data` that `msg` also has and that is not
    None is written into `msg`, converted to the destination's type (array('d') for sequences)."""
    for key in vars(data):
        if not hasattr(msg, key):
            continue
        new = getattr(data, key)
        if isinstance(new, PythonMsg):
            populate_msg(getattr(msg, key), new)
            continue
        if new is None:
            continue
        target = type(getattr(msg, key))
        setattr(msg, key, new if type(new) is target else (array("d", np.asarray(new, float).ravel()) if target is array else target(new)))
    return msg


def unpack_msg(msg, data):
    """`MPClabNode.unpack_msg` (base_node.py:258-282): every message field except the header that `data` knows."""
    for key in msg.get_fields_and_field_types().keys():
        if key == "header" or not hasattr(data, key):
            continue
        setattr(data, key, getattr(msg, key))


class InProcessBus:
    """Topic -> callbacks; `publish` calls the subscribers at once, in subscription order."""

    def __init__(self):
        self.subs: Dict[str, List[Callable]] = {}
        self.published: Dict[str, int] = {}

    def subscribe(self, topic: str, callback: Callable):
        self.subs.setdefault(topic, []).append(callback)

    def publish(self, topic: str, msg):
        self.published[topic] = self.published.get(topic, 0) + 1
        for cb in self.subs.get(topic, []):
            cb(msg)


class VehicleNode:
    """`VehicleNode` (vehicle_node.py:68-190) around an already planned `VehicleFollower` (its reference set, its
    controller set up against `others`); `timer_callback` is the 10 Hz tick."""

    def __init__(self, vehicle, num_vehicles: int, bus):
        self.vehicle, self.bus = vehicle, bus
        self.agent = vehicle.agent
        self.others = [f"vehicle_{i}" for i in range(num_vehicles) if f"vehicle_{i}" != self.agent]
        self.others_info = {o: False for o in self.others}
        for other in self.others:
            bus.subscribe(f"/{other}/pred", self.vehicle_pred_cb(other))
            bus.subscribe(f"/{other}/info", self.vehicle_info_cb(other))
        self.steps = 0

    def publish_prediction(self):
        pred = VehiclePrediction()
        pred.x, pred.y, pred.psi = (array("d", getattr(self.vehicle.pred, n)) for n in ("x", "y", "psi"))
        self.bus.publish(f"/{self.agent}/pred", populate_msg(VehiclePredictionMsg(), pred))

    def vehicle_pred_cb(self, other):
        def callback(msg):
            pred = VehiclePrediction()
            unpack_msg(msg, pred)
            pred.x, pred.y, pred.psi = np.array(pred.x), np.array(pred.y), np.array(pred.psi)
            self.vehicle.others_pred[other] = pred

        return callback

    def vehicle_info_cb(self, other):
        def callback(msg):
            self.others_info[other] = msg.data

        return callback

    def timer_callback(self):
        self.bus.publish(f"/{self.agent}/info", Bool(True))
        if all(self.others_info.values()):
            self.vehicle.step()
            self.steps += 1
            self.publish_prediction()
