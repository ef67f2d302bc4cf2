"""conflict_rez_amd -- MI355X-native batched OBCA trajectory-optimisation engine.

Drop-in for the collision-free planning hot path of XuShenLZ/conflict_rez
(`confrez/control`): same Python call surface (`Vehicle`, `VehicleFollower`,
`MultiDistributedFollower`) and `pytypes` structs, with the per-instance NLP solved by
hand-written gfx950 HIP kernels behind a C-ABI (`include/confrez_hip.h`).
"""
from .pytypes import VehiclePrediction, VehicleState  # noqa: F401
from .vehicle_types import VehicleBody, VehicleConfig  # noqa: F401
from .obstacle_types import GeofenceRegion, Polytope  # noqa: F401
