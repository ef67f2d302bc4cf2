"""Oracle (test infrastructure): kinematic-bicycle model, numpy restatement.

Follows the reference's `confrez/control/dynamic_model.py`:
  * `bicycle_ct`      <- `kinematic_bicycle_ct`  :5-27   (state order x,y,psi,v,delta; input a,w)
  * `bicycle_rk4`     <- `kinematic_bicycle_rk`  :30-58  (classical RK4, M=4 sub-steps, h=dt/M)
  * `plant_step`      <- `simulator`             :61-93  (IDAS over [0,dt]; restated as RK4 with
                                                         10 sub-steps, 6e-11 from the converged solution)
All functions are vectorised over a leading batch axis.
"""
import numpy as np


def bicycle_ct(z, u, wb):
    """zdot = f(z,u).  z[...,5] = (x,y,psi,v,delta), u[...,2] = (a,w)."""
    psi, v, delta = z[..., 2], z[..., 3], z[..., 4]
    out = np.empty_like(z)
    out[..., 0] = v * np.cos(psi)
    out[..., 1] = v * np.sin(psi)
    out[..., 2] = v / wb * np.tan(delta)
    out[..., 3] = u[..., 0]
    out[..., 4] = u[..., 1]
    return out


def bicycle_ct_jac(z, u, wb):
    """Returns (f, df/dz [...,5,5], df/du [...,5,2])."""
    psi, v, delta = z[..., 2], z[..., 3], z[..., 4]
    c, s, t = np.cos(psi), np.sin(psi), np.tan(delta)
    f = np.empty_like(z)
    f[..., 0] = v * c
    f[..., 1] = v * s
    f[..., 2] = v / wb * t
    f[..., 3] = u[..., 0]
    f[..., 4] = u[..., 1]
    fz = np.zeros(z.shape + (5,))
    fz[..., 0, 2] = -v * s
    fz[..., 0, 3] = c
    fz[..., 1, 2] = v * c
    fz[..., 1, 3] = s
    fz[..., 2, 3] = t / wb
    fz[..., 2, 4] = v / wb * (1.0 + t * t)
    fu = np.zeros(z.shape + (2,))
    fu[..., 3, 0] = 1.0
    fu[..., 4, 1] = 1.0
    return f, fz, fu


def bicycle_rk4(z, u, dt, wb, M=4):
    """Discrete map z+ = F(z,u): M classical RK4 sub-steps of length dt/M."""
    h = dt / M
    zk = np.array(z, dtype=float, copy=True)
    for _ in range(M):
        a1 = bicycle_ct(zk, u, wb)
        a2 = bicycle_ct(zk + h * a1 / 2, u, wb)
        a3 = bicycle_ct(zk + h * a2 / 2, u, wb)
        a4 = bicycle_ct(zk + h * a3, u, wb)
        zk = zk + h / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
    return zk


def bicycle_rk4_jac(z, u, dt, wb, M=4):
    """Returns (F, dF/dz [...,5,5], dF/du [...,5,2]) by forward sensitivity through RK4."""
    h = dt / M
    zk = np.array(z, dtype=float, copy=True)
    shp = zk.shape
    Sz = np.broadcast_to(np.eye(5), shp + (5,)).copy()
    Su = np.zeros(shp + (2,))

    def stage(zz, dzz, duu):
        f, fz, fu = bicycle_ct_jac(zz, u, wb)
        return f, fz @ dzz, fz @ duu + fu

    for _ in range(M):
        a1, a1z, a1u = stage(zk, Sz, Su)
        a2, a2z, a2u = stage(zk + h / 2 * a1, Sz + h / 2 * a1z, Su + h / 2 * a1u)
        a3, a3z, a3u = stage(zk + h / 2 * a2, Sz + h / 2 * a2z, Su + h / 2 * a2u)
        a4, a4z, a4u = stage(zk + h * a3, Sz + h * a3z, Su + h * a3u)
        zk = zk + h / 6 * (a1 + 2 * a2 + 2 * a3 + a4)
        Sz = Sz + h / 6 * (a1z + 2 * a2z + 2 * a3z + a4z)
        Su = Su + h / 6 * (a1u + 2 * a2u + 2 * a3u + a4u)
    return zk, Sz, Su


def plant_step(z, u, dt, wb, substeps=10):
    """Plant integration over one control interval (stand-in for CasADi's IDAS integrator)."""
    return bicycle_rk4(z, u, dt, wb, M=substeps)
