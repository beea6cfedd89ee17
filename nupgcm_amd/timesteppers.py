"""BDF1 / BDF2 timesteppers - mirrors /root/reference/src/timesteppers.jl:7-122."""
from __future__ import annotations


class AbstractTimestepper:
    def __repr__(self):
        return f"{type(self).__name__}: t={self.t}, t_start={self.t_start}, t_stop={self.t_stop}, Δt={self.dt}"


class BDF1(AbstractTimestepper):
    """src/timesteppers.jl:7-29"""

    def __init__(self, *, t_start, t_stop, dt, t=None, adaptive=False, CFL_factor=0.8):
        self.t_start, self.t_stop = float(t_start), float(t_stop)
        self.t = float(t_start if t is None else t)
        self.dt = float(dt)
        self.adaptive, self.CFL_factor = bool(adaptive), float(CFL_factor)


class BDF2(AbstractTimestepper):
    """src/timesteppers.jl:36-63 (adaptive stepping is not implemented for BDF2 in the reference either)"""
    adaptive = False

    def __init__(self, *, t_start, t_stop, dt, t=None):
        self.t_start, self.t_stop = float(t_start), float(t_stop)
        self.t = float(t_start if t is None else t)
        self.dt = float(dt)


def update_t(ts):
    """update_t! - src/timesteppers.jl:80-83"""
    ts.t += ts.dt
    return ts


def update_dt(ts, device_fe, x_inv, h_cells=None, u_min=0.01):
    """update_Δt! - src/timesteppers.jl:108-122: Δt = CFL_factor * min_K h_K / max(max_q |u|, u_min) for BDF1; a no-op
    for BDF2.  (The reference evaluates this on every step for any BDF1, adaptive or not - src/model.jl:131.)  The
    reduction over cells and quadrature points runs on the device."""
    if isinstance(ts, BDF1):
        ts.dt = ts.CFL_factor * device_fe.cfl_ratio(x_inv, u_min=u_min, h_cells=h_cells)
    return ts
