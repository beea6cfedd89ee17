"""ORACLE (test infrastructure) - the timestep recipe of /root/reference/src/model.jl:90-317 on top of fe_oracle.

`CONFIGS` holds the parameter sets of the reference's four regression scripts (test/bowl_*_tests.jl) plus the example's
(examples/bowl_mixing.jl:35-52,171); `setup()` builds every operator; `run()` does n steps of

    evolve!  (src/model.jl:213-285)   y = rhs_adv + theta rhs_diff + dt rhs_flux - (rhs_M + theta (rhs_h + rhs_v))
    invert!  (src/model.jl:302-317)   A [u;p] = B b + b0

with either the reference CPU() path's direct solves (src/iterative_solvers.jl:42-55; scipy SuperLU stands in for
UMFPACK) or the GPU() path's Krylov solves (krylov_oracle).  Quirks reproduced: the first step of a
BDF2 run uses the BDF1 left-hand side while the right-hand side already uses BDF2's theta; flux scaled by dt; u = 0
during step 1 unless the caller inverted first.  `first_step_lhs="bdf2"` is the compatibility switch that the exact
fixture bowl_surface_flux.jld2 (older revision) needs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import reverse_cuthill_mckee

from . import fe_oracle as fo
from . import krylov_oracle as ko

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _Model:
    pass


def load_mesh(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    m = _Model()
    import json
    m.dim = int(z["dim"])
    for k in ("coords", "node_phys", "cells", "facets", "facets_phys", "ridges", "ridges_phys"):
        setattr(m, k, z[k])
    m.periodic = z["periodic"] if "periodic" in z.files else None
    m.phys_names = json.loads(bytes(z["phys_names"]).decode())
    return m


def _H(x, alpha):
    return alpha * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)


def _kappa_bottom(alpha):
    return lambda x: 1e-2 + np.exp(-(x[..., 2] + _H(x, alpha)) / (0.1 * alpha))


def channel_basin_depth(x, alpha):
    """H((x, y, z)) of /root/reference/scratch/run.jl:54-97 (geom = :tub): channel of depth alpha W shoaling to the sill at
    y = -L/2 + L/4, basin alpha W (1 - ((x - W/2)/(W/2))^2), revolved parabola at the northern end; L = 2, W = 1."""
    X, Y = x[..., 0], x[..., 1]
    L, W = 2.0, 1.0
    Lc = L / 4
    Lf = 5 * Lc / 8
    H0 = alpha * W
    par = lambda s_, smax, szero: H0 * (1 - ((s_ - smax) / (szero - smax)) ** 2)
    Hb = par(X, W / 2, 0.0)
    Hc = np.where(Y <= -L / 2 + Lf, H0, par(Y, -L / 2 + Lf, -L / 2 + Lc))
    r = np.sqrt((X - W / 2) ** 2 + (Y - (L / 2 - W / 2)) ** 2)
    H = np.where(Y <= -L / 2 + Lc, np.maximum(Hc, Hb), np.where(Y <= L / 2 - W / 2, Hb, par(r, 0.0, W / 2)))
    return np.maximum(H, 0.0)


def channel_basin_config(surface="flux"):
    """Parameters of /root/reference/scratch/run.jl:28-172 (the production channel-basin run): alpha = 1/8, f = y,
    N2 = 0, P1 buoyancy, function-valued nu = 1 (hence the full-stress form), bottom-enhanced kappa, wind stress (:114),
    convection + eddy closures (:119-120), BDF1 with the adaptive CFL step (CFL_factor 0.8, :158).  surface="dirichlet" is
    run.jl's SurfaceDirichletBC (:116-118); surface="flux" is the BASELINE.json configs[4] variant, which swaps in a
    SurfaceFluxBC (no buoyancy Dirichlet tags)."""
    Om = 2 * np.pi / 86400
    a_e = 6.371e6
    beta = 2 * Om / a_e
    Ld = 2 * np.pi * a_e * 60 / 360
    f0 = beta * Ld
    H0, k0, Ke, N0, rho0, aT, g = 4e3, 1e-5, 1000.0, 1e-3, 1035.0, 2e-4, 9.81
    nu0 = Ke * f0 ** 2 / N0 ** 2
    tau0 = rho0 * N0 ** 2 * H0 ** 3 / Ld
    b0s = g * aT * 30 / (N0 ** 2 * H0)
    eps = np.sqrt(nu0 / f0 / H0 ** 2)
    mu_rho = (nu0 / k0) * (N0 * H0 / f0 / Ld) ** 2
    t0 = 1 / f0 / (N0 * H0 / f0 / Ld) ** 2
    alpha = 1 / 8
    kI, kB, d = 1.0, 1e2, 500 / 4000 * alpha
    kap = lambda x: kI + (kB - kI) * np.exp(-(x[..., 2] + channel_basin_depth(x, alpha)) / d)
    b_surface = lambda x: np.where(x[..., 1] > 0, 0.0, -b0s * x[..., 1] ** 2)
    cfg = dict(eps=eps, alpha=alpha, mu_rho=mu_rho, N2=0.0, f=lambda x: x[..., 1], nu=lambda x: 1.0 + 0 * x[..., 0],
               kappa=kap, tau_x=lambda x: np.where(x[..., 1] > -0.5, 0.0,
                                                   -0.2 / tau0 * (x[..., 1] + 1) * (x[..., 1] + 0.5) / 0.25 ** 2),
               b_order=1, dt=86400 / t0, conv=(0.2 / k0, 1e-3), eddy=(np.sqrt(1e-3), 10.0, 1.0), cfl_factor=0.8,
               b0=lambda x: b0s * x[..., 2] / alpha + b_surface(x) * np.exp(x[..., 2] / (alpha / 4)))
    if surface == "dirichlet":
        cfg.update(b_diri_tags=["coastline", "surface"], b_diri_fn=b_surface)
    else:
        cfg.update(b_diri_tags=[], b_diri_fn=None, surface_flux=lambda x: 1e-2 * b_surface(x))
    return cfg


U_TAGS = ["bottom", "coastline", "surface"]
U_MASKS = [(True, True, True), (True, True, True), (False, False, True)]

# parameter sets: test/bowl_mixing_tests.jl:16-32,67-68 ; test/bowl_dirichlet_tests.jl ; test/bowl_wind_tests.jl ;
# test/bowl_surface_flux_tests.jl ; examples/bowl_mixing.jl:35-52,171
CONFIGS = {
    "bowl_mixing": dict(eps=2e-1, alpha=0.5, mu_rho=10.0, N2=2.0, f=lambda x: 1 + 0.5 * x[..., 1], nu=1.0,
                        kappa="bottom", b_diri_tags=["coastline", "surface"], b_diri_fn=lambda x: 0 * x[..., 0],
                        dt=1e-4 * 10.0 / (0.5 * 0.2) ** 2, b0=None),
    "bowl_diri": dict(eps=np.sqrt(1e-1), alpha=0.5, mu_rho=1.0, N2=0.0, f=lambda x: 0.5 * x[..., 1], nu=1.0,
                      kappa=1.0, b_diri_tags=["coastline", "surface"], b_diri_fn=lambda x: x[..., 1],
                      dt=1e-1, b0=lambda x: x[..., 1]),
    "bowl_wind": dict(eps=np.sqrt(1e-1), alpha=0.5, mu_rho=1.0, N2=0.0, f=lambda x: 0.5 * x[..., 1], nu=1.0,
                      kappa="bottom", b_diri_tags=["coastline", "surface"], b_diri_fn=lambda x: 0 * x[..., 0],
                      tau_x=lambda x: -1e-1 * np.cos(np.pi * x[..., 1] / 2), dt=1e-1, b0=lambda x: x[..., 2] / 0.5),
    "bowl_surface_flux": dict(eps=np.sqrt(1e-1), alpha=0.5, mu_rho=1.0, N2=0.0, f=lambda x: 1 + 0 * x[..., 1], nu=1.0,
                              kappa=1e-2, b_diri_tags=[], b_diri_fn=None,
                              surface_flux=lambda x: 1e-3 * np.sin(np.pi * x[..., 0]), dt=1e-1,
                              b0=lambda x: x[..., 2] / 0.5),
    "example": dict(eps=2e-1, alpha=0.5, mu_rho=1.0, N2=2.0, f=lambda x: 1 + 0.5 * x[..., 1], nu=1.0,
                    kappa="bottom", b_diri_tags=["coastline", "surface"], b_diri_fn=lambda x: 0 * x[..., 0],
                    dt=1e-3, b0=None),
}


@dataclass
class System:
    name: str
    orc: fo.Oracle
    cfg: dict
    A: sp.csr_matrix
    B: sp.csr_matrix
    b0: np.ndarray
    M: sp.csr_matrix
    Kh: sp.csr_matrix
    Kv: sp.csr_matrix
    rhs_M: np.ndarray
    rhs_h: np.ndarray
    rhs_v: np.ndarray
    rhs_diff: np.ndarray
    rhs_flux: np.ndarray
    dt: float
    extra: dict = field(default_factory=dict)

    @property
    def c(self):
        return self.orc.alpha ** 2 * self.orc.eps ** 2 / self.orc.mu_rho

    def theta(self, scheme):
        """src/evolution.jl:187-193."""
        return self.dt * self.c * (1.0 if scheme == "BDF1" else 2.0 / 3.0)


def setup(name, mesh="mesh_bowl3D_h0.1", model=None, **override) -> System:
    """model: an already loaded / refined mesh model (anything with GmshModel's attributes) instead of a fixture name"""
    if name.startswith("channel_basin"):
        cfg = channel_basin_config("dirichlet" if name.endswith("dirichlet") else "flux")
    else:
        cfg = dict(CONFIGS[name])
    cfg.update(override)
    topo = fo.build_topo(load_mesh(mesh) if model is None else model)
    spc = fo.build_spaces(topo, U_TAGS, U_MASKS, cfg["b_diri_tags"], cfg["b_diri_fn"], b_order=cfg.get("b_order", 2))
    kap = _kappa_bottom(cfg["alpha"]) if isinstance(cfg["kappa"], str) else cfg["kappa"]
    orc = fo.Oracle(topo, spc, eps=cfg["eps"], alpha=cfg["alpha"], mu_rho=cfg["mu_rho"], N2=cfg["N2"], f=cfg["f"],
                    nu=cfg["nu"], kappa_h=kap, kappa_v=kap, tau_x=cfg.get("tau_x", 0.0), tau_y=cfg.get("tau_y", 0.0),
                    surface_flux=cfg.get("surface_flux"))
    M, rM = orc.M()
    Kh, rh = orc.K_h()
    Kv, rv = orc.K_v()
    # a function-valued nu selects the full-stress form (src/inversion.jl:172-181)
    A = orc.A_inversion(nu_q=fo._const_or_fn(cfg["nu"], orc.geo.xq)) if callable(cfg["nu"]) else orc.A_inversion()
    return System(name, orc, cfg, A, orc.B_inversion(), orc.b_inversion(), M, Kh, Kv, rM, rh, rv,
                  orc.rhs_diff(), orc.rhs_flux(), float(cfg["dt"]))


def rcm_perms(sysm: System):
    """src/dofs.jl:70-100: per-field RCM of the mass-matrix graphs; p_inversion = [p_u; nu + p_p].  Any valid RCM is
    acceptable to the reference (test/bowl_mixing_tests.jl:60 comment); scipy's is used."""
    s = sysm.orc.sp
    A = sysm.A
    Auu = A[:s.nu, :s.nu]
    p_u = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(Auu), symmetric_mode=False))
    App = (A[s.nu:, :s.nu] @ A[:s.nu, s.nu:])
    p_p = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(App), symmetric_mode=False))
    p_b = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(sysm.M), symmetric_mode=True))
    return np.concatenate([p_u, s.nu + p_p]), p_b


def cfl_dt(orc, u_free, cfl_factor, u_min=0.01):
    """update_dt! (src/timesteppers.jl:108-119): CFL_factor * min_K h_K / max(max_q |u|, u_min)"""
    un = orc.u_nodal(u_free)[orc.cn2]
    speed = np.linalg.norm(np.einsum("qi,cia->cqa", orc.N2q, un), axis=-1).max(axis=1)
    return cfl_factor * float((orc.h_cells() / np.maximum(speed, u_min)).min())


def run(sysm: System, nsteps, solver="direct", first_step_lhs="bdf1", invert_first=False, scheme="BDF2",
        krylov_kw=None, record=None, timer=None, cfl_factor=None, adaptive=False, conv=None, eddy=None):
    """Returns (u, p, b) free values in native order after `nsteps` steps from the configuration's initial condition.
    timer: optional dict; receives 'loop_seconds' = wall time of the step loop only (factorisations excluded, as the
    reference's CPU() path factorises at set-up time, src/inversion.jl:58, src/evolution.jl:152)."""
    import time as _time
    orc, s = sysm.orc, sysm.orc.sp
    b0fn = sysm.cfg["b0"]
    b = np.zeros(s.nb) if b0fn is None else orc.interpolate_b(b0fn)
    u = np.zeros(s.nu)
    p = np.zeros(s.np_)
    krylov_kw = dict(krylov_kw or {})
    nu = s.nu
    if solver == "direct":
        luA = spla.splu(sp.csc_matrix(sysm.A))
    else:
        h, _ = orc.precond_h()
        Pinv = 1.0 / h ** orc.topo.dim
        xinv = np.zeros(s.nu + s.np_)
        xb = np.zeros(s.nb)

    def invert(bv):
        nonlocal xinv
        y = sysm.B @ bv + sysm.b0
        if solver == "direct":
            x = luA.solve(y)
        else:
            x, st = ko.gmres(sysm.A, y, x0=xinv, M=Pinv, **krylov_kw)
            xinv = x
            if record is not None:
                record.append(("gmres", st["niter"], st["solved"]))
        return x[:nu], x[nu:]

    def alpha_bz(bv):
        """alpha d_z(N2 z + b) at the quadrature points (src/model.jl:229, 163)"""
        bn = orc.b_nodal(bv)[orc.cnb]
        return orc.alpha * (orc.N2 + np.einsum("cqi,ci->cq", orc.gradNb[..., 2], bn))

    kv0_q = None
    if conv is not None:                            # (kappa_c, N2min): src/inputs.jl:87-91
        kv0_q = fo._const_or_fn(orc.kappa_v, orc.geo.xq)

    if invert_first:
        u, p = invert(b)
    u_prev, b_prev = u.copy(), b.copy()
    lhs_cache = {}

    def lhs(theta):
        if theta not in lhs_cache:
            Amat = (sysm.M + theta * (sysm.Kh + sysm.Kv)).tocsr()
            lhs_cache[theta] = (Amat, spla.splu(sp.csc_matrix(Amat)) if solver == "direct" else 1.0 / Amat.diagonal())
        return lhs_cache[theta]

    if solver == "direct":
        lhs(sysm.theta("BDF1" if (scheme == "BDF2" and first_step_lhs == "bdf1") else scheme))
        lhs(sysm.theta(scheme))
    theta0 = sysm.theta(scheme)
    _t0 = _time.perf_counter()
    for i in range(1, nsteps + 1):
        if scheme == "BDF1" and cfl_factor is not None:
            # the reference updates dt on EVERY BDF1 step (src/model.jl:131) but rebuilds the LHS only when the
            # timestepper is adaptive (src/model.jl:251-261)
            sysm.dt = cfl_dt(orc, u, cfl_factor)
            if adaptive:
                lhs_cache.clear()
        if conv is not None:
            # evolve! with the convection closure (src/model.jl:229-261): kappa_v from the current b, then K_v, its lift,
            # rhs_diff and the LHS are rebuilt
            kq = kv0_q + conv[0] * (1 + np.tanh(-alpha_bz(b) / conv[1])) / 2
            sysm.Kv, sysm.rhs_v = orc.K_v(kappa=lambda x: kq)
            sysm.rhs_diff = orc.rhs_diff(kappa=lambda x: kq)
            lhs_cache.clear()
        theta_rhs = sysm.theta(scheme)
        if scheme == "BDF2" and i == 1 and first_step_lhs == "bdf1":
            theta_lhs = sysm.theta("BDF1")          # src/evolution.jl:110-111 + src/model.jl:134-137
        elif scheme == "BDF1" and cfl_factor is not None and not adaptive:
            theta_lhs = theta0                      # LHS still the one built with the initial dt (src/model.jl:251)
        else:
            theta_lhs = theta_rhs
        u_curr, b_curr = u.copy(), b.copy()
        radv = orc.advection_rhs(b, b_prev, u, u_prev, sysm.dt, scheme)
        y = radv + theta_rhs * sysm.rhs_diff + sysm.dt * sysm.rhs_flux \
            - (sysm.rhs_M + theta_rhs * (sysm.rhs_h + sysm.rhs_v))                      # src/model.jl:278
        Amat, fac = lhs(theta_lhs)
        if solver == "direct":
            b = fac.solve(y)
        else:
            b, st = ko.cg(Amat, y, x0=xb, M=fac, **{k: v for k, v in krylov_kw.items() if k in ("atol", "rtol")})
            xb = b
            if record is not None:
                record.append(("cg", st["niter"], st["solved"]))
        u, p = invert(b)
        if max(np.abs(u).max(), np.abs(b).max()) > 1e3 or not np.isfinite(u).all():
            raise RuntimeError("Blow-up detected, stopping simulation")                  # src/model.jl:149-153
        u_prev, b_prev = u_curr, b_curr
        if eddy is not None and i % 10 == 0:
            # nu_eddy from the new b, A re-assembled in the full-stress form (src/model.jl:160-170, src/inputs.jl:130-137)
            N2min, smoothing, nu_min = eddy
            fq = fo._const_or_fn(orc.f, orc.geo.xq)
            nu_e = fq * (fq / np.sqrt(N2min ** 2 + alpha_bz(b) ** 2))
            nuq = np.logaddexp(smoothing * nu_min, smoothing * nu_e) / smoothing
            sysm.A = orc.A_inversion(nu_q=nuq)
            if solver == "direct":
                luA = spla.splu(sp.csc_matrix(sysm.A))
    if timer is not None:
        timer["loop_seconds"] = _time.perf_counter() - _t0
    return u, p, b
