"""Shared builders for the GPU parity tests: the product-side (nupgcm_amd) objects of the reference's regression
configurations, mirroring oracle.recipe.CONFIGS (test/bowl_*_tests.jl and examples/bowl_mixing.jl of the reference)."""
import os

import numpy as np

import nupgcm_amd as npg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
U_TAGS = ["bottom", "coastline", "surface"]
U_VALS = [(0, 0, 0), (0, 0, 0), (0, 0, 0)]
U_MASKS = [(True, True, True), (True, True, True), (False, False, True)]


def H(x, alpha):
    return alpha * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)


def kappa_bottom(alpha):
    return lambda x: 1e-2 + np.exp(-(x[..., 2] + H(x, alpha)) / (0.1 * alpha))


def product_config(name):
    """(params, forcings, b_diri_tags, b_diri_vals, dt, b0) for a named configuration"""
    a = 0.5
    if name == "bowl_mixing":
        prm = npg.Parameters(eps=2e-1, alpha=a, mu_rho=10.0, N2=1 / a, f=lambda x: 1 + 0.5 * x[..., 1], H=lambda x: H(x, a))
        frc = npg.Forcings(1.0, kappa_bottom(a), kappa_bottom(a), 0.0, 0.0, npg.SurfaceDirichletBC(0.0))
        return prm, frc, ["coastline", "surface"], [0.0, 0.0], 1e-4 * 10.0 / (a * 0.2) ** 2, None
    if name == "example":
        prm = npg.Parameters(eps=2e-1, alpha=a, mu_rho=1.0, N2=1 / a, f=lambda x: 1 + 0.5 * x[..., 1], H=lambda x: H(x, a))
        frc = npg.Forcings(1.0, kappa_bottom(a), kappa_bottom(a), 0.0, 0.0, npg.SurfaceDirichletBC(0.0))
        return prm, frc, ["coastline", "surface"], [0.0, 0.0], 1e-3, None
    if name == "bowl_diri":
        bs = lambda x: x[..., 1]
        prm = npg.Parameters(eps=np.sqrt(1e-1), alpha=a, mu_rho=1.0, N2=0.0, f=lambda x: 0.5 * x[..., 1], H=lambda x: H(x, a))
        frc = npg.Forcings(1.0, 1.0, 1.0, 0.0, 0.0, npg.SurfaceDirichletBC(bs))
        return prm, frc, ["coastline", "surface"], [bs, bs], 1e-1, bs
    if name == "bowl_wind":
        prm = npg.Parameters(eps=np.sqrt(1e-1), alpha=a, mu_rho=1.0, N2=0.0, f=lambda x: 0.5 * x[..., 1], H=lambda x: H(x, a))
        frc = npg.Forcings(1.0, kappa_bottom(a), kappa_bottom(a), lambda x: -1e-1 * np.cos(np.pi * x[..., 1] / 2), 0.0,
                           npg.SurfaceDirichletBC(0.0))
        return prm, frc, ["coastline", "surface"], [0.0, 0.0], 1e-1, lambda x: x[..., 2] / a
    if name == "bowl_surface_flux":
        prm = npg.Parameters(eps=np.sqrt(1e-1), alpha=a, mu_rho=1.0, N2=0.0, f=lambda x: 1 + 0 * x[..., 1], H=lambda x: H(x, a))
        frc = npg.Forcings(1.0, 1e-2, 1e-2, 0.0, 0.0, npg.SurfaceFluxBC(lambda x: 1e-3 * np.sin(np.pi * x[..., 0])))
        return prm, frc, [], [], 1e-1, lambda x: x[..., 2] / a
    raise KeyError(name)


def build_fe_data(name, mesh="mesh_bowl3D_h0.1", perms=None):
    prm, frc, btags, bvals, dt, b0 = product_config(name)
    mesh = npg.Mesh(os.path.join(GOLDEN, mesh + ".npz"))
    spaces = npg.Spaces(mesh, u_diri_tags=U_TAGS, u_diri_vals=U_VALS, u_diri_masks=U_MASKS, b_diri_tags=btags,
                        b_diri_vals=bvals)
    return npg.FEData(mesh, spaces, perms=perms), prm, frc, dt, b0


def build_model(name, mesh="mesh_bowl3D_h0.1", nsteps=50, scheme="BDF2", arch=None, **inv_kw):
    arch = arch or npg.GPU()
    fed, prm, frc, dt, b0 = build_fe_data(name, mesh)
    ts = (npg.BDF2 if scheme == "BDF2" else npg.BDF1)(t_start=0.0, t_stop=nsteps * dt, dt=dt)
    inv = npg.InversionToolkit(arch, fed, prm, frc, **inv_kw)
    evo = npg.EvolutionToolkit(arch, fed, prm, frc, ts)
    model = npg.Model(arch, prm, frc, fed, inv, evo, ts)
    if b0 is not None:
        npg.set_b(model, b0)
    return model


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)
