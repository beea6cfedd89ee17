"""Named workloads of the benchmark / examples: the reference's bowl meshes (committed as arrays under tests/golden/, see
tests/golden/make_fixtures.py) and their uniform refinements, with the parameters of /root/reference/examples/bowl_mixing.jl."""
from __future__ import annotations

import os

import numpy as np

from . import gmsh_io, refine
from .evolution import EvolutionToolkit
from .fe import FEData, Mesh, Spaces
from .inputs import Forcings, Parameters, SurfaceDirichletBC
from .inversion import InversionToolkit
from .model import Model
from .timesteppers import BDF2

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# label -> (committed base mesh, refinement levels).  h halves per level.
BOWL_MESHES = {
    "bowl3D_h0.1": ("mesh_bowl3D_h0.1", 0), "bowl3D_h0.08": ("mesh_bowl3D_h0.08", 0),
    "bowl3D_h0.05": ("mesh_bowl3D_h0.1", 1), "bowl3D_h0.04": ("mesh_bowl3D_h0.08", 1),
    "bowl3D_h0.025": ("mesh_bowl3D_h0.1", 2), "bowl3D_h0.02": ("mesh_bowl3D_h0.08", 2),
    "bowl3D_h0.0125": ("mesh_bowl3D_h0.1", 3), "bowl3D_h0.01": ("mesh_bowl3D_h0.08", 3),
}
ALPHA = 0.5


def bowl_mesh_model(label) -> gmsh_io.GmshModel:
    base, levels = BOWL_MESHES[label]
    m = gmsh_io.load_npz(os.path.join(_DATA, base + ".npz"))
    return refine.refine(m, levels, refine.bowl_projector(ALPHA)) if levels else m


def _H(x):
    return ALPHA * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)


def example_parameters(mu_rho=1.0):
    """examples/bowl_mixing.jl:35-52: eps = 0.2, alpha = 1/2, mu_rho = 1, N2 = 1/alpha, f = 1 + 0.5 y, bottom-enhanced
    mixing kappa = 1e-2 + exp(-(z + H)/(0.1 alpha)), no wind, b = 0 at the surface."""
    kap = lambda x: 1e-2 + np.exp(-(x[..., 2] + _H(x)) / (0.1 * ALPHA))
    prm = Parameters(eps=2e-1, alpha=ALPHA, mu_rho=mu_rho, N2=1 / ALPHA, f=lambda x: 1.0 + 0.5 * x[..., 1], H=_H)
    frc = Forcings(1.0, kap, kap, 0.0, 0.0, SurfaceDirichletBC(0.0))
    return prm, frc


def example_fe_data(mesh_model):
    mesh = Mesh(mesh_model)
    spaces = Spaces(mesh, u_diri_tags=["bottom", "coastline", "surface"], u_diri_vals=[(0, 0, 0)] * 3,
                    u_diri_masks=[(True, True, True), (True, True, True), (False, False, True)],
                    b_diri_tags=["coastline", "surface"], b_diri_vals=[0.0, 0.0])
    return FEData(mesh, spaces)


def example_model(arch, label_or_model, dt=1e-3, t_stop=1e9, **inv_kw):
    """The full model of examples/bowl_mixing.jl:171-190 on the named mesh (BDF2, dt = 1e-3)."""
    mm = bowl_mesh_model(label_or_model) if isinstance(label_or_model, str) else label_or_model
    fed = example_fe_data(mm)
    prm, frc = example_parameters()
    ts = BDF2(t_start=0.0, t_stop=t_stop, dt=dt)
    inv = InversionToolkit(arch, fed, prm, frc, **inv_kw)
    evo = EvolutionToolkit(arch, fed, prm, frc, ts)
    return Model(arch, prm, frc, fed, inv, evo, ts)
