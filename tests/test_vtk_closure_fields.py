"""save_vtk's derived fields (src/IO.jl:31-57: alpha*b_z, nu, kappa_v): nodal recovery of d/dz of the P2 buoyancy and the
closure formulas of src/inputs.jl:87-91, 130-137, on the host (no GPU)."""
from types import SimpleNamespace as NS

import numpy as np

from nupgcm_amd import inputs, io, workloads


def test_closure_fields_at_the_nodes():
    fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
    x = fed.mesh.node_coords
    b = 2.0 * x[:, 2] + 0.5 * x[:, 0] + 0.3 * x[:, 2] ** 2          # in the P2 space: b_z = 2 + 0.6 z exactly
    f = lambda X: 1 + 0.5 * X[:, 1]                                   # noqa: E731
    frc = NS(eddy_param=inputs.EddyParameterization(f=f, N2min=1e-2, is_on=True),
             conv_param=inputs.ConvectionParameterization(kappa_c=1.0, N2min=1e-3, is_on=True), nu=1.0,
             kappa_v=lambda X: 1e-2 + np.exp(-(X[:, 2] + 0.5) / 0.05))
    model = NS(fe_data=fed, params=NS(alpha=0.5), forcings=frc)
    abz, nu, kv = io._nodal_closure_fields(model, b)
    ref = 0.5 * (2 + 0.6 * x[:, 2])
    assert np.abs(abz - ref).max() < 1e-12
    v = f(x) * f(x) / np.sqrt(1e-4 + ref ** 2)
    assert np.abs(nu - np.log(np.exp(10.0) + np.exp(10 * v)) / 10).max() < 1e-12       # LogSumExp limiter, nu_min = 1
    kv_ref = 1e-2 + np.exp(-(x[:, 2] + 0.5) / 0.05) + 1.0 * (1 + np.tanh(-ref / 1e-3)) / 2
    assert np.abs(kv - kv_ref).max() < 1e-12
    # closures off: the forcing functions themselves
    frc.eddy_param.is_on = frc.conv_param.is_on = False
    _, nu0, kv0 = io._nodal_closure_fields(model, b)
    assert np.array_equal(nu0, np.ones(len(x))) and np.abs(kv0 - (1e-2 + np.exp(-(x[:, 2] + 0.5) / 0.05))).max() < 1e-15
