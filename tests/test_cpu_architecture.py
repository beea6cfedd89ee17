"""BASELINE.json configs[0]: "bowl3D h = 0.1 mesh, CPU() architecture, 5 timesteps of bowl_mixing (plumbing, runs without a GPU)".

Model(CPU(), ...) drives libnupgcm_host.so - the host build of the C ABI (nupgcm_amd/csrc_host/: plain C++ / OpenMP element
kernels, SpMV, Krylov.jl's GMRES / CG) - and takes the reference's CPU() branches of iterative_solve! (src/iterative_solvers.jl:42-58):
a sparse LU as P wherever the reference factorises, `A \\ y` below 300 000 rows, Krylov otherwise.  The oracle is the CHECKER here
(its matrices, its direct-solve recipe), pinned by the reference's fixtures (tests/test_oracle_fixtures.py); the host library is
written independently of it (C++ restatement of csrc/fe.hip).  No GPU, no HIP library needed."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import nupgcm_amd as npg
from nupgcm_amd import _lib as L
from oracle import krylov_oracle as ko
from oracle import recipe as rc
from tests.helpers import build_fe_data, build_model, rel

pytestmark = pytest.mark.skipif(not os.path.exists(L.HOST_LIB_PATH), reason="libnupgcm_host.so not built (make -C nupgcm_amd/csrc_host)")


@pytest.fixture(scope="module")
def arch():
    a = npg.CPU()
    assert "host CPU" in a.ctx.name() and L.kind() == "host"
    return a


def _perm(A, pr, pc):
    return sp.csr_matrix(A)[pr][:, pc]


def test_host_library_exports_a_subset_of_the_abi_with_the_same_names():
    import ctypes as C
    lib = C.CDLL(L.HOST_LIB_PATH)
    declared = set(L.declared_symbols())
    have = {s for s in declared if hasattr(lib, s)}
    need = {"npg_ctx_create", "npg_vec_create", "npg_vec_upload_perm", "npg_vec_download_perm", "npg_vec_maxabs", "npg_csr_create",
            "npg_csr_create_from_csc", "npg_csr_combine", "npg_csr_inv_diag", "npg_spmv", "npg_gmres_solve", "npg_cg_solve",
            "npg_fe_create", "npg_fe_assemble_matrix", "npg_fe_evolution_rhs", "npg_fe_update_kappa_convection",
            "npg_fe_update_nu_eddy", "npg_fe_cfl_ratio", "npg_last_error"}
    assert need <= have and len(have) >= 60
    # nothing but ABI names is exported
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.HOST_LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert exported <= declared, exported - declared


def test_host_element_kernels_against_the_oracle(arch):
    """every matrix and vector of the set-up on bowl3D h = 0.1 (surface-flux configuration: non-zero flux, P2 buoyancy)"""
    S = rc.setup("bowl_surface_flux")
    fed, prm, frc, dt, b0 = build_fe_data("bowl_surface_flux")
    d = fed.dofs
    inv = npg.InversionToolkit(arch, fed, prm, frc)
    A = inv.solver.A.to_scipy_csr()
    ref = _perm(S.A, d.p_inversion, d.p_inversion)
    assert abs(A - ref).max() <= 1e-13 * abs(ref).max()
    B = inv.B.to_scipy_csr()
    refB = _perm(S.B, d.p_inversion, d.p_b)
    assert abs(B - refB).max() <= 1e-13 * abs(refB).max()
    assert np.linalg.norm(S.b0) == 0 and np.linalg.norm(inv.b.to_host()) == 0 or rel(inv.b.to_host(), S.b0[d.p_inversion]) < 1e-12
    ts = npg.BDF2(t_start=0.0, t_stop=1.0, dt=dt)
    evo = npg.EvolutionToolkit(arch, fed, prm, frc, ts)
    for got, want in ((evo.M, S.M), (evo.Kh, S.Kh), (evo.Kv, S.Kv)):
        refm = _perm(want, d.p_b, d.p_b)
        assert abs(got.to_scipy_csr() - refm).max() <= 1e-13 * abs(refm).max()
    assert rel(evo.rhs_flux.to_host(), S.rhs_flux[d.p_b]) < 1e-12
    assert isinstance(inv.solver.P, npg.iterative_solvers.LU) and isinstance(evo.solver.P, npg.iterative_solvers.LU)
    # advection right-hand side (BDF2) on random states
    rng = np.random.default_rng(5)
    b, bp = rng.standard_normal((2, d.nb))
    x, xp = rng.standard_normal((2, d.nu + d.np))
    dv = lambda v, p: npg.DeviceVector.from_host(arch.ctx, v, p)
    out = npg.DeviceVector(arch.ctx, d.nb)
    evo.fe.advection_rhs(L.NPG_BDF2, 0.1, prm.N2, dv(b, d.p_b), dv(bp, d.p_b), dv(x, d.p_inversion), dv(xp, d.p_inversion), out)
    assert rel(out.to_host(d.inv_p_b), S.orc.advection_rhs(b, bp, x[:d.nu], xp[:d.nu], 0.1, "BDF2")) < 1e-12


def test_five_timesteps_of_bowl_mixing_on_the_cpu_architecture(arch):
    """configs[0] itself: test/bowl_mixing_tests.jl's configuration, CPU(), 5 steps - direct solves on both systems, as the
    reference's CPU() path (lu + ldiv!); against the oracle's direct-solve recipe"""
    m = build_model("bowl_mixing", nsteps=5, arch=arch)
    npg.run(m)
    S = rc.setup("bowl_mixing")
    u, p, b = rc.run(S, 5)
    assert m.step_index == 6 and all(s[0].get("direct") and s[1].get("direct") for s in m.stats)
    assert rel(m.state.b, b) < 1e-9 and rel(m.state.u, u) < 1e-9 and rel(m.state.p, p) < 1e-9, \
        (rel(m.state.b, b), rel(m.state.u, u), rel(m.state.p, p))


def test_host_krylov_solvers_follow_krylov_jl(arch):
    """the Krylov branch (src/iterative_solvers.jl:58) on the host: restarted GMRES(20) with MGS and CG as Krylov.jl runs them -
    same iteration counts and residual histories as the oracle's restatement (both are MGS: unlike the device's CGS kernels the
    counts are equal, not within 10 %), same solutions"""
    S = rc.setup("bowl_surface_flux")
    fed, prm, frc, dt, b0 = build_fe_data("bowl_surface_flux")
    d, ctx = fed.dofs, arch.ctx
    Ah = sp.csr_matrix(_perm(S.A, d.p_inversion, d.p_inversion))
    A = npg.DeviceCSR.from_scipy(ctx, Ah)
    N = Ah.shape[0]
    h = fed.mesh.median_edge_length()
    y = Ah @ np.cos(np.arange(N, dtype=float)) * 1e-3
    ws = npg.GmresWorkspace(ctx, N, memory=20)
    x = npg.DeviceVector(ctx, N)
    st = ws.solve(A, npg.DeviceVector.from_host(ctx, y), x, npg.Diagonal(scalar=1 / h ** 3, n=N), atol=1e-6, rtol=1e-6, itmax=400)
    xo, so = ko.gmres(Ah, y, M=1 / h ** 3, atol=1e-6, rtol=1e-6, itmax=400)
    assert st["niter"] == so["niter"] == 400 and st["solved"] == so["solved"] == 0
    assert rel(ws.history(), np.asarray(so["residuals"])[:len(ws.history())]) < 1e-8 and rel(x.to_host(), xo) < 1e-8
    # CG on the evolution matrix with its Jacobi diagonal
    Mb = (S.M + S.theta("BDF2") * (S.Kh + S.Kv)).tocsr()
    Ab = sp.csr_matrix(_perm(Mb, d.p_b, d.p_b))
    rhs = Ab @ np.sin(np.arange(d.nb, dtype=float))
    cg = npg.CgWorkspace(ctx, d.nb)
    xb = npg.DeviceVector(ctx, d.nb)
    sc = cg.solve(npg.DeviceCSR.from_scipy(ctx, Ab), npg.DeviceVector.from_host(ctx, rhs), xb,
                  npg.Diagonal(npg.DeviceVector.from_host(ctx, 1 / Ab.diagonal())), atol=1e-10, rtol=1e-10)
    xbo, sco = ko.cg(Ab, rhs, M=1 / Ab.diagonal(), atol=1e-10, rtol=1e-10)
    assert sc["solved"] == 1 and sc["niter"] == sco["niter"] and rel(xb.to_host(), xbo) < 1e-9


def test_closures_on_the_cpu_architecture_take_the_backslash_branch(arch):
    """convection closure + adaptive BDF1 step on CPU(): the evolution LHS changes every step, so P stays the Jacobi diagonal and
    iterative_solve! takes `x .= A \\\\ y` (n < 300 000, src/iterative_solvers.jl:49-55); kappa_v, K_v, rhs_diff, the LHS and the CFL
    step come from the host library; 3 steps against the oracle"""
    from nupgcm_amd.inputs import ConvectionParameterization
    fed, prm, frc, dt, b0 = build_fe_data("bowl_diri")
    frc.conv_param = ConvectionParameterization(kappa_c=0.5, N2min=0.5)
    ts = npg.BDF1(t_start=0.0, t_stop=1e9, dt=dt, adaptive=True, CFL_factor=0.5)
    inv = npg.InversionToolkit(arch, fed, prm, frc)
    evo = npg.EvolutionToolkit(arch, fed, prm, frc, ts)
    assert isinstance(evo.solver.P, npg.Diagonal)
    m = npg.Model(arch, prm, frc, fed, inv, evo, ts)
    npg.set_b(m, b0)
    npg.run(m, n_steps=3)
    S = rc.setup("bowl_diri")
    u, p, b = rc.run(S, 3, solver="direct", scheme="BDF1", cfl_factor=0.5, adaptive=True, conv=(0.5, 0.5))
    assert abs(ts.dt - S.dt) <= 1e-9 * S.dt
    assert rel(m.state.b, b) < 1e-9 and rel(m.state.u, u) < 1e-9, (rel(m.state.b, b), rel(m.state.u, u))
