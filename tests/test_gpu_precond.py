"""General preconditioners of the inversion on the GPU (csrc/mg.hip) through the C ABI: flexible GMRES, the multigrid V-cycle
(new work) and the reference's BlockDiagonalPreconditioner (src/preconditioners.jl:53-125).

What is checked: (1) FGMRES against the host restatement iteration by iteration; (2) one V-cycle against the host restatement
built from the SAME operators (operation-level parity, 1e-10); (3) solutions against the fixture-pinned oracle's direct solve
within the stopping tolerance; (4) iteration counts far below the Diagonal(1/h^3) path's, on the reference's own meshes."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu

import nupgcm_amd as npg  # noqa: E402
from nupgcm_amd import multigrid as mgm  # noqa: E402
from nupgcm_amd import workloads  # noqa: E402
from oracle import mg_oracle as mo  # noqa: E402
from oracle import recipe as rc  # noqa: E402
from tests.helpers import rel  # noqa: E402


@pytest.fixture(scope="module")
def arch():
    a = npg.GPU()
    a.ctx
    return a


def test_fgmres_without_preconditioner_matches_restatement(arch):
    rng = np.random.default_rng(2)
    n = 3000
    A = sp.random(n, n, density=0.003, random_state=3, format="csr") + sp.diags(4.0 + rng.random(n))
    A = sp.csr_matrix(A)
    b = rng.standard_normal(n)
    x0 = 0.1 * rng.standard_normal(n)
    ctx = arch.ctx
    Ad = npg.DeviceCSR.from_scipy(ctx, A)
    ws = npg.FgmresWorkspace(ctx, n, memory=7)
    x = npg.DeviceVector.from_host(ctx, x0)
    st = ws.solve(Ad, npg.DeviceVector.from_host(ctx, b), x, None, atol=0.0, rtol=1e-10, scale=3.0)
    xr, sr = mo.fgmres(A, b, None, x0=x0, m=7, scale=3.0, atol=0.0, rtol=1e-10)
    assert st["solved"] == 1 and st["niter"] == sr["niter"] and st["npass"] == -(-st["niter"] // 7)
    h = ws.history()
    assert len(h) == st["niter"] + 1 and np.allclose(h, sr["residuals"], rtol=1e-6, atol=1e-13 * h[0])
    assert rel(x.to_host(), xr) < 1e-9 and rel(A @ x.to_host(), b) < 2e-10
    assert abs(st["rnorm"] - 3.0 * np.linalg.norm(b - A @ x.to_host())) < 1e-6 * st["rnorm0"] * 1e-4 + 1e-12
    # itmax is honoured and reported
    x2 = npg.DeviceVector(ctx, n)
    st2 = ws.solve(Ad, npg.DeviceVector.from_host(ctx, b), x2, None, atol=0.0, rtol=1e-14, itmax=5)
    assert st2["solved"] == 0 and st2["niter"] == 5 and st2["status"] == 2


@pytest.fixture(scope="module")
def two_level(arch):
    """bowl3D h = 0.05 over h = 0.1 (the reference's mesh and its red refinement), example parameters"""
    prm, frc = workloads.example_parameters()
    hier = [workloads.example_fe_data(m) for m in workloads.bowl_hierarchy_models("bowl3D_h0.05")]
    A = npg.build_A_inversion(arch, hier[-1], prm, frc.nu)
    As = A.to_scipy_csr()
    P = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=A, block_nodes=False, coarse_dense=False)
    return prm, frc, hier, A, As, P


def _host_levels(arch, prm, frc, hier):
    levels = []
    for k, fed in enumerate(hier):
        Ak = npg.build_A_inversion(arch, fed, prm, frc.nu).to_scipy_csr()
        Dinv = mgm.node_block_inverse(Ak[:fed.dofs.nu, :fed.dofs.nu], fed.dofs.n_full, fed.dofs.n_surf)
        levels.append(mo.Level(Ak, fed.dofs.nu, Dinv, P=None if k == 0 else mgm.prolongation(hier[k - 1], fed)))
    return levels


def test_vcycle_matches_host_restatement(arch, two_level):
    prm, frc, hier, A, As, P = two_level
    n = As.shape[0]
    levels = _host_levels(arch, prm, frc, hier)
    r = np.sin(np.arange(n) * 0.37) + 0.1
    for params in (dict(), dict(omega=3.0, jacobi_weight=0.6, schur_sweeps=2, nu1=1, nu2=3, coarse_sweeps=5),
                   dict(nu1=0, nu2=2, schur_sweeps=1)):
        P.set_params(**params)
        z = P.apply(npg.DeviceVector.from_host(arch.ctx, r), npg.DeviceVector(arch.ctx, n)).to_host()
        kw = dict(omega=2.5, jw=0.7, sweeps=3, nu1=2, nu2=2, coarse=20)
        kw.update({dict(jacobi_weight="jw", schur_sweeps="sweeps", coarse_sweeps="coarse").get(k, k): v
                   for k, v in params.items()})
        zr = mo.vcycle(levels, len(levels) - 1, r, **kw)
        assert rel(z, zr) < 1e-10, (params, rel(z, zr))
    P.set_params()


def test_multigrid_fgmres_solves_the_inversion(arch, two_level):
    """A [u; p] = B b + b0 on bowl3D h = 0.05 (134 866 unknowns): same stopping rule as the reference path, ~20 iterations
    instead of thousands; the solution satisfies the system to the tolerance and agrees with the Diagonal(1/h^3) GMRES
    solution within the two solvers' tolerances."""
    prm, frc, hier, A, As, P = two_level
    fed = hier[-1]
    n = As.shape[0]
    ctx = arch.ctx
    B = npg.build_B_inversion(arch, fed, prm).to_scipy_csr()
    b = fed.spaces.interpolate_b(lambda x: 0.1 * np.exp(-(x[..., 2] + prm.H(x)) / (0.1 * prm.alpha)))[fed.dofs.p_b]
    y = B @ b
    scale = 1.0 / fed.mesh.median_edge_length() ** 3
    ws = npg.FgmresWorkspace(ctx, n)
    x = npg.DeviceVector(ctx, n)
    st = ws.solve(A, npg.DeviceVector.from_host(ctx, y), x, P, atol=1e-6, rtol=1e-6, scale=scale)
    assert st["solved"] == 1 and st["niter"] <= 40, st
    xs = x.to_host()
    assert scale * np.linalg.norm(y - As @ xs) <= 1.001 * (1e-6 + 1e-6 * scale * np.linalg.norm(y))
    # the restatement takes the same number of iterations (identical algorithm) ...
    levels = _host_levels(arch, prm, frc, hier)
    xr, sr = mo.fgmres(As, y, lambda v: mo.vcycle(levels, 1, v), scale=scale)
    assert abs(sr["niter"] - st["niter"]) <= 1 and rel(xs, xr) < 1e-3
    # ... and the reference-configured path lands on the same solution, thousands of iterations later
    gm = npg.GmresWorkspace(ctx, n)
    xg = npg.DeviceVector(ctx, n)
    sg = gm.solve(A, npg.DeviceVector.from_host(ctx, y), xg, npg.Diagonal(scalar=scale, n=n), atol=1e-9, rtol=1e-9)
    assert sg["solved"] == 1 and sg["niter"] > 50 * st["niter"]
    xt = npg.DeviceVector(ctx, n)
    ws.solve(A, npg.DeviceVector.from_host(ctx, y), xt, P, atol=1e-10, rtol=1e-10, scale=scale)
    nu = fed.dofs.nu
    assert rel(xt.to_host()[:nu], xg.to_host()[:nu]) < 1e-5
    assert rel(xs[:nu], xg.to_host()[:nu]) < 5e-3            # at the reference's tolerance: the K5 floor class


def test_timestep_loop_with_multigrid_satisfies_the_oracle_equations(arch):
    """4 steps of the example configuration (invert!, then evolve! + invert!, BDF2) on bowl3D h = 0.05 with the
    multigrid-preconditioned inversion.  A sparse LU of this 134 866-unknown saddle point takes ten minutes on the host, so
    instead of comparing with the oracle's direct-solve trajectory (done on h = 0.1 below and, for the Diagonal path, on
    h = 0.08 in test_config2_bowl3D_h008_timestep_loop) every step of the GPU trajectory is substituted into the ORACLE's
    operators: it must satisfy the evolution equation (M + theta K) b+ = y(b, b-, u, u-) and the inversion equation
    A [u; p] = B b + b0 of the recipe (src/model.jl:213-317, incl. the BDF1-LHS first step) to the solver tolerance."""
    ms = workloads.bowl_hierarchy_models("bowl3D_h0.05")
    S = rc.setup("example", model=ms[-1])
    o = S.orc
    m = workloads.example_model(arch, "bowl3D_h0.05", preconditioner="multigrid", atol=1e-10, rtol=1e-10)
    m.evolution.solver.kwargs.update(atol=1e-12, rtol=1e-12)
    npg.set_b(m, lambda x: 0.05 * np.exp(-(x[..., 2] + 0.5 * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)) / 0.05) * (1 + x[..., 0]))
    npg.invert(m)
    hist = [(m.state.b, m.state.u, m.state.p)]
    for _ in range(4):
        npg.run(m, n_steps=1)
        hist.append((m.state.b, m.state.u, m.state.p))
    assert all(st[1]["solved"] == 1 and st[1]["niter"] <= 60 for st in m.stats), [st[1] for st in m.stats]
    th1, th2 = S.theta("BDF1"), S.theta("BDF2")
    K = S.Kh + S.Kv
    scale = 1.0 / o.precond_h()[0] ** 3
    for i in range(1, 5):
        b1, u1, _ = hist[i - 1]
        b2, u2, _ = hist[max(i - 2, 0)]
        bn, un, pn = hist[i]
        y = o.advection_rhs(b1, b2, u1, u2, S.dt, "BDF2") + th2 * S.rhs_diff + S.dt * S.rhs_flux \
            - (S.rhs_M + th2 * (S.rhs_h + S.rhs_v))
        lhs = S.M + (th1 if i == 1 else th2) * K
        assert np.linalg.norm(lhs @ bn - y) < 1e-9 * np.linalg.norm(y), (i, np.linalg.norm(lhs @ bn - y) / np.linalg.norm(y))
        rhs = S.B @ bn + S.b0
        res = scale * np.linalg.norm(S.A @ np.concatenate([un, pn]) - rhs)
        assert res <= 1.01 * (1e-10 + 1e-10 * scale * np.linalg.norm(rhs)) + 1e-12 * scale * np.linalg.norm(rhs), (i, res)
    assert rel(hist[4][0], hist[0][0]) > 1e-3              # the state did move


def test_smoother_only_preconditioner_on_the_reference_mesh_vs_direct(arch):
    """bowl3D h = 0.1 is the coarsest mesh there is, so its 'hierarchy' is the single level: the preconditioner degenerates to
    `coarse_sweeps` Braess-Sarazin steps.  Solution against the fixture-pinned oracle's direct solve."""
    prm, frc = workloads.example_parameters()
    fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
    S = rc.setup("example")
    d = fed.dofs
    inv = npg.InversionToolkit(arch, fed, prm, frc, preconditioner="multigrid", hierarchy=[fed], atol=1e-8, rtol=1e-8,
                               precond_kw=dict(coarse_sweeps=8))
    bfree = S.orc.interpolate_b(lambda x: 0.1 * np.exp(-(x[..., 2] + 0.5 * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)) / 0.05))
    npg.inversion.invert(inv, npg.DeviceVector.from_host(arch.ctx, bfree, d.p_b))
    st = inv.solver.workspace.stats
    x = inv.solver.x.to_host(d.inv_p_inversion)
    xd = spla.splu(sp.csc_matrix(S.A)).solve(S.B @ bfree + S.b0)
    assert st["solved"] == 1 and st["niter"] < 400
    assert rel(x[:d.nu], xd[:d.nu]) < 1e-5 and rel(x[d.nu:], xd[d.nu:]) < 1e-4


def test_block_diagonal_preconditioner_of_the_reference(arch):
    """BlockDiagonalPreconditioner (src/preconditioners.jl:53-93) behind FGMRES on bowl3D h = 0.1: its two blocks act as inner
    Jacobi-CG solves (checked against scipy on the same block matrices), the outer iteration count drops by an order of
    magnitude against Diagonal(1/h^3) at the parameters of the reference's log (alpha = eps = 1/2: 421 against 11 973 outer
    iterations, scratch/inversion_log.md:147-148; at eps = 0.2 it takes MORE outer iterations than Diagonal(1/h^3), which is
    why the reference leaves it switched off, src/inversion.jl:60), and the solution is the direct solve's."""
    prm, frc = workloads.example_parameters()
    prm.eps = 0.5                                # the log's case: alpha = 1/2, eps = 1/2, f = 1 + y/2, h = 0.2 alpha, N = 15 946
    fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
    S = rc.setup("example", eps=0.5)
    d, ctx = fed.dofs, arch.ctx
    P = npg.BlockDiagonalPreconditioner(arch, prm, fed, u_itmax=0, p_itmax=0, atol=1e-12, rtol=1e-10)
    # the blocks: friction-only velocity block and pressure mass / (alpha^2 eps^2), converged inner solves
    a2e2 = prm.alpha ** 2 * prm.eps ** 2
    F = sp.csr_matrix(rc.setup("example", eps=0.5, f=lambda x: 0 * x[..., 0]).A[:d.nu, :d.nu])[d.p_u][:, d.p_u]
    T = mgm.pressure_mass_matrix(fed) / a2e2
    vol = fed.mesh.detJ.sum() / 6
    assert 0.99 * vol < T.sum() * a2e2 < vol                              # the volume minus the pinned vertex's share
    r = np.cos(np.arange(d.nu + d.np) * 0.11)
    z = P.apply(npg.DeviceVector.from_host(ctx, r), npg.DeviceVector(ctx, d.nu + d.np)).to_host()
    assert rel(F @ z[:d.nu], r[:d.nu]) < 1e-8 and rel(T @ z[d.nu:], r[d.nu:]) < 1e-8
    inv = npg.InversionToolkit(arch, fed, prm, frc, preconditioner="block_diagonal", atol=1e-8, rtol=1e-8,
                               precond_kw=dict(u_itmax=100, p_itmax=0))
    bfree = S.orc.interpolate_b(lambda x: 0.1 * np.exp(-(x[..., 2] + 0.5 * (1 - x[..., 0] ** 2 - x[..., 1] ** 2)) / 0.05))
    npg.inversion.invert(inv, npg.DeviceVector.from_host(ctx, bfree, d.p_b))
    st = inv.solver.workspace.stats
    ref = npg.InversionToolkit(arch, fed, prm, frc, atol=1e-8, rtol=1e-8)
    npg.inversion.invert(ref, npg.DeviceVector.from_host(ctx, bfree, d.p_b))
    x = inv.solver.x.to_host(d.inv_p_inversion)
    xd = spla.splu(sp.csc_matrix(S.A)).solve(S.B @ bfree + S.b0)
    assert st["solved"] == 1 and st["niter"] * 5 < ref.solver.workspace.stats["niter"], (st, ref.solver.workspace.stats)
    assert rel(x[:d.nu], xd[:d.nu]) < 1e-5
    apps, inner = inv.solver.P.counters()
    assert apps == st["niter"] and inner > apps


def test_dense_inverse_preconditioner_on_the_reference_meshes(arch):
    """P = A^-1 explicit in HBM (NPG_PC_DENSE) on bowl3D h = 0.1: one application IS the direct solve (checked against the
    fixture-pinned oracle's sparse LU), flexible GMRES needs <= 3 iterations at any tolerance, and the 5-step timestep loop
    lands on the oracle's direct-solve trajectory."""
    prm, frc = workloads.example_parameters()
    fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
    S = rc.setup("example")
    d, ctx = fed.dofs, arch.ctx
    A = npg.build_A_inversion(arch, fed, prm, frc.nu)
    P = npg.DenseInversePreconditioner(arch, A)
    n = A.shape[0]
    r = np.cos(np.arange(n) * 0.23)
    z = P.apply(npg.DeviceVector.from_host(ctx, r), npg.DeviceVector(ctx, n)).to_host()
    zd = spla.splu(sp.csc_matrix(S.A)).solve(r[d.inv_p_inversion])          # oracle: native order
    assert rel(z[d.inv_p_inversion], zd) < 1e-8, rel(z[d.inv_p_inversion], zd)
    m = workloads.example_model(arch, "bowl3D_h0.1", preconditioner="dense_inverse", atol=1e-12, rtol=1e-12)
    m.evolution.solver.kwargs.update(atol=1e-13, rtol=1e-13)
    npg.invert(m)
    npg.run(m, n_steps=5)
    assert all(st[1]["solved"] == 1 and st[1]["niter"] <= 3 for st in m.stats), [st[1] for st in m.stats]
    u, p, b = rc.run(S, 5, solver="direct", invert_first=True)
    assert rel(m.state.b, b) < 1e-9 and rel(m.state.u, u) < 1e-7 and rel(m.state.p, p) < 1e-7, \
        (rel(m.state.b, b), rel(m.state.u, u), rel(m.state.p, p))
    with pytest.raises(npg._lib.DeviceError):                                    # node-blocked matrices are refused
        Ab = npg.build_A_inversion(arch, fed, prm, frc.nu)
        Ab.block_nodes(d.n_full, d.n_surf)
        npg.DenseInversePreconditioner(arch, Ab)


def test_vcycle_on_node_blocked_levels_matches_host_restatement(arch, two_level):
    """the same cycle with the level matrices stored by node blocks - the form production sizes get: the residuals run on the
    record tiles and t = Dinv r_u rides in their node-block epilogue (NbEpi), the velocity update in the (Dinv G) kernel's second
    output - against the host restatement, and against the plain-CSR cycle"""
    prm, frc, hier, A, As, P = two_level
    n = As.shape[0]
    levels = _host_levels(arch, prm, frc, hier)
    Ab = npg.build_A_inversion(arch, hier[-1], prm, frc.nu)
    assert Ab.block_nodes(hier[-1].dofs.n_full, hier[-1].dofs.n_surf)
    Pb = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=Ab, block_nodes=True, coarse_dense=False)
    assert all(a.storage()[0] > 0 for a in Pb.A)                       # every level's matrix is in record form
    r = np.sin(np.arange(n) * 0.37) + 0.1
    rv = npg.DeviceVector.from_host(arch.ctx, r)
    z = Pb.apply(rv, npg.DeviceVector(arch.ctx, n)).to_host()
    zr = mo.vcycle(levels, len(levels) - 1, r, omega=2.5, jw=0.7, sweeps=3, nu1=2, nu2=2, coarse=20)
    zp = P.apply(rv, npg.DeviceVector(arch.ctx, n)).to_host()
    assert rel(z, zr) < 1e-10 and rel(z, zp) < 1e-10, (rel(z, zr), rel(z, zp))


def test_multigrid_with_exact_coarse_solve(arch, two_level):
    """coarse_dense: the coarsest level solved by its dense inverse; the V-cycle then equals the restatement with an exact
    coarse solve, and the iteration count drops to the two-grid optimum"""
    prm, frc, hier, A, As, P = two_level
    n = As.shape[0]
    levels = _host_levels(arch, prm, frc, hier)
    lu0 = spla.splu(sp.csc_matrix(levels[0].A))
    r = np.sin(np.arange(n) * 0.37) + 0.1
    l = levels[1]
    x = mo.smooth(l, np.zeros(n), r, 2, 2.5, 0.7, 3)
    x = x + l.P @ lu0.solve(l.P.T @ (r - l.A @ x))
    zr = mo.smooth(l, x, r, 2, 2.5, 0.7, 3)
    # the inverse stored in full precision / rounded to fp32 / columns scaled and rounded to fp16 (11 bits: what None picks)
    for mode, bar in (("fp64", 1e-8), ("fp32", 1e-5), ("fp16", 2e-2)):
        P2 = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=A, block_nodes=False, coarse_dense=mode)
        z = P2.apply(npg.DeviceVector.from_host(arch.ctx, r), npg.DeviceVector(arch.ctx, n)).to_host()
        assert rel(z, zr) < bar, (mode, rel(z, zr))
        if mode == "fp32":
            assert rel(z, zr) > 1e-10                            # the fp32 copy is really what ran
        if mode == "fp16":
            assert rel(z, zr) > 1e-7 and "fp16" in repr(P2)


def test_multigrid_with_fp32_operator_values(arch, two_level):
    """mixed=True: the cycle's SpMVs read fp32 copies of the level operators (plain CSR and node-block storage); one
    application agrees with the fp64 cycle to fp32 rounding (and differs from it: the copies are what ran), the outer
    flexible GMRES - fp64 throughout - takes the same number of iterations to the same tolerance."""
    prm, frc, hier, A, As, P = two_level
    n = As.shape[0]
    ctx = arch.ctx
    r = np.sin(np.arange(n) * 0.37) + 0.1
    z64 = P.apply(npg.DeviceVector.from_host(ctx, r), npg.DeviceVector(ctx, n)).to_host()
    fed = hier[-1]
    y = npg.build_B_inversion(arch, fed, prm).to_scipy_csr() @ fed.spaces.interpolate_b(
        lambda x: 0.1 * np.exp(-(x[..., 2] + prm.H(x)) / (0.1 * prm.alpha)))[fed.dofs.p_b]
    scale = 1.0 / fed.mesh.median_edge_length() ** 3
    ws = npg.FgmresWorkspace(ctx, n)
    x0 = npg.DeviceVector(ctx, n)
    st0 = ws.solve(A, npg.DeviceVector.from_host(ctx, y), x0, P, atol=1e-6, rtol=1e-6, scale=scale)
    for blocks in (False, True):
        Ab = A
        if blocks:
            Ab = npg.build_A_inversion(arch, fed, prm, frc.nu)
            assert Ab.block_nodes(fed.dofs.n_full, fed.dofs.n_surf)
        Pm = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=Ab, block_nodes=False, coarse_dense=False, mixed=True)
        z32 = Pm.apply(npg.DeviceVector.from_host(ctx, r), npg.DeviceVector(ctx, n)).to_host()
        assert 1e-12 < rel(z32, z64) < 1e-5, rel(z32, z64)
        x1 = npg.DeviceVector(ctx, n)
        st1 = ws.solve(Ab, npg.DeviceVector.from_host(ctx, y), x1, Pm, atol=1e-6, rtol=1e-6, scale=scale)
        assert st1["solved"] == 1 and abs(st1["niter"] - st0["niter"]) <= 1, (st0, st1)
        assert scale * np.linalg.norm(y - As @ x1.to_host()) <= 1.001 * (1e-6 + 1e-6 * scale * np.linalg.norm(y))


def test_preconditioner_abi_argument_errors(arch):
    """the round-2 entry points validate on the host like the rest of the ABI: wrong shapes, wrong order of set-up calls, index
    maps out of range, node-block patterns that do not match - status code + message, never a device fault"""
    import ctypes as C

    from nupgcm_amd import _lib as L
    from nupgcm_amd.architectures import DeviceIndex
    ctx = arch.ctx
    lib = L.lib()
    I = npg.on_architecture(arch, sp.csr_matrix(sp.eye(12)))
    pc = C.c_void_p()
    assert lib.npg_precond_create(ctx.h, 7, 1, C.byref(pc)) != 0 and b"unknown kind" in lib.npg_last_error()
    assert lib.npg_precond_create(ctx.h, L.NPG_PC_MG, 0, C.byref(pc)) != 0
    P = mgm.GeneralPreconditioner(ctx, L.NPG_PC_MG, 2)
    r, z = npg.DeviceVector(ctx, 12), npg.DeviceVector(ctx, 12)
    with pytest.raises(L.DeviceError, match="not all set"):
        P.apply(r, z)
    G = npg.on_architecture(arch, sp.csr_matrix(np.ones((8, 4))))
    D = npg.on_architecture(arch, sp.csr_matrix(np.ones((4, 8))))
    Di = npg.on_architecture(arch, sp.csr_matrix(sp.eye(8)))
    S = npg.on_architecture(arch, sp.csr_matrix(sp.eye(4)))
    # level 1 before level 0; transfer operators at level 0; block shapes that do not add up
    assert lib.npg_precond_mg_set_level(P.h, 1, I.h, 8, G.h, D.h, Di.h, S.h, I.h, I.h) != 0
    assert lib.npg_precond_mg_set_level(P.h, 0, I.h, 8, G.h, D.h, Di.h, S.h, I.h, I.h) != 0
    assert lib.npg_precond_mg_set_level(P.h, 0, I.h, 7, G.h, D.h, Di.h, S.h, None, None) != 0
    assert lib.npg_precond_mg_set_level(P.h, 0, I.h, 8, G.h, D.h, Di.h, S.h, None, None) == 0
    assert lib.npg_precond_mg_set_level(P.h, 0, I.h, 8, G.h, D.h, Di.h, S.h, None, None) != 0      # already set
    assert lib.npg_precond_mg_set_params(P.h, -1.0, 0.7, 3, 2, 2, 20) != 0
    assert lib.npg_precond_mg_set_cycle(P.h, 3) != 0
    ws = npg.FgmresWorkspace(ctx, 12)
    with pytest.raises(L.DeviceError):
        ws.solve(I, npg.DeviceVector(ctx, 11), ws.x, None)                                        # short right-hand side
    with pytest.raises(L.DeviceError):
        npg.FgmresWorkspace(ctx, 12, memory=40)
    Pd = mgm.GeneralPreconditioner(ctx, L.NPG_PC_DENSE, 1)
    with pytest.raises(L.DeviceError, match="has not been set"):
        Pd.apply(r, z)
    assert lib.npg_precond_dense_set(Pd.h, G.h, 0) != 0                                           # not square
    sing = npg.on_architecture(arch, sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 4.0]])))
    assert lib.npg_precond_dense_set(Pd.h, sing.h, 0) != 0 and b"singular" in lib.npg_last_error()
    big = npg.on_architecture(arch, sp.identity(50000, format="csr"))          # n^2 > 2^31: refused before anything is allocated
    assert lib.npg_precond_dense_set(Pd.h, big.h, 0) != 0 and b"46 340" in lib.npg_last_error()
    # fixed-pattern helpers
    with pytest.raises(L.DeviceError, match="outside"):
        DeviceIndex(ctx, np.array([0, 5, 99]), 12)
    with pytest.raises(L.DeviceError):
        I.gather_values(G, DeviceIndex(ctx, np.zeros(12, dtype=np.int64), 32).__class__(ctx, np.zeros(3, dtype=np.int64), 32))
    with pytest.raises(L.DeviceError, match="node blocks"):
        L.check(lib.npg_csr_node_block_inverse(I.h, I.h, 2, 1))            # I holds 12 entries, the blocks would need 28
    with pytest.raises(L.DeviceError, match="outside S"):
        L.check(lib.npg_csr_triple_product(S.h, D.h, Di.h, G.h))           # D Dinv G is full, S's pattern is diagonal
    # a block solve that breaks down (NaN in the block: CG status 3) is an error of the application, not a NaN in z
    bad = sp.csr_matrix(sp.eye(12) * 2.0).tolil()
    bad[3, 3] = np.nan
    Ab = npg.on_architecture(arch, sp.csr_matrix(bad))
    Pb = mgm.GeneralPreconditioner(ctx, L.NPG_PC_BLOCKDIAG, 1)
    jac = npg.DeviceVector.from_host(ctx, np.full(12, 0.5))
    L.check(lib.npg_precond_blockdiag_set(Pb.h, 0, 0, Ab.h, jac.h, 12, 1e-10, 1e-10))
    r.fill(1.0)
    with pytest.raises(L.DeviceError, match="broke down"):
        Pb.apply(r, z)
    lib.npg_precond_destroy(pc) if pc else None


def test_zline_smoother_on_the_channel_basin(arch):
    """smoother="zline" (Braess-Sarazin whose velocity blocks are the unknowns of the nodes above one another) on the channel
    basin's three-level hierarchy (2 431 / 20 807 / 172 591 unknowns; alpha = 1/8: the anisotropic case): the device's block
    inverse equals the host's, one V-cycle equals the host restatement built from the same operators, and flexible GMRES takes
    less than half the node-block smoother's iterations to the same solution."""
    ctx = arch.ctx
    models = workloads.channel_basin_hierarchy_models(0.03125, 2)
    hier = [workloads.channel_basin_fe_data(m) for m in models]
    prm, frc, _, _, dt, b0 = workloads.channel_basin_parameters("flux")
    fed = hier[-1]
    d = fed.dofs
    n = d.nu + d.np
    A = npg.build_A_inversion(arch, fed, prm, frc.nu, structural=True)
    As = A.to_scipy_csr()
    Pz = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=A, block_nodes=False, coarse_dense=False, smoother="zline", omega=1.5)
    Pn = mgm.MultigridPreconditioner(arch, prm, frc, hier, A_fine=A, block_nodes=False, coarse_dense=False, omega=2.0)
    assert "z-line" in repr(Pz) and Pz.ops[-1].block_sizes.max() > 20 and Pz.ops[-1].Gh is None
    # operators: the device's line-block inverse against the host's, level by level; S on its line-built pattern
    levels = []
    for k, f in enumerate(hier):
        Ak = npg.build_A_inversion(arch, f, prm, frc.nu, structural=True).to_scipy_csr()
        bp, bd, _ = mgm.line_blocks(f)
        Dh = mgm.line_block_inverse(Ak[:f.dofs.nu, :f.dofs.nu], bp, bd)
        Dd = Pz.ops[k].Dinv.to_scipy_csr()
        assert abs(Dd - Dh).max() < 1e-10 * abs(Dh).max()
        lv = mo.Level(Ak, f.dofs.nu, Dh, P=None if k == 0 else mgm.prolongation(hier[k - 1], f))
        Sd = Pz.ops[k].S.to_scipy_csr()
        assert abs(Sd - lv.S).max() < 1e-10 * abs(lv.S).max()
        levels.append(lv)
    r = np.sin(np.arange(n) * 0.37) + 0.1
    z = Pz.apply(npg.DeviceVector.from_host(ctx, r), npg.DeviceVector(ctx, n)).to_host()
    zr = mo.vcycle(levels, len(levels) - 1, r, omega=1.5, jw=0.7, sweeps=3, nu1=2, nu2=2, coarse=20)
    assert rel(z, zr) < 1e-9, rel(z, zr)
    # the preconditioned solve
    y = npg.build_B_inversion(arch, fed, prm).to_scipy_csr() @ fed.spaces.interpolate_b(b0)[d.p_b]
    scale = 1.0 / fed.mesh.median_edge_length() ** 3
    ws = npg.FgmresWorkspace(ctx, n)
    out = {}
    for name, P in (("zline", Pz), ("node", Pn)):
        x = npg.DeviceVector(ctx, n)
        st = ws.solve(A, npg.DeviceVector.from_host(ctx, y), x, P, atol=1e-6, rtol=1e-6, scale=scale, itmax=400)
        assert st["solved"] == 1, (name, st)
        out[name] = (st["niter"], x.to_host())
    assert 2 * out["zline"][0] < out["node"][0], (out["zline"][0], out["node"][0])
    assert rel(out["zline"][1][:d.nu], out["node"][1][:d.nu]) < 1e-4
    with pytest.raises(ValueError):
        mgm.MultigridPreconditioner(arch, prm, frc, hier[-2:], smoother="plane")


def test_line_block_entry_points_on_a_synthetic_matrix(arch):
    """npg_csr_line_block_inverse / npg_csr_line_schur / npg_csr_product / npg_csr_set_lanes through the C ABI on a matrix that is
    no finite-element matrix at all: ragged blocks (1 .. 70 unknowns, interleaved index sets), results against numpy; products
    with the block inverse (dense pack, fp64 and fp32) against the CSR form; and the argument checks of each entry point."""
    from nupgcm_amd.architectures import DeviceIndex
    ctx = arch.ctx
    rng = np.random.default_rng(5)
    sizes = np.array([1, 2, 70, 3, 65, 17, 64, 5, 33, 128])
    nu, npp = int(sizes.sum()), 40
    perm = rng.permutation(nu)
    bp = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    bd = np.concatenate([np.sort(perm[bp[b]:bp[b + 1]]) for b in range(len(sizes))]).astype(np.int64)
    line_of = np.empty(nu, dtype=np.int64)
    for b in range(len(sizes)):
        line_of[bd[bp[b]:bp[b + 1]]] = b
    F = sp.random(nu, nu, 0.05, random_state=3, format="csr")
    F = sp.csr_matrix(F + F.T + sp.diags(np.asarray(abs(F + F.T).sum(axis=1)).ravel() + 1.0))
    G = sp.csr_matrix(sp.random(nu, npp, 0.08, random_state=4, format="csr"))
    D = sp.csr_matrix(sp.random(npp, nu, 0.08, random_state=6, format="csr"))
    A = sp.csr_matrix(sp.bmat([[F, G], [D, None]]))
    Ad = npg.DeviceCSR.from_scipy(ctx, A)
    irp, icol = mgm._line_block_pattern(nu, bp, bd, line_of)
    Dinv = npg.DeviceCSR.from_pattern(ctx, nu, nu, irp, icol)
    ip, idf = DeviceIndex(ctx, bp, nu + 1), DeviceIndex(ctx, bd, nu)
    L = npg._lib
    L.check(L.lib().npg_csr_line_block_inverse(Dinv.h, Ad.h, ip.h, idf.h))
    Dh = mgm.line_block_inverse(F, bp, bd)
    assert abs(Dinv.to_scipy_csr() - Dh).max() < 1e-11 * abs(Dh).max()
    # products with it: dense pack (default) in fp64, against scipy
    x = rng.standard_normal(nu)
    y = Dinv.mul(npg.DeviceVector.from_host(ctx, x)).to_host()
    assert rel(y, Dh @ x) < 1e-12
    # S = D Dinv G line by line against the generic triple product and numpy
    Gd, Dd = npg.DeviceCSR.from_scipy(ctx, G), npg.DeviceCSR.from_scipy(ctx, D)
    Sref = sp.csr_matrix(D @ Dh @ G)
    one = lambda M: sp.csr_matrix((np.ones(M.nnz), M.indices, M.indptr), shape=M.shape)
    Gs, Ds = sp.csr_matrix(G), sp.csr_matrix(D)
    Gs.sort_indices(); Ds.sort_indices()
    nl = len(sizes)
    Dl = sp.csr_matrix((np.ones(Ds.nnz), line_of[Ds.indices], Ds.indptr), shape=(npp, nl))
    Gc = sp.coo_matrix(Gs)
    Gl = sp.csr_matrix((np.ones(Gc.nnz), (line_of[Gc.row], Gc.col)), shape=(nl, npp))
    Gl.sort_indices()
    Sp = sp.csr_matrix(Dl @ Gl)
    Sp.sort_indices()
    S1 = npg.DeviceCSR.from_pattern(ctx, npp, npp, Sp.indptr, Sp.indices)
    S2 = npg.DeviceCSR.from_pattern(ctx, npp, npp, Sp.indptr, Sp.indices)
    woff = np.concatenate([[0], np.cumsum(sizes * np.diff(Gl.indptr))]).astype(np.int64)
    pos = np.empty(nu, dtype=np.int64)
    pos[bd] = np.arange(nu) - bp[line_of[bd]]
    rows = np.repeat(np.arange(npp, dtype=np.int64), np.diff(Ds.indptr))
    lines = line_of[Ds.indices]
    order = np.lexsort((Ds.indices, lines, rows)).astype(np.int64)
    key = rows[order] * nl + lines[order]
    cut = np.concatenate([[0], np.flatnonzero(np.diff(key)) + 1, [len(order)]]).astype(np.int64)
    seg_line = lines[order][cut[:-1]]
    seg_ptr = np.searchsorted(rows[order][cut[:-1]], np.arange(npp + 1)).astype(np.int64)
    idx = [DeviceIndex(ctx, Gl.indptr.astype(np.int64), Gl.nnz + 1), DeviceIndex(ctx, Gl.indices.astype(np.int64), npp),
           DeviceIndex(ctx, woff, int(woff[-1]) + 1), DeviceIndex(ctx, order, Ds.nnz), DeviceIndex(ctx, pos[Ds.indices[order]], int(sizes.max())),
           DeviceIndex(ctx, seg_ptr, len(seg_line) + 1), DeviceIndex(ctx, seg_line, nl), DeviceIndex(ctx, cut, Ds.nnz + 1)]
    Dd_sorted, Gd_sorted = npg.DeviceCSR.from_scipy(ctx, Ds), npg.DeviceCSR.from_scipy(ctx, Gs)      # (kept: the calls borrow them)
    L.check(L.lib().npg_csr_line_schur(S1.h, Dd_sorted.h, Dinv.h, Gd_sorted.h, *[i.h for i in idx]))
    L.check(L.lib().npg_csr_triple_product(S2.h, Dd.h, Dinv.h, Gd.h))
    assert abs(S1.to_scipy_csr() - Sref).max() < 1e-11 * abs(Sref).max() and abs(S2.to_scipy_csr() - Sref).max() < 1e-11 * abs(Sref).max()
    # C = A B on a fixed pattern; lanes per row
    Tp = sp.csr_matrix(one(Dh) @ one(Gs))
    Tp.sort_indices()
    T = npg.DeviceCSR.from_pattern(ctx, nu, npp, Tp.indptr, Tp.indices)
    L.check(L.lib().npg_csr_product(T.h, Dinv.h, Gd_sorted.h))
    assert abs(T.to_scipy_csr() - Dh @ G).max() < 1e-11 * abs(Dh @ G).max()
    xs = rng.standard_normal(npp)
    for lanes in (4, 8, 16, 32, 0):
        assert rel(T.set_lanes(lanes).mul(npg.DeviceVector.from_host(ctx, xs)).to_host(), (Dh @ G) @ xs) < 1e-12
    # argument checks
    with pytest.raises(L.DeviceError, match="lanes"):
        T.set_lanes(5)
    with pytest.raises(L.DeviceError, match="shapes"):
        L.check(L.lib().npg_csr_product(T.h, Dinv.h, Dd.h))
    with pytest.raises(L.DeviceError, match="outside"):
        Tsmall = npg.DeviceCSR.from_pattern(ctx, nu, npp, np.arange(nu + 1, dtype=np.int64), np.zeros(nu, dtype=np.int32))
        L.check(L.lib().npg_csr_product(Tsmall.h, Dinv.h, Gd_sorted.h))
    with pytest.raises(L.DeviceError, match="ascending"):
        bad = bd.copy()
        bad[bp[2]], bad[bp[2] + 1] = bad[bp[2] + 1], bad[bp[2]]
        ibad = DeviceIndex(ctx, bad, nu)
        L.check(L.lib().npg_csr_line_block_inverse(Dinv.h, Ad.h, ip.h, ibad.h))
    with pytest.raises(L.DeviceError, match="pattern|hold"):
        Dn = npg.DeviceCSR.from_pattern(ctx, nu, nu, *mgm._node_block_pattern(nu, 0, 0))
        L.check(L.lib().npg_csr_line_block_inverse(Dn.h, Ad.h, ip.h, idf.h))
    with pytest.raises(L.DeviceError, match="unknowns"):
        big = np.array([0, nu], dtype=np.int64)                 # one block of 388 unknowns: beyond the LDS budget
        ibig, iall = DeviceIndex(ctx, big, nu + 1), DeviceIndex(ctx, np.arange(nu), nu)
        L.check(L.lib().npg_csr_line_block_inverse(Dinv.h, Ad.h, ibig.h, iall.h))
    with pytest.raises(L.DeviceError, match="line-block"):
        L.check(L.lib().npg_csr_line_schur(S1.h, Dd_sorted.h, T.h, Gd.h, *[i.h for i in idx]))
    with pytest.raises(L.DeviceError, match="index arrays"):
        wrong = list(idx)
        wrong[5] = DeviceIndex(ctx, seg_ptr[:-1], len(seg_line) + 1)
        L.check(L.lib().npg_csr_line_schur(S1.h, Dd_sorted.h, Dinv.h, Gd.h, *[i.h for i in wrong]))
    # the handle now carries dense packs of its blocks, which products read instead of the CSR values: any OTHER writer of those
    # values would leave the packs stale and is refused
    for call in (lambda: L.check(L.lib().npg_csr_zero_values(Dinv.h)), lambda: Dinv.combine(1.0, Dinv, 0.5, Dinv, Dinv),
                 lambda: L.check(L.lib().npg_csr_node_block_inverse(Dinv.h, Ad.h, 0, 0))):
        with pytest.raises(L.DeviceError, match="line-block packs"):
            call()
    Fs = sp.csr_matrix(F.copy())
    Fs.data[Fs.indptr[bd[bp[3]]]:Fs.indptr[bd[bp[3]] + 1]] = 0.0                      # a zero row inside block 3: singular
    with pytest.raises(L.DeviceError, match="singular"):
        As = npg.DeviceCSR.from_scipy(ctx, sp.csr_matrix(sp.bmat([[Fs, G], [D, None]])))
        L.check(L.lib().npg_csr_line_block_inverse(Dinv.h, As.h, ip.h, idf.h))


@pytest.mark.gpu
def test_coarse_viscosity_is_the_child_volume_average_of_the_fine_one(arch):
    """npg_fe_restrict_coeff (the multigrid refresh with the eddy closure, multigrid.MultigridPreconditioner.refresh): the coarse
    engine's viscosity table = the volume-weighted average over a cell's eight children of the quadrature mean of the fine table -
    checked through what it is for: the full-stress coarse matrix assembled from the restricted table equals the one assembled
    from that average computed on the host, and differs from the one of the pointwise viscosity."""
    from nupgcm_amd.assembly import eval_at_quad_points
    from nupgcm_amd.inversion import device_fe
    prm, frc = workloads.example_parameters()
    hier = [workloads.example_fe_data(m) for m in workloads.bowl_hierarchy_models("bowl3D_h0.05")]
    coarse, fine = hier

    def nu(x):      # a strongly varying positive field
        return 1.0 + 0.9 * np.sin(7.0 * x[..., 0]) * np.cos(5.0 * x[..., 1]) + 4.0 * x[..., 2] ** 2

    fe_f, fe_c = device_fe(arch, fine), device_fe(arch, coarse)
    fe_f.set_coeff("nu", nu)
    fe_c.set_coeff("f", prm.f)
    fe_c.restrict_coeff(fe_f, "nu")
    A_dev = npg.build_A_inversion(arch, coarse, prm, None).to_scipy_csr()
    # the same average on the host
    m = fine.mesh
    tab = eval_at_quad_points(m, nu)
    wq = m.q_w / m.q_w.sum()
    mean = tab @ wq
    avg = (mean * m.detJ).reshape(-1, 8).sum(axis=1) / m.detJ.reshape(-1, 8).sum(axis=1)
    table = np.repeat(avg[:, None], len(wq), axis=1)
    A_host = npg.build_A_inversion(arch, coarse, prm, lambda x: table).to_scipy_csr()
    scale = abs(A_host).max()
    assert abs(A_dev - A_host).max() <= 1e-13 * scale
    A_point = npg.build_A_inversion(arch, coarse, prm, nu).to_scipy_csr()
    assert abs(A_dev - A_point).max() > 1e-3 * scale
    # argument errors
    L = npg._lib
    with pytest.raises(L.DeviceError, match="uniform refinement"):
        fe_f.restrict_coeff(fe_c, "nu")
    with pytest.raises(L.DeviceError, match="unknown coefficient"):
        fe_c.restrict_coeff(fe_f, "viscosity")
