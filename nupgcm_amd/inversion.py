"""InversionToolkit / invert! - mirrors /root/reference/src/inversion.jl:1-249 for the GPU() architecture.

    A [u; p] = B b + b0,   solved with left-preconditioned restarted GMRES(20) and P = Diagonal(1/h^dim)

Everything the reference assembles with Gridap on the host (A, B, the Dirichlet lift in b0) is assembled by HIP kernels
directly in the RCM-permuted numbering the solver uses; only the wind-stress surface integral (a set-up constant over
boundary triangles) is evaluated on the host."""
from __future__ import annotations

import os

import numpy as np

from . import _lib as L
from .architectures import CPU, GPU, DeviceVector, print_memory_status
from .assembly import DeviceFE
from .iterative_solvers import LU, Diagonal, GmresWorkspace, IterativeSolverToolkit, iterative_solve


def device_fe(arch, fe_data) -> DeviceFE:
    """one assembly engine per (FEData, device)"""
    cache = fe_data.__dict__.setdefault("_device_fe", {})
    if arch.device not in cache:
        cache[arch.device] = DeviceFE(arch.ctx, fe_data)
    return cache[arch.device]


def _is_function(v):
    return callable(v)


def build_A_inversion(arch, fe_data, params, nu, A=None, structural=False):
    """build_A_inversion(!) - src/inversion.jl:133-170.  Real nu: Laplacian form; function nu: full-stress form
    (src/inversion.jl:172-192).  Returns a DeviceCSR in p_inversion order (the `A[perm, perm]` of src/inversion.jl:37)."""
    fe = device_fe(arch, fe_data)
    full = _is_function(nu) or nu is None        # nu None: keep the device table (eddy refresh path)
    if nu is not None:
        fe.set_coeff("nu", nu)
    fe.set_coeff("f", params.f)
    if A is None:
        A = fe.new_matrix("A", structural=structural or full)
    return fe.assemble(L.NPG_MAT_A, A, scale=params.alpha ** 2 * params.eps ** 2, full_stress=full)


def build_B_inversion(arch, fe_data, params, lift=None):
    """build_B_inversion - src/inversion.jl:199-219: N x nb, rows in p_inversion order.  Unlike the reference (whose B
    consumes b in native order, src/inversion.jl:38), columns are in p_b order because the buoyancy vector stays on the
    device in the evolution solver's ordering."""
    fe = device_fe(arch, fe_data)
    B = fe.new_matrix("B")
    return fe.assemble(L.NPG_MAT_B, B, scale=1.0 / params.alpha, lift=lift)


def build_b_inversion(arch, fe_data, params, forcings, lift: DeviceVector):
    """build_b_inversion - src/inversion.jl:226-249: wind stress over the surface + the Dirichlet-b lift (already in
    `lift`, produced by the B assembly kernel)."""
    m, t = fe_data.mesh, fe_data.tables
    host = np.zeros(fe_data.dofs.nu + fe_data.dofs.np)
    for comp, tau in ((0, forcings.tau_x), (1, forcings.tau_y)):
        if callable(tau) or float(tau) != 0.0:
            fn = tau if callable(tau) else (lambda x, c=float(tau): np.full(x.shape[:-1], c))
            load = m.surface_load(lambda x: params.alpha * fn(x))
            pos = t.u_pos[:, comp]
            host[pos[pos >= 0]] += load[pos >= 0]
    if np.any(host != 0.0):
        wind = DeviceVector.from_host(arch.ctx, host)
        lift.axpby(1.0, wind, 1.0)
    return lift


class InversionToolkit:
    """src/inversion.jl:1-5: {B, b, solver}"""

    def __init__(self, arch, *args, atol=1e-6, rtol=1e-6, itmax=0, memory=20, history=True, verbose=False, restart=True,
                 reorth_eta=0.1, block_nodes=None, preconditioner="diagonal", hierarchy=None, precond_kw=None):
        """preconditioner: "diagonal" = the reference's GPU choice Diagonal(1/h^dim) (src/inversion.jl:42-54, default);
        "block_diagonal" = its experimental BlockDiagonalPreconditioner (src/inversion.jl:60, src/preconditioners.jl:53-93);
        "multigrid" = MultigridPreconditioner over `hierarchy` (FEData coarse ... fine, the last one being fe_data);
        "dense_inverse" = the explicit inverse in HBM (small meshes).  These
        are general operators: the workspace is then a flexible GMRES(memory) with the same stopping rule."""
        if not isinstance(arch, (GPU, CPU)):
            raise TypeError("InversionToolkit: arch must be GPU() or CPU()")
        on_gpu = isinstance(arch, GPU)
        if not on_gpu and preconditioner != "diagonal":
            raise NotImplementedError("CPU(): the general preconditioners (multigrid, dense inverse, block diagonal) are device work")
        if not restart:
            raise NotImplementedError("restart=false (growing Krylov basis) is not supported; the reference uses restart=true")
        if len(args) == 3:                          # InversionToolkit(arch, fe_data, params, forcings; kwargs...)
            fe_data, params, forcings = args
            b0 = DeviceVector(arch.ctx, fe_data.dofs.nu + fe_data.dofs.np)
            # the eddy closure re-assembles A in the full-stress form later: keep all nine component pairs in the pattern
            # then (Gridap's structural pattern, src/model.jl:160-170 assembles into it in place)
            A = build_A_inversion(arch, fe_data, params, forcings.nu, structural=forcings.eddy_param.is_on)
            if block_nodes is None and os.environ.get("NPG_BLOCK_NODES"):
                block_nodes = os.environ["NPG_BLOCK_NODES"] != "0"          # tuning override
            if not on_gpu:
                block_nodes = False                     # the record layouts are HBM layouts
            if block_nodes is None:
                # bandwidth-bound sizes only: below ~1e5 rows the solve is latency-bound and the extra stream costs time
                block_nodes = A.shape[0] >= 100000
            if block_nodes and not callable(forcings.nu) and not forcings.eddy_param.is_on:
                # constant nu: K_xx = K_yy = K_zz and C_xy = -C_yx per node pair, stored once (no-op if the structure
                # does not hold)
                A.block_nodes(fe_data.dofs.n_full, fe_data.dofs.n_surf)
            elif block_nodes and fe_data.dofs.n_full + fe_data.dofs.n_surf > 0 and os.environ.get("NPG_PACK_NODES", "1") != "0":
                # function-valued nu / eddy closure (full-stress form): a record-form companion with FULL node records that
                # follows every re-assembly; A itself stays plain (assembly target)
                A.pack_nodes(fe_data.dofs.n_full, fe_data.dofs.n_surf)
            B = build_B_inversion(arch, fe_data, params, lift=b0)
            b0 = build_b_inversion(arch, fe_data, params, forcings, b0)
            # GPU preconditioner: Diagonal(1/h^dim) with the median edge length (src/inversion.jl:42-54)
            h = fe_data.mesh.median_edge_length()
            P = Diagonal(scalar=1.0 / h ** getattr(fe_data.mesh, "dim", 3), n=A.shape[0])      # 1/h^dim: dim = 2 on the embedded 2-D meshes
            if not on_gpu and not forcings.eddy_param.is_on:
                P = LU(A)                               # "on CPU and fixed nu -> can just LU factor" (src/inversion.jl:55-58)
            if preconditioner != "diagonal":
                from . import multigrid as mgm
                self.scale = P.scalar                   # the residual keeps the reference's 1/h^dim scaling
                if preconditioner == "multigrid":
                    if not hierarchy or hierarchy[-1] is not fe_data:
                        raise ValueError("preconditioner='multigrid' needs hierarchy=[FEData coarse, ..., fe_data]")
                    P = mgm.MultigridPreconditioner(arch, params, forcings, hierarchy, A_fine=A, **(precond_kw or {}))
                elif preconditioner == "dense_inverse":
                    P = mgm.DenseInversePreconditioner(arch, A, **(precond_kw or {}))
                elif preconditioner == "block_diagonal":
                    P = mgm.BlockDiagonalPreconditioner(arch, params, fe_data, A, **(precond_kw or {}))
                else:
                    raise ValueError(f"unknown preconditioner {preconditioner!r}")
            print_memory_status(arch) if verbose else None
        elif len(args) == 4:                        # InversionToolkit(arch, A, P, B, b; kwargs...) src/inversion.jl:74-94
            A, P, B, b0 = args
        else:
            raise TypeError("InversionToolkit(arch, fe_data, params, forcings) or InversionToolkit(arch, A, P, B, b)")
        self.arch, self.B, self.b = arch, B, b0
        N = A.shape[0]
        y = DeviceVector(arch.ctx, N)
        kwargs = dict(atol=atol, rtol=rtol, itmax=itmax, history=history, verbose=int(verbose), restart=restart,
                      reorth_eta=reorth_eta)
        if isinstance(P, (Diagonal, LU)) or P is None:
            ws = GmresWorkspace(arch.ctx, N, memory=memory)
        else:
            from .multigrid import FgmresWorkspace
            ws = FgmresWorkspace(arch.ctx, N, memory=memory)
            kwargs["scale"] = getattr(self, "scale", 1.0)
        self.solver = IterativeSolverToolkit(A, P, y, ws, kwargs, "Inversion")

    def __repr__(self):
        return f"InversionToolkit:\n├── B: {self.B!r}\n├── b: {self.b!r}\n└── solver: IterativeSolverToolkit"


def invert(inversion: InversionToolkit, b: DeviceVector):
    """invert!(inversion, b) - src/inversion.jl:101-110: y = B b + b0 (one SpMV with the add folded in), then GMRES."""
    s = inversion.solver
    s.y.copy_from(inversion.b)
    inversion.B.mul(b, s.y, alpha=1.0, beta=1.0)
    iterative_solve(s)
    return inversion
