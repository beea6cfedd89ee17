"""EvolutionToolkit - mirrors /root/reference/src/evolution.jl:1-296 for the GPU() architecture.

    mu_rho (d_t b + u . grad b) = alpha^2 eps^2 [ div_h(kappa_h grad_h b) + d_z(kappa_v d_z b) ]
    (M + theta (Kh + Kv)) b^{n+1} = rhs,   Jacobi-preconditioned CG

M, Kh, Kv and their Dirichlet lift vectors, rhs_diff, and the combination A = M + theta (Kh + Kv) with its Jacobi
diagonal are all produced on the device in p_b order; rhs_flux (a surface integral, set-up constant) comes from the host."""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .architectures import CPU, GPU, DeviceVector
from .inputs import SurfaceFluxBC
from .inversion import device_fe
from .iterative_solvers import LU, CgWorkspace, Diagonal, IterativeSolverToolkit
from .timesteppers import BDF1, BDF2


def evolution_parameter(params, ts):
    """theta in A = M + theta (Kh + Kv) - src/evolution.jl:187-193"""
    c = params.alpha ** 2 * params.eps ** 2 / params.mu_rho
    return ts.dt * c if isinstance(ts, BDF1) else 2.0 / 3.0 * ts.dt * c


def build_rhs_flux(arch, params, forcings, fe_data):
    """build_rhs_flux - src/evolution.jl:280-296: alpha int_Gamma F d (zeros for a Dirichlet surface condition)"""
    nb = fe_data.dofs.nb
    host = np.zeros(nb)
    bc = forcings.b_surface_bc
    if isinstance(bc, SurfaceFluxBC):
        fn = bc.flux if callable(bc.flux) else (lambda x, c=float(bc.flux): np.full(x.shape[:-1], c))
        load = fe_data.mesh.surface_load(lambda x: params.alpha * fn(x))[:fe_data.spaces.nb_nodes]
        pos = fe_data.tables.b_pos
        host[pos[pos >= 0]] = load[pos >= 0]
    return DeviceVector.from_host(arch.ctx, host)


class EvolutionToolkit:
    """src/evolution.jl:1-17: {arch, M, Kh, Kv, rhs_diff, rhs_flux, rhs_M, rhs_h, rhs_v, solver}"""

    def __init__(self, arch, fe_data, params, forcings, ts, atol=1e-6, rtol=1e-6, itmax=0, history=True, verbose=False,
                 first_step_lhs="bdf1"):
        """first_step_lhs: "bdf1" = the current source (the first step of any run uses a BDF1 left-hand side,
        src/evolution.jl:110-111); "bdf2" = the timestepper's own LHS from the start - what the reference's exact state
        fixture test/data/bowl_surface_flux.jld2 (written by an older revision) encodes."""
        if not isinstance(arch, (GPU, CPU)):
            raise TypeError("EvolutionToolkit: arch must be GPU() or CPU()")
        self.arch, self.fe_data, self.params, self.forcings = arch, fe_data, params, forcings
        fe = self.fe = device_fe(arch, fe_data)
        ctx, nb = arch.ctx, fe_data.dofs.nb
        fe.set_coeff("kappa_h", forcings.kappa_h)
        fe.set_coeff("kappa_v", forcings.kappa_v)
        # build components (src/evolution.jl:80-84), already in p_b order (src/evolution.jl:91-99)
        self.rhs_M, self.rhs_h, self.rhs_v = (DeviceVector(ctx, nb) for _ in range(3))
        self.M = fe.assemble(L.NPG_MAT_M, fe.new_matrix("b"), lift=self.rhs_M)
        self.Kh = fe.assemble(L.NPG_MAT_KH, fe.new_matrix("b"), lift=self.rhs_h)
        self.Kv = fe.assemble(L.NPG_MAT_KV, fe.new_matrix("b"), lift=self.rhs_v)
        self.rhs_diff = fe.rhs_diff(params.N2, DeviceVector(ctx, nb))
        self.rhs_flux = build_rhs_flux(arch, params, forcings, fe_data)
        # LHS for the first step: always a BDF1 matrix (src/evolution.jl:110-111)
        ts1 = BDF1(t_start=ts.t_start, t_stop=ts.t_stop, dt=ts.dt) if first_step_lhs == "bdf1" else ts
        A = fe.new_matrix("b")
        P = Diagonal(DeviceVector(ctx, nb))
        collect_evolution_LHS_into(A, P, params, ts1, self.M, self.Kh, self.Kv)
        P = _cpu_factorisation(arch, forcings, ts1, A, P)
        y = DeviceVector(ctx, nb)
        ws = CgWorkspace(ctx, nb)
        kwargs = dict(atol=atol, rtol=rtol, itmax=itmax, history=history, verbose=int(verbose))
        self.solver = IterativeSolverToolkit(A, P, y, ws, kwargs, "Evolution")

    def __repr__(self):
        return (f"EvolutionToolkit:\n├── arch: {self.arch}\n├── M: {self.M!r}\n├── Kₕ: {self.Kh!r}\n├── Kᵥ: {self.Kv!r}\n"
                f"└── solver: IterativeSolverToolkit")


def _cpu_factorisation(arch, forcings, ts, A, P):
    """src/evolution.jl:148-153,166-171: on CPU() with fixed coefficients (no convection closure, no adaptive step) the
    preconditioner is lu(A) - iterative_solve! then back-substitutes; in every other case the Jacobi diagonal stays"""
    if isinstance(arch, CPU) and not forcings.conv_param.is_on and not getattr(ts, "adaptive", False):
        return LU(A)
    return P


def collect_evolution_LHS_into(A, P, params, ts, M, Kh, Kv):
    """A = M + theta (Kh + Kv); P = Diagonal(1 ./ diag(A)) - src/evolution.jl:143-177, as two device kernels on the shared
    pattern instead of a host sparse add + upload."""
    theta = evolution_parameter(params, ts)
    A.combine(1.0, M, theta, Kh, Kv)
    A.inv_diag(P.diag)
    return A, P


def collect_evolution_LHS(evolution: EvolutionToolkit, params, forcings, ts):
    """collect_evolution_LHS! - src/evolution.jl:133-142"""
    s = evolution.solver
    if isinstance(s.P, LU):             # CPU(): the Jacobi vector was replaced by the factorisation - combine, then factorise anew
        theta = evolution_parameter(params, ts)
        s.A.combine(1.0, evolution.M, theta, evolution.Kh, evolution.Kv)
        s.P = LU(s.A)
        return evolution
    collect_evolution_LHS_into(s.A, s.P, params, ts, evolution.M, evolution.Kh, evolution.Kv)
    return evolution
