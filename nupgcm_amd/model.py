"""State, Model, run!, evolve!, invert!, sync_flow!, set_b! - mirrors /root/reference/src/model.jl:1-317 for GPU().

Residency: the reference keeps the state in host FEFunctions and crosses the device boundary 4-6 times per step
(src/model.jl:243-244,275,282,312).  Here the state IS the two solver vectors in HBM (x_inv = [u; p] in p_inversion
order, x_b in p_b order) plus their previous-step copies; a timestep is device-only.  `state.u / p / b` download and
un-permute on access (what sync_flow! / src/model.jl:282 do every step in the reference)."""
from __future__ import annotations

import time

import numpy as np

from . import _lib as L
from .architectures import CPU, GPU, DeviceVector
from .evolution import EvolutionToolkit, collect_evolution_LHS, collect_evolution_LHS_into, evolution_parameter
from .inversion import InversionToolkit, build_A_inversion, invert as invert_toolkit
from .iterative_solvers import iterative_solve
from .timesteppers import BDF1, BDF2, update_dt, update_t


class BlowUp(RuntimeError):
    pass


class State:
    """src/model.jl:1-5.  u, p, b are host copies of the free values in the native (Gridap) DoF order."""

    def __init__(self, model):
        self._m = model

    @property
    def u(self):
        d = self._m.fe_data.dofs
        return self._m.inversion.solver.x.to_host(d.inv_p_inversion)[:d.nu]

    @property
    def p(self):
        d = self._m.fe_data.dofs
        return self._m.inversion.solver.x.to_host(d.inv_p_inversion)[d.nu:]

    @property
    def b(self):
        return self._m.b_vec.to_host(self._m.fe_data.dofs.inv_p_b)

    def __repr__(self):
        d = self._m.fe_data.dofs
        return f"State:\n├── u: {d.nu} DOFs\n├── p: {d.np} DOFs\n└── b: {d.nb} DOFs"


class Model:
    """Model(arch, params, forcings, fe_data, inversion[, evolution, timestepper]) - src/model.jl:18-62; starts from
    rest."""

    def __init__(self, arch, params, forcings, fe_data, inversion: InversionToolkit, evolution: EvolutionToolkit = None,
                 timestepper=None):
        if not isinstance(arch, (GPU, CPU)):
            raise TypeError("Model: arch must be GPU() or CPU()")
        self.arch, self.params, self.forcings, self.fe_data = arch, params, forcings, fe_data
        self.inversion, self.evolution, self.timestepper = inversion, evolution, timestepper
        ctx = arch.ctx
        # buoyancy lives in the evolution solver's x when there is one
        self.b_vec = evolution.solver.x if evolution is not None else DeviceVector(ctx, fe_data.dofs.nb)
        self.state = State(self)
        self.step_index = 1
        # True / 1: run! starts each inversion from the extrapolation 2 x_{n-1} - x_{n-2} of the last two solutions instead of
        # x_{n-1} alone (the reference's warm start); 2: from the quadratic one through the last three; off by default = the
        # reference's recipe
        self.extrapolate_guess = False
        self.stats = []
        self._prev = None
        self._u_view = inversion.solver.x.view(0, fe_data.dofs.nu)     # x[1:nu] = u (p_inversion = [p_u; nu + p_p])

    def __repr__(self):
        return f"Model:\n├── arch: {self.arch}\n├── fe_data: {self.fe_data!r}\n└── timestepper: {self.timestepper!r}"


def set_b(model: Model, b):
    """set_b!(model, b::Function | b::AbstractArray) - src/model.jl:77-88 (array in native free-DoF order)"""
    s, d = model.fe_data.spaces, model.fe_data.dofs
    vals = s.interpolate_b(b) if callable(b) else np.asarray(b, dtype=float)
    if vals.shape != (d.nb,):
        raise ValueError(f"set_b: expected {d.nb} free values, got {vals.shape}")
    model.b_vec.upload(vals, d.p_b)
    return model


def invert(model: Model, b: DeviceVector = None):
    """invert!(model[, b]) - src/model.jl:302-309 (sync_flow! is implicit: the flow lives in solver.x)"""
    invert_toolkit(model.inversion, model.b_vec if b is None else b)
    return model


def sync_flow(model: Model):
    """sync_flow! - src/model.jl:311-317: returns (u, p) on the host in native order"""
    return model.state.u, model.state.p


def _scheme(ts):
    return L.NPG_BDF1 if isinstance(ts, BDF1) else L.NPG_BDF2


def evolve(model: Model, x_inv_prev: DeviceVector, b_prev: DeviceVector):
    """evolve!(model, u_prev, b_prev) - src/model.jl:213-285, device-only:
       [convection: kappa_v closure -> Kv, rhs_v, rhs_diff reassembly]  ->  [LHS + Jacobi rebuild]  ->
       advection assembly + RHS combination (one call)  ->  CG."""
    ev, ts, prm, frc = model.evolution, model.timestepper, model.params, model.forcings
    solver, fe = ev.solver, ev.fe
    theta = evolution_parameter(prm, ts)
    if frc.conv_param.is_on:
        cp = frc.conv_param
        fe.update_kappa_convection(cp.kappa_c, cp.N2min, prm.alpha, prm.N2, model.b_vec)       # src/model.jl:229-232
        if hasattr(ev, "Kv_full"):        # distributed: assemble the replicated matrix, gather this rank's row block
            fe.assemble(L.NPG_MAT_KV, ev.Kv_full, lift=ev.rhs_v)
            ev.Kv.gather_values(ev.Kv_full, ev.Kv_map)
        else:
            fe.assemble(L.NPG_MAT_KV, ev.Kv, lift=ev.rhs_v)                                    # src/model.jl:235
        fe.rhs_diff(prm.N2, ev.rhs_diff)                                                       # src/model.jl:237
    if frc.conv_param.is_on or ts.adaptive:
        collect_evolution_LHS_into(solver.A, solver.P, prm, ts, ev.M, ev.Kh, ev.Kv)            # src/model.jl:251-261
    x_inv = model.inversion.solver.x
    # (mesh-partitioned models: the element kernels write every local row, the solver reads the owned ones - a view)
    fe.evolution_rhs(_scheme(ts), ts.dt, prm.N2, theta, model.b_vec, b_prev, x_inv, x_inv_prev, ev.rhs_diff,
                     ev.rhs_flux, ev.rhs_M, ev.rhs_h, ev.rhs_v, getattr(solver, "y_full", solver.y))   # src/model.jl:269-278
    iterative_solve(solver)                                                                    # src/model.jl:279
    return model


def run(model: Model, n_info=10, n_save=float("inf"), n_plot=float("inf"), advection=True, n_steps=None, log=None):
    """run!(model; n_info, n_save, n_plot, advection) - src/model.jl:90-211.  `n_steps` (extension) bounds the number of
    steps taken by this call so that a caller can time a fixed number of steps; state carries over between calls.
    Every n_save steps the state goes to <out_dir>/data/state_<i>.jld2 and .vtu (src/model.jl:194-197, io.set_out_dir);
    n_plot is accepted and ignored: sim_plots is plotting (out of scope)."""
    ts, prm, frc = model.timestepper, model.params, model.forcings
    inv_x, b = model.inversion.solver.x, model.b_vec
    fe = model.evolution.fe
    ctx = model.arch.ctx
    comm = getattr(model, "comm", None)          # partition.PartitionedModel: host-level reductions over the ranks
    if model._prev is None:
        # copies of previous and current u, b (src/model.jl:119-123)
        model._prev = dict(x_prev=inv_x.copy(), b_prev=b.copy(), x_curr=inv_x.copy(), b_curr=b.copy())
        if isinstance(ts, BDF1):
            model._h_cells = model.h_cells() if hasattr(model, "h_cells") else model.fe_data.mesh.h_cells()   # src/model.jl:100
    pv = model._prev
    t0 = t_last = time.time()
    taken = 0
    while ts.t < ts.t_stop and (n_steps is None or taken < n_steps):
        i = model.step_index
        if isinstance(ts, BDF1):
            update_dt(ts, fe, inv_x, h_cells=model.__dict__.pop("_h_cells", None))              # src/model.jl:131
            if comm is not None:
                ts.dt = comm.min(ts.dt)                           # partitioned mesh: every rank saw its own cells only
        if i == 2 and isinstance(ts, BDF2):
            collect_evolution_LHS(model.evolution, prm, frc, ts)                                # src/model.jl:134-137
        pv["x_curr"].copy_from(inv_x)                                                           # src/model.jl:140-141
        pv["b_curr"].copy_from(b)
        evolve(model, pv["x_prev"], pv["b_prev"])                                               # src/model.jl:144
        order = int(getattr(model, "extrapolate_guess", 0) or 0)
        if order >= 2 and not getattr(model, "_warned_quadratic_guess", False):
            import warnings
            model._warned_quadratic_guess = True
            warnings.warn("extrapolate_guess=2 (quadratic initial guess) is kept for experiments only: on bowl3D h = 0.02 restarted "
                          "GMRES(20) stagnated on what it leaves of the residual (up to 240 997 iterations in one step, "
                          "profiles/r03_bench_quadratic_guess_bowl3D_h0.02.json); use extrapolate_guess=1", RuntimeWarning)
        if order >= 2 and i > 2 and "x_prev2" in pv:
            # ... or the quadratic one, 3 x_{n-1} - 3 x_{n-2} + x_{n-3}
            inv_x.axpby(-3.0, pv["x_prev"], 3.0)
            inv_x.axpby(1.0, pv["x_prev2"], 1.0)
        elif order >= 1 and i > 1:
            # initial guess 2 x_{n-1} - x_{n-2} instead of the reference's x_{n-1} (x aliases workspace.x,
            # src/iterative_solvers.jl:26-29): changes the result only within the solver tolerance
            inv_x.axpby(-1.0, pv["x_prev"], 2.0)
        invert(model)                                                                           # src/model.jl:145
        update_t(ts)
        # blow-up guard (src/model.jl:149-153): device reductions over [u; p] and b
        xm, xnan = model._u_view.maxabs()
        bm, bnan = b.maxabs()
        blow = max(xm, bm) > 1e3 or xnan or bnan
        if comm is not None:
            blow = comm.any(blow)                                 # every rank leaves the loop together
        if blow:
            raise BlowUp("Blow-up detected, stopping simulation")
        if int(getattr(model, "extrapolate_guess", 0) or 0) >= 2:         # x_{n-2} of this step is x_{n-3} of the next
            if "x_prev2" not in pv:
                pv["x_prev2"] = pv["x_prev"].copy()
            else:
                pv["x_prev2"].copy_from(pv["x_prev"])
        pv["x_prev"], pv["x_curr"] = pv["x_curr"], pv["x_prev"]                                 # src/model.jl:156-157
        pv["b_prev"], pv["b_curr"] = pv["b_curr"], pv["b_prev"]
        if frc.eddy_param.is_on and advection and i % 10 == 0:                                  # src/model.jl:160-170
            ep = frc.eddy_param
            fe.update_nu_eddy(ep.N2min, prm.alpha, prm.N2, b)
            sol = model.inversion.solver
            if hasattr(model, "reassemble_A"):    # mesh-partitioned: the rank's cells into the rank's rows
                model.reassemble_A()
            elif hasattr(sol, "A_full"):    # distributed: re-assemble the replicated matrix, gather this rank's row block
                build_A_inversion(model.arch, model.fe_data, prm, None, A=sol.A_full)
                sol.A.gather_values(sol.A_full, sol.A_map)
            else:
                build_A_inversion(model.arch, model.fe_data, prm, None, A=sol.A)
            if hasattr(model.inversion.solver.P, "refresh"):         # an operator-dependent preconditioner follows A
                model.inversion.solver.P.refresh(model.inversion.solver.A, model)
        model.stats.append((model.evolution.solver.workspace.stats, model.inversion.solver.workspace.stats))
        if i % n_info == 0 and log is not None:
            t1 = time.time()
            log(f"t = {ts.t:.3e}/{ts.t_stop:.3e} (i = {i}, Δt = {ts.dt:.3e}); step ~ {(t1 - t_last) / n_info:.3e} s; "
                f"|u|max = {xm:.3e}; GMRES it = {model.stats[-1][1]['niter']}, CG it = {model.stats[-1][0]['niter']}")
            t_last = t1
        if n_save != float("inf") and i % int(n_save) == 0:                                     # src/model.jl:194-197
            from . import io as _io
            _io.save_checkpoint(model, i)                         # collective for distributed models; rank 0 writes
        model.step_index += 1
        taken += 1
    ctx.sync()
    return model
