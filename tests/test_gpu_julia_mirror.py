"""Line-for-line Python twin of ext/nuPGCMHIPExt.jl (which cannot run here: no Julia in the image).

Each function below carries the name of the Julia method it mirrors and issues the SAME C-ABI calls in the SAME order with the
SAME argument conventions: Julia's 1-based CSC arrays and permutations shifted at the call (`A.colptr .- 1`, `perm .- 1`),
vectors handed over in the reference's NATIVE (Gridap) DoF order with HOST permutations (src/dofs.jl:27-41), Gridap-style
cell DoF tables (free ids > 0, Dirichlet ids < 0) composed with the inverse permutations exactly as `devidx` does.  The
"reference side" is played by the fixture-pinned oracle (it holds what Gridap would: native-order CSC matrices, tables,
an RCM of its own), so this drives libnupgcm_hip.so the way the reference package would - raw ctypes, none of the
nupgcm_amd host classes - and checks three timesteps of run! against the oracle's direct-solve recipe."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

from nupgcm_amd import _lib as L  # noqa: E402  (the ctypes declarations of include/nupgcm_hip.h, nothing else)
from oracle import fe_oracle as fo  # noqa: E402
from oracle import recipe as rc  # noqa: E402

lib = L.lib
P64 = lambda a: a.ctypes.data_as(C.c_void_p)


class Ext:
    """module nuPGCMHIPExt"""

    def __init__(self):
        out = C.c_void_p()
        L.check(lib().npg_ctx_create(0, C.byref(out)))                       # ctx()
        self.ctx = out

    # ---- the ten methods of ext/nuPGCMCUDAExt.jl:24-33 ---------------------------------------------------------------
    def HIPVector(self, n):                                                  # HIPVector{Float64}(undef, n)
        out = C.c_void_p()
        L.check(lib().npg_vec_create(self.ctx, int(n), C.byref(out)))
        return out

    def on_architecture_GPU_Array(self, a):                                  # on_architecture(::GPU, a::Array{Float64})
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = self.HIPVector(len(a))
        L.check(lib().npg_vec_upload(v, P64(a)))
        return v

    def on_architecture_CPU_HIPVector(self, v):                              # on_architecture(::CPU, v::HIPVector)
        a = np.empty(lib().npg_vec_len(v))
        L.check(lib().npg_vec_download(v, P64(a)))
        return a

    def on_architecture_GPU_SparseMatrixCSC(self, colptr1, rowval1, nzval, m, n):
        """on_architecture(::GPU, A::SparseMatrixCSC{Float64,Int64}): the arrays arrive 1-based, as Julia stores them"""
        out = C.c_void_p()
        cp, rv = np.ascontiguousarray(colptr1 - 1, dtype=np.int64), np.ascontiguousarray(rowval1 - 1, dtype=np.int64)
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        L.check(lib().npg_csr_create_from_csc(self.ctx, m, n, P64(cp), P64(rv), P64(nz), 0, C.byref(out)))   # upload_csc(A, 0)
        return out

    def on_architecture_CPU_HIPSparseMatrixCSR(self, A):                     # back to a 1-based SparseMatrixCSC
        m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(lib().npg_csr_shape(A, C.byref(m), C.byref(n), C.byref(nnz)))
        cp, rv, nz = np.empty(n.value + 1, np.int64), np.empty(nnz.value, np.int64), np.empty(nnz.value)
        L.check(lib().npg_csr_to_csc(A, P64(cp), P64(rv), P64(nz)))
        return cp + 1, rv + 1, nz, m.value, n.value

    # ---- Krylov.GmresWorkspace / CgWorkspace (N, N, HIPVector{Float64}) ---------------------------------------------------
    def GmresWorkspace(self, n, memory=20):
        out = C.c_void_p()
        L.check(lib().npg_gmres_create(self.ctx, int(n), int(memory), C.byref(out)))
        return dict(h=out, x=self.HIPVector(n), kind="gmres", stats=None)

    def CgWorkspace(self, n):
        out = C.c_void_p()
        L.check(lib().npg_cg_create(self.ctx, int(n), C.byref(out)))
        return dict(h=out, x=self.HIPVector(n), kind="cg", stats=None)

    def precond_args(self, Pdiag):                                           # precond_args(P::Diagonal{Float64,<:HIPVector})
        val, flag = C.c_double(), C.c_int()
        L.check(lib().npg_vec_is_constant(Pdiag, C.byref(val), C.byref(flag)))
        return (1, val.value, None) if flag.value else (2, 0.0, Pdiag)

    def iterative_solve(self, tk):                         # iterative_solve!(tk::IterativeSolverToolkit{<:HIPSparseMatrixCSR})
        ws, kw = tk["workspace"], tk["kwargs"]
        kind, scalar, dh = self.precond_args(tk["P"])
        st = L.SolveStats()
        if ws["kind"] == "gmres":
            L.check(lib().npg_gmres_solve(ws["h"], tk["A"], kind, scalar, dh, tk["y"], tk["x"], kw["atol"], kw["rtol"],
                                          kw["itmax"], 0.1, C.byref(st)))
        else:
            L.check(lib().npg_cg_solve(ws["h"], tk["A"], kind, scalar, dh, tk["y"], tk["x"], kw["atol"], kw["rtol"],
                                       kw["itmax"], C.byref(st)))
        ws["stats"] = st.as_dict()
        return tk

    def getindex_perm(self, x, perm1):                                       # Base.getindex(x::HIPVector, perm::Vector{Int})
        a = np.empty(len(perm1))
        pm = np.ascontiguousarray(perm1 - 1, dtype=np.int64)
        L.check(lib().npg_vec_download_perm(x, P64(a), P64(pm)))
        return a

    def upload_perm(self, a, perm1):                                         # upload_perm(a, perm)
        v = self.HIPVector(len(perm1))
        a = np.ascontiguousarray(a, dtype=np.float64)
        pm = np.ascontiguousarray(perm1 - 1, dtype=np.int64)
        L.check(lib().npg_vec_upload_perm(v, P64(a), P64(pm)))
        return v

    # ---- the reference's own (generic) constructors, which now only meet methods defined above --------------------------------
    def IterativeSolverToolkit(self, A, P, y, workspace, kwargs, label):     # src/iterative_solvers.jl:26-29
        return dict(A=A, P=P, x=workspace["x"], y=y, workspace=workspace, kwargs=kwargs, label=label)

    def InversionToolkit(self, A, P, B, b, n, atol=1e-6, rtol=1e-6, itmax=0, memory=20):     # src/inversion.jl:74-94
        y = self.on_architecture_GPU_Array(np.zeros(n))
        ws = self.GmresWorkspace(n, memory)
        L.check(lib().npg_vec_fill(ws["x"], 0.0))                            # workspace.x .= zero(T) -> fill!
        return dict(B=B, b=b, solver=self.IterativeSolverToolkit(A, P, y, ws, dict(atol=atol, rtol=rtol, itmax=itmax),
                                                                 "Inversion"))

    def invert(self, inversion, b_free_values):            # invert!(inversion::InversionToolkit{<:HIPSparseMatrixCSR}, b)
        s = inversion["solver"]
        bd = self.on_architecture_GPU_Array(b_free_values)                   # native order
        L.check(lib().npg_vec_copy(s["y"], inversion["b"]))
        L.check(lib().npg_spmv(inversion["B"], bd, s["y"], 1.0, 1.0))
        self.iterative_solve(s)
        return inversion

    # ---- hip_fe(fe_data) ---------------------------------------------------------------------------------------------------
    @staticmethod
    def devidx(ids, inv_perm1, off=0):
        """free id k > 0 -> off + inv_perm[k] - 1; Dirichlet id -k < 0 stays -k"""
        ids = np.asarray(ids)
        return np.where(ids > 0, off + inv_perm1[np.maximum(ids, 1) - 1] - 1, ids).astype(np.int32)

    def hip_fe(self, g):
        d = L.FeDesc(ncell=g["nc"], nq=len(g["qw"]), nloc_b=10, grad_lambda=g["G"].ctypes.data, wdet=g["wdet"].ctypes.data,
                     qw=g["qw"].ctypes.data, N2=g["N2"].ctypes.data, dN2=g["dN2"].ctypes.data, Nb=g["N2"].ctypes.data,
                     dNb=g["dN2"].ctypes.data, N1=g["lam"].ctypes.data, cell_u=g["cu"].ctypes.data, cell_p=g["cp"].ctypes.data,
                     cell_b=g["cb"].ctypes.data, u_diri=g["ud"].ctypes.data, n_u_diri=len(g["ud"]), b_diri=g["bd"].ctypes.data,
                     n_b_diri=len(g["bd"]), n_inv=g["n_inv"], n_b=g["n_b"])
        out = C.c_void_p()
        L.check(lib().npg_fe_create(self.ctx, C.byref(d), C.byref(out)))
        return out

    def block_nodes(self, A, node, comp, rtol=1e-12):                        # block_nodes!(A::HIPSparseMatrixCSR, fe_data)
        node = np.ascontiguousarray(node, dtype=np.int64)
        comp = np.ascontiguousarray(comp, dtype=np.int32)
        flag = C.c_int()
        L.check(lib().npg_csr_block_nodes_dofs(A, P64(node), P64(comp), rtol, C.byref(flag)))
        return flag.value != 0

    def upload_csc0(self, A):                                                # `up` of device_evolution_matrices: drop_zeros = 0
        cp1, rv1, nz, m, n = _csc1(A)
        out = C.c_void_p()
        cp, rv = np.ascontiguousarray(cp1 - 1), np.ascontiguousarray(rv1 - 1)
        L.check(lib().npg_csr_create_from_csc(self.ctx, m, n, P64(cp), P64(rv), P64(np.ascontiguousarray(nz)), 0, C.byref(out)))
        return out

    def device_evolution_matrices(self, model):                              # device_evolution_matrices(ev, model)
        ev = model["evolution"]
        if "dm" not in ev:
            pattern = (abs(ev["M_host"]) + abs(ev["Kh_host"]) + abs(ev["Kv_host"])).tocsc()
            pattern.data[:] = 1.0

            def onpat(A):                                                    # A's values on the common pattern (explicit zeros kept)
                B = pattern.copy()
                B.data[:] = 0.0
                B = (B + sp.csc_matrix(A)).tocsc()
                out = pattern.copy().tocsc()
                out.sort_indices()
                B.sort_indices()
                full = sp.csc_matrix((np.zeros(out.nnz), out.indices, out.indptr), shape=out.shape)
                # scatter B's entries into the pattern's slots
                pos = {}
                for j in range(out.shape[1]):
                    lo, hi = out.indptr[j], out.indptr[j + 1]
                    pos.update({(int(i), j): lo + k for k, i in enumerate(out.indices[lo:hi])})
                for j in range(B.shape[1]):
                    for k in range(B.indptr[j], B.indptr[j + 1]):
                        full.data[pos[(int(B.indices[k]), j)]] = B.data[k]
                return full
            kv = np.ascontiguousarray(model["kappa_v0_q"], dtype=np.float64)
            L.check(lib().npg_fe_set_coeff(model["fe"], b"kappa_v", P64(kv)))
            ev["solver"]["A"] = self.upload_csc0(onpat(ev["A_host"]))
            ev["dm"] = dict(M=self.upload_csc0(onpat(ev["M_host"])), Kh=self.upload_csc0(onpat(ev["Kh_host"])),
                            Kv=self.upload_csc0(onpat(ev["Kv_host"])))
        return ev["dm"]

    def update_dt(self, model, h_cells, u_min=0.01):                         # update_Δt!(ts::BDF1, u, dΩ, h_cells::Vector{Float64})
        out = C.c_double()
        hc = np.ascontiguousarray(h_cells, dtype=np.float64)
        L.check(lib().npg_fe_cfl_ratio(model["fe"], P64(hc), u_min, model["inversion"]["solver"]["x"], C.byref(out)))
        model["dt"] = model["CFL_factor"] * out.value
        return model

    def evolve(self, model, u_prev, b_prev):                                 # evolve!(model::HIPModel, u_prev, b_prev)
        dofs, ev = model["dofs"], model["evolution"]
        solver = ev["solver"]
        fe = model["fe"]
        xi = lambda u: self.upload_perm(np.concatenate([u, np.zeros(dofs["np"])]), dofs["p_inversion"])
        bv = lambda b: self.upload_perm(b, dofs["p_b"])
        scheme = model.get("scheme", 2)
        if model.get("conv") is not None or model.get("adaptive"):
            theta = model["theta_of_dt"](model["dt"])
            model["theta"] = theta
            dm = self.device_evolution_matrices(model)
            if model.get("conv") is not None:
                kc, n2min = model["conv"]
                L.check(lib().npg_fe_update_kappa_convection(fe, None, kc, n2min, model["alpha"], model["N2"], bv(model["b"])))
                L.check(lib().npg_fe_assemble_matrix(fe, L.NPG_MAT_KV, 1.0, 0, dm["Kv"], ev["rhs_v"]))
                L.check(lib().npg_fe_assemble_rhs_diff(fe, model["N2"], ev["rhs_diff"]))
            L.check(lib().npg_csr_combine(solver["A"], 1.0, dm["M"], theta, dm["Kh"], dm["Kv"]))
            L.check(lib().npg_csr_inv_diag(solver["A"], solver["P"]))
        L.check(lib().npg_fe_evolution_rhs(fe, scheme, model["dt"], model["N2"], model["theta"], bv(model["b"]), bv(b_prev),
                                           xi(model["u"]), xi(u_prev), ev["rhs_diff"], ev["rhs_flux"], ev["rhs_M"],
                                           ev["rhs_h"], ev["rhs_v"], solver["y"]))
        self.iterative_solve(solver)
        model["b"] = self.getindex_perm(solver["x"], dofs["inv_p_b"])       # b.free_values .= solver.x[inv_perm]
        return model


def _csc1(A):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int64) + 1, A.indices.astype(np.int64) + 1, A.data.copy(), A.shape[0], A.shape[1]


def _gridap_tables(S):
    """what hip_fe reads from Gridap, from the oracle's spaces: 1-based free ids, negative Dirichlet ids, value arrays"""
    o, s = S.orc, S.orc.sp
    nn = len(s.u_dof)
    ud_id = np.zeros((nn, 3), dtype=np.int64)
    free = s.u_dof >= 0
    ud_id[free] = s.u_dof[free] + 1
    ud_id[~free] = -(np.arange((~free).sum()) + 1)
    bd_id = np.where(s.b_dof >= 0, s.b_dof + 1, 0)
    nd = np.nonzero(s.b_dof < 0)[0]
    bd_id[nd] = -(np.arange(len(nd)) + 1)
    pid = np.where(s.p_dof >= 0, s.p_dof + 1, -1)
    return dict(cell_u=ud_id[o.cn2].reshape(len(o.cn2), 30), cell_p=pid[o.topo.cells], cell_b=bd_id[o.cn2],
                u_diri=np.zeros((~free).sum()), b_diri=s.b_diri[nd])


def test_three_timesteps_through_the_julia_binding_sequence():
    S = rc.setup("bowl_diri")                                               # non-trivial Dirichlet buoyancy (b = y)
    o, s = S.orc, S.orc.sp
    nu, np_, nb = s.nu, s.np_, s.nb
    N = nu + np_
    p_inv0, p_b0 = rc.rcm_perms(S)                                           # the host's RCM (src/dofs.jl:70-100)
    p_inversion, p_b = p_inv0 + 1, p_b0 + 1                                  # Julia holds them 1-based
    inv_p_inversion, inv_p_b = np.argsort(p_inv0) + 1, np.argsort(p_b0) + 1
    ext = Ext()
    # InversionToolkit(arch, fe_data, params, forcings): host permutes, then on_architecture (src/inversion.jl:37-65)
    A = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(S.A[p_inv0][:, p_inv0]))
    B = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(S.B[p_inv0]))
    b0 = ext.on_architecture_GPU_Array(S.b0[p_inv0])
    h, _ = o.precond_h()
    Pinv = ext.on_architecture_GPU_Array(np.full(N, 1 / h ** 3))             # Diagonal(on_architecture(arch, 1/h^dim*ones(N)))
    assert ext.precond_args(Pinv)[0] == 1
    inversion = ext.InversionToolkit(A, Pinv, B, b0, N, atol=1e-10, rtol=1e-10)
    # round trip of on_architecture(CPU(), A): values AND Gridap's explicit zeros survive (the reference re-assembles into this pattern,
    # src/model.jl:112-113,166)
    cp1, rv1, nz, m, n = ext.on_architecture_CPU_HIPSparseMatrixCSR(A)
    back = sp.csc_matrix((nz, rv1 - 1, cp1 - 1), shape=(m, n))
    assert abs(back - S.A[p_inv0][:, p_inv0]).max() == 0 and len(nz) == S.A.nnz and (nz == 0.0).sum() > 0.2 * len(nz)
    # EvolutionToolkit (src/evolution.jl:55-131): host permutes by p_b, uploads the five vectors, BDF1 LHS for the first step
    th1, th2 = S.theta("BDF1"), S.theta("BDF2")
    perm = lambda Mx: Mx[p_b0][:, p_b0]

    def collect_evolution_LHS(theta):                                        # src/evolution.jl:143-177 on the host, then upload
        Ah = (perm(S.M) + theta * (perm(S.Kh) + perm(S.Kv))).tocsc()
        return (ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(Ah)),
                ext.on_architecture_GPU_Array(1.0 / Ah.diagonal()))
    Ae, Pe = collect_evolution_LHS(th1)
    assert ext.precond_args(Pe)[0] == 2
    ws = ext.CgWorkspace(nb)
    up = ext.on_architecture_GPU_Array
    evo = dict(rhs_diff=up(S.rhs_diff[p_b0]), rhs_flux=up(S.rhs_flux[p_b0]), rhs_M=up(S.rhs_M[p_b0]), rhs_h=up(S.rhs_h[p_b0]),
               rhs_v=up(S.rhs_v[p_b0]),
               solver=ext.IterativeSolverToolkit(Ae, Pe, up(np.zeros(nb)), ws, dict(atol=1e-12, rtol=1e-12, itmax=0),
                                                 "Evolution"))
    # hip_fe(fe_data): Gridap tables -> npg_fe_desc
    t = _gridap_tables(S)
    _, dN = fo.p2_basis(o.geo.lam, fo.TET_EDGES)
    g = dict(nc=len(o.topo.cells), G=np.ascontiguousarray(o.geo.G.reshape(-1)), wdet=np.ascontiguousarray(o.geo.detJ),
             qw=np.ascontiguousarray(o.geo.w), lam=np.ascontiguousarray(o.geo.lam), N2=np.ascontiguousarray(o.N2q),
             dN2=np.ascontiguousarray(dN), n_inv=N, n_b=nb,
             cu=np.ascontiguousarray(Ext.devidx(t["cell_u"], inv_p_inversion)),
             cp=np.ascontiguousarray(np.where(t["cell_p"] > 0, inv_p_inversion[nu + np.maximum(t["cell_p"], 1) - 1] - 1, -1)
                                     .astype(np.int32)),
             cb=np.ascontiguousarray(Ext.devidx(t["cell_b"], inv_p_b)),
             ud=np.ascontiguousarray(t["u_diri"]), bd=np.ascontiguousarray(t["b_diri"]))
    fe = ext.hip_fe(g)
    # run!: 3 BDF2 steps (src/model.jl:119-157), state in host vectors in native order as the reference keeps it
    model = dict(dofs=dict(np=np_, p_inversion=p_inversion, p_b=p_b, inv_p_b=inv_p_b), evolution=evo, fe=fe, dt=S.dt,
                 N2=o.N2, theta=th2, b=o.interpolate_b(S.cfg["b0"]), u=np.zeros(nu))
    u_prev, b_prev = model["u"].copy(), model["b"].copy()
    p = None
    for i in (1, 2, 3):
        if i == 2:                                                           # collect_evolution_LHS! (src/model.jl:134-137)
            evo["solver"]["A"], evo["solver"]["P"] = collect_evolution_LHS(th2)
        u_curr, b_curr = model["u"].copy(), model["b"].copy()
        ext.evolve(model, u_prev, b_prev)
        ext.invert(inversion, model["b"])
        x = ext.getindex_perm(inversion["solver"]["x"], inv_p_inversion)     # sync_flow! (src/model.jl:311-317)
        model["u"], p = x[:nu], x[nu:]
        u_prev, b_prev = u_curr, b_curr
        assert inversion["solver"]["workspace"]["stats"]["solved"] == 1 and ws["stats"]["solved"] == 1
    u, pr, b = rc.run(S, 3, solver="direct")
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    assert rel(model["b"], b) < 1e-8 and rel(model["u"], u) < 1e-6 and rel(p, pr) < 1e-6, \
        (rel(model["b"], b), rel(model["u"], u), rel(p, pr))


def _velocity_dof_nodes(s):
    """velocity_dof_nodes(fe_data): (node, component) of every free velocity DoF in the native numbering"""
    nodes, comps = np.nonzero(s.u_dof >= 0)
    node = np.full(s.nu, -1, dtype=np.int64)
    comp = np.zeros(s.nu, dtype=np.int32)
    node[s.u_dof[nodes, comps]] = nodes
    comp[s.u_dof[nodes, comps]] = comps
    return node, comp


def test_node_blocks_and_windowed_tiles_from_the_reference_order():
    """Model(arch::GPU, ...) of the extension: A_inversion arrives as the reference uploads it - CSC, rows and columns in ITS
    p_inversion (an RCM that comes out component by component, src/dofs.jl:70-100) - and npg_csr_block_nodes_dofs, given
    Gridap's (node, component) of every velocity DoF, stores it by node blocks with a windowed tile set while every vector keeps
    the caller's order: npg_spmv equals the host product, npg_gmres_solve the plain matrix's solve.  At 135 k unknowns, where
    the record layouts are in play (they start at 100 k rows)."""
    from nupgcm_amd import workloads                                         # (mesh only: refine.py; the FE side is the oracle's)
    S = rc.setup("example", model=workloads.bowl_mesh_model("bowl3D_h0.05"))
    o, s = S.orc, S.orc.sp
    nu, N = s.nu, s.nu + s.np_
    assert N >= 100000
    p_inv0, _ = rc.rcm_perms(S)
    A_host = sp.csr_matrix(S.A[p_inv0][:, p_inv0])
    ext = Ext()
    A = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(A_host))
    A_plain = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(A_host))
    node_native, comp_native = _velocity_dof_nodes(s)
    node, comp = np.full(N, -1, dtype=np.int64), np.zeros(N, dtype=np.int32)
    node[:nu], comp[:nu] = node_native[p_inv0[:nu]], comp_native[p_inv0[:nu]]
    # the caller's order is NOT node-blocked: hardly any node has its x, y, z rows next to each other
    together = (node[:nu - 2] == node[1:nu - 1]) & (node[:nu - 2] == node[2:nu]) & (comp[:nu - 2] == 0) & (comp[1:nu - 1] == 1) & (comp[2:nu] == 2)
    assert together.sum() < 0.05 * len(s.u_dof)
    assert ext.block_nodes(A, node, comp)
    a, b_, c = C.c_int64(), C.c_int64(), C.c_int64()
    L.check(lib().npg_csr_storage(A, C.byref(a), C.byref(b_), C.byref(c)))
    assert a.value > 0.9 * len(s.u_dof) * 0.8 and b_.value > 10 * a.value and c.value < 0.01 * A_host.nnz
    wt = [C.c_int64() for _ in range(4)]
    L.check(lib().npg_csr_window_info(A, *[C.byref(w) for w in wt]))
    assert wt[0].value > 0 and wt[1].value > 0
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N)
    dx, dy, dy0 = ext.on_architecture_GPU_Array(x), ext.HIPVector(N), ext.HIPVector(N)
    L.check(lib().npg_spmv(A, dx, dy, 1.0, 0.0))
    L.check(lib().npg_spmv(A_plain, dx, dy0, 1.0, 0.0))
    want = A_host @ x
    rel = lambda u, v: np.linalg.norm(u - v) / np.linalg.norm(v)
    y1 = ext.on_architecture_CPU_HIPVector(dy)
    assert rel(y1, want) < 1e-13 and rel(ext.on_architecture_CPU_HIPVector(dy0), want) < 1e-13
    L.check(lib().npg_spmv(A, dx, dy, -0.5, 2.0))                           # mul!(y, A, x, alpha, beta) in the caller's order
    assert rel(ext.on_architecture_CPU_HIPVector(dy), 2.0 * y1 - 0.5 * want) < 1e-13
    # the solve, through the binding sequence: same iterations (to the few per cent any perturbation moves them), same answer
    h, _ = o.precond_h()
    rhs = A_host @ np.cos(np.arange(N, dtype=float)) * 1e-3
    out = {}
    for name, M in (("blocked", A), ("plain", A_plain)):
        Pinv = ext.on_architecture_GPU_Array(np.full(N, 1 / h ** 3))
        inv = ext.InversionToolkit(M, Pinv, None, None, N)
        tk = inv["solver"]
        L.check(lib().npg_vec_upload(tk["y"], P64(np.ascontiguousarray(rhs))))
        ext.iterative_solve(tk)
        st = tk["workspace"]["stats"]
        xs = ext.on_architecture_CPU_HIPVector(tk["x"])
        assert st["solved"] == 1 and np.linalg.norm((rhs - A_host @ xs) / h ** 3) <= 1.5 * (1e-6 + 1e-6 * st["rnorm0"])
        L.check(lib().npg_vec_upload(tk["y"], P64(np.ascontiguousarray(1.01 * rhs))))
        ext.iterative_solve(tk)                                              # warm start from workspace.x, in the caller's order
        st2 = tk["workspace"]["stats"]
        assert st2["solved"] == 1 and st2["niter"] < st["niter"]
        out[name] = (st["niter"], xs)
    assert abs(out["blocked"][0] - out["plain"][0]) <= 0.08 * out["plain"][0], (out["blocked"][0], out["plain"][0])
    assert rel(out["blocked"][1], out["plain"][1]) < 5e-4


def test_closures_and_cfl_step_through_the_julia_binding_sequence():
    """evolve!(model::HIPModel) with the convection closure and update_Δt! of the extension: kappa_v from the current buoyancy,
    K_v + lift, rhs_diff, the LHS M + theta (K_h + K_v) and its Jacobi diagonal, and the CFL step are device calls (raw
    ctypes, the extension's sequence); four BDF1 steps with the adaptive step against the oracle's restatement of
    src/model.jl:229-261 + src/timesteppers.jl:108-119."""
    S = rc.setup("bowl_diri")
    o, s = S.orc, S.orc.sp
    nu, np_, nb = s.nu, s.np_, s.nb
    N = nu + np_
    conv, cfl = (0.5, 0.5), 0.5
    p_inv0, p_b0 = rc.rcm_perms(S)
    p_inversion, p_b = p_inv0 + 1, p_b0 + 1
    inv_p_inversion, inv_p_b = np.argsort(p_inv0) + 1, np.argsort(p_b0) + 1
    ext = Ext()
    A = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(S.A[p_inv0][:, p_inv0]))
    B = ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(S.B[p_inv0]))
    b0 = ext.on_architecture_GPU_Array(S.b0[p_inv0])
    h, _ = o.precond_h()
    inversion = ext.InversionToolkit(A, ext.on_architecture_GPU_Array(np.full(N, 1 / h ** 3)), B, b0, N, atol=1e-10, rtol=1e-10)
    perm = lambda Mx: sp.csc_matrix(Mx)[p_b0][:, p_b0]
    th1 = S.theta("BDF1")
    Ah = (perm(S.M) + th1 * (perm(S.Kh) + perm(S.Kv))).tocsc()
    up = ext.on_architecture_GPU_Array
    ws = ext.CgWorkspace(nb)
    evo = dict(rhs_diff=up(S.rhs_diff[p_b0]), rhs_flux=up(S.rhs_flux[p_b0]), rhs_M=up(S.rhs_M[p_b0]), rhs_h=up(S.rhs_h[p_b0]),
               rhs_v=up(S.rhs_v[p_b0]), M_host=perm(S.M), Kh_host=perm(S.Kh), Kv_host=perm(S.Kv), A_host=Ah,
               solver=ext.IterativeSolverToolkit(ext.on_architecture_GPU_SparseMatrixCSC(*_csc1(Ah)), up(1.0 / Ah.diagonal()),
                                                 up(np.zeros(nb)), ws, dict(atol=1e-12, rtol=1e-12, itmax=0), "Evolution"))
    t = _gridap_tables(S)
    _, dN = fo.p2_basis(o.geo.lam, fo.TET_EDGES)
    g = dict(nc=len(o.topo.cells), G=np.ascontiguousarray(o.geo.G.reshape(-1)), wdet=np.ascontiguousarray(o.geo.detJ),
             qw=np.ascontiguousarray(o.geo.w), lam=np.ascontiguousarray(o.geo.lam), N2=np.ascontiguousarray(o.N2q),
             dN2=np.ascontiguousarray(dN), n_inv=N, n_b=nb,
             cu=np.ascontiguousarray(Ext.devidx(t["cell_u"], inv_p_inversion)),
             cp=np.ascontiguousarray(np.where(t["cell_p"] > 0, inv_p_inversion[nu + np.maximum(t["cell_p"], 1) - 1] - 1, -1)
                                     .astype(np.int32)),
             cb=np.ascontiguousarray(Ext.devidx(t["cell_b"], inv_p_b)),
             ud=np.ascontiguousarray(t["u_diri"]), bd=np.ascontiguousarray(t["b_diri"]))
    fe = ext.hip_fe(g)
    dt0 = S.dt
    c = S.c
    model = dict(dofs=dict(np=np_, p_inversion=p_inversion, p_b=p_b, inv_p_b=inv_p_b), evolution=evo, inversion=inversion, fe=fe,
                 dt=dt0, N2=o.N2, alpha=o.alpha, theta=th1, theta_of_dt=lambda dt: dt * c, scheme=1, conv=conv, adaptive=True,
                 CFL_factor=cfl, kappa_v0_q=fo._const_or_fn(o.kappa_v, o.geo.xq), b=o.interpolate_b(S.cfg["b0"]), u=np.zeros(nu))
    u_prev, b_prev = model["u"].copy(), model["b"].copy()
    h_cells = o.h_cells()
    dts = []
    for i in range(1, 5):
        ext.update_dt(model, h_cells)                                        # src/model.jl:131
        dts.append(model["dt"])
        u_curr, b_curr = model["u"].copy(), model["b"].copy()
        ext.evolve(model, u_prev, b_prev)
        ext.invert(inversion, model["b"])
        x = ext.getindex_perm(inversion["solver"]["x"], inv_p_inversion)
        model["u"], p = x[:nu], x[nu:]
        u_prev, b_prev = u_curr, b_curr
        assert inversion["solver"]["workspace"]["stats"]["solved"] == 1 and ws["stats"]["solved"] == 1
    S2 = rc.setup("bowl_diri")
    u, pr, b = rc.run(S2, 4, solver="direct", scheme="BDF1", cfl_factor=cfl, adaptive=True, conv=conv)
    rel = lambda a, c_: np.linalg.norm(a - c_) / np.linalg.norm(c_)
    assert abs(dts[-1] - S2.dt) <= 1e-6 * S2.dt and len(set(np.round(dts, 12))) > 1      # the step really adapted
    assert rel(model["b"], b) < 1e-7 and rel(model["u"], u) < 1e-5 and rel(p, pr) < 1e-5, \
        (rel(model["b"], b), rel(model["u"], u), rel(p, pr))


def _hip_fe_inputs(S, inv_p_inversion, inv_p_b):
    """the tables hip_fe(fe_data) reads from Gridap, composed with the inverse permutations as `devidx` does"""
    o, s = S.orc, S.orc.sp
    t = _gridap_tables(S)
    _, dN = fo.p2_basis(o.geo.lam, fo.TET_EDGES)
    return dict(nc=len(o.topo.cells), G=np.ascontiguousarray(o.geo.G.reshape(-1)), wdet=np.ascontiguousarray(o.geo.detJ),
                qw=np.ascontiguousarray(o.geo.w), lam=np.ascontiguousarray(o.geo.lam), N2=np.ascontiguousarray(o.N2q),
                dN2=np.ascontiguousarray(dN), n_inv=s.nu + s.np_, n_b=s.nb,
                cu=np.ascontiguousarray(Ext.devidx(t["cell_u"], inv_p_inversion)),
                cp=np.ascontiguousarray(np.where(t["cell_p"] > 0, inv_p_inversion[s.nu + np.maximum(t["cell_p"], 1) - 1] - 1, -1)
                                        .astype(np.int32)),
                cb=np.ascontiguousarray(Ext.devidx(t["cell_b"], inv_p_b)),
                ud=np.ascontiguousarray(t["u_diri"]), bd=np.ascontiguousarray(t["b_diri"]))


def test_eddy_refresh_through_the_julia_binding_sequence():
    """build_A_inversion!(A::SparseMatrixCSC, dup, dvq, assembler, fe_data, params, nu) of the extension (the eddy closure's refresh
    every tenth step, src/model.jl:160-170): `eddy_state` uploads the host matrix run! holds - native order permuted by p_inversion,
    Gridap's structural pattern with its explicit zeros - and sets the Coriolis table; the refresh itself is nu_eddy at the quadrature
    points from the current buoyancy (native order, permuted at the upload) and the full-stress assembly IN PLACE.  Checked against
    the oracle's restatement of src/inputs.jl:130-137 + src/inversion.jl:172-181 entry by entry, twice (two buoyancy fields into
    the same device matrix), and through a solve with the refreshed matrix."""
    S = rc.setup("bowl_diri")
    o, s = S.orc, S.orc.sp
    nu, N = s.nu, s.nu + s.np_
    p_inv0, p_b0 = rc.rcm_perms(S)
    p_b = p_b0 + 1
    inv_p_inversion, inv_p_b = np.argsort(p_inv0) + 1, np.argsort(p_b0) + 1
    ext = Ext()
    fe = ext.hip_fe(_hip_fe_inputs(S, inv_p_inversion, inv_p_b))
    # eddy_state(model, A): params.f and eddy_param.f are the same function -> served; "f" table; M = upload_csc(A[p, p], 0)
    fq = np.ascontiguousarray(fo._const_or_fn(o.f, o.geo.xq), dtype=np.float64)
    L.check(lib().npg_fe_set_coeff(fe, b"f", P64(fq)))
    A_native = sp.csc_matrix(S.A)                      # what on_architecture(CPU(), solver.A)[iperm, iperm] gives run!: zeros kept
    M = ext.upload_csc0(A_native[p_inv0][:, p_inv0])
    N2min, smoothing, nu_min = 0.3, 10.0, 1.0          # ν_eddy(eddy_param, αbz; smoothing = 10, ν_min = 1)
    a2e2 = o.alpha ** 2 * o.eps ** 2
    rel = lambda a, c: np.linalg.norm(a - c) / np.linalg.norm(c)
    xq = o.geo.xq
    for k, bfn in enumerate((lambda x: 0.3 * x[..., 2] ** 2 + 0.1 * x[..., 0] * x[..., 2], lambda x: -0.8 * x[..., 2] + 0.2 * np.sin(3 * x[..., 1]))):
        b = o.interpolate_b(bfn)                       # model.state.b.free_values, native order
        bdev = ext.upload_perm(b, p_b)
        L.check(lib().npg_fe_update_nu_eddy(fe, N2min, o.alpha, o.N2, smoothing, nu_min, bdev))
        L.check(lib().npg_fe_assemble_matrix(fe, L.NPG_MAT_A, a2e2, 1, M, None))
        cp1, rv1, nz, m, n = ext.on_architecture_CPU_HIPSparseMatrixCSR(M)
        got = sp.csc_matrix((nz, rv1 - 1, cp1 - 1), shape=(m, n))
        # oracle: nu_eddy at the quadrature points (recipe.run's eddy branch), full-stress A
        bn = o.b_nodal(b)[o.cnb]
        abz = o.alpha * (o.N2 + np.einsum("cqi,ci->cq", o.gradNb[..., 2], bn))
        nu_e = fq * (fq / np.sqrt(N2min ** 2 + abz ** 2))
        nuq = np.logaddexp(smoothing * nu_min, smoothing * nu_e) / smoothing
        want = sp.csc_matrix(o.A_inversion(nu_q=nuq))[p_inv0][:, p_inv0]
        assert len(nz) == A_native.nnz, "the refresh must keep the pattern run! holds"
        assert abs(got - want).max() <= 1e-12 * abs(want).max(), (k, abs(got - want).max(), abs(want).max())
    # `solver.A = on_architecture(arch, A_inversion[perm, perm])` is answered with M (PENDING_A): the next invert! runs on it
    h, _ = o.precond_h()
    inv = ext.InversionToolkit(M, ext.on_architecture_GPU_Array(np.full(N, 1 / h ** 3)), None, None, N, atol=1e-9, rtol=1e-9)
    tk = inv["solver"]
    xs = np.cos(np.arange(N, dtype=float))
    rhs = np.asarray(want @ xs)
    L.check(lib().npg_vec_upload(tk["y"], P64(np.ascontiguousarray(rhs))))
    ext.iterative_solve(tk)
    st = tk["workspace"]["stats"]
    x = ext.on_architecture_CPU_HIPVector(tk["x"])
    assert st["solved"] == 1 and np.linalg.norm(rhs - want @ x) / h ** 3 <= 2 * (1e-9 + 1e-9 * st["rnorm0"])
