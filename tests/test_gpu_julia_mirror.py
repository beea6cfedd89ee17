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
        L.check(lib().npg_csr_create_from_csc(self.ctx, m, n, P64(cp), P64(rv), P64(nz), 1, C.byref(out)))
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

    def evolve(self, model, u_prev, b_prev):                                 # evolve!(model::HIPModel, u_prev, b_prev)
        dofs, ev = model["dofs"], model["evolution"]
        solver = ev["solver"]
        fe = model["fe"]
        xi = lambda u: self.upload_perm(np.concatenate([u, np.zeros(dofs["np"])]), dofs["p_inversion"])
        bv = lambda b: self.upload_perm(b, dofs["p_b"])
        L.check(lib().npg_fe_evolution_rhs(fe, 2, model["dt"], model["N2"], model["theta"], bv(model["b"]), bv(b_prev),
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
    # round trip of on_architecture(CPU(), A): values survive, explicit zeros are dropped
    cp1, rv1, nz, m, n = ext.on_architecture_CPU_HIPSparseMatrixCSR(A)
    back = sp.csc_matrix((nz, rv1 - 1, cp1 - 1), shape=(m, n))
    assert abs(back - S.A[p_inv0][:, p_inv0]).max() == 0 and back.nnz < S.A.nnz
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
