"""Device boundary: mirrors /root/reference/src/architectures.jl:4-20 and supersedes ext/nuPGCMCUDAExt.jl:24-33.

    on_architecture(GPU(), ndarray)          -> DeviceVector        (CuArray(a))
    on_architecture(CPU(), DeviceVector)     -> ndarray             (Array(a))
    on_architecture(GPU(), scipy sparse)     -> DeviceCSR           (CuSparseMatrixCSR(a); CSC -> CSR on upload)
    on_architecture(CPU(), DeviceCSR)        -> scipy.sparse.csc_matrix
    architecture(x), vector_type(arch, T), print_memory_status(arch)

The device objects are thin owners of the opaque handles of libnupgcm_hip.so; there is no host implementation behind
`GPU()` - without the library or without a gfx950 device these calls raise."""
from __future__ import annotations

import ctypes as C
import os
import resource

import numpy as np
import scipy.sparse as sp

from . import _lib as L


class AbstractArchitecture:
    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self).__name__)

    def __repr__(self):
        return f"{type(self).__name__}()"


class CPU(AbstractArchitecture):
    """The reference's CPU() architecture (src/architectures.jl:5).  As a marker it keeps its round-1 meaning -
    on_architecture(CPU(), device_object) brings data to the host.  As the architecture of a MODEL (BASELINE configs[0]: the
    reference's test configuration on CPU(), "runs without a GPU") its context lives in libnupgcm_host.so, the host build of the
    same C ABI (csrc_host/): Model(CPU(), ...) assembles with the host element kernels, multiplies with the host SpMV and solves
    as the reference's CPU() path does - sparse LU wherever it factorises (src/inversion.jl:55-58, src/evolution.jl:150-153,
    src/iterative_solvers.jl:42-55), host Krylov otherwise (:58).  One process runs on one architecture: asking for a CPU() context
    while GPU() handles are alive (or the other way round) is refused."""
    device = -1

    @property
    def ctx(self):
        if any(d >= 0 for d in _contexts):
            raise RuntimeError("CPU(): this process already holds GPU() contexts; one process runs on one architecture")
        L.select("host")
        return context(-1)


class GPU(AbstractArchitecture):
    """The MI355X architecture.  All GPU() instances of a process share one context per device (the way CUDA.jl has one
    implicit context); the device defaults to LOCAL_RANK so that one-process-per-GPU launches need no extra plumbing."""

    def __init__(self, device=None):
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else int(device)

    @property
    def ctx(self):
        if -1 in _contexts:
            raise RuntimeError("GPU(): this process already holds a CPU() context; one process runs on one architecture")
        L.select("hip")
        return context(self.device)


class Context:
    def __init__(self, device):
        h = C.c_void_p()
        L.check(L.lib().npg_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = device
        self.rank, self.nranks = 0, 1

    def sync(self):
        L.check(L.lib().npg_ctx_sync(self.h))

    def name(self):
        buf = C.create_string_buffer(256)
        L.check(L.lib().npg_device_name(self.h, buf, 256))
        return buf.value.decode()

    def mem_status(self):
        f, t = C.c_size_t(), C.c_size_t()
        L.check(L.lib().npg_mem_status(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def stream(self):
        return L.lib().npg_ctx_stream(self.h)

    def timer_start(self):
        L.check(L.lib().npg_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        L.check(L.lib().npg_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(unique_id, 128)
        L.check(L.lib().npg_comm_init(self.h, buf, rank, nranks))
        self.rank, self.nranks = rank, nranks

    def comm_info(self):
        """dict describing the communicator: rank, nranks, rccl_ranks (ncclCommCount), device, in_cycle_transport"""
        import json
        buf = C.create_string_buffer(1024)
        L.check(L.lib().npg_comm_info(self.h, buf, 1024))
        return json.loads(buf.value.decode())

    def disable_peer(self):
        """auto transport: drop the peer windows; RCCL carries the in-cycle traffic from here on (npg_comm_disable_peer)"""
        L.check(L.lib().npg_comm_disable_peer(self.h))

    def allreduce_sum(self, values):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        L.check(L.lib().npg_comm_allreduce_sum(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size))
        return a


_contexts: dict = {}


def context(device=0) -> Context:
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    L.check(L.lib().npg_comm_unique_id(buf))
    return buf.raw


class DeviceVector:
    """fp64 vector in HBM (the reference's CuVector{Float64})."""

    def __init__(self, ctx: Context, n: int):
        self.ctx = ctx
        h = C.c_void_p()
        L.check(L.lib().npg_vec_create(ctx.h, int(n), C.byref(h)))
        self.h = h
        self.n = int(n)

    @classmethod
    def from_host(cls, ctx, a, perm=None):
        a = L.as_f64(a)
        v = cls(ctx, len(a) if perm is None else len(perm))
        v.upload(a, perm)
        return v

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_vec_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def view(self, offset, n):
        """non-owning window of this vector (keeps the parent alive)"""
        v = DeviceVector.__new__(DeviceVector)
        h = C.c_void_p()
        L.check(L.lib().npg_vec_view(self.h, int(offset), int(n), C.byref(h)))
        v.ctx, v.h, v.n, v._parent = self.ctx, h, int(n), self
        return v

    def __len__(self):
        return self.n

    @property
    def shape(self):
        return (self.n,)

    dtype = np.dtype(np.float64)

    def upload(self, a, perm=None):
        a = L.as_f64(a)
        if perm is None:
            if a.size != self.n:
                raise ValueError(f"upload: vector has {self.n} entries, got {a.size}")
            L.check(L.lib().npg_vec_upload(self.h, L.ptr(a)))
        else:
            p = L.as_i64(perm)
            if p.size != self.n or (p.size and (p.min() < 0 or p.max() >= a.size)):
                raise ValueError("upload: permutation does not match")
            L.check(L.lib().npg_vec_upload_perm(self.h, L.ptr(a), L.ptr(p)))
        return self

    def to_host(self, perm=None):
        out = np.empty(self.n)
        if perm is None:
            L.check(L.lib().npg_vec_download(self.h, L.ptr(out)))
        else:
            p = L.as_i64(perm)
            if p.size != self.n:
                raise ValueError("to_host: permutation does not match")
            L.check(L.lib().npg_vec_download_perm(self.h, L.ptr(out), L.ptr(p)))
        return out

    def __getitem__(self, perm):
        """`x[inv_perm]` of src/model.jl:282,312: a gathered copy (returned on the host, where the reference sends it
        next anyway)."""
        return self.to_host(np.asarray(perm))

    def fill(self, a):
        L.check(L.lib().npg_vec_fill(self.h, float(a)))
        return self

    def copy_from(self, other: "DeviceVector"):
        L.check(L.lib().npg_vec_copy(self.h, other.h))
        return self

    def copy(self):
        return DeviceVector(self.ctx, self.n).copy_from(self)

    def axpby(self, a, x: "DeviceVector", b):
        """self = a x + b self"""
        L.check(L.lib().npg_vec_axpby(self.h, float(a), x.h, float(b)))
        return self

    def lincomb(self, coefs, xs):
        """self = sum_k coefs[k] xs[k]  (one fused kernel; src/inversion.jl:104, src/model.jl:278)"""
        n = len(xs)
        cf = (C.c_double * n)(*[float(c) for c in coefs])
        hs = (C.c_void_p * n)(*[x.h for x in xs])
        L.check(L.lib().npg_vec_lincomb(self.h, n, cf, hs))
        return self

    def mul(self, d: "DeviceVector", x: "DeviceVector"):
        L.check(L.lib().npg_vec_mul(self.h, d.h, x.h))
        return self

    def dot(self, other):
        out = C.c_double()
        L.check(L.lib().npg_vec_dot(self.h, other.h, C.byref(out)))
        return out.value

    def norm(self):
        out = C.c_double()
        L.check(L.lib().npg_vec_nrm2(self.h, C.byref(out)))
        return out.value

    def maxabs(self):
        out, nan = C.c_double(), C.c_int()
        L.check(L.lib().npg_vec_maxabs(self.h, C.byref(out), C.byref(nan)))
        return out.value, bool(nan.value)

    def __repr__(self):
        return f"{self.n}-element DeviceVector{{Float64}}"


class DeviceIndex:
    """device-resident int64 index array (npg_index): entries are checked against `bound` when it is created"""

    def __init__(self, ctx, host, bound):
        a = L.as_i64(host)
        h = C.c_void_p()
        L.check(L.lib().npg_index_create(ctx.h, a.size, L.ptr(a), int(bound), C.byref(h)))
        self.h, self.ctx, self.n = h, ctx, a.size

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_index_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DeviceILU0:
    """ILU(0) factors of a plain-CSR DeviceCSR in HBM - KrylovPreconditioners.kp_ilu0(P) of the reference's GPU P-block
    (src/preconditioners.jl:101-107): level-scheduled factorisation and triangular solves (csrc/ilu.hip)."""

    def __init__(self, A: "DeviceCSR"):
        h = C.c_void_p()
        L.check(L.lib().npg_ilu0_create(A.ctx.h, A.h, C.byref(h)))
        self.h, self.ctx, self.A = h, A.ctx, A
        lo, up, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().npg_ilu0_info(self.h, C.byref(lo), C.byref(up), C.byref(nnz)))
        self.levels, self.nnz, self.stats = (lo.value, up.value), nnz.value, None

    def refactor(self, A: "DeviceCSR" = None):
        """A's values changed (same pattern): factorise again"""
        L.check(L.lib().npg_ilu0_refactor(self.h, (A or self.A).h))
        return self

    def ldiv(self, r: DeviceVector, z: DeviceVector = None):
        """z = U^-1 L^-1 r  (ldiv!(z, P_prec, r))"""
        z = z or DeviceVector(self.ctx, r.n)
        L.check(L.lib().npg_ilu0_apply(self.h, r.h, z.h))
        return z

    def factors(self):
        """the factors' values in A's pattern as a scipy CSR (strictly lower: L without its unit diagonal, rest: U)"""
        import scipy.sparse as sp
        M = self.A.to_scipy_csr()
        M.sort_indices()
        v = np.empty(self.nnz)
        L.check(L.lib().npg_ilu0_factors(self.h, L.ptr(v)))
        return sp.csr_matrix((v, M.indices, M.indptr), shape=M.shape)

    def cg(self, A: "DeviceCSR", b: DeviceVector, x: DeviceVector, atol=1e-6, rtol=1e-6, itmax=0):
        """cg(A, b, x; M = self, ldiv = true), x in/out (warm start) - npg_cg_ilu0_solve"""
        st = L.SolveStats()
        L.check(L.lib().npg_cg_ilu0_solve(self.h, A.h, b.h, x.h, float(atol), float(rtol), int(itmax), C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_ilu0_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DeviceCSR:
    """fp64 CSR matrix with int32 column indices in HBM (the reference's CuSparseMatrixCSR{Float64,Int32})."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.h = handle
        m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().npg_csr_shape(self.h, C.byref(m), C.byref(n), C.byref(nnz)))
        self.shape = (m.value, n.value)
        self.nnz = nnz.value
        self._parent = None

    dtype = np.dtype(np.float64)

    @classmethod
    def from_scipy(cls, ctx, A, drop_zeros=False):
        A = sp.csc_matrix(A)
        A.sort_indices()
        h = C.c_void_p()
        cp, ri, nz = L.as_i64(A.indptr), L.as_i64(A.indices), L.as_f64(A.data)
        L.check(L.lib().npg_csr_create_from_csc(ctx.h, A.shape[0], A.shape[1], L.ptr(cp), L.ptr(ri), L.ptr(nz),
                                                int(bool(drop_zeros)), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_pattern(cls, ctx, m, n, rowptr, colind, val=None):
        rp, ci = L.as_i64(rowptr), L.as_i32(colind)
        h = C.c_void_p()
        v = None if val is None else L.as_f64(val)
        L.check(L.lib().npg_csr_create(ctx.h, int(m), int(n), L.ptr(rp), L.ptr(ci), None if v is None else L.ptr(v),
                                       C.byref(h)))
        return cls(ctx, h)

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_csr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def clone(self):
        h = C.c_void_p()
        L.check(L.lib().npg_csr_clone(self.h, C.byref(h)))
        B = DeviceCSR(self.ctx, h)
        B._parent = self           # the clone shares this matrix's pattern arrays
        return B

    def to_scipy_csc(self):
        m, n = self.shape
        cp, ri, nz = np.empty(n + 1, np.int64), np.empty(self.nnz, np.int64), np.empty(self.nnz)
        L.check(L.lib().npg_csr_to_csc(self.h, L.ptr(cp), L.ptr(ri), L.ptr(nz)))
        return sp.csc_matrix((nz, ri, cp), shape=(m, n))

    def to_scipy_csr(self):
        m, n = self.shape
        rp, ci, v = np.empty(m + 1, np.int64), np.empty(self.nnz, np.int32), np.empty(self.nnz)
        L.check(L.lib().npg_csr_download(self.h, L.ptr(rp), L.ptr(ci), L.ptr(v)))
        return sp.csr_matrix((v, ci, rp), shape=(m, n))

    def block_nodes(self, n_full, n_surf, rtol=1e-12):
        """store the velocity block node by node: one {c, K, C} record per coupled node pair (npg_csr_block_nodes); returns
        True if the structure held and the matrix is now node-blocked"""
        flag = C.c_int()
        L.check(L.lib().npg_csr_block_nodes(self.h, int(n_full), int(n_surf), float(rtol), C.byref(flag)))
        self.paired = bool(flag.value)
        return self.paired

    def set_ghost_nodes(self, first_col, ncomp):
        """a rank's row block: ghost columns [first_col[g], first_col[g] + ncomp[g]) are the components of ONE velocity node
        (npg_csr_set_ghost_nodes; before block_nodes) - owned-ghost node couplings then become node records in the windowed tiles"""
        fc = np.ascontiguousarray(first_col, dtype=np.int32)
        nc = np.ascontiguousarray(ncomp, dtype=np.int32)
        L.check(L.lib().npg_csr_set_ghost_nodes(self.h, len(fc), fc.ctypes.data_as(C.c_void_p), nc.ctypes.data_as(C.c_void_p)))
        return self

    def pack_nodes(self, n_full, n_surf):
        """attach a record-form companion with FULL node records (function-valued viscosity; npg_csr_pack_nodes): the matrix
        stays plain for assembly / download, products and solves read the companion, which follows every re-assembly"""
        flag = C.c_int()
        L.check(L.lib().npg_csr_pack_nodes(self.h, int(n_full), int(n_surf), C.byref(flag)))
        self.packed = bool(flag.value)
        return self.packed

    def block_nodes_dofs(self, node_of_dof, comp_of_dof, rtol=1e-12):
        """npg_csr_block_nodes_dofs: node-block storage of a matrix in ANY DoF order (node_of_dof[i] < 0: not a velocity DoF);
        the library renumbers internally and keeps the permutation in the handle - mul and GmresWorkspace.solve keep taking
        vectors in this matrix's own order."""
        nd = np.ascontiguousarray(node_of_dof, dtype=np.int64)
        cd = np.ascontiguousarray(comp_of_dof, dtype=np.int32)
        if nd.shape != (self.shape[0],) or cd.shape != nd.shape:
            raise ValueError("block_nodes_dofs: one node label and one component per row")
        flag = C.c_int()
        L.check(L.lib().npg_csr_block_nodes_dofs(self.h, nd.ctypes.data_as(C.c_void_p), cd.ctypes.data_as(C.c_void_p), float(rtol),
                                                 C.byref(flag)))
        self.paired = bool(flag.value)
        return self.paired

    def pair_xy(self, npairs, rtol=1e-12):
        """two-component special case of block_nodes: rows 2q, 2q+1 for q < npairs"""
        return self.block_nodes(0, npairs, rtol)

    def storage(self):
        """(block nodes, 20-byte {c, K, C} records, 12-byte CSR entries) as laid out in HBM"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().npg_csr_storage(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def coupling_records(self):
        """28-byte records: {c, d_x, d_y, d_z} of the rows behind the block rows + {m, a_x, a_y, a_z} column records of the
        block rows (0: all of that is plain CSR)"""
        a = C.c_int64()
        L.check(L.lib().npg_csr_coupling_records(self.h, C.byref(a)))
        return a.value

    def stored_spmv_bytes(self):
        """bytes one SpMV streams from HBM with this layout (matrix arrays in the form the kernels read + x once + y once)"""
        mb = C.c_int64()
        L.check(L.lib().npg_csr_spmv_bytes(self.h, C.byref(mb)))
        m, n = self.shape
        return mb.value + 8 * n + 8 * m

    def set_lanes(self, lanes):
        """lanes per row in the tiled SpMV's segmented sums: 4, 8, 16, 32 (0: the default rule) - npg_csr_set_lanes"""
        L.check(L.lib().npg_csr_set_lanes(self.h, int(lanes)))
        return self

    def window_info(self):
        """the windowed tile set of a node-blocked matrix (csrc/spmv_window.h): dict(tiles, block_tiles, distinct, bytes) -
        bytes = what one product of the Krylov kernels' gather-layout instance streams from HBM (matrix + x + y); tiles = 0
        without such a set"""
        a, b, c, d = (C.c_int64() for _ in range(4))
        L.check(L.lib().npg_csr_window_info(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        m, n = self.shape
        return dict(tiles=a.value, block_tiles=b.value, distinct=c.value, bytes=d.value + (4 * c.value + 8 * m if a.value else 0))

    def mul_gather32(self, x: DeviceVector, y: DeviceVector = None, windowed=True, reps=1):
        """A * float32(x) in fp64 arithmetic: the product of the Krylov kernels' gather-layout instance (npg_spmv_gather32),
        on the windowed tile set or on the ordinary tiles"""
        if y is None:
            y = DeviceVector(self.ctx, self.shape[0])
        L.check(L.lib().npg_spmv_gather32(self.h, x.h, y.h, int(bool(windowed)), int(reps)))
        return y

    def mul(self, x: DeviceVector, y: DeviceVector = None, alpha=1.0, beta=0.0):
        """mul!(y, A, x) / A*x"""
        if y is None:
            y = DeviceVector(self.ctx, self.shape[0])
        L.check(L.lib().npg_spmv(self.h, x.h, y.h, float(alpha), float(beta)))
        return y

    def __matmul__(self, x):
        return self.mul(x)

    def combine(self, a, X, b, Y, Z):
        """self = a X + b (Y + Z) on a shared pattern"""
        L.check(L.lib().npg_csr_combine(self.h, float(a), X.h, float(b), Y.h, Z.h))
        return self

    def gather_values(self, src, index):
        """self.val[k] = src.val[index[k]] (npg_csr_gather_values): this matrix holds a fixed subset of src's entries"""
        L.check(L.lib().npg_csr_gather_values(self.h, src.h, index.h))
        return self

    def inv_diag(self, out: DeviceVector = None):
        if out is None:
            out = DeviceVector(self.ctx, self.shape[0])
        L.check(L.lib().npg_csr_inv_diag(self.h, out.h))
        return out

    def __repr__(self):
        return f"{self.shape[0]}x{self.shape[1]} DeviceCSR{{Float64,Int32}} with {self.nnz} stored entries"


def on_architecture(arch, a, **kw):
    if isinstance(arch, CPU):
        if isinstance(a, DeviceVector):
            return a.to_host()
        if isinstance(a, DeviceCSR):
            return a.to_scipy_csc()
        return a
    if isinstance(arch, GPU):
        if isinstance(a, (DeviceVector, DeviceCSR)):
            return a
        if sp.issparse(a):
            return DeviceCSR.from_scipy(arch.ctx, a, **kw)
        return DeviceVector.from_host(arch.ctx, np.asarray(a, dtype=np.float64))
    raise TypeError(f"unknown architecture {arch!r}")


def architecture(a):
    if isinstance(a, (DeviceVector, DeviceCSR)):
        return GPU(a.ctx.device) if a.ctx.device >= 0 else CPU()
    return CPU()


def vector_type(arch, T=np.float64):
    if isinstance(arch, GPU):
        return lambda n: DeviceVector(arch.ctx, n)
    return lambda n: np.zeros(n, dtype=T)


def print_memory_status(arch):
    if isinstance(arch, GPU):
        f, t = arch.ctx.mem_status()
        print(f"GPU memory usage: {(t - f) / 2**30:.3f} GiB / {t / 2**30:.3f} GiB ({arch.ctx.name()})")
    else:
        print(f"CPU memory usage: {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.3f} GB")
