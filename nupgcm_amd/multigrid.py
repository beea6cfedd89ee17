"""General (inexact) preconditioners for the inversion and the flexible GMRES workspace that drives them.

`BlockDiagonalPreconditioner` mirrors /root/reference/src/preconditioners.jl:53-125 (an inner Jacobi-CG per block:
friction-only velocity block, pressure mass matrix / (alpha^2 eps^2)).  `MultigridPreconditioner` is new work (SURVEY 8f rank
1): a geometric multigrid V-cycle on the whole saddle-point system over the red-refinement hierarchy the large bowl meshes
are built from (nupgcm_amd.refine), with a node-block Braess-Sarazin smoother - see csrc/mg.hip.  Both are applied by
libnupgcm_hip.so; this module only prepares their operators at set-up time:

  * per level: FEData (own RCM / node-block ordering), the level's A assembled by the device kernels, and the FIXED patterns
    (host index work on the global sparsity pattern) of its blocks G = A[u, p], D = A[p, u], of the inverse Dinv of the
    node-block diagonal of A[u, u] and of S = D Dinv G - the values are computed on the device (gathers, node-block inverse,
    fixed-pattern triple product), at set-up and again whenever the eddy closure re-assembles A,
  * between levels: P2 (velocity) and P1 (pressure) nodal interpolation from the parent cell of every fine node, composed with
    both levels' device orderings; the pressure of every level is pinned at its own last vertex (src/dofs.jl:57), so the
    interpolated coarse pressure is shifted by its value there (constants are in the null space of the gradient block).
"""
from __future__ import annotations

import ctypes as C

import os

import numpy as np
import scipy.sparse as sp

from . import _lib as L
from . import refine
from .architectures import DeviceCSR, DeviceVector
from .fe import _TET_EDGE_A, _TET_EDGE_B, FEData, Mesh
from .inversion import build_A_inversion, device_fe


class GeneralPreconditioner:
    """an npg_precond handle: an inexact operator application, usable by FgmresWorkspace only"""

    def __init__(self, ctx, kind, nparts):
        h = C.c_void_p()
        L.check(L.lib().npg_precond_create(ctx.h, int(kind), int(nparts), C.byref(h)))
        self.h, self.ctx = h, ctx
        self._keep = []            # device objects the library borrows

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_precond_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def apply(self, r: DeviceVector, z: DeviceVector):
        L.check(L.lib().npg_precond_apply(self.h, r.h, z.h))
        return z

    def counters(self):
        a, b = C.c_int64(), C.c_int64()
        L.check(L.lib().npg_precond_counters(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def cycle_bytes(self):
        """bytes one application streams as its operators are laid out (npg_precond_cycle_bytes; 0 before the first application)"""
        a = C.c_int64()
        L.check(L.lib().npg_precond_cycle_bytes(self.h, C.byref(a)))
        return a.value


# ---- reference's block-diagonal preconditioner -----------------------------------------------------------------------------
def pressure_mass_matrix(fe_data: FEData):
    """int p q over the P1 pressure space in device (p_p) order - `assemble_matrix(a, P_trial, P_test)[p_p, p_p]`
    (src/preconditioners.jl:83-85); P1 on a tetrahedron: |K| (1 + delta_ij) / 20."""
    m, t, d = fe_data.mesh, fe_data.tables, fe_data.dofs
    vol = m.detJ / 6.0
    loc = vol[:, None, None] * (1.0 + np.eye(4)) / 20.0
    pos = t.p_pos[m.cells]                                           # (nc, 4) device index in [u; p] or -1
    rows = np.broadcast_to(pos[:, :, None], loc.shape)
    cols = np.broadcast_to(pos[:, None, :], loc.shape)
    ok = (rows >= 0) & (cols >= 0)
    return sp.csr_matrix((loc[ok], (rows[ok] - d.nu, cols[ok] - d.nu)), shape=(d.np, d.np))


class BlockDiagonalPreconditioner(GeneralPreconditioner):
    """BlockDiagonalPreconditioner(arch, params, fe_data, A_inversion) - src/preconditioners.jl:53-93: velocity block = CG on
    the friction-only A[1:nu, 1:nu] assembled with nu = 1 (:74-80), pressure block = CG on the pressure mass matrix /
    (alpha^2 eps^2) with its diagonal (:83-88, itmax = 0).  u_precond: what preconditions the velocity block's CG - "ilu0", the
    reference's GPU recipe (P_block_setup(::GPU): kp_ilu0, ldiv = true, itmax = 100, :101-107; DeviceILU0), or "jacobi", the
    variant its log lists as `BlockDiagonal(I/h^3)` / the commented-out Diagonal(1 ./ diag(A)) (:109-115) - the default here,
    because a level-scheduled triangular solve costs one dependent launch per level (thousands on a P2 friction block)."""

    def __init__(self, arch, params, fe_data, A_inversion=None, u_itmax=100, p_itmax=0, atol=1e-6, rtol=1e-6, u_precond="jacobi"):
        super().__init__(arch.ctx, L.NPG_PC_BLOCKDIAG, 2)
        d, ctx = fe_data.dofs, arch.ctx
        # a PRIVATE assembly engine: the friction-only matrix needs nu = 1, f = 0, and the engine cached per FEData is
        # shared with the solver (a later build_A_inversion(..., nu=None) keeps whatever table the device holds)
        from .assembly import DeviceFE
        fe = DeviceFE(ctx, fe_data)
        fe.set_coeff("nu", 1.0)
        fe.set_coeff("f", 0.0)                                       # friction_only = true
        Afr = fe.assemble(L.NPG_MAT_A, fe.new_matrix("A"), scale=params.alpha ** 2 * params.eps ** 2).to_scipy_csr()
        del fe
        F = sp.csr_matrix(Afr[:d.nu, :d.nu])
        F.eliminate_zeros()                                          # dropzeros!(A)  (:80)
        T = pressure_mass_matrix(fe_data) / (params.alpha ** 2 * params.eps ** 2)
        for k, (off, M) in enumerate(((0, F), (d.nu, T))):
            Ad = DeviceCSR.from_scipy(ctx, M)
            jac = DeviceVector.from_host(ctx, 1.0 / M.diagonal())
            self._keep += [Ad, jac]
            L.check(L.lib().npg_precond_blockdiag_set(self.h, k, int(off), Ad.h, jac.h, int(u_itmax if k == 0 else p_itmax),
                                                      float(atol), float(rtol)))
            if k == 0 and u_precond == "ilu0":
                from .architectures import DeviceILU0
                self.ilu = DeviceILU0(Ad)
                self._keep.append(self.ilu)
                L.check(L.lib().npg_precond_blockdiag_set_ilu0(self.h, 0, self.ilu.h))
            elif k == 0 and u_precond != "jacobi":
                raise ValueError(f"BlockDiagonalPreconditioner: u_precond = {u_precond!r} (\"jacobi\" or \"ilu0\")")
        self.u_precond = u_precond


class DenseInversePreconditioner(GeneralPreconditioner):
    """P = A^-1 held explicitly in HBM (n^2 doubles) - npg_precond_dense_set.  For the reference's own small meshes (16 k /
    31 k unknowns: 2 / 8 GB), where the Krylov path is bound by kernel latency; the device counterpart of the CPU() path's
    `lu(A)` + `ldiv!` (src/inversion.jl:55-58).  `refresh(A)` follows a re-assembled matrix (eddy closure)."""

    def __init__(self, arch, A: DeviceCSR, storage="fp64"):
        super().__init__(arch.ctx, L.NPG_PC_DENSE, 1)
        self.storage = storage
        self.refresh(A)

    def refresh(self, A: DeviceCSR, model=None):
        L.check(L.lib().npg_precond_dense_set(self.h, A.h, int(self.storage == "fp32")))
        self.n = A.shape[0]
        return self

    def __repr__(self):
        b = 4 if self.storage == "fp32" else 8
        return f"DenseInversePreconditioner(n={self.n}, {self.storage} storage, {b * self.n ** 2 / 2 ** 30:.1f} GiB)"


# ---- multigrid -----------------------------------------------------------------------------------------------------------------
def _p2_shape(lam):
    return np.concatenate([lam * (2 * lam - 1), 4 * lam[..., _TET_EDGE_A] * lam[..., _TET_EDGE_B]], axis=-1)


def nodal_interpolation(mesh_c: Mesh, mesh_f: Mesh):
    """(P2: nn_f x nn_c, P1: nv_f x nv_c) interpolation matrices between a mesh and its red refinement (refine.refine_once:
    fine cell 8 i + k is child k of coarse cell i).  Every fine P2 node is evaluated in its parent cell with the parent's P2
    (P1) shape functions at the node's barycentric coordinates there (vertices: multiples of 1/2, edge nodes: of 1/4)."""
    ncf = mesh_f.ncell
    if ncf != 8 * mesh_c.ncell:
        raise ValueError("nodal_interpolation: the fine mesh is not a uniform red refinement of the coarse one")
    par, kid = np.arange(ncf) // 8, np.arange(ncf) % 8
    variant = getattr(mesh_f.model, "child_variant", None)
    if variant is None or len(variant) != mesh_c.ncell:
        raise ValueError("nodal_interpolation: the fine mesh does not carry refine_once's child_variant of this coarse mesh")
    lam = refine.CHILD_BARY[np.asarray(variant, dtype=np.int64)[par], kid]   # (ncf, 4 fine verts, 4 coarse verts), model order
    fm = mesh_f.vertex_of[np.asarray(mesh_f.model.cells, dtype=np.int64)]
    lam = np.take_along_axis(lam, np.argsort(fm, axis=1, kind="stable")[:, :, None], axis=1)
    cm = mesh_c.vertex_of[np.asarray(mesh_c.model.cells, dtype=np.int64)][par]
    lam = np.take_along_axis(lam, np.argsort(cm, axis=1, kind="stable")[:, None, :], axis=2)
    lam10 = np.concatenate([lam, 0.5 * (lam[:, _TET_EDGE_A, :] + lam[:, _TET_EDGE_B, :])], axis=1)

    def build(vals, rows, cols, shape):
        rows, cols = np.broadcast_to(rows, vals.shape).ravel(), np.broadcast_to(cols, vals.shape).ravel()
        key, first = np.unique(rows * shape[1] + cols, return_index=True)       # a conforming field: any cell's value
        M = sp.csr_matrix((vals.ravel()[first], (key // shape[1], key % shape[1])), shape=shape)
        M.eliminate_zeros()
        return M

    P2 = build(_p2_shape(lam10), mesh_f.cell_nodes[:, :, None], mesh_c.cell_nodes[par][:, None, :], (mesh_f.nn, mesh_c.nn))
    P1 = build(lam, mesh_f.cells[:, :, None], mesh_c.cells[par][:, None, :], (mesh_f.nv, mesh_c.nv))
    return P2, P1


def prolongation(fed_c: FEData, fed_f: FEData):
    """[u; p] interpolation (n_f x n_c) in the DEVICE orderings of both levels"""
    P2, P1 = nodal_interpolation(fed_c.mesh, fed_f.mesh)
    tc, tf, dc, df = fed_c.tables, fed_f.tables, fed_c.dofs, fed_f.dofs
    nc, nf = dc.nu + dc.np, df.nu + df.np
    P2 = P2.tocoo()
    rows, cols, vals = [], [], []
    for a in range(3):
        r, c = tf.u_pos[P2.row, a], tc.u_pos[P2.col, a]
        ok = (r >= 0) & (c >= 0)
        rows.append(r[ok]); cols.append(c[ok]); vals.append(P2.data[ok])
    # pressure: free coarse vertices -> all fine vertices, minus the value at the fine level's pinned vertex
    free_c = np.nonzero(tc.p_pos >= 0)[0]
    Pn = P1.tocsr()[:, free_c]                                       # nv_f x np_c
    pinned = np.nonzero(tf.p_pos < 0)[0]
    if len(pinned) != 1:
        raise ValueError("prolongation: expected exactly one pinned pressure vertex")
    shift = Pn[pinned[0]]
    Pn = (Pn - sp.csr_matrix(np.ones((Pn.shape[0], 1))) @ shift).tocoo()
    r, c = tf.p_pos[Pn.row], tc.p_pos[free_c][Pn.col]
    ok = r >= 0
    rows.append(r[ok]); cols.append(c[ok]); vals.append(Pn.data[ok])
    P = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nf, nc))
    P.eliminate_zeros()
    P.sort_indices()
    return P


def node_block_inverse(F, n_full, n_surf):
    """inverse of the node-block diagonal of the velocity block in the node-block DoF order of nupgcm_amd.fe:
    3 x 3 blocks for the first n_full nodes, 2 x 2 for the next n_surf, 1 x 1 for the rest"""
    F = sp.csr_matrix(F)
    nu = F.shape[0]
    rows, cols, vals = [], [], []
    for start, count, sz in ((0, n_full, 3), (3 * n_full, n_surf, 2), (3 * n_full + 2 * n_surf, nu - 3 * n_full - 2 * n_surf, 1)):
        if count == 0:
            continue
        st = start + sz * np.arange(count)
        blk = np.empty((count, sz, sz))
        for i in range(sz):
            for j in range(sz):
                blk[:, i, j] = np.asarray(F[st + i, st + j]).ravel()
        inv = np.linalg.inv(blk)
        for i in range(sz):
            for j in range(sz):
                rows.append(st + i); cols.append(st + j); vals.append(inv[:, i, j])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nu, nu))


MAX_LINE_UNKNOWNS = 136          # csrc/csr.hip kMaxLineBlock: one workgroup inverts a block in LDS (148 of the CU's 160 KB)


def line_blocks(fed: FEData, decimals=7, max_unknowns=MAX_LINE_UNKNOWNS):
    """(block_ptr, block_dofs, line_of_dof): the velocity unknowns (positions in the device order) grouped by the (x, y) of their
    nodes - the unknowns of the nodes above one another form one block (a structured-to-tet mesh such as the channel basin's
    stacks its nodes in vertical lines; on an unstructured mesh the blocks fall back to the single nodes).  Blocks are numbered
    by their first unknown; each block's unknowns ascend.  A line with more than `max_unknowns` unknowns (the device inverts a
    block in the LDS of one CU: 136 = 45 three-component nodes, about 22 P2 layers) is cut into consecutive vertical segments of at
    most that many - whole nodes, top down: finer vertical resolutions degrade to shorter blocks instead of failing the set-up
    (ADVICE round 4)."""
    s, d = fed.spaces, fed.dofs
    nu = d.nu
    node_of = np.full(nu, -1, dtype=np.int64)
    for a in range(3):
        nodes = np.nonzero(s.u_dof[:, a] >= 0)[0]
        node_of[d.inv_p_u[s.u_dof[nodes, a]]] = nodes
    xy = np.round(fed.mesh.node_coords[node_of][:, :2], decimals)
    _, grp = np.unique(xy, axis=0, return_inverse=True)
    grp = np.asarray(grp).ravel()
    if max_unknowns and np.bincount(grp).max() > max_unknowns:
        # rank of every unknown's NODE within its line, top down; a segment holds max_unknowns // 3 nodes
        z = fed.mesh.node_coords[node_of][:, 2]
        o = np.lexsort((node_of, -z, grp))
        new_node = np.r_[True, (node_of[o][1:] != node_of[o][:-1]) | (grp[o][1:] != grp[o][:-1])]
        node_rank = np.cumsum(new_node) - 1
        line_start = np.r_[True, grp[o][1:] != grp[o][:-1]]
        first_rank = np.maximum.accumulate(np.where(line_start, node_rank, 0))
        seg = np.empty(nu, dtype=np.int64)
        seg[o] = (node_rank - first_rank) // max(1, max_unknowns // 3)
        long_line = np.bincount(grp)[grp] > max_unknowns
        _, grp = np.unique(np.stack([grp, np.where(long_line, seg, 0)], axis=1), axis=0, return_inverse=True)
        grp = np.asarray(grp).ravel()
    first = np.full(grp.max() + 1, nu, dtype=np.int64)
    np.minimum.at(first, grp, np.arange(nu))
    line_of = np.argsort(np.argsort(first))[grp]                  # blocks renumbered by their first unknown
    order = np.lexsort((np.arange(nu), line_of))
    sizes = np.bincount(line_of)
    bp = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=bp[1:])
    return bp, order.astype(np.int64), line_of.astype(np.int64)


def line_block_inverse(F, block_ptr, block_dofs):
    """host counterpart of npg_csr_line_block_inverse (tests, distributed set-up): inverse of the block diagonal of F with the
    blocks of line_blocks(); blocks of equal size are inverted together"""
    F = sp.csr_matrix(F)
    nu = F.shape[0]
    sizes = np.diff(block_ptr)
    rows, cols, vals = [], [], []
    for n in np.unique(sizes):
        bs = np.nonzero(sizes == n)[0]
        idx = block_dofs[block_ptr[bs][:, None] + np.arange(n)[None, :]]            # (nblk, n)
        blk = np.empty((len(bs), n, n))
        for i in range(n):
            for j in range(n):
                blk[:, i, j] = np.asarray(F[idx[:, i], idx[:, j]]).ravel()
        inv = np.linalg.inv(blk)
        rows.append(np.repeat(idx, n, axis=1).ravel()); cols.append(np.tile(idx, (1, n)).ravel()); vals.append(inv.ravel())
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nu, nu))


def _line_block_pattern(nu, block_ptr, block_dofs, line_of):
    """CSR pattern of the line-block diagonal: row i holds every unknown of its block, ascending"""
    sizes = np.diff(block_ptr)
    size = sizes[line_of]
    rp = np.zeros(nu + 1, dtype=np.int64)
    np.cumsum(size, out=rp[1:])
    col = block_dofs[np.repeat(block_ptr[line_of], size) + (np.arange(rp[-1]) - np.repeat(rp[:-1], size))]
    return rp, col.astype(np.int32)


def _node_block_pattern(nu, n_full, n_surf):
    """CSR pattern (rowptr, col) of the node-block diagonal: 3 x 3 / 2 x 2 / 1 x 1 blocks in the node-block DoF order"""
    r3, r2 = 3 * n_full, 3 * n_full + 2 * n_surf
    size = np.concatenate([np.full(r3, 3), np.full(r2 - r3, 2), np.ones(nu - r2, dtype=np.int64)]).astype(np.int64)
    rp = np.zeros(nu + 1, dtype=np.int64)
    np.cumsum(size, out=rp[1:])
    first = np.concatenate([3 * (np.arange(r3) // 3), r3 + 2 * (np.arange(r2 - r3) // 2), np.arange(r2, nu)])
    col = np.repeat(first, size) + (np.arange(rp[-1]) - np.repeat(rp[:-1], size))
    return rp, col.astype(np.int32)


class _LevelOperators:
    """The smoother's operators of one level on FIXED patterns, (re)computed on the device from the level's plain-CSR A:
    G = A[u, p] and D = A[p, u] by entry gathers, Dinv by the node-block inverse kernel, S = D Dinv G by the fixed-pattern
    triple product.  The host only ever touches index arrays (the global sparsity pattern), at set-up."""

    def __init__(self, ctx, fed, pattern, smoother="node"):
        """smoother: "node" - Dinv inverts the node-block diagonal of the velocity block; "zline" - the blocks are the unknowns of
        the nodes above one another (line_blocks; npg_csr_line_block_inverse): the anisotropic meshes' smoother"""
        from .architectures import DeviceIndex
        d = fed.dofs
        nu, n = d.nu, d.nu + d.np
        rp, ci, shape = pattern
        nnz = len(ci)
        tags = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.float64), ci, rp), shape=shape)
        Gt, Dt = sp.csr_matrix(tags[:nu, nu:]), sp.csr_matrix(tags[nu:, :nu])
        Gt.sort_indices(); Dt.sort_indices()
        self.nu, self.n_full, self.n_surf = nu, d.n_full, d.n_surf
        self.G = DeviceCSR.from_pattern(ctx, nu, n - nu, Gt.indptr, Gt.indices)
        self.D = DeviceCSR.from_pattern(ctx, n - nu, nu, Dt.indptr, Dt.indices)
        self.mapG = DeviceIndex(ctx, np.rint(Gt.data).astype(np.int64) - 1, nnz)
        self.mapD = DeviceIndex(ctx, np.rint(Dt.data).astype(np.int64) - 1, nnz)
        one = lambda M: sp.csr_matrix((np.ones(M.nnz, dtype=np.float32), M.indices, M.indptr), shape=M.shape)
        self.smoother = smoother
        if smoother == "zline":
            bp, bd, line_of = line_blocks(fed)
            self.blocks = (DeviceIndex(ctx, bp, nu + 1), DeviceIndex(ctx, bd, nu))
            self.block_sizes = np.diff(bp)
            irp, icol = _line_block_pattern(nu, bp, bd, line_of)
            self.Dinv = DeviceCSR.from_pattern(ctx, nu, nu, irp, icol)
            # pattern of S = D Dinv G through the lines: pressure node p reaches p' when both touch one line
            nl = len(bp) - 1
            Dl = sp.csr_matrix((np.ones(Dt.nnz, dtype=np.float32), line_of[Dt.indices], Dt.indptr), shape=(n - nu, nl))
            Gc = sp.coo_matrix(Gt)
            Gl = sp.csr_matrix((np.ones(Gc.nnz, dtype=np.float32), (line_of[Gc.row], Gc.col)), shape=(nl, n - nu))
            Sp = sp.csr_matrix(Dl @ Gl)
            self.Gh = None                     # Dinv G would hold a whole line's pressure neighbours per row: not formed
            # index arrays of npg_csr_line_schur (S = D Dinv G line by line): the lines' distinct pressure columns and dense W
            # blocks, and D's entries sorted by (row, line) with their positions inside the lines
            Gl.sort_indices()
            sizes = np.diff(bp)
            woff = np.zeros(nl + 1, dtype=np.int64)
            np.cumsum(sizes * np.diff(Gl.indptr), out=woff[1:])
            pos_in_line = np.empty(nu, dtype=np.int64)
            pos_in_line[bd] = np.arange(nu) - bp[line_of[bd]]
            rows = np.repeat(np.arange(n - nu, dtype=np.int64), np.diff(Dt.indptr))
            lines = line_of[Dt.indices]
            order = np.lexsort((Dt.indices, lines, rows)).astype(np.int64)
            key = rows[order] * nl + lines[order]
            cut = np.concatenate([[0], np.flatnonzero(np.diff(key)) + 1, [len(order)]]).astype(np.int64) if len(order) else np.zeros(1, np.int64)
            seg_line = lines[order][cut[:-1]]
            seg_ptr = np.searchsorted(rows[order][cut[:-1]], np.arange(n - nu + 1)).astype(np.int64)
            self.schur = [DeviceIndex(ctx, Gl.indptr.astype(np.int64), Gl.nnz + 1), DeviceIndex(ctx, Gl.indices.astype(np.int64), n - nu),
                          DeviceIndex(ctx, woff, int(woff[-1]) + 1), DeviceIndex(ctx, order, max(Dt.nnz, 1)),
                          DeviceIndex(ctx, pos_in_line[Dt.indices[order]], int(sizes.max())), DeviceIndex(ctx, seg_ptr, len(seg_line) + 1),
                          DeviceIndex(ctx, seg_line, nl), DeviceIndex(ctx, cut, Dt.nnz + 1)]
        elif smoother == "node":
            irp, icol = _node_block_pattern(nu, d.n_full, d.n_surf)
            self.Dinv = DeviceCSR.from_pattern(ctx, nu, nu, irp, icol)
            Ip = sp.csr_matrix((np.ones(len(icol), dtype=np.float32), icol, irp), shape=(nu, nu))
            Sp = sp.csr_matrix(one(Dt) @ Ip @ one(Gt))
            # Dinv G (the smoother's velocity update with ONE application of Dinv per step: npg_precond_mg_set_scaled_gradient);
            # the components of a node couple to the same pressure nodes, so this is G's pattern wherever the node blocks are full
            Hp = sp.csr_matrix(Ip @ one(Gt))
            Hp.sort_indices()
            self.Gh = DeviceCSR.from_pattern(ctx, nu, n - nu, Hp.indptr, Hp.indices)
        else:
            raise ValueError(f"smoother = {smoother!r} (\"node\" or \"zline\")")
        Sp.sort_indices()
        self.S = DeviceCSR.from_pattern(ctx, n - nu, n - nu, Sp.indptr, Sp.indices)

    def update(self, A: DeviceCSR):
        if os.environ.get("NPG_MG_TIMING") == "1":
            import time
            ctx, t = A.ctx, [time.time()]
            def lap(what):
                ctx.sync()
                t.append(time.time())
                print(f"[npg mg] level of {A.shape[0]} unknowns: {what} {1e3 * (t[-1] - t[-2]):.1f} ms", flush=True)
            self.G.gather_values(A, self.mapG); self.D.gather_values(A, self.mapD); lap("G, D gathered")
            if self.smoother == "zline":
                L.check(L.lib().npg_csr_line_block_inverse(self.Dinv.h, A.h, self.blocks[0].h, self.blocks[1].h))
            else:
                L.check(L.lib().npg_csr_node_block_inverse(self.Dinv.h, A.h, int(self.n_full), int(self.n_surf)))
            lap("Dinv")
            self._schur(); lap("S = D Dinv G")
            if self.Gh is not None:
                L.check(L.lib().npg_csr_product(self.Gh.h, self.Dinv.h, self.G.h)); lap("Dinv G")
            return self
        self.G.gather_values(A, self.mapG)
        self.D.gather_values(A, self.mapD)
        if self.smoother == "zline":
            L.check(L.lib().npg_csr_line_block_inverse(self.Dinv.h, A.h, self.blocks[0].h, self.blocks[1].h))
        else:
            L.check(L.lib().npg_csr_node_block_inverse(self.Dinv.h, A.h, int(self.n_full), int(self.n_surf)))
        self._schur()
        if self.Gh is not None:
            L.check(L.lib().npg_csr_product(self.Gh.h, self.Dinv.h, self.G.h))
        return self

    def _schur(self):
        """S = D Dinv G on S's fixed pattern: line by line for the z-line blocks, the generic triple product for node blocks"""
        if self.smoother == "zline" and os.environ.get("NPG_MG_LINE_SCHUR", "1") != "0":
            L.check(L.lib().npg_csr_line_schur(self.S.h, self.D.h, self.Dinv.h, self.G.h, *[ix.h for ix in self.schur]))
        else:
            L.check(L.lib().npg_csr_triple_product(self.S.h, self.D.h, self.Dinv.h, self.G.h))


def injection(mesh_c: Mesh, mesh_f: Mesh, p1):
    """fine node index of every coarse node (P1: vertices, P2: vertices + edge nodes): the coarse nodes are a subset of the
    fine ones - read off the rows of the nodal interpolation that hold a single unit entry"""
    if p1:
        # vertices: refine_once keeps the coarse nodes' ids, so coarse vertex t (master node g) is fine vertex vertex_of_f[g]
        masters_c = np.full(mesh_c.nv, -1, dtype=np.int64)
        geo = np.arange(len(mesh_c.vertex_of))
        own = np.asarray(getattr(mesh_c.model, "periodic", None) if mesh_c.periodic else geo) == geo
        masters_c[mesh_c.vertex_of[geo[own]]] = geo[own]
        return mesh_f.vertex_of[masters_c]
    P2, _ = nodal_interpolation(mesh_c, mesh_f)
    P = sp.csr_matrix(P2)
    single = np.nonzero((np.diff(P.indptr) == 1))[0]
    rows = single[np.abs(P.data[P.indptr[single]] - 1.0) < 1e-12]
    inj = np.full(P.shape[1], -1, dtype=np.int64)
    inj[P.indices[P.indptr[rows]]] = rows
    if (inj < 0).any():
        raise ValueError("injection: a coarse node has no coinciding fine node")
    return inj


class MultigridPreconditioner(GeneralPreconditioner):
    """MultigridPreconditioner(arch, params, forcings, hierarchy): hierarchy = [FEData coarse, ..., FEData fine], each the
    red refinement of the one before (workloads.bowl_hierarchy_models / channel_basin_hierarchy_models); the last one is the
    model's own fe_data.  `A_fine` is the solver's matrix (it may be stored by node blocks); the coarser operators are
    re-discretised on their own meshes."""

    def __init__(self, arch, params, forcings, hierarchy, A_fine: DeviceCSR = None, omega=2.5, jacobi_weight=0.7,
                 schur_sweeps=3, nu1=2, nu2=2, coarse_sweeps=20, block_nodes=None, cycle="V", coarse_dense=None,
                 mixed=False, scaled_gradient=None, smoother=None, coarse_nu="average"):
        """smoother: "node" (Braess-Sarazin on the node blocks of the velocity block) or "zline" (its blocks are the unknowns of
        the nodes above one another - the anisotropic, structured-in-z meshes: on the channel basin's three-level hierarchy a
        third of the outer iterations; None: NPG_MG_SMOOTHER, else "node").
        scaled_gradient: hand the smoother Dinv G, so that a step applies Dinv once instead of twice (None: on unless
        NPG_MG_SCALED_GRADIENT=0).
        coarse_nu: how the coarser levels follow the eddy closure at a refresh: "average" (the level above's viscosity averaged
        over a cell's children, on the device) or "inject" (rounds 3-4: the closure re-evaluated on the injected buoyancy).
        mixed: the cycle's SpMVs read fp32 copies of the level operators' values (npg_precond_mg_set_mixed); vectors,
        sums and the outer flexible GMRES stay fp64.
        coarse_dense: solve the coarsest level exactly with its dense inverse (DenseInversePreconditioner's machinery)
        instead of `coarse_sweeps` smoothing steps; None = whenever a hierarchy's coarsest level has <= 40 000 unknowns
        (<= 12 GiB; measured at 2.15 M unknowns: 19 instead of 32 iterations for a cold solve, 88 instead of 128 ms).
        A function-valued nu (full-stress form) is re-discretised on every level like a constant one.  With the eddy closure
        on, `refresh(A, model)` (called by run! after each re-assembly of A, src/model.jl:160-170) re-assembles EVERY level
        with the new viscosity (the buoyancy is injected into the coarser meshes) and recomputes the smoothers' operators
        on the device."""
        super().__init__(arch.ctx, L.NPG_PC_MG, len(hierarchy))
        ctx = arch.ctx
        self.arch, self.prm, self.frc, self.hierarchy = arch, params, forcings, hierarchy
        self.levels, self.ops, self.A = [], [], []
        self._top = hierarchy[-1]
        full = callable(forcings.nu) or forcings.eddy_param.is_on
        prev = None
        if scaled_gradient is None:
            scaled_gradient = os.environ.get("NPG_MG_SCALED_GRADIENT", "1") != "0"
        if smoother is None:
            smoother = os.environ.get("NPG_MG_SMOOTHER", "node")
        self.smoother = smoother
        self.scaled_gradient = bool(scaled_gradient) and smoother == "node"
        for lev, fed in enumerate(hierarchy):
            d = fed.dofs
            top = lev == len(hierarchy) - 1
            A = build_A_inversion(arch, fed, params, forcings.nu, structural=full)   # plain CSR, [u; p] device order
            ops = _LevelOperators(ctx, fed, fed.pattern_A(structural=full), smoother=smoother).update(A)
            nu = d.nu
            if top and A_fine is not None:
                A = A_fine
            elif not full and (block_nodes if block_nodes is not None else A.shape[0] >= 100000):
                A.block_nodes(d.n_full, d.n_surf)
            Pd = Rd = None
            if prev is not None:
                P = prolongation(prev, fed)
                Pd, Rd = DeviceCSR.from_scipy(ctx, P), DeviceCSR.from_scipy(ctx, sp.csr_matrix(P.T))
            self._keep += [Pd, Rd]
            self.ops.append(ops)
            self.A.append(A)
            L.check(L.lib().npg_precond_mg_set_level(self.h, lev, A.h, int(nu), ops.G.h, ops.D.h, ops.Dinv.h, ops.S.h,
                                                     None if Pd is None else Pd.h, None if Rd is None else Rd.h))
            if self.scaled_gradient:
                L.check(L.lib().npg_precond_mg_set_scaled_gradient(self.h, lev, ops.Gh.h))
            self.levels.append(dict(n=d.nu + d.np, nu=nu, S_nnz=ops.S.nnz))
            prev = fed
        if os.environ.get("NPG_MG_PARAMS"):          # tuning: "schur_sweeps=6,coarse_sweeps=60,omega=1.8"
            kv = dict(item.split("=") for item in os.environ["NPG_MG_PARAMS"].split(","))
            omega, jacobi_weight = float(kv.get("omega", omega)), float(kv.get("jacobi_weight", jacobi_weight))
            schur_sweeps, nu1, nu2 = int(kv.get("schur_sweeps", schur_sweeps)), int(kv.get("nu1", nu1)), int(kv.get("nu2", nu2))
            coarse_sweeps, cycle = int(kv.get("coarse_sweeps", coarse_sweeps)), kv.get("cycle", cycle)
        self.set_params(omega, jacobi_weight, schur_sweeps, nu1, nu2, coarse_sweeps, cycle)
        if coarse_dense is None:
            limit = min(46340, int(os.environ.get("NPG_MG_COARSE_DENSE_MAX", 40000)))     # (46 340: the library's limit, rocSOLVER's 32-bit offsets)
            coarse_dense = "fp16" if len(hierarchy) > 1 and self.levels[0]["n"] <= limit else False
        # True / "fp32": inverse stored in fp32 (half the bytes per V-cycle; a coarse-grid correction inside a preconditioner
        # needs no more: same 19 iterations, 4.0 instead of 4.7 ms each at 2.15 M unknowns); "fp64": full precision;
        # "fp16" (what None picks): every column scaled by its largest magnitude and stored in fp16 - a quarter of the bytes,
        # the same iteration counts step by step at 2.15 M unknowns, 3.14 instead of 3.50 ms per outer iteration
        # (profiles/r04_multigrid_cycle.txt)
        if coarse_dense and os.environ.get("NPG_MG_COARSE_DENSE"):
            coarse_dense = os.environ["NPG_MG_COARSE_DENSE"]
        self._dense_mode = 0 if not coarse_dense else {"fp64": 1, "fp16": 3}.get(coarse_dense, 2)
        if self._dense_mode:
            L.check(L.lib().npg_precond_mg_set_coarse_dense(self.h, self._dense_mode))
        self.coarse_dense = bool(coarse_dense)
        self.coarse_nu = coarse_nu
        self.mixed = bool(mixed) or os.environ.get("NPG_MG_MIXED", "0") == "1"
        if self.mixed:
            L.check(L.lib().npg_precond_mg_set_mixed(self.h, 1))
        self._inj = None
        if forcings.eddy_param.is_on and len(hierarchy) > 1:      # what refresh() needs, computed at set-up
            p1 = self._top.spaces.b_order == 1
            self._inj = [injection(hierarchy[k].mesh, hierarchy[k + 1].mesh, p1) for k in range(len(hierarchy) - 1)]

    def _update_level(self, lev, A):
        ops = self.ops[lev].update(A)
        L.check(L.lib().npg_precond_mg_update_level(self.h, lev, A.h, ops.G.h, ops.D.h, ops.Dinv.h, ops.S.h))

    def refresh(self, A: DeviceCSR, model=None):
        """the solver's matrix has been re-assembled (eddy closure): recompute the finest level's smoother from it and,
        given the model, re-assemble the coarser levels with the child-volume average of the fine viscosity (coarse_nu="average",
        the default) or with the eddy viscosity of the injected buoyancy ("inject")"""
        top = len(self.levels) - 1
        self.A[top] = A
        self._update_level(top, A)
        ep = self.frc.eddy_param
        if model is None or top == 0 or not ep.is_on:
            return self
        if os.environ.get("NPG_MG_COARSE_NU", self.coarse_nu) == "average":
            # (the fine engine's viscosity table is the one the caller has just re-assembled A from: model.run, src/model.jl:160-170)
            return self._refresh_coarse_levels_averaged()
        if self._inj is None:
            p1 = self._top.spaces.b_order == 1
            self._inj = [injection(self.hierarchy[k].mesh, self.hierarchy[k + 1].mesh, p1) for k in range(top)]
        s_f = self._top.spaces
        nodal = np.where(s_f.b_dof >= 0, model.state.b[np.maximum(s_f.b_dof, 0)], s_f.b_diri_val)   # fine nodal values
        for lev in range(top - 1, -1, -1):
            fed = self.hierarchy[lev]
            nodal = nodal[self._inj[lev]]
            s = fed.spaces
            fe = device_fe(self.arch, fed)
            bl = DeviceVector.from_host(self.ctx, nodal[s.b_dof >= 0], fed.dofs.p_b)
            fe.update_nu_eddy(ep.N2min, self.prm.alpha, self.prm.N2, bl)
            build_A_inversion(self.arch, fed, self.prm, None, A=self.A[lev])
            self._update_level(lev, self.A[lev])
        self._refresh_coarse_dense()
        return self

    def _refresh_coarse_dense(self):
        """the coarsest level's dense inverse after a re-assembly: rebuilt (getrf + getri of the level's matrix) - or, with
        dense_refresh_every = k > 1 (NPG_MG_DENSE_REFRESH_EVERY), only at every k-th refresh: in between the cycle keeps the inverse
        of the previous operator as its coarse solve (an approximate one: the smoother's operators and the level matrices above
        are current), which a coarsest level of several 1e4 unknowns needs - its inversion takes seconds"""
        if not self.coarse_dense:
            return
        self._dense_refreshes = getattr(self, "_dense_refreshes", 0) + 1
        every = int(os.environ.get("NPG_MG_DENSE_REFRESH_EVERY", getattr(self, "dense_refresh_every", 1)))
        if every <= 0 or (every > 1 and self._dense_refreshes % every != 0):
            return
        L.check(L.lib().npg_precond_mg_set_coarse_dense(self.h, self._dense_mode))

    def _refresh_coarse_levels_averaged(self):
        """coarse levels re-discretised with the AVERAGE of the level above's eddy viscosity over each cell's eight children
        (volume-weighted; npg_fe_restrict_coeff, level by level on the device) instead of the viscosity of the buoyancy injected
        into the coarse mesh: nu_eddy = f^2 / sqrt(N2min^2 + (alpha (N2 + d_z b))^2) is a strongly non-linear function of d_z b, and
        d_z of the injected buoyancy is not the average of the fine d_z b.  A Galerkin coarse operator R A P sees the fine
        viscosity; the cell average is what a re-discretisation on the fixed patterns can take of that.  Channel basin h = 0.01:
        16 -> 19-20 outer iterations after a re-assembly instead of 16 -> 31-35 (profiles/r05_coarse_viscosity.txt)."""
        top = len(self.levels) - 1
        for lev in range(top - 1, -1, -1):
            fedc = self.hierarchy[lev]
            fe = device_fe(self.arch, fedc)
            fe.restrict_coeff(device_fe(self.arch, self.hierarchy[lev + 1]), "nu")
            build_A_inversion(self.arch, fedc, self.prm, None, A=self.A[lev])
            self._update_level(lev, self.A[lev])
        self._refresh_coarse_dense()
        return self

    def set_params(self, omega=2.5, jacobi_weight=0.7, schur_sweeps=3, nu1=2, nu2=2, coarse_sweeps=20, cycle="V"):
        L.check(L.lib().npg_precond_mg_set_params(self.h, float(omega), float(jacobi_weight), int(schur_sweeps), int(nu1),
                                                  int(nu2), int(coarse_sweeps)))
        L.check(L.lib().npg_precond_mg_set_cycle(self.h, {"V": 1, "W": 2}[cycle]))
        self.params = dict(omega=omega, jacobi_weight=jacobi_weight, schur_sweeps=schur_sweeps, nu1=nu1, nu2=nu2,
                           coarse_sweeps=coarse_sweeps, cycle=cycle)

    def __repr__(self):
        dense = {1: "fp64", 2: "fp32", 3: "scaled fp16"}.get(self._dense_mode, "")
        coarse = f"dense inverse ({dense} storage)" if self.coarse_dense else "smoothing steps"
        return (f"MultigridPreconditioner({[lv['n'] for lv in self.levels]}, {self.params}, coarsest level: {coarse}"
                f"{', z-line smoother' if self.smoother == 'zline' else ''}"
                f"{', fp32 operator values inside the cycle' if self.mixed else ''})")


class FgmresWorkspace:
    """right-preconditioned flexible GMRES(memory) - npg_fgmres_*; the Krylov workspace for a GeneralPreconditioner"""

    def __init__(self, ctx, n, memory=20):
        h = C.c_void_p()
        L.check(L.lib().npg_fgmres_create(ctx.h, int(n), int(memory), C.byref(h)))
        self.h, self.ctx, self.n, self.memory = h, ctx, n, memory
        self.x = DeviceVector(ctx, n)
        self.stats = None

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_fgmres_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def solve(self, A: DeviceCSR, y: DeviceVector, x: DeviceVector, P, atol=1e-6, rtol=1e-6, itmax=0, scale=1.0,
              **_ignored):
        st = L.SolveStats()
        L.check(L.lib().npg_fgmres_solve(self.h, A.h, None if P is None else P.h, y.h, x.h, float(scale), float(atol),
                                         float(rtol), int(itmax), C.byref(st)))
        self.stats = st.as_dict()
        return self.stats

    def history(self):
        buf = np.empty(int(self.stats["niter"]) + 2 if self.stats else 2)
        k = L.lib().npg_fgmres_history(self.h, L.ptr(buf), buf.size)
        return buf[:max(k, 0)]
