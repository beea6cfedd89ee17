"""Host-side finite-element substrate: what Gridap / GridapGmsh / CuthillMcKee hold for the reference
(/root/reference/src/meshes.jl:29-39, src/spaces.jl:31-72, src/dofs.jl:27-100), reduced to the integer tables, reference
shape tables and sparsity patterns that the device kernels consume.  No integration happens here: every volume integral
is evaluated by libnupgcm_hip.so; the only host-side quadrature is the boundary (surface-triangle) load vectors, which
are set-up-time constants.

Conventions reproduced from the un-vendored dependencies (pinned through the oracle tests):
cells = sorted node ids, edges numbered at first encounter over local pairs (1,2),(1,3),(2,3),(1,4),(2,4),(3,4), DoFs =
vertices then edges, vector components interleaved per node, last pressure vertex fixed, `Measure(.,4)` = Keast 11-point
rule on tets and the 3x3 collapsed Gauss-Jacobi x Gauss-Legendre rule on boundary triangles.

3-D tetrahedral meshes only (the hot path's configurations); every array is built with vectorised numpy / scipy.sparse.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
from scipy.special import roots_jacobi, roots_legendre

from . import gmsh_io

_TET_EDGE_A = np.array([0, 0, 1, 0, 1, 2])
_TET_EDGE_B = np.array([1, 2, 2, 3, 3, 3])
_TRI_EDGE_A = np.array([0, 0, 1])
_TRI_EDGE_B = np.array([1, 2, 2])


# ---- reference tables -------------------------------------------------------------------------------------------------
def tet_quadrature_degree4():
    """Keast 11-point rule, barycentric points (11,4) and weights on the unit tet (sum 1/6)."""
    a, b = 1.0 / 14.0, 11.0 / 14.0
    c, d = 0.399403576166799, 0.100596423833201
    lam = [[0.25] * 4]
    lam += [[b if i == k else a for i in range(4)] for k in range(4)]
    lam += [[c if i in pr else d for i in range(4)] for pr in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))]
    w = [-74.0 / 5625.0] + [343.0 / 45000.0] * 4 + [56.0 / 2250.0] * 6
    return np.array(lam), np.array(w)


def tri_quadrature_degree4():
    """3x3 collapsed rule on the unit triangle (sum 1/2); barycentric w.r.t. the face's vertices sorted by node id."""
    tj, wj = roots_jacobi(3, 1, 0)
    tl, wl = roots_legendre(3)
    x = np.repeat((tj + 1) / 2, 3)
    y = np.tile((tl + 1) / 2, 3) * (1 - x)
    w = np.repeat(wj / 4, 3) * np.tile(wl / 2, 3)
    return np.stack([1 - x - y, x, y], axis=1), w


def p2_tables(lam, ea, eb):
    """P2 nodal basis at barycentric points: values (nq, nloc) and d/d(lambda_k) (nq, nloc, nv)."""
    nq, nv = lam.shape
    N = np.concatenate([lam * (2 * lam - 1), 4 * lam[:, ea] * lam[:, eb]], axis=1)
    dN = np.zeros((nq, nv + len(ea), nv))
    for k in range(nv):
        dN[:, k, k] = 4 * lam[:, k] - 1
    for e in range(len(ea)):
        dN[:, nv + e, ea[e]] = 4 * lam[:, eb[e]]
        dN[:, nv + e, eb[e]] = 4 * lam[:, ea[e]]
    return N, dN


# ---- mesh ----------------------------------------------------------------------------------------------------------------
class Mesh:
    """Mesh(ifile; degree=4, surface_tags=["surface"]) - src/meshes.jl:29-39.  `ifile` is a Gmsh 4.1 `.msh`, the `.npz`
    form written by gmsh_io.save_npz, or an already parsed GmshModel."""

    def __init__(self, ifile, degree=4, surface_tags=("surface",)):
        if degree != 4:
            raise NotImplementedError("only Measure(., 4) - the reference's default and only used degree - is tabulated")
        model = ifile if isinstance(ifile, gmsh_io.GmshModel) else gmsh_io.load_model(ifile)
        if model.dim not in (2, 3):
            raise NotImplementedError("nupgcm_amd covers the 3-D (tetrahedral) and embedded 2-D (triangular) meshes of the hot path")
        self.model = model
        self.dim = int(model.dim)
        self.surface_tags = tuple(surface_tags)
        if self.dim == 2:
            self._init_embedded_2d(model)
            return
        # Periodic meshes (meshes/channel_basin.jl:103-108): GridapGmsh glues paired nodes into one topological vertex while
        # cells keep their own node coordinates.  `cells` / `coords` are the TOPOLOGY (vertex ids, one coordinate per vertex:
        # the master's); `cell_geo` / `geo_coords` are the GEOMETRY (the Gmsh nodes of each cell, in the order of its sorted
        # vertices).  Without periodicity the two coincide.
        self.geo_coords = np.ascontiguousarray(model.coords, dtype=np.float64)
        per = getattr(model, "periodic", None)
        if per is None:
            self.vertex_of = np.arange(len(self.geo_coords), dtype=np.int64)
            masters = self.vertex_of
        else:
            masters, self.vertex_of = np.unique(np.asarray(per, dtype=np.int64), return_inverse=True)
        self.periodic = per is not None
        self.coords = self.geo_coords[masters]
        self.nv = len(self.coords)
        cg = np.asarray(model.cells, dtype=np.int64)
        ct = self.vertex_of[cg]
        order = np.argsort(ct, axis=1, kind="stable")
        self.cells = np.take_along_axis(ct, order, axis=1)
        self.cell_geo = np.take_along_axis(cg, order, axis=1)
        if (self.cells[:, 1:] == self.cells[:, :-1]).any():
            raise ValueError("a cell touches a periodic node and its own image: the mesh needs >= 3 cells per period")
        nc = len(self.cells)
        # first-encounter edge numbering
        a = self.cells[:, _TET_EDGE_A].ravel()
        b = self.cells[:, _TET_EDGE_B].ravel()
        key = a * self.nv + b                                  # a < b because cells are sorted
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))
        self.cell_edges = rank[inv].reshape(nc, 6)
        self.edges = np.empty((len(uniq), 2), dtype=np.int64)
        self.edges[rank] = np.stack([uniq // self.nv, uniq % self.nv], axis=1)
        self.ne = len(self.edges)
        self.nn = self.nv + self.ne
        self._edge_key_sorted = uniq
        self._edge_rank = rank
        self.cell_nodes = np.hstack([self.cells, self.nv + self.cell_edges])      # (nc, 10) P2 nodes
        # an edge node sits at the midpoint of the edge as its first cell sees it (for an edge that crosses the periodic
        # seam the two vertex coordinates would be a period apart)
        ga = self.cell_geo[:, _TET_EDGE_A].ravel()[first]
        gb = self.cell_geo[:, _TET_EDGE_B].ravel()[first]
        mid = np.empty((self.ne, 3))
        mid[rank] = 0.5 * (self.geo_coords[ga] + self.geo_coords[gb])
        self.node_coords = np.vstack([self.coords, mid])
        # labels
        self.phys_names = list(model.phys_names)
        emask = np.zeros(self.ne, dtype=np.uint32)
        fg = np.asarray(model.facets, dtype=np.int64).reshape(-1, 3)
        ft = self.vertex_of[fg]
        order = np.argsort(ft, axis=1, kind="stable")
        fac = np.take_along_axis(ft, order, axis=1)            # sorted vertex ids; geometry in the same order
        self._facets_geo = np.take_along_axis(fg, order, axis=1)
        fph = np.asarray(model.facets_phys, dtype=np.uint32)
        for (i, j) in ((0, 1), (0, 2), (1, 2)):
            eid = self.edge_ids(fac[:, i], fac[:, j])
            ok = eid >= 0
            np.bitwise_or.at(emask, eid[ok], fph[ok])
        rid = np.sort(self.vertex_of[np.asarray(model.ridges, dtype=np.int64).reshape(-1, 2)], axis=1)
        if len(rid):
            eid = self.edge_ids(rid[:, 0], rid[:, 1])
            ok = eid >= 0
            emask[eid[ok]] = np.asarray(model.ridges_phys, dtype=np.uint32)[ok]     # a matching 1-D element wins
        vmask = np.zeros(self.nv, dtype=np.uint32)
        np.bitwise_or.at(vmask, self.vertex_of, np.asarray(model.node_phys, dtype=np.uint32))   # a vertex and its images
        self.node_mask = np.concatenate([vmask, emask])
        self._facets, self._facets_phys = fac, fph
        # geometry of the (sorted) cells
        X = self.geo_coords[self.cell_geo]
        J = np.transpose(X[:, 1:, :] - X[:, :1, :], (0, 2, 1))
        self.detJ = np.abs(np.linalg.det(J))
        gref = np.linalg.inv(J)
        self.grad_lambda = np.ascontiguousarray(np.concatenate([-gref.sum(axis=1, keepdims=True), gref], axis=1))
        self.q_lam, self.q_w = tet_quadrature_degree4()
        self.N2, self.dN2 = p2_tables(self.q_lam, _TET_EDGE_A, _TET_EDGE_B)
        self.N1 = self.q_lam.copy()
        self.dN1 = np.broadcast_to(np.eye(4), (len(self.q_w), 4, 4)).copy()

    def _init_embedded_2d(self, model):
        """Triangles embedded in 3-D (the reference's bowl2D meshes, test/bowl_mixing_tests.jl:112-114), with GridapGmsh's
        conventions for them: cells keep their RAW vertex order (only 3-D simplices in 3-D are re-oriented by sorting), edges
        are numbered at first encounter in local order (0,1) (0,2) (1,2), an edge carries the tag of the matching 1-D boundary
        element, gradients are tangential (pseudo-inverse of the 3 x 2 Jacobian), |J| = sqrt(det J'J), Measure(., 4) = the
        3 x 3 collapsed rule.  The DEVICE sees every triangle as a tetrahedron whose fourth barycentric coordinate is
        identically zero: the element kernels are table-driven (shape values, barycentric derivatives, barycentric gradients,
        weights all come from the host), so with grad(lambda_4) = 0, lambda_4 = 0 at every quadrature point and the four
        P2 / one P1 functions of the missing vertex constrained to zero (FEData pads the DoF tables) they integrate exactly
        the triangle forms - no 2-D instances of the kernels are needed."""
        if getattr(model, "periodic", None) is not None:
            raise NotImplementedError("periodic embedded 2-D meshes")
        self.geo_coords = np.ascontiguousarray(model.coords, dtype=np.float64)
        self.vertex_of = np.arange(len(self.geo_coords), dtype=np.int64)
        self.periodic = False
        self.coords = self.geo_coords
        self.nv = len(self.coords)
        self.cells = np.asarray(model.cells, dtype=np.int64).copy()         # raw order
        self.cell_geo = self.cells
        nc = len(self.cells)
        a, b = self.cells[:, _TRI_EDGE_A].ravel(), self.cells[:, _TRI_EDGE_B].ravel()
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = lo * self.nv + hi
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(len(uniq))       # first-encounter numbering
        self.cell_edges = rank[inv].reshape(nc, 3)
        self.edges = np.empty((len(uniq), 2), dtype=np.int64)
        self.edges[rank] = np.stack([uniq // self.nv, uniq % self.nv], axis=1)
        self.ne = len(self.edges)
        self.nn = self.nv + self.ne
        self._edge_key_sorted, self._edge_rank = uniq, rank
        self.cell_nodes = np.hstack([self.cells, self.nv + self.cell_edges])  # (nc, 6) P2 nodes
        self.node_coords = np.vstack([self.coords, 0.5 * (self.coords[self.edges[:, 0]] + self.coords[self.edges[:, 1]])])
        self.phys_names = list(model.phys_names)
        emask = np.zeros(self.ne, dtype=np.uint32)
        fg = np.asarray(model.facets, dtype=np.int64).reshape(-1, 2)          # boundary elements are 1-D here
        fac = np.sort(fg, axis=1)
        fph = np.asarray(model.facets_phys, dtype=np.uint32)
        eid = self.edge_ids(fac[:, 0], fac[:, 1])
        ok = eid >= 0
        emask[eid[ok]] = fph[ok]
        vmask = np.zeros(self.nv, dtype=np.uint32)
        np.bitwise_or.at(vmask, self.vertex_of, np.asarray(model.node_phys, dtype=np.uint32))
        self.node_mask = np.concatenate([vmask, emask])
        self._facets, self._facets_phys, self._facets_geo = fac, fph, fac
        X = self.geo_coords[self.cell_geo]                                    # (nc, 3, 3)
        J = np.transpose(X[:, 1:, :] - X[:, :1, :], (0, 2, 1))                # (nc, 3, 2): columns = edge vectors
        JtJ = np.einsum("cik,cil->ckl", J, J)
        self.detJ = np.sqrt(np.linalg.det(JtJ))
        gref = np.einsum("ckl,cil->cki", np.linalg.inv(JtJ), J)               # pseudo-inverse rows: tangential gradients
        g3 = np.concatenate([-gref.sum(axis=1, keepdims=True), gref], axis=1) # (nc, 3, 3)
        self.grad_lambda = np.ascontiguousarray(np.concatenate([g3, np.zeros((nc, 1, 3))], axis=1))   # lambda_4 = 0: no gradient
        lam3, self.q_w = tri_quadrature_degree4()
        self.q_lam = np.concatenate([lam3, np.zeros((len(lam3), 1))], axis=1)
        self.N2, self.dN2 = p2_tables(self.q_lam, _TET_EDGE_A, _TET_EDGE_B)   # the tetrahedron's tables on its face lambda_4 = 0
        self.N1 = self.q_lam.copy()
        self.dN1 = np.broadcast_to(np.eye(4), (len(self.q_w), 4, 4)).copy()

    @property
    def ncell(self):
        return len(self.cells)

    def edge_ids(self, a, b):
        """edge id of node pairs (a < b), -1 where the pair is not an edge of the mesh"""
        key = np.asarray(a) * self.nv + np.asarray(b)
        pos = np.searchsorted(self._edge_key_sorted, key)
        pos = np.minimum(pos, len(self._edge_key_sorted) - 1)
        hit = self._edge_key_sorted[pos] == key
        return np.where(hit, self._edge_rank[pos], -1)

    def tag_bit(self, name):
        return self.phys_names.index(name)

    def has_tag(self, name):
        return (self.node_mask >> self.tag_bit(name)) & 1 == 1

    def quad_points(self):
        """physical quadrature points (ncell, nq, 3) - where the host evaluates the user's coefficient closures"""
        k = self.cell_geo.shape[1]
        return np.einsum("qk,cki->cqi", self.q_lam[:, :k], self.geo_coords[self.cell_geo])

    def boundary_faces(self, names, with_geometry=False):
        """boundary triangles carrying any of `names` as sorted vertex ids (and, on request, their Gmsh nodes in that order)"""
        sel = np.zeros(len(self._facets), dtype=bool)
        for nm in names:
            sel |= (self._facets_phys >> self.tag_bit(nm)) & 1 == 1
        return (self._facets[sel], self._facets_geo[sel]) if with_geometry else self._facets[sel]

    def surface_load(self, g, names=None):
        """int_Gamma g phi_i dGamma for every P2 node (length nn) over the boundary triangles carrying `names`
        (dGamma = Measure(BoundaryTriangulation(model, tags=surface_tags), 4), src/meshes.jl:35-36)."""
        faces, fgeo = self.boundary_faces(self.surface_tags if names is None else names, with_geometry=True)
        out = np.zeros(self.nn)
        if len(faces) == 0:
            return out
        if getattr(self, "dim", 3) == 2:
            # boundary = segments: 3-point Gauss-Legendre (exact to degree 5), P2 on the segment = two ends + the edge node
            t, w = roots_legendre(3)
            t, w = (t + 1) / 2, w / 2
            N = np.stack([(1 - t) * (1 - 2 * t), t * (2 * t - 1), 4 * t * (1 - t)], axis=1)        # (nq, 3)
            X = self.geo_coords[fgeo]                                                              # (nf, 2, 3)
            length = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
            xq = X[:, None, 0, :] * (1 - t)[None, :, None] + X[:, None, 1, :] * t[None, :, None]
            en = self.edge_ids(faces[:, 0], faces[:, 1])
            nodes = np.stack([faces[:, 0], faces[:, 1], self.nv + en], axis=1)
            vals = np.einsum("f,q,fq,qi->fi", length, w, np.asarray(g(xq), dtype=float) * np.ones(xq.shape[:2]), N)
            np.add.at(out, nodes, vals)
            return out
        lam, w = tri_quadrature_degree4()
        N, _ = p2_tables(lam, _TRI_EDGE_A, _TRI_EDGE_B)
        X = self.geo_coords[fgeo]
        area2 = np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
        en = np.stack([self.edge_ids(faces[:, i], faces[:, j]) for (i, j) in ((0, 1), (0, 2), (1, 2))], axis=1)
        nodes = np.hstack([faces, self.nv + en])
        xq = np.einsum("qk,fki->fqi", lam, X)
        vals = np.einsum("f,q,fq,qi->fi", area2, w, np.asarray(g(xq), dtype=float) * np.ones(xq.shape[:2]), N)
        np.add.at(out, nodes, vals)
        return out

    def h_cells(self):
        """compute_h_cells (src/meshes.jl:127-134): longest edge of each cell"""
        X = self.geo_coords[self.cell_geo]
        return np.linalg.norm(X[:, :, None, :] - X[:, None, :, :], axis=-1).max(axis=(1, 2))

    def median_edge_length(self):
        """The `h` of src/inversion.jl:44-49: median over the unique edges built from local pairs (1,2),(2,3),(3,1) only
        (all_edges, src/meshes.jl:94-108, written for triangles) - `hs[length(hs) // 2]` 1-based."""
        t, g = self.cells, self.cell_geo
        e = np.sort(np.vstack([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)      # (a triangle's three edges, dim 2 or 3)
        eg = np.vstack([g[:, [0, 1]], g[:, [1, 2]], g[:, [2, 0]]])
        _, first = np.unique(e[:, 0] * self.nv + e[:, 1], return_index=True)
        eg = eg[first]                                          # lengths from the cell's own nodes (periodic seam)
        hs = np.sort(np.linalg.norm(self.geo_coords[eg[:, 0]] - self.geo_coords[eg[:, 1]], axis=1))
        return float(hs[len(hs) // 2 - 1])

    def __repr__(self):
        return f"Mesh: {self.ncell} tets, {self.nv} vertices, {self.ne} edges, tags {self.phys_names}"


# ---- spaces --------------------------------------------------------------------------------------------------------------
class Spaces:
    """Spaces(mesh; u_diri_tags, u_diri_masks, u_diri_vals, b_diri_tags, b_diri_vals, u_order=2, b_order=2) -
    src/spaces.jl:31-72.  Velocity P2 (vector), pressure P1 with the last vertex fixed (zero-mean space), buoyancy P2 or
    P1.  Dirichlet values: velocity values are constants per tag; buoyancy values are functions of x (or numbers)."""

    def __init__(self, mesh: Mesh, u_diri_tags=(), u_diri_masks=(), u_diri_vals=None, b_diri_tags=(), b_diri_vals=None,
                 u_order=2, b_order=2):
        if u_order != 2 or b_order not in (1, 2):
            raise NotImplementedError("u_order must be 2 and b_order 1 or 2 (the reference's P2-P1-P2/P1 element)")
        self.mesh, self.b_order = mesh, b_order
        nn = mesh.nn
        diri = np.zeros((nn, 3), dtype=bool)
        uval = np.zeros((nn, 3))
        vals = u_diri_vals if u_diri_vals is not None else [(0.0, 0.0, 0.0)] * len(u_diri_tags)
        for tag, cm, v in reversed(list(zip(u_diri_tags, u_diri_masks, vals))):     # the first listed tag wins
            has = mesh.has_tag(tag)
            for c in range(3):
                if cm[c]:
                    diri[has, c] = True
                    uval[has, c] = float(v[c])
        self.u_dof = np.full((nn, 3), -1, dtype=np.int64)
        self.u_dof[~diri] = np.arange((~diri).sum())
        self.u_diri_val = uval
        self.nu = int((~diri).sum())
        self.p_dof = np.arange(mesh.nv, dtype=np.int64)
        self.p_dof[-1] = -1
        self.np = mesh.nv - 1
        nbn = nn if b_order == 2 else mesh.nv
        self.nb_nodes = nbn
        bdiri = np.zeros(nbn, dtype=bool)
        bval = np.zeros(nbn)
        bvals = b_diri_vals if b_diri_vals is not None else [0.0] * len(b_diri_tags)
        for tag, fn in reversed(list(zip(b_diri_tags, bvals))):
            has = mesh.has_tag(tag)[:nbn]
            bdiri |= has
            bval[has] = fn(mesh.node_coords[:nbn][has]) if callable(fn) else float(fn)
        self.b_dof = np.full(nbn, -1, dtype=np.int64)
        self.b_dof[~bdiri] = np.arange((~bdiri).sum())
        self.b_diri_val = bval
        self.nb = int((~bdiri).sum())
        self.cell_b_nodes = mesh.cell_nodes if b_order == 2 else mesh.cells

    def interpolate_b(self, fn):
        """free values of the interpolant of fn (set_b!, src/model.jl:77-83)"""
        x = self.mesh.node_coords[:self.nb_nodes]
        v = fn(x) if callable(fn) else np.full(len(x), float(fn))
        return np.asarray(v, dtype=float)[self.b_dof >= 0]

    def __repr__(self):
        return f"Spaces: nu={self.nu}, np={self.np}, nb={self.nb} (b_order={self.b_order})"


def get_n_dofs(x):
    """get_n_dofs (src/dofs.jl:51-63)"""
    return x.nu, x.np, x.nb


# ---- DoF handler -----------------------------------------------------------------------------------------------------------
def _node_adjacency(cell_nodes, nn):
    nc, k = cell_nodes.shape
    Cm = sp.csr_matrix((np.ones(nc * k, dtype=np.int32), cell_nodes.ravel(), np.arange(0, nc * k + 1, k)), shape=(nc, nn))
    Adj = (Cm.T @ Cm).tocsr()
    Adj.data[:] = 1
    return Adj


def _selector(node_dof, ncols):
    """sparse (nnodes x ncols) 0/1 matrix mapping node -> its DoF column (rows of constrained nodes are empty)"""
    nodes = np.nonzero(node_dof >= 0)[0]
    return sp.csr_matrix((np.ones(len(nodes), dtype=np.int8), (nodes, node_dof[nodes])), shape=(len(node_dof), ncols))


class DoFHandler:
    """DoFHandler (src/dofs.jl:1-41): RCM permutations per field from the mass-matrix graphs (compute_dof_perms,
    src/dofs.jl:70-100; any valid RCM is acceptable to the reference, test/bowl_mixing_tests.jl:60) and
    p_inversion = [p_u; nu + p_p]."""

    def __init__(self, spaces: Spaces, perms=None):
        m = spaces.mesh
        self.nu, self.np, self.nb = spaces.nu, spaces.np, spaces.nb
        self.adj2 = _node_adjacency(m.cell_nodes, m.nn)                      # P2 node graph
        self.adjb = self.adj2 if spaces.b_order == 2 else _node_adjacency(m.cells, m.nv)
        if perms is None:
            Su = [_selector(spaces.u_dof[:, a], self.nu) for a in range(3)]
            Mu = sum(S.T @ self.adj2 @ S for S in Su)                         # u.v couples equal components only
            Sp = _selector(spaces.p_dof, self.np)
            Mp = Sp.T @ self.adj2[:m.nv, :m.nv] @ Sp
            Sb = _selector(spaces.b_dof, self.nb)
            Mb = Sb.T @ self.adjb @ Sb
            p_u = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(Mu), symmetric_mode=True), dtype=np.int64)
            p_p = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(Mp), symmetric_mode=True), dtype=np.int64)
            p_b = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(Mb), symmetric_mode=True), dtype=np.int64)
            p_u, self.n_full, self.n_surf = _node_block_order(p_u, spaces)
        else:
            p_u, p_p, p_b = (np.asarray(p, dtype=np.int64) for p in perms)
            self.n_full = self.n_surf = 0
        self.p_u, self.p_p, self.p_b = p_u, p_p, p_b
        self.inv_p_u, self.inv_p_p, self.inv_p_b = (_invperm(p) for p in (p_u, p_p, p_b))
        self.p_inversion = np.concatenate([p_u, self.nu + p_p])
        self.inv_p_inversion = _invperm(self.p_inversion)

    def __repr__(self):
        return f"DoFHandler with (nu={self.nu}, np={self.np}, nb={self.nb}) DOFs"


def _node_block_order(p_u, spaces):
    """The velocity mass-matrix graph has one connected component per vector component, so an RCM of it comes out
    component-blocked.  Re-arrange it node by node, keeping the RCM order of the nodes (taken from the x component):

        [ (x, y, z) of every node with three free components | (x, y) of every node with free x, y only | the rest ]

    (in the reference configurations: interior nodes; surface nodes, where w = 0 is imposed; nothing).  This is still "a
    valid RCM-type ordering" for the reference, and it is what lets the device store the friction entries K_xx = K_yy = K_zz
    and the antisymmetric Coriolis entries of a node pair once and fetch a node's components with one gather
    (npg_csr_block_nodes).  Returns (permutation, n_full, n_surf)."""
    free = spaces.u_dof >= 0                                   # (nn, 3)
    nu = len(p_u)
    node_of = np.full(nu, -1, dtype=np.int64)
    comp_of = np.full(nu, -1, dtype=np.int64)
    for a in range(3):
        nodes = np.nonzero(free[:, a])[0]
        node_of[spaces.u_dof[nodes, a]] = nodes
        comp_of[spaces.u_dof[nodes, a]] = a
    xs = p_u[comp_of[p_u] == 0]                                 # x DoFs in RCM order
    if len(xs) == 0:
        return p_u, 0, 0
    order = node_of[xs]                                         # nodes in RCM order
    full = order[free[order].all(axis=1)]
    surf = order[free[order, 0] & free[order, 1] & ~free[order, 2]]
    used = np.zeros(nu, dtype=bool)
    tri = spaces.u_dof[full][:, :3].reshape(-1)                # x, y, z of node 0, x, y, z of node 1, ...
    par = spaces.u_dof[surf][:, :2].reshape(-1)
    used[tri] = True
    used[par] = True
    rest = p_u[~used[p_u]]
    return np.concatenate([tri, par, rest]).astype(np.int64), len(full), len(surf)


def _invperm(p):
    inv = np.empty_like(p)
    inv[p] = np.arange(len(p))
    return inv


@dataclass
class DeviceTables:
    """Everything npg_fe_create wants, with DoF ids already composed with the RCM permutations."""
    cell_u: np.ndarray      # (nc, 10, 3) int32
    cell_p: np.ndarray      # (nc, 4) int32
    cell_b: np.ndarray      # (nc, nloc_b) int32
    u_diri: np.ndarray
    b_diri: np.ndarray
    u_pos: np.ndarray       # (nn, 3) device index of each velocity DoF or -1
    p_pos: np.ndarray       # (nv,)
    b_pos: np.ndarray       # (nb_nodes,)


class FEData:
    """FEData(mesh, spaces) - src/dofs.jl:104-124."""

    def __init__(self, mesh: Mesh, spaces: Spaces, perms=None):
        self.mesh, self.spaces = mesh, spaces
        self.dofs = DoFHandler(spaces, perms)
        self.tables = self._device_tables()

    def _device_tables(self) -> DeviceTables:
        m, s, d = self.mesh, self.spaces, self.dofs
        u_pos = np.where(s.u_dof >= 0, d.inv_p_inversion[np.maximum(s.u_dof, 0)], -1)
        p_pos = np.where(s.p_dof >= 0, d.inv_p_inversion[s.nu + np.maximum(s.p_dof, 0)], -1)
        b_pos = np.where(s.b_dof >= 0, d.inv_p_b[np.maximum(s.b_dof, 0)], -1)
        # Dirichlet entries: -1 - k with k indexing the value tables
        udn, udc = np.nonzero(s.u_dof < 0)
        u_code = u_pos.copy()
        u_code[udn, udc] = -1 - np.arange(len(udn))
        u_diri = s.u_diri_val[udn, udc]
        bdn = np.nonzero(s.b_dof < 0)[0]
        b_code = b_pos.copy()
        b_code[bdn] = -1 - np.arange(len(bdn))
        b_diri = s.b_diri_val[bdn]
        cell_u, cell_p, cell_b = u_code[m.cell_nodes], p_pos[m.cells], b_code[s.cell_b_nodes]
        if getattr(m, "dim", 3) == 2:
            # the device's element is the tetrahedron: a triangle fills its face lambda_4 = 0 (Mesh._init_embedded_2d); the
            # local functions of the missing vertex - P2 slots 3 (vertex) and 7, 8, 9 (its edges), P1 slot 3 - are
            # homogeneous Dirichlet DoFs: one extra zero at the end of each value table, no row, no column
            nc = len(cell_p)
            zu, zb = -1 - len(u_diri), -1 - len(b_diri)
            u_diri, b_diri = np.append(u_diri, 0.0), np.append(b_diri, 0.0)
            cu = np.full((nc, 10, 3), zu, dtype=np.int64)
            cu[:, [0, 1, 2, 4, 5, 6]] = cell_u                     # vertices 0..2, edges (0,1) (0,2) (1,2)
            cell_u = cu
            cell_p = np.concatenate([cell_p, np.full((nc, 1), -1, dtype=np.int64)], axis=1)
            if cell_b.shape[1] == 6:
                cb = np.full((nc, 10), zb, dtype=np.int64)
                cb[:, [0, 1, 2, 4, 5, 6]] = cell_b
            else:
                cb = np.concatenate([cell_b, np.full((nc, 1), zb, dtype=np.int64)], axis=1)
            cell_b = cb
        return DeviceTables(cell_u=np.ascontiguousarray(cell_u, dtype=np.int32),
                            cell_p=np.ascontiguousarray(cell_p, dtype=np.int32),
                            cell_b=np.ascontiguousarray(cell_b, dtype=np.int32),
                            u_diri=np.ascontiguousarray(u_diri), b_diri=np.ascontiguousarray(b_diri),
                            u_pos=u_pos, p_pos=p_pos, b_pos=b_pos)

    # ---- sparsity patterns, in device (permuted) numbering, rows sorted -----------------------------------------
    def _cached(self, key, build):
        cache = self.__dict__.setdefault("_pattern_cache", {})
        if key not in cache:
            cache[key] = build()
        return cache[key]

    def pattern_A(self, structural=False):
        return self._cached(("A", bool(structural)), lambda: self._pattern_A(structural))

    def pattern_B(self, structural=False):
        return self._cached(("B", bool(structural)), lambda: self._pattern_B(structural))

    def pattern_b(self):
        return self._cached(("b",), self._pattern_b)

    def _pattern_A(self, structural=False):
        """Pattern of A_inversion.  structural=False: the numerically non-zero pattern of the constant-nu (Laplacian)
        form - same-component friction, (x,y) Coriolis pairs, u-p and p-u couplings.  structural=True: all nine component
        pairs, as Gridap stores them and as the full-stress form (function-valued nu) needs."""
        m, t, d = self.mesh, self.tables, self.dofs
        N = d.nu + d.np
        S = [_selector(t.u_pos[:, a], N) for a in range(3)]
        Sp = _selector(np.concatenate([t.p_pos, np.full(m.ne, -1)]), N)
        Adj = self.dofs.adj2
        pairs = [(a, c) for a in range(3) for c in range(3)] if structural else [(0, 0), (1, 1), (2, 2), (0, 1), (1, 0)]
        P = sum(S[a].T @ Adj @ S[c] for (a, c) in pairs)
        for a in range(3):
            up = S[a].T @ Adj @ Sp
            P = P + up + up.T
        return _finish_pattern(P)

    def _pattern_B(self, structural=False):
        m, t, d = self.mesh, self.tables, self.dofs
        N = d.nu + d.np
        nbn = self.spaces.nb_nodes
        Sb = _selector(t.b_pos, d.nb)
        Adj = self.dofs.adj2[:, :nbn]
        comps = range(3) if structural else (2,)
        P = sum(_selector(t.u_pos[:, a], N).T @ Adj @ Sb for a in comps)
        return _finish_pattern(P)

    def _pattern_b(self):
        Sb = _selector(self.tables.b_pos, self.dofs.nb)
        return _finish_pattern(Sb.T @ self.dofs.adjb @ Sb)

    def __repr__(self):
        return f"FEData: {self.mesh!r}; {self.spaces!r}"


def _finish_pattern(P):
    P = sp.csr_matrix(P)
    P.sum_duplicates()
    P.sort_indices()
    return P.indptr.astype(np.int64), P.indices.astype(np.int32), P.shape
