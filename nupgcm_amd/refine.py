"""Uniform red refinement of a tagged tetrahedral Gmsh model (each tet -> 8, each boundary triangle -> 4, each tagged line
-> 2), with optional projection of the new boundary nodes onto the analytic geometry.

Why it exists: the benchmark configuration "bowl3D refined h ~ 0.02" (BASELINE.json configs[3]) has no committed mesh
and Gmsh (/root/reference/meshes/mesh_bowl3D.jl drives it) is not available offline, so the refined meshes are derived
from the reference's committed h = 0.08 / h = 0.1 bowl meshes.  Physical tags are inherited the way a Gmsh refinement would
classify the new nodes: on a tagged curve -> the curve's tag, else on a boundary surface -> the surface's tag, else the
volume's tag."""
from __future__ import annotations

import numpy as np

from .gmsh_io import GmshModel

_E = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])


# The eight children of refine_once in terms of the parent's vertices (model order): child k's vertex v is the midpoint of
# parent vertices CHILD_VERTS[d][k][v] (a vertex itself when both coincide).  The four corner children are the same for every
# parent; the inner octahedron is cut along one of its three diagonals d = 0: m01-m23, 1: m02-m13, 2: m03-m12 - the SHORTEST
# one (in the parent's straight geometry), the standard choice that keeps the children's shape from degrading (a fixed
# diagonal brings the worst cell of the bowl meshes from 0.126 to 0.036 of a regular tetrahedron's volume / longest-edge^3,
# the shortest one keeps 0.126 on every level).  CHILD_BARY[d, k, v, :] are the barycentric coordinates of child vertices in
# the parent - what nupgcm_amd.multigrid needs to interpolate between the levels of a refinement hierarchy.
_M01, _M02, _M03, _M12, _M13, _M23 = (0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)
_CORNERS = (((0, 0), _M01, _M02, _M03), (_M01, (1, 1), _M12, _M13), (_M02, _M12, (2, 2), _M23), (_M03, _M13, _M23, (3, 3)))
_INNER = (((_M01, _M23, _M02, _M03), (_M01, _M23, _M03, _M13), (_M01, _M23, _M13, _M12), (_M01, _M23, _M12, _M02)),
          ((_M02, _M13, _M01, _M03), (_M02, _M13, _M03, _M23), (_M02, _M13, _M23, _M12), (_M02, _M13, _M12, _M01)),
          ((_M03, _M12, _M01, _M02), (_M03, _M12, _M02, _M23), (_M03, _M12, _M23, _M13), (_M03, _M12, _M13, _M01)))
CHILD_VERTS = tuple(_CORNERS + _INNER[d] for d in range(3))
CHILD_BARY = np.zeros((3, 8, 4, 4))
for _d in range(3):
    for _k, _kid in enumerate(CHILD_VERTS[_d]):
        for _v, (_a, _b) in enumerate(_kid):
            CHILD_BARY[_d, _k, _v, _a] += 0.5
            CHILD_BARY[_d, _k, _v, _b] += 0.5


def _mid_ids(nv, a, b):
    """ids of midpoint nodes for node pairs (a, b): returns (unique_keys, lookup(a, b) -> nv + index)"""
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    return lo * nv + hi


def refine_once(model: GmshModel, project=None) -> GmshModel:
    if model.dim != 3:
        raise NotImplementedError("refine_once: tetrahedral meshes only")
    nv = len(model.coords)
    cells = np.asarray(model.cells, dtype=np.int64)
    keys = _mid_ids(nv, cells[:, _E[:, 0]], cells[:, _E[:, 1]])               # (nc, 6)
    uniq, inv = np.unique(keys.ravel(), return_inverse=True)
    mid = (nv + inv).reshape(-1, 6)                                            # m01 m02 m03 m12 m13 m23
    coords = np.vstack([model.coords, 0.5 * (model.coords[uniq // nv] + model.coords[uniq % nv])])
    # the inner octahedron of every parent is cut along its shortest diagonal (ties: the first)
    d3 = np.stack([np.linalg.norm(coords[mid[:, a]] - coords[mid[:, b]], axis=1) for a, b in ((0, 5), (1, 4), (2, 3))], axis=1)
    variant = np.argmin(d3, axis=1).astype(np.int8)
    node = np.concatenate([cells, mid], axis=1)                 # columns 0-3: vertices, 4-9: m01 m02 m03 m12 m13 m23
    col = {(0, 0): 0, (1, 1): 1, (2, 2): 2, (3, 3): 3, _M01: 4, _M02: 5, _M03: 6, _M12: 7, _M13: 8, _M23: 9}
    new_cells = np.empty((len(cells), 8, 4), dtype=np.int64)
    for d in range(3):
        sel = variant == d
        for k, kid in enumerate(CHILD_VERTS[d]):
            for v, pair in enumerate(kid):
                new_cells[sel, k, v] = node[sel, col[pair]]
    new_cells = new_cells.reshape(-1, 4)

    def lookup(a, b):
        pos = np.searchsorted(uniq, _mid_ids(nv, a, b))
        return nv + pos

    # every new node starts with the volume's tag, then surface, then curve tags override
    interior = 0
    if "interior" in model.phys_names:
        interior = 1 << model.phys_names.index("interior")
    node_phys = np.concatenate([np.asarray(model.node_phys, dtype=np.uint32),
                                np.full(len(uniq), interior, dtype=np.uint32)])
    fac = np.asarray(model.facets, dtype=np.int64).reshape(-1, 3)
    fph = np.asarray(model.facets_phys, dtype=np.uint32)
    a, b, c = fac.T
    mab, mbc, mca = lookup(a, b), lookup(b, c), lookup(c, a)
    for m in (mab, mbc, mca):
        node_phys[m] = fph
    new_fac = np.stack([np.stack(t, axis=1) for t in ((a, mab, mca), (mab, b, mbc), (mca, mbc, c), (mab, mbc, mca))],
                       axis=1).reshape(-1, 3)
    new_fph = np.repeat(fph, 4)
    rid = np.asarray(model.ridges, dtype=np.int64).reshape(-1, 2)
    rph = np.asarray(model.ridges_phys, dtype=np.uint32)
    if len(rid):
        mr = lookup(rid[:, 0], rid[:, 1])
        node_phys[mr] = rph
        new_rid = np.stack([np.stack((rid[:, 0], mr), axis=1), np.stack((mr, rid[:, 1]), axis=1)], axis=1).reshape(-1, 2)
        new_rph = np.repeat(rph, 2)
    else:
        new_rid, new_rph = rid, rph
    out = GmshModel(3, coords, node_phys, new_cells, new_fac, new_fph, new_rid, new_rph, list(model.phys_names))
    if model.periodic is not None:
        # the image of an edge of the paired face is an edge of the master face (the two faces carry the same triangulation):
        # its midpoint is the image of that edge's midpoint; every other new node is its own master
        per = np.asarray(model.periodic, dtype=np.int64)
        ea, eb = uniq // nv, uniq % nv
        ma, mb = per[ea], per[eb]
        mkey = np.minimum(ma, mb) * nv + np.maximum(ma, mb)
        pos = np.minimum(np.searchsorted(uniq, mkey), len(uniq) - 1)
        paired = (ma != ea) & (mb != eb) & (uniq[pos] == mkey)
        out.periodic = np.concatenate([per, np.where(paired, nv + pos, nv + np.arange(len(uniq)))])
    out.child_variant = variant                                 # per PARENT cell: which diagonal its inner children share
    if project is not None:
        out.coords = project(out)
    return out


def refine(model: GmshModel, levels: int, project=None) -> GmshModel:
    for _ in range(levels):
        model = refine_once(model, project)
    return model


def bowl_projector(alpha):
    """Put boundary nodes back on the bowl of /root/reference/meshes/mesh_bowl3D.jl:12-28: bottom z = -alpha (1 - x^2 - y^2)
    (the Bezier generator revolved about z is exactly this paraboloid), coastline = unit circle at z = 0, surface z = 0."""

    def project(m: GmshModel):
        x = m.coords.copy()
        bit = {n: 1 << i for i, n in enumerate(m.phys_names)}
        ph = np.asarray(m.node_phys)
        coast = (ph & bit.get("coastline", 0)) != 0
        surf = ((ph & bit.get("surface", 0)) != 0) & ~coast
        bot = ((ph & bit.get("bottom", 0)) != 0) & ~coast
        r = np.linalg.norm(x[coast, :2], axis=1)
        x[coast, :2] /= r[:, None]
        x[coast, 2] = 0.0
        x[surf, 2] = 0.0
        x[bot, 2] = -alpha * (1.0 - x[bot, 0] ** 2 - x[bot, 1] ** 2)
        return x

    return project


def channel_basin_projector(alpha):
    """Put new boundary nodes of a refined channel-basin mesh (nupgcm_amd.channel_basin) back on the geometry of
    /root/reference/scratch/run.jl:54-97: bottom z = -H(x, y) (the vertical wall at y = -L/2 keeps its nodes), surface and
    coast z = 0, the coast of the northern end on its circle.  H is x-periodic, so paired nodes stay a period apart."""
    from .channel_basin import L_DOMAIN, W_DOMAIN, depth

    def project(m: GmshModel):
        x = m.coords.copy()
        bit = {n: 1 << i for i, n in enumerate(m.phys_names)}
        ph = np.asarray(m.node_phys)
        coast = (ph & bit.get("coastline", 0)) != 0
        surf = ((ph & bit.get("surface", 0)) != 0) & ~coast
        wall = np.abs(x[:, 1] + L_DOMAIN / 2) < 1e-12
        bot = ((ph & bit.get("bottom", 0)) != 0) & ~coast & ~wall
        x[coast | surf, 2] = 0.0
        yc, r0 = L_DOMAIN / 2 - W_DOMAIN / 2, W_DOMAIN / 2
        arc = coast & (x[:, 1] > yc + 1e-12)
        d = x[arc, :2] - np.array([W_DOMAIN / 2, yc])
        x[arc, :2] = np.array([W_DOMAIN / 2, yc]) + d * (r0 / np.linalg.norm(d, axis=1))[:, None]
        x[bot, 2] = -depth(x[bot, 0], x[bot, 1], alpha)
        return x

    return project
