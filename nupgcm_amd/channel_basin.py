"""Tagged, x-periodic tetrahedral mesh of the channel + basin domain of the reference's production configuration.

Stands in for the Gmsh script /root/reference/meshes/channel_basin_no_flat_round_end.jl:4-86 (same construction as
meshes/channel_basin.jl:4-175; periodic pairing at :103-108 / :62-67, physical groups at :110-124 / :69-78), which needs
Gmsh + OpenCASCADE - neither is available offline - and whose meshes are not committed.  What is matched:

  * the domain: 0 <= x <= W = 1, -L/2 <= y <= L/2 = 1, depth H(x, y) of scratch/run.jl:54-97 (`geom = :tub`): a
    re-entrant channel of full depth alpha W for y <= -L/2 + 5 L_channel / 8, shoaling parabolically to the sill at
    y = -L/2 + L_channel where it meets the basin H = alpha W (1 - ((x - W/2)/(W/2))^2), closed at the northern end by
    the revolved parabola about (W/2, L/2 - W/2); a vertical wall at y = -L/2;
  * the physical tags "bottom", "surface", "coastline", "interior" with Gmsh's entity semantics (a node carries the tag
    of the lowest-dimensional entity it lies on: coastline curve > bottom / surface > interior);
  * the periodic pairing: the face x = W of the channel is the image of the face x = 0 under the translation (W, 0, 0)
    (`gmsh.model.mesh.setPeriodic(2, [5], [4], translation)`): `periodic[n]` is the master of node n (itself if none).
    GridapGmsh identifies the vertices of paired nodes in the grid topology while every cell keeps its own node
    coordinates; nupgcm_amd.fe.Mesh and oracle.fe_oracle.build_topo do the same.

Construction (structured-to-tet, no mesh generator needed): a logically rectangular horizontal grid - uniform on the
rectangle, squircle-mapped onto the northern half disc so that its outer grid lines lie ON the coast - with one column
of nodes per grid point.  Column (i, j) has n_ij = max(1, round(H_ij / dz)) layers (0 on the coast), its nodes evenly
spaced between z = 0 and z = -H_ij, so elements stay isotropic (vertical size ~ dz everywhere) and the bottom nodes sit
exactly on z = -H.  Each horizontal triangle carries one prism per layer, cut into three tets with the quadrilateral
diagonals fixed by the (periodic-master) column ids, so neighbouring prisms - and the two sides of the periodic face -
agree; where a column has run out of layers its nodes coincide ("pinching") and the degenerate tets are dropped.
"""
from __future__ import annotations

import numpy as np

from .gmsh_io import GmshModel

PHYS_NAMES = ["bottom", "surface", "coastline", "interior"]
_BOT, _SURF, _COAST, _INT = (np.uint32(1 << i) for i in range(4))

L_DOMAIN, W_DOMAIN = 2.0, 1.0


def depth(x, y, alpha):
    """H((x, y, z)) of /root/reference/scratch/run.jl:54-97 (geom = :tub), vectorised; 0 outside the basin's disc end."""
    x, y = np.broadcast_arrays(np.asarray(x, dtype=float), np.asarray(y, dtype=float))
    L, W = L_DOMAIN, W_DOMAIN
    Lc = L / 4
    Lf = 5 * Lc / 8
    H0 = alpha * W

    def parabola(s, s_max, s_zero):
        return H0 * (1 - ((s - s_max) / (s_zero - s_max)) ** 2)

    Hb = parabola(x, W / 2, 0.0)
    Hc = np.where(y <= -L / 2 + Lf, H0, parabola(y, -L / 2 + Lf, -L / 2 + Lc))
    r = np.sqrt((x - W / 2) ** 2 + (y - (L / 2 - W / 2)) ** 2)
    out = np.where(y <= -L / 2 + Lc, np.maximum(Hc, Hb), np.where(y <= L / 2 - W / 2, Hb, parabola(r, 0.0, W / 2)))
    return np.maximum(out, 0.0)


def channel_basin_model(h, alpha, dz=None) -> GmshModel:
    """Mesh with horizontal spacing ~h and vertical spacing ~dz (default h: isotropic, as the Gmsh meshes)."""
    dz = float(h if dz is None else dz)
    L, W = L_DOMAIN, W_DOMAIN
    m = max(2, int(round(0.5 / h)))          # rows per 0.5 in y; nx = 2 m columns in x (even: the disc mapping needs it)
    nx, ny_rect, ny = 2 * m, 3 * m, 4 * m    # rows 0..3m: rectangle -1 <= y <= 0.5; rows 3m..4m: half disc
    jsill = m                                # y = -L/2 + L_channel: the side walls of the basin start here
    I, J = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), indexing="ij")
    X = I / nx * W
    Y = -L / 2 + J / ny_rect * 1.5
    disc = J > ny_rect
    u = 2.0 * I / nx - 1.0
    v = np.where(disc, (J - ny_rect) / m, 0.0)
    X = np.where(disc, W / 2 + W / 2 * u * np.sqrt(1 - v * v / 2), X)
    Y = np.where(disc, (L / 2 - W / 2) + W / 2 * v * np.sqrt(1 - u * u / 2), Y)
    coast = ((I == 0) | (I == nx)) & (J >= jsill) | (J == ny)
    Hc = np.where(coast, 0.0, depth(X, Y, alpha))
    if np.any(Hc[~coast] <= 0):
        raise ValueError("channel_basin_model: a non-coast column has no depth; h is too coarse for this alpha")
    nlay = np.where(coast, 0, np.maximum(1, np.rint(Hc / dz).astype(np.int64)))
    nlay[nx, :jsill + 1] = nlay[0, :jsill + 1]            # the periodic image has the same column (H is x-periodic)
    # geometric nodes: column-major, level 0 (z = 0) first
    start = np.zeros((nx + 1) * (ny + 1) + 1, dtype=np.int64)
    np.cumsum((nlay + 1).ravel(), out=start[1:])
    nnode = int(start[-1])
    col_id = (I * (ny + 1) + J)
    cstart = start[:-1].reshape(nx + 1, ny + 1)
    colof = np.repeat(np.arange(start.size - 1), (nlay + 1).ravel())
    lev = np.arange(nnode) - start[colof]
    nl_n = nlay.ravel()[colof]
    coords = np.empty((nnode, 3))
    coords[:, 0] = X.ravel()[colof]
    coords[:, 1] = Y.ravel()[colof]
    coords[:, 2] = -Hc.ravel()[colof] * np.where(nl_n > 0, lev / np.maximum(nl_n, 1), 0.0)
    # tags
    is_coast = coast.ravel()[colof]
    on_wall = (J == 0).ravel()[colof]
    phys = np.full(nnode, _INT, dtype=np.uint32)
    phys[lev == 0] = _SURF
    phys[(lev == nl_n) & ~is_coast] = _BOT
    phys[on_wall & (lev > 0)] = _BOT
    phys[on_wall & (lev == 0)] = _COAST
    phys[is_coast] = _COAST
    # periodic pairing: face x = W of the channel (rows 0..jsill) -> face x = 0
    periodic = np.arange(nnode, dtype=np.int64)
    mcol = col_id.copy()
    mcol[nx, :jsill + 1] = col_id[0, :jsill + 1]
    for j in range(jsill + 1):
        k = np.arange(nlay[nx, j] + 1)
        periodic[cstart[nx, j] + k] = cstart[0, j] + k
    # horizontal triangles: diagonal towards the nearer side wall so that the two corner squares of the disc end are cut
    # from their corner node
    i0, j0 = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    i0, j0 = i0.ravel(), j0.ravel()
    c00, c10, c01, c11 = col_id[i0, j0], col_id[i0 + 1, j0], col_id[i0, j0 + 1], col_id[i0 + 1, j0 + 1]
    east = i0 >= nx // 2
    tri = np.concatenate([
        np.stack([c00[east], c10[east], c11[east]], axis=1), np.stack([c00[east], c11[east], c01[east]], axis=1),
        np.stack([c00[~east], c10[~east], c01[~east]], axis=1), np.stack([c10[~east], c11[~east], c01[~east]], axis=1)])
    nl_c = nlay.ravel()
    tri = tri[nl_c[tri].max(axis=1) > 0]                       # all three columns on the coast: no water
    # order the columns of each triangle by master column id: fixes every quadrilateral's diagonal consistently
    mflat = mcol.ravel()
    order = np.argsort(mflat[tri], axis=1, kind="stable")
    tri = np.take_along_axis(tri, order, axis=1)
    cs = start[:-1]
    cells = []
    for k in range(int(nl_c.max())):
        act = tri[nl_c[tri].max(axis=1) > k]
        T = cs[act] + np.minimum(k, nl_c[act])
        B = cs[act] + np.minimum(k + 1, nl_c[act])
        for tet in ((T[:, 0], T[:, 1], T[:, 2], B[:, 0]), (T[:, 1], T[:, 2], B[:, 0], B[:, 1]),
                    (T[:, 2], B[:, 0], B[:, 1], B[:, 2])):
            t = np.stack(tet, axis=1)
            s = np.sort(t, axis=1)
            cells.append(t[(s[:, 1:] != s[:, :-1]).all(axis=1)])
    cells = np.concatenate(cells)
    # boundary triangles: faces met once in the periodic topology; all three nodes at z = 0 -> surface, else bottom
    topo = periodic[cells]
    loc = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    fg = cells[:, loc].reshape(-1, 3)
    ft = np.sort(topo[:, loc].reshape(-1, 3), axis=1)
    o = np.lexsort((ft[:, 2], ft[:, 1], ft[:, 0]))
    fs = ft[o]
    same = (fs[1:] == fs[:-1]).all(axis=1)
    once = np.ones(len(fs), dtype=bool)
    once[1:] &= ~same
    once[:-1] &= ~same
    facets = fg[o[once]]
    top = (lev[facets] == 0).all(axis=1)
    facets_phys = np.where(top, _SURF, _BOT).astype(np.uint32)
    # coastline segments: edges of exactly one surface triangle (coast proper and the top of the southern wall)
    sf = periodic[facets[top]]
    eg = facets[top][:, [[0, 1], [0, 2], [1, 2]]].reshape(-1, 2)
    et = np.sort(sf[:, [[0, 1], [0, 2], [1, 2]]].reshape(-1, 2), axis=1)
    o = np.lexsort((et[:, 1], et[:, 0]))
    es = et[o]
    same = (es[1:] == es[:-1]).all(axis=1)
    once = np.ones(len(es), dtype=bool)
    once[1:] &= ~same
    once[:-1] &= ~same
    ridges = eg[o[once]]
    ridges_phys = np.full(len(ridges), _COAST, dtype=np.uint32)
    return GmshModel(3, coords, phys, cells.astype(np.int64), facets.astype(np.int64), facets_phys,
                     ridges.astype(np.int64), ridges_phys, list(PHYS_NAMES), periodic=periodic)
