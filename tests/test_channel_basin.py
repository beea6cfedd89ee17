"""BASELINE.json configs[4] on the CPU side: the x-periodic channel-basin mesh (nupgcm_amd.channel_basin, standing in for
/root/reference/meshes/channel_basin.jl), the periodic vertex identification in the product's host FE substrate against
the oracle's independent restatement, and size-independent properties of the periodic discretisation."""
import numpy as np
import pytest
import scipy.sparse as sp

from nupgcm_amd import channel_basin as cb
from nupgcm_amd import fe as pfe
from nupgcm_amd import gmsh_io, workloads
from oracle import recipe as rc

ALPHA = 1 / 8


@pytest.fixture(scope="module")
def model():
    return cb.channel_basin_model(0.1, ALPHA, dz=0.04)


@pytest.fixture(scope="module")
def both(model):
    fed = workloads.channel_basin_fe_data(model, "dirichlet")
    S = rc.setup("channel_basin_dirichlet", model=model)
    return fed, S


@pytest.mark.parametrize("h,dz", [(0.1, 0.04), (0.125, 0.125), (0.05, 0.05)])
def test_mesh_is_conforming_tagged_and_periodic(h, dz):
    m = cb.channel_basin_model(h, ALPHA, dz=dz)
    per = m.periodic
    n = len(m.coords)
    assert per.shape == (n,) and np.array_equal(per[per], per)
    slave = np.nonzero(per != np.arange(n))[0]
    # setPeriodic(2, [5], [4], translation by (W, 0, 0)) - meshes/channel_basin.jl:103-108
    assert len(slave) > 0 and np.allclose(m.coords[slave] - m.coords[per[slave]], [1.0, 0.0, 0.0], atol=1e-14)
    assert (m.coords[slave, 1] <= -0.5 + 1e-12).all()                   # only the channel is re-entrant
    assert np.array_equal(m.node_phys[slave], m.node_phys[per[slave]])
    topo = per[m.cells]
    s = np.sort(topo, axis=1)
    assert (s[:, 1:] != s[:, :-1]).all()
    loc = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    f, cnt = np.unique(np.sort(topo[:, loc].reshape(-1, 3), axis=1), axis=0, return_counts=True)
    assert cnt.max() == 2                                               # conforming: no face with three cells
    bf = f[cnt == 1]
    assert len(bf) == len(m.facets)
    assert np.array_equal(np.unique(np.sort(per[m.facets], axis=1), axis=0), bf)
    X = m.coords[m.cells]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6
    assert vol.min() > 0
    # volume of the water body: the integral of the depth profile of scratch/run.jl:54-97 (converges from below: chords)
    xs, ys = (np.arange(1000) + 0.5) / 1000, -1 + (np.arange(2000) + 0.5) / 1000
    Xg, Yg = np.meshgrid(xs, ys, indexing="ij")
    Hh = cb.depth(Xg, Yg, ALPHA)
    Hh[(Yg > 0.5) & (np.hypot(Xg - 0.5, Yg - 0.5) > 0.5)] = 0
    exact = Hh.sum() / 1e6
    assert 0.93 * exact < vol.sum() <= exact * 1.001
    # tags with Gmsh's entity semantics
    bit = {nm: 1 << i for i, nm in enumerate(m.phys_names)}
    z, xy = m.coords[:, 2], m.coords[:, :2]
    coast, surf, bot = (m.node_phys & bit[k] != 0 for k in ("coastline", "surface", "bottom"))
    assert np.allclose(z[coast | surf], 0) and (z[bot] < 0).all()
    Hn = cb.depth(xy[:, 0], xy[:, 1], ALPHA)
    wall = np.isclose(xy[:, 1], -1.0)
    assert np.allclose(z[bot & ~wall], -Hn[bot & ~wall], atol=1e-12)      # bottom nodes sit on z = -H
    assert (Hn[coast & ~wall] < 1e-12).all() and (Hn[surf] > 0).all()
    fz = m.coords[m.facets][:, :, 2]
    assert ((fz == 0).all(axis=1) == (m.facets_phys == bit["surface"])).all()
    assert (m.ridges_phys == bit["coastline"]).all() and np.allclose(m.coords[m.ridges][:, :, 2], 0)


def test_topology_and_numbering_match_oracle(both):
    fed, S = both
    m, t = fed.mesh, S.orc.topo
    assert m.periodic and m.nv < len(m.geo_coords)
    assert np.array_equal(m.cells, t.cells) and np.array_equal(m.edges, t.edges)
    assert np.array_equal(m.cell_edges, t.cell_edges) and np.array_equal(m.node_mask, t.node_mask())
    assert np.array_equal(m.node_coords, t.p2_coords())
    assert np.array_equal(m.geo_coords[m.cell_geo], t.cell_X)
    s, o = fed.spaces, S.orc.sp
    assert (s.nu, s.np, s.nb) == (o.nu, o.np_, o.nb)
    assert np.array_equal(s.u_dof, o.u_dof) and np.array_equal(s.p_dof, o.p_dof) and np.array_equal(s.b_dof, o.b_dof)
    assert np.array_equal(s.b_diri_val, o.b_diri)
    assert np.allclose(m.grad_lambda, S.orc.geo.G, rtol=1e-13, atol=1e-13) and np.allclose(m.detJ, S.orc.geo.detJ, rtol=1e-13)
    assert abs(m.median_edge_length() - S.orc.precond_h()[0]) < 1e-15
    assert np.allclose(m.h_cells(), S.orc.h_cells())
    g = lambda x: 1e-3 * np.sin(2 * np.pi * x[..., 0]) + x[..., 1] ** 2
    assert np.allclose(fed.mesh.surface_load(g), S.orc.surface_integral(g), rtol=1e-13, atol=1e-16)


def test_patterns_cover_oracle_matrices(both):
    fed, S = both
    d = fed.dofs
    perm = lambda A, pr, pc: (lambda B: (B.sort_indices(), B)[1])(sp.csr_matrix(A)[pr][:, pc])
    Ao = perm(S.A, d.p_inversion, d.p_inversion)
    rp, ci, shape = fed.pattern_A(structural=True)
    assert shape == Ao.shape and np.array_equal(rp, Ao.indptr) and np.array_equal(ci, Ao.indices)
    Bo = perm(S.B, d.p_inversion, d.p_b)
    rpb, cib, shb = fed.pattern_B(structural=True)
    assert shb == Bo.shape and np.array_equal(rpb, Bo.indptr) and np.array_equal(cib, Bo.indices)
    Mo = perm(S.M, d.p_b, d.p_b)
    rpm, cim, _ = fed.pattern_b()
    assert np.array_equal(rpm, Mo.indptr) and np.array_equal(cim, Mo.indices)


def test_periodic_discretisation_properties(model):
    """What the periodic identification must deliver, independent of any numbering: an x-periodic field that the space
    represents is seen as smooth across the seam.  With every buoyancy node free: (i) K_h annihilates constants, rows of M
    sum to the lumped volumes; (ii) for b = y (linear, x-periodic) the interior rows of K_h b vanish - also for the rows of
    seam vertices, whose patches wrap around; (iii) with b = x (NOT periodic) the seam rows do not vanish."""
    S = rc.setup("channel_basin", model=model, kappa=1.0, b_order=2)
    o = S.orc
    Kh, _ = o.K_h()
    M, _ = o.M()
    X = o.topo.cell_X
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6
    assert abs(M.sum() - vol.sum()) < 1e-13 and abs(Kh @ np.ones(Kh.shape[0])).max() < 1e-12
    x = o.topo.p2_coords()
    nmask = o.topo.node_mask()
    interior = nmask == (1 << o.topo.phys_names.index("interior"))
    seam = interior & (np.isclose(x[:, 0], 0.0))
    assert seam.sum() > 0
    r = Kh @ x[:, 1]
    assert abs(r[interior]).max() < 1e-12 * abs(Kh).max()
    r = Kh @ x[:, 0]
    assert abs(r[seam]).max() > 1e-3 and abs(r[interior & ~seam & (x[:, 0] > 0.15) & (x[:, 0] < 0.85)]).max() < 1e-12


def test_npz_roundtrip_keeps_the_pairing(model, tmp_path):
    gmsh_io.save_npz(model, str(tmp_path / "cb.npz"))
    m2 = gmsh_io.load_npz(str(tmp_path / "cb.npz"))
    assert np.array_equal(m2.periodic, model.periodic) and np.array_equal(m2.cells, model.cells)
    assert pfe.Mesh(m2).nv == pfe.Mesh(model).nv


def test_over_long_vertical_lines_are_cut_into_segments():
    """multigrid.line_blocks: the z-line smoother's blocks are inverted in the LDS of one CU (at most 136 unknowns); a line with
    more than that - a mesh with more than ~22 P2 layers - is cut into consecutive vertical segments of whole nodes instead of
    failing the set-up.  Every block stays inside one (x, y) column, its unknowns ascend, the segments of a column are contiguous in z
    and together they still cover every velocity unknown once."""
    from nupgcm_amd import channel_basin as cb
    from nupgcm_amd import multigrid as mgm
    from nupgcm_amd import workloads
    fed = workloads.channel_basin_fe_data(cb.channel_basin_model(0.25, 1 / 8, dz=0.004), "flux")
    bp0, _, _ = mgm.line_blocks(fed, max_unknowns=None)
    assert np.diff(bp0).max() > mgm.MAX_LINE_UNKNOWNS                     # the mesh really has over-long lines
    bp, bd, line_of = mgm.line_blocks(fed)
    assert np.diff(bp).max() <= mgm.MAX_LINE_UNKNOWNS and len(bp) > len(bp0)
    assert np.array_equal(np.sort(bd), np.arange(fed.dofs.nu))
    s, d = fed.spaces, fed.dofs
    node_of = np.full(d.nu, -1, dtype=np.int64)
    for a in range(3):
        nodes = np.nonzero(s.u_dof[:, a] >= 0)[0]
        node_of[d.inv_p_u[s.u_dof[nodes, a]]] = nodes
    xyz = fed.mesh.node_coords[node_of]
    zr = {}
    for b in range(len(bp) - 1):
        idx = bd[bp[b]:bp[b + 1]]
        assert (np.diff(idx) > 0).all() and (line_of[idx] == line_of[idx[0]]).all()
        assert np.ptp(np.round(xyz[idx, 0], 7)) == 0 and np.ptp(np.round(xyz[idx, 1], 7)) == 0
        zr.setdefault((round(xyz[idx[0], 0], 7), round(xyz[idx[0], 1], 7)), []).append((xyz[idx, 2].min(), xyz[idx, 2].max()))
    for segs in zr.values():                                             # segments of one column do not interleave in z
        segs.sort()
        assert all(a[1] <= b[0] + 1e-12 for a, b in zip(segs, segs[1:]))
