"""Host-side FE substrate of the product (nupgcm_amd.fe) against the fixture-pinned oracle: numbering, labels, patterns,
boundary load vectors.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from nupgcm_amd import fe as pfe
from nupgcm_amd import gmsh_io
from oracle import fe_oracle as fo
from oracle import recipe as rc

U_TAGS = ["bottom", "coastline", "surface"]
U_MASKS = [(True, True, True), (True, True, True), (False, False, True)]


@pytest.fixture(scope="module")
def both(golden_dir):
    mesh = pfe.Mesh(f"{golden_dir}/mesh_bowl3D_h0.1.npz")
    spaces = pfe.Spaces(mesh, u_diri_tags=U_TAGS, u_diri_masks=U_MASKS, b_diri_tags=["coastline", "surface"],
                        b_diri_vals=[lambda x: x[..., 1]] * 2)
    fed = pfe.FEData(mesh, spaces)
    S = rc.setup("bowl_diri")
    return fed, S


def test_topology_matches_oracle(both):
    fed, S = both
    m, t = fed.mesh, S.orc.topo
    assert np.array_equal(m.cells, t.cells)
    assert np.array_equal(m.edges, t.edges)
    assert np.array_equal(m.cell_edges, t.cell_edges)
    assert np.array_equal(m.node_mask, t.node_mask())
    assert np.allclose(m.node_coords, t.p2_coords(), rtol=0, atol=0)


def test_numbering_matches_oracle(both):
    fed, S = both
    s, o = fed.spaces, S.orc.sp
    assert (s.nu, s.np, s.nb) == (o.nu, o.np_, o.nb) == (14792, 1154, 5864)
    assert np.array_equal(s.u_dof, o.u_dof) and np.array_equal(s.p_dof, o.p_dof) and np.array_equal(s.b_dof, o.b_dof)
    assert np.array_equal(s.b_diri_val, o.b_diri)


def test_geometry_and_tables(both):
    fed, S = both
    m, o = fed.mesh, S.orc
    assert np.allclose(m.grad_lambda, o.geo.G, rtol=1e-14, atol=1e-14)
    assert np.allclose(m.detJ, o.geo.detJ, rtol=1e-14)
    lam, w = fo.keast11()
    assert np.allclose(m.q_w, w) and np.allclose(m.q_lam, lam) and abs(m.q_w.sum() - 1 / 6) < 1e-15
    assert np.allclose(m.N2, o.N2q, atol=1e-15)
    assert abs(m.median_edge_length() - 0.1046478656618976) < 1e-15
    assert np.allclose(m.h_cells(), o.h_cells())


def test_surface_load(both):
    fed, S = both
    g = lambda x: 1e-3 * np.sin(np.pi * x[..., 0]) + x[..., 1] ** 2
    assert np.allclose(fed.mesh.surface_load(g), S.orc.surface_integral(g), rtol=1e-13, atol=1e-16)


def _perm_pattern(A, prow, pcol):
    A = sp.csr_matrix(A)[prow][:, pcol]
    A.sort_indices()
    return A


def test_patterns_cover_oracle_matrices(both):
    fed, S = both
    d = fed.dofs
    assert sorted(d.p_inversion) == list(range(d.nu + d.np)) and sorted(d.p_b) == list(range(d.nb))
    # structural pattern == Gridap's stored pattern (explicit zeros included)
    Ao = _perm_pattern(S.A, d.p_inversion, d.p_inversion)
    rp, ci, shape = fed.pattern_A(structural=True)
    assert shape == Ao.shape and np.array_equal(rp, Ao.indptr) and np.array_equal(ci, Ao.indices)
    # numeric pattern: contains every numerically non-zero entry, is contained in the structural one
    An = Ao.copy()
    An.eliminate_zeros()
    rpn, cin, _ = fed.pattern_A()
    Pn = sp.csr_matrix((np.ones(len(cin)), cin, rpn), shape=shape)
    assert (abs(An) > 0).multiply(Pn).nnz == An.nnz
    assert Pn.nnz <= Ao.nnz and Pn.nnz < 0.75 * Ao.nnz
    Bo = _perm_pattern(S.B, d.p_inversion, d.p_b)
    rpb, cib, shb = fed.pattern_B(structural=True)
    assert shb == Bo.shape and np.array_equal(rpb, Bo.indptr) and np.array_equal(cib, Bo.indices)
    Mo = _perm_pattern(S.M, d.p_b, d.p_b)
    rpm, cim, _ = fed.pattern_b()
    assert np.array_equal(rpm, Mo.indptr) and np.array_equal(cim, Mo.indices)


def test_device_tables_are_consistent(both):
    fed, S = both
    t, d, s = fed.tables, fed.dofs, fed.spaces
    assert t.cell_u.shape == (fed.mesh.ncell, 10, 3) and t.cell_u.dtype == np.int32
    free = t.cell_u >= 0
    assert t.cell_u[free].max() < d.nu            # velocity DoFs occupy the first nu slots of [u; p]
    assert (t.cell_p[t.cell_p >= 0] >= d.nu).all()
    # Dirichlet codes point at the right values
    neg = t.cell_b < 0
    vals = t.b_diri[-1 - t.cell_b[neg]]
    assert np.array_equal(vals, s.b_diri_val[s.cell_b_nodes][neg])


def test_gmsh_roundtrip(tmp_path, golden_dir):
    m = gmsh_io.load_npz(f"{golden_dir}/mesh_bowl3D_h0.1.npz")
    gmsh_io.save_npz(m, str(tmp_path / "m.npz"))
    m2 = gmsh_io.load_npz(str(tmp_path / "m.npz"))
    assert np.array_equal(m.cells, m2.cells) and m.phys_names == m2.phys_names


def test_node_block_dof_order(both):
    """fe._node_block_order: [x,y,z of every node with three free components | x,y of the nodes with free x,y only | rest],
    nodes in the RCM order of the x component - a permutation, with the components of a node adjacent."""
    fed = both[0]
    s, d = fed.spaces, fed.dofs
    free = s.u_dof >= 0
    nfull, nsurf = int(free.all(axis=1).sum()), int((free[:, 0] & free[:, 1] & ~free[:, 2]).sum())
    assert (d.n_full, d.n_surf) == (nfull, nsurf) and nfull > 0 and nsurf > 0
    assert np.array_equal(np.sort(d.p_u), np.arange(d.nu))
    node_of = np.full(d.nu, -1)
    comp_of = np.full(d.nu, -1)
    for a in range(3):
        n = np.nonzero(free[:, a])[0]
        node_of[s.u_dof[n, a]] = n
        comp_of[s.u_dof[n, a]] = a
    tri = d.p_u[:3 * nfull].reshape(-1, 3)
    assert np.array_equal(comp_of[tri], np.tile([0, 1, 2], (nfull, 1)))
    assert (node_of[tri] == node_of[tri][:, :1]).all() and free[node_of[tri[:, 0]]].all()
    par = d.p_u[3 * nfull:3 * nfull + 2 * nsurf].reshape(-1, 2)
    assert np.array_equal(comp_of[par], np.tile([0, 1], (nsurf, 1)))
    assert (node_of[par[:, 0]] == node_of[par[:, 1]]).all() and not free[node_of[par[:, 0]], 2].any()
