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
