"""Pins the CPU oracle (oracle/) against the reference's own golden data (tests/golden/*.npz, extracted from
/root/reference/test/data/*.jld2 by tests/golden/make_fixtures.py).  Known-answer tests K1-K4 of SURVEY.md section 8c."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import fe_oracle as fo
from oracle import recipe as rc


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("mesh,expect", [("mesh_bowl3D_h0.1", (14792, 1154, 5864)),
                                         ("mesh_bowl3D_h0.08", (29353, 2042, 11211)),
                                         ("mesh_bowl2D_h0.1", (990, 108, 349))])
def test_dof_counts(mesh, expect):
    topo = fo.build_topo(rc.load_mesh(mesh))
    s = fo.build_spaces(topo, rc.U_TAGS, rc.U_MASKS, ["coastline", "surface"], lambda x: 0 * x[..., 0])
    assert (s.nu, s.np_, s.nb) == expect


def test_dof_count_flux_bc():
    s = fo.build_spaces(fo.build_topo(rc.load_mesh("mesh_bowl3D_h0.1")), rc.U_TAGS, rc.U_MASKS)
    assert s.nb == 7434          # length of b in bowl_surface_flux.jld2


def test_K1_A_inversion_2D(golden_dir):
    """test/bowl_mixing_tests.jl:50-64: assembled un-permuted A_inversion of the 2-D bowl, values AND pattern."""
    topo = fo.build_topo(rc.load_mesh("mesh_bowl2D_h0.1"))
    s = fo.build_spaces(topo, rc.U_TAGS, rc.U_MASKS, ["coastline", "surface"], lambda x: 0 * x[..., 0])
    orc = fo.Oracle(topo, s, eps=0.2, alpha=0.5, mu_rho=10, N2=2, f=lambda x: 1 + 0.5 * x[..., 1], nu=1.0)
    A = orc.A_inversion()
    z = np.load(f"{golden_dir}/A_bowl_mixing_2D.npz")
    Af = sp.csc_matrix((z["nzval"], z["rowval"] - 1, z["colptr"] - 1), shape=(int(z["m"]), int(z["n"]))).tocsr()
    assert A.nnz == Af.nnz == 38712
    A.sort_indices()
    Af.sort_indices()
    assert np.array_equal(A.indptr, Af.indptr) and np.array_equal(A.indices, Af.indices)
    assert sp.linalg.norm(A - Af) / sp.linalg.norm(Af) < 1e-14


@pytest.fixture(scope="module")
def flux_system():
    return rc.setup("bowl_surface_flux")


def test_K2_inversion_identity(flux_system, golden_dir):
    S = flux_system
    z = np.load(f"{golden_dir}/state_bowl_surface_flux.npz")
    rhs = S.B @ z["b"] + S.b0
    r = S.A @ np.concatenate([z["u"], z["p"]]) - rhs
    assert np.linalg.norm(r) / np.linalg.norm(rhs) < 1e-13
    assert S.A.nnz == 1154824 and S.A.shape == (15946, 15946)


def test_precond_scalar(flux_system):
    h, ne = flux_system.orc.precond_h()
    assert ne == 4625
    assert abs(h - 0.1046478656618976) < 1e-15


@pytest.mark.slow
def test_K3_surface_flux_50_steps(flux_system, golden_dir):
    """The exact state fixture encodes a BDF2 left-hand side on step 1 (older revision of the reference)."""
    z = np.load(f"{golden_dir}/state_bowl_surface_flux.npz")
    u, p, b = rc.run(flux_system, 50, first_step_lhs="bdf2")
    assert rel(b, z["b"]) < 1e-10 and rel(u, z["u"]) < 1e-10 and rel(p, z["p"]) < 1e-10
    # current-source first step (BDF1 LHS): documented deviation from that fixture
    u1, p1, b1 = rc.run(flux_system, 50, first_step_lhs="bdf1")
    assert 3e-4 < rel(b1, z["b"]) < 1e-3 and 3e-3 < rel(u1, z["u"]) < 8e-3


@pytest.mark.slow
@pytest.mark.parametrize("name,fixture,eu_max,eb_max", [("bowl_mixing", "bowl_mixing_3D", 1e-5, 1e-4),
                                                        ("bowl_mixing", "bowl_mixing_2D", 1e-5, 1e-4),   # test/bowl_mixing_tests.jl:109
                                                        ("bowl_diri", "bowl_diri", 2e-4, 4e-4),
                                                        ("bowl_wind", "bowl_wind", 4e-4, 1e-3)])
def test_K4_reference_bar(name, fixture, eu_max, eb_max, golden_dir):
    """The reference's own metric (squared relative L2, < 1e-3): test/bowl_mixing_tests.jl:101-103 and siblings."""
    S = rc.setup(name, mesh="mesh_bowl2D_h0.1") if fixture.endswith("2D") else rc.setup(name)
    z = np.load(f"{golden_dir}/state_{fixture}.npz")
    u, p, b = rc.run(S, 50)
    eu = S.orc.l2_sq_u(u, z["u"]) / S.orc.l2_sq_u(z["u"])
    eb = S.orc.l2_sq_b(b, z["b"]) / S.orc.l2_sq_b(z["b"])
    assert eu < eu_max < 1e-3 + 1e-12 and eb < eb_max < 1e-3 + 1e-12


def test_c_openmp_krylov_restatement_matches_the_numpy_oracle():
    """oracle/krylov_c.c (the all-cores CPU baseline of bench.py) takes the same iterates as oracle/krylov_oracle.py"""
    import os
    import subprocess

    from oracle import krylov_c as kc
    from oracle import krylov_oracle as ko
    subprocess.run(["make", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")],
                   check=True, capture_output=True)
    S = rc.setup("bowl_surface_flux")
    h, _ = S.orc.precond_h()
    y = S.B @ S.orc.interpolate_b(S.cfg["b0"]) + S.b0
    x0 = 1e-3 * np.sin(np.arange(len(y)))
    x1, s1 = ko.gmres(S.A, y, x0=x0, M=1 / h ** 3, itmax=130)
    x2, s2 = kc.gmres(S.A, y, x0=x0, M=1 / h ** 3, itmax=130)
    assert s1["niter"] == s2["niter"] == 130 and not s2["solved"]
    assert np.allclose(s1["residuals"], s2["residuals"], rtol=1e-9) and np.linalg.norm(x1 - x2) < 1e-12 * np.linalg.norm(x1)
    A = (S.M + 0.01 * (S.Kh + S.Kv)).tocsr()
    b = S.M @ np.ones(A.shape[0])
    x1, s1 = ko.cg(A, b, M=1 / A.diagonal())
    x2, s2 = kc.cg(A, b, M=1 / A.diagonal())
    assert s1["niter"] == s2["niter"] and s2["solved"] and np.linalg.norm(x1 - x2) < 1e-12 * np.linalg.norm(x1)


def test_jld2_layout_checkpoint_io(tmp_path):
    """nupgcm_amd._hdf5 writes what `jldsave(ofile; u, p, b, t)` lays out (src/IO.jl:8): user block of 512 bytes with the JLD2
    header line, Float64 datasets u, p, b and a scalar t - and reads the reference's own state files bit for bit."""
    import os
    import subprocess

    from nupgcm_amd import _hdf5
    p = str(tmp_path / "state.jld2")
    d = dict(u=np.arange(7.0) / 3, p=np.ones(3), b=np.linspace(0, 1, 4), t=np.float64(2.5))
    _hdf5.write_flat(p, d)
    back = _hdf5.read_flat(p, ("u", "p", "b", "t", "absent"))
    assert set(back) == {"u", "p", "b", "t"} and all(np.array_equal(back[k].ravel(), np.ravel(d[k])) for k in d)
    raw = open(p, "rb").read(520)
    assert raw.startswith(b"HDF5-based Julia Data Format, version 0.1.1\x00") and raw[512:516] == b"\x89HDF"
    if os.path.exists("/opt/conda/bin/h5dump"):
        hdr = subprocess.run(["/opt/conda/bin/h5dump", "-H", "-B", p], capture_output=True, text=True).stdout
        assert "USERBLOCK_SIZE 512" in hdr and "SUPERBLOCK_VERSION 2" in hdr
        assert hdr.count("H5T_IEEE_F64LE") == 4 and "SCALAR" in hdr
    ref = "/root/reference/test/data/bowl_surface_flux.jld2"
    if os.path.exists(ref):                       # only in the build container: the file itself does not travel
        r = _hdf5.read_flat(ref, ("u", "p", "b", "t"))
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_bowl_surface_flux.npz"))
        assert all(np.array_equal(r[k].ravel(), z[k].ravel()) for k in ("u", "p", "b", "t"))


def test_iperm_fixture_of_the_2d_matrix():
    """test/data/A_bowl_mixing_2D.jld2 also pins `iperm`, the inverse of the reference's RCM p_inversion
    (test/bowl_mixing_tests.jl:58-63: `A[iperm, iperm]` must reproduce the un-permuted assembly).  The fixture matrix is the
    UN-permuted one (K1 compares it entry by entry with the oracle's native-order assembly), so what the permutation pins
    is its own structure: a permutation of 1:N that keeps the velocity and pressure blocks apart
    (p_inversion = [p_u; nu + p_p], src/dofs.jl:38) and - being an RCM - narrows the band of the permuted velocity block."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "A_bowl_mixing_2D.npz"))
    iperm = z["iperm"] - 1
    N = int(z["m"])
    assert np.array_equal(np.sort(iperm), np.arange(N))
    A = sp.csc_matrix((z["nzval"], z["rowval"] - 1, z["colptr"] - 1), shape=(N, N)).tocsr()
    nu = 990                                        # 2-D fixture sizes: nu = 990, np = 108
    perm = np.argsort(iperm)                        # p_inversion
    assert set(perm[:nu]) == set(range(nu)) and set(perm[nu:]) == set(range(nu, N))
    Auu = A[:nu, :nu]
    band = lambda M: int(np.abs(M.tocoo().row - M.tocoo().col).max())
    assert band(Auu[perm[:nu]][:, perm[:nu]]) < band(Auu) / 2
