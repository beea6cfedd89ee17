"""State checkpoints and VTK output - mirrors /root/reference/src/IO.jl:1-59 (save_state, set_state_from_file!, save_vtk).

The reference stores {u, p, b, t} (free values in the native Gridap DoF order) with JLD2, i.e. in an HDF5 file with a 512-byte
JLD2 header block.  `save_state` writes exactly that layout through libhdf5 (nupgcm_amd/_hdf5.py), and `set_state_from_file`
reads it - so checkpoints can be exchanged with the reference in both directions (its own test/data/*.jld2 state files load
directly).  A path ending in `.npz` selects a NumPy archive with the same four names instead (the committed fixtures under
tests/golden/state_*.npz use it).  `save_vtk` writes an unstructured-grid `.vtu` (ASCII XML) with quadratic tetrahedra, as
`writevtk(...; order=2)` does: u at the P2 nodes, p (P1, interpolated to the edge mid-points), the full buoyancy N2 z + b
and the time as field data, plus the reference's derived fields alpha b_z, nu, kappa_v (src/IO.jl:28-52; nodal recovery of
b_z, closure formulas of src/inputs.jl).  `set_out_dir` mirrors set_out_dir! (src/nuPGCM.jl:26-54): run!'s n_save checkpoints
go to <out_dir>/data/state_<i>.jld2 / .vtu (src/model.jl:194-197)."""
from __future__ import annotations

import numpy as np

# VTK_QUADRATIC_TETRA (type 24) lists the edge nodes as (0,1) (1,2) (0,2) (0,3) (1,3) (2,3); fe.Mesh numbers a cell's edges
# (0,1) (0,2) (1,2) (0,3) (1,3) (2,3)
_VTK_P2 = np.array([0, 1, 2, 3, 4, 6, 5, 7, 8, 9])

out_dir = "."          # src/nuPGCM.jl:26


def set_out_dir(d):
    """set_out_dir!(dir) - src/nuPGCM.jl:36-54: creates dir, dir/images and dir/data"""
    import os
    global out_dir
    out_dir = str(d)
    for sub in ("", "images", "data"):
        os.makedirs(os.path.join(out_dir, sub), exist_ok=True)
    return out_dir


def save_checkpoint(model, i):
    """What run! does every n_save steps (src/model.jl:194-197): state_%016d.jld2 + state_%016d.vtu under out_dir/data."""
    import os
    os.makedirs(os.path.join(out_dir, "data"), exist_ok=True)
    base = os.path.join(out_dir, "data", "state_%016d" % i)
    return save_state(model, base + ".jld2"), save_vtk(model, base + ".vtu")


def save_state(model, ofile):
    """save_state(model, ofile) - src/IO.jl:1-10"""
    s = model.state
    t = 0.0 if model.timestepper is None else float(model.timestepper.t)
    ofile = str(ofile)
    u, p, b = s.u, s.p, s.b                     # (distributed models: each a collective gather of the owned slices)
    if getattr(model.arch.ctx, "rank", 0) != 0:
        return ofile                            # one writer
    if ofile.endswith(".npz"):
        np.savez(ofile, u=u, p=p, b=b, t=t)
    else:
        from . import _hdf5
        _hdf5.write_flat(ofile, dict(u=u, p=p, b=b, t=np.float64(t)))           # jldsave(ofile; u, p, b, t)
    return ofile                                                               # the path actually written


def set_state_from_file(model, ifile):
    """set_state_from_file!(model, ifile) - src/IO.jl:12-23.  As in the reference only the current fields and the time are
    restored: a run resumed from a checkpoint starts its time stepper afresh (previous-step copies = current state)."""
    ifile = str(ifile)
    if ifile.endswith(".npz"):
        z = dict(np.load(ifile))
    else:
        from . import _hdf5
        z = _hdf5.read_flat(ifile, ("u", "p", "b", "t"))                         # jldopen(ifile)["u"] ...
    d = model.fe_data.dofs
    u, p, b = (np.asarray(z[k], dtype=float) for k in ("u", "p", "b"))
    if u.shape != (d.nu,) or p.shape != (d.np,) or b.shape != (d.nb,):
        raise ValueError(f"{ifile}: expected {d.nu}/{d.np}/{d.nb} free values of u/p/b, got {u.shape}/{p.shape}/{b.shape}")
    model.inversion.solver.x.upload(np.concatenate([u, p]), d.p_inversion)
    model.b_vec.upload(b, d.p_b)
    sol = model.inversion.solver
    if hasattr(sol, "load_owned_from_full"):                  # distributed: the solver's own slice follows the full vector
        sol.load_owned_from_full()
    if model.evolution is not None and hasattr(model.evolution.solver, "load_owned_from_full"):
        model.evolution.solver.load_owned_from_full()
    if model.timestepper is not None and "t" in z:
        model.timestepper.t = float(np.asarray(z["t"]).reshape(-1)[0])
    model._prev = None
    model.step_index = 1
    return model


def _nodal_fields(model):
    m, s, prm = model.fe_data.mesh, model.fe_data.spaces, model.params
    st = model.state
    u = np.zeros((m.nn, 3))
    free = s.u_dof >= 0
    u[free] = st.u[s.u_dof[free]]
    u[~free] = s.u_diri_val[~free]
    pv = np.zeros(m.nv)                                            # pressure: P1, the last vertex is the fixed one (p = 0)
    pf = s.p_dof >= 0
    pv[pf] = st.p[s.p_dof[pf]]
    p = np.concatenate([pv, 0.5 * (pv[m.edges[:, 0]] + pv[m.edges[:, 1]])])
    nbn = len(s.b_dof)
    bn = np.where(s.b_dof >= 0, st.b[np.maximum(s.b_dof, 0)], s.b_diri_val)
    if nbn == m.nv:                                               # P1 buoyancy -> P2 nodes
        bn = np.concatenate([bn, 0.5 * (bn[m.edges[:, 0]] + bn[m.edges[:, 1]])])
    b_full = prm.N2 * m.node_coords[:, 2] + bn
    return u, p, b_full


def _nodal_closure_fields(model, b_full):
    """alpha*b_z, nu, kappa_v at the P2 nodes (src/IO.jl:31-42).  The reference hands Gridap cell fields, which writevtk
    evaluates cell by cell; here the cell-wise values at a node (d/dz of the P2 interpolant of b_full through the cell's own
    geometry) are averaged over the cells sharing it, and the closures (src/inputs.jl:87-91, 130-137) or the forcing
    functions are evaluated on that."""
    from .fe import _TET_EDGE_A, _TET_EDGE_B, _TRI_EDGE_A, _TRI_EDGE_B, p2_tables
    m, prm, frc = model.fe_data.mesh, model.params, model.forcings
    k = m.cells.shape[1]                                             # 4 (tetrahedra) or 3 (embedded triangles)
    ea, eb = (_TET_EDGE_A, _TET_EDGE_B) if k == 4 else (_TRI_EDGE_A, _TRI_EDGE_B)
    lam = np.vstack([np.eye(k), 0.5 * (np.eye(k)[ea] + np.eye(k)[eb])])   # the nodes of a cell
    _, dN = p2_tables(lam, ea, eb)                                   # (nodes, basis, barycentric)
    Gz = m.grad_lambda[:, :k, 2]                                     # (nc, k): d lambda_k / d z (tangential on embedded meshes)
    bc = b_full[m.cell_nodes]                                        # (nc, 10 | 6)
    dlam = np.einsum("qak,ca->cqk", dN, bc)                          # d b / d lambda_k at the cell's nodes
    bz = np.einsum("cqk,ck->cq", dlam, Gz)                           # d b / d z
    acc = np.zeros(m.nn)
    cnt = np.zeros(m.nn)
    np.add.at(acc, m.cell_nodes.ravel(), bz.ravel())
    np.add.at(cnt, m.cell_nodes.ravel(), 1.0)
    abz = prm.alpha * acc / np.maximum(cnt, 1.0)
    x = m.node_coords

    def ev(fn):
        return np.asarray(fn(x), dtype=float) * np.ones(m.nn) if callable(fn) else np.full(m.nn, float(fn))

    ep, cp = frc.eddy_param, frc.conv_param
    if ep.is_on:
        f = ev(ep.f)
        v = f * (f / np.sqrt(ep.N2min ** 2 + abz * abz))
        nu = np.logaddexp(10.0 * 1.0, 10.0 * v) / 10.0               # smoothing = 10, nu_min = 1 (src/inputs.jl:130)
    else:
        nu = ev(frc.nu)
    kv = ev(frc.kappa_v)
    if cp.is_on:
        kv = kv + cp.kappa_c * (1.0 + np.tanh(-abz / cp.N2min)) / 2.0
    return abz, nu, kv


def save_vtk(model, ofile):
    """save_vtk(model; ofile) - src/IO.jl:25-59 with order = 2 (quadratic tetrahedra): fields u, p, b = N2 z + b',
    alpha*b_z, nu, kappa_v and t."""
    m = model.fe_data.mesh
    u, p, b = _nodal_fields(model)                 # (distributed models: collective gathers of the state)
    if getattr(model.arch.ctx, "rank", 0) != 0:
        return ofile                               # one writer
    abz, nu, kv = _nodal_closure_fields(model, b)
    t = 0.0 if model.timestepper is None else float(model.timestepper.t)
    tet = m.cell_nodes.shape[1] == 10
    # VTK_QUADRATIC_TETRA (24) / VTK_QUADRATIC_TRIANGLE (22: edge nodes (0,1) (1,2) (2,0); ours (0,1) (0,2) (1,2))
    conn = m.cell_nodes[:, _VTK_P2] if tet else m.cell_nodes[:, [0, 1, 2, 3, 5, 4]]
    nc, npc, vtype = len(conn), conn.shape[1], (24 if tet else 22)

    def arr(a, fmt):
        return "\n".join(" ".join(fmt % v for v in row) for row in np.atleast_2d(a))

    with open(ofile, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n')
        f.write("<UnstructuredGrid>\n<FieldData>\n")
        f.write(f'<DataArray type="Float64" Name="t" NumberOfTuples="1" format="ascii">{t!r}</DataArray>\n</FieldData>\n')
        f.write(f'<Piece NumberOfPoints="{m.nn}" NumberOfCells="{nc}">\n<Points>\n')
        f.write('<DataArray type="Float64" NumberOfComponents="3" format="ascii">\n' + arr(m.node_coords, "%.17g")
                + "\n</DataArray>\n</Points>\n<Cells>\n")
        f.write('<DataArray type="Int64" Name="connectivity" format="ascii">\n' + arr(conn, "%d") + "\n</DataArray>\n")
        f.write('<DataArray type="Int64" Name="offsets" format="ascii">\n'
                + arr((npc * np.arange(1, nc + 1)).reshape(-1, 1), "%d") + "\n</DataArray>\n")
        f.write('<DataArray type="UInt8" Name="types" format="ascii">\n' + arr(np.full((nc, 1), vtype), "%d")
                + "\n</DataArray>\n</Cells>\n<PointData>\n")
        f.write('<DataArray type="Float64" Name="u" NumberOfComponents="3" format="ascii">\n' + arr(u, "%.17g")
                + "\n</DataArray>\n")
        f.write('<DataArray type="Float64" Name="p" format="ascii">\n' + arr(p.reshape(-1, 1), "%.17g") + "\n</DataArray>\n")
        f.write('<DataArray type="Float64" Name="b" format="ascii">\n' + arr(b.reshape(-1, 1), "%.17g") + "\n</DataArray>\n")
        for name, a in (("alpha*b_z", abz), ("nu", nu), ("kappa_v", kv)):
            f.write(f'<DataArray type="Float64" Name="{name}" format="ascii">\n' + arr(a.reshape(-1, 1), "%.17g")
                    + "\n</DataArray>\n")
        f.write("</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
    return ofile
