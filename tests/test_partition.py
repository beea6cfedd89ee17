"""Mesh partition host logic on CPU (nupgcm_amd.partition): node-aligned ownership, the cells / local numberings / local
patterns a rank keeps, and the two halo plans per field - checked on the reference's bowl3D h = 0.1 mesh against the global
objects, with two gloo ranks moving the ghost values the way the device transports do."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nupgcm_amd import workloads
from nupgcm_amd.distributed import RowPartition
from nupgcm_amd.partition import NodePartition, RankLayout, halo_plans

from .test_distributed_plan import _exchange, _free_port


@pytest.fixture(scope="module")
def fed():
    return workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))


@pytest.mark.parametrize("world", [2, 3, 5])
def test_node_partition_owns_every_dof_once_and_aligns_the_fields(fed, world):
    d, t = fed.dofs, fed.tables
    part = NodePartition(fed, world)
    io, bo = part.inv_owner(), part.b_owner()
    assert io.shape == (d.nu + d.np,) and bo.shape == (d.nb,) and io.min() == 0 and io.max() == world - 1
    own = [part.inv_owned(r) for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate(own)), np.arange(d.nu + d.np))
    assert np.array_equal(np.sort(np.concatenate([part.b_owned(r) for r in range(world)])), np.arange(d.nb))
    # one owner per NODE: velocity components, pressure and buoyancy of a node sit on the same rank
    for a in range(3):
        on = t.u_pos[:, a] >= 0
        assert np.array_equal(io[t.u_pos[on, a]], part.node_owner[on])
    on = t.b_pos >= 0
    assert np.array_equal(bo[t.b_pos[on]], part.node_owner[:len(t.b_pos)][on])
    # equal SpMV work, whole nodes per rank in the [full | surface] order npg_csr_block_nodes wants
    rowlen = np.diff(fed.pattern_A()[0])
    work = np.array([rowlen[o].sum() for o in own])
    assert work.max() <= 1.02 * work.mean()
    nf = sum(part.local_nodes(r)[0] for r in range(world))
    ns = sum(part.local_nodes(r)[1] for r in range(world))
    assert (nf, ns) == (d.n_full, d.n_surf)
    # and the point of it: far fewer ghosts than cutting each field's own RCM sequence
    rowpart = RowPartition(d.nu, d.np, d.nb, world, d.n_full, d.n_surf)
    g_node = sum(len(RankLayout(fed, part, r).inv.g_sol) for r in range(world))
    g_row = sum(len(RankLayout(fed, rowpart, r).inv.g_sol) for r in range(world))
    assert g_node < 0.5 * g_row, (g_node, g_row)


@pytest.mark.parametrize("world", [2, 4])
def test_rank_layout_cells_tables_and_patterns(fed, world):
    d, t = fed.dofs, fed.tables
    part = NodePartition(fed, world)
    rpA, ciA, shA = fed.pattern_A()
    PA = sp.csr_matrix((np.ones(len(ciA)), ciA, rpA), shape=shA)
    rpB, ciB, shB = fed.pattern_B()
    PB = sp.csr_matrix((np.ones(len(ciB)), ciB, rpB), shape=shB)
    ncells = 0
    for r in range(world):
        lay = RankLayout(fed, part, r)
        ncells += len(lay.cells)
        # every cell that carries an owned DoF is kept, and the kept cells' DoFs all have a local number
        lt = lay.local_tables(fed)
        assert lt.cell_u.shape == (len(lay.cells), 10, 3) and lt.cell_u.max() < lay.inv.n_loc and lt.cell_b.max() < lay.b.n_loc
        gi = lay.inv.globals()
        assert np.array_equal(gi[lt.cell_u[lt.cell_u >= 0]], t.cell_u[lay.cells][lt.cell_u >= 0])
        touched = np.zeros(d.nu + d.np, dtype=bool)
        cu = t.cell_u[lay.cells]
        touched[cu[cu >= 0]] = True
        all_cu = t.cell_u.reshape(len(t.cell_u), -1)
        mine = np.isin(np.where(all_cu >= 0, all_cu, -1), lay.inv.owned[lay.inv.owned < d.nu]).any(axis=1)
        assert set(np.nonzero(mine)[0]) <= set(lay.cells)
        # the local pattern of A: the owned rows of the global one, columns inside [owned | solver ghosts]
        rp, ci, shape = lay.local_pattern(fed.pattern_A(), lay.inv, lay.inv)
        assert shape == (lay.inv.n_own, lay.inv.n_sol)
        Pl = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=shape)
        G = PA[lay.inv.owned][:, gi[:lay.inv.n_sol]]
        assert (abs(Pl - G)).nnz == 0 and Pl.nnz == PA[lay.inv.owned].nnz
        # B: inversion rows x ALL local buoyancy columns
        rp, ci, shape = lay.local_pattern(fed.pattern_B(), lay.inv, lay.b, solver_cols=False)
        Bl = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=shape)
        assert (abs(Bl - PB[lay.inv.owned][:, lay.b.globals()])).nnz == 0 and Bl.nnz == PB[lay.inv.owned].nnz
    # one ghost layer: the ranks' cell sets overlap, but by far less than replication
    assert fed.mesh.ncell < ncells < 1.8 * fed.mesh.ncell


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
        d = fed.dofs
        part = NodePartition(fed, world)
        lay = RankLayout(fed, part, rank)
        for f, owner, n in ((lay.inv, lay.owner_inv, d.nu + d.np), (lay.b, lay.owner_b, d.nb)):
            plan_s, plan_e = halo_plans(dist, rank, f, owner)
            xg = np.sin(0.1 * np.arange(n))                      # a global field; every rank starts with its owned slice
            x = np.zeros(f.n_loc)
            x[:f.n_own] = xg[f.owned]
            _exchange(rank, x[:f.n_sol], f.n_own, plan_s)         # the solver's plan fills the solver ghosts of the VIEW
            assert np.array_equal(x[f.n_own:f.n_sol], xg[f.g_sol])
            _exchange(rank, x, f.n_sol, plan_e)                   # the extra plan fills what is behind them
            assert np.array_equal(x, xg[f.globals()])
            assert all(int(i) < f.n_own for i in plan_e["send_idx"])       # only truly owned entries are ever sent
        # the distributed SpMV of the inversion pattern (values = 1): owned rows of the global product
        rp, ci, shape = lay.local_pattern(fed.pattern_A(), lay.inv, lay.inv)
        Al = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=shape)
        rpA, ciA, shA = fed.pattern_A()
        Ag = sp.csr_matrix((np.ones(len(ciA)), ciA, rpA), shape=shA)
        xg = np.cos(0.01 * np.arange(shA[0]))
        xl = np.zeros(lay.inv.n_sol)
        xl[:lay.inv.n_own] = xg[lay.inv.owned]
        _exchange(rank, xl, lay.inv.n_own, halo_plans(dist, rank, lay.inv, lay.owner_inv)[0])
        assert np.allclose(Al @ xl, (Ag @ xg)[lay.inv.owned], rtol=1e-13, atol=1e-13)
        q.put((rank, "ok"))
    except Exception as e:                                        # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_plans_of_the_partitioned_fields_with_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_second_multigrid_level_inherits_ownership_through_the_injection(world):
    """The distributed multigrid's second partitioned level (DistributedMultigridPreconditioner, distributed_levels = 2): a coarse
    node belongs to the rank that owns the fine node it coincides with.  bowl3D h = 0.05 over h = 0.1 on CPU: every coarse unknown
    has one owner, a node's fields sit together, every rank gets rows, the coarse cut follows the fine one, and the rows of the
    prolongation a rank owns reach only coarse columns it owns or holds as ghosts of ITS coarse cells' neighbours' owners."""
    from nupgcm_amd import multigrid as mgm
    models = workloads.bowl_hierarchy_models("bowl3D_h0.05")
    fed_c, fed_f = (workloads.example_fe_data(m) for m in models)
    part_f = NodePartition(fed_f, world)
    inj = mgm.injection(fed_c.mesh, fed_f.mesh, False)
    # a coarse node IS a fine node (new boundary nodes are projected back on the bowl: the edge nodes there moved a little)
    dx = np.abs(fed_c.mesh.node_coords - fed_f.mesh.node_coords[inj]).max(axis=1)
    assert dx.max() < 0.02 and (dx < 1e-12).mean() > 0.7
    part_c = NodePartition(fed_c, world, node_owner=part_f.node_owner[inj])
    assert np.array_equal(part_c.node_owner, part_f.node_owner[inj])
    d = fed_c.dofs
    io = part_c.inv_owner()
    own = [part_c.inv_owned(r) for r in range(world)]
    assert np.array_equal(np.sort(np.concatenate(own)), np.arange(d.nu + d.np)) and all(len(o) > 0 for o in own)
    t = fed_c.tables
    for a in range(3):
        on = t.u_pos[:, a] >= 0
        assert np.array_equal(io[t.u_pos[on, a]], part_c.node_owner[on])
    # the rank's rows of P (fine owned x coarse): columns owned here or by the owner of a neighbouring coarse node
    P = sp.csr_matrix(mgm.prolongation(fed_c, fed_f))
    io_f = part_f.inv_owner()
    for r in range(world):
        lay_c = RankLayout(fed_c, part_c, r)
        Pr = P[part_f.inv_owned(r)]
        cols = np.unique(Pr.indices)
        foreign = cols[io[cols] != r]
        # ghosts of the transfer are few (the cut is shared) and all of them are unknowns of this rank's coarse cells or their
        # immediate neighbours - the halo plan of the transfer (npg_precond_mg_set_transfer_dist) moves exactly these
        assert len(foreign) < 0.25 * len(cols), (r, len(foreign), len(cols))
        known = np.concatenate([lay_c.inv.owned, lay_c.inv.g_sol, lay_c.inv.g_ext])
        assert np.isin(foreign, known).mean() > 0.9
    assert io_f.max() == world - 1


def test_ghost_nodes_of_a_rank_are_whole_nodes_of_a_neighbour():
    """RankLayout.ghost_nodes (what npg_csr_set_ghost_nodes takes): every listed node is 3 (full node) or 2 (surface node) ADJACENT
    ghost columns holding consecutive global DoFs of one node, owned by another rank; most velocity ghosts belong to such a node"""
    from nupgcm_amd import partition, workloads
    fed = workloads.example_fe_data(workloads.bowl_mesh_model("bowl3D_h0.1"))
    d = fed.dofs
    part = partition.NodePartition(fed, 3)
    nf3, nbr = 3 * d.n_full, 3 * d.n_full + 2 * d.n_surf
    for r in range(3):
        lay = partition.RankLayout(fed, part, r)
        first, ncomp = lay.ghost_nodes(fed)
        g = np.asarray(lay.inv.g_sol)
        assert len(first) > 0 and set(ncomp.tolist()) <= {2, 3} and (np.diff(first) >= ncomp[:-1]).all()     # disjoint, ascending
        k = first - lay.inv.n_own
        for a in range(3):
            sel = ncomp > a
            assert (g[k[sel] + a] == g[k[sel]] + a).all()
        assert (g[k[ncomp == 3]] % 3 == 0).all() and (g[k[ncomp == 3]] < nf3).all()
        assert ((g[k[ncomp == 2]] - nf3) % 2 == 0).all() and (g[k[ncomp == 2]] >= nf3).all() and (g[k[ncomp == 2]] < nbr).all()
        assert (lay.owner_inv[g[k]] != r).all()
        nvel = int((g < nbr).sum())
        assert ncomp.sum() >= 0.9 * nvel, (ncomp.sum(), nvel)
