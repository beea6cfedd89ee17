"""Multi-GPU host logic on CPU: the row partition, local blocks and halo plan of nupgcm_amd.distributed, exercised with
two gloo ranks (the data movement that RCCL does on the GPUs is emulated with torch.distributed send/recv and the local
SpMV with scipy): the distributed SpMV and inner products must reproduce the serial ones exactly."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nupgcm_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _system(n_u=300, n_p=60, seed=0):
    """a banded, structurally symmetric 'saddle point' matrix in [u; p] ordering"""
    rng = np.random.default_rng(seed)
    N = n_u + n_p
    A = sp.diags([rng.standard_normal(N - abs(k)) for k in (-7, -2, -1, 0, 1, 2, 7)], (-7, -2, -1, 0, 1, 2, 7)).tolil()
    for i in range(n_p):                       # pressure rows couple to a few velocity dofs, and back
        js = rng.choice(n_u, 6, replace=False)
        A[n_u + i, js] = rng.standard_normal(6)
        A[js, n_u + i] = rng.standard_normal(6)
    return sp.csr_matrix(A), n_u, n_p


def _exchange(rank, x_loc, n_own, plan):
    """what npg_halo_exchange does, with gloo point-to-point messages"""
    reqs, bufs = [], []
    for i, q in enumerate(plan["peers"]):
        s0, s1 = plan["send_ptr"][i], plan["send_ptr"][i + 1]
        r0, r1 = plan["recv_ptr"][i], plan["recv_ptr"][i + 1]
        if s1 > s0:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(x_loc[plan["send_idx"][s0:s1]])), int(q)))
        if r1 > r0:
            b = torch.empty(r1 - r0, dtype=torch.float64)
            bufs.append((r0, r1, b))
            reqs.append(dist.irecv(b, int(q)))
    for r in reqs:
        r.wait()
    for r0, r1, b in bufs:
        x_loc[n_own + r0:n_own + r1] = b.numpy()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, n_u, n_p = _system()
        part = D.RowPartition(n_u, n_p, 40, world)
        owner = part.inv_owner()
        owned = part.inv_owned(rank)
        A_loc, ghosts = D.local_block(A, owned, owner)
        allg = [None] * world
        dist.all_gather_object(allg, ghosts)
        plan = D.halo_plan(rank, owned, owner, allg)
        x = np.sin(np.arange(A.shape[0], dtype=float))
        x_loc = np.zeros(len(owned) + len(ghosts))
        x_loc[:len(owned)] = x[owned]
        _exchange(rank, x_loc, len(owned), plan)
        assert np.array_equal(x_loc[len(owned):], x[ghosts])          # the ghosts arrived, in plan order
        y_loc = A_loc @ x_loc
        assert np.allclose(y_loc, (A @ x)[owned], rtol=1e-14, atol=1e-14)
        # inner product = all-reduce of owned partial sums
        t = torch.tensor([float(y_loc @ y_loc)], dtype=torch.float64)
        dist.all_reduce(t)
        assert abs(t.item() - float((A @ x) @ (A @ x))) <= 1e-12 * t.item()
        # all-gather by segments reproduces the global vector
        full = np.zeros(A.shape[0])
        for (r, lo, go, ln) in part.inv_segments():
            buf = torch.from_numpy(np.ascontiguousarray(y_loc[lo:lo + ln])) if r == rank else torch.empty(ln, dtype=torch.float64)
            dist.broadcast(buf, src=r)
            full[go:go + ln] = buf.numpy()
        assert np.allclose(full, A @ x, rtol=1e-14, atol=1e-14)
        q.put((rank, "ok", len(ghosts), len(plan["peers"])))
    except Exception as e:                                             # pragma: no cover
        q.put((rank, repr(e), 0, 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_spmv_with_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), res
    assert all(r[2] > 0 for r in res)                                  # every rank really has ghosts


def test_partition_covers_everything():
    part = D.RowPartition(1001, 97, 350, 8)
    inv = np.concatenate([part.inv_owned(r) for r in range(8)])
    assert sorted(inv) == list(range(1001 + 97))
    assert np.array_equal(np.concatenate([part.b_owned(r) for r in range(8)]), np.arange(350))
    own = part.inv_owner()
    assert all((own[part.inv_owned(r)] == r).all() for r in range(8))
    seg = part.inv_segments()
    assert sum(s[3] for s in seg) == 1001 + 97 and len(seg) == 16
    assert (part.b_owner() == np.repeat(np.arange(8), np.diff(part.b_bounds))).all()


def test_local_block_and_plan_serial():
    A, n_u, n_p = _system(seed=3)
    part = D.RowPartition(n_u, n_p, 10, 4)
    owner = part.inv_owner()
    blocks = [D.local_block(A, part.inv_owned(r), owner) for r in range(4)]
    ghosts = [b[1] for b in blocks]
    x = np.cos(np.arange(A.shape[0], dtype=float))
    for r in range(4):
        owned = part.inv_owned(r)
        plan = D.halo_plan(r, owned, owner, ghosts)
        A_loc, gh = blocks[r]
        assert A_loc.shape == (len(owned), len(owned) + len(gh))
        assert not np.isin(gh, owned).any() and (np.diff(owner[gh]) >= 0).all()
        x_loc = np.concatenate([x[owned], x[gh]])
        assert np.allclose(A_loc @ x_loc, (A @ x)[owned], rtol=1e-14, atol=1e-14)
        # what the peers will send me is exactly my ghost segment, peer by peer
        for i, q in enumerate(plan["peers"]):
            peer_plan = D.halo_plan(int(q), part.inv_owned(int(q)), owner, ghosts)
            j = list(peer_plan["peers"]).index(r)
            sent = part.inv_owned(int(q))[peer_plan["send_idx"][peer_plan["send_ptr"][j]:peer_plan["send_ptr"][j + 1]]]
            assert np.array_equal(sent, gh[plan["recv_ptr"][i]:plan["recv_ptr"][i + 1]])


def test_partition_boundaries_on_node_starts():
    """with the node-block DoF order every rank owns whole velocity nodes: [its full nodes | its surface nodes | rest]"""
    n_full, n_surf, rest = 1001, 77, 5
    nu = 3 * n_full + 2 * n_surf + rest
    for world in (1, 2, 3, 5, 8):
        part = D.RowPartition(nu, 300, 500, world, n_full, n_surf)
        assert part.u_bounds[0] == 0 and part.u_bounds[-1] == nu and np.all(np.diff(part.u_bounds) > 0)
        for b in part.u_bounds[1:-1]:
            assert (b % 3 == 0) if b < 3 * n_full else ((b - 3 * n_full) % 2 == 0 or b >= 3 * n_full + 2 * n_surf)
        tot = np.array([part.local_nodes(r) for r in range(world)]).sum(axis=0)
        assert tuple(tot) == (n_full, n_surf)
        for r in range(world):
            nf, ns = part.local_nodes(r)
            assert 3 * nf + 2 * ns <= part.u_bounds[r + 1] - part.u_bounds[r]
        assert np.array_equal(np.sort(np.concatenate([part.inv_owned(r) for r in range(world)])), np.arange(nu + 300))


def test_periodic_mesh_plan_and_value_map():
    """The channel-basin mesh (BASELINE configs[4]): the partition of its RCM-ordered inversion system, the halo plan and the
    entry map that refreshes a rank's row block from a re-assembled global matrix (npg_csr_gather_values).  The periodic
    seam needs no special case - a rank's ghosts are whatever columns its rows reference - but it shows: some rank's
    neighbour set is not just {rank - 1, rank + 1}."""
    from nupgcm_amd import channel_basin, workloads
    fed = workloads.channel_basin_fe_data(channel_basin.channel_basin_model(0.1, 1 / 8, dz=0.04), "flux")
    rp, ci, shape = fed.pattern_A(structural=True)
    rng = np.random.default_rng(5)
    A = sp.csr_matrix((rng.standard_normal(len(ci)), ci, rp), shape=shape)
    d = fed.dofs
    world = 4
    part = D.RowPartition(d.nu, d.np, d.nb, world, d.n_full, d.n_surf)
    owner = part.inv_owner()
    blocks = [D.local_block(A, part.inv_owned(r), owner, with_map=True) for r in range(world)]
    ghosts = [b[1] for b in blocks]
    x = np.cos(np.arange(A.shape[0], dtype=float))
    A2 = sp.csr_matrix((rng.standard_normal(len(ci)), ci, rp), shape=shape)     # "re-assembled": same pattern, new values
    peers = []
    for r in range(world):
        owned = part.inv_owned(r)
        A_loc, gh, amap = blocks[r]
        assert np.array_equal(A_loc.data, A.data[amap])                         # the map reproduces the block ...
        A_new = sp.csr_matrix((A2.data[amap], A_loc.indices, A_loc.indptr), shape=A_loc.shape)
        x_loc = np.concatenate([x[owned], x[gh]])
        assert np.allclose(A_new @ x_loc, (A2 @ x)[owned], rtol=1e-13, atol=1e-13)      # ... and refreshes it
        plan = D.halo_plan(r, owned, owner, ghosts)
        assert plan["recv_ptr"][-1] == len(gh)
        peers.append(set(int(q) for q in plan["peers"]))
    assert all(r in peers[q] for r in range(world) for q in peers[r])            # the plan is symmetric
    assert any(p - {r - 1, r + 1} for r, p in enumerate(peers))                  # the seam (and the pressure rows) reach further
