"""Multi-GPU execution of the hot path: one process per GPU, RCCL over xGMI (new work - the reference is single-device).

Decomposition.  The systems are already RCM-ordered (src/dofs.jl:27-41), so a rank owns CONTIGUOUS row blocks: of the
inversion system a slice of the velocity rows and a slice of the pressure rows (p_inversion = [p_u; nu + p_p]), of the
evolution system a slice of the buoyancy rows.  The Krylov solves - more than 99 % of a timestep - run distributed:

  * SpMV: the rank's rows as an n_owned x (n_owned + n_ghost) CSR block; ghost entries of the input vector are filled from
    the neighbours before every SpMV (npg_halo_exchange: one pack kernel + one grouped ncclSend/ncclRecv per neighbour),
  * inner products: every kernel folds its partial sums to one 32-double row which ncclAllReduce sums over the ranks
    (one all-reduce per GMRES iteration - h = V'w and ||w||^2 together, ||w - V h||^2 by Pythagoras - and two per CG
    iteration; latency-bound: 256-byte messages).

The state ([u; p] and b) is REPLICATED: after each solve the owned slices are all-gathered (8 N bytes per step), and each
rank evaluates the element kernels (advection right-hand side, < 1 % of a step) on the full mesh - so the element layer
needs no ghost-cell plan.  `RowPartition`, `local_block` and `halo_plan` are pure host logic (numpy/scipy) and are
exercised on CPU with gloo (tests/test_distributed_plan.py); everything that touches data on the device goes through
libnupgcm_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib as L


class RowPartition:
    """Contiguous, near-equal row blocks per field.  With the node-block DoF order of nupgcm_amd.fe (n_full nodes of three
    components, then n_surf nodes of two) the velocity boundaries are moved to node starts, so that every rank owns whole
    nodes and can store its owned-by-owned velocity block by node records (npg_csr_block_nodes)."""

    def __init__(self, nu, np_, nb, nranks, n_full=0, n_surf=0):
        self.nu, self.np, self.nb, self.nranks = int(nu), int(np_), int(nb), int(nranks)
        self.n_full, self.n_surf = int(n_full), int(n_surf)
        self.u_bounds = np.linspace(0, nu, nranks + 1).astype(np.int64)
        nf3, nbr = 3 * self.n_full, 3 * self.n_full + 2 * self.n_surf
        for i in range(1, nranks):
            b = int(self.u_bounds[i])
            if b < nf3:
                b -= b % 3
            elif b < nbr:
                b -= (b - nf3) % 2
            self.u_bounds[i] = b
        self.p_bounds = np.linspace(0, np_, nranks + 1).astype(np.int64)
        self.b_bounds = np.linspace(0, nb, nranks + 1).astype(np.int64)

    def local_nodes(self, r):
        """(full nodes, surface nodes) whose rows rank r owns - they lead its local numbering, in this order"""
        a, b = int(self.u_bounds[r]), int(self.u_bounds[r + 1])
        nf3, nbr = 3 * self.n_full, 3 * self.n_full + 2 * self.n_surf
        nfull = (min(b, nf3) - min(a, nf3)) // 3
        nsurf = (min(b, nbr) - min(max(a, nf3), nbr)) // 2 if b > nf3 else 0
        return nfull, nsurf

    def inv_owned(self, r):
        """global row ids of the inversion system [u; p] owned by rank r (ascending)"""
        return np.concatenate([np.arange(self.u_bounds[r], self.u_bounds[r + 1]),
                               self.nu + np.arange(self.p_bounds[r], self.p_bounds[r + 1])])

    def b_owned(self, r):
        return np.arange(self.b_bounds[r], self.b_bounds[r + 1])

    def inv_owner(self):
        o = np.empty(self.nu + self.np, dtype=np.int32)
        for r in range(self.nranks):
            o[self.inv_owned(r)] = r
        return o

    def b_owner(self):
        return (np.searchsorted(self.b_bounds, np.arange(self.nb), side="right") - 1).astype(np.int32)

    def inv_segments(self):
        """(rank, local_off, global_off, len) of every owned slice - what npg_comm_allgather_segments takes"""
        seg = []
        for r in range(self.nranks):
            nu_r = self.u_bounds[r + 1] - self.u_bounds[r]
            seg.append((r, 0, int(self.u_bounds[r]), int(nu_r)))
            seg.append((r, int(nu_r), self.nu + int(self.p_bounds[r]), int(self.p_bounds[r + 1] - self.p_bounds[r])))
        return seg

    def b_segments(self):
        return [(r, 0, int(self.b_bounds[r]), int(self.b_bounds[r + 1] - self.b_bounds[r])) for r in range(self.nranks)]


def local_block(A, owned, owner, with_map=False):
    """Rows `owned` (ascending global ids) of the square global CSR matrix A, renumbered [owned | ghosts].
    Returns (A_loc, ghosts) with ghosts sorted by (owner rank, global id); with_map: also the position of every local entry
    in A's value array (A_loc.data == A.data[map]) - what npg_csr_gather_values needs to refresh the block from a
    re-assembled global matrix."""
    A = sp.csr_matrix(A)
    if with_map:
        A_loc, ghosts = local_block(A, owned, owner)
        tag = sp.csr_matrix((np.arange(1, A.nnz + 1, dtype=np.float64), A.indices, A.indptr), shape=A.shape)
        T_loc, _ = local_block(tag, owned, owner)                 # same pattern, same ordering of the entries
        return A_loc, ghosts, np.rint(T_loc.data).astype(np.int64) - 1
    R = A[owned]
    cols = np.unique(R.indices)
    is_owned = np.zeros(A.shape[1], dtype=bool)
    is_owned[owned] = True
    ghosts = cols[~is_owned[cols]]
    ghosts = ghosts[np.lexsort((ghosts, owner[ghosts]))]
    lut = np.full(A.shape[1], -1, dtype=np.int64)
    lut[owned] = np.arange(len(owned))
    lut[ghosts] = len(owned) + np.arange(len(ghosts))
    A_loc = sp.csr_matrix((R.data, lut[R.indices], R.indptr), shape=(len(owned), len(owned) + len(ghosts)))
    A_loc.sort_indices()
    return A_loc, ghosts


def halo_plan(rank, owned, owner, ghosts_by_rank):
    """From every rank's ghost list: whom this rank receives from / sends to.
    Returns dict(peers, send_ptr, send_idx, recv_ptr): recv segments are consecutive in this rank's ghost order."""
    my_ghosts = ghosts_by_rank[rank]
    g_owner = owner[my_ghosts]
    lut = np.full(len(owner), -1, dtype=np.int64)
    lut[owned] = np.arange(len(owned))
    peers, send_ptr, send_idx, recv_ptr = [], [0], [], [0]
    for q in range(len(ghosts_by_rank)):
        if q == rank:
            continue
        n_recv = int((g_owner == q).sum())                      # my ghosts owned by q (contiguous: sorted by owner)
        needs = ghosts_by_rank[q][owner[ghosts_by_rank[q]] == rank]      # q's ghosts that I own, in q's ghost order
        if n_recv == 0 and len(needs) == 0:
            continue
        peers.append(q)
        send_idx.append(lut[needs])
        send_ptr.append(send_ptr[-1] + len(needs))
        recv_ptr.append(recv_ptr[-1] + n_recv)
    sidx = np.concatenate(send_idx).astype(np.int32) if send_idx else np.zeros(0, np.int32)
    assert recv_ptr[-1] == len(my_ghosts) and (sidx >= 0).all()
    return dict(peers=np.asarray(peers, np.int32), send_ptr=np.asarray(send_ptr, np.int64), send_idx=sidx,
                recv_ptr=np.asarray(recv_ptr, np.int64))


# ---- device side ----------------------------------------------------------------------------------------------------------
class Halo:
    def __init__(self, ctx, n_owned, n_ghost, plan):
        self.ctx, self.n_owned, self.n_ghost = ctx, int(n_owned), int(n_ghost)
        h = C.c_void_p()
        k = self._keep = {k: np.ascontiguousarray(v) for k, v in plan.items()}
        L.check(L.lib().npg_halo_create(ctx.h, self.n_owned, self.n_ghost, len(k["peers"]), L.ptr(k["peers"]),
                                        L.ptr(k["send_ptr"]), L.ptr(k["send_idx"]), L.ptr(k["recv_ptr"]), C.byref(h)))
        self.h = h

    def exchange(self, x):
        L.check(L.lib().npg_halo_exchange(self.h, x.h))

    def __del__(self):
        try:
            if self.h:
                L.lib().npg_halo_destroy(self.h)
                self.h = None
        except Exception:
            pass


def allgather_segments(ctx, local, segments, full):
    seg = np.asarray(segments, dtype=np.int64).reshape(-1, 4)
    r, lo, go, ln = (np.ascontiguousarray(seg[:, i]) for i in range(4))
    r32 = np.ascontiguousarray(r, dtype=np.int32)
    L.check(L.lib().npg_comm_allgather_segments(ctx.h, local.h, len(seg), L.ptr(r32), L.ptr(lo), L.ptr(go), L.ptr(ln),
                                                full.h))


class DistributedSolverToolkit:
    """IterativeSolverToolkit whose solve runs on this rank's row block.  `x` is the FULL replicated solution (what the
    model and the element kernels read); `x_loc` = [owned | ghosts] is the solver's own vector (warm start)."""

    def __init__(self, A_loc, P, y, workspace, kwargs, label, x_full, halo, segments, y_range=None):
        self.A, self.P, self.y, self.workspace, self.kwargs, self.label = A_loc, P, y, workspace, dict(kwargs), label
        self.x, self.halo, self.segments, self.y_range = x_full, halo, segments, y_range
        from .architectures import DeviceVector
        self.x_loc = DeviceVector(A_loc.ctx, halo.n_owned + halo.n_ghost)
        self.x_own = self.x_loc.view(0, halo.n_owned)

    def load_owned_from_full(self):
        """x_loc[owned] <- x (used after set_b! / state restores)"""
        me = self.A.ctx.rank
        for (r, lo, go, ln) in self.segments:
            if r == me and ln:
                self.x_loc.view(lo, ln).copy_from(self.x.view(go, ln))

    def solve(self):
        y = self.y if self.y_range is None else self.y.view(*self.y_range)
        self.workspace.solve(self.A, y, self.x_loc, self.P, **self.kwargs)
        allgather_segments(self.A.ctx, self.x_own, self.segments, self.x)


def distribute_model(model, dist, block_nodes=None):
    """Turn a freshly built single-GPU Model into its distributed form (call on every rank, before the first solve).
    `dist` is torch.distributed (initialised); RCCL is bootstrapped from it.  block_nodes: store each rank's
    owned-by-owned velocity block by node records (None: from 100 000 global rows, as InversionToolkit does)."""
    from .architectures import DeviceCSR, DeviceVector, comm_unique_id
    from .iterative_solvers import CgWorkspace, Diagonal, GmresWorkspace
    arch, ctx = model.arch, model.arch.ctx
    rank, world = dist.get_rank(), dist.get_world_size()
    closures = model.forcings.conv_param.is_on or model.forcings.eddy_param.is_on
    # RCCL bootstrap: rank 0's unique id travels over the launcher's process group
    ids = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(ids[0], rank, world)
    d = model.fe_data.dofs
    part = RowPartition(d.nu, d.np, d.nb, world, d.n_full, d.n_surf)
    model.partition = part
    if block_nodes is None:
        block_nodes = d.nu + d.np >= 100000
    block_nodes = block_nodes and not callable(model.forcings.nu) and not model.forcings.eddy_param.is_on

    from .architectures import DeviceIndex
    fed = model.fe_data
    full_stress = callable(model.forcings.nu) or model.forcings.eddy_param.is_on

    def make(pattern, A_dev, owned, owner):
        """This rank's row block of a replicated device matrix WITHOUT moving values through the host: the block's pattern,
        ghost list and entry map come from the global sparsity pattern (host index arrays only), the values from one
        device gather out of the matrix every rank assembled with the element kernels."""
        rp, ci, shape = pattern
        tags = sp.csr_matrix((np.arange(1, len(ci) + 1, dtype=np.float64), ci, rp), shape=shape)
        T_loc, ghosts = local_block(tags, owned, owner) if shape[0] == shape[1] else (sp.csr_matrix(tags[owned]), None)
        T_loc.sort_indices()
        amap = DeviceIndex(ctx, np.rint(T_loc.data).astype(np.int64) - 1, A_dev.nnz)
        A_loc = DeviceCSR.from_pattern(ctx, T_loc.shape[0], T_loc.shape[1], T_loc.indptr, T_loc.indices)
        A_loc.gather_values(A_dev, amap)
        if ghosts is None:
            return A_loc, None, amap
        allg = [None] * world
        dist.all_gather_object(allg, ghosts)
        plan = halo_plan(rank, owned, owner, allg)
        return A_loc, Halo(ctx, len(owned), len(ghosts), plan), amap

    # ---- inversion -------------------------------------------------------------------------------------------------
    inv, s = model.inversion, model.inversion.solver
    owned = part.inv_owned(rank)
    if s.A.storage()[0]:
        raise ValueError("distribute_model: build the model with block_nodes=False (the global matrix is cut into row blocks)")
    A_full = s.A
    A_loc_dev, halo, a_map = make(fed.pattern_A(structural=full_stress), A_full, owned, part.inv_owner())
    B_loc_dev, _, _ = make(fed.pattern_B(), inv.B, owned, None)
    b0_loc = inv.b.to_host()[owned]
    x_full = s.x
    ws = GmresWorkspace(ctx, len(owned), memory=s.workspace.memory)
    L.check(L.lib().npg_gmres_set_halo(ws.h, halo.h))
    inv.B = B_loc_dev
    inv.b = DeviceVector.from_host(ctx, b0_loc)
    if block_nodes:
        # owned nodes lead the local numbering in the global order [full | surface]; couplings to ghosts stay CSR
        A_loc_dev.block_nodes(*part.local_nodes(rank))
    inv.solver = DistributedSolverToolkit(A_loc_dev, Diagonal(scalar=s.P.scalar, n=len(owned)),
                                          DeviceVector(ctx, len(owned)), ws, s.kwargs, s.label, x_full, halo,
                                          part.inv_segments())
    if closures:
        # the closures re-assemble the GLOBAL matrices (every rank runs the element kernels on the replicated state); a
        # rank's block is a fixed subset of their entries and follows by one gather (src/model.jl:160-170,229-261)
        inv.solver.A_full, inv.solver.A_map = A_full, a_map
    # ---- evolution -------------------------------------------------------------------------------------------------
    ev, se = model.evolution, model.evolution.solver
    bo = part.b_owned(rank)
    pat_b = fed.pattern_b()
    A_dev, halo_b, b_map = make(pat_b, se.A, bo, part.b_owner())
    # M, Kh, Kv share A_evo's pattern, hence its ghost set, column numbering and entry map
    Kv_full = ev.Kv

    def loc(Mdev):
        Ml = A_dev.clone()
        return Ml.gather_values(Mdev, b_map)

    ev.M, ev.Kh, ev.Kv = loc(ev.M), loc(ev.Kh), loc(ev.Kv)
    if closures:
        ev.Kv_full, ev.Kv_map = Kv_full, b_map
    wsb = CgWorkspace(ctx, len(bo))
    L.check(L.lib().npg_cg_set_halo(wsb.h, halo_b.h))
    P = Diagonal(A_dev.inv_diag(DeviceVector(ctx, len(bo))))
    ev.solver = DistributedSolverToolkit(A_dev, P, se.y, wsb, se.kwargs, se.label, se.x, halo_b, part.b_segments(),
                                         y_range=(int(part.b_bounds[rank]), len(bo)))
    # a model that already holds a state (set_b!, invert! before distributing): the solvers' own slices follow it
    inv.solver.load_owned_from_full()
    ev.solver.load_owned_from_full()
    model._prev = None
    ctx.sync()                     # (the library's own stream; torch is only the launcher / bootstrap)
    # leave set-up together: device-side waits of the peer transport are bounded (NPG_PEER_TIMEOUT_S), and the first
    # collective should not have to sit out another rank's host-side set-up
    dist.barrier()
    model.comm_layout = dict(n_owned_inv=int(len(owned)), n_ghost_inv=int(halo.n_ghost), n_owned_b=int(len(bo)),
                             n_ghost_b=int(halo_b.n_ghost), peers_inv=[int(q) for q in halo._keep["peers"]])
    return model


def example_model(arch, mesh_model, dist, dt=1e-3, block_nodes=None, **kw):
    from . import workloads
    # the global matrix stays plain CSR (row blocks are gathered out of it on the device); the local blocks get the node records
    return distribute_model(workloads.example_model(arch, mesh_model, dt=dt, block_nodes=False, **kw), dist,
                            block_nodes=block_nodes)


def channel_basin_model(arch, mesh_model, dist, **kw):
    """BASELINE configs[4] distributed: scratch/run.jl on the x-periodic channel-basin mesh (the periodic seam shows up as
    one more neighbour in the halo plan - a rank's off-rank columns are whatever its rows reference), closures refreshed
    through the replicated global matrices."""
    from . import workloads
    return distribute_model(workloads.channel_basin_model(arch, mesh_model=mesh_model, **kw), dist, block_nodes=False)
