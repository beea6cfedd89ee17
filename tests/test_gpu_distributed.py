"""Multi-rank rehearsal of the distributed timestep on ONE GPU: 2 to 4 ranks share the device (RCCL cannot place two ranks on
one device) and exchange either through the PEER-WINDOW transport - the production kernels and protocol of comm.hip: hipIpc-
mapped windows, push / flag / unpack kernels, one-kernel all-reduce, the cycle replayed from one hipGraph - or through the
host-driven shared-memory loop-back transport.  Checks that the distributed SpMV (halo), the distributed GMRES/CG and the
replicated state reproduce the single-GPU run, and that both transports produce the same bits."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import nupgcm_amd as npg
from nupgcm_amd import workloads

from .helpers import rel

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def serial():
    arch = npg.GPU()
    m = workloads.example_model(arch, "bowl3D_h0.1")
    A = m.inversion.solver.A
    xg = np.sin(0.37 * np.arange(A.shape[0]))
    y = A.mul(npg.DeviceVector.from_host(arch.ctx, xg)).to_host()
    npg.invert(m)
    npg.run(m, n_steps=3)
    return m, y


def _launch(world, out, nsteps, mode, transport, split=False, timeout=600):
    env = dict(os.environ, NPG_COMM_TRANSPORT=transport, NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2", NPG_PEER_TIMEOUT_S="60")
    if split:
        # (NPG_HALO_OVERLAP=1: the two-launch form is no longer the default when fewer than half of a rank's tiles are interior,
        #  round 5 - these tests exercise it on purpose)
        env.update(NPG_GMRES_SPLIT="1", NPG_HALO_OVERLAP_VERBOSE="1", NPG_HALO_OVERLAP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_rehearsal_worker.py"), out, str(nsteps),
           mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_peer_and_shm_transports_give_the_same_bits(tmp_path):
    """3 ranks, node-record blocks, the split cycle with the interior tiles beside the exchange: the peer-window transport
    (graph-replayed cycle, device-side all-reduce in rank order) and the host-driven rehearsal transport run the same kernels
    and sum in the same order - identical iterates, iteration counts and state."""
    outs = {}
    for tr in ("peer", "shm"):
        out = str(tmp_path / tr)
        r = _launch(3, out, 3, "blocks", tr, split=True)
        assert "halo overlap on" in r.stderr
        outs[tr] = [np.load(f"{out}.rank{k}.npz") for k in range(3)]
    for a, b in zip(outs["peer"], outs["shm"]):
        assert np.array_equal(a["gm"], b["gm"]) and np.array_equal(a["cg"], b["cg"])
        assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["p"], b["p"]) and np.array_equal(a["b"], b["b"])
        assert np.array_equal(a["y_loc"], b["y_loc"])
    assert all(str(z["transport"]) == "peer" for z in outs["peer"]) and all(str(z["transport"]) == "shm" for z in outs["shm"])


@pytest.mark.parametrize("world,blocks,split,transport", [(2, False, False, "peer"), (3, False, False, "shm"),
                                                          (3, True, False, "peer"), (4, True, False, "peer"),
                                                          (3, True, True, "peer")])
def test_multi_rank_rehearsal(serial, world, blocks, split, transport, tmp_path):
    """split: force the split GMRES cycle that rank blocks of >= 8192 rows use - its Arnoldi step runs the tiles without
    ghost columns before the halo exchange has completed and the others behind it (gmres.hip, launch_cycle_L)."""
    ref, y = serial
    out = str(tmp_path / "dist")
    r = _launch(world, out, 3, "blocks" if blocks else "csr", transport, split=split)
    if split:
        assert "halo overlap on" in r.stderr, r.stderr[-2000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    # every row owned exactly once, and A x assembled from the ranks' owned rows equals the serial product
    owned = np.concatenate([z["owned"] for z in ranks])
    assert np.array_equal(np.sort(owned), np.arange(len(y)))
    yd = np.empty_like(y)
    for z in ranks:
        yd[z["owned"]] = z["y_loc"]
        assert z["n_ghost"] > 0
        assert (z["storage"][1] > 0) == blocks                   # node records in the local block iff asked for
    assert rel(yd, y) < 1e-14
    # replicated state: identical on all ranks, equal to the single-GPU run up to the Krylov tolerance
    for z in ranks[1:]:
        assert np.array_equal(z["b"], ranks[0]["b"]) and np.array_equal(z["u"], ranks[0]["u"])
        assert np.array_equal(z["gm"], ranks[0]["gm"])
    z = ranks[0]
    assert z["solved"].all()
    ref_gm = np.array([s[1]["niter"] for s in ref.stats])
    assert np.all(np.abs(z["gm"] - ref_gm) <= 0.1 * ref_gm + 20), (z["gm"], ref_gm)
    assert rel(z["b"], ref.state.b) < 1e-6
    assert rel(z["u"], ref.state.u) < 1e-3 and rel(z["p"], ref.state.p) < 1e-3


@pytest.mark.parametrize("world,mode", [(2, "partcsr"), (3, "part"), (4, "part")])
def test_mesh_partitioned_model_rehearsal(serial, world, mode, tmp_path):
    """nupgcm_amd.partition: every rank keeps its cells (one ghost layer), assembles its own rows with the element kernels,
    holds its slice of the state [owned | solver ghosts | further ghosts] and refreshes ghosts instead of all-gathering.
    Against the single-GPU run: the distributed SpMV of the locally ASSEMBLED blocks, the 3-step trajectory, and the
    footprint - a rank's matrices are ~1/N of the serial ones, its cells ~1/N + the ghost layer."""
    ref, y = serial
    out = str(tmp_path / "part")
    _launch(world, out, 3, mode, "peer")
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    owned = np.concatenate([z["owned"] for z in ranks])
    assert np.array_equal(np.sort(owned), np.arange(len(y)))
    yd = np.empty_like(y)
    for z in ranks:
        yd[z["owned"]] = z["y_loc"]
        assert (z["storage"][1] > 0) == (mode == "part")
    assert rel(yd, y) < 1e-13                     # locally assembled row blocks + halo == the serial matrix
    for z in ranks[1:]:                           # the gathered state is the same object on every rank
        assert np.array_equal(z["b"], ranks[0]["b"]) and np.array_equal(z["u"], ranks[0]["u"])
        assert np.array_equal(z["gm"], ranks[0]["gm"])
    z = ranks[0]
    assert z["solved"].all()
    ref_gm = np.array([s[1]["niter"] for s in ref.stats])
    assert np.all(np.abs(z["gm"] - ref_gm) <= 0.1 * ref_gm + 20), (z["gm"], ref_gm)
    assert rel(z["b"], ref.state.b) < 1e-6
    assert rel(z["u"], ref.state.u) < 1e-3 and rel(z["p"], ref.state.p) < 1e-3
    # footprint: rows are partitioned (not replicated) and only one layer of cells is shared
    lay = np.array([zz["layout"] for zz in ranks])            # n_own_inv, g_sol, g_ext, n_own_b, g_sol_b, g_ext_b, cells, ncell, bytes
    assert lay[:, 0].sum() == len(y) and lay[:, 6].max() < lay[0, 7] * (1.0 / world + 0.3)
    sol = ref.inversion.solver
    serial_bytes = sum(M.stored_spmv_bytes() for M in (sol.A, ref.inversion.B, ref.evolution.M, ref.evolution.Kh,
                                                       ref.evolution.Kv, ref.evolution.solver.A))
    assert lay[:, 8].max() < 1.25 * serial_bytes / world, (lay[:, 8], serial_bytes)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_multigrid_rehearsal(world, tmp_path):
    """The multigrid-preconditioned inversion on the partitioned mesh (partition.DistributedMultigridPreconditioner: finest
    level row-partitioned, coarse level replicated, flexible GMRES with all-reduced Gram-Schmidt sums) against the one-GPU
    multigrid model of the same hierarchy: the same outer iteration counts step by step and the same trajectory to the solver
    tolerance - it is the same algorithm, only the summation orders differ."""
    arch = npg.GPU()
    label, nsteps = "bowl3D_h0.05", 4
    ref = workloads.example_model(arch, label, preconditioner="multigrid")
    npg.invert(ref)
    npg.run(ref, n_steps=nsteps)
    ref_its = [s[1]["niter"] for s in ref.stats]
    out = str(tmp_path / "dmg")
    env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               NPG_PEER_TIMEOUT_S="90", NPG_MG_DIST_CHECK="1")     # (the check: the device-side operator refresh against the host's)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_mg_worker.py"), out, str(nsteps), label]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    for z in ranks[1:]:
        assert np.array_equal(z["u"], ranks[0]["u"]) and np.array_equal(z["its"], ranks[0]["its"])
    z = ranks[0]
    assert z["solved"].all() and "row-partitioned" in str(z["precond"])
    assert list(z["its"]) == ref_its, (list(z["its"]), ref_its)
    assert rel(z["b"], ref.state.b) < 1e-6 and rel(z["u"], ref.state.u) < 2e-3 and rel(z["p"], ref.state.p) < 2e-3


def test_two_distributed_multigrid_levels(tmp_path):
    """Three-level hierarchy (bowl3D h = 0.1 -> 0.05 -> 0.025, 1.02 M unknowns) on 3 ranks with the TWO finest levels
    row-partitioned (partition.DistributedMultigridPreconditioner(distributed_levels=2): the second level's ownership inherited
    through the injection, its rows assembled on the rank's coarse cells, prolongation and restriction between the two levels as
    row blocks with halo plans of their own) and only the coarsest one replicated: the same outer iteration counts, step by step,
    as the one-GPU multigrid model of the same hierarchy, the same trajectory to the solver tolerance."""
    arch = npg.GPU()
    label, nsteps, world = "bowl3D_h0.025", 3, 3
    ref = workloads.example_model(arch, label, preconditioner="multigrid")
    npg.invert(ref)
    npg.run(ref, n_steps=nsteps)
    ref_its = [s[1]["niter"] for s in ref.stats]
    out = str(tmp_path / "dmg2")
    env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               NPG_PEER_TIMEOUT_S="90", NPG_MG_DIST_CHECK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_mg_worker.py"), out, str(nsteps), label, "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    for z in ranks[1:]:
        assert np.array_equal(z["u"], ranks[0]["u"]) and np.array_equal(z["its"], ranks[0]["its"])
    z = ranks[0]
    assert z["solved"].all()
    # every rank holds a share of the SECOND level too (rows, ghosts of its iterate, of P's and R's inputs)
    for zz in ranks:
        assert zz["mg2"][0] > 0.2 * 134866 / world and zz["mg2"][2] > 0 and zz["mg2"][3] > 0
    assert sum(int(zz["mg2"][0]) for zz in ranks) == 134866
    assert list(z["its"]) == ref_its, (list(z["its"]), ref_its)
    assert rel(z["b"], ref.state.b) < 1e-6 and rel(z["u"], ref.state.u) < 2e-3 and rel(z["p"], ref.state.p) < 2e-3


def test_distributed_multigrid_follows_the_eddy_closure(tmp_path):
    """BASELINE configs[4] with converged inversions on 3 ranks: the distributed multigrid (finest level partitioned, coarse
    level replicated) behind flexible GMRES, both closures on.  At step 10 the eddy closure re-assembles A in the full-stress
    form on every rank's cells (src/model.jl:160-170) and the preconditioner refreshes - the distributed level from the
    re-assembled rows, the replicated level with the injected buoyancy's viscosity - so the outer iteration counts stay those of
    the one-GPU multigrid model through and beyond the refresh, and the trajectories agree to the solver tolerance."""
    world, nsteps, label = 3, 12, "channel_basin_h0.0625"
    arch = npg.GPU()
    # (the distributed preconditioner smooths on node blocks: the serial reference is given the same cycle, not the channel
    #  workloads' z-line default)
    ref = workloads.channel_basin_model(arch, h=0.0625, levels=1, element_precision="fp64", itmax=0,
                                        precond_kw=dict(smoother="node", omega=2.0))
    npg.run(ref, n_steps=nsteps)
    ref_its = [s[1]["niter"] for s in ref.stats]
    assert all(s[1]["solved"] == 1 for s in ref.stats)
    out = str(tmp_path / "dmge")
    env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               NPG_PEER_TIMEOUT_S="90", NPG_MG_DIST_CHECK="1")     # (the check: the device-side operator refresh against the host's)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_mg_worker.py"), out, str(nsteps), label]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    z = ranks[0]
    for zz in ranks[1:]:
        assert np.array_equal(zz["its"], z["its"])
    assert z["solved"].all()
    # the same algorithm with other summation orders: the counts may differ by one here and there, not drift
    assert max(abs(int(a) - int(b)) for a, b in zip(z["its"], ref_its)) <= 2, (list(z["its"]), ref_its)
    assert abs(int(np.sum(z["its"])) - sum(ref_its)) <= 0.05 * sum(ref_its)
    assert rel(z["b"], ref.state.b) < 1e-4 and rel(z["u"], ref.state.u) < 5e-3 and rel(z["p"], ref.state.p) < 5e-3


def test_distributed_multigrid_with_the_zline_smoother(tmp_path):
    """the z-line smoother on a partitioned level: every rank inverts the pieces of the vertical lines it owns (a line cut by a
    rank boundary smooths in pieces, so the counts are not the one-GPU cycle's to the digit) - channel basin, 3 ranks, through the
    eddy closure's re-assembly at step 10: close to the one-GPU z-line counts, far below the node-block smoother's, same trajectory"""
    world, nsteps, label = 3, 12, "channel_basin_h0.0625"
    arch = npg.GPU()
    kw = dict(h=0.0625, levels=1, element_precision="fp64", itmax=0)
    ref = workloads.channel_basin_model(arch, precond_kw=dict(smoother="zline", mixed=False, omega=1.7, coarse_sweeps=20), **kw)
    npg.run(ref, n_steps=nsteps)
    node = workloads.channel_basin_model(arch, precond_kw=dict(smoother="node", omega=2.0), **kw)
    npg.run(node, n_steps=nsteps)
    ref_its, node_its = [s[1]["niter"] for s in ref.stats], [s[1]["niter"] for s in node.stats]
    assert all(s[1]["solved"] == 1 for s in ref.stats)
    out = str(tmp_path / "dmgz")
    env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               NPG_PEER_TIMEOUT_S="90", NPG_TEST_SMOOTHER="zline", NPG_MG_DIST_CHECK="1")     # (device-side refresh of the line pieces against the host's)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_mg_worker.py"), out, str(nsteps), label]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    z = ranks[0]
    for zz in ranks[1:]:
        assert np.array_equal(zz["its"], z["its"])
    assert z["solved"].all() and "z-line" in str(z["precond"])
    assert bool(z["devplan"]), "the z-line level has no device plan: its refresh went through the host"
    print("distributed z-line", list(z["its"]), "one GPU z-line", ref_its, "node blocks", node_its)
    # (at this size - two levels, lines of five nodes - the node-block smoother is only a fifth behind; the gap opens with depth)
    assert int(np.sum(z["its"])) <= 1.35 * sum(ref_its) and int(np.sum(z["its"])) < sum(node_its), (list(z["its"]), ref_its, node_its)
    assert rel(z["b"], ref.state.b) < 1e-4 and rel(z["u"], ref.state.u) < 5e-3 and rel(z["p"], ref.state.p) < 5e-3


def test_two_distributed_levels_follow_the_eddy_closure(tmp_path):
    """the same one refinement finer, with a three-level hierarchy (2 431 / 20 807 / 172 591 unknowns; three levels under
    h = 0.0625 start from a mesh too coarse to precondition this system at all) whose TWO finest levels are partitioned over 3 ranks:
    at step 10 the second level's engine takes the injected buoyancy's viscosity on the rank's coarse cells, its rows and its
    smoother are rebuilt like the finest level's, and the counts stay the one-GPU model's through the refresh"""
    world, nsteps, label = 3, 12, "channel_basin_h0.03125"
    arch = npg.GPU()
    ref = workloads.channel_basin_model(arch, h=0.03125, levels=2, element_precision="fp64", itmax=2000,
                                        precond_kw=dict(smoother="node", omega=2.0))
    npg.run(ref, n_steps=nsteps)
    ref_its = [s[1]["niter"] for s in ref.stats]
    assert all(s[1]["solved"] == 1 for s in ref.stats)
    out = str(tmp_path / "dmge2")
    env = dict(os.environ, NPG_COMM_TRANSPORT="peer", NPG_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               NPG_PEER_TIMEOUT_S="90", NPG_MG_DIST_CHECK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dist_mg_worker.py"), out, str(nsteps), label, "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    z = ranks[0]
    for zz in ranks[1:]:
        assert np.array_equal(zz["its"], z["its"])
    assert z["solved"].all() and all(int(zz["mg2"][0]) > 0 for zz in ranks)           # every rank holds rows of the second level
    assert max(abs(int(a) - int(b)) for a, b in zip(z["its"], ref_its)) <= 2, (list(z["its"]), ref_its)
    assert abs(int(np.sum(z["its"])) - sum(ref_its)) <= 0.05 * sum(ref_its)
    assert rel(z["b"], ref.state.b) < 1e-4 and rel(z["u"], ref.state.u) < 5e-3 and rel(z["p"], ref.state.p) < 5e-3


@pytest.mark.parametrize("records", [False, True])
def test_channel_basin_mesh_partitioned(tmp_path, records, monkeypatch):
    """BASELINE configs[4] on 3 ranks with the mesh partitioned: closures re-evaluated and K_v / the full-stress A
    re-assembled on each rank's own cells (src/model.jl:160-170,229-261), CFL step from the global minimum.
    records: every rank's row block with its record-form companion (full node records, npg_csr_pack_nodes - what production
    sizes get), which follows the re-assembly of step 10."""
    from nupgcm_amd import channel_basin
    if records:
        monkeypatch.setenv("NPG_BLOCK_NODES", "1")          # (the workers inherit it; the serial reference below stays plain)
    world, nsteps = 3, 11
    arch = npg.GPU()
    mm = channel_basin.channel_basin_model(0.0625, workloads.CB_ALPHA)
    monkeypatch.setenv("NPG_PACK_NODES", "0")
    ref = workloads.channel_basin_model(arch, mesh_model=mm, element_precision="fp64")
    npg.run(ref, n_steps=nsteps)
    monkeypatch.delenv("NPG_PACK_NODES")
    out = str(tmp_path / "pcb")
    _launch(world, out, nsteps, "pchannel", "peer")
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    for z in ranks[1:]:
        assert np.array_equal(z["b"], ranks[0]["b"]) and np.array_equal(z["u"], ranks[0]["u"]) and z["dt"] == ranks[0]["dt"]
    z = ranks[0]
    assert list(z["gm"]) == [1000] * nsteps
    assert abs(z["dt"] - ref.timestepper.dt) < 3e-3 * ref.timestepper.dt
    assert rel(z["b"], ref.state.b) < 1e-3 and rel(z["u"], ref.state.u) < 1e-2, (rel(z["b"], ref.state.b), rel(z["u"], ref.state.u))


def test_peer_transport_repeats_under_gpu_contention(tmp_path):
    """The peer-window transport must not depend on WHEN its workgroups get to run.  Round 3 found (tools/dist_repeat_probe.py,
    profiles/r03_halo_epoch_race.txt) that a push workgroup of the one-kernel halo exchange dispatched late - after the
    rank's own consumers had completed the epoch, which needs only the NEIGHBOURS' pushes - read the advanced epoch and wrote
    the other window slot: stale ghosts in one SpMV, visible as run-to-run differences of the unconverged channel-basin
    inversions (itmax = 1000) whenever another process held the CUs.  Here this process keeps the card busy while the 3-rank
    channel run goes through the peer windows twice and through the host-driven rehearsal transport once: identical bits."""
    import threading
    from nupgcm_amd import workloads as wl
    arch = npg.GPU()
    busy = wl.example_model(arch, "bowl3D_h0.05")
    stop = []

    def spin():
        while not stop:
            npg.run(busy, n_steps=1)

    th = threading.Thread(target=spin)
    th.start()
    try:
        res = {}
        for tag, tr in (("peer1", "peer"), ("peer2", "peer"), ("shm", "shm")):
            out = str(tmp_path / tag)
            _launch(3, out, 3, "channel", tr)
            res[tag] = np.load(f"{out}.rank0.npz")
    finally:
        stop.append(1)
        th.join()
    for tag in ("peer2", "shm"):
        for key in ("u", "p", "b", "gm", "cg", "hist_gm"):
            assert np.array_equal(res[tag][key], res["peer1"][key]), (tag, key)
    assert str(res["peer1"]["transport"]) == "peer" and str(res["shm"]["transport"]) == "shm"


def test_channel_basin_closures_and_periodic_seam_distributed(tmp_path):
    """BASELINE configs[4] on 3 ranks (rehearsal transport): the x-periodic mesh, P1 buoyancy, full-stress A, BDF1 with the CFL
    step, the convection closure every step and the eddy closure's re-assembly of A at step 10 - each re-assembled as the
    replicated global matrix and gathered into the rank's row block - against the single-GPU run of the same model."""
    from nupgcm_amd import channel_basin
    world, nsteps = 3, 11
    arch = npg.GPU()
    mm = channel_basin.channel_basin_model(0.0625, workloads.CB_ALPHA)
    ref = workloads.channel_basin_model(arch, mesh_model=mm, element_precision="fp64")
    npg.run(ref, n_steps=nsteps)
    out = str(tmp_path / "cb")
    _launch(world, out, nsteps, "channel", "peer")
    ranks = [np.load(f"{out}.rank{k}.npz") for k in range(world)]
    for z in ranks[1:]:
        assert np.array_equal(z["b"], ranks[0]["b"]) and np.array_equal(z["u"], ranks[0]["u"]) and z["dt"] == ranks[0]["dt"]
    z = ranks[0]
    assert list(z["gm"]) == [1000] * nsteps                       # run.jl's itmax ends every inversion, on every rank count
    # (every inversion stops UNCONVERGED at the cap: the iterates, and with them the CFL step, depend on the summation order
    #  of the reductions at the 1e-3 level)
    assert abs(z["dt"] - ref.timestepper.dt) < 3e-3 * ref.timestepper.dt
    print("channel, 3 ranks vs one GPU: rel(b) =", rel(z["b"], ref.state.b), "rel(u) =", rel(z["u"], ref.state.u))
    assert rel(z["b"], ref.state.b) < 1e-3 and rel(z["u"], ref.state.u) < 1e-2, (rel(z["b"], ref.state.b), rel(z["u"], ref.state.u))
    # the periodic seam: with contiguous RCM row blocks on a non-periodic mesh the middle rank of three talks to its two
    # neighbours; here some rank also holds columns across the seam
    assert max(len(zz["peers"]) for zz in ranks) == 2 and all(len(zz["peers"]) >= 1 for zz in ranks)
