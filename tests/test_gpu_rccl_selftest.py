"""The product transports on a one-GPU box: a one-rank communicator made to go through every RCCL call site, and through the
peer-window kernels with the rank as its own neighbour (tests/rccl_selftest_worker.py, NPG_COMM_SELFTEST=1)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_comm_call_sites_run_on_a_one_rank_communicator(transport):
    env = dict(os.environ, NPG_COMM_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NPG_HALO_OVERLAP_VERBOSE="1",
               NPG_PEER_TIMEOUT_S="20")
    env.pop("NPG_COMM_TRANSPORT", None)
    if transport == "peer":
        env["NPG_COMM_TRANSPORT"] = "peer"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_selftest_worker.py")], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"{transport.upper()} self-test OK" in r.stdout
    assert "halo overlap on" in r.stderr          # the split cycle really took the two-stream path
