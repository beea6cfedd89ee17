"""The product transport (RCCL) on a one-GPU box: a one-rank communicator made to go through every RCCL call site
(tests/rccl_selftest_worker.py, NPG_COMM_SELFTEST=1)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_call_sites_run_on_a_one_rank_communicator():
    env = dict(os.environ, NPG_COMM_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NPG_HALO_OVERLAP_VERBOSE="1")
    env.pop("NPG_COMM_TRANSPORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_selftest_worker.py")], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL self-test OK" in r.stdout
    assert "halo overlap on" in r.stderr          # the split cycle really took the two-stream path
