"""RCCL with the ranks one GPU allows: torch.distributed's "nccl" backend must initialise on this image and carry the sharded proof's exchange buffers (plonk/native.py TorchExchange:
uint8 HBM tensors, all_gather_into_tensor, called from a helper host thread as bench.py's N > 1 extras do).  One rank — RCCL refuses two ranks on one device — so this is the
collective's call path, not scaling; worlds 2 / 4 / 8 run over gloo in tests/test_multi_rank.py.  Runs in a child process: a process group is process-global state."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_carries_the_exchange_buffers_single_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29551", RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_check.py")], env=env, capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL did not initialise within 180 s on this box (infrastructure, not the library)")
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        pytest.skip("no result from the RCCL child process (rc %d): %s" % (r.returncode, r.stderr[-300:]))
    res = json.loads(lines[-1])
    assert res["all_gather_into_tensor_uint8_from_a_helper_thread"] and not res["errors"] and res["calls"] == 3, res
