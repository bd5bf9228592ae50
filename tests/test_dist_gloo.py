"""CPU: the multi-GPU path (window sharding + one all_gather of result records) with gloo, world 2."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_windows", [7, 600])
def test_shard_and_gather_world2(n_windows):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29511 + n_windows % 50), os.path.join(ROOT, "tests", "dist_worker.py"), str(n_windows)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert f"GATHER_OK {n_windows} 2" in out.stdout
