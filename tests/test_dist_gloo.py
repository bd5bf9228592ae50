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


def _run_exchange(n_windows, want, extra_env=None, port=29571):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "exchange_worker.py"), str(n_windows), want]
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)


def test_record_exchange_asked_for_rccl_falls_back_to_gloo():
    """Two ranks ask for RCCL where it cannot come up (this container has no GPU): rank 0's probe says so before anything touches a
    device, the verdict travels over the gloo control plane, every rank takes the gloo data plane and the output names the reason."""
    out = _run_exchange(11, "nccl")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("EXCHANGE_OK")][-1]
    assert "EXCHANGE_OK 11 2 | gloo (fallback: RCCL probe job failed:" in line and "GPU(s) visible" in line


def test_record_exchange_injected_probe_failure_and_launcher_reason():
    out = _run_exchange(5, "nccl", {"TWX_INJECT_RCCL_FAIL": "probe"}, port=29572)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "gloo (fallback: RCCL probe job failed: injected failure" in out.stdout
    # the launcher's second job: --backend gloo with the reason handed down in the environment
    out = _run_exchange(5, "gloo", {"TWX_COLLECTIVE_FALLBACK_REASON": "the RCCL job ended with status 3; restarted by the launcher"}, port=29573)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "| gloo (fallback: the RCCL job ended with status 3; restarted by the launcher)" in out.stdout
    out = _run_exchange(5, "gloo", port=29574)
    assert out.returncode == 0 and out.stdout.strip().endswith("| gloo")


def test_launcher_restarts_a_failed_rccl_job_on_gloo(tmp_path):
    """launch.spawn_with_fallback: the first job (RCCL asked for) loses a rank -> status != 0 -> ONE fresh job with --backend gloo."""
    script = tmp_path / "job.py"
    script.write_text(
        "import argparse, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "from amaranth_twstft_amd import collective, launch\n"
        "ap = argparse.ArgumentParser(); ap.add_argument('--backend', default='nccl'); a = ap.parse_args()\n"
        "if not launch.is_rank():\n"
        "    sys.exit(launch.spawn_with_fallback(2, os.path.abspath(__file__), sys.argv[1:], backend=a.backend))\n"
        "rank, _, world = launch.rank_world()\n"
        "ex = collective.RecordExchange(rank, world, want=a.backend, reason=os.environ.get('TWX_COLLECTIVE_FALLBACK_REASON') or None).prepare()\n"
        "ex.bring_up(torch.device('cpu'))\n"
        "if rank == 0: print('JOB_OK', ex.describe(), flush=True)\n"
        "ex.close()\n" % ROOT)
    env = dict(os.environ, TWX_INJECT_RCCL_FAIL="exit:1")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "JOB_OK gloo (fallback: the RCCL job ended with status" in out.stdout
    assert "starting it once more with the record exchange on gloo" in out.stderr
