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


def test_launcher_does_not_restart_a_job_that_failed_outside_the_exchange(tmp_path):
    """A job whose rank fails for a reason of its own (a bad argument, an assertion: no mark of an RCCL step left behind, no signal) is NOT
    run a second time: its status comes back unchanged and the side effects happen once (advisor, round 5)."""
    script = tmp_path / "job.py"
    script.write_text(
        "import argparse, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from amaranth_twstft_amd import launch\n"
        "ap = argparse.ArgumentParser(); ap.add_argument('--backend', default='nccl'); a = ap.parse_args()\n"
        "if not launch.is_rank():\n"
        "    sys.exit(launch.spawn_with_fallback(2, os.path.abspath(__file__), sys.argv[1:], backend=a.backend))\n"
        "rank, _, world = launch.rank_world()\n"
        "open(os.path.join(%r, 'ran_%%d_%%d' %% (rank, os.getpid())), 'w').close()\n"
        "sys.exit(7 if rank == 1 else 0)\n" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "not restarted" in out.stderr and "starting it once more" not in out.stderr
    assert len([f for f in os.listdir(tmp_path) if f.startswith("ran_")]) == 2         # two ranks, once


def test_visible_gpus_counts_from_the_kfd_topology_without_hip(tmp_path, monkeypatch):
    """collective.visible_gpus(): the GPU nodes of the kfd topology (nodes with SIMDs) whose render node this process may open, narrowed by
    *_VISIBLE_DEVICES — no torch, no HIP in the counting process (advisor, round 5: rank 0 must not initialise HSA before the probe's
    environment is applied)."""
    from amaranth_twstft_amd import collective
    nodes, dri = tmp_path / "nodes", tmp_path / "dri"
    dri.mkdir()
    for i, (simd, minor) in enumerate([(0, 0), (0, 0), (1024, 128), (1024, 129), (1024, 130)]):     # two CPU nodes, three GPUs
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if simd else 64, simd, minor))
    for m in (128, 129):                                                                 # the third GPU's render node is not ours
        (dri / ("renderD%d" % m)).write_text("")
    monkeypatch.setattr(collective, "KFD_NODES", str(nodes))
    monkeypatch.setattr(collective, "DRI_DIR", str(dri))
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert collective.visible_gpus() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert collective.visible_gpus() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,5")
    assert collective.visible_gpus() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert collective.visible_gpus() == 0
