"""One process per GPU: start the ranks of a multi-GPU job from a plain ``python script.py --gpus N`` call.

The reference runs its three correlators as three background ``octave`` processes
(acquisition/goprocess.sh:9-11); here a job over N GPUs is N ranks under ``torch.distributed.run``
(RCCL over xGMI for the final gather of result records).  A caller that is not already a rank
(no ``RANK`` in the environment) re-launches itself through :func:`spawn_ranks` BEFORE it touches
the GPU — a process that has initialised HIP must never be replaced or re-exec'ed — relays the
children's output and exits with their return code.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def is_rank() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) outside it."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, script: str, argv: list[str], module: str | None = None, timeout: float | None = None) -> int:
    """Run ``script argv`` (or ``-m module argv``) as ``n`` ranks on this node; returns the launcher's exit code.
    stdout/stderr of the ranks pass straight through (rank 0 prints the job's output)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC, needed by RCCL across processes on this driver
    env["MASTER_ADDR"] = "127.0.0.1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", module] if module else [script]) + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode
