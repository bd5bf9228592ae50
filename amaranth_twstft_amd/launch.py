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


def spawn_with_fallback(n: int, script: str, argv: list[str], module: str | None = None, backend: str = "nccl", timeout: float | None = None) -> int:
    """:func:`spawn_ranks`, and when that job ends with a non-zero status although its ranks already fall back by themselves
    (amaranth_twstft_amd/collective.py) — a rank killed inside RCCL, a bootstrap that aborted the process — ONE more job, fresh, with
    ``--backend gloo`` and the reason in ``TWX_COLLECTIVE_FALLBACK_REASON`` (the output then reads ``gloo (fallback: ...)``).
    This launcher never touches the GPU; no rank is ever re-executed, the second job is all new processes."""
    rc = spawn_ranks(n, script, argv, module=module, timeout=timeout)
    if rc == 0 or backend != "nccl" or os.environ.get("TWX_NO_JOB_FALLBACK", "0") != "0":
        return rc
    sys.stderr.write(f"[launch] the {n}-rank job ended with status {rc}; starting it once more with the record exchange on gloo\n")
    args, skip = [], False
    for x in argv:                                  # drop a --backend the caller gave, in either spelling
        if skip:
            skip = False
        elif x == "--backend":
            skip = True
        elif not x.startswith("--backend="):
            args.append(x)
    os.environ["TWX_COLLECTIVE_FALLBACK_REASON"] = f"the RCCL job ended with status {rc}; restarted by the launcher"
    try:
        return spawn_ranks(n, script, args + ["--backend", "gloo"], module=module, timeout=timeout)
    finally:
        os.environ.pop("TWX_COLLECTIVE_FALLBACK_REASON", None)
