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


MARKER_ENV = "TWX_COLLECTIVE_MARKER"       # a directory: rank r keeps the file `rank<r>` there while it is inside an RCCL bring-up or collective


def mark_collective(rank: int, inside: bool) -> None:
    """Called by ``collective.RecordExchange`` around every step that can take the process down inside RCCL (bootstrap, first
    collective, rehearsal): the launcher restarts a failed job on gloo ONLY when such a mark was left behind."""
    d = os.environ.get(MARKER_ENV)
    if not d:
        return
    path = os.path.join(d, f"rank{rank}")
    try:
        if inside:
            with open(path, "w") as f:
                f.write("inside RCCL\n")
        elif os.path.exists(path):
            os.remove(path)
    except OSError:
        pass


def spawn_with_fallback(n: int, script: str, argv: list[str], module: str | None = None, backend: str = "nccl", timeout: float | None = None) -> int:
    """:func:`spawn_ranks`, and ONE more job, fresh, with ``--backend gloo`` when the first one died INSIDE RCCL although its ranks
    already fall back by themselves (amaranth_twstft_amd/collective.py): a rank that left its mark in ``TWX_COLLECTIVE_MARKER`` (it
    entered an RCCL bring-up step or collective and never left it — killed there, or the bootstrap aborted the process), or a job that
    ended on a signal.  Any other failure — bad arguments, a missing capture, an assertion, out of memory — has nothing to do with the
    exchange: its status is returned as it is, nothing runs twice.  The second job's output reads ``gloo (fallback: ...)`` through
    ``TWX_COLLECTIVE_FALLBACK_REASON``.  This launcher never touches the GPU; no rank is ever re-executed, the second job is all new
    processes."""
    import shutil
    import tempfile
    if backend != "nccl" or os.environ.get("TWX_NO_JOB_FALLBACK", "0") != "0":
        return spawn_ranks(n, script, argv, module=module, timeout=timeout)
    marks = tempfile.mkdtemp(prefix="twx_rccl_marks_")
    os.environ[MARKER_ENV] = marks
    try:
        rc = spawn_ranks(n, script, argv, module=module, timeout=timeout)
        if rc == 0:
            return rc
        left = sorted(os.listdir(marks))
        by_signal = rc < 0 or rc > 128
        if not left and not by_signal:
            sys.stderr.write(f"[launch] the {n}-rank job ended with status {rc} outside the record exchange: not restarted\n")
            return rc
        why = (f"{', '.join(left)} died inside RCCL" if left else f"the job ended on signal {-rc if rc < 0 else rc - 128}")
    finally:
        os.environ.pop(MARKER_ENV, None)
        shutil.rmtree(marks, ignore_errors=True)
    sys.stderr.write(f"[launch] the {n}-rank job ended with status {rc} ({why}); starting it once more with the record exchange on gloo\n")
    args, skip = [], False
    for x in argv:                                  # drop a --backend the caller gave, in either spelling
        if skip:
            skip = False
        elif x == "--backend":
            skip = True
        elif not x.startswith("--backend="):
            args.append(x)
    os.environ["TWX_COLLECTIVE_FALLBACK_REASON"] = f"the RCCL job ended with status {rc} ({why}); restarted by the launcher"
    try:
        return spawn_ranks(n, script, args + ["--backend", "gloo"], module=module, timeout=timeout)
    finally:
        os.environ.pop("TWX_COLLECTIVE_FALLBACK_REASON", None)
