"""The record exchange of a one-process-per-GPU job, built so that it cannot lose the job.

The only exchange of the path is one all_gather of 240-byte result records per capture (the reference's analogue
is the hand-off between worker threads, processing/CPP/main.cpp:180-187,488-497, and three jobs side by side,
acquisition/goprocess.sh:9-11).  RCCL over xGMI carries it when RCCL can be brought up — but a communicator that cannot be formed
(IPC handles, a dead link, a driver setting) must cost a flag in the output, not the run:

* control plane: a ``gloo`` process group over 127.0.0.1, created first and before anything touches the GPU; barriers, the
  max-over-ranks of the timing and every agreement below go over it;
* RCCL probe: rank 0, which has not touched a GPU yet, starts a FRESH child job of ``world`` ranks
  (``python -m amaranth_twstft_amd.collective --probe``) that forms the communicator and runs an all_reduce and an all_gather, under a
  time limit, first with the environment as it is and then with ``HSA_ENABLE_IPC_MODE_LEGACY`` toggled; the verdict (and the
  environment that worked) is broadcast over gloo and applied by every rank before its first HIP call.  No process that has
  initialised the GPU is ever replaced or re-executed; a probe that hangs is killed by its own process group id;
* data plane: a ``nccl`` (= RCCL) sub-group, brought up with an asynchronous all_reduce and a rehearsal of the real gather, each
  with a deadline, each followed by an agreement (gloo MIN) so that all ranks take the same branch.  Any failure — probe,
  bring-up, rehearsal — turns the data plane to gloo and is reported as ``backend: "gloo (fallback: <text>)"``.

``TWX_INJECT_RCCL_FAIL`` = ``probe`` | ``probe_child:<rank>`` | ``init:<rank>`` | ``rehearsal:<rank>`` makes the named step fail
on purpose (a one-GPU box has no other way into these branches; two ranks on one device already fail the probe by themselves);
``exit:<rank>`` ends that rank with status 3 when RCCL was asked for — the whole job then fails, which is what the launcher's own
second line of defence (``launch.spawn_with_fallback``: one fresh job with ``--backend gloo``) is there for.
"""
from __future__ import annotations

import datetime
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE_OK = "TWX_RCCL_PROBE_OK"
KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"      # (module attributes: the CPU tests point them at a fake tree)
DRI_DIR = "/dev/dri"


def _inject(step: str, rank: int | None = None) -> bool:
    v = os.environ.get("TWX_INJECT_RCCL_FAIL", "")
    if rank is None:
        return v == step
    return v == f"{step}:{rank}"


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_list(n_physical: int) -> int | None:
    """How many devices the *_VISIBLE_DEVICES variables leave of ``n_physical``; None when none of them is set."""
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(name)
        if v is None:
            continue
        items = [x.strip() for x in v.split(",") if x.strip() != ""]
        n = 0
        for x in items:                                  # ordinals beyond the machine (or -1) end the list, as the runtime reads it
            if x.lstrip("-").isdigit() and not (0 <= int(x) < n_physical):
                break
            n += 1
        return min(n, n_physical)
    return None


def visible_gpus() -> int:
    """Device count WITHOUT loading HIP or torch into this process: the GPU nodes of /sys/class/kfd/kfd/topology (nodes with SIMDs),
    narrowed by ROCR_/HIP_/CUDA_VISIBLE_DEVICES.  The probe's verdict — the environment RCCL came up with — is applied before the first
    HIP call of every rank; a count taken through the runtime would have initialised HSA in rank 0 only, with the OLD environment.  Where
    the topology cannot be read the count comes from torch.cuda.device_count(), which must then have left HIP uninitialised."""
    base = KFD_NODES
    try:
        n = 0
        for node in sorted(os.listdir(base)):
            try:
                with open(os.path.join(base, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) <= 0:
                continue                                     # a CPU node
            # a GPU of the machine that this container may not open (a slice of an 8-GPU node) is not ours
            minor = int(props.get("drm_render_minor", "0"))
            if minor > 0 and DRI_DIR and not os.access(os.path.join(DRI_DIR, "renderD%d" % minor), os.R_OK | os.W_OK):
                continue
            n += 1
        if n > 0:
            lim = _visible_list(n)
            return n if lim is None else lim
    except OSError:
        pass
    import torch
    n = torch.cuda.device_count()
    assert not torch.cuda.is_initialized(), "counting the devices initialised HIP: the RCCL probe's environment could no longer be applied"
    return n


def _run_group(cmd, env, timeout):
    """Child job in its own session; on a time-out the whole process group (torchrun + its ranks) is killed by ITS id."""
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True, cwd=ROOT)
    try:
        out, _ = p.communicate(timeout=timeout)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        return -9, (out or "") + f"\n[killed after {timeout:.0f} s]"


def _short(text: str, n: int = 300) -> str:
    """The informative tail of a child's output: the last lines that carry an error, squeezed."""
    lines = [l.strip() for l in (text or "").splitlines() if l.strip()]
    keep = [l for l in lines if any(k in l for k in ("Error", "error", "failed", "FAILED", "invalid", "Duplicate", "killed", "Traceback", "NCCL", "RCCL"))]
    s = " | ".join((keep or lines)[-4:])
    return s[-n:]


def probe_rccl(world: int, timeout: float | None = None) -> dict:
    """Can ``world`` fresh processes, one per GPU, form an RCCL communicator here?  Called by a process that has NOT touched the GPU.
    Returns {"ok": bool, "env": {name: value-or-None to apply before HIP initialises}, "tried": [...], "error": text, "seconds": s}."""
    t0 = time.perf_counter()
    timeout = float(os.environ.get("TWX_RCCL_PROBE_TIMEOUT_S", "120")) if timeout is None else timeout
    if _inject("probe"):
        return {"ok": False, "env": {}, "tried": [], "error": "injected failure (TWX_INJECT_RCCL_FAIL=probe)", "seconds": 0.0}
    ndev = visible_gpus()
    # (TWX_RCCL_PROBE_SHARE=1: start the probe job anyway, its ranks sharing devices — RCCL then refuses by itself, which is how a
    # one-GPU box gets a REAL RCCL failure through the child-job path)
    if world > ndev and not (ndev >= 1 and os.environ.get("TWX_RCCL_PROBE_SHARE") == "1"):
        return {"ok": False, "env": {}, "tried": [], "seconds": 0.0,
                "error": f"{world} ranks but {ndev} GPU(s) visible: RCCL needs one device per rank"}
    cur = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    variants = [("as is (HSA_ENABLE_IPC_MODE_LEGACY=%s)" % ("unset" if cur is None else cur), {})]
    if os.environ.get("TWX_RCCL_PROBE_VARIANTS", "1") != "0":
        variants.append(("HSA_ENABLE_IPC_MODE_LEGACY unset", {"HSA_ENABLE_IPC_MODE_LEGACY": None}) if cur is not None
                        else ("HSA_ENABLE_IPC_MODE_LEGACY=0", {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}))
    tried, errors = [], []
    for name, delta in variants:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                 "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "GROUP_WORLD_SIZE",
                                                                 "ROLE_NAME", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
                                                                 "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING", "TORCHELASTIC_ERROR_FILE")}
        for k, v in delta.items():
            if v is None:
                env.pop(k, None)
            else:
                env[k] = v
        env["MASTER_ADDR"] = "127.0.0.1"
        env["PYTHONPATH"] = ROOT + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", "amaranth_twstft_amd.collective", "--probe"]
        rc, out = _run_group(cmd, env, timeout)
        tried.append(name)
        if rc == 0 and out.count(PROBE_OK) >= 1:
            return {"ok": True, "env": delta, "tried": tried, "error": "", "seconds": round(time.perf_counter() - t0, 1)}
        errors.append(f"{name}: rc {rc}: {_short(out)}")
    return {"ok": False, "env": {}, "tried": tried, "error": "; ".join(errors)[-600:], "seconds": round(time.perf_counter() - t0, 1)}


def _probe_child():
    """One rank of the probe job: communicator + all_reduce + all_gather over RCCL, nothing else."""
    import torch
    import torch.distributed as dist
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    if _inject("probe_child", rank):
        sys.stderr.write(f"probe rank {rank}: injected failure (TWX_INJECT_RCCL_FAIL=probe_child:{rank})\n")
        sys.exit(3)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=90))
    x = torch.full((1,), float(rank + 1), device=dev)
    dist.all_reduce(x)
    rec = torch.full((600, 240), rank, dtype=torch.uint8, device=dev)               # the size of the real exchange
    allr = torch.empty((world * 600, 240), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(allr, rec)
    torch.cuda.synchronize()
    ok = float(x.item()) == world * (world + 1) / 2 and all(int(allr[r * 600, 0]) == r for r in range(world))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.stderr.write(f"probe rank {rank}: collectives returned wrong data\n")
        sys.exit(4)
    if rank == 0:
        print(PROBE_OK, flush=True)


class RecordExchange:
    """Control plane (gloo) + data plane (RCCL, or gloo when RCCL cannot be brought up) of one rank.

    Create it BEFORE the first HIP call of the process (``prepare``), choose the device, then ``bring_up(device)``.
    ``want``: "nccl" (RCCL with fall-back) or "gloo" (tests: several ranks on one GPU)."""

    def __init__(self, rank: int, world: int, want: str = "nccl", reason: str | None = None):
        self.rank, self.world, self.want = rank, world, want
        self.backend = "none" if world == 1 else "gloo"
        self.fallback: str | None = reason          # why the data plane is not RCCL although it was asked for
        self.probe: dict | None = None
        self._ctl = None
        self._data = None
        self.device = None

    # ---- step 1: control plane and the probe; nothing here touches the GPU
    def prepare(self, force: bool = False):
        if self.world == 1 and not force:
            return self
        if self.want == "nccl" and _inject("exit", self.rank):
            from . import launch
            launch.mark_collective(self.rank, True)        # stands for a rank killed inside RCCL: the mark stays behind
            sys.stderr.write(f"rank {self.rank}: injected job failure (TWX_INJECT_RCCL_FAIL=exit:{self.rank})\n")
            sys.exit(3)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=datetime.timedelta(minutes=30))
        self._ctl = dist.group.WORLD
        if self.want == "nccl" and self.fallback is None:
            verdict = [None]
            if self.rank == 0:
                verdict[0] = probe_rccl(self.world) if (self.world > 1 and os.environ.get("TWX_RCCL_PROBE", "1") != "0") \
                    else {"ok": True, "env": {}, "tried": [], "error": "", "seconds": 0.0, "skipped": True}
            dist.broadcast_object_list(verdict, src=0)
            self.probe = verdict[0]
            if not self.probe["ok"]:
                self.fallback = "RCCL probe job failed: " + self.probe["error"]
            else:
                for k, v in self.probe["env"].items():              # the environment the probe found working, before HIP initialises
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return self

    def _agree(self, ok: bool) -> bool:
        import torch
        import torch.distributed as dist
        f = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=self._ctl)
        return bool(int(f.item()) == 1)

    def _gather_texts(self, text: str) -> str:
        import torch.distributed as dist
        texts = [None] * self.world
        dist.all_gather_object(texts, text, group=self._ctl)
        return "; ".join(f"rank {r}: {t}" for r, t in enumerate(texts) if t)[:600]

    def _wait(self, work, seconds: float) -> bool:
        t0 = time.perf_counter()
        while not work.is_completed():
            if time.perf_counter() - t0 > seconds:
                return False
            time.sleep(0.002)
        return True

    def _drop_data_group(self):
        pg, self._data = self._data, None
        if pg is None:
            return
        try:                                           # a communicator with a collective that never completed is aborted, not destroyed
            be = pg._get_backend(self.device)
            (getattr(be, "abort", None) or getattr(be, "_abort", None) or (lambda: None))()
        except Exception:
            pass

    # ---- step 2: the data plane, after the rank has chosen its device
    def bring_up(self, device, rehearsal=None, limit: float | None = None):
        """``device``: torch.device of this rank.  ``rehearsal``: (gathered, local) device tensors of the real exchange, run once."""
        self.device = device
        if self._ctl is None:
            return self
        import torch
        import torch.distributed as dist
        limit = float(os.environ.get("TWX_RCCL_INIT_TIMEOUT_S", "120")) if limit is None else limit
        if self.want == "nccl" and self.fallback is None:
            from . import launch
            err = ""
            launch.mark_collective(self.rank, True)        # (a process that dies in here is what the launcher's restart is for)
            try:
                # (new_group is collective over the control plane: every rank enters it, whatever happens next)
                self._data = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(limit, 30.0)), device_id=device)
                if _inject("init", self.rank):
                    raise RuntimeError(f"injected failure (TWX_INJECT_RCCL_FAIL=init:{self.rank})")
                w = dist.all_reduce(torch.ones(1, device=device), group=self._data, async_op=True)
                if not self._wait(w, limit):
                    raise TimeoutError(f"first RCCL all_reduce did not complete within {limit:.0f} s")
            except Exception as e:                      # noqa: BLE001 - whatever RCCL raises ends up in the output line
                err = f"{type(e).__name__}: {e}"[:300]
            ok = self._agree(not err)
            if ok and rehearsal is not None:
                try:
                    if _inject("rehearsal", self.rank):
                        raise RuntimeError(f"injected failure (TWX_INJECT_RCCL_FAIL=rehearsal:{self.rank})")
                    w = dist.all_gather_into_tensor(rehearsal[0], rehearsal[1], group=self._data, async_op=True)
                    if not self._wait(w, limit):
                        raise TimeoutError(f"rehearsal all_gather did not complete within {limit:.0f} s")
                except Exception as e:                  # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"[:300]
                ok = self._agree(not err)
            launch.mark_collective(self.rank, False)
            if ok:
                self.backend = "nccl"
            else:
                self.fallback = "RCCL bring-up failed: " + (self._gather_texts(err) or "a peer failed")
                self._drop_data_group()
        return self

    # ---- use
    @property
    def active(self) -> bool:
        return self._ctl is not None

    def describe(self) -> str:
        if self.backend == "nccl":
            return "nccl (RCCL)"
        if self.fallback:
            return f"gloo (fallback: {self.fallback})"
        return self.backend

    def all_gather_records(self, gathered, local):
        """``gathered`` [world*n, B] and ``local`` [n, B] uint8 tensors on this rank's device; returns when ``gathered`` is complete."""
        import torch
        import torch.distributed as dist
        if self.backend == "nccl":
            from . import launch
            launch.mark_collective(self.rank, True)
            dist.all_gather_into_tensor(gathered, local, group=self._data)
            torch.cuda.current_stream().synchronize()
            launch.mark_collective(self.rank, False)
        else:
            host = torch.empty(gathered.shape, dtype=gathered.dtype)
            dist.all_gather_into_tensor(host, local.cpu(), group=self._ctl)
            gathered.copy_(host)

    def barrier(self):
        if self._ctl is not None:
            import torch.distributed as dist
            dist.barrier(group=self._ctl)

    def max_float(self, x: float) -> float:
        if self._ctl is None:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._ctl)
        return float(t.item())

    def all_true(self, ok: bool) -> bool:
        return ok if self._ctl is None else self._agree(ok)

    def all_objects(self, obj):
        if self._ctl is None:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self._ctl)
        return out

    def close(self):
        if self._ctl is None:
            return
        import torch.distributed as dist
        try:
            dist.barrier(group=self._ctl)
        finally:
            dist.destroy_process_group()
            self._ctl = self._data = None


def pin_to_device(lib, device_index: int) -> dict:
    """Bind this rank to the CPUs of its GPU's NUMA node (twx_pin_thread_to_device); reports what was done."""
    import ctypes as C
    node, ncpu = C.c_int32(-1), C.c_int32(0)
    rc = lib.twx_pin_thread_to_device(int(device_index), C.byref(node), C.byref(ncpu))
    return {"numa_node": int(node.value), "cpus_bound": int(ncpu.value), "ok": rc == 0}


if __name__ == "__main__":
    if "--probe" in sys.argv:
        _probe_child()
