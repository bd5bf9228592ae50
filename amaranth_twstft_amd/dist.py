"""Window sharding across GPUs and the final gather of per-window results.

The fixed-window contract (processing/Octave/godual_ranging.m:75-102) makes every 1-s window
independent: each rank (one process per GPU) takes a contiguous block of windows — one
contiguous extent of the capture file — computes the code spectrum redundantly, and the only
exchange is one all_gather of the fixed-size ``twx_result`` records (RCCL over xGMI when the
tensors live on GPUs, gloo on CPU in the tests).  There is no data-path collective.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

RESULT_BYTES = C.sizeof(L.twx_result)


def shard_windows(n_windows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [start, stop) of windows for ``rank``; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(n_windows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_shard(n_windows: int, world: int) -> int:
    return -(-n_windows // world)


def gather_results(local, n_windows: int, rank: int, world: int, device=None, per_window: int = 1, exchange=None):
    """all_gather the local result records → numpy structured bytes for all ``n_windows``.

    ``local``: torch uint8 tensor [n_local*per_window, RESULT_BYTES] (on the rank's GPU for RCCL, on CPU
    for gloo); ``per_window`` records belong to one window (the channels of the all-channel mode).
    Shards are padded to a common length so a single ``all_gather_into_tensor`` suffices
    (≈ 240 B × windows: latency-bound, one collective per capture).  ``exchange``: a :class:`collective.RecordExchange`
    (RCCL with its gloo fall-back) to carry it; without one the default process group is used as it is.
    """
    import torch
    import torch.distributed as dist
    cap = max_shard(n_windows, world) * per_window
    pad = torch.zeros((cap, RESULT_BYTES), dtype=torch.uint8, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        out = pad.unsqueeze(0)
    else:
        flat = torch.empty((world * cap, RESULT_BYTES), dtype=torch.uint8, device=local.device)
        if exchange is not None:
            exchange.all_gather_records(flat, pad)
        else:
            dist.all_gather_into_tensor(flat, pad)
        out = flat.view(world, cap, RESULT_BYTES)
    out = out.cpu().numpy()
    pieces = []
    for r in range(world):
        s, e = shard_windows(n_windows, r, world)
        pieces.append(out[r, : (e - s) * per_window])
    return np.concatenate(pieces, axis=0)


def results_from_bytes(buf: np.ndarray):
    """[n, RESULT_BYTES] uint8 → list of correlator.WindowResult."""
    from .correlator import _to_result
    n = buf.shape[0]
    arr = (L.twx_result * n).from_buffer_copy(np.ascontiguousarray(buf).tobytes())
    return [_to_result(arr[i]) for i in range(n)]
