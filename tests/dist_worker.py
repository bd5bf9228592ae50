"""Worker for tests/test_dist_gloo.py: run under torch.distributed.run with the gloo backend.

Each rank fabricates the result records of its shard (deterministic function of the global window
index), gathers them with amaranth_twstft_amd.dist.gather_results and rank 0 checks the order."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import _lib as L  # noqa: E402
from amaranth_twstft_amd import dist as D  # noqa: E402


def main():
    n_windows = int(sys.argv[1])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    s, e = D.shard_windows(n_windows, rank, world)
    recs = (L.twx_result * max(e - s, 1))()
    for i, w in enumerate(range(s, e)):
        recs[i].indice0 = 3 * (1311765 - w)
        recs[i].correction = w * 1e-3
        recs[i].df = 1780.75 + w
        recs[i].status = rank
    local = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).view(-1, D.RESULT_BYTES)[: e - s].clone()
    allb = D.gather_results(local, n_windows, rank, world)
    if rank == 0:
        res = D.results_from_bytes(allb)
        assert len(res) == n_windows
        for w, r in enumerate(res):
            assert r.indice == 3 * (1311765 - w) and abs(r.df - (1780.75 + w)) < 1e-12 and abs(r.correction - w * 1e-3) < 1e-15
        print("GATHER_OK", n_windows, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
