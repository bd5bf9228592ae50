"""Worker for tests/test_dist_gloo.py::test_record_exchange_*: one rank of a job that ASKS for RCCL on a box where RCCL cannot
come up (no GPU here; on a GPU box with fewer devices than ranks likewise).  The exchange must carry on over gloo, flagged."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import _lib as L  # noqa: E402
from amaranth_twstft_amd import collective, launch  # noqa: E402
from amaranth_twstft_amd import dist as D  # noqa: E402


def main():
    n_windows, want = int(sys.argv[1]), sys.argv[2]
    rank, _, world = launch.rank_world()
    ex = collective.RecordExchange(rank, world, want=want, reason=os.environ.get("TWX_COLLECTIVE_FALLBACK_REASON") or None).prepare()
    ex.bring_up(torch.device("cpu"))
    s, e = D.shard_windows(n_windows, rank, world)
    recs = (L.twx_result * max(e - s, 1))()
    for i, w in enumerate(range(s, e)):
        recs[i].indice0 = 3 * (1311765 - w)
        recs[i].status = rank
    local = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).view(-1, D.RESULT_BYTES)[: e - s].clone()
    allb = D.gather_results(local, n_windows, rank, world, exchange=ex)
    res = D.results_from_bytes(allb)
    ok = len(res) == n_windows and all(r.indice == 3 * (1311765 - w) for w, r in enumerate(res))
    tmax = ex.max_float(float(rank))
    agree = ex.all_true(ok)
    texts = ex.all_objects(ex.describe())
    if rank == 0:
        assert agree and tmax == world - 1 and len(set(texts)) == 1, (agree, tmax, texts)
        print("EXCHANGE_OK", n_windows, world, "|", ex.describe(), flush=True)
    ex.close()


if __name__ == "__main__":
    main()
