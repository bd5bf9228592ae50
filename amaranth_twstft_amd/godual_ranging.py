"""Script-level drop-in for processing/Octave/godual_ranging.m (file-in / delay-out contract).

    python -m amaranth_twstft_amd.godual_ranging [--datalocation DIR] [--codelocation DIR] [--remote 0|1] [--OP 0|1] [--gpus N]

Same flow as the reference script (godual_ranging.m:57-133): every capture ``1*.bin`` in
``datalocation`` (int16 ``[I1 Q1 I2 Q2]``) is correlated window by window against the code
``n*.bin`` picked by the parity of OP+remote (:60); one TSV row per window goes to stdout (:74,96,98),
the quadratic-fit residual statistics of :104-113 follow, and ``<capture>.mat`` (``remote<capture>.mat``
when remote=1) holds the result vectors (:126-131).  Already processed captures are skipped (the
idempotence rule of acquisition/claudio_aligned_code_ranging_separate.m:119).  Environment variables
``OP``, ``processing_dir``, ``codelocation`` are honoured like in the newer reference scripts (:12-25).
All sample arithmetic runs in libtwstft_hip.so.

``--gpus N`` (BASELINE.json configs[3]; the reference's analogue is three parallel octave processes,
acquisition/goprocess.sh:9-11): one process per GPU under torch.distributed.run, every rank correlates a
contiguous block of the capture's windows (one contiguous file extent: ``skip_samples = start*N``), the
fixed-size result records are exchanged with ONE all_gather (RCCL over xGMI) and rank 0 writes the same
``.mat``/TSV a single-GPU run writes — byte for byte, since every window is computed independently.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

from . import launch, prn, results_io
from .correlator import Correlator, band_godual


def _process_capture(cor, cap, band, all_channels, rank, world, exchange):
    """Records of every window of ``cap`` (all ranks return the full list; one collective)."""
    from . import dist as D
    nch = 2
    if hasattr(cor, "n_contexts"):            # --single-process: the library's own multi-GPU driver (twx_multi_*)
        allb = cor.process_file(cap, n_channels=nch, channel=-1 if all_channels else 0, band=band, raw_records=True)
        res = D.results_from_bytes(allb)
        return (res[0::2], res[1::2]) if all_channels else (res, None)
    nwin = os.path.getsize(cap) // (cor.n * 4 * nch)
    start, stop = D.shard_windows(nwin, rank, world)
    recs = cor.process_file(cap, n_channels=nch, channel=-1 if all_channels else 0, band=band,
                            skip_samples=start * cor.n, max_windows=stop - start, raw_records=True)
    per = nch if all_channels else 1
    if recs.shape[0] != (stop - start) * per:
        raise RuntimeError(f"rank {rank}: {recs.shape[0]} records for windows {start}..{stop} of {cap}")
    if world > 1:
        import torch
        local = torch.from_numpy(recs)
        if exchange.backend == "nccl":
            local = local.cuda()
        allb = D.gather_results(local, nwin, rank, world, per_window=per, exchange=exchange)
    else:
        allb = recs
    res = D.results_from_bytes(allb)
    if all_channels:
        return res[0::2], res[1::2]
    return res, None


def run(datalocation="./", codelocation="./codes/", remote=0, OP=0, fs=5e6, Nint=1, out=sys.stdout, device=-1,
        backend="nccl", single_process_gpus=0):
    rank, local_rank, world = launch.rank_world()
    if single_process_gpus:
        rank, local_rank, world = 0, 0, 1
    ex = None
    if world > 1:
        # the record exchange: gloo control plane + RCCL probe job BEFORE this rank touches its GPU, then the RCCL data plane with
        # its fall-back to gloo (collective.py) — a communicator that cannot be formed costs a line on stderr, not the capture
        from .collective import RecordExchange, pin_to_device
        ex = RecordExchange(rank, world, want=backend, reason=os.environ.get("TWX_COLLECTIVE_FALLBACK_REASON") or None).prepare()
        import torch
        if device < 0:
            device = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        ex.bring_up(torch.device("cuda", device))
        from . import _lib
        pin_to_device(_lib.load(), device)
        if rank == 0 and ex.fallback:
            sys.stderr.write(f"[godual_ranging] record exchange: {ex.describe()}\n")
    caps = sorted(glob.glob(os.path.join(datalocation, "1*.bin")))
    codes = sorted(glob.glob(os.path.join(codelocation, "n*.bin")) + glob.glob(os.path.join(codelocation, "n*.bin.gz")))
    if not codes:
        raise FileNotFoundError(f"no code file n*.bin in {codelocation}")
    codefile = codes[(OP + remote) % 2 % len(codes)]              # LTFB=odd OP=even (godual_ranging.m:60)
    chips = prn.read_code_file(codefile)
    done = []
    say = out.write if rank == 0 else (lambda s: None)
    if single_process_gpus:
        # one host process, N devices (threads + RCCL inside the library); a box with fewer GPUs repeats them, as the
        # multi-rank gloo tests do
        from .multi import MultiCorrelator
        try:
            import torch
            nvis = max(1, torch.cuda.device_count())
        except Exception:
            nvis = 1
        make = lambda: MultiCorrelator(chips, [i % nvis for i in range(single_process_gpus)], fs=fs, Nint=Nint, var_ddof=1)
    else:
        make = lambda: Correlator(chips, fs=fs, Nint=Nint, var_ddof=1, device=device)     # Octave var (N-1)
    with make() as cor:
        band = band_godual(fs, cor.n, remote=remote, OP=OP)
        for cap in caps:
            base = os.path.basename(cap)
            nom = os.path.join(datalocation, ("remote" if remote == 1 else "") + base.replace(".bin", ".mat"))
            if os.path.exists(nom) or os.path.exists(nom + ".gz"):
                say(f"{nom} already done\n")
                continue
            say(base + "\n")
            # both channels of every window from one pass over the file (:91,95); remote: measurement channel only
            r1, r2 = _process_capture(cor, cap, band, remote != 1, rank, world, ex)
            if rank == 0:
                for row in results_io.tsv_rows(r1, r2, fs, Nint):
                    say(row)
                for line in results_io.residual_report(r1, r2, fs, Nint):
                    say(line)
                results_io.save_mat(nom, r1, r2, code=prn.chips_to_code(chips), remote=remote)
            if ex is not None:
                ex.barrier()              # the .mat exists before any rank looks at the next capture's "already done"
            done.append(nom)
    if ex is not None:
        ex.close()
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--datalocation", default=os.environ.get("processing_dir", "./"))
    ap.add_argument("--codelocation", default=os.environ.get("codelocation", "./codes/"))
    ap.add_argument("--remote", type=int, default=0)
    ap.add_argument("--OP", type=int, default=int(os.environ.get("OP", "0") or 0))
    ap.add_argument("--fs", type=float, default=5e6)
    ap.add_argument("--Nint", type=int, default=1)
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU) sharing every capture's windows")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) or gloo (several ranks on one GPU, tests)")
    ap.add_argument("--single-process", action="store_true", help="--gpus N inside THIS process: the library's twx_multi driver "
                    "(one context + host thread per device, RCCL gather) instead of N ranks under torch.distributed.run")
    a = ap.parse_args(argv)
    if a.single_process:
        run(a.datalocation, a.codelocation, a.remote, a.OP, a.fs, a.Nint, single_process_gpus=max(1, a.gpus))
        return
    if a.gpus > 1 and not launch.is_rank():
        args = list(sys.argv[1:] if argv is None else argv)
        sys.exit(launch.spawn_with_fallback(a.gpus, "", args, module="amaranth_twstft_amd.godual_ranging", backend=a.backend))
    run(a.datalocation, a.codelocation, a.remote, a.OP, a.fs, a.Nint, backend=a.backend)


if __name__ == "__main__":
    main()
