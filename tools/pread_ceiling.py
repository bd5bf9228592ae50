#!/usr/bin/env python3
"""What the host gives a reader of a page-cache (tmpfs) file, no GPU and no library involved: N threads, each preadv-ing its share of a
3.84-GB file (192 windows of 5e6 int16 IQ samples) in 8-MB pieces into its own buffer — the ceiling of twx_process_file's ingest —
beside plain memcpy of the same bytes between two user buffers (what the host-buffer entry twx_process_windows does).
    python tools/pread_ceiling.py [threads ...]"""
import os, sys, tempfile, threading, time
import numpy as np
threads = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]
SIZE = 192 * 20_000_000
PIECE = 8 << 20
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    path = os.path.join(td, "cap.bin")
    blk = np.random.default_rng(1).integers(-400, 400, 20_000_000 // 2, dtype=np.int16).tobytes()
    with open(path, "wb") as f:
        for _ in range(192):
            f.write(blk)
    fd = os.open(path, os.O_RDONLY)
    src = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)          # the same bytes in user memory
    for n in threads:
        share = (SIZE // n + 4095) & ~4095
        def reader(i, how):
            buf = bytearray(PIECE); mv = memoryview(buf); dst = np.frombuffer(buf, dtype=np.uint8)
            lo, hi = i * share, min(SIZE, (i + 1) * share)
            for off in range(lo, hi, PIECE):
                k = min(PIECE, hi - off)
                if how == "pread":
                    os.preadv(fd, [mv[:k]], off)
                else:
                    np.copyto(dst[:k], src[off:off + k])
        for how in ("pread", "memcpy"):
            best = 0.0
            for rep in range(2):
                ts = [threading.Thread(target=reader, args=(i, how)) for i in range(n)]
                t0 = time.perf_counter()
                for t in ts: t.start()
                for t in ts: t.join()
                best = max(best, SIZE / (time.perf_counter() - t0) / 1e9)
            print(f"{how:6s} {n:3d} threads: {best:6.1f} GB/s = {best / 4 * 1e3:6.0f} Msample/s of int16 IQ", flush=True)
    os.close(fd)
