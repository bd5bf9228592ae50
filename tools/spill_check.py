#!/usr/bin/env python3
"""Register-spill audit of the shipped gfx950 code objects (no GPU needed).

    python tools/spill_check.py [--all] [file.so ...]

Walks the `.hip_fatbin` section of every shared object given (default: the library and every plan plug-in), unbundles the
gfx950 code objects (clang offload bundles, plain or zlib/zstd-compressed), reads the kernel metadata with
`llvm-readelf --notes` and lists every kernel with `.vgpr_spill_count` / `.sgpr_spill_count` > 0 or scratch
(`.private_segment_fixed_size`) > 0.  A kernel that spills in a streaming pass pays its scratch traffic out of the same HBM
the pass is bound by; tests/test_build_artifacts.py fails the build on any vector spill.  Exit status 1 when something spills.
"""
from __future__ import annotations

import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("TWX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
CMAGIC = b"CCOB"


def _section(path: str, name: str) -> bytes:
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        m = re.match(r"\s*\[\s*\d+\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m and m.group(1) == name:
            off, size = int(m.group(3), 16), int(m.group(4), 16)
            with open(path, "rb") as f:
                f.seek(off)
                return f.read(size)
    return b""


def _decompress(blob: bytes) -> bytes:
    """One compressed bundle (CCOB header, clang's CompressedOffloadBundle) -> the plain bundle."""
    ver = struct.unpack_from("<H", blob, 4)[0]
    method = struct.unpack_from("<H", blob, 6)[0]
    if ver == 1:
        hdr = 4 + 2 + 2 + 4 + 8
    elif ver == 2:
        hdr = 4 + 2 + 2 + 4 + 4 + 8
    else:
        hdr = 4 + 2 + 2 + 8 + 8 + 8
    body = blob[hdr:]
    if method == 0:
        import zlib
        return zlib.decompress(body)
    try:
        import zstandard  # type: ignore
        return zstandard.ZstdDecompressor().decompress(body, max_output_size=1 << 31)
    except ImportError:
        p = subprocess.run(["zstd", "-d", "-c"], input=body, capture_output=True)
        if p.returncode != 0:
            raise RuntimeError("compressed offload bundle and neither `zstandard` nor `zstd` available")
        return p.stdout


def code_objects(path: str):
    """Yields (triple, bytes) of every device code object bundled into `path`."""
    fat = _section(path, ".hip_fatbin")
    pos = 0
    while pos < len(fat):
        i_plain, i_comp = fat.find(MAGIC, pos), fat.find(CMAGIC, pos)
        cands = [i for i in (i_plain, i_comp) if i >= 0]
        if not cands:
            break
        i = min(cands)
        if i == i_comp and (i_plain < 0 or i_comp < i_plain):
            # compressed bundle: its total size is in the header for version >= 2; version 1 runs to the next magic
            ver = struct.unpack_from("<H", fat, i + 4)[0]
            if ver >= 2:
                tot = struct.unpack_from("<I" if ver == 2 else "<Q", fat, i + 8)[0]
            else:
                nxt = fat.find(CMAGIC, i + 4)
                tot = (nxt if nxt >= 0 else len(fat)) - i
            bundle = _decompress(fat[i:i + tot])
            pos = i + tot
            base = 0
        else:
            bundle, base = fat, i
            pos = i + len(MAGIC)
        n = struct.unpack_from("<Q", bundle, base + len(MAGIC))[0]
        p = base + len(MAGIC) + 8
        end = base
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", bundle, p)
            triple = bundle[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if size and "amdgcn" in triple:
                yield triple, bundle[base + off: base + off + size]
            end = max(end, base + off + size)
        if bundle is fat:
            pos = max(pos, end)


def kernels_of(obj: bytes):
    """[(name, {field: int})] from the AMDGPU metadata note of one code object."""
    with tempfile.NamedTemporaryFile(suffix=".co", dir="/tmp") as t:
        t.write(obj)
        t.flush()
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", t.name], capture_output=True, text=True).stdout
    # amdhsa.kernels is a YAML list: an entry starts with "  - .key:" and its own keys sit at "    .key:" (argument lists are deeper)
    kern, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"(  - | {4})\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        if m.group(1) == "  - ":
            cur = {}
            kern.append(cur)
        if cur is not None:
            cur[m.group(2)] = m.group(3).strip().strip("'\"")
    return [(k.get("name", "?"), k) for k in kern if "vgpr_count" in k]


def audit(paths):
    rows = []
    for path in paths:
        for triple, obj in code_objects(path):
            for name, k in kernels_of(obj):
                g = lambda f: int(k.get(f, "0") or 0)
                rows.append({"file": os.path.relpath(path, ROOT), "kernel": name, "vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"),
                             "vgpr_spill": g("vgpr_spill_count"), "sgpr_spill": g("sgpr_spill_count"), "scratch": g("private_segment_fixed_size"),
                             "lds": g("group_segment_fixed_size")})
    return rows


def demangle(names):
    import shutil
    exe = shutil.which("llvm-cxxfilt", path=LLVM) or shutil.which("c++filt")
    if not exe or not names:
        return list(names)
    p = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else list(names)


def default_paths():
    pk = os.path.join(ROOT, "amaranth_twstft_amd")
    return [os.path.join(pk, "libtwstft_hip.so")] + sorted(glob.glob(os.path.join(pk, "plans", "*.so")))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = audit(args or default_paths())
    bad = [r for r in rows if r["vgpr_spill"] or r["scratch"]]
    show = rows if "--all" in sys.argv else bad
    names = demangle([r["kernel"] for r in show])
    for r, nm in zip(show, names):
        print("%-40s vgpr %3d sgpr %3d lds %6d  vgpr_spill %3d sgpr_spill %3d scratch %4d  %s" % (r["file"][-40:], r["vgpr"], r["sgpr"], r["lds"], r["vgpr_spill"],
                                                                                               r["sgpr_spill"], r["scratch"], nm[:160]))
    print("%d kernels in %d files; %d with vector spills or scratch" % (len(rows), len(set(r["file"] for r in rows)), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
