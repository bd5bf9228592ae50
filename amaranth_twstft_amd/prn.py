"""PRN replica definition (host side).

Mirrors the reference's LFSR semantics so code files written by either side are
interchangeable:

* Fibonacci LFSR, right shift, feedback bit = parity(state & taps) shifted into the MSB,
  seed 1, output = state LSB *before* the shift, one byte (0/1) per chip
  (reference: amaranth_twstft/common.py:15-30 ``unary_xor``/``nextstate``,
  common.py:59-73 ``write_prn_seq``; C twin tools/mseq_calculator.c:9-18 ``lfsr_next``).
* Code files are raw bytes 0/1, named ``noiselen<N>_bitlen<B>_taps<T>.bin``
  (reference: experiments/221207_twoway_codes/codes/).

The device-side generator (``twx_lfsr_chips`` in csrc/) produces the same bytes; this module is
the host mirror used to write/read code files and by the tests.
"""
from __future__ import annotations

import gzip
import os

import numpy as np


def nextstate(current: int, taps: int, bit_len: int) -> int:
    """One LFSR step (same name and argument meaning as reference common.py:23)."""
    bit = bin(current & taps).count("1") & 1
    return (current >> 1) | (bit << (bit_len - 1))


def lfsr_chips(bit_len: int, taps: int, noiselen: int, seed: int = 1) -> np.ndarray:
    """``noiselen`` chips (uint8 0/1) of the LFSR(bit_len, taps) sequence started at ``seed``."""
    if not (1 <= bit_len <= 32):
        raise ValueError("bit_len must be in 1..32")
    if taps <= 0 or taps >> bit_len:
        raise ValueError("taps must be a non-zero bit_len-bit mask")
    out = np.empty(noiselen, dtype=np.uint8)
    # The output stream obeys s[i+B] = XOR_{t in taps} s[i+t]; over GF(2) the same relation holds
    # with every offset multiplied by 2^m (Frobenius), which lets numpy extend the stream in
    # blocks of 2^m*(B - t_max) chips once the first 2^m*B chips exist.
    tap_pos = [t for t in range(bit_len) if (taps >> t) & 1]
    m = 10
    head = min(noiselen, (bit_len << m))
    state = seed
    msb = bit_len - 1
    for i in range(head):
        out[i] = state & 1
        bit = bin(state & taps).count("1") & 1
        state = (state >> 1) | (bit << msb)
    if noiselen > head:
        step = 1 << m
        blk = step * (bit_len - tap_pos[-1])
        big = bit_len * step
        pos = head
        while pos < noiselen:
            n = min(blk, noiselen - pos)
            base = pos - big
            acc = out[base + tap_pos[0] * step: base + tap_pos[0] * step + n].copy()
            for t in tap_pos[1:]:
                acc ^= out[base + t * step: base + t * step + n]
            out[pos:pos + n] = acc
            pos += n
    return out


def lfsr_chips_slow(bit_len: int, taps: int, noiselen: int, seed: int = 1) -> np.ndarray:
    """Bit-serial form of :func:`lfsr_chips` (kept as the plain statement of the recurrence)."""
    out = np.empty(noiselen, dtype=np.uint8)
    state = seed
    for i in range(noiselen):
        out[i] = state & 1
        state = nextstate(state, taps, bit_len)
    return out


def lfsr_period(bit_len: int, taps: int, seed: int = 1) -> int:
    """Cycle length from ``seed`` (reference: tools/mseq_calculator.c:29-38)."""
    state = seed
    n = 0
    while True:
        state = nextstate(state, taps, bit_len)
        n += 1
        if state == seed or state == 0:
            return n


def write_prn_seq(bitlen: int, noiselen: int, taps_a: int, taps_b: int | None = None, path: str | None = None) -> str:
    """Write a code file with the reference's naming (common.py:59-73): BPSK = one byte per chip; QPSK (``taps_b``) =
    the two sequences interleaved a0 b0 a1 b1 … (read back as ``code(1:2:end)``, ``code(2:2:end)`` by
    experiments/220822_qpsk_vs_bpsk/goqpsk.m:10-11)."""
    if path is None:
        path = f"prn{taps_a}{f'.{taps_b}q' if taps_b else 'b'}psk{bitlen}bits.bin"
    a = lfsr_chips(bitlen, taps_a, noiselen)
    if taps_b:
        out = np.empty(2 * noiselen, dtype=np.uint8)
        out[0::2] = a
        out[1::2] = lfsr_chips(bitlen, taps_b, noiselen)
        out.tofile(path)
    else:
        a.tofile(path)
    return path


def _gf2_matmul(a: list[int], b: list[int]) -> list[int]:
    """Rows as bit masks: (a·b)[i] = XOR of the rows b[j] over the set bits j of a[i]."""
    out = []
    for row in a:
        acc, j = 0, 0
        while row:
            if row & 1:
                acc ^= b[j]
            row >>= 1
            j += 1
        out.append(acc)
    return out


def _gf2_matpow(m: list[int], e: int) -> list[int]:
    n = len(m)
    res = [1 << i for i in range(n)]
    while e:
        if e & 1:
            res = _gf2_matmul(res, m)
        m = _gf2_matmul(m, m)
        e >>= 1
    return res


def lfsr_is_maximal(bit_len: int, taps: int) -> bool:
    """True when LFSR(bit_len, taps) runs through all 2^bit_len - 1 non-zero states — the property
    ``m_seq_codes`` (common.py:32-57) and tools/mseq_calculator.c:29-38 establish by stepping through the whole cycle;
    here by the order of the transition matrix: M^(2^n-1) = I and M^((2^n-1)/p) != I for every prime p | 2^n-1."""
    n = bit_len
    if not (taps & 1):
        return False                                  # singular transition: state 1 is not on a cycle through 1
    # state' = (state >> 1) | (parity(state & taps) << (n-1)) as a matrix acting on bit vectors (row i = image of bit i)
    rows = []
    for i in range(n):
        s = 1 << i
        bit = bin(s & taps).count("1") & 1
        rows.append((s >> 1) | (bit << (n - 1)))
    order = (1 << n) - 1
    ident = [1 << i for i in range(n)]
    if _gf2_matpow(rows, order) != ident:
        return False
    m, p, primes = order, 2, []
    while p * p <= m:
        if m % p == 0:
            primes.append(p)
            while m % p == 0:
                m //= p
        p += 1
    if m > 1:
        primes.append(m)
    return all(_gf2_matpow(rows, order // q) != ident for q in primes)


def m_seq_codes(bit_len: int, limit: int = 10) -> list[int]:
    """The first ``limit`` tap masks (odd values, ascending — the candidates and the order of common.py:32-57) that make
    LFSR(bit_len) a maximum-length sequence generator."""
    codes = []
    for code in range(1, 1 << bit_len, 2):
        if lfsr_is_maximal(bit_len, code):
            codes.append(code)
            if len(codes) == limit:
                break
    return codes


def read_code_file(path: str) -> np.ndarray:
    """Read a chip file (bytes 0/1; ``.gz`` transparently) → uint8 array.

    Reference readers: ``fread(f,inf,'int8')`` godual_ranging.m:63; ``list(fd.read())``
    experiments/221219_twoway/processing/godual_ranging.py:71-74.
    """
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        data = f.read()
    chips = np.frombuffer(data, dtype=np.uint8).copy()
    if chips.size == 0:
        raise ValueError(f"empty code file {path}")
    if chips.max() > 1:
        raise ValueError(f"{os.path.basename(path)}: chips must be bytes 0/1")
    return chips


def chips_to_code(chips: np.ndarray, sps: int = 2) -> np.ndarray:
    """``repelems`` ×sps and ``2*code-1`` (godual_ranging.m:64-65) — the ``code`` vector saved in the .mat."""
    return np.repeat(np.asarray(chips, dtype=np.float64), sps) * 2.0 - 1.0
