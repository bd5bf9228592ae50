"""CPU: the tracked-ranging control flow of the product library (amaranth_twstft_amd/csrc/twx_tracked_core.h: speculative
batches + serial re-alignment fix-up, search_df, the ``lo`` per-chunk band arg-max) compiled with g++ and driven through C
callbacks that the ORACLE answers — the same loop that runs the device must reproduce oracle.ranging_tracked exactly."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from amaranth_twstft_amd import synth
from oracle import twstft_oracle as orc
from tests.helpers import chips_for

FS = 5e6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Meas(C.Structure):
    _fields_ = [("indice0", C.c_longlong)] + [(k, C.c_double) for k in
                                               ("correction", "xre", "xim", "snr_r", "snr_i", "puissance", "pcode", "pnoise")]


class Code(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("xre", "xim", "indice1", "correction1", "snr_r", "snr_i", "puissance1")]


class Params(C.Structure):
    _fields_ = [("n", C.c_longlong), ("L", C.c_longlong), ("r", C.c_int), ("carrier", C.c_int), ("indice_floor", C.c_int),
                ("pad", C.c_int), ("fs", C.c_double), ("band_lo", C.c_double), ("band_hi", C.c_double), ("df_threshold", C.c_double)]


LOAD = C.CFUNCTYPE(C.c_int, C.c_longlong, C.c_longlong, C.POINTER(C.c_int))
MEASURE = C.CFUNCTYPE(C.c_int, C.c_longlong, C.c_int, C.c_double, C.POINTER(Meas))
SQBINS = C.CFUNCTYPE(C.c_int, C.c_longlong, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_double))
SQBAND = C.CFUNCTYPE(C.c_int, C.c_longlong, C.c_longlong, C.c_longlong, C.POINTER(C.c_double))
CAND = C.CFUNCTYPE(C.c_int, C.c_longlong, C.c_double, C.POINTER(C.c_double))
SLIDE = C.CFUNCTYPE(C.c_int, C.c_longlong, C.c_longlong)


class Callbacks(C.Structure):
    _fields_ = [("load_chunk", LOAD), ("measure", MEASURE), ("sq_bins", SQBINS), ("sq_band", SQBAND), ("candidate_snr", CAND),
                ("slide_tail", SLIDE)]


class Summary(C.Structure):
    _fields_ = [(k, C.c_longlong) for k in ("n_codes", "n_chunks", "n_moved", "kbon", "batches")] + [("pcode", C.c_double), ("pnoise", C.c_double)]


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = tmp_path_factory.mktemp("trk") / "tracked_emul.so"
    # TWX_EMUL_SANITIZE=1 (tests/test_sanitizers.py, in an interpreter with libasan preloaded): the same control flow with ASan + UBSan
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"] if os.environ.get("TWX_EMUL_SANITIZE") else ["-O2"]
    subprocess.run(["g++", *san, "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "amaranth_twstft_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "tracked_emul.cpp"), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.trk_emul_run.restype = C.c_int
    lib.trk_emul_run.argtypes = [C.POINTER(Params), C.POINTER(Callbacks), C.c_longlong, C.c_longlong, C.POINTER(Summary)]
    lib.trk_emul_fetch.restype = None
    lib.trk_emul_fetch.argtypes = [C.c_void_p] * 4
    lib.trk_emul_freq.restype = C.c_double
    lib.trk_emul_freq.argtypes = [C.c_double, C.c_longlong, C.c_longlong]
    lib.trk_emul_band.restype = None
    lib.trk_emul_band.argtypes = [C.c_double, C.c_longlong, C.c_double, C.c_double, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    lib.trk_emul_median.restype = C.c_double
    lib.trk_emul_median.argtypes = [C.c_void_p, C.c_longlong]
    return lib


class OracleBackend:
    """The six Backend operations answered by the oracle on a host copy of the sample buffer."""

    def __init__(self, raw, chips, n, Lc):
        self.raw, self.n, self.L = np.asarray(raw).reshape(-1), n, Lc
        self.buf = np.zeros(Lc + n + 64, dtype=np.complex128)
        self.code = orc.make_code(chips, 2)
        self.fc = orc.make_fcode(self.code, "claudio")
        self.temps = np.arange(n) / FS
        self.measures = 0

        def load_chunk(pos, carry, full):
            chunk = self.raw[pos:pos + 2 * Lc]
            full[0] = int(chunk.size == 2 * Lc)
            if full[0]:
                self.buf[carry:carry + Lc] = chunk[0::2].astype(np.float64) + 1j * chunk[1::2].astype(np.float64)
            return 0

        def measure(start, count, df, out):
            self.measures += 1
            for j in range(count):
                d = self.buf[start + j * n:start + (j + 1) * n]
                o = orc.processing_claudio(d - d.mean(), df, self.temps, self.fc, self.code, Nint=1, ddof=1)
                out[j].indice0, out[j].correction = o["indice"], o["correction"]
                out[j].xre, out[j].xim = o["xval"].real, o["xval"].imag
                out[j].snr_r, out[j].snr_i, out[j].puissance = o["SNRr"], o["SNRi"], o["puissance"]
                out[j].pcode, out[j].pnoise = o["puissancecode"], o["puissancenoise"]
            return 0

        def sq_bins(ns, bins, nb, out):
            f = np.fft.fft(self.buf[:ns] ** 2)
            for i in range(nb):
                v = f[bins[i] % ns]
                out[2 * i], out[2 * i + 1] = v.real, v.imag
            return 0

        def sq_band(offset, k_lo, nk, mag):
            f = np.abs(np.fft.fft(self.buf[offset:offset + Lc] ** 2))
            np.ctypeslib.as_array(mag, shape=(nk,))[:] = f[np.arange(k_lo, k_lo + nk) % Lc]
            return 0

        def candidate_snr(offset, dftmp, snr):
            y = self.buf[offset:offset + n] * np.exp(-2j * np.pi * dftmp * self.temps)
            prnmap = np.abs(np.fft.ifft(self.fc * np.conj(np.fft.fft(y))))
            b = int(prnmap.argmax())
            sig = prnmap[b]
            prnmap[max(b - 5, 0):b + 6] = 0
            snr[0] = sig ** 2 / np.var(prnmap, ddof=1)
            return 0

        def slide_tail(src, count):
            self.buf[:count] = self.buf[src:src + count].copy()
            return 0

        self.cb = Callbacks(LOAD(load_chunk), MEASURE(measure), SQBINS(sq_bins), SQBAND(sq_band), CAND(candidate_snr), SLIDE(slide_tail))


def _run(emul, raw, chips, n, Lc, band=(-8000.0, 8000.0), carrier=0, indice_floor=0, skip=0, kbon=-1):
    be = OracleBackend(raw, chips, n, Lc)
    p = Params(n, Lc, 3, carrier, indice_floor, 0, FS, band[0], band[1], 20.0)
    s = Summary()
    rc = emul.trk_emul_run(C.byref(p), C.byref(be.cb), skip, kbon, C.byref(s))
    assert rc == 0
    codes = (Code * max(s.n_codes, 1))()
    df = np.zeros(max(s.n_chunks, 1)); moved = np.zeros(max(s.n_moved, 1), dtype=np.int64); mv = np.zeros(max(s.n_moved, 1))
    emul.trk_emul_fetch(C.cast(codes, C.c_void_p), df.ctypes.data, moved.ctypes.data, mv.ctypes.data)
    c = [codes[i] for i in range(s.n_codes)]
    return dict(xval=[complex(x.xre, x.xim) for x in c], indice1=[x.indice1 for x in c], correction1=[x.correction1 for x in c],
                SNR1r=[x.snr_r for x in c], SNR1i=[x.snr_i for x in c], puissance1=[x.puissance1 for x in c],
                df=list(df[:s.n_chunks]), moved=list(moved[:s.n_moved]), movedval=list(mv[:s.n_moved]), kbon=s.kbon,
                batches=s.batches, measures=be.measures)


def _capture(ncodes, delay, seed, df=30.0):
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    p = synth.SynthParams(delay_q8=delay * 256, fstep=synth.fstep_for_df(df, FS), phi0=5, amp=500,
                          noise_gain=synth.noise_gain_for_sigma(300.0), seed=seed)
    return chips, n, synth.synth_channel(n * ncodes, chips, 2, p)


def test_axis_band_and_median_helpers_match_numpy(emul):
    """freq=linspace(-fs/2,fs/2-1,fs*ls) (:132), k=find(...) (:134-141) and median() as the C++ loop forms them."""
    for L in (600000, 10_000_000):
        f = np.linspace(-FS / 2, FS / 2 - 1.0, L)
        idx = np.r_[0:5, L // 2 - 3:L // 2 + 3, L - 5:L, np.random.default_rng(1).integers(0, L, 200)]
        got = np.array([emul.trk_emul_freq(FS, L, int(i)) for i in idx])
        assert np.array_equal(got, f[idx])                       # bit-identical doubles
        for lo, hi in ((-8000.0, 8000.0), (-20000.0, 20000.0), (92000.0, 108000.0), (-108000.0, -92000.0), (100000.0, 120000.0)):
            k = np.nonzero((f < hi) & (f > lo))[0]
            k0, nk = C.c_longlong(), C.c_longlong()
            emul.trk_emul_band(FS, L, lo, hi, C.byref(k0), C.byref(nk))
            assert (k0.value, nk.value) == (int(k[0]), k.size)
    rng = np.random.default_rng(2)
    for m in (1, 2, 7, 1000, 1001):
        v = rng.standard_normal(m)
        assert emul.trk_emul_median(v.ctypes.data, m) == np.median(v)


def test_tracked_control_flow_matches_the_oracle_loop(emul):
    chips, n, a = _capture(45, 1500, 4)
    _, _, b = _capture(45, 1500 + 777, 5)                 # a delay jump in the middle forces a second re-alignment
    raw = np.concatenate((a, b))
    Lc = 30 * n
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc)
    got = _run(emul, raw, chips, n, Lc)
    assert got["kbon"] == want["kbon"] > 0 and got["df"] == want["df"]
    assert got["moved"] == want["moved"] and len(want["moved"]) >= 2
    assert got["indice1"] == want["indice1"] and got["movedval"] == want["movedval"]
    assert np.allclose(got["correction1"], want["correction1"]) and np.allclose(got["xval"], want["xval"])
    for key in ("SNR1r", "SNR1i", "puissance1"):
        assert np.allclose(got[key], want[key])
    # speculation: one batch per chunk plus one per re-alignment (batch cut + 1-window re-measure)
    assert got["batches"] <= len(want["df"]) + 2 * len(want["moved"])


def test_skip_and_known_carrier(emul):
    """fseek(f,30*fs*2*2) (:128) only moves the chunk search_df sees (the file is re-read from its start, :156-159); a carrier
    handed over beforehand skips the search and honours the skip."""
    chips, n, raw = _capture(70, 900, 8)
    Lc = 30 * n
    for skip in (0, 5 * n):
        want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, skip_samples=skip)
        got = _run(emul, raw, chips, n, Lc, skip=skip)
        assert got["kbon"] == want["kbon"] > 0 and got["indice1"] == want["indice1"] and got["df"] == want["df"]
    kb = want["kbon"]
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, skip_samples=7 * n, kbon=kb)
    got = _run(emul, raw, chips, n, Lc, skip=7 * n, kbon=kb)
    assert got["indice1"] == want["indice1"] and got["moved"] == want["moved"] and len(want["indice1"]) > 30


def test_no_carrier_found_ends_after_three_tries(emul):
    rng = np.random.default_rng(3)
    chips = chips_for(14, 43, 10000)
    raw = np.round(rng.standard_normal(2 * 20000 * 70) * 300).astype(np.int16)
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=30 * 20000)
    got = _run(emul, raw, chips, 20000, 30 * 20000)
    assert want["kbon"] == got["kbon"] == -1 and got["indice1"] == want["indice1"] == []


@pytest.mark.parametrize("mode,OP", [("lo", 0), ("re", 0), ("re", 1)])
def test_lo_and_re_siblings(emul, mode, OP):
    """claudio_aligned_code_lo_separate.m (per-chunk full-band carrier, floor() of the lag, no search) and the remote band of
    claudio_aligned_code_re_separate.m through the same C++ loop, against the oracle's restatement of those scripts."""
    m = orc.tracked_mode(mode, OP)
    car = (m["band"][0] + m["band"][1]) / 4          # a carrier whose doubled line sits mid-band
    if mode == "lo":
        car = 1234.5
    chips, n, a = _capture(40, 1500, 14, df=car)
    _, _, b = _capture(40, 1500 + 333, 15, df=car)
    raw = np.concatenate((a, b))
    Lc = 30 * n
    kw = dict(band=m["band"], carrier=m["carrier"], indice_floor=m["indice_floor"])
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, **kw)
    got = _run(emul, raw, chips, n, Lc, band=m["band"], carrier=1 if m["carrier"] == "chunk_band" else 0,
               indice_floor=int(m["indice_floor"]))
    assert len(want["indice1"]) >= 55 and len(want["moved"]) >= 2
    assert got["df"] == want["df"] and abs(want["df"][0] - car) < 3.0      # the axis of :132 assumes 1-Hz bins; short test chunks are off by ~2 Hz at 100 kHz
    assert got["kbon"] == want["kbon"] and (want["kbon"] > 0) == (mode != "lo")
    assert got["indice1"] == want["indice1"] and got["moved"] == want["moved"] and got["movedval"] == want["movedval"]
    assert np.allclose(got["xval"], want["xval"])
    if mode == "lo":
        assert all(float(v).is_integer() for v in want["indice1"])


@pytest.mark.parametrize("delay,jump_at,jump", [(500, None, 0), (15000, 29, 4000), (300, 28, -250), (12000, 57, 9000)])
def test_edge_paths_of_the_realignment(emul, delay, jump_at, jump):
    """The branches of :176-193 that ordinary captures do not reach: a peak in the upper half of the code (``dindex-indice1+1 < 0``
    -> one code period is skipped, :180-182), a re-alignment asked for by the LAST code of a chunk (the script would index past
    the chunk; restated as: keep the first measurement, end the chunk), jumps backwards."""
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    Lc = 30 * n

    def seg(ncodes, d, seed):
        p = synth.SynthParams(delay_q8=d * 256, fstep=synth.fstep_for_df(30.0, FS), phi0=5, amp=500,
                              noise_gain=synth.noise_gain_for_sigma(300.0), seed=seed)
        return synth.synth_channel(n * ncodes, chips, 2, p)

    if jump_at is None:
        raw = seg(70, delay, 41)
    else:
        raw = np.concatenate((seg(jump_at, delay, 42), seg(70 - jump_at, delay + jump, 43)))
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc)
    got = _run(emul, raw, chips, n, Lc)
    assert want["kbon"] > 0 and len(want["indice1"]) >= 55 and len(want["moved"]) >= (1 if jump_at is None else 2)
    assert got["indice1"] == want["indice1"] and got["moved"] == want["moved"] and got["movedval"] == want["movedval"]
    assert got["df"] == want["df"] and np.allclose(got["xval"], want["xval"])
    if jump_at is None:                                           # claudio convention: the peak of a small delay sits at n - delay
        assert want["movedval"][0] > n / 2                       # the upper-half branch was taken


@pytest.mark.parametrize("seed", range(6))
def test_randomised_captures_through_the_same_loop(emul, seed):
    """Random delays, carriers, jump positions and sizes, chunk lengths and skips, all three flavours: the C++ loop and the
    oracle's restatement must agree on every stored lag, every re-alignment and every carrier."""
    rng = np.random.default_rng(100 + seed)
    nchips, n = 10000, 20000
    chips = chips_for(14, 43, nchips)
    mode = ("ranging", "lo", "re")[seed % 3]
    OP = int(rng.integers(0, 2))
    m = orc.tracked_mode(mode, OP)
    car = float(rng.uniform(-3000, 3000)) if mode != "re" else (m["band"][0] + m["band"][1]) / 4 + float(rng.uniform(-500, 500))
    segs, d = [], int(rng.integers(100, n - 100))
    for k in range(int(rng.integers(2, 5))):
        p = synth.SynthParams(delay_q8=d * 256, fstep=synth.fstep_for_df(car, FS), phi0=int(rng.integers(0, 2 ** 31)), amp=500,
                              noise_gain=synth.noise_gain_for_sigma(300.0), seed=1000 * seed + k)
        segs.append(synth.synth_channel(n * int(rng.integers(12, 30)), chips, 2, p))
        d = int((d + rng.integers(-3000, 3000)) % n)
    raw = np.concatenate(segs)
    Lc = n * int(rng.integers(8, 20))
    skip = int(rng.integers(0, 3)) * n
    kw = dict(band=m["band"], carrier=m["carrier"], indice_floor=m["indice_floor"])
    want = orc.ranging_tracked(raw, chips, fs=FS, ls_samples=Lc, skip_samples=skip, **kw)
    got = _run(emul, raw, chips, n, Lc, band=m["band"], carrier=1 if m["carrier"] == "chunk_band" else 0,
               indice_floor=int(m["indice_floor"]), skip=skip)
    assert got["kbon"] == want["kbon"] and got["df"] == want["df"]
    assert got["indice1"] == want["indice1"] and got["moved"] == want["moved"] and got["movedval"] == want["movedval"]
    assert np.allclose(got["xval"], want["xval"]) and np.allclose(got["correction1"], want["correction1"])
    assert len(want["indice1"]) >= 8
