"""CPU ORACLE for the TWSTFT correlation hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``amaranth_twstft_amd``) never does, and fails loudly when the
HIP library is missing.

This is a plain numpy (fp64, ``np.fft`` = pocketfft, the same library the reference's numpy
path uses) restatement of the reference algorithm.  Every function cites the reference lines it
follows (paths relative to the reference repository root).

Parity pinning (see tests/golden/README.md, tools/make_golden.py):
  * PRN generation is pinned by the reference's own code files (SHA-256 / prefix fixtures made
    from experiments/221207_twoway_codes/codes/*.bin.gz, 231001_DLL_PLL/{0,1}.bin,
    220706_TWSTFT/*.bin) and by ``amaranth_twstft/common.py:nextstate`` imported at
    fixture-generation time.
  * ``processing()`` is pinned by outputs of the reference's own numpy code run in the build
    container on seeded synthetic captures: ``experiments/221219_twoway/processing/
    godual_ranging.py:processing`` (full-precision return values, fine-frequency step on) and
    ``experiments/221207_twoway_codes/processing/godual_ranging.py:ranging`` (printed rows);
    with the zero-mean 0/1 replica of ``make_code_variant`` by ``experiments/220830_OP/
    godual_ranging_OP.py:ranging`` (printed lag, correction and complex peak sample).
  * The reference holds NO test vectors of its own for the correlator, and the
    Octave-only variants (``processing_claudio``, ``search_df``, ``ranging_tracked``, ``go_1s_session``, the QPSK form of ``make_code_variant``,
    ``peak_refine_polyfit``, ``epl_step`` / ``octave_xcorr``, ``ranging_vitesse`` / ``interp1_linear``, the off-peak and
    squared-spectrum SNR estimators ``snr_offpeak`` / ``snr_square``), the C++-only Hamming window
    and the 231001_DLL_PLL acquisition/tracking restatements (incl. the receiver programs ``rx_second`` — rxcomplex.cpp and, with
    ``real=True``, rx.cpp with its interference cancellation ``rx_mai_up`` / ``rx_mai_out``) have no runnable twin here:
    for those functions parity is UNPINNED (hand restatement, cross-checked by identities) — EXCEPT what the reference's own result
    archives pin (round 5, tools/make_golden_archives.py -> tests/golden/ref_archives.*, tests/test_ref_archives.py): the parabola of
    ``peak_refine`` / ``peak_refine_polyfit(.., 1)`` against 3 850 stored (xvalm1, xval, xvalp1, correction) sets of 220616_Besancon and the
    ``freq_axis`` grid against its 101 distinct df values; ``make_code(lfsr_chips(17, 15 | 9, 100000))`` against the `code` variable of
    230315_analysis_100k and SNR1r + SNR1i = puissancecode / puissancenoise; the re-alignment rule of ``ranging_tracked``
    (claudio...separate.m:175-185: the limits 43 / n/2 / n-2, the -30 dB gate, movedval = indice1 + 1, the undivided re-measured
    index) against all 2 087 production records of 2401_{OP,LTFB} and 240527 (16.9 M codes, 3 358 moves), and the window arithmetic of the
    code loop (:166-200: dindex, +21, the + length(fcode) wrap, the dold carry, codes per chunk) by replaying 70 whole records (109 moves) through the
    product's loop (twx_tracked_core.h — the same loop tests/test_tracked_host.py holds against ``ranging_tracked``); ``go_1s_session`` runs on two
    real sessions of 240527 (no stored output exists to compare with: the numbers are physically checked, not pinned).
"""
from __future__ import annotations

import numpy as np

# FFT backend of the restatement: numpy's pocketfft (what the reference's numpy path calls) by default;
# ``use_fft_backend("scipy", workers=-1)`` switches to scipy.fft with its internal thread pool — the multi-threaded
# CPU baseline of SURVEY.md §8(d) (pyfftw ``threads=4`` in experiments/221207_twoway_codes/processing/godual_ranging_fftw.py:62).
_backend = {"fft": np.fft.fft, "ifft": np.fft.ifft}


def use_fft_backend(name: str = "numpy", workers: int | None = None) -> None:
    if name == "numpy":
        _backend.update(fft=np.fft.fft, ifft=np.fft.ifft)
    elif name == "scipy":
        import scipy.fft as sf
        _backend.update(fft=lambda x: sf.fft(x, workers=workers), ifft=lambda x: sf.ifft(x, workers=workers))
    else:
        raise ValueError(name)


def _fft(x):
    return _backend["fft"](x)


def _ifft(x):
    return _backend["ifft"](x)


# --------------------------------------------------------------------------------------------
# PRN replica
# --------------------------------------------------------------------------------------------

def lfsr_next(lfsr: int, taps: int, bits: int) -> int:
    """tools/mseq_calculator.c:9-18 ``lfsr_next`` ≡ amaranth_twstft/common.py:23-30 ``nextstate``."""
    masked = lfsr & taps
    bit = 0
    for i in range(bits):
        bit ^= (masked >> i) & 1
    return (lfsr >> 1) | (bit << (bits - 1))


def lfsr_chips(bitlen: int, taps: int, noiselen: int) -> np.ndarray:
    """amaranth_twstft/common.py:59-73 ``write_prn_seq`` (BPSK branch): seed 1, byte = state%2."""
    out = np.empty(noiselen, dtype=np.uint8)
    a = 1
    for i in range(noiselen):
        out[i] = a % 2
        a = lfsr_next(a, taps, bitlen)
    return out


def make_code(chips: np.ndarray, sps: int = 2) -> np.ndarray:
    """processing/Octave/godual_ranging.m:63-65: repelems ×2, ``2*code-1`` → ±1 (float64)."""
    return np.repeat(np.asarray(chips, dtype=np.float64), sps) * 2.0 - 1.0


def make_code_variant(chips, chips_q=None, sps: int = 2, unipolar: bool = False, zero_mean: bool = False) -> np.ndarray:
    """Replica variants of the experiment scripts.  The real zero-mean 0/1 replica is PINNED: it is also what
    experiments/220830_OP/godual_ranging_OP.py:17-24 builds, and tests/golden/ref220830_op_ranging.json holds that
    script's own output (tools/make_golden.py gen_220830; tests/test_oracle_golden.py).  The QPSK form is unpinned (Octave only).
    0/1 levels and ``code=code-mean(code)`` (experiments/220616_Besancon/godual.m:5-7, mean removed after repelems);
    complex QPSK code ``codec=codei+j*codeq; codec=codec-mean(codec); repelems`` (experiments/220822_qpsk_vs_bpsk/
    goqpsk.m:10-14, mean removed before repelems — the same thing, the hold does not change the mean)."""
    lev = (lambda c: np.asarray(c, dtype=np.float64)) if unipolar else (lambda c: 2.0 * np.asarray(c, dtype=np.float64) - 1.0)
    code = lev(chips).astype(np.complex128 if chips_q is not None else np.float64)
    if chips_q is not None:
        code = code + 1j * lev(chips_q)
    code = np.repeat(code, sps)
    if zero_mean:
        code = code - code.mean()
    return code


def make_fcode(code: np.ndarray, convention: str = "godual") -> np.ndarray:
    """Code spectrum.

    ``godual``:  ``fcode=conj(fft(code'))``  processing/Octave/godual_ranging.m:66,
                 experiments/221219_twoway/processing/godual_ranging.py:78.
    ``claudio``: ``fcode=fft(code')``  acquisition/claudio_aligned_code_ranging_separate.m:124.
    ``hamming``: godual × Hamming window over the spectrum index, processing/CPP/main.cpp:717-719
                 (sigpack ``hamming(N)`` = 0.54-0.46*cos(2*pi*i/(N-1))).  UNPINNED.
    """
    f = _fft(code)
    if convention == "godual":
        return np.conj(f)
    if convention == "claudio":
        return f
    if convention == "hamming":
        n = len(code)
        w = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))
        return np.conj(f) * w
    raise ValueError(convention)


def freq_axis(fs: float, n: int) -> np.ndarray:
    """``freq=linspace(-fs/2,fs/2,length(code))`` godual_ranging.m:73 (spacing fs/(n-1), sic)."""
    return np.linspace(-fs / 2, fs / 2, num=n, dtype=float)


def band_godual(freq: np.ndarray, remote: int = 0, OP: int = 0) -> np.ndarray:
    """Search band ``k`` of godual_ranging.m:83-89 (0-based indices into the fftshifted axis)."""
    if remote != 1:
        return np.nonzero((freq < 20000) & (freq > -20000))[0]
    if OP == 1:
        return np.nonzero((freq > -120000) & (freq < -80000))[0]
    return np.nonzero((freq < 120000) & (freq > 80000))[0]


def band_numpy(freq: np.ndarray, foffset: float = 0.0, frange: float = 8000.0) -> np.ndarray:
    """``k`` of experiments/221219_twoway/processing/godual_ranging.py:80-81."""
    return np.nonzero((freq < 2 * (foffset + frange)) & (freq > 2 * (foffset - frange)))[0]


# --------------------------------------------------------------------------------------------
# capture reader
# --------------------------------------------------------------------------------------------

def deinterleave(raw: np.ndarray, n_channels: int, channel: int) -> np.ndarray:
    """int16 ``[I1 Q1 I2 Q2 …]`` → complex128 of one channel.

    godual_ranging.m:76-79 (2 channels), godual_ranging.py:100-102,112-113;
    single-channel files acquisition/claudio_aligned_code_ranging_separate.m:148-151.
    """
    raw = np.asarray(raw).reshape(-1, 2 * n_channels)
    return raw[:, 2 * channel].astype(np.float64) + 1j * raw[:, 2 * channel + 1].astype(np.float64)


# --------------------------------------------------------------------------------------------
# the per-window kernel
# --------------------------------------------------------------------------------------------

def coarse_df(d: np.ndarray, k: np.ndarray, freq: np.ndarray):
    """godual_ranging.m:14-15 / godual_ranging.py:21-23. Returns (0-based shifted index, df)."""
    d2_fft = np.fft.fftshift(np.abs(_fft(d * d)))
    tmp = int(d2_fft[k].argmax() + k[0])
    return tmp, freq[tmp] / 2


def fine_df(y: np.ndarray, fs: float) -> float:
    """Fine frequency from the phase drift, experiments/221219_twoway/processing/
    godual_ranging.py:26-27 (Octave twin 221219…/godual_ranging.m:19-21; commented out in
    processing/Octave/godual_ranging.m:19-24).  Needs len(y) >= fs/3."""
    a = np.polyfit(np.arange(1, fs // 3, 10) / fs,
                   np.convolve(np.angle(y[0:int(fs // 3):10]), np.ones(100) / 100)[49:-50], 1)
    return a[0] / 2 / np.pi


def xcorr_interp(ffty: np.ndarray, fcode: np.ndarray, Nint: int) -> np.ndarray:
    """godual_ranging.m:25-28 in the direct-scatter form of godual_ranging.py:31-37:
    spectrum product, zero-pad to (2*Nint+1)*N around Nyquist, inverse FFT."""
    n = len(ffty)
    multmp = ffty * fcode
    if Nint == 0:
        return _ifft(multmp)
    interpolation = np.zeros((2 * Nint + 1) * n, dtype=complex)
    interpolation[:n // 2] = multmp[:n // 2]
    interpolation[-(n // 2):] = multmp[-(n // 2):]
    return _ifft(interpolation)


def peak_refine(prnmap: np.ndarray, wrap: bool = True):
    """godual_ranging.m:29-33. ``indice`` is 0-based here (Octave's is this +1).

    The reference indexes ``indice-1`` / ``indice+1`` without bounds handling (numpy wraps -1,
    raises at the end; Octave raises at both ends); ``wrap=True`` treats the map as circular,
    which is what acquisition/claudio_aligned_code_ranging_separate.m:71-80 does explicitly.
    """
    m = len(prnmap)
    indice = int(np.abs(prnmap).argmax())
    xval = prnmap[indice]
    xvalm1 = prnmap[(indice - 1) % m] if wrap else prnmap[indice - 1]
    xvalp1 = prnmap[(indice + 1) % m] if wrap else prnmap[indice + 1]
    correction = (abs(xvalm1) - abs(xvalp1)) / (abs(xvalm1) + abs(xvalp1) - 2 * abs(xval)) / 2
    return indice, correction, xval, xvalm1, xvalp1


def peak_refine_polyfit(prnmap: np.ndarray, indice: int, half_width: int) -> float:
    """``correction1_1/_2/_3`` of experiments/221207_twoway_codes/processing/godual_ranging.m:73-78 (circular indices)."""
    n = len(prnmap)
    idx = (indice + np.arange(-half_width, half_width + 1)) % n
    u = np.polyfit(np.arange(-half_width, half_width + 1, dtype=float), np.abs(prnmap[idx]), 2)
    return float(-u[1] / 2 / u[0])


def snr_wipeoff(ffty: np.ndarray, code: np.ndarray, indice: int, Nint: int,
                rot: int = -1, ddof: int = 0):
    """godual_ranging.m:38-48 / godual_ranging.py:55-64.

    ``rot`` is the rotate offset relative to the 0-based peak index: -1 in godual_ranging.m:43
    (1-based ``indice-1``), godual_ranging.py(221219):59 and processing/CPP/main.cpp:332; the
    221207 numpy file uses -2 (ch1, :104) and 0 (ch2, :122).
    ``ddof``: numpy ``np.var`` is population (0); Octave ``var`` is N-1 (1).
    Returns (SNRr, SNRi, puissancecode, puissancenoise).
    """
    n = len(ffty)
    r = 2 * Nint + 1
    yint = np.zeros(r * n, dtype=complex)
    yint[:n // 2] = ffty[:n // 2]
    yint[-(n // 2):] = ffty[-(n // 2):]
    yinti = _ifft(yint)
    codetmp = np.repeat(code, r)
    s = (indice + rot) % (r * n)
    yincode = np.concatenate((yinti[s:], yinti[:s])) * codetmp
    var = np.var(yincode, ddof=ddof)
    mr = np.mean(np.real(yincode))
    mi = np.mean(np.imag(yincode))
    return mr ** 2 / var, mi ** 2 / var, mr ** 2 + mi ** 2, var


def processing(d, k, freq, temps, fcode, code, Nint=1, fs=5e6, fine_freq=False,
               snr_rot=-1, ddof=0, df=None):
    """``processing(d,k)`` of processing/Octave/godual_ranging.m:12-49
    (≡ experiments/221219_twoway/processing/godual_ranging.py:18-65 when ``fine_freq=True``).

    ``d`` is the mean-removed complex window.  If ``df`` is given the coarse estimate is skipped
    (code-phase-only variant, cf. ``processing(d,df)`` of claudio_aligned_code_ranging_separate.m:49).
    Returns a dict with 0-based ``indice``.
    """
    if df is None:
        _, dftmp = coarse_df(d, k, freq)
    else:
        dftmp = float(df)
    lo = np.exp(-1j * 2 * np.pi * dftmp * temps)
    y = d * lo
    if fine_freq:
        dfleftover = fine_df(y, fs)
        lo = np.exp(-1j * 2 * np.pi * dfleftover * temps)
        y = y * lo
        dftmp += dfleftover
    ffttmp = _fft(y)
    prnmap = xcorr_interp(ffttmp, fcode, Nint)
    indice, correction, xval, xvalm1, xvalp1 = peak_refine(prnmap)
    SNRr, SNRi, pcode, pnoise = snr_wipeoff(ffttmp, code, indice, Nint, rot=snr_rot, ddof=ddof)
    return dict(indice=indice, correction=correction, xval=xval, xvalm1=xvalm1, xvalp1=xvalp1,
                df=dftmp, SNRr=SNRr, SNRi=SNRi, puissance=np.var(y, ddof=ddof),
                puissancecode=pcode, puissancenoise=pnoise)


def ranging(raw, chips, fs=5e6, sps=2, Nint=1, n_channels=2, channels=(0, 1), band="godual",
            remote=0, OP=0, foffset=0.0, frange=8000.0, **kw):
    """Window loop of godual_ranging.m:59-102 over an in-memory int16 capture.

    ``raw``: int16 array of interleaved samples.  A short final window is dropped
    (godual_ranging.m:81,102).  Returns {channel: [result dict per window]}.
    """
    code = make_code(chips, sps)
    fcode = make_fcode(code, "godual")
    n = len(code)
    freq = freq_axis(fs, n)
    k = band_godual(freq, remote, OP) if band == "godual" else band_numpy(freq, foffset, frange)
    temps = np.arange(n) / fs
    raw = np.asarray(raw).reshape(-1, 2 * n_channels)
    nwin = raw.shape[0] // n
    out = {c: [] for c in channels}
    for p in range(nwin):
        blk = raw[p * n:(p + 1) * n]
        for c in channels:
            d = deinterleave(blk, n_channels, c)
            d = d - np.mean(d)
            out[c].append(processing(d, k, freq, temps, fcode, code, Nint=Nint, fs=fs, **kw))
    return out


# --------------------------------------------------------------------------------------------
# the other SNR estimators the reference compares (experiments/220830_OP/process_OP.m, 221127_SNR/simu_snr.m) — UNPINNED (Octave only)
# --------------------------------------------------------------------------------------------

def snr_offpeak(prnmap: np.ndarray, indice: int, length: int = 1001) -> float:
    """``bruit1(p)=var(prnmap01(indice1(p)+20:indice1(p)+1020))`` behind the guard ``if ((indice1(p)+1020)<length(prnmap01))``
    (experiments/220830_OP/process_OP.m:119-121; ``bruit2`` with 10020, i.e. ``length`` = 10001, :138): the variance (Octave ``var``:
    N-1, complex deviations by modulus) of the correlation map off its peak — the "noise" of the cross-correlation SNR estimate that
    experiments/221127_SNR/README.md shows saturating at high SNR.  ``indice`` 0-based; NaN where the guard fails (the script leaves
    the entry unset)."""
    if (indice + 1) + 20 + length - 1 < len(prnmap):
        return float(np.var(prnmap[indice + 20: indice + 20 + length], ddof=1))
    return float("nan")


def snr_square(d: np.ndarray, k: np.ndarray, length: int = 10001):
    """``[valmax_square(p),df(p)]=max(d22(freqindex)); tmpdf(p)=df(p)+freqindex(1)-1; noise_square(p)=var(d22(tmpdf(p)+20:tmpdf(p)+10020))``
    with ``d22=fftshift(abs(fft(d1.^2)))`` (process_OP.m:94-97): peak of the squared signal's spectrum inside the search band and the
    variance (N-1) of the magnitudes of the ``length`` bins from 20 bins above it.  Returns (valmax_square, noise_square, 0-based shifted
    index); noise NaN where the range leaves the spectrum (Octave would stop with an index error)."""
    d22 = np.fft.fftshift(np.abs(_fft(d * d)))
    tmp = int(d22[k].argmax() + k[0])
    noise = float(np.var(d22[tmp + 20: tmp + 20 + length], ddof=1)) if tmp + 20 + length - 1 <= len(d22) - 1 else float("nan")
    return float(d22[tmp]), noise, tmp


# --------------------------------------------------------------------------------------------
# velocity-compensated window (experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m) — UNPINNED (Octave only, no stored output)
# --------------------------------------------------------------------------------------------

def interp1_linear(y: np.ndarray, xq: np.ndarray) -> np.ndarray:
    """Octave ``interp1([0:n-1], y, xq)`` (default method 'linear', no extrapolation): NaN outside [0, n-1]."""
    n = len(y)
    xq = np.asarray(xq, dtype=float)
    inside = (xq >= 0) & (xq <= n - 1)
    i0 = np.clip(np.floor(xq).astype(np.int64), 0, n - 2)
    f = xq - i0
    out = y[i0] + (y[i0 + 1] - y[i0]) * f
    out = out.astype(complex if np.iscomplexobj(y) else float)
    out[~inside] = np.nan
    return out


def ranging_vitesse(raw, chips, fs=5e6, sps=2, vitesse=-3.25e-9, n_channels=2, channel=0, band_hz=(96200.0, 106200.0), t0=0.0, dt=0):
    """The window loop of experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m:17-74 for its channel 1 (the returned signal):
    replica = 0/1 chips held ``sps`` samples minus its mean (:7-10), ``d1-mean(d1)`` (:25), carrier from ``fft(d1.^2)`` inside
    ``(freq<106200)&(freq>96200)`` (:31-33), mix (:34-37), THEN linear resampling on the stretched axis
    ``interp1([0:N-1],y,[0:N-1]*1/(1-vitesse)+t0)`` (:40) with the carried offset ``t0=t0+length(y)*vitesse`` (:41) and the edge rule
    ``if isnan(yi(end)) yi(end)=yi(end-1)``, ``if isnan(yi(1)) yi(1)=yi(2)`` (:42-43), ``ifft(fft(yi).*fcode)`` (:46, no interpolation),
    arg-max, the three peak samples, the 3-point polyfit vertex (:56-57), ``indice1=indice1+dt`` (:68) and the wrap of ``t0`` into (-1, 1)
    with ``dt`` following it (:70-71).  Returns one dict per window: 0-based ``indice`` (before ``dt``), ``dt`` as added to this window,
    ``t0`` as used by it, ``solution = indice + 1 + dt + correction`` (the script's 1-based ``solution12``), and ``nan`` when more than the
    two edge samples fell outside the window (the script's ``max`` then returns index 1 of an all-NaN map and its next line fails)."""
    code = make_code_variant(chips, sps=sps, unipolar=True, zero_mean=True)
    fcode = np.conj(_fft(code))
    n = len(code)
    freq = freq_axis(fs, n)
    k = np.nonzero((freq < band_hz[1]) & (freq > band_hz[0]))[0]
    temps = np.arange(n) / fs
    raw = np.asarray(raw).reshape(-1, 2 * n_channels)
    nwin = raw.shape[0] // n
    out = []
    for p in range(nwin):
        d1 = deinterleave(raw[p * n:(p + 1) * n], n_channels, channel)
        d1 = d1 - np.mean(d1)
        tmp, df = coarse_df(d1, k, freq)
        y = d1 * np.exp(-1j * 2 * np.pi * df * temps)
        yi = interp1_linear(y, np.arange(n) * 1 / (1 - vitesse) + t0)
        t0_used = t0
        t0 = t0 + n * vitesse
        if np.isnan(yi[-1]):
            yi[-1] = yi[-2]
        if np.isnan(yi[0]):
            yi[0] = yi[1]
        rec = dict(df=df, df_index=tmp, t0=t0_used, dt=dt)
        if np.isnan(yi).any():
            rec.update(nan=True, indice=0, correction=np.nan, xval=complex(np.nan, np.nan), xvalm1=complex(np.nan, np.nan), xvalp1=complex(np.nan, np.nan), solution=np.nan)
        else:
            prnmap = _ifft(_fft(yi) * fcode)
            indice, _, xval, xvalm1, xvalp1 = peak_refine(prnmap)
            correction = peak_refine_polyfit(prnmap, indice, 1)
            rec.update(nan=False, indice=indice, correction=correction, xval=xval, xvalm1=xvalm1, xvalp1=xvalp1, solution=indice + 1 + dt + correction)
        out.append(rec)
        if t0 >= 1:
            t0 -= 1; dt -= 1
        if t0 <= -1:
            t0 += 1; dt += 1
    return out, t0, dt


# --------------------------------------------------------------------------------------------
# "claudio" production variant (reversed-conjugate convention) — UNPINNED (Octave only)
# --------------------------------------------------------------------------------------------

def processing_claudio(d, df, temps, fcode_claudio, code, Nint=1, ddof=1):
    """acquisition/claudio_aligned_code_ranging_separate.m:49-102 for one code per call
    (len(d) == len(code)); ``fcode_claudio = fft(code)`` (:124).  0-based ``indice``.
    """
    n = len(d)
    r = 2 * Nint + 1
    lo = np.exp(-1j * 2 * np.pi * df * temps)
    y = d * lo
    ffty = _fft(y)
    multmp = fcode_claudio * np.conj(ffty)                       # :59
    pad = np.zeros(r * n, dtype=complex)
    pad[:n // 2] = multmp[:n // 2]
    pad[-(n // 2):] = multmp[-(n // 2):]
    prnmap = _ifft(pad)                                     # :60-61
    yint = np.zeros(r * n, dtype=complex)
    yint[:n // 2] = ffty[:n // 2]
    yint[-(n // 2):] = ffty[-(n // 2):]
    yint = _ifft(yint)                                      # :62-65
    codetmp = np.repeat(code, r)
    indice = int(np.abs(prnmap).argmax())                         # :69
    xval = prnmap[indice]
    xvalm1 = prnmap[indice - 1] if indice >= 1 else prnmap[-1]    # :71-75
    xvalp1 = prnmap[indice + 1] if indice + 2 < r * n else prnmap[0]  # :76-80 (1-based '<')
    correction = (abs(xvalm1) - abs(xvalp1)) / (abs(xvalm1) + abs(xvalp1) - 2 * abs(xval)) / 2
    ind1 = indice + 1                                             # Octave 1-based
    if ind1 > 2:                                                  # :90-94
        rot = np.concatenate((codetmp[ind1 - 2:], codetmp[:ind1 - 2]))
    else:
        rot = codetmp
    yincode = rot * yint
    var = np.var(yincode, ddof=ddof)
    mr, mi = np.mean(yincode.real), np.mean(yincode.imag)
    return dict(indice=indice, correction=correction, xval=xval, xvalm1=xvalm1, xvalp1=xvalp1,
                df=df, SNRr=mr ** 2 / var, SNRi=mi ** 2 / var, puissance=np.var(y, ddof=ddof),
                puissancecode=mr ** 2 + mi ** 2, puissancenoise=var)


def search_df(d, k, df_threshold, freq, temps, fcode_claudio):
    """acquisition/claudio_aligned_code_ranging_separate.m:27-47. ``k``/result 0-based; 0 → -1."""
    n = len(fcode_claudio)
    kbon = -1
    d2 = np.fft.fftshift(np.abs(_fft(d ** 2)))
    ktmp = np.nonzero(d2[k] > np.median(d2[k]) * df_threshold)[0] + k[0]
    if 0 < len(ktmp) < 100:
        for kk in ktmp:
            dftmp = freq[kk] / 2
            lo = np.exp(-1j * 2 * np.pi * dftmp * temps)
            y = d[:n] * lo
            prnmap = np.abs(_ifft(fcode_claudio * np.conj(_fft(y))))
            b = int(prnmap.argmax())
            prnsig = prnmap[b]
            prnmap[max(b - 5, 0):b + 6] = 0
            prnvar = np.var(prnmap, ddof=1)
            if prnsig ** 2 / prnvar > 100:
                kbon = int(kk)
    return kbon


def _oround(x: float) -> int:
    """Octave ``round`` (half away from zero)."""
    return int(np.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def tracked_mode(mode: str, OP: int = 0):
    """The constants that tell the three tracked scripts apart (all else is the same text):
    ``ranging`` acquisition/claudio_aligned_code_ranging_separate.m (remote=0, ranging=1: band :135, 30-s skip :128),
    ``re``      acquisition/claudio_aligned_code_re_separate.m (remote=1, ranging=0: band :137-141, skip :127),
    ``lo``      acquisition/claudio_aligned_code_lo_separate.m (remote=0: band :106, no skip, carrier = arg-max of the
                fresh chunk's squared spectrum over the whole band :126-129, ``floor`` of the lag :134).
    Returns dict(band=(lo, hi) open interval in Hz, carrier, indice_floor, skip_seconds, prefix, code_index) with
    ``code_index`` = the 0-based position in ``dir('n*.bin')`` (:107 / lo :82)."""
    if mode == "ranging":
        return dict(band=(-8000.0, 8000.0), carrier="search_df", indice_floor=False, skip_seconds=30.0,
                    prefix="rangingclaudio", code_index=(OP + 0 + 2) % 2)
    if mode == "re":
        band = (-108000.0, -92000.0) if OP == 1 else (92000.0, 108000.0)
        return dict(band=band, carrier="search_df", indice_floor=False, skip_seconds=30.0,
                    prefix="remoteclaudio", code_index=(OP + 1) % 2)
    if mode == "lo":
        return dict(band=(-20000.0, 20000.0), carrier="chunk_band", indice_floor=True, skip_seconds=0.0,
                    prefix="localclaudio", code_index=OP % 2)
    raise ValueError(mode)


def ranging_tracked(raw, chips, fs=5e6, sps=2, Nint=1, ls_samples=None, band_hz=8000.0, df_threshold=20.0,
                    skip_samples=0, kbon=None, band=None, carrier="search_df", indice_floor=False):
    """Tracked multi-code loop of acquisition/claudio_aligned_code_ranging_separate.m:143-205 — UNPINNED
    (Octave only).  ``raw`` = single-channel interleaved int16 ``[I Q]…`` file contents; ``ls_samples``
    = samples per chunk (``fs*ls`` :157).  1-based bookkeeping is kept as in the script, including
    its quirks: ``indice1`` is the 1-based peak index divided by (2Nint+1) (:168) except after a
    re-alignment, where the raw 1-based index of the second measurement is stored (:179); after the
    carrier search the file is re-read from its START (:153-155), so ``skip_samples`` only moves the
    chunk that ``search_df`` sees.  Deviation: where the script would index past the chunk after a
    re-alignment (Octave aborts), the first measurement is kept and the chunk ends.
    ``carrier="chunk_band"`` + ``indice_floor`` restate the ``lo`` sibling
    (acquisition/claudio_aligned_code_lo_separate.m:117-164): no ``search_df``; every chunk takes its carrier from
    the arg-max of ``fftshift(abs(fft(d.^2)))`` of the FRESH chunk over the whole band ``k`` (:126,129, before ``dold``
    is prepended :127), and the stored lag is ``floor(indice/(2*Nint+1))`` (:134).  ``band`` = (lo, hi) open interval in
    Hz (``tracked_mode``); default ±``band_hz``.
    Returns a dict of per-code lists plus ``df`` per chunk, ``kbon`` (0-based), ``moved``/``movedval``.
    """
    raw = np.asarray(raw).reshape(-1)
    code = make_code(chips, sps)
    n = len(code)
    fc = make_fcode(code, "claudio")
    temps = np.arange(n) / fs
    L = int(ls_samples if ls_samples is not None else fs * 2)
    freq = np.linspace(-fs / 2, fs / 2 - 1.0, L)                    # :131 (fs/fs = 1)
    if band is None:
        band = (-band_hz, band_hz)
    k = np.nonzero((freq < band[1]) & (freq > band[0]))[0]          # :134-141 / lo :105-113
    r = 2 * Nint + 1
    out = dict(xval=[], indice1=[], correction1=[], SNR1r=[], SNR1i=[], puissance1=[], df=[], moved=[], movedval=[], kbon=-1)
    pos = skip_samples * 2
    dold = np.zeros(0, dtype=complex)
    df_found = kbon is not None or carrier == "chunk_band"
    if kbon is not None:
        out["kbon"] = int(kbon)
    p = 1
    guard = 0
    while True:
        chunk = raw[pos:pos + 2 * L]
        pos += 2 * L
        if len(chunk) != 2 * L:
            break
        d = chunk[0::2].astype(np.float64) + 1j * chunk[1::2].astype(np.float64)
        if not df_found:
            kb = search_df(d, k, df_threshold, freq, temps, fc)
            if kb >= 0:
                df_found = True
                out["kbon"] = kb
            chunk = raw[0:2 * L]                                       # fclose/fopen/fread :153-155
            pos = 2 * L
            d = chunk[0::2].astype(np.float64) + 1j * chunk[1::2].astype(np.float64)
            guard += 1
            if not df_found and guard > 2:                             # the script would loop forever here
                break
        if df_found:
            if carrier == "chunk_band":                                # lo :126-129
                d2 = np.fft.fftshift(np.abs(_fft(d ** 2)))
                df = freq[int(np.argmax(d2[k])) + k[0]] / 2
                d = np.concatenate((dold, d))
            else:
                kb = out["kbon"]
                d = np.concatenate((dold, d))
                d2 = np.fft.fftshift(np.abs(_fft(d ** 2)))
                j = int(np.argmax(d2[kb - 3:kb + 4]))
                df = freq[j + kb - 3] / 2
            out["df"].append(df)
            dindex = 1.0
            while True:
                s0 = _oround(dindex)
                dpart = d[s0 - 1:s0 - 1 + n]
                dpart = dpart - dpart.mean()
                o = processing_claudio(dpart, df, temps, fc, code, Nint=Nint, ddof=1)
                ind = (o["indice"] + 1) / r
                if indice_floor:                                       # lo :134
                    ind = float(np.floor(ind))
                stop = False
                if 10 * np.log10(o["SNRi"] + o["SNRr"]) > -30:
                    if (43 < ind < n / 2) or (n / 2 < ind < n - 2):
                        out["moved"].append(p)
                        out["movedval"].append(ind + 1)
                        dnew = dindex
                        if dnew - ind + 1 < 0:
                            dnew += n
                        dnew = dnew - ind + 21
                        s1 = _oround(dnew)
                        if s1 >= 1 and s1 - 1 + n <= len(d):
                            dindex = dnew
                            dpart = d[s1 - 1:s1 - 1 + n]
                            dpart = dpart - dpart.mean()
                            o = processing_claudio(dpart, df, temps, fc, code, Nint=Nint, ddof=1)
                            ind = float(o["indice"] + 1)
                        else:
                            stop = True
                out["xval"].append(o["xval"]); out["indice1"].append(ind); out["correction1"].append(o["correction"])
                out["SNR1r"].append(o["SNRr"]); out["SNR1i"].append(o["SNRi"]); out["puissance1"].append(o["puissance"])
                p += 1
                dindex += n
                if stop or dindex + n - 1 > len(d):
                    break
            dold = d[_oround(dindex) - 1:] if dindex < len(d) else np.zeros(0, dtype=complex)
    return out


# --------------------------------------------------------------------------------------------
# Two-way combination, acquisition/go_1s.m — UNPINNED (Octave only; no recorded result files in the reference)
# --------------------------------------------------------------------------------------------

def _o_find(mask):
    """Octave ``find`` → 1-based indices."""
    return np.nonzero(np.asarray(mask))[0] + 1


def _o_idx(v, k1):
    """``v(k)`` with 1-based index vector ``k``."""
    return np.asarray(v)[np.asarray(k1, dtype=int) - 1]


def _o_polyfit_yf(x, y, n):
    """``[~,S]=polyfit(x,y,n); S.yf`` (fitted values)."""
    return np.polyval(np.polyfit(np.asarray(x, dtype=float), np.asarray(y, dtype=float), n), np.asarray(x, dtype=float))


def go_1s_session(op_lo, op_re, lt_lo, lt_re, fs=5e6, N=1):
    """The per-session body of acquisition/go_1s.m:77-268 on the four result records already loaded (dicts with the
    tracked scripts' row vectors ``xval1 indice1 correction1 SNR1r SNR1i``): OP local (:78-101), OP remote (:103-124),
    LTFB local (:137-150), LTFB remote (:152-174), equalised lengths (:176-182), ``res2`` from the quadratic fits
    (:183-190), ``res`` (:192-194), the code-ambiguity shifts exactly as written (:208-211), statistics (:234-241,
    269-274: ``pkg load nan`` is active :16, so mean/median/std skip NaN) and the rows of the ``<MJD>.1s`` file
    (:251-268, without the date column: ``cpt`` instead).  1-based index vectors are kept as in the script.
    Returns None where the script skips the session (:102 ``length(oplo)>102``, a missing variable)."""
    r = 2 * N + 1
    xval1, indice1, correction1 = (np.asarray(op_lo[k]).reshape(-1) for k in ("xval1", "indice1", "correction1"))
    k = _o_find(np.abs(xval1) > np.max(np.abs(xval1)) / 2)                     # :80
    kk = _o_find(np.diff(k) > 1)                                               # :81
    if kk.size:
        k = k[11 - 1:kk[0]]                                                    # :83  k(11:kk(1))
    else:
        k = k[11 - 1:len(k) - 1]                                               # :86  k(11:end-1)
    oplo = (_o_idx(indice1, k) + _o_idx(correction1, k) / r) / fs * 1e9        # :88,90
    kk = _o_find(np.abs(np.diff(oplo)) > 2)                                    # :94
    if kk.size:
        kk = kk[0]
        if kk > 1:
            oplo = oplo[:kk - 1]                                               # :99
    if not len(oplo) > 102:                                                    # :102
        return None
    xval1, indice1, correction1 = (np.asarray(op_re[q]).reshape(-1) for q in ("xval1", "indice1", "correction1"))
    xv = _o_idx(xval1, k)                                                      # :107
    kkk = _o_find(np.abs(xv) > np.max(np.abs(xv)) / 2)                         # :109
    kkkk = _o_find(np.diff(kkk) > 1)
    if kkkk.size:                                                              # :111-118
        k = k[:kkkk[0]]
        if kkkk[0] < len(oplo):
            oplo = oplo[:kkkk[0]]
    opre = (_o_idx(indice1, k) + _o_idx(correction1, k) / r) / fs * 1e9        # :120,122
    opre = opre[:len(oplo)]                                                    # :123
    snrop = float(np.median(10 * np.log10(np.abs(_o_idx(op_re["SNR1r"], k) + _o_idx(op_re["SNR1i"], k)) * fs)))   # :124
    xval1, indice1, correction1 = (np.asarray(lt_lo[q]).reshape(-1) for q in ("xval1", "indice1", "correction1"))
    k = _o_find(np.abs(xval1) > np.max(np.abs(xval1)) / 2)                     # :139
    kk = _o_find(np.diff(k) > 1)
    if kk.size:
        k = k[11 - 1:kk[0]]                                                    # :142
    else:
        k = k[11 - 1:len(k) - 1]                                               # :145
    ltlo = (_o_idx(indice1, k) + _o_idx(correction1, k) / r) / fs * 1e9        # :147,149
    xval1, indice1, correction1 = (np.asarray(lt_re[q]).reshape(-1) for q in ("xval1", "indice1", "correction1"))
    xv = _o_idx(xval1, k)                                                      # :158
    kkk = _o_find(np.abs(xv) > np.max(np.abs(xv)) / 2)                         # :159
    kkkk = _o_find(np.diff(kkk) > 1)
    if kkkk.size:                                                              # :161-165
        k = k[:kkkk[0]]
        ltlo = ltlo[:kkkk[0]]
    if len(kkk) < len(ltlo):                                                   # :166-170
        k = k[:kkk[-1]]
        ltlo = ltlo[:kkk[-1]]
    ltre = (_o_idx(indice1, k) + _o_idx(correction1, k) / r) / fs * 1e9        # :171,173
    snrlt = float(np.median(10 * np.log10(np.abs(_o_idx(lt_re["SNR1r"], k) + _o_idx(lt_re["SNR1i"], k)) * fs)))   # :174
    if len(oplo) > len(ltlo):                                                  # :176-182
        oplo = oplo[:len(ltlo)]
        opre = opre[:len(ltlo)]
    else:
        ltlo = ltlo[:len(oplo)]
        ltre = ltre[:len(oplo)]
    res2 = None
    if len(opre) > 2 and len(ltre) > 2:                                        # :183-190
        x = np.arange(1, len(opre) + 1)
        res2 = 0.5 * ((_o_polyfit_yf(x, opre, 2) - oplo) - (_o_polyfit_yf(x, ltre, 2) - ltlo))
        res2[np.abs(res2 - np.nanmedian(res2)) > 5] = np.nan
    res = 0.5 * ((opre - oplo) - (ltre - ltlo))                                # :192
    res[np.abs(res - np.nanmedian(res)) > 5] = np.nan                          # :193-194
    ki = res > np.nanmedian(res) + 10                                          # :208-211 (as written: the second test hits every element)
    res[ki] = res[ki] - 200 / r
    ki = res > np.nanmedian(res) - 10
    res[ki] = res[ki] + 200 / r
    out = dict(oplo=oplo, opre=opre, ltlo=ltlo, ltre=ltre, res=res, res2=res2, snrop=snrop, snrlt=snrlt,
               resmean=float(np.nanmean(res)), resstd=float(np.nanstd(res, ddof=1)))                              # :234,239
    rows = []
    cpt = 0
    for kq in range(1, len(opre) - 25 + 1, 25):                                # :255  k=1:25:length(opre)-25
        x = np.arange(kq - 1, kq + 25 - 2 + 1) / 25                            # [k-1:k+25-2]/25
        row = [float(cpt)]
        for series in (oplo, opre, ltlo, ltre):                                # :256-263  yf(13)
            row.append(float(_o_polyfit_yf(x, series[kq - 1:kq + 25 - 1], 1)[13 - 1]))
        rows.append(row)
        cpt += 1
    out["rows"] = np.array(rows).reshape(-1, 5)
    res25 = np.convolve(res, np.ones(25) / 25)                                 # :269  conv(res,ones(25,1)/25)(25:end-25)
    res25 = res25[25 - 1:len(res25) - 25]
    out["resmean25"] = float(np.nanmean(res25)) if res25.size else float("nan")
    out["resstd25"] = float(np.nanstd(res25, ddof=1)) if res25.size > 1 else float("nan")
    if len(opre) > 2:                                                          # :275-277
        t = np.arange(len(opre)) / 25
        out["opslope"] = np.polyfit(t, opre, 1)
        out["ltslope"] = np.polyfit(t, ltre, 1)
    return out


def go_1s_text(mjd: float, rows) -> str:
    """The ``<MJD>.1s`` file body, go_1s.m:252-267 (``fprintf`` formats verbatim)."""
    txt = "# MJD\t\tOPlocal\tOPremote\tLTFBlocal\tLTBBremote\n"
    for row in rows:
        txt += "%f\t%f\t%f\t%f\t%f\n" % (mjd + row[0] / 86400, row[1], row[2], row[3], row[4])
    return txt


# --------------------------------------------------------------------------------------------
# delay × Doppler cross-ambiguity (experiments/231001_DLL_PLL/rxcomplex.cpp) — UNPINNED
# --------------------------------------------------------------------------------------------

def caf_bins(y0, fcode, fs, f_lo, f_hi, f_step):
    """Per-Doppler-bin circular xcorr peak, the loop of rxcomplex.cpp:534-563 restated on the
    godual replica (Nint=0): for each trial offset mix → FFT → × fcode → IFFT → arg-max.
    ``y0`` is the mean-removed window.  Returns (freqs, peak |.|, peak lag)."""
    n = len(y0)
    t = np.arange(n) / fs
    nb = int(np.floor((f_hi - f_lo) / f_step + 1e-9)) + 1
    freqs = f_lo + f_step * np.arange(nb)
    pk = np.empty(nb)
    lag = np.empty(nb, dtype=np.int64)
    for i, f in enumerate(freqs):
        y = y0 * np.exp(-1j * 2 * np.pi * f * t)
        m = np.abs(_ifft(_fft(y) * fcode))
        lag[i] = int(m.argmax())
        pk[i] = m[lag[i]]
    return freqs, pk, lag


def caf_bins_shift(y0, fcode, k_lo, k_hi):
    """Same surface on the integer-bin Doppler grid f = k*fs/N: a frequency shift by k bins is a
    circular shift of FFT(y) by k, so one forward FFT serves all bins."""
    Y = _fft(y0)
    ks = np.arange(k_lo, k_hi + 1)
    pk = np.empty(len(ks))
    lag = np.empty(len(ks), dtype=np.int64)
    for i, kk in enumerate(ks):
        m = np.abs(_ifft(np.roll(Y, -kk) * fcode))
        lag[i] = int(m.argmax())
        pk[i] = m[lag[i]]
    return ks, pk, lag


def sliding_dot(y, code, nlag):
    """Direct sliding dot products for ±nlag lags (the ``cblas_dgemm`` of rxcomplex.cpp:605 with
    ``PRN_mapping`` :989-999 replicas): out[l] = (1/n) * sum_i y[i] * code[(i - (l-nlag)) mod n]."""
    n = len(y)
    out = np.empty(2 * nlag + 1, dtype=complex)
    for li in range(2 * nlag + 1):
        out[li] = np.dot(y, np.roll(code, li - nlag)) / n
    return out


# --------------------------------------------------------------------------------------------
# FIR front end (no reference twin: parameters from experiments/2403/zmq_rx.py:208-215) — UNPINNED
# --------------------------------------------------------------------------------------------

def fir_lowpass_hamming(fs, cutoff, transition):
    """GNU Radio ``firdes.low_pass(1, fs, cutoff, transition, WIN_HAMMING)`` design rule:
    ntaps = int(53 * fs / (22 * transition)) made odd; windowed sinc, unit DC gain."""
    ntaps = int(53.0 * fs / (22.0 * transition))
    if ntaps % 2 == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    n = np.arange(-m, m + 1)
    w = 0.54 - 0.46 * np.cos(2 * np.pi * (n + m) / (ntaps - 1))
    fw = 2 * np.pi * cutoff / fs
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(n == 0, fw / np.pi, np.sin(n * fw) / (n * np.pi)) * w
    return h / h.sum()


def fir_decimate(x, taps, dec):
    """y[m] = sum_j taps[j] * x[m*dec + j]  ('valid' part, direct form)."""
    nout = (len(x) - len(taps)) // dec + 1
    idx = np.arange(nout)[:, None] * dec + np.arange(len(taps))[None, :]
    return (x[idx] * taps[None, ::1]).sum(axis=1)


# --------------------------------------------------------------------------------------------
# Acquisition stage of experiments/231001_DLL_PLL/rxcomplex.cpp, restated function by function — UNPINNED
# (the program needs fftw3 / gsl / cblas, none of which exists in the build image; its own results are not in the
# repository).  Names follow the C++ functions; all arrays complex128 / float64.
# --------------------------------------------------------------------------------------------

RX_NINTERP = 2      # #define Ninterp 2, rxcomplex.cpp:29


def rx_short2double(smp, nobs: int, dec: int = 1):
    """``short2double`` rxcomplex.cpp:914-963 (USRP X310 branch): int16 ``[IA QA IB QB]`` samples -> two complex
    streams of ``nobs`` samples each, i.e. x2 interpolated through the FFT domain.  As written there the lower half of
    the ``nobs/2``-point spectrum stays where it is UNSCALED while the upper half moves to the top of the ``nobs``-point
    spectrum divided by ``nobs/2`` (:931-936) — restated as is; the inverse transform is divided by ``nobs`` (:957-959)."""
    smp = np.asarray(smp).reshape(-1)
    half = nobs // RX_NINTERP
    out = []
    for off in (0, 2):
        tmp = np.zeros(nobs, dtype=np.complex128)
        i = np.arange(half)
        tmp[:half] = smp[4 * i * dec + off] / 32768.0 + 1j * (smp[4 * i * dec + off + 1] / 32768.0)      # :922-928
        tmp[:half] = _fft(tmp[:half])                                                                     # plan1 :931-933
        j = np.arange(half // 2)
        tmp[nobs - j - 1] = tmp[half - j - 1] / float(half)                                               # :934-936
        tmp[half - j - 1] = 0.0                                                                           # :937-938
        tmp = _ifft(tmp) * nobs                                                                           # FFTW backward is unnormalised :940-942
        out.append(tmp / float(nobs))                                                                     # :957-959
    return out[0], out[1]


def rx_short2double_real(smp, nobs: int, dec: int = 1):
    """``short2double`` of the REAL-sample program, experiments/231001_DLL_PLL/rx.cpp:892-900: the I sample of physical
    channel A (offset 0) and B (offset 2) of every frame, /32768; no interpolation."""
    smp = np.asarray(smp).reshape(-1)
    i = np.arange(nobs)
    return smp[4 * i * dec] / 32768.0, smp[4 * i * dec + 2] / 32768.0


def rx_mai_up(nobs: int, ld: int, pt: int, amp, c_wav, pidx, ff: float, pmod, wav):
    """``MAI_up`` rx.cpp:1011-1020: for ``i >= pt``, period ``p=(i-pt)/ld``, code sample ``k=(i-pidx[p]-pt+ld)%ld``:
    ``wav[i] += 0.5*amp[p]*real(c_wav[k])*cos(2 pi (ff*i + pmod[p]))`` (in place)."""
    i = np.arange(pt, nobs)
    p = (i - pt) // ld
    k = (i - np.asarray(pidx)[p] - pt + ld) % ld
    wav[pt:] += 0.5 * np.asarray(amp)[p] * np.asarray(c_wav).real[k] * np.cos(2.0 * 3.141592653589793 * (ff * i.astype(np.float64) + np.asarray(pmod)[p]))
    return wav


def rx_mai_out(inp, out):
    """``MAI_out`` rx.cpp:1022-1027: ``out = in - out``."""
    return np.asarray(inp) - out


def rx_prn_sampling(nobs: int, code, rc: float, fs: float, clen: int, delay_ns: float = 0.0):
    """``PRN_sampling`` :965-978: ``idx=floor(fmod((i/fs-delay*1e-9)*rc, clen))`` (wrapped), value ``code[idx]``."""
    i = np.arange(nobs, dtype=np.float64)
    idx = np.floor(np.fmod((i / fs - delay_ns * 1.0e-9) * float(rc), float(clen))).astype(np.int64)
    idx = np.where(idx < 0, idx + clen, np.where(idx >= clen, idx - clen, idx))
    return np.asarray(code, dtype=np.float64)[idx].astype(np.complex128)


def _rx_band(n: int, df: float, fmax: float, fmin: float):
    """Pass-band test shared by ``lowpass`` :1025-1026 and ``cross_spectrum`` :1007-1008: signed bin index,
    ``idx*df < fmax && idx*df > fmin && idx != 0``."""
    i = np.arange(n)
    idx = np.where(i >= n // 2, i - n, i)
    f = idx.astype(np.float64) * df
    return (f < fmax) & (f > fmin) & (idx != 0)


def rx_lowpass(obs, df: float, fmax: float, fmin: float):
    """``lowpass`` :1020-1037: brick wall on a spectrum, pass-band scaled by 1/nobs."""
    n = len(obs)
    return np.where(_rx_band(n, df, fmax, fmin), obs / float(n), 0.0)


def rx_replica(code_pm1, nobs: int, nfft: int, rc: float, fs: float, clen: int, fltmax: float, fltmin: float, dec_a: int = 1):
    """Channel set-up :414-437.  Returns (``dev_wav_acq`` = FFT of the zero-padded sampled code (real part only is
    copied, ``memcpy_acq`` :980-987, BEFORE the filter is applied to ``dev_wav_t``), ``psbb`` = mean power of the
    low-pass filtered waveform :431-432, the filtered waveform itself)."""
    wav_t = rx_prn_sampling(nobs, code_pm1, rc, fs, clen, 0.0)                                   # :416
    wav_acq = np.zeros(nfft, dtype=np.complex128)                                                   # :418
    m = nobs // dec_a
    wav_acq[:m] = wav_t[:m * dec_a:dec_a].real                                                      # :420, :984-986
    filt = _ifft(rx_lowpass(_fft(wav_t), fs / float(nobs), fltmax, fltmin)) * nobs                 # :422-428 (unnormalised backward)
    psbb = float(np.sqrt(np.sum(np.abs(filt) ** 2)) ** 2 / float(nobs))                            # :431-432 (dznrm2 squared / nobs)
    return _fft(wav_acq), psbb, filt                                                                # :434-437


def rx_downconv_acq(nfft: int, ff: float, phi: float, smp, dec: int = 1):
    """``downconv_acq`` :1039-1049: ``sqrt(2)*smp[i*dec]*exp(-2 pi j (ff*i+phi))`` with the constant as written."""
    i = np.arange(nfft, dtype=np.float64)
    ang = -1.0 * 2.0 * 3.141592653589793 * (ff * i + phi)                                          # #define PI :31
    x = np.asarray(smp)[:nfft * dec:dec]
    return 1.4142135624 * (x.real * np.cos(ang) - x.imag * np.sin(ang)) + 1j * 1.4142135624 * (x.real * np.sin(ang) + x.imag * np.cos(ang))


def rx_cross_spectrum(obs, prn, df: float, fmax: float, fmin: float):
    """``cross_spectrum`` :1001-1018: ``obs*conj(prn)/n^2`` inside the pass-band, 0 elsewhere."""
    n = len(obs)
    return np.where(_rx_band(n, df, fmax, fmin), obs * np.conj(prn) / float(n) / float(n), 0.0)


def rx_izamax(x) -> int:
    """``cblas_izamax`` :553: first index of the largest ``|re|+|im|`` (BLAS cabs1), 0-based."""
    return int(np.argmax(np.abs(x.real) + np.abs(x.imag)))


def rx_acq_bin(smp, idx: int, fcc: float, wav_acq_f, nfft: int, fs: float, fltmax: float, fltmin: float, dec_a: int = 1):
    """Body of the Doppler loop :543-556 for one trial carrier ``fcc``: (pk, pk_idx)."""
    obs = rx_downconv_acq(nfft, fcc / (fs / float(dec_a)), 0.0, np.asarray(smp)[idx:], dec_a)      # :543
    obs = _fft(obs)                                                                                  # :544-546
    obs = rx_cross_spectrum(obs, wav_acq_f, (fs / float(dec_a)) / float(nfft), fltmax, fltmin)       # :548
    obs = _ifft(obs) * nfft                                                                          # :549-551 (unnormalised backward)
    pk_idx = rx_izamax(obs)                                                                          # :553
    return float(abs(obs[pk_idx])), pk_idx                                                           # :554 (dznrm2 of one element)


def rx_acquire(smp, idx: int, wav_acq_f, nobs: int, nfft: int, fs: float, fc_init: float, frange: float, fstep: float,
               fltmax: float, fltmin: float, dec_a: int = 1):
    """Acquisition sweep :521-567 (the random code-aligned offset ``idx`` of :529 is an argument): coarse sweep
    fc±range in ``step``, keep the strictly largest peak, then halve the step with range = step until step < 1 Hz.
    Returns (fc, pk, pt) with ``pt = pk_idx % (nobs/dec_a)`` :561."""
    pk_best, fc, pt = 0.0, float(fc_init), 0
    while True:
        flow, fhigh = fc - frange, fc + frange                       # :536-537
        fcc = flow
        while fcc <= fhigh:                                          # :538
            pk, pk_idx = rx_acq_bin(smp, idx, fcc, wav_acq_f, nfft, fs, fltmax, fltmin, dec_a)
            if pk > pk_best:                                         # :556-562
                fc, pk_best, pt = fcc, pk, pk_idx % (nobs // dec_a)
            fcc += fstep
        fstep = fstep / 2.0                                          # :565-567
        frange = fstep
        if fstep < 1.0:
            break
    return fc, pk_best, pt


def rx_gate(pk: float, psbb: float, px: float, snr_min: float):
    """:570-573: peak power ``8*pk^2/psbb`` and the lock test ``(1+snr_min)*pk > snr_min*px``."""
    p = 8.0 * pk * pk / psbb
    return p, (1.0 + snr_min) * p > snr_min * px


def rx_power(smp, fs: float, dec_a: int = 1):
    """Received power :481-489: ``zdotc(smp, smp)`` over every ``dec_a``-th sample divided by ``fs/dec_a``."""
    x = np.asarray(smp)[::dec_a]
    return float(np.real(np.vdot(x, x))) / (fs / float(dec_a))


# --------------------------------------------------------------------------------------------
# Tracking epoch of experiments/231001_DLL_PLL/rxcomplex.cpp:593-745, restated function by function — UNPINNED
# --------------------------------------------------------------------------------------------

def rx_prn_mapping(snobs: int, knobs: int, prn):
    """``PRN_mapping`` :989-999 for ``nobs = snobs*(2*knobs+1)``: row ``l`` of the (2*knobs+1) x snobs replica matrix is the
    waveform delayed by ``l-knobs`` samples (circular): ``pn[l*snobs+i] = real(prn[(i - (l-knobs)) mod snobs])``."""
    prn = np.asarray(prn)
    out = np.empty((2 * knobs + 1, snobs))
    i = np.arange(snobs)
    for l in range(2 * knobs + 1):
        idx = i - (l - knobs)
        idx = np.where(idx >= snobs, idx - snobs, idx)
        idx = np.where(idx < 0, idx + snobs, idx)
        out[l] = prn[idx].real
    return out


def rx_downconv_trk(nobs: int, ld: int, ff: float, phi: float, smp):
    """``downconv_trk`` :1051-1061: ``sqrt(2)*smp[i]*exp(-2 pi j (ff*i+phi))`` for ``nobs`` samples, returned as the complex
    matrix [nobs/ld, ld] (the program stores real and imaginary rows interleaved for the dgemm)."""
    i = np.arange(nobs, dtype=np.float64)
    ang = -1.0 * 2.0 * 3.141592653589793 * (ff * i + phi)
    x = np.asarray(smp)[:nobs]
    re = 1.4142135624 * (x.real * np.cos(ang) - x.imag * np.sin(ang))
    im = 1.4142135624 * (x.real * np.sin(ang) + x.imag * np.cos(ang))
    return (re + 1j * im).reshape(nobs // ld, ld)


def rx_kth_smallest(a, k: int) -> float:
    """``kth_smallest`` :840-865 (Wirth's selection) returns the element of rank ``k`` of the array = ``sorted(a)[k]``."""
    return float(np.sort(np.asarray(a, dtype=np.float64))[k])


def rx_fit_wlinear(x, w, y):
    """``gsl_fit_wlinear(x,1,w,1,y,1,n,&c0,&c1,...,&chisq)``: weighted least squares over the entries with ``w > 0``."""
    x, w, y = (np.asarray(v, dtype=np.float64) for v in (x, w, y))
    m = w > 0
    W = w[m].sum()
    xm, ym = (w[m] * x[m]).sum() / W, (w[m] * y[m]).sum() / W
    dx, dy = x[m] - xm, y[m] - ym
    c1 = (w[m] * dx * dy).sum() / (w[m] * dx * dx).sum()
    c0 = ym - xm * c1
    return float(c0), float(c1), float((w[m] * (y[m] - (c0 + c1 * x[m])) ** 2).sum())


def rx_track_epoch(smp, wav_t, st: dict, nobs: int, bps: int, nlag: int, fs: float):
    """One pass of the tracking branch :589-745 for a locked channel.  ``smp`` = the complex sample stream (``ci.dev_smp``),
    ``wav_t`` = the sampled (and filtered) code waveform of ``nobs`` samples (``ci.dev_wav_t``), ``st`` = dict with the
    ``channel_info`` fields ``fc pt last_phi psbb duration`` (updated in place as the program does).  Returns the printed
    quantities (``freq phi cnt gd dg sdgd pk``) or None where ``cnt*2 <= bps`` (:667)."""
    pt, fc = int(st["pt"]), float(st["fc"])
    nl = 2 * nlag + 1
    alpha = 1.0 / float(nobs)                                                         # :593
    phi = float(np.fmod(float(pt) * fc / fs, 1.0))                                    # :594
    obs = rx_downconv_trk(nobs * (bps - 1), nobs, fc / fs, phi, np.asarray(smp)[pt:])   # :602
    wav = rx_prn_mapping(nobs, nlag, wav_t)                                           # :430
    res = alpha * (obs @ wav.T)                                                       # cblas_dgemm :605: [bps-1, nl], re and im columns
    cor = res.real ** 2 + res.imag ** 2                                               # get_cor_and_phi :1063-1072
    ph = np.arctan2(res.imag, res.real) / 2.0 / 3.141592653589793
    res_gd, res_phi, ps, w, ttag_phi, res_amp = (np.zeros(bps) for _ in range(6))
    pk_idx = np.zeros(bps, dtype=np.int64)
    cnt = 0
    for p in range(bps - 1):
        k = int(np.argmax(np.abs(cor[p])))                                            # cblas_idamax :630
        ttag_phi[p] = float(p) * st["duration"] + float(pt) / fs                      # :632
        ps[p] = cor[p, k] / st["psbb"]                                                # :633
        if k - 2 >= 0 and k + 2 < nl:                                                 # :634
            res_phi[p] = ph[p, k]
            pk_idx[p] = k - nlag                                                      # rx.cpp:638
            res_amp[p] = np.sqrt(2.0 * cor[p, k]) / st["psbb"]                        # rx.cpp:640
            c = cor[p]
            res_gd[p] = ((c[k - 1] - c[k + 1]) / (c[k - 1] - 2.0 * c[k] + c[k + 1])
                         - (c[k - 2] - c[k + 2]) / (c[k - 2] - 2.0 * c[k] + c[k + 2])
                         + float(pt + k - nlag)) * 1.0e+9 / fs                        # :649-659
            w[p] = 1.0
            cnt += 1
    st["cnt_last"] = cnt                                                              # what the "lock lost" line prints (:787)
    if not cnt * 2 > bps:                                                             # :667
        return None
    raw_phi = res_phi.copy()                                                          # rx.cpp:664 (the real-sample program's SIC records)
    sel = [res_gd[p] for p in range(bps) if w[p] > 0.0]                               # :692-698
    ii = len(sel)
    c0 = rx_kth_smallest(sel, ii // 2)                                                # :699
    stddev = (rx_kth_smallest(sel, ii * 3 // 4) - rx_kth_smallest(sel, ii // 4)) / 1.349
    cnt = 0
    for p in range(bps - 1):                                                          # :703-716
        if w[p] != 0.0:
            if abs(res_gd[p] - c0) < 3.0 * stddev:
                cnt += 1
                while abs(res_phi[p] - st["last_phi"]) > 0.25:
                    if res_phi[p] > st["last_phi"]:
                        res_phi[p] -= 0.5
                    else:
                        res_phi[p] += 0.5
                st["last_phi"] = res_phi[p]
            else:
                w[p] = 0.0
    c0, c1, _ = rx_fit_wlinear(ttag_phi, w, res_phi)                                  # :728
    st["fc_prev"] = st["fc"]
    st["fc"] = st["fc"] + round(c1)                                                   # :730 (C round: half away from zero; c1 is never a tie here)
    st["df"] = c1 - round(c1)
    st["phi"] = float(np.fmod(c0 + 1000.0, 1.0))
    ttag_gd = np.arange(bps, dtype=np.float64) * st["duration"]                       # :410
    g0, g1, chisq = rx_fit_wlinear(ttag_gd, w, res_gd)                                # :739
    out = dict(freq=st["fc"] + st["df"], phi=st["phi"], cnt=cnt, sdgd=float(np.sqrt(chisq / float(cnt))),
               gd=g0 + 0.5 * g1, dg=g1, pk=float(np.mean(ps[w > 0.0])))              # :740-742, average() :887-901
    st["pt_prev"] = pt
    st["pt"] = int(np.floor((g0 + g1) * fs / 1.0e+9 + 0.5)) if (g0 + g1) >= 0 else -int(np.floor(-(g0 + g1) * fs / 1.0e+9 + 0.5))   # round() :744
    # rx.cpp:664-666,752-757: what MAI_up reads — peak lags, amplitudes, and the raw phases with the carrier update taken out
    rec = np.zeros(bps)
    pp = np.arange(bps - 1)
    rec[:bps - 1] = raw_phi[:bps - 1] - (st["fc"] + st["df"] - st["fc_prev"]) * ((pp + 1) * nobs + st["pt_prev"]).astype(np.float64) / fs
    st["mai"] = dict(pk_idx=pk_idx, amp=res_amp, phase=rec)
    return out


# --------------------------------------------------------------------------------------------
# The receiver as a program, experiments/231001_DLL_PLL/rxcomplex.cpp:263-835, and (real=True) its real-sample sibling rx.cpp with
# the successive interference cancellation of rx.cpp:505-518 — UNPINNED (same reason as above): parameter
# parser, per-channel set-up, the per-second loop with the acquisition -> tracking hand-over, the text of the .dat rows and of
# the rxcomplex.log lines.  Checker of twx_rx_* (tests/test_gpu_rx.py, tests/test_rx_host.py).
# --------------------------------------------------------------------------------------------

def rx_v2todbm(v2: float) -> float:
    """``v2todBm`` :1236-1240."""
    return 10.0 * np.log10(v2 * 1000.0 / 25.0) if v2 > 0.0 else 0.0


def _c_sscanf(line: str, fmt: str):
    """``sscanf(line, fmt)`` for the conversions the program uses (``%s %d %lf`` separated by blanks): every directive skips white
    space; ``%s`` takes the run of non-blank characters, ``%d`` the longest decimal-integer prefix (strtol), ``%lf`` the longest
    floating-point prefix (strtod: digits, point, exponent, inf / nan) — and the NEXT directive goes on right behind it, inside the
    same token if something is left of it ("100.5" read with ``%d %lf`` gives 100 and .5).  Returns the converted values; the scan
    stops at the first directive that cannot convert."""
    import re
    out, pos = [], 0
    num = re.compile(r"[+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|[iI][nN][fF](?:[iI][nN][iI][tT][yY])?|[nN][aA][nN])")
    for d in fmt.split():
        while pos < len(line) and line[pos] in " \t\n\r\v\f":
            pos += 1
        if pos >= len(line):
            break
        if d == "%s":
            m = re.compile(r"\S+").match(line, pos)
            out.append(m.group(0))
        elif d == "%d":
            m = re.compile(r"[+-]?\d+").match(line, pos)
            if not m:
                break
            out.append(int(m.group(0)))
        elif d == "%lf":
            m = num.match(line, pos)
            if not m:
                break
            out.append(float(m.group(0)))
        else:
            raise ValueError(d)
        pos = m.end()
    return out


def rx_parse_param(lines):
    """The parameter-file loop :263-296: '#' lines skipped (:267), 9 tokens split on " ;\r\n" (:270-273), ``str[0]`` in A/B and
    ``str[2]`` in N/S (:274), the ``sscanf`` of :280 — restated with C's conversion rules (``_c_sscanf``: it splits on white space
    only, and a numeric directive stops where its number stops, so a row written with ';', or an integer field written "100.5",
    scans differently from how it tokenises) — and the value ranges of :288 (rows outside them are skipped).  A row whose scan does
    not yield all nine values is skipped (the program goes on with the previous row's values there — not restated).  Lines are cut at
    199 characters as ``fgets(str, 200, …)`` cuts them (:265).  Returns dicts ``ch mode pn fc_init kcps fltkhz frange fstep snr_min_db``."""
    import re
    rows = []
    pieces = []
    for line in lines:                                                   # fgets(str, 200): a longer line comes in pieces of 199 characters
        while len(line) > 199:
            pieces.append(line[:199])
            line = line[199:]
        if line:
            pieces.append(line)
    for line in pieces:
        if line[:1] == "#":
            continue
        toks = [t for t in re.split(r"[ ;\r\n]+", line) if t]
        if len(line) < 3 or line[0] not in "AB" or line[2] not in "NS" or len(toks) != 9:
            continue
        v = _c_sscanf(line, "%s %s %d %lf %d %lf %lf %lf %lf")
        if len(v) != 9:
            continue
        _, _, pn, fc_init, kcps, fltkhz, frange, fstep, snr = v
        if not (0 <= pn <= 131 and kcps == 2500 and -200000.0 <= fc_init < 200000.0 and 0.0 <= frange < 200000.0 and frange > fstep and snr > -100.0):
            continue
        rows.append(dict(ch=line[0], mode=line[2], pn=pn, fc_init=fc_init, kcps=kcps, fltkhz=fltkhz, frange=frange, fstep=fstep, snr_min_db=snr))
    return rows


def rx_channel_setup(row: dict, code_bytes, sps: int, dec_a: int = 1) -> dict:
    """``channel_info`` after :290-437 for one accepted row; ``code_bytes`` = the file SDRcode reads (:866-884), 0/1 per chip."""
    ci = dict(is_chA=row["ch"] == "A", cid=row["pn"], rc=row["kcps"] * 1000, fc_init=row["fc_init"], is_sic=row.get("mode", "N") == "S")
    shown = ci["cid"] + 50 if ci["is_sic"] else ci["cid"]                             # rx.cpp:442,708: SIC rows print PRN + 50
    ci["shown"] = shown
    if row["pn"] < 100:
        ci.update(clen=10000, duration=0.004, nlag=14)                                # :299-304
    else:
        ci.update(clen=100000, duration=0.04, nlag=28)                                # :305-311
    fs = float(sps)
    ci["bps"] = ci["rc"] // ci["clen"]                                               # :363
    ci["nobs"] = sps // ci["bps"]                                                     # :364
    ci["fltmax"], ci["fltmin"] = float(ci["rc"]), -float(ci["rc"])                    # :367-368
    nfft = 1
    while True:                                                                        # :369-374
        nfft *= 2
        if nfft > ci["nobs"] * 2 // dec_a:
            break
    ci["nfft"] = nfft
    rng = 1.0
    while rng < row["frange"]:
        rng *= 2.0                                                                     # :375-376
    stp = 1.0
    while stp < row["fstep"]:
        stp *= 2.0                                                                     # :377-378
    ci.update(range=rng, step=stp, snr_min=10.0 ** (row["snr_min_db"] / 10.0))         # :379
    ci.update(is_trk=False, is_first=False, df=0.0, last_phi=0.0, fc=0.0, pt=0, pk=0.0, gd=0.0, dg=0.0, sdgd=0.0, cnt=0)
    code = 1 - 2 * np.asarray(code_bytes[:ci["clen"]], dtype=np.int64)                 # host_code :879
    wav_acq_f, psbb, filt = rx_replica(code, ci["nobs"], nfft, float(ci["rc"]), fs, ci["clen"], ci["fltmax"], ci["fltmin"], dec_a)
    ci.update(wav_acq_f=wav_acq_f, psbb=psbb, wav_t=filt)
    ci["dat_name"] = "ch%s.pn%02d.%dkcps.dat" % ("A" if ci["is_chA"] else "B", shown, ci["rc"] // 1000)     # :720
    ci["log_set"] = "set param   : Ch. %s, PRN#%2d, %8.0f %4d %5.0f %5.0f %5.0f %3.0f\n" % (
        "A" if ci["is_chA"] else "B", shown, ci["fc_init"], ci["rc"] // 1000, ci["fltmax"] * 1.0e-3, ci["range"], ci["step"], ci["snr_min"])   # :441
    return ci


def rx_second(cis, raw_second, sps: int, dec_a: int, acq_idx, real: bool = False):
    """One pass of the loop body :468-832.  ``raw_second``: ``sps/Ninterp`` frames ``[IA QA IB QB]`` of int16; ``acq_idx(i, ci)``
    returns the sample offset the program draws with rand() (:529).  Returns one dict per channel: ``status`` (the names of the
    TWX_RX_* states), the ``channel_info`` values after the pass, ``dat_row`` / ``log`` texts where the program writes them."""
    fs = float(sps)
    if real:                                                                                  # rx.cpp:478 — and, below, its SIC block :505-518
        smp_a, smp_b = rx_short2double_real(np.asarray(raw_second).reshape(-1), sps)
    else:
        smp_a, smp_b = rx_short2double(np.asarray(raw_second).reshape(-1), sps)             # :477
    pwr = {True: rx_power(smp_a, fs, dec_a), False: rx_power(smp_b, fs, dec_a)}               # :481-489
    out = []
    for i, ci in enumerate(cis):
        smp = smp_a if ci["is_chA"] else smp_b
        ci["px"] = pwr[ci["is_chA"]]
        if real and ci["is_sic"]:                                                             # rx.cpp:505-518
            mai = np.zeros(sps)
            for k in range(i):
                o = cis[k]
                if o["is_chA"] == ci["is_chA"] and o["cid"] != ci["cid"] and not o["is_sic"] and o["is_trk"] and not o["is_first"]:
                    m = o["mai"]
                    rx_mai_up(sps, o["nobs"], o["pt_prev"], m["amp"], o["wav_t"], m["pk_idx"], (o["fc"] + o["df"]) / fs, m["phase"], mai)
            smp = rx_mai_out(smp, mai)
            ci["px"] = rx_power(smp, fs, dec_a)                                               # rx.cpp:515-516
        chs = "A" if ci["is_chA"] else "B"
        ev = dict(status=None, dat_row="", log="", acq_idx=0)
        if not ci["is_trk"]:                                                                  # :521-586
            idx = int(acq_idx(i, ci))
            fc, pk, pt = rx_acquire(smp, idx, ci["wav_acq_f"], ci["nobs"], ci["nfft"], fs, ci["fc_init"], ci["range"], ci["step"],
                                    ci["fltmax"], ci["fltmin"], dec_a)
            ci["fc"], ci["pt"] = fc, pt
            ci["pk"], locked = rx_gate(pk, ci["psbb"], ci["px"], ci["snr_min"])                 # :570-573
            ev["acq_idx"] = idx
            if locked:
                ci["pt"] = ci["pt"] * dec_a                                                   # :575
                ci["gd"] = float(ci["pt"]) * 1.0e+9 / fs
                ci["is_trk"], ci["is_first"] = True, True
                ev["log"] = "acquisition : Ch. %s, PRN#%2d, %3d %8.0f %7.0f %6d %8.3f %8.3f\n" % (
                    chs, ci["shown"], idx // 2 // ci["nobs"], ci["fc"], ci["gd"], ci["pt"], rx_v2todbm(ci["pk"]), rx_v2todbm(ci["px"]))   # :582
                ev["status"] = "acquired"
            else:
                ev["status"] = "no signal"
        else:                                                                                 # :589-790
            res = rx_track_epoch(smp, ci["wav_t"], ci, ci["nobs"], ci["bps"], ci["nlag"], fs)
            if res is not None:
                ci.update(gd=res["gd"], dg=res["dg"], sdgd=res["sdgd"], pk=res["pk"], cnt=res["cnt"])
                if not ci["is_first"]:
                    ev["dat_row"] = "%14.6f %11.8f %3d %5.3f %14.6f %11.6f %8.4f %7.3f %7.3f\n" % (
                        ci["fc"] + ci["df"], ci["phi"], ci["cnt"], 0.0, ci["gd"], ci["dg"], ci["sdgd"], rx_v2todbm(ci["pk"]),
                        rx_v2todbm(ci["px"] - ci["pk"]))                                        # :735,747,752
                    ev["status"] = "tracked"
                else:
                    ev["log"] = "code lock   : Ch. %s, PRN#%2d, count = %d / %d\n" % (chs, ci["shown"], ci["cnt"], ci["bps"])      # :761
                    ci["is_first"] = False                                                    # :766
                    ev["status"] = "code lock"
            else:
                ev["log"] = "%s : Ch. %s, PRN#%2d, count = %d / %d\n" % ("acq failed " if ci["is_first"] else "lock lost  ", chs, ci["shown"],
                                                                          ci.get("cnt_last", 0), ci["bps"])                      # :779,787
                ev["status"] = "acq failed" if ci["is_first"] else "lock lost"
                ci["is_trk"], ci["last_phi"] = False, 0.0                                      # :792-793
        ev.update(fc=ci["fc"], df=ci["df"], phi=ci.get("phi", 0.0), gd=ci["gd"], dg=ci["dg"], sdgd=ci["sdgd"], pk=ci["pk"], px=ci["px"],
                  pt=ci["pt"], cnt=ci["cnt"])
        out.append(ev)
    return out


# --------------------------------------------------------------------------------------------
# C++ twin: file-level carrier estimate, processing/CPP/main.cpp:363-450 — UNPINNED (fftw3/matio/sigpack absent)
# --------------------------------------------------------------------------------------------

def cpp_file_df(raw, fs: float = 5e6, N: int = 25, remote: int = 0, foffset: float = 0.0):
    """``GoRanging::df``: every N-th ``[I1 Q1 I2 Q2]`` sample of the whole capture, mixed by ``foffset`` (time base
    accumulated ``t += N/fs`` :392), minus the mean of the RAW samples (:384,416-418), squared, FFT of the arbitrary
    length ``nrec`` (:419-421), halves swapped by two memcpy of ``nrec/2`` elements (:423-424), channel 1 arg-max inside
    ``freq < 2*8000`` / ``freq <= -2*8000`` (:401-406,427-430), channel 2 over everything (:443); returns
    (``freq[pos]/2 + foffset``, same for channel 2 or None)."""
    raw = np.asarray(raw).reshape(-1, 4)
    nrec = raw.shape[0] // N
    rec = raw[: nrec * N][::N]
    t = np.zeros(nrec)
    acc = 0.0
    step = float(N) / fs
    for i in range(nrec):
        t[i] = acc
        acc += step
    tlo = complex(0, -1) * float(np.float32(2.0)) * np.pi                    # :28
    lo = np.exp(tlo * foffset * t)
    start, end = -fs / 2 / N, fs / 2 / N
    freq = start + ((end - start) / (nrec - 1)) * np.arange(nrec)
    freq[-1] = end                                                              # linspace :734-757
    kmax = kmin = 0
    for i in range(nrec):
        if freq[i] < 2 * 8000.0:
            kmax = i
        if freq[i] <= -2 * 8000.0:
            kmin = i
    res = [None, None]
    h = nrec // 2
    for ch in ((0,) if remote else (0, 1)):
        dx = rec[:, 2 * ch].astype(np.float64) + 1j * rec[:, 2 * ch + 1]
        x = dx * lo - dx.sum() / nrec
        f = _fft(x * x)
        out = np.zeros(nrec, dtype=complex)
        out[:h] = f[h:2 * h]
        out[h:2 * h] = f[:h]
        pos = int(np.abs(out[kmin:kmax]).argmax()) + kmin if ch == 0 else int(np.abs(out).argmax())
        res[ch] = float(freq[pos] / 2.0 + float(np.float32(foffset)))
    return res[0], res[1]


# --------------------------------------------------------------------------------------------
# Early / prompt / late tracking loop of experiments/230503_100kchips_withcode/gotracking_inv2.m:149-235 — UNPINNED (Octave only)
# --------------------------------------------------------------------------------------------

def octave_xcorr(a, b):
    """``xcorr(a,b)`` of two equal-length vectors (octave-signal): c[k + N] = sum_n a[n+k] conj(b[n]), k = -(N-1) ... N-1 — returned with
    the 2N+1 entries of ``xcorr(a,b,N)`` (lags -N and +N are zero)."""
    a = np.asarray(a, dtype=complex).reshape(-1)
    b = np.asarray(b, dtype=complex).reshape(-1)
    n = a.size
    full = np.correlate(a, b, mode="full") if n <= 4096 else None
    if full is None:                                             # FFT evaluation of the same sum
        m = 2 * n
        c = _ifft(_fft(np.concatenate([a, np.zeros(n)])) * np.conj(_fft(np.concatenate([b, np.zeros(n)]))))
        full = np.concatenate([c[n + 1:], c[:n]])                # lags -(n-1) ... n-1
    return np.concatenate([[0.0], full, [0.0]])


def epl_step(state: dict, x, al, ap, ae, fs=5e6, freq0=0.0, T_blk=80e-3, delay_spacing=0.5, B_DLL=2.0, B_PLL=20.0):
    """One pass of the main loop (:151-249).  ``state``: l (1-based), doppler_freq (list), time_end, code_phase, carrier_phase —
    updated in place; returns the quantities of the block (1-based arg-max indices as Octave's ``max``)."""
    zeta, omega_n = 1 / np.sqrt(2), B_PLL / .53                                            # :146-147
    x = np.asarray(x, dtype=complex).reshape(-1)
    l = state["l"]
    fd = state["doppler_freq"][l - 1]
    time = state["time_end"] + np.arange(1, x.size + 1) / fs                               # :156
    xx = x * np.exp(1j * (2 * np.pi * (-freq0 + fd) * time))                              # :157-159
    zl, zp, ze = (octave_xcorr(a, xx) for a in (al, ap, ae))                               # :161-163, MAXLAG = points_per_code
    bbl, bbp, bbe = (int(np.abs(z).argmax()) + 1 for z in (zl, zp, ze))                    # :175-177
    vl, vp, ve = zl[bbl - 1], zp[bbp - 1], ze[bbe - 1]
    code_phase_error = delay_spacing * (abs(ve) - abs(vl)) / (abs(ve) + abs(vl) + 2 * abs(vp))   # :187
    filtered_code_phase_error = T_blk * B_DLL / .25 * code_phase_error                    # :193
    out = dict(l=l, bbl=bbl, bbp=bbp, bbe=bbe, zl=vl, zp=vp, ze=ve, code_phase_error=code_phase_error)
    out["measured_code_phase"] = state["code_phase"] + code_phase_error
    out["filtered_code_phase"] = state["code_phase"] + filtered_code_phase_error
    delta_theta = np.arctan(vp.imag / vp.real) / (2 * np.pi)                               # :201
    out["delta_theta"] = float(delta_theta)
    out["sortie"] = float(np.arctan2(vp.imag, vp.real) / (2 * np.pi))                      # :202
    out["measured_carrier_phase"] = state["carrier_phase"] + delta_theta
    out["filtered_carrier_phase"] = state["carrier_phase"] + (2 * zeta * omega_n * T_blk - 3 / 2 * omega_n ** 2 * T_blk ** 2) * delta_theta   # :210
    out["measured_doppler_freq"] = fd + T_blk / 2 * delta_theta                            # :204,216
    out["doppler_freq"] = fd + omega_n ** 2 * T_blk * delta_theta                          # :211,217,249
    state["code_phase"], state["carrier_phase"] = out["filtered_code_phase"], out["filtered_carrier_phase"]
    state["time_end"] = float(time[-1])
    state["l"] = l + 1
    state["doppler_freq"].append(out["doppler_freq"])
    return out


def codestep_step(state: dict, x, fs=5e6, freq0=0.0, coef=(1.0, 0.0, 0.05 / 6, 0.0)):
    """One pass of the main loop of experiments/230503_100kchips_withcode/gotracking_test.m:121-187 — the SECOND DLL experiment: the block is
    mixed by ``exp(1j*2*pi*(-freq0+freq)*time)`` (:126-127), correlated with the late / prompt / early replicas over ``MAXLAG`` lags
    (``xcorr(al,xx,MAXLAG)``, :128-130: every lag in the first pass, 20 afterwards, :158), the COHERENT discriminator
    ``d=(|ze|^2-|zl|^2)/(|ze|^2+|zl|^2)`` (:156) steps the code — the prompt replica is rotated by one sample and late / early rebuilt
    from it (:159-168) — and the arctangent of the prompt peak drives the loop filter ``freq=coef_uk_1*freqm1+coef_uk_2*freqm2+
    coef_ek*yp+coef_ek_1*ym1`` (:171-180; ``coef`` = the script's TEST values, :66-72).  ``state``: l, maxlag, ap, al, ae, freq, freqm1,
    freqm2, ym1, time_end — updated in place.  UNPINNED (Octave only)."""
    x = np.asarray(x, dtype=complex).reshape(-1)
    n = x.size
    time = state["time_end"] + np.arange(1, n + 1) / fs                                    # :125
    xx = x * np.exp(1j * (2 * np.pi * (-freq0 + state["freq"]) * time))                   # :126-127
    ml = min(state["maxlag"], n)
    cut = lambda z: z[n - ml: n + ml + 1]                                                  # xcorr(a,xx,MAXLAG): lags -MAXLAG ... MAXLAG
    zl, zp, ze = (cut(octave_xcorr(a, xx)) for a in (state["al"], state["ap"], state["ae"]))
    bbl, bbp, bbe = (int(np.abs(z).argmax()) for z in (zl, zp, ze))                        # 0-based here; Octave's are +1
    vl, vp, ve = zl[bbl], zp[bbp], ze[bbe]
    d = (abs(ve ** 2) - abs(vl) ** 2) / (abs(ve) ** 2 + abs(vl) ** 2)                      # :156
    state["maxlag"] = 20                                                                   # :158
    ap = state["ap"]
    if d < -0.5:                                                                           # :159-163
        ap = np.concatenate([ap[-1:], ap[:-1]])
    if d > 0.5:                                                                            # :164-168 (after the first test, on the possibly rotated ap, as the script does)
        ap = np.concatenate([ap[1:], ap[:1]])
    if d < -0.5 or d > 0.5:
        state["ap"] = ap
        state["al"] = np.concatenate([ap[-1:], ap[:-1]])
        state["ae"] = np.concatenate([ap[1:], ap[:1]])
    yp = float(np.arctan(np.divide(vp.imag, vp.real))) if vp.real != 0 else float(np.sign(vp.imag) * np.pi / 2 if vp.imag != 0 else np.nan)   # :172
    yyp = float(np.arctan2(vp.imag, vp.real))                                              # :173
    freq = coef[0] * state["freqm1"] + coef[1] * state["freqm2"] + coef[2] * yp + coef[3] * state["ym1"]      # :174
    out = dict(l=state["l"], bbl=bbl + 1, bbp=bbp + 1, bbe=bbe + 1, zl=vl, zp=vp, ze=ve, d=float(d), u=float(abs(vp)), yp=yp, yyp=yyp, freq=float(freq),
               stepped=int(d > 0.5) - int(d < -0.5))
    state["freqm2"], state["freqm1"], state["ym1"], state["freq"] = state["freqm1"], freq, yp, freq      # :175-178
    state["time_end"] = float(time[-1])
    state["l"] += 1
    return out
