#!/usr/bin/env python3
"""Round 5, review item 4 (compact Bz with a rigorous two-tier arg-max): how many lags would the rigorous bound leave to re-examine?

numpy emulation of the pipeline's intermediate Bz[rho][k1][q2] (N = N1 x N2 = 625 x 8000, R = 3 phases) for pure-noise and for
signal windows: Bz is rounded to fp16 per (row, phase) block with a power-of-two block exponent, the last pass (length-N1 transforms)
runs on the rounded values, and the candidates are all lags m with |z~_m| + B_m >= max_n (|z~_n| - B_n), B = 2^-11 sum_k1 |Bz~| of the
lag's column (the worst-case error of the column sum).  Prints per window: the lag of the exact map, whether the fp16 arg-max equals it,
the size of the candidate set.  CPU only (tools/, not product code)."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amaranth_twstft_amd import prn, synth   # noqa: E402

N1, N2, R = 625, 8000, 3
N = N1 * N2
FS = 5e6


def bz_of(y, fcode):
    """Bz[rho][k1][q2] with prnmap[R*(q1*N2+q2)+rho] = sum_k1 Bz[rho][k1][q2] * exp(2 pi i k1 q1 / N1) (polyphase form)."""
    P = np.fft.fft(y) * fcode                                # k natural
    k = np.arange(N)
    ks = np.where(k < N // 2, k, k - N)                      # signed bin
    out = []
    for rho in range(R):
        Pr = P * np.exp(2j * np.pi * ks * rho / (R * N))     # phase ramp of the fractional lag
        A = Pr.reshape(N2, N1).T                             # [k1][k2], k = k1 + N1 k2
        rows = np.fft.ifft(A, axis=1) * N2                   # sum over k2: e^{+2 pi i k2 q2 / N2}
        tw = np.exp(2j * np.pi * np.outer(np.arange(N1), np.arange(N2)) / N)      # W_N^{-k1 q2} (inverse sign)
        out.append(rows * tw)
    return out


def last_pass(bz):
    z = np.empty((R, N1, N2), dtype=complex)
    for rho in range(R):
        z[rho] = np.fft.ifft(bz[rho], axis=0) * N1           # sum over k1: e^{+2 pi i k1 q1 / N1} -> [q1][q2]
    # lag m = R*(q1*N2 + q2) + rho
    return np.transpose(z, (1, 2, 0)).reshape(-1) / (R * N)


def fp16_blocks(b):
    """per (row k1) block exponent, fp16 mantissas: what k_rowd<MID> would store"""
    mx = np.abs(b).max(axis=1, keepdims=True)
    e = np.ceil(np.log2(np.maximum(mx, 1e-300) / 32768.0))
    s = 2.0 ** e
    q = (b.real / s).astype(np.float16).astype(np.float64) + 1j * (b.imag / s).astype(np.float16).astype(np.float64)
    return q * s


def main():
    chips = prn.lfsr_chips(22, 3, N // 2)
    code = np.repeat(2.0 * chips - 1.0, 2)
    fcode = np.conj(np.fft.fft(code))
    print("window kind      exact_lag   fp16_argmax_equal   candidates   bound/max")
    for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        noise_only = w % 2 == 0
        p = synth.SynthParams(delay_q8=(1311765 - w) * 256, fstep=synth.fstep_for_df(0.0, FS), phi0=w, amp=0 if noise_only else 200,
                              noise_gain=synth.noise_gain_for_sigma(400.0), seed=500 + w, stream=0)
        raw = np.asarray(synth.synth_channel(N, chips, 2, p)).astype(np.float64)
        raw = raw.reshape(-1, 2)
        y = raw[:, 0] + 1j * raw[:, 1]
        y -= y.mean()
        bz = bz_of(y, fcode)
        z = last_pass(bz)
        q = [fp16_blocks(b) for b in bz]
        zq = last_pass(q)
        # rigorous per-column bound: |sum_k1 err| <= sum_k1 |err|, |err| <= 2^-11 |Bz~| (+ subnormal floor, negligible with block exponents)
        B = np.stack([np.abs(b).sum(axis=0) for b in q]) * 2.0 ** -11 * np.sqrt(2) / (R * N)            # [rho][q2]; sqrt 2: both components
        Bm = np.broadcast_to(B.T[None, :, :], (N1, N2, R)).reshape(-1)
        a, aq = np.abs(z), np.abs(zq)
        thr = (aq - Bm).max()
        cand = int(np.count_nonzero(aq + Bm >= thr))
        assert np.abs(a - aq).max() <= Bm.max() + 1e-12 * a.max()
        print(f"{w:3d}   {'noise ' if noise_only else 'signal'}   {int(a.argmax()):9d}   {str(bool(a.argmax() == aq.argmax())):17s}   {cand:10d}   {Bm.max() / a.max():.2e}")


if __name__ == "__main__":
    main()
