"""CPU: the integer synthetic-capture generator is deterministic and statistically sane."""
import hashlib

import numpy as np

from amaranth_twstft_amd import prn, synth


def test_generator_is_bit_stable():
    chips = prn.lfsr_chips(13, 27, 5000)
    p = synth.SynthParams(delay_q8=1234 * 256 + 17, fstep=synth.fstep_for_df(843.75, 5e6), phi0=99, amp=300,
                          noise_gain=synth.noise_gain_for_sigma(600.0), seed=1, stream=3)
    raw = synth.synth_channel(10000, chips, 2, p)
    assert raw.dtype == np.int16 and raw.shape == (10000, 2)
    # frozen digest: any change to the generator invalidates tests/golden (regenerate deliberately)
    assert hashlib.sha256(raw.tobytes()).hexdigest() == "0030cde0239315098836b725d387d2614bd28626ddaa42baa0da64a83b3b212a"
    assert abs(raw.astype(float).std() - np.sqrt(600.0 ** 2 + 300.0 ** 2 / 2)) < 15
    # window offset n0 continues the same stream
    a = synth.synth_channel(4000, chips, 2, p, n0=6000)
    assert np.array_equal(a, raw[6000:])


def test_carrier_and_delay_are_where_requested():
    chips = prn.lfsr_chips(13, 27, 5000)
    n = 10000
    fs = 5e6
    fstep = synth.fstep_for_df(2000.0, fs)
    p = synth.SynthParams(delay_q8=777 * 256, fstep=fstep, phi0=0, amp=1000, noise_gain=0)
    raw = synth.synth_channel(n, chips, 2, p).astype(float)
    d = raw[:, 0] + 1j * raw[:, 1]
    code = np.repeat(chips.astype(float), 2) * 2 - 1
    y = d * np.exp(-2j * np.pi * synth.df_of_fstep(fstep, fs) * np.arange(n) / fs)
    z = np.fft.ifft(np.fft.fft(y) * np.conj(np.fft.fft(code)))
    assert int(np.abs(z).argmax()) == 777
    assert abs(abs(z[777]) / n - 1000) < 2
    assert abs(synth.df_of_fstep(fstep, fs) - 2000.0) < fs / 2 ** 32
