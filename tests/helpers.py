"""Shared helpers for the parity tests (synthetic inputs from fixture descriptions)."""
import hashlib
import json
import os

import numpy as np

from amaranth_twstft_amd import prn, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


_chip_cache = {}


def chips_for(bitlen, taps, nchips):
    key = (bitlen, taps, nchips)
    if key not in _chip_cache:
        _chip_cache[key] = prn.lfsr_chips(bitlen, taps, nchips)
    return _chip_cache[key]


def capture_from_desc(desc, check_sha=None):
    """Rebuild the int16 capture a fixture describes (see tools/make_golden.py:synth_desc)."""
    chips = chips_for(desc["bitlen"], desc["taps"], desc["nchips"])
    chans = [synth.SynthParams(**c) for c in desc["channels"]]
    raw = synth.synth_capture(desc["n"], chips, desc["sps"], chans)
    if check_sha is not None:
        assert hashlib.sha256(raw.tobytes()).hexdigest() == check_sha, "synthetic generator drifted"
    return chips, raw


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)
