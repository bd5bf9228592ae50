"""GPU (-m gpu): the DLL/PLL receiver as a program (twx_rx_*, experiments/231001_DLL_PLL/rxcomplex.cpp:263-835): parameter rows in,
acquisition in the first second, code lock in the second, .dat rows after — every second against the oracle's restatement of the
loop body (oracle.rx_second: UNPINNED, the C++ program needs fftw3 / gsl / cblas).  sdr.param sizes: 5 Msps two-channel capture,
x2 interpolation to 10 Msps, 100 000-chip codes at 2.5 Mchip/s (nobs = 400 000, nfft = 2^20), +-28 lags x 24 code periods."""
import ctypes as C
import os

import numpy as np
import pytest

from amaranth_twstft_amd import _lib as L
from amaranth_twstft_amd import receiver, synth
from oracle import twstft_oracle as orc
from tests.helpers import chips_for
from tests.test_gpu_configs import _synth_dev

pytestmark = pytest.mark.gpu
N_IN, SPS, CLEN = 5_000_000, 10_000_000, 100_000


def _capture(seconds, jump_after=None, jump=0):
    """Two physical channels, continuous over `seconds` seconds: A carries code 0 (LFSR 17 taps 9 = 0.bin), B code 1 (taps 15).
    From second `jump_after` on, channel A arrives `jump` samples (5 Msps) later."""
    import torch
    dev = torch.device("cuda", 0)
    chips = {"A": chips_for(17, 9, CLEN), "B": chips_for(17, 15, CLEN)}
    chans = {"A": synth.SynthParams(delay_q8=123_457 * 256, fstep=synth.fstep_for_df(1307.25, 5e6), phi0=11, amp=400,
                                    noise_gain=synth.noise_gain_for_sigma(900.0), seed=31, stream=0),
             "B": synth.SynthParams(delay_q8=77_001 * 256, fstep=synth.fstep_for_df(-260.5, 5e6), phi0=5, amp=500,
                                    noise_gain=synth.noise_gain_for_sigma(700.0), seed=31, stream=1)}
    out = np.empty((seconds, N_IN, 4), dtype=np.int16)
    tmp = torch.empty((N_IN, 2), dtype=torch.int16, device=dev)
    for s in range(seconds):
        for c, key in enumerate(("A", "B")):
            p = chans[key]
            if key == "A" and jump_after is not None and s >= jump_after:
                p = synth.SynthParams(**{**p.__dict__, "delay_q8": p.delay_q8 + jump * 256})
            _synth_dev(tmp, N_IN, torch.from_numpy(chips[key]).to(dev), CLEN, 2, [p], n0=s * N_IN)
            torch.cuda.synchronize()
            out[s, :, 2 * c:2 * c + 2] = tmp.cpu().numpy()
    return chips, out


def test_receiver_program_against_the_oracle_loop(tmp_path):
    seconds = 5
    chips, cap = _capture(seconds)
    chips["A"].tofile(tmp_path / "0.bin")                       # SDRcode reads <pn-100>.bin (:875)
    chips["B"].tofile(tmp_path / "1.bin")
    param = tmp_path / "sdr.param"
    param.write_text("# ch mode pn fc kcps lpf range step snr\n"
                     "A N 100 0001186 2500 1250 2000 256 -18\n"        # code 0 on channel A: present
                     "B N 101 -000186 2500 1250 2000 256 -18\n"        # code 1 on channel B: present
                     "A N 101 0000186 2500 1250 1000 256 -18\n")       # code 1 on channel A: absent, re-acquires every second
    rows = receiver.parse_param(str(param))
    assert len(rows) == 3
    outdir = tmp_path / "out"
    outdir.mkdir()
    block = 3
    with receiver.Receiver(rows, code_dir=str(tmp_path), out_dir=str(outdir), acq_block=block) as rx:
        infos = [rx.channel(i) for i in range(3)]
        got = [rx.second(cap[s]) for s in range(seconds)]
    # ---- oracle: the same rows, the same offsets
    orows = orc.rx_parse_param(param.read_text().splitlines(True))
    cis = [orc.rx_channel_setup(r, chips["A"] if r["pn"] == 100 else chips["B"], SPS) for r in orows]
    for info, ci in zip(infos, cis):
        assert (info.nobs, info.nfft, info.bps, info.nlag, info.clen) == (ci["nobs"], ci["nfft"], ci["bps"], ci["nlag"], ci["clen"])
        assert (info.range, info.step) == (ci["range"], ci["step"]) and abs(info.snr_min - ci["snr_min"]) < 1e-15
        assert abs(info.psbb - ci["psbb"]) <= 1e-9 * ci["psbb"] and info.dat_name.decode() == ci["dat_name"]
    want = [orc.rx_second(cis, cap[s].reshape(-1), SPS, 1, lambda i, ci: block * ci["nobs"]) for s in range(seconds)]
    names = {L.TWX_RX_NO_SIGNAL: "no signal", L.TWX_RX_ACQUIRED: "acquired", L.TWX_RX_CODE_LOCK: "code lock", L.TWX_RX_TRACKED: "tracked",
             L.TWX_RX_ACQ_FAILED: "acq failed", L.TWX_RX_LOCK_LOST: "lock lost"}
    status = [[names[r.status] for r in sec] for sec in got]
    assert status == [[e["status"] for e in sec] for sec in want]
    assert status[0] == ["acquired", "acquired", "no signal"] and status[1] == ["code lock", "code lock", "no signal"]
    assert all(st == ["tracked", "tracked", "no signal"] for st in status[2:])
    for s in range(seconds):
        for i in range(3):
            g, w = got[s][i], want[s][i]
            assert abs(g.px - w["px"]) <= 1e-5 * w["px"]                                  # received power :481-489
            if w["status"] in ("acquired", "no signal"):
                assert (g.fc, g.pt, g.acq_idx) == (w["fc"], w["pt"], w["acq_idx"])      # carrier of the sweep and cblas_izamax lag: exact
                assert abs(g.pk - w["pk"]) <= 1e-5 * w["pk"]
                if w["status"] == "acquired":
                    assert g.gd == w["gd"]
            else:
                assert (g.fc, g.pt, g.cnt) == (w["fc"], w["pt"], w["cnt"])                # integer carrier, code phase, usable periods
                assert abs(g.fc + g.df - (w["fc"] + w["df"])) <= 5e-4 and abs(g.phi - w["phi"]) <= 2e-4
                assert abs(g.gd - w["gd"]) <= 0.05 and abs(g.dg - w["dg"]) <= 0.1 and abs(g.sdgd - w["sdgd"]) <= 0.1
                assert abs(g.pk - w["pk"]) <= 1e-5 * w["pk"]
            if w["status"] == "tracked":
                gr, wr = g.dat_row.decode(), w["dat_row"]
                assert gr.endswith("\n") and len(gr.split()) == len(wr.split()) == 9
                gv, wv = [float(x) for x in gr.split()], [float(x) for x in wr.split()]
                assert gv[2] == wv[2] and gv[3] == wv[3] == 0.0                           # cnt, ib*duration
                assert abs(gv[0] - wv[0]) <= 1e-3 and abs(gv[4] - wv[4]) <= 0.05 and abs(gv[7] - wv[7]) <= 2e-3 and abs(gv[8] - wv[8]) <= 2e-3
            else:
                assert g.dat_row == b""
    # the truth of the generator: carrier offsets and code phases (delay in 5-Msps samples -> x2)
    assert abs(got[-1][0].fc + got[-1][0].df - 1307.25) < 0.2 and abs(got[-1][1].fc + got[-1][1].df + 260.5) < 0.2
    assert abs(got[-1][0].gd - 2 * 123_457 * 100.0) < 100.0 and abs(got[-1][1].gd - 2 * 77_001 * 100.0) < 100.0      # ns at 10 Msps
    # ---- files: rows appended to ch?.pn???.2500kcps.dat, the log lines of the program
    for i in (0, 1):
        text = (outdir / infos[i].dat_name.decode()).read_text()
        assert text == "".join(got[s][i].dat_row.decode() for s in range(2, seconds))
    assert not (outdir / infos[2].dat_name.decode()).exists()
    log = (outdir / "rxcomplex.log").read_text().splitlines(True)
    assert log[:3] == [ci["log_set"] for ci in cis]
    assert [l[:11] for l in log[3:]] == ["acquisition", "acquisition", "code lock  ", "code lock  "]
    wl = [e["log"] for sec in want for e in sec if e["log"]]
    assert [l.split(",")[:2] for l in log[3:]] == [l.split(",")[:2] for l in wl] and log[5:] == wl[2:]      # code-lock lines equal to the character
    acq_g, acq_w = log[3].split(), wl[0].split()
    assert acq_g[:8] == acq_w[:8]                                                        # Ch, PRN, block, carrier, gd, pt


def test_receiver_file_loop_and_lock_lost(tmp_path):
    """twx_rx_file = `./rxcomplex data.bin sdr.param`: whole seconds of a capture FILE.  After three seconds the signal arrives 28
    samples of the 10-Msps stream later: every code period then peaks on the edge of the +-28-lag window, no period is usable
    (:634) -> 'lock lost' (:783-793), and the channel acquires again in the next second.  (Noise alone does not unlock the
    program: its tracking branch has no SNR test, the one at :634 is commented out.)"""
    seconds = 5
    chips, cap = _capture(seconds, jump_after=3, jump=14)
    path = tmp_path / "data.bin"
    with open(path, "wb") as f:
        cap.tofile(f)
        np.zeros(1000, dtype=np.int16).tofile(f)                  # a short tail: not a whole second, not processed
    row = receiver.make_row("A", 100, 1186.0, 2000.0, 256.0, -18.0, code=chips["A"])
    with receiver.Receiver([row], out_dir=str(tmp_path), seed=7) as rx:
        reps = rx.run_file(str(path))
        assert len(reps) == seconds
        st = [r[0].status for r in reps]
        assert st == [L.TWX_RX_ACQUIRED, L.TWX_RX_CODE_LOCK, L.TWX_RX_TRACKED, L.TWX_RX_LOCK_LOST, L.TWX_RX_ACQUIRED]
        assert 0 <= reps[0][0].acq_idx < SPS and reps[0][0].acq_idx % 400_000 == 0 and reps[0][0].n_trials == 17 + 3 * 8
        assert reps[3][0].cnt == 0 and reps[4][0].pt == (reps[0][0].pt + 28) % 400_000
    log = (tmp_path / "rxcomplex.log").read_text()
    assert "lock lost   : Ch. A, PRN#100, count =" in log and len((tmp_path / "chA.pn100.2500kcps.dat").read_text().splitlines()) == 1


def test_receiver_refuses_what_the_program_cannot_do(tmp_path):
    chips = chips_for(17, 9, CLEN)
    with pytest.raises(L.TwxError, match="SIC rows need ninterp = 1"):
        receiver.Receiver([receiver.make_row("A", 100, 186.0, 2000.0, 256.0, -18.0, mode="S", code=chips)])
    with pytest.raises(L.TwxError, match="Code filename error"):
        receiver.Receiver([receiver.make_row("A", 100, 186.0, 2000.0, 256.0, -18.0)], code_dir=str(tmp_path))
    with pytest.raises(L.TwxError, match="no code source"):
        receiver.Receiver([receiver.make_row("B", 7, 186.0, 2000.0, 256.0, -18.0)])
    with pytest.raises(L.TwxError, match="ranges of rxcomplex.cpp:288"):
        receiver.Receiver([receiver.make_row("A", 100, 186.0, 100.0, 256.0, -18.0, code=chips)])


def test_receiver_b210_build_and_device_resident_seconds():
    """dec_a = 2 (the N210 / B210 build, rxcomplex.cpp:226-231: every second sample of the stream in the acquisition, nfft = 2^19, pt * dec_a
    at the hand-over :575) and seconds that already sit in device memory (twx_rx_second_dev), against the oracle loop with the same dec_a."""
    import torch
    seconds = 3
    chips, cap = _capture(seconds)
    row = receiver.make_row("B", 101, -186.0, 2000.0, 256.0, -18.0, code=chips["B"])
    with receiver.Receiver([row], dec_a=2, acq_block=5) as rx:
        info = rx.channel(0)
        assert info.nfft == 1 << 19 and info.nobs == 400_000
        dev = torch.device("cuda", 0)
        got = []
        for s in range(seconds):
            d = torch.from_numpy(cap[s]).to(dev)
            got.append(rx.second_dev(d.data_ptr())[0])
            assert rx.stream_dev(1) != 0 and rx.stream_dev(0) == 0          # only physical channel B is interpolated
    orow = dict(ch="B", mode="N", pn=101, fc_init=-186.0, kcps=2500, fltkhz=1250.0, frange=2000.0, fstep=256.0, snr_min_db=-18.0)
    cis = [orc.rx_channel_setup(orow, chips["B"], SPS, dec_a=2)]
    assert abs(info.psbb - cis[0]["psbb"]) <= 1e-9 * cis[0]["psbb"]
    want = [orc.rx_second(cis, cap[s].reshape(-1), SPS, 2, lambda i, ci: 5 * ci["nobs"])[0] for s in range(seconds)]
    assert [g.status for g in got] == [L.TWX_RX_ACQUIRED, L.TWX_RX_CODE_LOCK, L.TWX_RX_TRACKED]
    assert [w["status"] for w in want] == ["acquired", "code lock", "tracked"]
    assert (got[0].fc, got[0].pt, got[0].gd) == (want[0]["fc"], want[0]["pt"], want[0]["gd"]) and got[0].pt % 2 == 0
    for g, w in zip(got[1:], want[1:]):
        assert (g.fc, g.pt, g.cnt) == (w["fc"], w["pt"], w["cnt"]) and abs(g.gd - w["gd"]) <= 0.05 and abs(g.fc + g.df - (w["fc"] + w["df"])) <= 5e-4
    assert abs(got[-1].gd - 2 * 77_001 * 100.0) < 100.0


def test_real_sample_program_with_interference_cancellation(tmp_path):
    """cfg.ninterp = 1: the real-sample program experiments/231001_DLL_PLL/rx.cpp — the I samples only, no interpolation (nobs = 200 000,
    nfft = 2^19), rxreal.log — and its successive interference cancellation (rx.cpp:505-518, MAI_up :1011-1020, MAI_out :1022-1027).
    Physical channel A carries code 0 strongly and code 1 weakly.  The 'N' row for code 1 never passes the -14 dB gate (the strong code
    is its noise); the 'S' row for the same code does, from the second in which code 0 is past its code lock: the received power of
    the cleaned stream drops, the weak code is acquired, locked and tracked.  Every second against oracle.rx_second(real=True)."""
    import torch
    seconds = 4
    dev = torch.device("cuda", 0)
    chips = {0: chips_for(17, 9, CLEN), 1: chips_for(17, 15, CLEN)}
    par = {0: synth.SynthParams(delay_q8=123_457 * 256, fstep=synth.fstep_for_df(1307.25, 5e6), phi0=11, amp=1500,
                                noise_gain=synth.noise_gain_for_sigma(600.0), seed=31, stream=0),
           1: synth.SynthParams(delay_q8=77_001 * 256, fstep=synth.fstep_for_df(-260.5, 5e6), phi0=5, amp=300, noise_gain=0, seed=31, stream=1)}
    cap = np.zeros((seconds, N_IN, 4), dtype=np.int16)
    tmp = torch.empty((N_IN, 2), dtype=torch.int16, device=dev)
    for s in range(seconds):
        acc = np.zeros((N_IN, 2), dtype=np.int32)
        for c in (0, 1):
            _synth_dev(tmp, N_IN, torch.from_numpy(chips[c]).to(dev), CLEN, 2, [par[c]], n0=s * N_IN)
            torch.cuda.synchronize()
            acc += tmp.cpu().numpy()
        cap[s, :, 0:2] = acc                                    # physical channel B stays silent
    chips[0].tofile(tmp_path / "0.bin")
    chips[1].tofile(tmp_path / "1.bin")
    param = tmp_path / "sdr.param"
    param.write_text("A N 100 0001186 2500 1250 2000 256 -18\n"         # the strong code
                     "A S 101 -000186 2500 1250 2000 256 -14\n"         # the weak code behind the cancellation: printed as PRN 151
                     "A N 101 -000186 2500 1250 2000 256 -14\n")        # the weak code without it
    rows = receiver.parse_param(str(param))
    assert [r.mode for r in rows] == [b"N", b"S", b"N"]
    outdir = tmp_path / "out"
    outdir.mkdir()
    block = 3
    with receiver.Receiver(rows, code_dir=str(tmp_path), out_dir=str(outdir), acq_block=block, real=True) as rx:
        infos = [rx.channel(i) for i in range(3)]
        got = []
        for s in range(seconds):
            got.append(rx.second(cap[s]))
            assert rx.stream_dev(0) != 0 and rx.stream_dev(1) == 0 and rx.stream_dev(2) != 0
    assert [(i.nobs, i.nfft, i.bps, i.nlag, i.is_sic) for i in infos] == [(200_000, 1 << 19, 25, 28, 0), (200_000, 1 << 19, 25, 28, 1), (200_000, 1 << 19, 25, 28, 0)]
    assert [i.dat_name.decode() for i in infos] == ["chA.pn100.2500kcps.dat", "chA.pn151.2500kcps.dat", "chA.pn101.2500kcps.dat"]
    # ---- oracle
    orows = orc.rx_parse_param(param.read_text().splitlines(True))
    cis = [orc.rx_channel_setup(r, chips[0] if r["pn"] == 100 else chips[1], N_IN) for r in orows]
    for info, ci in zip(infos, cis):
        assert abs(info.psbb - ci["psbb"]) <= 1e-9 * ci["psbb"] and info.dat_name.decode() == ci["dat_name"]
    want = [orc.rx_second(cis, cap[s].reshape(-1), N_IN, 1, lambda i, ci: block * ci["nobs"], real=True) for s in range(seconds)]
    status = [[receiver.STATUS[r.status] for r in sec] for sec in got]
    assert status == [[e["status"] for e in sec] for sec in want]
    assert status == [["acquired", "no signal", "no signal"], ["code lock", "acquired", "no signal"], ["tracked", "code lock", "no signal"],
                      ["tracked", "tracked", "no signal"]]
    for s in range(seconds):
        for i in range(3):
            g, w = got[s][i], want[s][i]
            assert abs(g.px - w["px"]) <= (1e-4 if i == 1 else 1e-5) * w["px"], (s, i)      # row 1: power of the CLEANED stream (rx.cpp:515-516)
            if w["status"] in ("acquired", "no signal"):
                assert (g.fc, g.pt, g.acq_idx) == (w["fc"], w["pt"], w["acq_idx"]), (s, i)
                assert abs(g.pk - w["pk"]) <= 1e-4 * w["pk"]
            else:
                assert (g.fc, g.pt, g.cnt) == (w["fc"], w["pt"], w["cnt"]), (s, i)
                assert abs(g.fc + g.df - (w["fc"] + w["df"])) <= 2e-3 and abs(g.gd - w["gd"]) <= 0.2 and abs(g.pk - w["pk"]) <= 1e-4 * w["pk"]
    # what the cancellation buys: the cleaned stream's power is less than half of channel A's from the second after code 0's code lock
    assert got[0][1].px == got[0][0].px and all(got[s][1].px < 0.55 * got[s][0].px for s in (1, 2, 3))
    assert all(got[s][2].px == got[s][0].px for s in range(seconds))
    # the truth of the generator (real samples: the carrier's sign is not observable, the sweep settles on +260.5)
    assert abs(got[-1][0].fc + got[-1][0].df - 1307.25) < 0.2 and abs(abs(got[-1][1].fc + got[-1][1].df) - 260.5) < 0.2
    assert abs(got[-1][0].gd - 123_457 * 200.0) < 100.0 and abs(got[-1][1].gd - 77_001 * 200.0) < 100.0            # ns at 5 Msps
    # ---- files: rxreal.log, PRN + 50 for the SIC row
    log = (outdir / "rxreal.log").read_text().splitlines(True)
    assert not (outdir / "rxcomplex.log").exists()
    assert log[:3] == [ci["log_set"] for ci in cis] and "PRN#151" in log[1]
    wl = [e["log"] for sec in want for e in sec if e["log"]]
    assert [l[:11] for l in log[3:]] == ["acquisition", "code lock  ", "acquisition", "code lock  "] == [l[:11] for l in wl]
    assert [l.split(",")[:2] for l in log[3:]] == [l.split(",")[:2] for l in wl]
    assert (outdir / "chA.pn151.2500kcps.dat").read_text() == got[3][1].dat_row.decode()
    assert len((outdir / "chA.pn100.2500kcps.dat").read_text().splitlines()) == 2 and not (outdir / "chA.pn101.2500kcps.dat").exists()


def test_command_line_programs_equal_the_library_calls(tmp_path):
    """apps/bin/rxcomplex_hip and rx_hip (`./rxcomplex data.bin sdr.param`, `./rx …`): child processes on a capture file; the .dat
    rows and log files they append are byte-identical to what the same seed gives through the library calls, and stdout carries the
    programs' lines (rxcomplex.cpp:804-831): the PWR line and one line per parameter row and second."""
    import re
    import subprocess
    seconds = 3
    chips, cap = _capture(seconds)
    chips["A"].tofile(tmp_path / "0.bin")
    chips["B"].tofile(tmp_path / "1.bin")
    cap.tofile(tmp_path / "data.bin")
    (tmp_path / "sdr.param").write_text("A N 100 0001186 2500 1250 2000 256 -18\nB N 101 -000186 2500 1250 2000 256 -18\nA N 101 0000186 2500 1250 1000 256 -18\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for prog, real, logname in (("rxcomplex_hip", False, "rxcomplex.log"), ("rx_hip", True, "rxreal.log")):
        out_cli, out_lib = tmp_path / (prog + "_cli"), tmp_path / (prog + "_lib")
        out_cli.mkdir(); out_lib.mkdir()
        env = dict(os.environ, TWX_RX_SEED="11", TWX_RX_CODES=str(tmp_path), TWX_RX_OUT=str(out_cli))
        r = subprocess.run([os.path.join(root, "apps", "bin", prog)], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rows = receiver.parse_param(str(tmp_path / "sdr.param"))
        with receiver.Receiver(rows, code_dir=str(tmp_path), out_dir=str(out_lib), seed=11, real=real) as rx:
            reps = rx.run_file(str(tmp_path / "data.bin"))
            last_lines = [rx.console_line(i, reps[-1][i]) for i in range(3)]
        assert len(reps) == seconds and [x.status for x in reps[-1]][:2] == [L.TWX_RX_TRACKED, L.TWX_RX_TRACKED]
        names = sorted(p.name for p in out_lib.iterdir())
        assert names == sorted(p.name for p in out_cli.iterdir()) and logname in names and len(names) == 3
        for n in names:
            assert (out_cli / n).read_bytes() == (out_lib / n).read_bytes(), n
        lines = r.stdout.splitlines()
        assert lines[0] == "./data.bin" and sum(l.startswith("PWR A:") for l in lines) == seconds
        body = [l for l in lines[1:] if l and not l.startswith("PWR A:")]
        assert len(body) == 3 * seconds and body[-3:] == [l.rstrip("\n") for l in last_lines]
        assert re.fullmatch(r"A: \d\.\d+ \d\.\d+ A: #100  2\.5 Mcps SNR +-?\d+\.\d\d > -18\.00, analyzing", body[0])
        assert re.fullmatch(r"A: #100  2\.5 Mcps +1307\.\d{3} Hz +\d+\.\d{3} \( ?\d+\.\d{3}\) ns SNR +-?\d+\.\d\d dB", body[-3])
        assert body[-1].endswith(", no signal") and "#101" in body[-1]
        pw = [l for l in lines if l.startswith("PWR A:")][-1]
        assert re.fullmatch(r"PWR A: +-?\d+\.\d\d dBm , PWR B: +-?\d+\.\d\d dBm", pw)


def test_randomised_receiver_scenarios():
    """Random scenarios through both receiver programs at half the sample rate (fs_in = 2.5 Msps: one sample per chip in, nobs =
    200 000 / 100 000, the same code paths at a quarter of the oracle's cost): which codes are on which physical channel and how
    strong (well above the gate, or absent), carriers inside the search range, delays, a delay jump in some second (lock lost ->
    re-acquisition), the X310 / B210 builds (dec_a), rxcomplex / rx with 'S' rows behind 'N' rows of the other code — the state of
    every row after every second (status, integer carrier, code phase, usable periods) equal to oracle.rx_second, the measured
    quantities within the fp32 tolerances.  TWX_SWEEP_OPTIONS raises the count."""
    rng = np.random.default_rng(161803 + int(os.environ.get("TWX_SWEEP_SEED", "0")))
    ncomb = int(os.environ.get("TWX_SWEEP_OPTIONS", "3"))
    fs_in, n_in = 2.5e6, 2_500_000
    codes = {100: chips_for(17, 9, CLEN), 101: chips_for(17, 15, CLEN)}
    names = {v: k for k, v in receiver.STATUS.items()}
    for it in range(ncomb):
        real = bool(rng.integers(0, 2))
        dec_a = int(rng.integers(1, 3))
        sps = n_in * (1 if real else 2)
        seconds = int(rng.integers(3, 6))
        # what is on the air: per physical channel a list of (pn, amp, carrier, delay)
        air = {"A": [], "B": []}
        for chn in ("A", "B"):
            for pn in (100, 101):
                if rng.integers(0, 3) == 0:
                    continue
                # real samples cannot tell +f from -f (the two sweep peaks are equal up to rounding): the real program's rows search
                # a range that holds only one of them
                car = float(rng.uniform(300, 450)) if real else float(rng.uniform(-400, 400))
                air[chn].append([pn, int(rng.choice([900, 1500])), car, int(rng.integers(0, CLEN))])
        jump_sec = int(rng.integers(2, seconds + 2))                     # beyond the capture: no jump
        cap = np.zeros((seconds, n_in, 4), dtype=np.int16)
        for s in range(seconds):
            for c, chn in enumerate(("A", "B")):
                acc = np.zeros((n_in, 2), dtype=np.int32)
                for j, (pn, amp, car, dly) in enumerate(air[chn]):
                    d = dly + (37 if s >= jump_sec else 0)
                    p = synth.SynthParams(delay_q8=d * 256, fstep=synth.fstep_for_df(car, fs_in), phi0=3 + j, amp=amp,
                                          noise_gain=synth.noise_gain_for_sigma(500.0) if j == 0 else 0, seed=1000 * it + 10 * c + j, stream=c)
                    acc += synth.synth_channel(n_in, codes[pn], 1, p, n0=s * n_in).reshape(n_in, 2)
                if not air[chn]:
                    p = synth.SynthParams(delay_q8=0, fstep=0, phi0=0, amp=0, noise_gain=synth.noise_gain_for_sigma(500.0), seed=1000 * it + 10 * c, stream=c)
                    acc += synth.synth_channel(n_in, codes[100], 1, p, n0=s * n_in).reshape(n_in, 2)
                cap[s, :, 2 * c:2 * c + 2] = np.clip(acc, -32768, 32767)
        # rows: every (channel, code) pair in random order, 'S' for some rows of the real program that follow an 'N' row of the other code
        pairs = [(chn, pn) for chn in ("A", "B") for pn in (100, 101)]
        rng.shuffle(pairs)
        pairs = pairs[: int(rng.integers(1, 5))]
        rows, orows = [], []
        for chn, pn in pairs:
            mode = "S" if real and rng.integers(0, 2) and any(o["ch"] == chn and o["pn"] != pn and o["mode"] == "N" for o in orows) else "N"
            fc_init = float(rng.integers(340, 411)) if real else float(rng.integers(-100, 101))
            frange, fstep = (200.0, 64.0) if real else (512.0, 128.0)
            rows.append(receiver.make_row(chn, pn, fc_init, frange, fstep, -14.0, mode=mode, code=codes[pn]))
            orows.append(dict(ch=chn, mode=mode, pn=pn, fc_init=fc_init, kcps=2500, fltkhz=1250.0, frange=frange, fstep=fstep, snr_min_db=-14.0))
        block = int(rng.integers(0, 5))
        tag = f"combination {it}: {'rx' if real else 'rxcomplex'} dec_a={dec_a} {seconds} s, air {air}, jump at {jump_sec}, rows {[(o['ch'], o['mode'], o['pn']) for o in orows]}"
        with receiver.Receiver(rows, fs_in=fs_in, acq_block=block, dec_a=dec_a, real=real) as rx:
            got = [rx.second(cap[s]) for s in range(seconds)]
        cis = [orc.rx_channel_setup(o, codes[o["pn"]], sps, dec_a) for o in orows]
        try:
            for s in range(seconds):
                want = orc.rx_second(cis, cap[s].reshape(-1), sps, dec_a, lambda i, ci: block * ci["nobs"], real=real)
                for i, (g, w) in enumerate(zip(got[s], want)):
                    assert g.status == names[w["status"]], (s, i, receiver.STATUS[g.status], w["status"])
                    assert (g.fc, g.pt) == (w["fc"], w["pt"]), (s, i, g.fc, w["fc"], g.pt, w["pt"])
                    assert abs(g.px - w["px"]) <= 2e-4 * w["px"] and abs(g.pk - w["pk"]) <= 2e-4 * abs(w["pk"]) + 1e-12, (s, i)
                    if w["status"] in ("tracked", "code lock"):
                        assert g.cnt == w["cnt"] and abs(g.fc + g.df - (w["fc"] + w["df"])) <= 5e-3 and abs(g.gd - w["gd"]) <= 0.5, (s, i)
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from e
