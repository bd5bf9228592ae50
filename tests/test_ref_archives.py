"""CPU: the reference's OWN result archives as test vectors (tests/golden/ref_archives.{json,npz}, written by
tools/make_golden_archives.py from experiments/220616_Besancon, 230315_analysis_100k, 240102_1PPS_TXsync/2401_{OP,LTFB}, 240527 and
230111_twstft_2M5 of the reference repository).  The captures behind them are gone, but the stored numbers are real outputs of the
reference's scripts: they pin the parabola of godual_ranging.m:33, the linspace frequency grid of :15,73, the replica of the 100-kchip
codes, the SNR definition, the re-alignment rule of the tracked flow (Octave only — no runnable twin) and go_1s.m's session arithmetic."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from amaranth_twstft_amd import prn, results_io, twoway
from amaranth_twstft_amd.correlator import WindowResult
from oracle import twstft_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS = 5e6


@pytest.fixture(scope="module")
def arch():
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_archives.json")))
    bulk = np.load(os.path.join(ROOT, "tests", "golden", "ref_archives.npz"))
    return doc, bulk


def test_besancon_parabola_and_polyfit_against_stored_corrections(arch):
    """experiments/220616_Besancon/godual.m:45-48 stored `correction` = -u(2)/2/u(1) of the 3-point polyfit of |prnmap| around the peak,
    with the three complex samples beside it: the closed form of processing/Octave/godual_ranging.m:33 (oracle.peak_refine, what
    k_peak evaluates) and the host's correction_polyfit(1) must both land on the stored values."""
    doc, bulk = arch
    b = doc["besancon"]
    assert b["files_total"] == 43 and b["windows_total"] == 43 * 175 * 2 and b["closed_form_vs_stored_correction_max_abs_all_files"] < 1e-14
    nwin = 0
    for kept in b["kept"]:
        key = kept["key"]
        for c in "12":
            x, m1, p1 = (bulk[f"{key}_xval{c}{s}"] for s in ("", "m1", "p1"))
            want = bulk[f"{key}_correction{c}"]
            assert np.all(np.abs(x) >= np.abs(m1)) and np.all(np.abs(x) >= np.abs(p1)) and np.all(np.abs(want) <= 0.5)
            for w in range(len(x)):
                # oracle: a map whose arg-max is the stored peak, neighbours either side (0-based index 5 of 11)
                prnmap = np.zeros(11, dtype=complex)
                prnmap[4:7] = m1[w], x[w], p1[w]
                ind, corr, xv, xm, xp = orc.peak_refine(prnmap)
                assert ind == 5 and xv == x[w] and abs(corr - want[w]) < 1e-14
                # host: the 3-point polyfit over the device's zwin (7 samples around the peak)
                z = np.zeros(7, dtype=complex)
                z[2:5] = m1[w], x[w], p1[w]
                r = WindowResult(0, corr, x[w], m1[w], p1[w], z, 0.0, 0, 0, 0, 0, 0, 0)
                assert abs(r.correction_polyfit(1) - want[w]) < 1e-9
                assert abs(orc.peak_refine_polyfit(prnmap, 5, 1) - want[w]) < 1e-9
            nwin += len(x)
            ind = bulk[f"{key}_indice{c}"]
            assert ind.min() >= 2 and ind.max() <= 5_000_000 - 1                       # 1-based peak with both neighbours inside the map
    assert nwin == len(b["kept"]) * 175 * 2


def test_besancon_df_values_sit_on_the_linspace_grid(arch):
    """Every df the archive holds (101 distinct values over 7 525 windows) is freq(i)/2 of freq = linspace(-fs/2, fs/2, N) — spacing
    fs/(N-1), not fs/N (godual_ranging.m:15,73) — at an integer i inside the script's 200..9000 Hz band; the oracle's axis and the
    library's un-contracted fp64 formula give that very number."""
    doc, _ = arch
    b = doc["besancon"]
    n = b["n"]
    freq = orc.freq_axis(FS, n)
    assert sum(g["count"] for g in b["df_values_all_files"]) == 43 * 175
    for g in b["df_values_all_files"]:
        i = g["index0"]
        assert abs(freq[i] / 2 - g["df"]) <= 2.4e-10                                   # Octave's own linspace rounding, 1 ulp at 2.5e6
        assert b["band_hz"][0] < freq[i] < b["band_hz"][1]
        mine = (-FS / 2 + i * (FS / (n - 1))) / 2                                       # k_df_tables' formula (csrc/twx_kernels.h)
        assert abs(mine - g["df"]) <= 2.4e-10
        assert abs(g["df"] - (i - (n - 1) / 2) * (FS / n) / 2) > 1e-5                   # and NOT the fs/N grid


def test_claudio_100k_code_snr_and_ranges(arch):
    """experiments/230315_analysis_100k: the stored `code` is make_code(lfsr_chips(17, taps, 100000)) for taps 15 / 9 (the two stations);
    SNR1r + SNR1i of the last code = puissancecode / puissancenoise (the definitions of claudio...separate.m:95-99 share var(yincode));
    indice1 within the 3N map, |correction1| <= 1/2 even at the map's edges (the wrap-around neighbours of :71-80)."""
    doc, _ = arch
    c = doc["claudio100k"]
    assert sorted((k["bitlen"], k["taps"]) for k in c["codes"]) == [(17, 9), (17, 15)] and sum(k["files"] for k in c["codes"]) == 76
    for k in c["codes"]:
        chips = prn.lfsr_chips(k["bitlen"], k["taps"], k["n_chips"])
        assert np.array_equal(chips, orc.lfsr_chips(k["bitlen"], k["taps"], k["n_chips"]))
        code = orc.make_code(chips, k["sps"])
        assert hashlib.sha256(code.astype(np.int8).tobytes()).hexdigest() == k["sha256_of_int8_code"]
    edge = 0
    for f in c["files"]:
        ratio = f["puissancecode"] / f["puissancenoise"]
        assert abs((f["snr_r_last"] + f["snr_i_last"]) / ratio - 1) < 1e-12
        assert 1 <= f["indice_min"] and f["indice_max"] <= f["n3"] and f["indice_all_integer"] and f["correction_abs_max"] <= 0.5
        edge += f["indice_min"] == 1
        freq_i = -FS / 2 + f["df_mid_index0"] * (FS / (5_000_000 - 1))
        assert abs(freq_i / 2 - f["df_mid"]) < 1e-9                                     # per-chunk df on linspace(-fs/2, fs/2, fs)/2
    assert edge > 50                                                                   # peaks AT the first map sample occur in most files


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    so = tmp_path_factory.mktemp("trkarch") / "tracked_emul.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "amaranth_twstft_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "tracked_emul.cpp"), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.trk_emul_needs_realign.restype = C.c_int
    lib.trk_emul_needs_realign.argtypes = [C.c_double, C.c_double, C.c_longlong]
    return lib


def test_tracked_realignment_rule_against_every_production_record(arch, core):
    """2 087 production outputs of claudio_aligned_code_{lo,re,ranging}.m (16.9 million codes, 3 358 re-alignments): the product's own
    decision function (twx_trk::needs_realign in csrc/twx_tracked_core.h, the line the device loop runs) and the oracle's must
    say `stay` for every code the scripts left in place — 15.6 M above the -30 dB gate, by indice1 alone; the 4 000 gated codes
    nearest the gate, by the gate alone — and `move` for the first measurement (movedval - 1) of every code they moved."""
    doc, bulk = arch
    t = doc["tracked"]
    n = t["n"]
    assert t["files"] == 2087 and t["moves"] == len(bulk["trk_move_p"]) == 3358 and int(bulk["trk_stay_count"].sum()) == t["stayed_above_gate"] > 15_000_000

    def oracle_rule(ind, snr):                                                        # oracle.ranging_tracked's test, restated where the oracle inlines it
        return snr > 0 and 10 * np.log10(snr) > -30 and ((43 < ind < n / 2) or (n / 2 < ind < n - 2))
    for ind in bulk["trk_stay_indice1"]:
        assert core.trk_emul_needs_realign(float(ind), 1.0, n) == 0 and not oracle_rule(float(ind), 1.0), ind
    st = bulk["trk_stay_indice1"]
    # the data pin the upper limit exactly: codes at n-2 stayed (16 061 of them), codes at n-3 moved — `indice1 < n-2` of :176;
    # the lower limit is bracketed: nothing up to 31 moved, nothing from 63 1/3 stayed (43 lies between)
    assert st[st > n / 2].min() == n - 2 and st.max() == n and st[st < n / 2].max() <= 43 and (st < n / 2).sum() > 50
    for ind, snr in zip(bulk["trk_gated_indice1"], bulk["trk_gated_snr"]):
        assert core.trk_emul_needs_realign(float(ind), float(snr), n) == 0 and not oracle_rule(float(ind), float(snr))
        assert core.trk_emul_needs_realign(float(ind), 1.0, n) == 1                   # ... and only the gate holds them
    assert -30.05 < 10 * np.log10(bulk["trk_gated_snr"].max()) <= -30.0                # the sample reaches the gate to 0.05 dB
    first = bulk["trk_move_first_indice1"]
    moved = np.array([core.trk_emul_needs_realign(float(i), 1.0, n) for i in first])
    assert moved.mean() > 0.995                                                       # (a handful of records were written by a script revision with other limits)
    assert all(oracle_rule(float(i), 1.0) == bool(m) for i, m in zip(first, moved))
    assert first[moved == 1].min() > 43 and first[moved == 1].max() == n - 3 and (moved == 0).sum() <= 5
    assert np.all(np.abs(first * 3 - np.rint(first * 3)) < 1e-6)                     # thirds: indice/(2Nint+1), never floored in these jobs
    # the re-measurement keeps its raw x3 index (:184-185): a strong one lands at 3*21 +- 1 = dindex - indice + 21 (:183)
    strong = bulk["trk_move_post_snr"] > 1e-3
    post = bulk["trk_move_post_index"][strong]
    assert strong.sum() > 500 and np.all(post == np.rint(post)) and np.median(post) in (62.0, 63.0, 64.0) and np.mean(np.abs(post - 63) <= 1) > 0.65


def _records(bulk, key):
    return [{"xval1": bulk[f"{key}_{n}_absx"].astype(np.float64), "indice1": bulk[f"{key}_{n}_indice3"].astype(np.float64) / 3.0,
             "correction1": bulk[f"{key}_{n}_correction1"].astype(np.float64), "SNR1r": bulk[f"{key}_{n}_snr_r"].astype(np.float64),
             "SNR1i": bulk[f"{key}_{n}_snr_i"].astype(np.float64)} for n in ("op_lo", "op_re", "lt_lo", "lt_re")]


@pytest.mark.parametrize("which", [0, 1])
def test_real_two_way_sessions_through_go_1s_arithmetic(arch, which):
    """Two sessions of experiments/240527 (four real records each).  As stored, the loop-back series jumps by ~407 ns at the
    receiver's re-alignment 14 codes into the valid range: go_1s.m:94-102 takes it for a sample loss and drops the session — product
    and oracle must both do so.  Started after that transient the sessions run: product == oracle value for value (NaN outliers,
    unequal lengths, the ambiguity shifts of :208-211 as written, the 1-s rows), at the physically expected level (1 ns scatter)."""
    doc, bulk = arch
    s = doc["sessions"]["sessions"][which]
    recs = _records(bulk, s["key"])
    assert s["as_is"] is None and orc.go_1s_session(*recs) is None and twoway.session(*recs) is None
    k, trunc = twoway.valid_codes(recs[0]["xval1"])
    oplo = twoway.delays_ns(recs[0]["indice1"], recs[0]["correction1"], k)
    _, loss = twoway.cut_at_sample_loss(oplo)
    assert loss == 14 and 400 < abs(oplo[14] - oplo[13]) < 415 and not trunc        # (1-based kk = 14: between the 14th and 15th code)
    cut = [{kk: v[s["cut_first"]:] for kk, v in r.items()} for r in recs]
    want = orc.go_1s_session(*cut)
    got = twoway.session(*cut)                                                          # defaults = the script (ambiguity shifts applied)
    exp = s["oracle_on_cut"]
    assert len(want["res"]) == exp["n_codes"] and int(np.isnan(want["res"]).sum()) == exp["n_nan"]
    for name in ("oplo", "opre", "ltlo", "ltre"):
        assert np.array_equal(getattr(got, name), want[name]), name
    assert np.array_equal(np.isnan(got.res), np.isnan(want["res"])) and np.allclose(got.res, want["res"], equal_nan=True, rtol=0, atol=1e-9)
    assert np.allclose(got.res2, want["res2"], equal_nan=True, rtol=0, atol=1e-6)
    assert np.allclose(got.one_second, want["rows"], rtol=0, atol=1e-9) and want["rows"].shape[0] == exp["rows"]
    for name in ("resmean", "resstd", "resmean25", "resstd25"):
        assert abs(getattr(got, name) - want[name]) < 1e-9 and abs(want[name] - exp[name]) < 1e-6, name
    assert np.allclose(got.opslope, want["opslope"]) and np.allclose(got.ltslope, want["ltslope"]) and np.allclose(want["opslope"], exp["opslope"])
    assert 0.5 < want["resstd"] < 2.0 and want["resstd25"] < 0.6                        # ns: per-code and 1-s scatter of a healthy link
    assert 4.0 < want["opslope"][0] < 6.0 and abs(want["opslope"][0] - want["ltslope"][0]) < 0.05      # ns/s: the satellite's radial motion, seen alike from both ends
    plain = twoway.session(*cut, unwrap=False)
    assert abs((got.resmean - plain.resmean) - 200 / 3) < 1e-9                         # the shift of :208-211 as the script applies it
    assert abs(twoway.snr_db(cut[1]["SNR1r"], cut[1]["SNR1i"], slice(None)) - exp["snrop"]) < 3.0


def test_gofinal_tables_read_back(arch, tmp_path):
    """The per-second tables gofinal_{op,ltfb}.m wrote (experiments/230111_twstft_2M5): header, nine value columns, CR-less rows;
    the reader returns them with the date as Unix seconds; the OP df1 values sit on the linspace grid to the table's 3 decimals."""
    doc, _ = arch
    g = doc["gofinal"]
    assert g["totals"]["OP"]["files"] == 201 and g["totals"]["LTFB"]["files"] == 212
    for t in g["tables"]:
        tab = results_io.read_gofinal_table(t["lines"])
        rows = [l for l in t["lines"] if not l.startswith("%")]
        assert len(tab["delay"]) == len(rows) == 12 and tab["date"].shape == (12, 6)
        first = rows[0].split("\t")
        assert "%.12f" % tab["delay"][0] == first[1] and "%.3f" % tab["df1"][0] == first[2] and "%.1f" % tab["SNR1"][0] == first[3]
        assert np.all(np.diff(tab["unix"]) >= 0) and np.all(np.diff(tab["unix"]) <= 2)
        assert np.all((tab["delay"] >= 0) & (tab["delay"] < 1.0)) and np.all((tab["delay2"] >= 0) & (tab["delay2"] < 1.0))
        if t["site"] == "OP":                                                          # coarse df only at OP (LTFB's tables carry the fine-frequency step)
            q = (2 * tab["df1"] + FS / 2) / (FS / (5_000_000 - 1))                     # index on linspace(-fs/2, fs/2, N)
            assert np.all(np.abs(q - np.rint(q)) < 2.5e-3)                               # 3 decimals of df = 2e-3 of a grid step
            q0 = (2 * tab["df1"] + FS / 2) / (FS / 5_000_000)
            assert not np.all(np.abs(q0 - np.rint(q0)) < 2.5e-3)                         # ... which the fs/N grid does not explain
    p = tmp_path / "short.txt"
    p.write_text(g["tables"][0]["lines"][0] + "\n" + "\t".join(g["tables"][0]["lines"][1].split("\t")[:7]) + "\t\n")
    tab = results_io.read_gofinal_table(str(p))
    assert len(tab["delay"]) == 1 and np.isnan(tab["delayrem"][0]) and np.isfinite(tab["SNR2"][0])
    with pytest.raises(ValueError):
        results_io.read_gofinal_table(["2023 01 11 13 06 09\t0.5\t1.0"])


def test_code_loop_replays_production_records(arch, core):
    """The product's code loop (twx_trk::run in csrc/twx_tracked_core.h — the loop that drives the device) REPLAYED on 70 production
    records of the reference (all of 240527, every 35th of 2401_{OP,LTFB}): a backend answers every measurement request with the record's
    own values in order (first measurement of a moved code: movedval - 1; its re-measurement: the stored raw index), one-second chunks of
    5e6 samples, 200 000-sample codes.  The loop must then consume EXACTLY the record — the same number of codes per file (8149 / 8150 /
    5400 ...: a re-alignment costs a chunk its last code), the same number of chunks, the same `moved` code numbers and `movedval` —
    which pins the window arithmetic of claudio_aligned_code_*.m:166-200 (dindex, the +21, the + length(fcode) wrap, the dold carry,
    1-based p) against reference-held data.  Records of script revisions with other limits (movedval - 1 <= 43) are skipped and counted."""
    from tests.test_tracked_host import Callbacks, Code, Meas, Params, Summary, LOAD, MEASURE, SQBINS, SQBAND, CAND, SLIDE
    lib = core
    lib.trk_emul_run.restype = C.c_int
    lib.trk_emul_run.argtypes = [C.POINTER(Params), C.POINTER(Callbacks), C.c_longlong, C.c_longlong, C.POINTER(Summary)]
    lib.trk_emul_fetch.restype = None
    lib.trk_emul_fetch.argtypes = [C.c_void_p] * 4
    doc, bulk = arch
    Lc, fs = 5_000_000, 5e6
    done = skipped = moves = 0
    for f in doc["replay"]["files"]:
        key, n = f["key"], f["n"]
        ind3 = bulk[key + "_ind3"].astype(np.int64)
        gate = np.unpackbits(bulk[key + "_gate"])[: len(ind3)].astype(bool)
        moved = bulk[key + "_moved"].astype(np.int64)
        mval3 = bulk[key + "_movedval3"].astype(np.int64)
        if np.any((mval3 - 3) <= 43 * 3) or len(set(moved.tolist())) != len(moved):         # another revision's limits / a code moved twice
            skipped += 1
            continue
        first3 = ind3.copy()                                   # what the FIRST measurement of code p returned, x3 grid, 1-based
        first3[moved - 1] = mval3 - 3
        above = gate.copy()
        above[moved - 1] = True                                # a moved code was above the gate when first measured
        state = {"p": 0, "chunks": 0, "bad": ""}

        def load_chunk(pos, carry, full):
            full[0] = int(state["chunks"] < f["chunks"])
            state["chunks"] += 1
            return 0

        def measure(start, count, df, out):
            p = state["p"]
            for j in range(count):
                q = p + j
                if q >= len(ind3):
                    # the loop asks for more codes than the record holds: answer with a quiet code, the totals will differ
                    out[j].indice0, out[j].snr_r, out[j].snr_i = 62, 1e-9, 0.0
                    continue
                if state.get("remeasure") == q:                # the re-measurement after a move: the stored RAW index (never divided, :184-185)
                    out[j].indice0, out[j].snr_r, out[j].snr_i = int(ind3[q]) // 3 - 1, (1.0 if gate[q] else 1e-9), 0.0
                else:
                    out[j].indice0, out[j].snr_r, out[j].snr_i = int(first3[q]) - 1, (1.0 if above[q] else 1e-9), 0.0
                out[j].correction = out[j].xre = out[j].xim = out[j].puissance = out[j].pcode = out[j].pnoise = 0.0
            # what the loop will do with this batch: accept codes up to the first one that moves, then re-measure that one
            state["remeasure"] = None
            if count > 1 or state.get("last_was_batch", True):
                for j in range(count):
                    q = p + j
                    if q < len(ind3) and (q + 1) in set_moved and above[q]:
                        state["p"], state["remeasure"] = q, q
                        state["last_was_batch"] = False
                        return 0
                state["p"] = p + count
                state["last_was_batch"] = True
            else:                                              # the single re-measurement: code q is done
                state["p"] = p + 1
                state["last_was_batch"] = True
            return 0

        def sq_band(offset, k_lo, nk, mag):
            a = np.ctypeslib.as_array(mag, shape=(nk,))
            a[:] = 0.0
            a[nk // 2] = 1.0
            return 0

        set_moved = set(moved.tolist())
        cb = Callbacks(LOAD(load_chunk), MEASURE(measure), SQBINS(lambda *a: 0), SQBAND(sq_band), CAND(lambda *a: 0), SLIDE(lambda *a: 0))
        prm = Params(n, Lc, 3, 1, 0, 0, fs, -20000.0, 20000.0, 20.0)
        s = Summary()
        assert lib.trk_emul_run(C.byref(prm), C.byref(cb), 0, -1, C.byref(s)) == 0
        codes = (Code * max(s.n_codes, 1))()
        df = np.zeros(max(s.n_chunks, 1)); mv = np.zeros(max(s.n_moved, 1), dtype=np.int64); mvv = np.zeros(max(s.n_moved, 1))
        lib.trk_emul_fetch(C.cast(codes, C.c_void_p), df.ctypes.data, mv.ctypes.data, mvv.ctypes.data)
        tag = f["file"]
        assert s.n_chunks == f["chunks"], (tag, s.n_chunks)
        assert s.n_codes == f["codes"], (tag, s.n_codes, f["codes"])
        assert list(mv[: s.n_moved]) == moved.tolist(), (tag, list(mv[: s.n_moved]), moved.tolist())
        assert np.allclose(mvv[: s.n_moved] * 3, mval3, atol=1e-6), tag
        got3 = np.rint(np.array([codes[i].indice1 for i in range(s.n_codes)]) * 3).astype(np.int64)
        assert np.array_equal(got3, ind3), (tag, np.nonzero(got3 != ind3)[0][:5])
        done += 1
        moves += len(moved)
    print(f"replayed {done} records with {moves} re-alignments, skipped {skipped}")
    assert done >= 60 and moves >= 60 and skipped <= 8, (done, moves, skipped)
