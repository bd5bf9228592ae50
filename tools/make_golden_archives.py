#!/usr/bin/env python3
"""Golden vectors from the RESULT ARCHIVES the reference repository itself holds (build container only: reads /root/reference).

The captures behind these files are not in the repository, so nothing here can be re-run — but the stored numbers are real outputs
of the reference's scripts, and they pin arithmetic that no runnable twin exists for (the Octave-only tracked flow, the two-way
combination).  Data only is written (numbers, hashes, file names): tests/golden/ref_archives.json + ref_archives.npz.

  besancon   experiments/220616_Besancon/*.mat.gz       godual.m outputs: xval / xvalm1 / xvalp1 / correction / indice / df, 2 channels
  claudio100k experiments/230315_analysis_100k/*.mat.gz  claudioltfbremote.m outputs: per-code vectors + the 200 000-sample code
  tracked    experiments/240102_1PPS_TXsync/2401_{OP,LTFB}/*.mat.gz + 240527/{op,ltfb}: outputs of the three claudio_aligned_code_* jobs
  sessions   experiments/240527/{op,ltfb}: two complete two-way sessions (four records each) for go_1s.m's arithmetic
  gofinal    experiments/230111_twstft_2M5/{OP,LTFB}/*.txt.gz: the per-second tables gofinal_{op,ltfb}.m wrote

    python tools/make_golden_archives.py
"""
from __future__ import annotations

import glob
import gzip
import hashlib
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
EXP = os.path.join(REF, "experiments")
GOLD = os.path.join(ROOT, "tests", "golden")
FS = 5e6


def load(path):
    from scipy.io import loadmat
    m = loadmat(io.BytesIO(gzip.open(path).read()))
    return {k: np.asarray(v).reshape(-1) for k, v in m.items() if not k.startswith("__")}


def rel(path):
    return os.path.relpath(path, REF)


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def grid_index(df, n, hi):
    """Index i with linspace(-fs/2, hi, n)[i]/2 == df (nearest), and the residual."""
    freq = np.linspace(-FS / 2, hi, n)
    i = int(np.abs(freq / 2 - df).argmin())
    return i, float(freq[i] / 2 - df)


def gen_besancon(bulk):
    """godual.m (experiments/220616_Besancon/godual.m:28-50): df = freq(arg-max)/2 on linspace(-fs/2,fs/2,N), band 200..9000 Hz;
    indice = arg-max of |ifft(fft(y).*fcode)| (1-based, N = 5e6, no interpolation); correction = -u(2)/2/u(1) of the 3-point
    polyfit of the magnitudes — the parabola vertex the closed form of processing/Octave/godual_ranging.m:33 gives."""
    files = sorted(glob.glob(os.path.join(EXP, "220616_Besancon", "*.mat.gz")))
    worst, nwin, dfs = 0.0, 0, {}
    for f in files:
        m = load(f)
        for c in "12":
            x, m1, p1 = (np.abs(m[f"xval{c}{s}"]) for s in ("", "m1", "p1"))
            corr = (m1 - p1) / (m1 + p1 - 2 * x) / 2
            worst = max(worst, float(np.abs(corr - m[f"correction{c}"]).max()))
            nwin += len(x)
        for d in m["df"]:
            dfs[float(d)] = dfs.get(float(d), 0) + 1
    grid = []
    for d in sorted(dfs):
        i, r = grid_index(d, 5_000_000, FS / 2)
        grid.append({"df": d, "count": dfs[d], "index0": i, "residual": r})
    keep = files[::4]
    for j, f in enumerate(keep):
        m = load(f)
        for c in "12":
            for s in ("", "m1", "p1"):
                bulk[f"bes{j}_xval{c}{s}"] = m[f"xval{c}{s}"].astype(np.complex128)
            bulk[f"bes{j}_correction{c}"] = m[f"correction{c}"].astype(np.float64)
            bulk[f"bes{j}_indice{c}"] = m[f"indice{c}"].astype(np.int64)
        bulk[f"bes{j}_df"] = m["df"].astype(np.float64)
    return {"source": "experiments/220616_Besancon/*.mat.gz (outputs of godual.m in that directory)", "files_total": len(files), "windows_total": nwin,
            "closed_form_vs_stored_correction_max_abs_all_files": worst, "n": 5_000_000, "band_hz": [200, 9000],
            "df_values_all_files": grid, "kept": [{"file": rel(f), "sha256_16": sha(f), "key": f"bes{j}"} for j, f in enumerate(keep)]}


def identify_code(code):
    from amaranth_twstft_amd import prn
    assert np.array_equal(code[0::2], code[1::2]) and set(np.unique(code)) == {-1.0, 1.0}
    chips = ((code[0::2] + 1) / 2).astype(np.uint8)
    for bitlen in (17, 22, 16, 18):
        for taps in range(1, 256):
            try:
                c = prn.lfsr_chips(bitlen, taps, len(chips))
            except Exception:
                continue
            if np.array_equal(c, chips):
                return bitlen, taps
    raise AssertionError("code not an LFSR sequence of the searched family")


def gen_claudio100k(bulk):
    """claudioltfbremote.m in experiments/230315_analysis_100k (an early version of the tracked flow): per 40-ms code xval1 /
    indice1 / correction1 / SNR1r / SNR1i / puissance1, per 1-s chunk df on linspace(-fs/2,fs/2,fs)/2; `code` is the ±1 replica at
    2 samples per chip; puissancecode / puissancenoise are those of the LAST code measured."""
    files = sorted(glob.glob(os.path.join(EXP, "230315_analysis_100k", "*.mat.gz")))
    codes, rows = {}, []
    for f in files:
        m = load(f)
        h = hashlib.sha256(m["code"].astype(np.int8).tobytes()).hexdigest()
        if h not in codes:
            bitlen, taps = identify_code(m["code"])
            codes[h] = {"bitlen": bitlen, "taps": taps, "n_chips": len(m["code"]) // 2, "sps": 2, "sha256_of_int8_code": h, "files": 0}
        codes[h]["files"] += 1
        n3 = 3 * len(m["code"])
        i0, r0 = grid_index(float(m["df"][len(m["df"]) // 2]), 5_000_000, FS / 2)
        rows.append({"file": rel(f), "code": h[:12], "n_codes": int(len(m["indice1"])), "n_chunks": int(len(m["df"])),
                     "snr_r_last": float(m["SNR1r"][-1]), "snr_i_last": float(m["SNR1i"][-1]), "puissancecode": float(m["puissancecode"][0]),
                     "puissancenoise": float(m["puissancenoise"][0]), "indice_min": float(m["indice1"].min()), "indice_max": float(m["indice1"].max()),
                     "indice_all_integer": bool(np.all(m["indice1"] == np.floor(m["indice1"]))), "correction_abs_max": float(np.abs(m["correction1"]).max()),
                     "df_mid": float(m["df"][len(m["df"]) // 2]), "df_mid_index0": i0, "df_mid_residual": r0, "n3": n3})
    return {"source": "experiments/230315_analysis_100k/*.mat.gz (outputs of claudioltfbremote.m in that directory)", "codes": list(codes.values()), "files": rows}


def gen_tracked(bulk):
    """The production outputs of claudio_aligned_code_{lo,re,ranging}.m — ALL of them: `moved` (1-based code numbers whose window was
    re-aligned) and `movedval` (= the first measurement's indice1 + 1, acquisition/claudio_aligned_code_ranging_separate.m:178-179);
    indice1 of a code NOT moved is indice/(2Nint+1); of a moved one the re-measurement's raw 1-based index on the x3 grid (:184-185
    never divide it).  The re-alignment test (:175-176) depends on (indice1, SNR > -30 dB) only, so the 17 million codes are kept as
      stay_*   histogram of indice1 over the codes that stayed although measured above -30 dB,
      gated_*  the stayed codes at or below -30 dB whose indice1 WOULD have moved them (the 4000 closest to the gate),
      move_*   every move: first measurement's indice1 (movedval - 1), the re-measurement's raw index, its SNR, the code length."""
    files = sorted(glob.glob(os.path.join(EXP, "240527", "*", "*.mat.gz"))) + sorted(glob.glob(os.path.join(EXP, "240102_1PPS_TXsync", "2401_*", "*.mat.gz")))
    hist = {}
    gated_i, gated_s, mv_pre, mv_post, mv_snr, mv_p = [], [], [], [], [], []
    nfile = ncodes = 0
    for f in files:
        m = load(f)
        if "moved" not in m or "xval1" not in m:
            continue
        nfile += 1
        n = len(m["code"])
        assert n == 200000
        ind = m["indice1"]
        ncodes += len(ind)
        mv = m["moved"].astype(np.int64)
        snr = m["SNR1r"] + m["SNR1i"]
        stay = np.ones(len(ind), bool)
        stay[mv - 1] = False
        with np.errstate(divide="ignore", invalid="ignore"):
            db = 10 * np.log10(snr)
        above = stay & (db > -30)
        u, c = np.unique(ind[above], return_counts=True)
        for x, k in zip(u, c):
            hist[float(x)] = hist.get(float(x), 0) + int(k)
        rng = ((ind > 43) & (ind < n / 2)) | ((ind < n - 2) & (ind > n / 2))
        g = stay & ~(db > -30) & rng
        gated_i.append(ind[g]); gated_s.append(snr[g])
        mv_pre.append(m["movedval"] - 1); mv_post.append(ind[mv - 1]); mv_snr.append(snr[mv - 1]); mv_p.append(mv)
    gi, gs = np.concatenate(gated_i), np.concatenate(gated_s)
    order = np.argsort(-gs)[:4000]
    bulk["trk_stay_indice1"] = np.array(sorted(hist), dtype=np.float64)
    bulk["trk_stay_count"] = np.array([hist[k] for k in sorted(hist)], dtype=np.int64)
    bulk["trk_gated_indice1"], bulk["trk_gated_snr"] = gi[order].astype(np.float64), gs[order].astype(np.float64)
    bulk["trk_move_first_indice1"] = np.concatenate(mv_pre).astype(np.float64)
    bulk["trk_move_post_index"] = np.concatenate(mv_post).astype(np.float64)
    bulk["trk_move_post_snr"] = np.concatenate(mv_snr).astype(np.float64)
    bulk["trk_move_p"] = np.concatenate(mv_p).astype(np.int32)
    return {"source": "experiments/240527/{op,ltfb}/*.mat.gz and experiments/240102_1PPS_TXsync/2401_{OP,LTFB}/*.mat.gz (all files with a `moved` variable)",
            "files": nfile, "codes_total": int(ncodes), "n": 200000, "stayed_above_gate": int(bulk["trk_stay_count"].sum()),
            "stayed_gated_in_move_range": int(len(gi)), "moves": int(len(bulk["trk_move_p"]))}


def compact(rec, lo=0, hi=None):
    """A tracked record in the form the session test needs: |xval1|, 3*indice1 as integers (the stored values are thirds or raw x3
    indices), correction1, SNR — float32 where go_1s.m only thresholds or takes medians."""
    sl = slice(lo, hi)
    i3 = np.rint(rec["indice1"][sl] * 3)
    assert np.abs(i3 - rec["indice1"][sl] * 3).max() < 1e-6
    return {"absx": np.abs(rec["xval1"][sl]).astype(np.float32), "indice3": i3.astype(np.int32), "correction1": rec["correction1"][sl].astype(np.float32),
            "snr_r": rec["SNR1r"][sl].astype(np.float32), "snr_i": rec["SNR1i"][sl].astype(np.float32)}


def expand(bulk, key):
    """The fixture form back into a record (what tests/test_ref_archives.py does, too)."""
    return {"xval1": bulk[key + "_absx"].astype(np.float64), "indice1": bulk[key + "_indice3"].astype(np.float64) / 3.0,
            "correction1": bulk[key + "_correction1"].astype(np.float64), "SNR1r": bulk[key + "_snr_r"].astype(np.float64), "SNR1i": bulk[key + "_snr_i"].astype(np.float64)}


def gen_sessions(bulk):
    """Two sessions (OP local / OP remote / LTFB local / LTFB remote) for acquisition/go_1s.m:77-268, restated by oracle.go_1s_session.
    As they stand the loop-back series jump by ~407 ns at the receiver's re-alignment, 14 codes into the valid range: go_1s.m:94-101
    reads that as a sample loss and :102 drops the session (every session of the 2401 and 240527 archives tried ends that way).  Cut to
    start after that transient (first 450 codes removed from all four records) the sessions run to the end; the second one is kept
    only up to code 1400 (a shorter session with another shape).  The expected numbers are the ORACLE's on the fixture data."""
    from oracle import twstft_oracle as orc
    root = os.path.join(EXP, "240527")
    out = []
    for j, (t, hi) in enumerate((("171680781", None), ("171681033", 1400))):
        paths = [sorted(glob.glob(os.path.join(root, d, f"{kind}claudio{t}*")))[0] for d, kind in (("op", "local"), ("op", "remote"), ("ltfb", "local"), ("ltfb", "remote"))]
        names = ("op_lo", "op_re", "lt_lo", "lt_re")
        for name, p in zip(names, paths):
            for k, v in compact(load(p), 0, hi).items():
                bulk[f"ses{j}_{name}_{k}"] = v
        recs = [expand(bulk, f"ses{j}_{n}") for n in names]
        full = orc.go_1s_session(*recs)
        g = orc.go_1s_session(*[{k: v[450:] for k, v in r.items()} for r in recs])
        out.append({"key": f"ses{j}", "files": [rel(p) for p in paths], "codes_kept": [0, hi], "as_is": None if full is None else "runs", "cut_first": 450,
                    "oracle_on_cut": {"n_codes": int(len(g["res"])), "n_nan": int(np.isnan(g["res"]).sum()), "resmean": g["resmean"], "resstd": g["resstd"],
                                      "resmean25": g["resmean25"], "resstd25": g["resstd25"], "snrop": g["snrop"], "snrlt": g["snrlt"], "rows": int(g["rows"].shape[0]),
                                      "first_row": [float(x) for x in g["rows"][0]], "opslope": [float(x) for x in g["opslope"]], "ltslope": [float(x) for x in g["ltslope"]]}})
    return {"source": "experiments/240527/{op,ltfb}/{local,remote}claudio*.mat.gz", "sessions": out}


def gen_replay(bulk):
    """Whole production records for a replay of the code loop (tests/test_ref_archives.py drives csrc/twx_tracked_core.h with them): per code
    the measured index on the x3 grid (round(3 indice1): thirds for codes that stayed, the raw re-measured index for moved ones) and whether
    it was above the -30 dB gate; per file `moved` (1-based code numbers), round(3 movedval) and the number of 1-s chunks (= length(df)).
    All ten files of 240527 and every 35th of 2401_{OP,LTFB}."""
    files = sorted(glob.glob(os.path.join(EXP, "240527", "*", "*.mat.gz"))) + sorted(glob.glob(os.path.join(EXP, "240102_1PPS_TXsync", "2401_*", "*.mat.gz")))[::35]
    out = []
    for f in files:
        m = load(f)
        if "moved" not in m or "xval1" not in m:
            continue
        i = len(out)
        i3 = np.rint(m["indice1"] * 3)
        assert np.abs(i3 - m["indice1"] * 3).max() < 1e-6
        snr = m["SNR1r"] + m["SNR1i"]
        with np.errstate(divide="ignore", invalid="ignore"):
            gate = 10 * np.log10(snr) > -30
        bulk[f"rp{i}_ind3"] = i3.astype(np.int32)
        bulk[f"rp{i}_gate"] = np.packbits(gate)
        bulk[f"rp{i}_moved"] = m["moved"].astype(np.int32)
        bulk[f"rp{i}_movedval3"] = np.rint(m["movedval"] * 3).astype(np.int32)
        out.append({"key": f"rp{i}", "file": rel(f), "codes": int(len(i3)), "chunks": int(len(m["df"])), "moves": int(len(m["moved"])), "n": int(len(m["code"]))})
    return {"source": "experiments/240527/{op,ltfb}/*.mat.gz (all) and every 35th file of experiments/240102_1PPS_TXsync/2401_{OP,LTFB}", "files": out}


def gen_gofinal():
    """Per-second tables written by gofinal_op.m / gofinal_ltfb.m (experiments/230111_twstft_2M5/gofinal_ltfb.m:86-89): header line,
    then `Y m d H M S <tab> delay <tab> df1 <tab> SNR1 <tab> delay2 <tab> df2 <tab> SNR2 <tab> delayrem <tab> df1rem <tab> SNR1rem`; rows
    of windows without a remote solution end after SNR2.  The first rows of four tables are kept verbatim as data; totals over ALL
    tables come from an independent split()-based read."""
    out = {"source": "experiments/230111_twstft_2M5/{OP,LTFB}/*.txt.gz", "tables": [], "totals": {}}
    for site in ("OP", "LTFB"):
        files = sorted(glob.glob(os.path.join(EXP, "230111_twstft_2M5", site, "1*.txt.gz")))        # (LTFB/ also holds an unrelated abstract.txt.gz)
        nrow = nshort = 0
        sdelay = 0.0
        for f in files:
            for ln in gzip.open(f, "rt").read().splitlines():
                if ln.startswith("%") or not ln.strip():
                    continue
                parts = ln.split("\t")
                vals = [x for x in parts[1:] if x.strip()]
                nrow += 1
                nshort += len(vals) < 9
                sdelay += float(vals[0])
        out["totals"][site] = {"files": len(files), "rows": nrow, "rows_without_remote": nshort, "sum_delay": sdelay}
        for f in (files[0], files[len(files) // 2]):
            lines = gzip.open(f, "rt").read().splitlines()[:13]
            out["tables"].append({"file": rel(f), "site": site, "lines": lines})
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    bulk = {}
    doc = {"generator": "tools/make_golden_archives.py (build container; reads the reference's result archives, writes numbers only)",
           "besancon": gen_besancon(bulk), "claudio100k": gen_claudio100k(bulk), "tracked": gen_tracked(bulk), "sessions": gen_sessions(bulk),
           "replay": gen_replay(bulk), "gofinal": gen_gofinal()}
    np.savez_compressed(os.path.join(GOLD, "ref_archives.npz"), **bulk)
    json.dump(doc, open(os.path.join(GOLD, "ref_archives.json"), "w"), indent=1)
    print("wrote", os.path.getsize(os.path.join(GOLD, "ref_archives.npz")), "bytes npz,", os.path.getsize(os.path.join(GOLD, "ref_archives.json")), "bytes json")
    print("besancon worst |closed form - stored|:", doc["besancon"]["closed_form_vs_stored_correction_max_abs_all_files"])
    print("codes:", doc["claudio100k"]["codes"])
    print("tracked:", {k: v for k, v in doc["tracked"].items() if k != "source"})
    for s in doc["sessions"]["sessions"]:
        print(s["key"], s["as_is"], s["oracle_on_cut"])
    print("replay files:", len(doc["replay"]["files"]), "moves:", sum(f["moves"] for f in doc["replay"]["files"]))
    print(doc["gofinal"]["totals"])


if __name__ == "__main__":
    main()
