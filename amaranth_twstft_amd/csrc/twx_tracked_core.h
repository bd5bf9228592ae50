// twx_tracked_core.h — control flow of the tracked multi-code ranging scripts, host C++ only (no HIP in this file):
//   acquisition/claudio_aligned_code_ranging_separate.m:143-205  (search_df :27-47, per-chunk carrier :166-169,
//       40-ms code loop with re-alignment :170-193, `dold` carry :196-200), its `_re_` twin (same text, other band), and
//   acquisition/claudio_aligned_code_lo_separate.m:117-164       (carrier = full-band arg-max of the fresh chunk :126-129,
//       floor() of the lag :134).
// Every operation on samples goes through `Backend`: twx_tracked.hip implements it on the device (libtwstft_hip.so),
// tests/cpu/tracked_emul.cpp with callbacks into the CPU oracle, so the SAME loop is checked against
// oracle.ranging_tracked without a GPU.
//
// Evaluation order differs from the script: the codes of a chunk are measured in one batched call on the assumption
// that the window does not move, the records are then scanned in order and the batch is cut at the first code that asks
// for a re-alignment ("coarse-parallel + serial fix-up").  The script's 1-based bookkeeping and its quirks
// are kept: `indice1` is the 1-based peak index divided by 2*Nint+1 except after a re-alignment, where the raw index of
// the second measurement stays (:174 vs :185); `kbon` indexes the axis of the ls-second chunk although d2 is taken
// over [dold; d] (:168-169); after the carrier search the capture is re-read from its start (:156-159).
#pragma once
#include <math.h>
#include <algorithm>
#include <limits>
#include <vector>

namespace twx_trk {

enum { CARRIER_SEARCH_DF = 0, CARRIER_CHUNK_BAND = 1 };

struct Params {
    long long n = 0;            // samples per code period, length(fcode)
    long long L = 0;            // samples per chunk, fs*ls (:148)
    int r = 3;                  // 2*Nint+1
    double fs = 5e6;
    double band_lo = -8000.0, band_hi = 8000.0;   // k=find((freq>lo)&(freq<hi)) (:134-141)
    int carrier = CARRIER_SEARCH_DF;
    int indice_floor = 0;       // lo script :134
    double df_threshold = 20.0; // :20
};

// one processing(d,df) record as the loop reads it (0-based arg-max in the (2Nint+1)N grid)
struct Meas { long long indice0; double correction, xre, xim, snr_r, snr_i, puissance, pcode, pnoise; };
struct Code { double xre, xim, indice1, correction1, snr_r, snr_i, puissance1; };

struct Output {
    std::vector<Code> codes;          // xval1 indice1 correction1 SNR1r SNR1i puissance1, one per code (p)
    std::vector<double> df;           // df(pfreq), one per chunk
    std::vector<long long> moved;     // p of every re-alignment (1-based, as the script stores it)
    std::vector<double> movedval;
    long long kbon = -1;              // 0-based index into the shifted axis of the chunk; -1: none (the script's 0)
    long long batches = 0;            // batched measure() calls (diagnostic)
    double pcode = std::numeric_limits<double>::quiet_NaN(), pnoise = std::numeric_limits<double>::quiet_NaN();
    void clear() { *this = Output(); }
};

// The device (or the test's oracle) side.  The backend keeps ONE sample buffer of at least L + n samples: the loop asks
// for a chunk to be placed behind `carry` samples of the previous one and addresses samples by their 0-based position.
// Every function returns 0 or a negative status that run() hands back unchanged.
struct Backend {
    virtual ~Backend() {}
    // int16 positions [pos_i16, pos_i16 + 2L) of the capture -> buffer samples [carry, carry + L); *full = 0 at a short read
    virtual int load_chunk(long long pos_i16, long long carry, int* full) = 0;
    // processing(dpart - mean(dpart), df) for `count` consecutive code periods starting at buffer sample `start`
    virtual int measure(long long start, int count, double df, Meas* out) = 0;
    // fft(d.^2) of buffer samples [0, n_samples) at the signed DFT bins `bins` -> (re, im) pairs
    virtual int sq_bins(long long n_samples, const long long* bins, int nb, double* re_im) = 0;
    // abs(fft(d.^2)) of the L samples from buffer sample `offset`, nk consecutive signed bins from k_lo
    virtual int sq_band(long long offset, long long k_lo, long long nk, double* mag) = 0;
    // search_df's test of one candidate (:34-43) on the n samples from `offset`, no mean removal: prnsig^2/var(prnmap)
    virtual int candidate_snr(long long offset, double dftmp, double* snr) = 0;
    // called once after the last candidate_snr() of a search (a backend that switched state for the search restores it)
    virtual int search_done() { return 0; }
    // buffer samples [from, from + count) -> [0, count)   (dold=d(round(dindex):end), :196-199)
    virtual int slide_tail(long long from, long long count) = 0;
};

inline long long oround(double x) {           // Octave round(): half away from zero
    const double a = floor(fabs(x) + 0.5);
    return (long long)(x >= 0 ? a : -a);
}

// freq=linspace(-fs/2,fs/2-fs/fs,fs*ls) (:132): base + i*delta, last element = the limit, as Octave and numpy form it
// (multiply, then add: no fused multiply-add)
struct FreqAxis {
    double start = 0, stop = 0, step = 0; long long L = 0;
    FreqAxis() {}
    FreqAxis(double fs, long long L_) : start(-fs / 2), stop(fs / 2 - 1.0), L(L_) { step = L > 1 ? (stop - start) / (double)(L - 1) : 0.0; }
    double at(long long i) const {
        if (i == L - 1 && L > 1) return stop;
        volatile double prod = (double)i * step;
        return prod + start;
    }
};

// k=find((freq<hi)&(freq>lo)): first index and count of the (contiguous) run, 0-based
inline void band_indices(const FreqAxis& f, double lo, double hi, long long* k0, long long* nk) {
    long long a = -1, b = -1;
    for (long long i = 0; i < f.L; ++i) {
        const double v = f.at(i);
        if (v > lo && v < hi) { if (a < 0) a = i; b = i; }
        else if (a >= 0) break;
    }
    *k0 = a; *nk = a < 0 ? 0 : b - a + 1;
}

inline double median_of(std::vector<double> v) {       // Octave/numpy median: mean of the two middle values for even counts
    const size_t m = v.size();
    if (!m) return std::numeric_limits<double>::quiet_NaN();
    std::nth_element(v.begin(), v.begin() + m / 2, v.end());
    const double hi = v[m / 2];
    if (m & 1) return hi;
    const double lo = *std::max_element(v.begin(), v.begin() + m / 2);
    return (lo + hi) / 2;
}

// search_df(d,k,df_threshold) (:27-47) on the chunk at buffer offset 0.  *kbon: 0-based shifted index or -1.
inline int search_df(const Params& P, Backend& be, const FreqAxis& freq, long long k0, long long nk, long long* kbon) {
    *kbon = -1;
    if (nk < 1) return 0;
    std::vector<double> d2k((size_t)nk);
    if (int rc = be.sq_band(0, k0 - P.L / 2, nk, d2k.data())) return rc;     // shifted index i <-> bin i - floor(L/2)
    const double thr = median_of(d2k) * P.df_threshold;
    std::vector<long long> ktmp;
    for (long long i = 0; i < nk; ++i) if (d2k[(size_t)i] > thr) ktmp.push_back(i + k0);
    if (ktmp.empty() || ktmp.size() >= 100) return 0;
    int rc = 0;
    for (long long kk : ktmp) {
        double snr = 0;
        if ((rc = be.candidate_snr(0, freq.at(kk) / 2, &snr)) != 0) break;
        if (snr > 100) *kbon = kk;
    }
    const int rc2 = be.search_done();
    return rc ? rc : rc2;
}

// The capture loop.  skip_samples: complex samples skipped before the first chunk (fseek(f,30*fs*2*2), :128);
// kbon_hint >= 0: a carrier bin known beforehand (no search).
// The re-alignment test of the code loop (claudio_aligned_code_ranging_separate.m:175-176): a code measured above -30 dB whose peak
// sits neither within the first 43 samples nor in the last two of the n-sample window makes the loop move its window.
// `ind` = indice/(2Nint+1) on the 1-based sample grid, `snr` = SNR1r + SNR1i (linear).
inline bool needs_realign(double ind, double snr, long long n) {
    return snr > 0 && 10 * log10(snr) > -30 &&
           ((ind > 43 && ind < (double)n / 2) || (ind < (double)n - 2 && ind > (double)n / 2));
}

inline int run(const Params& P, Backend& be, long long skip_samples, long long kbon_hint, Output& out) {
    out.clear();
    const long long n = P.n, Lc = P.L;
    const int r = P.r;
    const FreqAxis freq(P.fs, Lc);
    long long k0 = -1, nk = 0;
    band_indices(freq, P.band_lo, P.band_hi, &k0, &nk);
    const bool searching = P.carrier == CARRIER_SEARCH_DF;
    if (!searching && nk < 1) return -1;
    bool df_found = !searching || kbon_hint >= 0;
    if (kbon_hint >= 0) out.kbon = kbon_hint;
    long long pos = skip_samples * 2;
    long long carry = 0;                 // samples of dold at the head of the buffer
    long long p = 1;
    int guard = 0;
    std::vector<Meas> res;
    std::vector<double> mag;
    for (;;) {
        int full = 0;
        // while the carrier is unknown the chunk goes to the head of the buffer (dold is empty then)
        if (int rc = be.load_chunk(pos, df_found ? carry : 0, &full)) return rc;
        pos += 2 * Lc;
        if (!full) break;
        if (!df_found) {
            long long kb = -1;
            if (int rc = search_df(P, be, freq, k0, nk, &kb)) return rc;
            if (kb >= 0) { df_found = true; out.kbon = kb; }
            // the script closes the file and reads its first chunk again (:156-159)
            if (int rc = be.load_chunk(0, 0, &full)) return rc;
            pos = 2 * Lc;
            if (!full) break;
            ++guard;
            if (!df_found && guard > 2) break;       // the script would spin on the first chunk for ever
            if (!df_found) continue;
        }
        const long long Ld = carry + Lc;
        double df;
        if (searching) {                              // :168-169 on [dold; d]
            const long long kb = out.kbon;
            long long bins[7]; double v[14];
            for (int i = 0; i < 7; ++i) bins[i] = kb - 3 + i - Ld / 2;
            if (int rc = be.sq_bins(Ld, bins, 7, v)) return rc;
            int best = 0; double bm = -1;
            for (int i = 0; i < 7; ++i) { const double m = hypot(v[2 * i], v[2 * i + 1]); if (m > bm) { bm = m; best = i; } }
            df = freq.at(best + kb - 3) / 2;
        } else {                                      // lo :126,129 on the fresh chunk, whole band
            mag.resize((size_t)nk);
            if (int rc = be.sq_band(carry, k0 - Lc / 2, nk, mag.data())) return rc;
            long long best = 0;
            for (long long i = 1; i < nk; ++i) if (mag[(size_t)i] > mag[(size_t)best]) best = i;
            df = freq.at(best + k0) / 2;
        }
        out.df.push_back(df);
        double dindex = 1.0;
        bool done = false;
        while (!done) {
            const long long s0 = oround(dindex);
            int J = 1;
            while (dindex + (double)J * (double)n + (double)n - 1 <= (double)Ld) ++J;
            res.resize((size_t)J);
            if (int rc = be.measure(s0 - 1, J, df, res.data())) return rc;
            ++out.batches;
            bool cut = false;
            for (int j = 0; j < J && !cut; ++j) {
                Meas g = res[(size_t)j];
                double ind = (double)(g.indice0 + 1) / (double)r;          // :174
                if (P.indice_floor) ind = floor(ind);                       // lo :134
                bool stop = false;
                const double snr = g.snr_i + g.snr_r;
                if (needs_realign(ind, snr, n)) {                                                               // :175-176
                    out.moved.push_back(p);
                    out.movedval.push_back(ind + 1);
                    const double dcur = dindex + (double)j * (double)n;
                    double dnew = dcur - ind + 1 < 0 ? dcur + (double)n : dcur;   // :180-182
                    dnew = dnew - ind + 21;                                          // :183
                    const long long s1 = oround(dnew);
                    if (s1 >= 1 && s1 - 1 + n <= Ld) {
                        Meas m2;
                        if (int rc = be.measure(s1 - 1, 1, df, &m2)) return rc;     // :184-185
                        ++out.batches;
                        out.codes.push_back(Code{m2.xre, m2.xim, (double)(m2.indice0 + 1), m2.correction, m2.snr_r, m2.snr_i, m2.puissance});
                        out.pcode = m2.pcode; out.pnoise = m2.pnoise;
                        ++p;
                        dindex = dnew + (double)n;
                        done = dindex + (double)n - 1 > (double)Ld;
                        cut = true;                                                  // later codes of this batch are stale
                        break;
                    }
                    stop = true;       // the script would index past the chunk here (Octave aborts): keep the first measurement
                }
                out.codes.push_back(Code{g.xre, g.xim, ind, g.correction, g.snr_r, g.snr_i, g.puissance});
                out.pcode = g.pcode; out.pnoise = g.pnoise;
                ++p;
                if (stop) { dindex = dindex + (double)(j + 1) * (double)n; done = true; cut = true; }
            }
            if (!cut) { dindex += (double)J * (double)n; done = true; }
        }
        if (dindex < (double)Ld) {                    // dold=d(round(dindex):end) (:196-199)
            const long long s = oround(dindex) - 1;
            if (int rc = be.slide_tail(s, Ld - s)) return rc;
            carry = Ld - s;
        } else carry = 0;
    }
    return 0;
}

}  // namespace twx_trk
