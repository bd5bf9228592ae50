// tracked_emul.cpp — the tracked-ranging control flow of the product library (amaranth_twstft_amd/csrc/twx_tracked_core.h)
// compiled for the CPU with a Backend made of C callbacks: tests/test_tracked_host.py plugs the oracle in and checks
// that the SAME loop that drives the device reproduces oracle.ranging_tracked (no GPU needed).
//   g++ -O2 -std=c++17 -shared -fPIC -Iamaranth_twstft_amd/csrc tests/cpu/tracked_emul.cpp -o tracked_emul.so
#include "twx_tracked_core.h"

extern "C" {

struct EmulParams { long long n, L; int r, carrier, indice_floor, pad; double fs, band_lo, band_hi, df_threshold; };
struct EmulCallbacks {
    int (*load_chunk)(long long pos_i16, long long carry, int* full);
    int (*measure)(long long start, int count, double df, twx_trk::Meas* out);
    int (*sq_bins)(long long n_samples, const long long* bins, int nb, double* re_im);
    int (*sq_band)(long long offset, long long k_lo, long long nk, double* mag);
    int (*candidate_snr)(long long offset, double dftmp, double* snr);
    int (*slide_tail)(long long from, long long count);
};
struct EmulSummary { long long n_codes, n_chunks, n_moved, kbon, batches; double pcode, pnoise; };

}  // extern "C"

namespace {
struct CbBackend : twx_trk::Backend {
    EmulCallbacks cb;
    int load_chunk(long long pos, long long carry, int* full) override { return cb.load_chunk(pos, carry, full); }
    int measure(long long start, int count, double df, twx_trk::Meas* out) override { return cb.measure(start, count, df, out); }
    int sq_bins(long long ns, const long long* bins, int nb, double* o) override { return cb.sq_bins(ns, bins, nb, o); }
    int sq_band(long long off, long long k_lo, long long nk, double* mag) override { return cb.sq_band(off, k_lo, nk, mag); }
    int candidate_snr(long long off, double df, double* snr) override { return cb.candidate_snr(off, df, snr); }
    int slide_tail(long long from, long long count) override { return cb.slide_tail(from, count); }
};
twx_trk::Output g_out;
}  // namespace

extern "C" {

int trk_emul_run(const EmulParams* p, const EmulCallbacks* cb, long long skip_samples, long long kbon_hint, EmulSummary* s) {
    twx_trk::Params P;
    P.n = p->n; P.L = p->L; P.r = p->r; P.fs = p->fs; P.band_lo = p->band_lo; P.band_hi = p->band_hi;
    P.carrier = p->carrier; P.indice_floor = p->indice_floor; P.df_threshold = p->df_threshold;
    CbBackend be; be.cb = *cb;
    const int rc = twx_trk::run(P, be, skip_samples, kbon_hint, g_out);
    s->n_codes = (long long)g_out.codes.size(); s->n_chunks = (long long)g_out.df.size(); s->n_moved = (long long)g_out.moved.size();
    s->kbon = g_out.kbon; s->batches = g_out.batches; s->pcode = g_out.pcode; s->pnoise = g_out.pnoise;
    return rc;
}

void trk_emul_fetch(twx_trk::Code* codes, double* df, long long* moved, double* movedval) {
    for (size_t i = 0; i < g_out.codes.size(); ++i) codes[i] = g_out.codes[i];
    for (size_t i = 0; i < g_out.df.size(); ++i) df[i] = g_out.df[i];
    for (size_t i = 0; i < g_out.moved.size(); ++i) { moved[i] = g_out.moved[i]; movedval[i] = g_out.movedval[i]; }
}

double trk_emul_freq(double fs, long long L, long long i) { return twx_trk::FreqAxis(fs, L).at(i); }
void trk_emul_band(double fs, long long L, double lo, double hi, long long* k0, long long* nk) {
    twx_trk::band_indices(twx_trk::FreqAxis(fs, L), lo, hi, k0, nk);
}
int trk_emul_needs_realign(double ind, double snr, long long n) { return twx_trk::needs_realign(ind, snr, n) ? 1 : 0; }
double trk_emul_median(const double* v, long long n) { return twx_trk::median_of(std::vector<double>(v, v + n)); }

}  // extern "C"
