// twx_track_core.h — the tracking epoch's host arithmetic (rxcomplex.cpp:620-745; the per-period records of rx.cpp:664-666,752-757):
// plain C++ over the handful of correlation results, no HIP.  Included by csrc/twx_aux.hip (twx_track_update[_mai],
// twx_track_epoch_*) and by tests/cpu/track_fuzz.cpp, which runs it under -fsanitize=address,undefined on the CPU.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/twstft_hip.h"

namespace twx_track {
// gsl_fit_wlinear: weighted least squares y = c0 + c1 x over the entries with w > 0; chisq = sum w (y - c0 - c1 x)^2
inline void fit_wlinear(const std::vector<double>& x, const std::vector<double>& w, const std::vector<double>& y, double* c0, double* c1, double* chisq) {
    double W = 0, wx = 0, wy = 0;
    for (size_t i = 0; i < x.size(); ++i) if (w[i] > 0) { W += w[i]; wx += w[i] * x[i]; wy += w[i] * y[i]; }
    const double xm = wx / W, ym = wy / W;
    double sxx = 0, sxy = 0;
    for (size_t i = 0; i < x.size(); ++i) if (w[i] > 0) { const double dx = x[i] - xm, dy = y[i] - ym; sxx += w[i] * dx * dx; sxy += w[i] * dx * dy; }
    *c1 = sxy / sxx; *c0 = ym - *c1 * xm;
    double chi = 0;
    for (size_t i = 0; i < x.size(); ++i) if (w[i] > 0) { const double d = y[i] - (*c0 + *c1 * x[i]); chi += w[i] * d * d; }
    *chisq = chi;
}
inline int track_update_impl(const double* cor, const double* phi, int bps, int nlag, twx_track_state* st, twx_track_result* out,
                      long long nobs = 0, const twx_track_mai* mai = nullptr) {
    if (!cor || !phi || !st || !out || bps < 2 || nlag < 2 || !(st->fs > 0) || !(st->duration > 0)) return TWX_E_ARG;
    if (mai && (!mai->pk_idx || !mai->amp || !mai->phase || nobs < 1)) return TWX_E_ARG;
    const int nl = 2 * nlag + 1;
    const double psbb = st->psbb != 0.0 ? st->psbb : 1.0;
    std::vector<double> res_gd((size_t)bps, 0.0), res_phi((size_t)bps, 0.0), ps((size_t)bps, 0.0), w((size_t)bps, 0.0), ttag_phi((size_t)bps, 0.0), ttag_gd((size_t)bps);
    std::vector<double> amp(mai ? (size_t)bps : 0, 0.0);
    std::vector<int> pki(mai ? (size_t)bps : 0, 0);
    memset(out, 0, sizeof *out);
    int cnt = 0;
    for (int p = 0; p < bps - 1; ++p) {
        const double* c = cor + (size_t)p * nl;
        int k = 0;                                                         // cblas_idamax: first index of the largest |value| (:630)
        for (int i = 1; i < nl; ++i) if (fabs(c[i]) > fabs(c[k])) k = i;
        ttag_phi[(size_t)p] = (double)p * st->duration + (double)st->pt / st->fs;      // :632
        ps[(size_t)p] = c[k] / psbb;                                                     // :633
        if (k - 2 >= 0 && k + 2 < nl) {                                                  // :634
            res_phi[(size_t)p] = phi[(size_t)p * nl + k];
            if (mai) { pki[(size_t)p] = k - nlag; amp[(size_t)p] = sqrt(2.0 * c[k]) / psbb; }              // rx.cpp:638,640
            res_gd[(size_t)p] = ((c[k - 1] - c[k + 1]) / (c[k - 1] - 2.0 * c[k] + c[k + 1])          // high-resolution correlator (:649-659)
                                 - (c[k - 2] - c[k + 2]) / (c[k - 2] - 2.0 * c[k] + c[k + 2])
                                 + (double)(st->pt + k - nlag)) * 1.0e+9 / st->fs;
            w[(size_t)p] = 1.0;
            ++cnt;
        }
    }
    out->cnt = cnt;                                                                      // what the "lock lost" line prints (:787)
    if (!(cnt * 2 > bps)) return TWX_OK;                                                 // :667: not enough usable periods
    std::vector<double> sel;
    for (int p = 0; p < bps; ++p) if (w[(size_t)p] > 0.0) sel.push_back(res_gd[(size_t)p]);          // :692-698
    // kth_smallest = order statistics (:840-865).  A NaN delay (0/0 on a flat-topped peak, or NaN samples) sorts last: "<" alone
    // is not an ordering once NaNs are in, and std::sort may then run off the array
    std::sort(sel.begin(), sel.end(), [](double a, double b) { return std::isnan(b) ? !std::isnan(a) : (!std::isnan(a) && a < b); });
    const int ii = (int)sel.size();
    const double med = sel[(size_t)(ii / 2)];
    const double stddev = (sel[(size_t)(ii * 3 / 4)] - sel[(size_t)(ii / 4)]) / 1.349;   // :699-700
    double last_phi = st->last_phi;
    const std::vector<double> raw_phi = mai ? res_phi : std::vector<double>();           // rx.cpp:664, before the BPSK adjustment
    cnt = 0;
    for (int p = 0; p < bps - 1; ++p) {                                                  // :703-716
        if (w[(size_t)p] == 0.0) continue;
        if (fabs(res_gd[(size_t)p] - med) < 3.0 * stddev) {
            ++cnt;
            int guard = 0;
            while (fabs(res_phi[(size_t)p] - last_phi) > 0.25 && guard++ < 1000000)
                res_phi[(size_t)p] += res_phi[(size_t)p] > last_phi ? -0.5 : 0.5;
            last_phi = res_phi[(size_t)p];
        } else w[(size_t)p] = 0.0;
    }
    // fewer than two periods left (or all at one time tag): the weighted line through them has no slope — the program would
    // write NaN into fc / df / pt; here the epoch counts as unusable and the state stays as it was
    if (cnt < 2) { out->cnt = cnt; return TWX_OK; }
    st->last_phi = last_phi;
    double c0, c1, chi;
    fit_wlinear(ttag_phi, w, res_phi, &c0, &c1, &chi);                                   // :728
    st->fc_prev = st->fc;
    st->fc += round(c1);                                                                 // :730-732
    st->df = c1 - round(c1);
    st->phi = fmod(c0 + 1000.0, 1.0);
    for (int p = 0; p < bps; ++p) ttag_gd[(size_t)p] = (double)p * st->duration;         // :410
    double g0, g1;
    fit_wlinear(ttag_gd, w, res_gd, &g0, &g1, &chi);                                     // :739
    out->freq = st->fc + st->df; out->phi = st->phi; out->cnt = cnt;
    out->sdgd = sqrt(chi / (double)cnt);                                                 // :740
    out->gd = g0 + 0.5 * g1; out->dg = g1;                                               // :741-742
    st->pt_prev = st->pt;
    st->pt = (int64_t)llround((g0 + g1) * st->fs / 1.0e+9);                              // :744
    double acc = 0; int na = 0;
    for (int p = 0; p < bps; ++p) if (w[(size_t)p] > 0.0) { acc += ps[(size_t)p]; ++na; }            // average() :887-901
    out->pk = na ? acc / (double)na : 0.0;
    out->updated = 1;
    if (mai) {                                                                           // rx.cpp:664-666,752-757
        for (int p = 0; p < bps; ++p) { mai->pk_idx[p] = pki[(size_t)p]; mai->amp[p] = amp[(size_t)p]; }
        for (int p = 0; p < bps - 1; ++p)
            mai->phase[p] = raw_phi[(size_t)p] - (st->fc + st->df - st->fc_prev) * (double)((long long)(p + 1) * nobs + st->pt_prev) / st->fs;
        mai->phase[bps - 1] = 0.0;
    }
    return TWX_OK;
}
}  // namespace twx_track
