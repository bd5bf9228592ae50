// CPU: the tracking epoch's host arithmetic (csrc/twx_track_core.h = twx_track_update[_mai]) under -fsanitize=address,undefined.
// Thousands of correlation matrices of every shape the entry accepts — clean peaks, peaks on the window's edges, flat tops, all
// zeros, NaN / infinity entries (what a stream of uninitialised samples gives), one to all periods usable — with and without the
// per-period records of the interference cancellation.  Checked: no sanitizer report, the documented invariants (cnt <= bps-1;
// updated => the state's numbers are finite when the input was; not updated => state untouched; record arrays written only on
// update and only inside their bps entries — guard words around them).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "twx_track_core.h"

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    long updated = 0, rejected = 0, total = 0;
    for (int it = 0; it < 20000; ++it) {
        const int bps = 2 + (int)(rng() % 40), nlag = 2 + (int)(rng() % 30), nl = 2 * nlag + 1;
        const int kind = (int)(rng() % 8);
        std::vector<double> cor((size_t)(bps - 1) * nl), phi(cor.size());
        bool finite = true;
        for (int p = 0; p < bps - 1; ++p) {
            int pk = (int)(rng() % nl);
            if (kind == 1) pk = (int)(rng() % 2) ? 0 : nl - 1;                // edges: never usable
            if (kind == 2) pk = 2 + (int)(rng() % (nl - 4));                  // always usable
            for (int l = 0; l < nl; ++l) {
                double v = 0.01 * U(rng) + std::exp(-0.5 * (l - pk) * (l - pk));
                if (kind == 3) v = 0.0;                                        // all zeros
                if (kind == 4) v = 1.0;                                        // flat
                if (kind == 5 && std::abs(l - pk) <= 1) v = 2.0;               // flat top: 0/0 in the narrow correlator
                if (kind == 6 && rng() % 50 == 0) { v = (rng() % 2) ? NAN : INFINITY; finite = false; }
                cor[(size_t)p * nl + l] = v;
                phi[(size_t)p * nl + l] = U(rng) - 0.5 + (kind == 7 ? 0.5 * (double)(rng() % 2) : 0.0);
            }
        }
        if (kind == 5) finite = false;
        twx_track_state st;
        memset(&st, 0, sizeof st);
        st.fs = 1e7; st.duration = 0.04; st.psbb = (rng() % 5) ? 0.7 : 0.0; st.fc = 1000.0; st.pt = (int64_t)(rng() % 400000); st.last_phi = U(rng) - 0.5;
        const twx_track_state before = st;
        twx_track_result out;
        const bool with_mai = rng() % 2;
        const int G = 4;                                                       // guard entries on both sides of every record array
        std::vector<int32_t> pk_idx((size_t)bps + 2 * G, 0x5a5a5a5a);
        std::vector<double> amp((size_t)bps + 2 * G, -77.0), ph((size_t)bps + 2 * G, -77.0);
        twx_track_mai mai{pk_idx.data() + G, amp.data() + G, ph.data() + G};
        const int rc = twx_track::track_update_impl(cor.data(), phi.data(), bps, nlag, &st, &out, 400000, with_mai ? &mai : nullptr);
        ++total;
        if (rc != 0) { printf("unexpected status %d (bps %d nlag %d)\n", rc, bps, nlag); return 1; }
        if (out.cnt < 0 || out.cnt > bps - 1) { printf("cnt %d outside 0..%d\n", out.cnt, bps - 1); return 1; }
        for (int g = 0; g < G; ++g)
            if (pk_idx[(size_t)g] != 0x5a5a5a5a || pk_idx[(size_t)bps + G + g] != 0x5a5a5a5a || amp[(size_t)g] != -77.0 || amp[(size_t)bps + G + g] != -77.0 ||
                ph[(size_t)g] != -77.0 || ph[(size_t)bps + G + g] != -77.0) { printf("record arrays written outside their bps entries\n"); return 1; }
        if (!out.updated) {
            ++rejected;
            if (memcmp(&st, &before, sizeof st) != 0) { printf("state changed by an epoch that did not update\n"); return 1; }
            if (with_mai && (pk_idx[G] != 0x5a5a5a5a || amp[G] != -77.0)) { printf("records written by an epoch that did not update\n"); return 1; }
        } else {
            ++updated;
            if (finite && !(std::isfinite(st.fc) && std::isfinite(st.df) && std::isfinite(out.gd) && std::isfinite(out.sdgd) && std::isfinite(out.pk))) {
                printf("non-finite state from finite input (kind %d bps %d nlag %d)\n", kind, bps, nlag); return 1;
            }
            if (with_mai) {
                for (int p = 0; p < bps; ++p) if (mai.pk_idx[p] < -nlag || mai.pk_idx[p] > nlag) { printf("peak lag outside the window\n"); return 1; }
                if (mai.phase[bps - 1] != 0.0 || mai.amp[bps - 1] != 0.0) { printf("the last record entry must be zero\n"); return 1; }
            }
        }
    }
    // argument errors
    twx_track_state st; memset(&st, 0, sizeof st); twx_track_result out; double z[57 * 24] = {0};
    st.fs = 1e7; st.duration = 0.04;
    if (twx_track::track_update_impl(nullptr, z, 25, 28, &st, &out) == 0 || twx_track::track_update_impl(z, z, 1, 28, &st, &out) == 0 ||
        twx_track::track_update_impl(z, z, 25, 1, &st, &out) == 0) { printf("bad arguments accepted\n"); return 1; }
    twx_track_mai half{nullptr, z, z};
    if (twx_track::track_update_impl(z, z, 25, 28, &st, &out, 400000, &half) == 0) { printf("half-filled record struct accepted\n"); return 1; }
    printf("track ok: %ld epochs, %ld updated, %ld rejected\n", total, updated, rejected);
    return updated > 1000 && rejected > 1000 ? 0 : 1;
}
