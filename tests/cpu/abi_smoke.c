/* Plain-C99 client of include/twstft_hip.h: what a C/C++ host (processing/CPP/main.cpp's role) links
 * against.  Generates one synthetic window on the device, runs processing(d,k), prints the result; then a 60-code capture
 * through the tracked flow (twx_tracked_*).
 *     gcc -std=c99 -Iinclude tests/cpu/abi_smoke.c -Lamaranth_twstft_amd -ltwstft_hip -o abi_smoke
 * argv: n_chips bitlen taps delay_samples */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "twstft_hip.h"

int main(int argc, char** argv) {
    const long n_chips = argc > 1 ? atol(argv[1]) : 10000;
    const int bitlen = argc > 2 ? atoi(argv[2]) : 14, taps = argc > 3 ? atoi(argv[3]) : 43;
    const long delay = argc > 4 ? atol(argv[4]) : 1234;
    const long n = 2 * n_chips;
    twx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.fs = 5e6; cfg.sps = 2; cfg.nint = 1; cfg.chips = NULL; cfg.n_chips = n_chips;
    cfg.lfsr_bitlen = bitlen; cfg.lfsr_taps = taps; cfg.convention = TWX_CONV_GODUAL; cfg.precision = TWX_F32;
    cfg.var_ddof = 1; cfg.snr_rot = -1; cfg.device = -1;
    twx_ctx* ctx = NULL;
    int rc = twx_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "twx_create: %d %s\n", rc, twx_last_error(NULL)); return 2; }
    uint8_t* chips = (uint8_t*)malloc((size_t)n_chips);
    if (twx_lfsr_chips(bitlen, taps, n_chips, chips)) return 3;
    void* chips_dev = twx_dev_alloc((size_t)n_chips);
    void* iq_dev = twx_dev_alloc((size_t)n * 4);
    void* res_dev = twx_ctx_alloc(ctx, sizeof(twx_result));      /* context-owned: released by twx_destroy at the latest */
    if (!chips_dev || !iq_dev || !res_dev) return 4;
    if (twx_memcpy_h2d(chips_dev, chips, (size_t)n_chips)) return 5;
    /* {delay_q8, fstep, phi0, amp, noise_gain, seed, stream, 0}: 0 Hz offset, amplitude 400, noise off */
    const int64_t params[8] = {delay * 256, 0, 12345, 400, 0, 7, 0, 0};
    if ((rc = twx_synth_capture_dev(iq_dev, n, 0, (const uint8_t*)chips_dev, n_chips, 2, 1, params, NULL))) { fprintf(stderr, "synth %d\n", rc); return 6; }
    const double df = 0.0;
    if ((rc = twx_process_windows_dev(ctx, iq_dev, 1, 1, 0, NULL, &df, (twx_result*)res_dev))) { fprintf(stderr, "process: %s\n", twx_last_error(ctx)); return 7; }
    if (twx_synchronize(ctx)) return 8;
    twx_result r;
    if (twx_memcpy_d2h(&r, res_dev, sizeof r)) return 9;
    printf("indice0=%lld correction=%.6f xval=%.6e%+.6ej SNRr=%.4e\n", (long long)r.indice0, r.correction, r.xval[0], r.xval[1], r.SNRr);
    /* the tracked production flow (claudio_aligned_code_lo_separate.m) from the same C host: a capture of 60 code periods,
     * generated on the device, handed over as a host buffer; chunks of 25 code periods */
    int ok = r.indice0 == 3 * delay;
    {
        const long ncodes = 60, total = n * ncodes;
        void* cap_dev = twx_dev_alloc((size_t)total * 4);
        int16_t* cap = (int16_t*)malloc((size_t)total * 4);
        const int64_t p2[8] = {delay * 256, 1030792 /* ~1200 Hz at 5 Msps */, 1, 900, 0, 9, 0, 0};
        if (!cap_dev || !cap) return 10;
        if (twx_synth_capture_dev(cap_dev, total, 0, (const uint8_t*)chips_dev, n_chips, 2, 1, p2, NULL) || twx_synchronize(ctx)) return 11;
        if (twx_memcpy_d2h(cap, cap_dev, (size_t)total * 4)) return 12;
        twx_tracked_config tc;
        memset(&tc, 0, sizeof tc);
        if (twx_tracked_defaults(TWX_TRK_LO, 0, 5e6, &tc)) return 13;
        tc.chips = chips; tc.n_chips = n_chips; tc.chunk_samples = 25 * n; tc.precision = TWX_F32; tc.device = -1;
        twx_tracked* trk = NULL;
        if ((rc = twx_tracked_create(&tc, &trk))) { fprintf(stderr, "twx_tracked_create: %s\n", twx_tracked_last_error(NULL)); return 14; }
        twx_tracked_summary sum;
        if ((rc = twx_tracked_host(trk, cap, total, 0, -1, &sum))) { fprintf(stderr, "twx_tracked_host: %s\n", twx_tracked_last_error(trk)); return 15; }
        twx_tracked_code* codes = (twx_tracked_code*)malloc(sizeof(twx_tracked_code) * (size_t)(sum.n_codes + 1));
        double dfv[8]; int64_t moved[8]; double movedval[8];
        if (sum.n_chunks > 8 || sum.n_moved > 8 || twx_tracked_fetch(trk, codes, dfv, moved, movedval)) return 16;
        printf("tracked(lo): codes=%lld chunks=%lld moved=%lld first_moved_p=%lld df0=%.3f indice1[2]=%.1f\n", (long long)sum.n_codes,
               (long long)sum.n_chunks, (long long)sum.n_moved, (long long)(sum.n_moved ? moved[0] : -1), dfv[0], codes[2].indice1);
        /* the first code sees the peak at `delay` and re-aligns the window to sample 21 (:183); every later lag is floor(64/3) = 21 */
        ok = ok && sum.n_chunks == 2 && sum.n_codes >= 48 && sum.n_moved == 1 && moved[0] == 1 && codes[2].indice1 == 21.0 && dfv[0] > 1190 && dfv[0] < 1210;
        free(codes); free(cap); twx_dev_free(cap_dev);
        twx_tracked_destroy(trk);
    }
    twx_dev_free(chips_dev); twx_dev_free(iq_dev); free(chips);
    twx_destroy(ctx);
    return ok ? 0 : 1;
}
