/* Plain-C99 client of include/twstft_hip.h: what a C/C++ host (processing/CPP/main.cpp's role) links
 * against.  Generates one synthetic window on the device, runs processing(d,k), prints the result.
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
    void* res_dev = twx_dev_alloc(sizeof(twx_result));
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
    twx_dev_free(chips_dev); twx_dev_free(iq_dev); twx_dev_free(res_dev); free(chips);
    twx_destroy(ctx);
    return r.indice0 == 3 * delay ? 0 : 1;
}
