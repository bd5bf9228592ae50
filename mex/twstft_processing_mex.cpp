// twstft_processing_mex.cpp — MEX / Octave gateway to libtwstft_hip.so (argument marshalling only).
//
// Cannot be compiled in the build image (no mex.h / mkoctfile); build on the MATLAB/Octave host:
//     mkoctfile --mex -I../include twstft_processing_mex.cpp -L../amaranth_twstft_amd -ltwstft_hip
//     mex -R2018a -I../include twstft_processing_mex.cpp -L../amaranth_twstft_amd -ltwstft_hip
//
// Usage from the scripts (drop-in for processing(d,k) of processing/Octave/godual_ranging.m:12):
//     [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval] = ...
//         twstft_processing_mex(raw_int16, nchan, chan, k_or_df, code_chips, fs, Nint)
//   raw_int16 : int16 vector as returned by fread(f, N*2*nchan, 'int16=>int16') for whole windows
//   chan      : 1-based channel, or 0 = every channel from one upload (outputs become nchan x nwin)
//   k_or_df   : [k_lo k_hi] 1-based indices into the fftshifted axis (as find(...) gives), or a scalar df (Hz)
//   code_chips: the code file bytes (0/1), before repelems
//   outputs are 1 x nwin; indice is 1-based like Octave's max().
#if __has_include("mex.h")
#include <string.h>
#include <vector>
#include "mex.h"
#include "twstft_hip.h"

static twx_ctx* g_ctx = nullptr;
static std::vector<uint8_t> g_chips;
static double g_fs = 0;
static int g_nint = -1;

static void cleanup(void) {
    if (g_ctx) { twx_destroy(g_ctx); g_ctx = nullptr; }
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs != 7) mexErrMsgIdAndTxt("twstft:args", "7 inputs expected");
    if (!mxIsInt16(prhs[0])) mexErrMsgIdAndTxt("twstft:args", "raw samples must be int16");
    const int16_t* raw = (const int16_t*)mxGetData(prhs[0]);
    const int nch = (int)mxGetScalar(prhs[1]);
    const int ch = (int)mxGetScalar(prhs[2]) - 1;
    const double fs = mxGetScalar(prhs[5]);
    const int nint = (int)mxGetScalar(prhs[6]);
    const size_t nchips = mxGetNumberOfElements(prhs[4]);
    std::vector<uint8_t> chips(nchips);
    const double* cd = mxIsDouble(prhs[4]) ? mxGetPr(prhs[4]) : nullptr;
    const uint8_t* cb = cd ? nullptr : (const uint8_t*)mxGetData(prhs[4]);
    for (size_t i = 0; i < nchips; ++i) chips[i] = cd ? (uint8_t)cd[i] : cb[i];
    if (!g_ctx || chips != g_chips || fs != g_fs || nint != g_nint) {   // context cached across calls
        cleanup();
        twx_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.fs = fs; cfg.sps = 2; cfg.nint = nint; cfg.chips = chips.data(); cfg.n_chips = (int64_t)nchips;
        cfg.convention = TWX_CONV_GODUAL; cfg.precision = TWX_F32; cfg.var_ddof = 1 /* Octave var */; cfg.snr_rot = -1; cfg.device = -1;
        int rc = twx_create(&cfg, &g_ctx);
        if (rc) mexErrMsgIdAndTxt("twstft:create", "%s", twx_last_error(nullptr));
        g_chips = chips; g_fs = fs; g_nint = nint;
        mexAtExit(cleanup);
        mexLock();
    }
    twx_info info;
    twx_get_info(g_ctx, &info);
    const int64_t nwin = (int64_t)(mxGetNumberOfElements(prhs[0]) / (size_t)(info.n * 2 * nch));
    const int nco = ch < 0 ? nch : 1;                    // chan = 0: all channels, results [window][channel]
    std::vector<twx_result> res((size_t)(nwin > 0 ? nwin * nco : 1));
    twx_band band; std::vector<double> dfv;
    const bool estimate = mxGetNumberOfElements(prhs[3]) == 2;
    if (estimate) { band.k_lo = (int64_t)mxGetPr(prhs[3])[0] - 1; band.k_hi = (int64_t)mxGetPr(prhs[3])[1] - 1; }
    else dfv.assign((size_t)(nwin > 0 ? nwin * nco : 1), mxGetScalar(prhs[3]));
    int rc = twx_process_windows(g_ctx, raw, nwin, nch, ch, estimate ? &band : nullptr, estimate ? nullptr : dfv.data(), res.data());
    if (rc) mexErrMsgIdAndTxt("twstft:process", "%s", twx_last_error(g_ctx));
    double* o[8];
    for (int i = 0; i < 8 && i < (nlhs > 0 ? nlhs : 1); ++i) { plhs[i] = mxCreateDoubleMatrix((mwSize)nco, (mwSize)nwin, mxREAL); o[i] = mxGetPr(plhs[i]); }
    mxArray* xv = nullptr;
    if (nlhs > 8) { xv = mxCreateDoubleMatrix((mwSize)nco, (mwSize)nwin, mxCOMPLEX); plhs[8] = xv; }
    for (int64_t w = 0; w < nwin * nco; ++w) {          // column-major nco x nwin == the library's [window][channel] order
        const twx_result& r = res[(size_t)w];
        const double vals[8] = {(double)r.indice0 + 1.0, r.correction, r.SNRr, r.SNRi, r.df, r.puissance, r.puissancecode, r.puissancenoise};
        for (int i = 0; i < 8 && i < (nlhs > 0 ? nlhs : 1); ++i) o[i][w] = vals[i];
#if MX_HAS_INTERLEAVED_COMPLEX
        if (xv) { mxComplexDouble* c = mxGetComplexDoubles(xv); c[w].real = r.xval[0]; c[w].imag = r.xval[1]; }
#else
        if (xv) { mxGetPr(xv)[w] = r.xval[0]; mxGetPi(xv)[w] = r.xval[1]; }
#endif
    }
}
#else
#error "mex.h not found: build this file with mkoctfile --mex or mex on the MATLAB/Octave host (see INTEGRATION.md)"
#endif
