// twstft_processing_mex.cpp — MEX / Octave gateway to libtwstft_hip.so (argument marshalling only).
//
// Build on the MATLAB/Octave host (the build image has neither; tests run it against tests/cpu/mex_fake/mex.h):
//     mkoctfile --mex -I../include twstft_processing_mex.cpp -L../amaranth_twstft_amd -ltwstft_hip
//     mex -I../include twstft_processing_mex.cpp -L../amaranth_twstft_amd -ltwstft_hip
//
// Three call forms, told apart by the class of the first argument.
//
// (A) the reference's own signatures — `d` is the complex double column the scripts build, mean already removed:
//     [indice,correction,SNRr,SNRi,df,puissance,puissancecode,puissancenoise,xval] = ...
//         twstft_processing_mex(d, k, codeb, fs, Nint)                       % processing(d,k), godual_ranging.m:12
//     [xval,indice,correction,SNRr,SNRi,puissance,puissancecode,puissancenoise] = ...
//         twstft_processing_mex(d, df, codeb, fs, Nint, 'claudio')           % processing(d,df), claudio…separate.m:49
//   d     : n_windows*length(code) complex doubles (one or several code lengths); outputs are 1 x n_windows
//   k     : the index vector find(...) returns (1-based, fftshifted axis) or its two ends [k(1) k(end)]; a scalar
//           is a carrier offset df in Hz (one value per window also accepted when numel == n_windows ~= 2)
//   codeb : the code file bytes (0/1) before repelems;  convention 'godual' (default) or 'claudio':
//           selects fft(y).*conj(fft(code)) or fft(code).*conj(fft(y)) and the order of the outputs above.
//
// (B) raw int16 windows straight from fread (mean removal, godual_ranging.m:80, happens on the GPU):
//     [...] = twstft_processing_mex(raw_int16, nchan, chan, k_or_df, codeb, fs, Nint [, convention])
//   chan  : 1-based channel, or 0 = every channel from one upload (outputs become nchan x n_windows)
//
// (C) a whole capture FILE (the file-in / delay-out contract of godual_ranging.m:70-103: every device reads its own extent):
//     [...] = twstft_processing_mex('file', path, nchan, chan, k_or_df, codeb, fs, Nint [, convention] [, ngpu [, skip_samples [, max_windows]]])
//   k_or_df: band ends, or ONE carrier offset for the whole file;  outputs nchan(or 1) x n_windows_processed
//
// ngpu (forms B and C, a trailing numeric argument after the optional convention string; default 1): the windows are cut
// into ngpu contiguous blocks, one per device (devices 0..ngpu-1, wrapping where the box has fewer), each on its own host
// thread inside the library, and the records come back through one RCCL all-gather (twx_multi_*, include/twstft_hip.h) —
// the same outputs as with one GPU, record for record.
//
// (D) options of the cached context (one GPU), kept across calls and re-applied when the context is rebuilt:
//     twstft_processing_mex('option', 'vitesse', [vitesse t0 dt])     % the velocity-compensated window of godual_ranging_OP_vitesse.m:4,40-43,68-71
//                                                                      % (twx_set_resample; vitesse = 0: off); with one output: the carried [vitesse t0 dt]
//     twstft_processing_mex('option', 'replica', 'unipolar_zero_mean') % code=code-mean(code) on the 0/1 bytes (godual_ranging_OP_vitesse.m:7-10,
//                                                                      % 220616_Besancon/godual.m:5-7); 'bipolar' (default): 2*code-1
//     twstft_processing_mex('option', 'snr_estimators', [Lb Ls])       % bruit / noise_square lengths of process_OP.m:97,119-121 (0 0: off)
//     twstft_processing_mex('option', 'selfcheck', v)                  % TWX_OPT_SELFCHECK
// (E) what the last call left besides its outputs, one value per record:
//     [bruit, valmax_square, noise_square, status, dt] = twstft_processing_mex('extra')
//
// indice is 1-based like Octave's max(); variances use Octave's N-1 normalisation.
#if __has_include("mex.h")
#include <string.h>
#include <algorithm>
#include <vector>
#include "mex.h"
#include "twstft_hip.h"

static twx_ctx* g_ctx = nullptr;          // ngpu == 1
static twx_multi* g_multi = nullptr;      // ngpu > 1: one context + host thread per device
static std::vector<uint8_t> g_chips;
static double g_fs = 0;
static int g_nint = -1, g_conv = -1, g_ngpu = 0;
// call forms D / E
static double g_vit[3] = {0, 0, 0};
static int g_replica = 0, g_replica_built = -1, g_selfcheck = 0, g_est[2] = {0, 0};
static std::vector<twx_result> g_last;
static void apply_options(bool vitesse_too) {
    if (!g_ctx) return;
    if (twx_set_option(g_ctx, TWX_OPT_BRUIT_LEN, g_est[0]) || twx_set_option(g_ctx, TWX_OPT_NOISE_SQUARE_LEN, g_est[1])) mexErrMsgIdAndTxt("twstft:option", "%s", twx_last_error(g_ctx));
    if (g_selfcheck >= 0 && twx_set_option(g_ctx, TWX_OPT_SELFCHECK, g_selfcheck) && g_selfcheck > 0) mexErrMsgIdAndTxt("twstft:option", "%s", twx_last_error(g_ctx));
    if (vitesse_too && twx_set_resample(g_ctx, g_vit[0], g_vit[1], (int64_t)g_vit[2])) mexErrMsgIdAndTxt("twstft:option", "%s", twx_last_error(g_ctx));
}

static void cleanup(void) {
    if (g_ctx) { twx_destroy(g_ctx); g_ctx = nullptr; }
    if (g_multi) { twx_multi_destroy(g_multi); g_multi = nullptr; }
    g_ngpu = 0;
}
static twx_ctx* any_ctx(void) { return g_multi ? twx_multi_context(g_multi, 0) : g_ctx; }
static const char* last_error(void) { return g_multi ? twx_multi_last_error(g_multi) : twx_last_error(g_ctx); }

static int parse_convention(const mxArray* a) {
    char b[32] = {0};
    if (!mxIsChar(a) || mxGetString(a, b, sizeof b)) mexErrMsgIdAndTxt("twstft:args", "convention must be 'godual' or 'claudio'");
    if (!strcmp(b, "godual")) return TWX_CONV_GODUAL;
    if (!strcmp(b, "claudio")) return TWX_CONV_CLAUDIO;
    mexErrMsgIdAndTxt("twstft:args", "unknown convention '%s'", b);
    return TWX_CONV_GODUAL;
}

static void ensure_context(const mxArray* codeb, double fs, int nint, int conv, int ngpu = 1) {
    const size_t nchips = mxGetNumberOfElements(codeb);
    if (nchips == 0) mexErrMsgIdAndTxt("twstft:args", "empty code");
    std::vector<uint8_t> chips(nchips);
    const double* cd = mxIsDouble(codeb) ? mxGetPr(codeb) : nullptr;
    const uint8_t* cb = cd ? nullptr : (const uint8_t*)mxGetData(codeb);
    for (size_t i = 0; i < nchips; ++i) chips[i] = cd ? (uint8_t)cd[i] : cb[i];
    if ((g_ctx || g_multi) && chips == g_chips && fs == g_fs && nint == g_nint && conv == g_conv && ngpu == g_ngpu && g_replica == g_replica_built) return;   // cached across calls
    if (ngpu > 1 && (g_vit[0] != 0 || g_est[0] || g_est[1] || g_selfcheck > 0)) mexErrMsgIdAndTxt("twstft:args", "the options of call form D apply to one GPU (ngpu = 1)");
    cleanup();
    twx_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.fs = fs; cfg.sps = 2; cfg.nint = nint; cfg.chips = chips.data(); cfg.n_chips = (int64_t)nchips;
    cfg.convention = conv; cfg.precision = TWX_F32; cfg.var_ddof = 1 /* Octave var */; cfg.snr_rot = -1; cfg.device = -1;
    if (g_replica == 1) { cfg.code_levels = TWX_CODE_UNIPOLAR; cfg.flags |= TWX_FLAG_CODE_ZERO_MEAN; }
    if (ngpu > 1) {
        if (twx_multi_create(&cfg, nullptr, ngpu, 0, &g_multi)) mexErrMsgIdAndTxt("twstft:create", "%s", twx_multi_last_error(nullptr));
    } else if (twx_create(&cfg, &g_ctx)) mexErrMsgIdAndTxt("twstft:create", "%s", twx_last_error(nullptr));
    g_chips = chips; g_fs = fs; g_nint = nint; g_conv = conv; g_ngpu = ngpu; g_replica_built = g_replica;
    apply_options(true);                       // (a rebuilt context starts the carried t0 / dt where the last 'vitesse' option put them)
    mexAtExit(cleanup);
    if (!mexIsLocked()) mexLock();             // once: `clear mex` can unload after the exit handler has run
}

// k_or_df -> band (two ends of a 1-based index vector) or per-record carrier offsets
static bool parse_band_or_df(const mxArray* a, size_t nrec, twx_band* band, std::vector<double>* dfv) {
    const size_t ne = mxGetNumberOfElements(a);
    const double* v = mxGetPr(a);
    if (ne == 0) mexErrMsgIdAndTxt("twstft:args", "empty k / df");
    if (ne == 1) { dfv->assign(nrec ? nrec : 1, v[0]); return false; }
    if (ne == nrec && ne != 2) { dfv->assign(v, v + ne); return false; }
    band->k_lo = (int64_t)v[0] - 1; band->k_hi = (int64_t)v[ne - 1] - 1;       // find(...) is contiguous: its ends define the band
    return true;
}

static void emit(int nlhs, mxArray* plhs[], const std::vector<twx_result>& res, size_t rows, size_t cols, int conv) {
    // output order of the two reference functions
    //   godual : indice correction SNRr SNRi df puissance puissancecode puissancenoise [xval]
    //   claudio: xval indice correction SNRr SNRi puissance puissancecode puissancenoise [df]
    const int nout = nlhs > 0 ? (nlhs > 9 ? 9 : nlhs) : 1;
    const int xpos = conv == TWX_CONV_CLAUDIO ? 0 : 8;
    g_last.assign(res.begin(), res.begin() + (long)std::min(res.size(), rows * cols));
    double* o[9] = {0};
    double* oi = nullptr;
    for (int i = 0; i < nout; ++i) {
        plhs[i] = mxCreateDoubleMatrix((mwSize)rows, (mwSize)cols, i == xpos ? mxCOMPLEX : mxREAL);
        o[i] = mxGetPr(plhs[i]);
        if (i == xpos) oi = mxGetPi(plhs[i]);
    }
    for (size_t w = 0; w < rows * cols; ++w) {           // column-major rows x cols == the library's [window][channel] order
        const twx_result& r = res[w];
        const double g[8] = {(double)r.indice0 + 1.0, r.correction, r.SNRr, r.SNRi, r.df, r.puissance, r.puissancecode, r.puissancenoise};
        const double c[8] = {(double)r.indice0 + 1.0, r.correction, r.SNRr, r.SNRi, r.puissance, r.puissancecode, r.puissancenoise, r.df};
        for (int i = 0; i < nout; ++i) {
            if (i == xpos) { o[i][w] = r.xval[0]; oi[w] = r.xval[1]; }
            else o[i][w] = conv == TWX_CONV_CLAUDIO ? c[i - 1] : g[i];
        }
    }
}

static bool forms_d_e(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char what[16] = {0};
    if (nrhs < 1 || !mxIsChar(prhs[0]) || mxGetString(prhs[0], what, sizeof what)) return false;
    if (!strcmp(what, "extra")) {
        const size_t n = g_last.size();
        std::vector<twx_extra> ex(n ? n : 1);
        const bool have = g_ctx && (g_est[0] || g_est[1]) && n && twx_fetch_extra(g_ctx, ex.data(), (int64_t)n) == 0;
        const int nout = nlhs > 0 ? (nlhs > 5 ? 5 : nlhs) : 1;
        for (int i = 0; i < nout; ++i) {
            plhs[i] = mxCreateDoubleMatrix(1, (mwSize)n, mxREAL);
            double* o = mxGetPr(plhs[i]);
            for (size_t w = 0; w < n; ++w) {
                const double nanv = mxGetNaN();
                o[w] = i == 0 ? (have ? ex[w].bruit : nanv) : i == 1 ? (have ? ex[w].valmax_square : nanv) : i == 2 ? (have ? ex[w].noise_square : nanv)
                     : i == 3 ? (double)g_last[w].status : (double)g_last[w].dt;
            }
        }
        return true;
    }
    if (strcmp(what, "option")) return false;
    char name[32] = {0};
    if (nrhs < 2 || !mxIsChar(prhs[1]) || mxGetString(prhs[1], name, sizeof name)) mexErrMsgIdAndTxt("twstft:args", "usage: ('option', name [, value])");
    if (!strcmp(name, "vitesse")) {
        if (nrhs >= 3) {
            const size_t ne = mxGetNumberOfElements(prhs[2]);
            const double* v = mxGetPr(prhs[2]);
            g_vit[0] = ne > 0 ? v[0] : 0; g_vit[1] = ne > 1 ? v[1] : 0; g_vit[2] = ne > 2 ? v[2] : 0;
            if (g_ctx && twx_set_resample(g_ctx, g_vit[0], g_vit[1], (int64_t)g_vit[2])) mexErrMsgIdAndTxt("twstft:option", "%s", twx_last_error(g_ctx));
        }
        if (nlhs > 0 || nrhs < 3) {
            double v = g_vit[0], t0 = g_vit[1]; int64_t dt = (int64_t)g_vit[2];
            if (g_ctx) twx_get_resample(g_ctx, &v, &t0, &dt);
            plhs[0] = mxCreateDoubleMatrix(1, 3, mxREAL);
            double* o = mxGetPr(plhs[0]); o[0] = v; o[1] = t0; o[2] = (double)dt;
            g_vit[1] = t0; g_vit[2] = (double)dt;          // a context rebuilt later carries on from here
        }
    } else if (!strcmp(name, "replica")) {
        char r[32] = {0};
        if (nrhs < 3 || !mxIsChar(prhs[2]) || mxGetString(prhs[2], r, sizeof r)) mexErrMsgIdAndTxt("twstft:args", "replica: 'bipolar' or 'unipolar_zero_mean'");
        if (!strcmp(r, "bipolar")) g_replica = 0;
        else if (!strcmp(r, "unipolar_zero_mean")) g_replica = 1;
        else mexErrMsgIdAndTxt("twstft:args", "unknown replica '%s'", r);
    } else if (!strcmp(name, "snr_estimators")) {
        if (nrhs < 3 || mxGetNumberOfElements(prhs[2]) < 2) mexErrMsgIdAndTxt("twstft:args", "snr_estimators: [bruit_len noise_square_len]");
        g_est[0] = (int)mxGetPr(prhs[2])[0]; g_est[1] = (int)mxGetPr(prhs[2])[1];
        apply_options(false);
    } else if (!strcmp(name, "selfcheck")) {
        if (nrhs < 3) mexErrMsgIdAndTxt("twstft:args", "selfcheck: a value");
        g_selfcheck = (int)mxGetScalar(prhs[2]);
        apply_options(false);
    } else mexErrMsgIdAndTxt("twstft:args", "unknown option '%s'", name);
    return true;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (forms_d_e(nlhs, plhs, nrhs, prhs)) return;
    if (nrhs < 5) mexErrMsgIdAndTxt("twstft:args", "usage: (d, k_or_df, codeb, fs, Nint [, convention]) or (raw_int16, nchan, chan, k_or_df, codeb, fs, Nint [, convention] [, ngpu]) "
                                                   "or ('file', path, nchan, chan, k_or_df, codeb, fs, Nint [, convention] [, ngpu [, skip_samples [, max_windows]]])");
    const bool file_form = mxIsChar(prhs[0]);
    const bool raw_form = mxIsInt16(prhs[0]);
    const int base = file_form ? 4 : raw_form ? 3 : 1;   // position of k_or_df
    if (nrhs < base + 4) mexErrMsgIdAndTxt("twstft:args", "wrong number of inputs");
    const double fs = mxGetScalar(prhs[base + 2]);
    const int nint = (int)mxGetScalar(prhs[base + 3]);
    int opt = base + 4;                                  // optional tail: convention string, then numbers
    int conv = TWX_CONV_GODUAL;
    if (opt < nrhs && mxIsChar(prhs[opt])) conv = parse_convention(prhs[opt++]);
    double tail[3] = {1, -1, -1};                        // ngpu, skip_samples, max_windows
    const int ntail = nrhs - opt;
    if (ntail < 0 || ntail > (file_form ? 3 : raw_form ? 1 : 0)) mexErrMsgIdAndTxt("twstft:args", "wrong number of inputs");
    for (int i = 0; i < ntail; ++i) {
        if (mxIsChar(prhs[opt + i])) mexErrMsgIdAndTxt("twstft:args", "the convention string comes before ngpu");
        tail[i] = mxGetScalar(prhs[opt + i]);
    }
    const int ngpu = (int)tail[0];
    if (ngpu < 1 || ngpu > 64) mexErrMsgIdAndTxt("twstft:args", "ngpu must be 1..64");
    ensure_context(prhs[base + 1], fs, nint, conv, ngpu);
    twx_info info;
    twx_get_info(any_ctx(), &info);
    twx_band band; std::vector<double> dfv;
    if (file_form) {
        char what[16] = {0}, path[4096] = {0};
        if (mxGetString(prhs[0], what, sizeof what) || strcmp(what, "file")) mexErrMsgIdAndTxt("twstft:args", "a string first argument must be 'file'");
        if (!mxIsChar(prhs[1]) || mxGetString(prhs[1], path, sizeof path)) mexErrMsgIdAndTxt("twstft:args", "bad path");
        const int nch = (int)mxGetScalar(prhs[2]);
        const int ch = (int)mxGetScalar(prhs[3]) - 1;
        if (nch < 1 || ch < -1 || ch >= nch) mexErrMsgIdAndTxt("twstft:args", "bad channel");
        const int64_t skip = tail[1] > 0 ? (int64_t)tail[1] : 0;
        FILE* f = fopen(path, "rb");
        if (!f) mexErrMsgIdAndTxt("twstft:process", "cannot open %s", path);
        fseek(f, 0, SEEK_END);
        const int64_t have = ((int64_t)ftell(f) / (4 * nch) - skip) / info.n;
        fclose(f);
        const int64_t cap = have < 0 ? 0 : (tail[2] >= 0 && (int64_t)tail[2] < have ? (int64_t)tail[2] : have);
        const size_t nco = ch < 0 ? (size_t)nch : 1;
        std::vector<twx_result> res((size_t)(cap > 0 ? cap * nco : 1));
        const bool est = parse_band_or_df(prhs[4], 1, &band, &dfv);
        int64_t done = 0;
        const int rc = g_multi ? twx_multi_process_file(g_multi, path, nch, ch, skip, est ? &band : nullptr, est ? 0.0 : dfv[0], res.data(), cap, &done)
                               : twx_process_file(g_ctx, path, nch, ch, skip, est ? &band : nullptr, est ? 0.0 : dfv[0], res.data(), cap, &done);
        if (rc) mexErrMsgIdAndTxt("twstft:process", "%s", last_error());
        emit(nlhs, plhs, res, nco, (size_t)done, conv);
    } else if (raw_form) {
        const int16_t* raw = (const int16_t*)mxGetData(prhs[0]);
        const int nch = (int)mxGetScalar(prhs[1]);
        const int ch = (int)mxGetScalar(prhs[2]) - 1;
        if (nch < 1 || ch < -1 || ch >= nch) mexErrMsgIdAndTxt("twstft:args", "bad channel");
        const int64_t nwin = (int64_t)(mxGetNumberOfElements(prhs[0]) / (size_t)(info.n * 2 * nch));
        const size_t nco = ch < 0 ? (size_t)nch : 1;     // chan = 0: all channels, results [window][channel]
        std::vector<twx_result> res((size_t)(nwin > 0 ? nwin * nco : 1));
        const bool est = parse_band_or_df(prhs[3], (size_t)nwin * nco, &band, &dfv);
        if (g_multi ? twx_multi_process_windows(g_multi, raw, nwin, nch, ch, est ? &band : nullptr, est ? nullptr : dfv.data(), res.data())
                    : twx_process_windows(g_ctx, raw, nwin, nch, ch, est ? &band : nullptr, est ? nullptr : dfv.data(), res.data()))
            mexErrMsgIdAndTxt("twstft:process", "%s", last_error());
        emit(nlhs, plhs, res, nco, (size_t)nwin, conv);
    } else {
        if (!mxIsDouble(prhs[0])) mexErrMsgIdAndTxt("twstft:args", "d must be a (complex) double vector or int16 raw samples");
        const size_t ne = mxGetNumberOfElements(prhs[0]);
        if (ne == 0 || ne % (size_t)info.n) mexErrMsgIdAndTxt("twstft:args", "length(d) must be a multiple of the code length (%lld)", (long long)info.n);
        const int64_t nwin = (int64_t)(ne / (size_t)info.n);
        std::vector<twx_result> res((size_t)nwin);
        const bool est = parse_band_or_df(prhs[1], (size_t)nwin, &band, &dfv);
        std::vector<double> zero;
        int rc;
#if MX_HAS_INTERLEAVED_COMPLEX
        if (mxIsComplex(prhs[0])) {
            const double* c = (const double*)mxGetComplexDoubles(prhs[0]);
            rc = twx_process_complex(g_ctx, c, c + 1, 2, nwin, est ? &band : nullptr, est ? nullptr : dfv.data(), res.data());
        } else
#endif
        {
            const double* re = mxGetPr(prhs[0]);
            const double* im = mxIsComplex(prhs[0]) ? mxGetPi(prhs[0]) : nullptr;
            if (!im) { zero.assign(ne, 0.0); im = zero.data(); }       // a real d is a complex d with zero imaginary part
            rc = twx_process_complex(g_ctx, re, im, 1, nwin, est ? &band : nullptr, est ? nullptr : dfv.data(), res.data());
        }
        if (rc) mexErrMsgIdAndTxt("twstft:process", "%s", twx_last_error(g_ctx));
        emit(nlhs, plhs, res, 1, (size_t)nwin, conv);
    }
}
#else
#error "mex.h not found: build this file with mkoctfile --mex or mex on the MATLAB/Octave host (see INTEGRATION.md)"
#endif
