// twstft_tracked_mex.cpp — MEX / Octave gateway to the library's tracked-ranging entry points (twx_tracked_*,
// include/twstft_hip.h): one call per capture FILE replaces everything the three production scripts do between
// fopen() and save() — acquisition/claudio_aligned_code_ranging_separate.m:143-205, claudio_aligned_code_re_separate.m
// (same lines) and claudio_aligned_code_lo_separate.m:117-164.  Argument marshalling only.
//
//   mkoctfile --mex -I../include twstft_tracked_mex.cpp -L../amaranth_twstft_amd -ltwstft_hip
//
//   [xval1,indice1,correction1,SNR1r,SNR1i,puissance1,df,moved,movedval,kbon,puissancecode,puissancenoise] = ...
//       twstft_tracked_mex(capturefile, codeb, mode [, OP [, fs [, Nint [, skip_seconds]]]])
//
//   capturefile : single-channel int16 [I Q] capture (the scripts' *_<channel>.bin)
//   codeb       : code file bytes (0/1) before repelems
//   mode        : 'ranging' | 're' | 'lo'  — which script's band / carrier rule / lag rounding (twx_tracked_defaults)
//   OP          : station flag (sign of the remote band), default 0;  fs default 5e6;  Nint default 1
//   skip_seconds: default = the script's own (30 s; 'lo': none)
// Outputs are the scripts' workspace variables: 1 x ncodes rows (xval1 complex), df 1 x nchunks, moved/movedval
// 1 x nmoved, kbon 1-based (0 = no carrier found), the two scalars of the last code.
#if __has_include("mex.h")
#include <string.h>
#include <string>
#include <vector>
#include "mex.h"
#include "twstft_hip.h"

static twx_tracked* g_trk = nullptr;
static std::vector<uint8_t> g_chips;
static double g_fs = 0;
static int g_nint = -1, g_mode = -1, g_op = -1;

static void cleanup(void) {
    if (g_trk) { twx_tracked_destroy(g_trk); g_trk = nullptr; }
}

static std::string text_arg(const mxArray* a, const char* what) {
    char b[4096] = {0};
    if (!mxIsChar(a) || mxGetString(a, b, sizeof b)) mexErrMsgIdAndTxt("twstft:args", "%s must be a string", what);
    return b;
}

static int parse_mode(const mxArray* a) {
    const std::string m = text_arg(a, "mode");
    if (m == "ranging") return TWX_TRK_RANGING;
    if (m == "re") return TWX_TRK_RE;
    if (m == "lo") return TWX_TRK_LO;
    mexErrMsgIdAndTxt("twstft:args", "mode must be 'ranging', 're' or 'lo' (got '%s')", m.c_str());
    return TWX_TRK_RANGING;
}

static void ensure_tracker(const mxArray* codeb, int mode, int op, double fs, int nint) {
    const size_t nchips = mxGetNumberOfElements(codeb);
    if (nchips == 0) mexErrMsgIdAndTxt("twstft:args", "empty code");
    std::vector<uint8_t> chips(nchips);
    const double* cd = mxIsDouble(codeb) ? mxGetPr(codeb) : nullptr;
    const uint8_t* cb = cd ? nullptr : (const uint8_t*)mxGetData(codeb);
    for (size_t i = 0; i < nchips; ++i) chips[i] = cd ? (uint8_t)cd[i] : cb[i];
    if (g_trk && chips == g_chips && fs == g_fs && nint == g_nint && mode == g_mode && op == g_op) return;   // cached across captures
    cleanup();
    twx_tracked_config cfg;
    memset(&cfg, 0, sizeof cfg);
    if (twx_tracked_defaults(mode, op, fs, &cfg)) mexErrMsgIdAndTxt("twstft:args", "bad mode / fs");
    cfg.nint = nint; cfg.chips = chips.data(); cfg.n_chips = (int64_t)nchips;
    cfg.precision = TWX_F32; cfg.device = -1;
    if (twx_tracked_create(&cfg, &g_trk)) mexErrMsgIdAndTxt("twstft:create", "%s", twx_tracked_last_error(nullptr));
    g_chips = chips; g_fs = fs; g_nint = nint; g_mode = mode; g_op = op;
    mexAtExit(cleanup);
    if (!mexIsLocked()) mexLock();
}

static mxArray* row(size_t n, mxComplexity c = mxREAL) { return mxCreateDoubleMatrix(1, (mwSize)n, c); }

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 3 || nrhs > 7) mexErrMsgIdAndTxt("twstft:args", "usage: twstft_tracked_mex(capturefile, codeb, mode [, OP [, fs [, Nint [, skip_seconds]]]])");
    const std::string path = text_arg(prhs[0], "capturefile");
    const int mode = parse_mode(prhs[2]);
    const int op = nrhs > 3 ? (int)mxGetScalar(prhs[3]) : 0;
    const double fs = nrhs > 4 ? mxGetScalar(prhs[4]) : 5e6;
    const int nint = nrhs > 5 ? (int)mxGetScalar(prhs[5]) : 1;
    ensure_tracker(prhs[1], mode, op, fs, nint);
    const int64_t skip = nrhs > 6 ? (int64_t)(mxGetScalar(prhs[6]) * fs) : -1;
    twx_tracked_summary s;
    if (twx_tracked_file(g_trk, path.c_str(), skip, -1, &s)) mexErrMsgIdAndTxt("twstft:tracked", "%s", twx_tracked_last_error(g_trk));
    std::vector<twx_tracked_code> codes((size_t)s.n_codes + 1);
    std::vector<double> df((size_t)s.n_chunks + 1), mv((size_t)s.n_moved + 1);
    std::vector<int64_t> moved((size_t)s.n_moved + 1);
    if (twx_tracked_fetch(g_trk, codes.data(), df.data(), moved.data(), mv.data())) mexErrMsgIdAndTxt("twstft:tracked", "%s", twx_tracked_last_error(g_trk));
    const size_t nc = (size_t)s.n_codes, nk = (size_t)s.n_chunks, nm = (size_t)s.n_moved;
    const int nout = nlhs > 0 ? (nlhs > 12 ? 12 : nlhs) : 1;
    for (int i = 0; i < nout; ++i) {
        switch (i) {
            case 0: {
                plhs[0] = row(nc, mxCOMPLEX);
                double *re = mxGetPr(plhs[0]), *im = mxGetPi(plhs[0]);
                for (size_t p = 0; p < nc; ++p) { re[p] = codes[p].xval[0]; im[p] = codes[p].xval[1]; }
                break;
            }
            case 1: case 2: case 3: case 4: case 5: {
                plhs[i] = row(nc);
                double* o = mxGetPr(plhs[i]);
                for (size_t p = 0; p < nc; ++p) {
                    const twx_tracked_code& c = codes[p];
                    o[p] = i == 1 ? c.indice1 : i == 2 ? c.correction1 : i == 3 ? c.SNR1r : i == 4 ? c.SNR1i : c.puissance1;
                }
                break;
            }
            case 6: { plhs[6] = row(nk); for (size_t q = 0; q < nk; ++q) mxGetPr(plhs[6])[q] = df[q]; break; }
            case 7: { plhs[7] = row(nm); for (size_t q = 0; q < nm; ++q) mxGetPr(plhs[7])[q] = (double)moved[q]; break; }
            case 8: { plhs[8] = row(nm); for (size_t q = 0; q < nm; ++q) mxGetPr(plhs[8])[q] = mv[q]; break; }
            case 9: { plhs[9] = row(1); mxGetPr(plhs[9])[0] = (double)(s.kbon + 1); break; }
            case 10: { plhs[10] = row(1); mxGetPr(plhs[10])[0] = s.puissancecode; break; }
            default: { plhs[11] = row(1); mxGetPr(plhs[11])[0] = s.puissancenoise; break; }
        }
    }
}
#else
#error "mex.h not found: build this file with mkoctfile --mex or mex on the MATLAB/Octave host (see INTEGRATION.md)"
#endif
