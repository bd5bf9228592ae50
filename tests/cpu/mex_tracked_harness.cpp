// mex_tracked_harness.cpp — runs mexFunction() of mex/twstft_tracked_mex.cpp on tests/cpu/mex_fake/mex.h (no Octave):
//   mex_tracked_harness capture.bin chips.bin out.bin mode OP fs Nint [skip_seconds]
// out.bin: int32 count, then per output {int32 m, n, cplx; double re[m*n]; double im[m*n] if cplx}  (12 outputs).
// The gateway is called twice on the same capture: the second call must reuse the cached tracker (lock count 1).
// Build: g++ -std=c++17 -Itests/cpu/mex_fake -Iinclude tests/cpu/mex_tracked_harness.cpp mex/twstft_tracked_mex.cpp -Lamaranth_twstft_amd -ltwstft_hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mex.h"

static mxArray* scalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->re[0] = v; return a; }
static mxArray* text(const char* s) { mxArray* a = new mxArray; a->cls = mxCHAR_CLASS; a->str = s; return a; }

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: see the header of %s\n", __FILE__); return 2; }
    FILE* cf = fopen(argv[2], "rb");
    if (!cf) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    mxArray* code = new mxArray; code->cls = mxUINT8_CLASS;
    for (int c; (c = fgetc(cf)) != EOF;) code->u8.push_back((uint8_t)c);
    fclose(cf);
    code->m = code->u8.size(); code->n = 1;
    std::vector<const mxArray*> in = {text(argv[1]), code, text(argv[4]), scalar(atof(argv[5])), scalar(atof(argv[6])), scalar(atof(argv[7]))};
    if (argc > 8) in.push_back(scalar(atof(argv[8])));
    mxArray* out[12] = {0};
    try {
        mexFunction(12, out, (int)in.size(), in.data());
        mexFunction(12, out, (int)in.size(), in.data());
    } catch (const MexError& e) {
        fprintf(stderr, "MEX error %s: %s\n", e.id.c_str(), e.msg.c_str());
        return 3;
    }
    if (mex_fake_state().locks != 1) { fprintf(stderr, "mexLock count %d (expected 1)\n", mex_fake_state().locks); return 4; }
    FILE* f = fopen(argv[3], "wb");
    int32_t cnt = 12;
    fwrite(&cnt, 4, 1, f);
    for (int i = 0; i < 12; ++i) {
        int32_t hdr[3] = {(int32_t)out[i]->m, (int32_t)out[i]->n, out[i]->cplx ? 1 : 0};
        fwrite(hdr, 4, 3, f);
        fwrite(out[i]->re.data(), 8, out[i]->re.size(), f);
        if (out[i]->cplx) fwrite(out[i]->im.data(), 8, out[i]->im.size(), f);
    }
    fclose(f);
    if (mex_fake_state().at_exit) mex_fake_state().at_exit();
    return 0;
}
