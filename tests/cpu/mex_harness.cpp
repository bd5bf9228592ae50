// mex_harness.cpp — runs the MEX gateway's mexFunction() for real, on top of tests/cpu/mex_fake/mex.h (no MATLAB/Octave):
//   mex_harness raw     capture.bin chips.bin out.bin nchan chan  klo khi|df  fs Nint [convention]
//   mex_harness complex capture.bin chips.bin out.bin nchan chan  klo khi|df  fs Nint [convention]
//   mex_harness file    capture.bin chips.bin out.bin nchan chan  klo khi|df  fs Nint [convention] [ngpu=N] [skip=S] [max=M]
// Trailing opt=<name>:<v1>[,<v2>...] arguments are call form D ('option', name, value) issued BEFORE every main call (string values for
// `replica`), extra=1 appends the five outputs of call form E ('extra') to out.bin, state=1 one more array: the carried [vitesse t0 dt].
// `raw` also takes a trailing ngpu=N (several devices from the one process, twx_multi_*); `file` hands over the PATH (call form C).
// `raw` hands the int16 capture over as fread(...,'int16=>int16') would; `complex` does what the reference scripts do
// before calling processing(): de-interleave one channel into a complex double column and remove each window's mean
// (godual_ranging.m:77-80).  klo/khi are 1-based (find() output); "df <value>" passes a scalar carrier offset instead.
// out.bin: int32 count, then per output {int32 m, n, cplx; double re[m*n]; double im[m*n] if cplx}.
// Build: g++ -std=c++17 -Itests/cpu/mex_fake -Iinclude tests/cpu/mex_harness.cpp mex/twstft_processing_mex.cpp -Lamaranth_twstft_amd -ltwstft_hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "mex.h"
#include "twstft_hip.h"

static std::vector<uint8_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}
static mxArray* scalar(double v) { mxArray* a = mxCreateDoubleMatrix(1, 1, mxREAL); a->re[0] = v; return a; }

int main(int argc, char** argv) {
    if (argc < 11) { fprintf(stderr, "usage: see the header of %s\n", __FILE__); return 2; }
    const std::string mode = argv[1];
    const std::vector<uint8_t> cap = slurp(argv[2]), chipb = slurp(argv[3]);
    const int nch = atoi(argv[5]), ch = atoi(argv[6]);
    int ai = 7;
    mxArray* kdf;
    if (!strcmp(argv[ai], "df")) { kdf = scalar(atof(argv[ai + 1])); ai += 2; }
    else { kdf = mxCreateDoubleMatrix(1, 2, mxREAL); kdf->re[0] = atof(argv[ai]); kdf->re[1] = atof(argv[ai + 1]); ai += 2; }
    const double fs = atof(argv[ai]); const int nint = atoi(argv[ai + 1]); ai += 2;
    mxArray* conv = nullptr;
    if (ai < argc && !strchr(argv[ai], '=')) { conv = new mxArray; conv->cls = mxCHAR_CLASS; conv->str = argv[ai]; ++ai; }
    double ngpu = 0, skip = -1, maxw = -1;
    std::vector<std::string> opts; bool want_extra = false, want_state = false;
    for (; ai < argc; ++ai) {
        if (!strncmp(argv[ai], "opt=", 4)) opts.push_back(argv[ai] + 4);
        else if (!strcmp(argv[ai], "extra=1")) want_extra = true;
        else if (!strcmp(argv[ai], "state=1")) want_state = true;
        else if (!strncmp(argv[ai], "ngpu=", 5)) ngpu = atof(argv[ai] + 5);
        else if (!strncmp(argv[ai], "skip=", 5)) skip = atof(argv[ai] + 5);
        else if (!strncmp(argv[ai], "max=", 4)) maxw = atof(argv[ai] + 4);
        else { fprintf(stderr, "unknown argument %s\n", argv[ai]); return 2; }
    }
    mxArray* code = new mxArray; code->cls = mxUINT8_CLASS; code->m = chipb.size(); code->n = 1; code->u8 = chipb;
    const size_t nshort = cap.size() / 2;
    const int16_t* raw = reinterpret_cast<const int16_t*>(cap.data());
    std::vector<const mxArray*> in;
    mxArray* a0 = new mxArray;
    if (mode == "file") {
        a0->cls = mxCHAR_CLASS; a0->str = "file";
        mxArray* pa = new mxArray; pa->cls = mxCHAR_CLASS; pa->str = argv[2];
        in = {a0, pa, scalar(nch), scalar(ch), kdf, code, scalar(fs), scalar(nint)};
    } else if (mode == "raw") {
        a0->cls = mxINT16_CLASS; a0->m = nshort; a0->n = 1; a0->i16.assign(raw, raw + nshort);
        in = {a0, scalar(nch), scalar(ch), kdf, code, scalar(fs), scalar(nint)};
    } else {
        const size_t n = chipb.size() * 2, nsamp = nshort / (2 * (size_t)nch), nwin = nsamp / n;
        a0->m = nwin * n; a0->n = 1; a0->cplx = true; a0->re.resize(nwin * n); a0->im.resize(nwin * n);
        for (size_t w = 0; w < nwin; ++w) {
            double mr = 0, mi = 0;
            for (size_t i = 0; i < n; ++i) {                          // d1=d(1:2:end) of [I1 Q1 I2 Q2]; d1=d1-mean(d1)
                const size_t s = ((w * n + i) * (size_t)nch + (size_t)(ch - 1)) * 2;
                a0->re[w * n + i] = raw[s]; a0->im[w * n + i] = raw[s + 1];
                mr += raw[s]; mi += raw[s + 1];
            }
            mr /= (double)n; mi /= (double)n;
            for (size_t i = 0; i < n; ++i) { a0->re[w * n + i] -= mr; a0->im[w * n + i] -= mi; }
        }
        in = {a0, kdf, code, scalar(fs), scalar(nint)};
    }
    if (conv) in.push_back(conv);
    if (ngpu > 0 || skip >= 0 || maxw >= 0) {
        in.push_back(scalar(ngpu > 0 ? ngpu : 1));
        if (skip >= 0 || maxw >= 0) in.push_back(scalar(skip >= 0 ? skip : 0));
        if (maxw >= 0) in.push_back(scalar(maxw));
    }
    mxArray* out[16] = {0};
    int nlhs = 9;
    auto str = [](const std::string& t) { mxArray* a = new mxArray; a->cls = mxCHAR_CLASS; a->str = t; return a; };
    auto set_options = [&]() {
        for (const std::string& o : opts) {
            const size_t c = o.find(':');
            const std::string name = o.substr(0, c), val = c == std::string::npos ? "" : o.substr(c + 1);
            mxArray* v;
            if (name == "replica") v = str(val);
            else {
                std::vector<double> nums;
                for (size_t p = 0; p < val.size();) { size_t q = val.find(',', p); if (q == std::string::npos) q = val.size(); nums.push_back(atof(val.substr(p, q - p).c_str())); p = q + 1; }
                v = mxCreateDoubleMatrix(1, nums.size(), mxREAL); v->re = nums;
            }
            const mxArray* oin[3] = {str("option"), str(name), v};
            mexFunction(0, nullptr, 3, oin);
        }
    };
    try {
        set_options();
        mexFunction(nlhs, out, (int)in.size(), in.data());
        set_options();                                                // (the carried state of 'vitesse' starts over: both calls do the same work)
        mexFunction(nlhs, out, (int)in.size(), in.data());            // second call: the cached context is reused, lock count stays 1
        if (want_extra) { const mxArray* ein[1] = {str("extra")}; mexFunction(5, out + nlhs, 1, ein); nlhs += 5; }
        if (want_state) { const mxArray* sin_[2] = {str("option"), str("vitesse")}; mexFunction(1, out + nlhs, 2, sin_); nlhs += 1; }
    } catch (const MexError& e) {
        fprintf(stderr, "MEX error %s: %s\n", e.id.c_str(), e.msg.c_str());
        return 3;
    }
    if (mex_fake_state().locks != 1) { fprintf(stderr, "mexLock count %d (expected 1)\n", mex_fake_state().locks); return 4; }
    FILE* f = fopen(argv[4], "wb");
    int32_t cnt = nlhs;
    fwrite(&cnt, 4, 1, f);
    for (int i = 0; i < nlhs; ++i) {
        int32_t hdr[3] = {(int32_t)out[i]->m, (int32_t)out[i]->n, out[i]->cplx ? 1 : 0};
        fwrite(hdr, 4, 3, f);
        fwrite(out[i]->re.data(), 8, out[i]->re.size(), f);
        if (out[i]->cplx) fwrite(out[i]->im.data(), 8, out[i]->im.size(), f);
    }
    fclose(f);
    if (mex_fake_state().at_exit) mex_fake_state().at_exit();         // what the host does when the MEX file is cleared
    return 0;
}
