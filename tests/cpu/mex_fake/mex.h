/* Functional stand-in for the MEX API subset used by mex/twstft_processing_mex.cpp: mxArray is a small C++ object
 * over std::vector, so the gateway's mexFunction() can be COMPILED AND RUN without MATLAB/Octave
 * (tests/cpu/mex_harness.cpp builds the arguments, calls mexFunction and dumps the outputs; the -m gpu test
 * compares them with the ctypes path).  Header-only; errors are C++ exceptions of type MexError.
 * Not a MATLAB/Octave header: the real one comes from the host that builds the gateway. */
#ifndef TWX_TEST_MEX_FAKE_H
#define TWX_TEST_MEX_FAKE_H
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxINT16_CLASS = 10, mxUINT8_CLASS = 9, mxCHAR_CLASS = 4 } mxClassID;
#define MX_HAS_INTERLEAVED_COMPLEX 0

struct mxArray_tag {
    mxClassID cls = mxDOUBLE_CLASS;
    size_t m = 0, n = 0;
    bool cplx = false;
    std::vector<double> re, im;          /* separate real / imaginary storage, as Octave's mxGetPr / mxGetPi expose it */
    std::vector<int16_t> i16;
    std::vector<uint8_t> u8;
    std::string str;
};
typedef struct mxArray_tag mxArray;

struct MexError { std::string id, msg; };
struct MexFakeState { void (*at_exit)(void) = nullptr; int locks = 0; };
inline MexFakeState& mex_fake_state() { static MexFakeState s; return s; }

inline int mxIsInt16(const mxArray* a) { return a->cls == mxINT16_CLASS; }
inline int mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
inline int mxIsChar(const mxArray* a) { return a->cls == mxCHAR_CLASS; }
inline int mxIsComplex(const mxArray* a) { return a->cplx; }
inline size_t mxGetNumberOfElements(const mxArray* a) { return a->cls == mxCHAR_CLASS ? a->str.size() : a->m * a->n; }
inline size_t mxGetM(const mxArray* a) { return a->m; }
inline size_t mxGetN(const mxArray* a) { return a->n; }
inline void* mxGetData(const mxArray* a) {
    if (a->cls == mxINT16_CLASS) return (void*)a->i16.data();
    if (a->cls == mxUINT8_CLASS) return (void*)a->u8.data();
    return (void*)a->re.data();
}
inline double* mxGetPr(const mxArray* a) { return const_cast<double*>(a->re.data()); }
inline double* mxGetPi(const mxArray* a) { return a->cplx ? const_cast<double*>(a->im.data()) : nullptr; }
inline double mxGetScalar(const mxArray* a) {
    if (a->cls == mxINT16_CLASS) return a->i16.empty() ? 0.0 : (double)a->i16[0];
    if (a->cls == mxUINT8_CLASS) return a->u8.empty() ? 0.0 : (double)a->u8[0];
    return a->re.empty() ? 0.0 : a->re[0];
}
inline mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag) {
    mxArray* a = new mxArray;
    a->m = m; a->n = n; a->cplx = flag == mxCOMPLEX;
    a->re.assign(m * n, 0.0);
    if (a->cplx) a->im.assign(m * n, 0.0);
    return a;
}
inline int mxGetString(const mxArray* a, char* buf, mwSize len) {
    if (a->cls != mxCHAR_CLASS || len == 0) return 1;
    const size_t k = a->str.size() < len - 1 ? a->str.size() : len - 1;
    memcpy(buf, a->str.data(), k); buf[k] = 0;
    return a->str.size() >= len;
}
inline void mxDestroyArray(mxArray* a) { delete a; }
inline double mxGetNaN(void) { return __builtin_nan(""); }
inline void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    char b[1024]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap);
    throw MexError{id, b};
}
inline int mexAtExit(void (*fn)(void)) { mex_fake_state().at_exit = fn; return 0; }
inline void mexLock(void) { ++mex_fake_state().locks; }
inline void mexUnlock(void) { if (mex_fake_state().locks > 0) --mex_fake_state().locks; }
inline int mexIsLocked(void) { return mex_fake_state().locks > 0; }

#ifdef __cplusplus
extern "C"
#endif
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
