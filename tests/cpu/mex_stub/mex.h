/* Declarations-only stand-in for the MEX API subset used by mex/twstft_processing_mex.cpp, so the
 * shim can be TYPE-CHECKED in an image without MATLAB/Octave (tests/test_abi_and_host.py).  Nothing
 * here is linked or executed; the real header comes from the MATLAB/Octave host. */
#ifndef TWX_TEST_MEX_STUB_H
#define TWX_TEST_MEX_STUB_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
#define MX_HAS_INTERLEAVED_COMPLEX 0
int mxIsInt16(const mxArray*);
int mxIsDouble(const mxArray*);
void* mxGetData(const mxArray*);
double* mxGetPr(const mxArray*);
double* mxGetPi(const mxArray*);
double mxGetScalar(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity flag);
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
void mexLock(void);
#ifdef __cplusplus
}
#endif
#endif
