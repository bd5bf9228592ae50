// twx_plans.h — registry of the compiled FFT pass kernels (one entry per plan length, per
// precision).  Each twx_inst_*.hip translation unit instantiates the kernels of one plan and
// registers type-erased launchers here; twx_api.hip picks N = N1*N2 from what is registered.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace twx {

// Per-kernel timing: while a profiling scope is open on this thread the next launch carries the scope's two
// events, which then hold the dispatch's own begin/end timestamps (the durations rocprofv3 reports), instead
// of bracketing the launch with two extra barrier packets.
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
LaunchEvents& launch_events();          // thread-local, defined in twx_api.hip
#define TWX_LAUNCH(kernel, grid, block, stream, ...)                                                              \
    do {                                                                                                          \
        ::twx::LaunchEvents& le_ = ::twx::launch_events();                                                        \
        if (le_.start) {                                                                                          \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, le_.start, le_.stop, 0, __VA_ARGS__);           \
            le_.start = le_.stop = nullptr;                                                                       \
        } else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                   \
    } while (0)

enum { IN_I16 = 0, IN_CHIPS = 1, IN_C32 = 2, IN_C64 = 3, IN_C64S = 4, IN_I16RS = 5 };
// IN_I16RS (COL_MIX only): inptr points to a ResamplePtr — the velocity-compensated window of twx_set_resample, aux = nch
struct ResamplePtr { const void* iq; const double* t0; const int* edge; double c; };
// IN_C64S: inptr points to a SplitPtr (real / imaginary double arrays), aux = element stride
struct SplitPtr { const double* re; const double* im; };

struct ColOps {
    int L, W, NT, f64;
    // mode: COL_*, intype: IN_*, aux: nch (IN_I16) / sps (IN_CHIPS); args: ColFwdArgs<T>*
    int (*fwd)(int mode, int intype, const void* inptr, int aux, const void* args, unsigned nblk, hipStream_t s);
    int (*inv)(const void* args /*ColInvArgs<T>*/, unsigned nblk, hipStream_t s);
};
struct RowOps {
    int L, NT, f64;
    int S, R[4];       // radices in forward stage order
    int (*run)(int mode, const void* args /*RowArgs<T>*/, unsigned nblk, hipStream_t s);
    int (*caf)(const void* args /*CafArgs<T>*/, unsigned nblk, hipStream_t s);
    int (*rowd)(int mode, const void* args /*RowDArgs<T>*/, unsigned nblk, hipStream_t s);   // nullptr if the plan has no DIF form
};

void register_col(const ColOps& o);
void register_row(const RowOps& o);
const ColOps* find_col(int L, int f64, int W = 0);
const RowOps* find_row(int L, int f64);
// pick N1 (column plan) × N2 (row plan) for n at the given precision
bool choose_split(long long n, int f64, const ColOps** col, const RowOps** row);

}  // namespace twx
