// Instantiates the column-pass kernels of ONE plan: compile with
//   -DTWX_PLAN='Plan<625,25,25>' -DTWX_W=16 -DTWX_NT=448
#include <stdlib.h>
#include <type_traits>
#include "twx_kernels.h"
#include "twx_plans.h"

#ifndef TWX_FWD3_SQUARE
#define TWX_FWD3_SQUARE 1     // the component-wise column pass for the squared-signal pass as well: with its table values through LDS it beats k_col_fwd
                              // there too (round 6: 0.126 -> 0.098 ms per 8 windows; without them it was slower, 0.129 against 0.117 in round 3)
#endif
namespace twx {
namespace {
using P = TWX_PLAN;
using PR = typename Rev<P>::type;
constexpr int W = TWX_W;
constexpr int NT = TWX_NT;
static_assert(P::max_tasks * W <= NT, "one task per thread per stage");

template <typename T, int MODE, class In>
int launch_fwd(In in, const ColFwdArgs<T>& a, unsigned nblk, hipStream_t s) {
    // measured on MI355X (N1 = 625): round 3 MIX 0.1245 -> 0.119 ms per 8 windows, SQUARE 0.117 -> 0.129; round 6, table values through LDS: both
    // (int16 captures only: the complex-double loader of twx_process_complex needs two registers more than the 80 that six waves per SIMD leave)
    if constexpr (P::S == 2 && std::is_same<T, float>::value && NT >= 384 && (MODE == COL_MIX || (TWX_FWD3_SQUARE && MODE == COL_SQUARE && In::has_raw)) && !std::is_same<In, InCplxSplit>::value && !InTraits<In>::resample) {
        static const bool split = [] { const char* e = getenv("TWX_COLFWD3"); return !e || atoi(e) != 0; }();
        if (split) {                         // component-wise exchange: three or four workgroups per CU (twx_kernels.h)
            TWX_LAUNCH((k_col_fwd3<P, T, W, MODE, In, NT>), dim3(nblk), dim3(NT), s, in, a);
            return (int)hipGetLastError();
        }
    }
    TWX_LAUNCH((k_col_fwd<P, T, W, MODE, In, NT>), dim3(nblk), dim3(NT), s, in, a);
    return (int)hipGetLastError();
}

template <typename T> int fwd(int mode, int intype, const void* inptr, int aux, const void* args, unsigned nblk, hipStream_t s) {
    const ColFwdArgs<T>& a = *reinterpret_cast<const ColFwdArgs<T>*>(args);
    if (intype == IN_I16) {
        InI16 in{reinterpret_cast<const short2*>(inptr), aux};
        if (mode == COL_MIX) return launch_fwd<T, COL_MIX>(in, a, nblk, s);
        if (mode == COL_SQUARE) return launch_fwd<T, COL_SQUARE>(in, a, nblk, s);
        if (mode == COL_PLAIN) return launch_fwd<T, COL_PLAIN>(in, a, nblk, s);
    } else if (intype == IN_CHIPS && mode == COL_PLAIN) {
        InChips in{reinterpret_cast<const unsigned char*>(inptr), aux};
        return launch_fwd<T, COL_PLAIN>(in, a, nblk, s);
    } else if (intype == IN_C64S) {
        const SplitPtr* sp = reinterpret_cast<const SplitPtr*>(inptr);
        InCplxSplit in{sp->re, sp->im, aux};
        if (mode == COL_MIX) return launch_fwd<T, COL_MIX>(in, a, nblk, s);
        if (mode == COL_SQUARE) return launch_fwd<T, COL_SQUARE>(in, a, nblk, s);
    } else if (intype == IN_I16RS && mode == COL_MIX) {
        const ResamplePtr* rp = reinterpret_cast<const ResamplePtr*>(inptr);
        InI16Resample in{reinterpret_cast<const short2*>(rp->iq), aux, rp->t0, rp->edge, rp->c};
        return launch_fwd<T, COL_MIX>(in, a, nblk, s);
    } else if (intype == IN_C32) {
        InCplx<float> in{reinterpret_cast<const cpx<float>*>(inptr)};
        if (mode == COL_MIX) return launch_fwd<T, COL_MIX>(in, a, nblk, s);
        if (mode == COL_PLAIN) return launch_fwd<T, COL_PLAIN>(in, a, nblk, s);
    } else if (intype == IN_C64) {
        InCplx<double> in{reinterpret_cast<const cpx<double>*>(inptr)};
        if (mode == COL_PLAIN) return launch_fwd<T, COL_PLAIN>(in, a, nblk, s);
    }
    return -1;
}
template <typename T> int inv(const void* args, unsigned nblk, hipStream_t s) {
    const ColInvArgs<T>& a = *reinterpret_cast<const ColInvArgs<T>*>(args);
    if constexpr (PR::S == 2 && std::is_same<T, float>::value && NT >= 384) {
        static const bool split = [] { const char* e = getenv("TWX_COLINV3"); return !e || atoi(e) != 0; }();
        if (split) {                         // component-wise exchange: three or four workgroups per CU (twx_kernels.h)
            if (a.norm1) TWX_LAUNCH((k_col_inv3<PR, T, W, NT, 1>), dim3(nblk), dim3(NT), s, a);
            else TWX_LAUNCH((k_col_inv3<PR, T, W, NT, 0>), dim3(nblk), dim3(NT), s, a);
            return (int)hipGetLastError();
        }
    }
    if (a.norm1) TWX_LAUNCH((k_col_inv<PR, T, W, NT, 1>), dim3(nblk), dim3(NT), s, a);
    else TWX_LAUNCH((k_col_inv<PR, T, W, NT, 0>), dim3(nblk), dim3(NT), s, a);
    return (int)hipGetLastError();
}

struct Reg {
    Reg() {
        register_col(ColOps{P::L, W, NT, 0, &fwd<float>, &inv<float>});
#ifndef TWX_NO_F64
        register_col(ColOps{P::L, W, NT, 1, &fwd<double>, &inv<double>});
#endif
    }
} reg_instance;
}  // namespace
}  // namespace twx
