// Instantiates the row-pass kernels of ONE plan: compile with
//   -DTWX_PLAN='Plan<8000,20,20,20>' -DTWX_NT=448 -DTWX_PADQ=20
#include <type_traits>
#include "twx_kernels.h"
#include "twx_plans.h"

namespace twx {
namespace {
using P = TWX_PLAN;
constexpr int NT = TWX_NT;
constexpr int PADQ = TWX_PADQ;
static_assert(P::max_tasks <= NT, "one task per thread per stage");

template <class PP> struct HasRowD {
    static constexpr bool value = (PP::S == 2 && PP::radix(0) == PP::radix(1) && 64 / PP::radix(1) >= 1) ||
                                  (PP::S == 3 && PP::radix(1) == PP::radix(2));
};

template <typename T> int run(int mode, const void* args, unsigned nblk, hipStream_t s) {
    const RowArgs<T>& a = *reinterpret_cast<const RowArgs<T>*>(args);
    if (mode == ROW_STORE) TWX_LAUNCH((k_row<P, T, ROW_STORE, PADQ, NT>), dim3(nblk), dim3(NT), s, a);
    else if (mode == ROW_BAND) TWX_LAUNCH((k_row<P, T, ROW_BAND, PADQ, NT>), dim3(nblk), dim3(NT), s, a);
    else if (mode == ROW_MID) {
        // complex double through the Stockham form needs more registers than a lane has where the row is long (8000: 138 spilled, 556 B of
        // scratch per lane in a pass that is bound by the same HBM): plans with the DIF/DIT form run that one, always (twx_api.hip ignores
        // TWX_ROWD=0 for them), and the spilling instantiation does not exist
        if constexpr (std::is_same<T, double>::value && HasRowD<P>::value) return -1;
        else TWX_LAUNCH((k_row<P, T, ROW_MID, PADQ, NT>), dim3(nblk), dim3(NT), s, a);
    } else return -1;
    return (int)hipGetLastError();
}


template <typename T> int caf(const void* args, unsigned nblk, hipStream_t s) {
    const CafArgs<T>& a = *reinterpret_cast<const CafArgs<T>*>(args);
    if constexpr (HasRowD<P>::value && P::S == 3) {
        if (a.Yperm) {                                     // DIF/DIT form: the caller passes nblk = N1 * ceil(nbins / bpw)
            constexpr int NTD = RowD<P, T>::NT_MIN;
            if (a.lds_pad > 0) hipLaunchKernelGGL((k_rowd_caf<P, T, NTD>), dim3(nblk), dim3(NTD), (size_t)a.lds_pad, s, a);
            else TWX_LAUNCH((k_rowd_caf<P, T, NTD>), dim3(nblk), dim3(NTD), s, a);
            return (int)hipGetLastError();
        }
    }
    if (a.Yperm) return -1;
    TWX_LAUNCH((k_row_caf<P, T, PADQ, NT>), dim3(nblk), dim3(NT), s, a);
    return (int)hipGetLastError();
}


template <typename T> int rowd(int mode, const void* args, unsigned nblk, hipStream_t s) {
    if constexpr (HasRowD<P>::value) {
        constexpr int NTD = RowD<P, T>::NT_MIN;
        RowDArgs<T> a = *reinterpret_cast<const RowDArgs<T>*>(args);
        a.total_rows = nblk;
        if constexpr (RowD<P, T>::R0 == 1) {     // short rows: several rows per workgroup
            // complex double: four waves, one per SIMD, so that a lane may use 512 registers (v, pr, csr and the row loads in flight are
            // 79 complex doubles: at 448 threads the 256-register cap spilled 24 of them)
            constexpr int NTS = std::is_same<T, float>::value ? 448 : 256, G = (NTS / 64) * RowD<P, T>::BPW;
            const unsigned grid = (nblk + G - 1) / G;
            if (mode == ROW_BAND) TWX_LAUNCH((k_rowd_small<P, T, ROW_BAND, NTS>), dim3(grid), dim3(NTS), s, a, nblk);
            else if (mode == ROW_MID) TWX_LAUNCH((k_rowd_small<P, T, ROW_MID, NTS>), dim3(grid), dim3(NTS), s, a, nblk);
            else return -1;
        } else {
        if (mode == ROW_BAND) {
            // a narrow search band in fp32: the few bins a row can contribute are summed over the workgroup's threads, the row never enters LDS
            if constexpr (std::is_same<T, float>::value && rowd_bandsum_ok<P>()) {
                if (a.nprune > 0 && a.wm) {
                    constexpr int NTB = BandsumDeal<RowD<P, T>::M / 16, TWX_BANDSUM_TPT>::threads;
                    TWX_LAUNCH((k_rowd_bandsum<P, T, TWX_BANDSUM_TPT>), dim3(nblk), dim3(NTB), s, a);
                    return (int)hipGetLastError();
                }
            }
            TWX_LAUNCH((k_rowd<P, T, ROW_BAND, NTD>), dim3(nblk), dim3(NTD), s, a);
        } else if (mode == ROW_MID) {
            unsigned grid = nblk;                          // resident workgroups walk the rows (a.total_rows = nblk)
            if (rowd_mid_resident<P, T>() && a.r.pf_stride > 0 && (unsigned)a.r.pf_stride < nblk) grid = (unsigned)a.r.pf_stride;
            if (a.chk_rows) TWX_LAUNCH((k_rowd<P, T, ROW_MID, NTD, true>), dim3(grid), dim3(NTD), s, a);      // TWX_OPT_SELFCHECK: Parseval per row
            else TWX_LAUNCH((k_rowd<P, T, ROW_MID, NTD>), dim3(grid), dim3(NTD), s, a);
        } else return -1;
        }
        return (int)hipGetLastError();
    } else {
        return -1;
    }
}

struct Reg {
    Reg() {
        register_row(RowOps{P::L, NT, 0, P::S, {P::radix(0), P::radix(1), P::radix(2), P::radix(3)}, &run<float>, &caf<float>, HasRowD<P>::value ? &rowd<float> : nullptr});
#ifndef TWX_NO_F64
        register_row(RowOps{P::L, NT, 1, P::S, {P::radix(0), P::radix(1), P::radix(2), P::radix(3)}, &run<double>, &caf<double>, HasRowD<P>::value ? &rowd<double> : nullptr});
#endif
    }
} reg_instance;
}  // namespace
}  // namespace twx
