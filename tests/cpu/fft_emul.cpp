// CPU emulation of the LDS Stockham stages in csrc/twx_fft.h: every "thread" of a workgroup is
// run in a loop, phase by phase (a phase boundary = a workgroup barrier on the device).
// Compares against a naive O(L^2) DFT in double.  Built and run by tests/test_fft_emul.py.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../amaranth_twstft_amd/csrc/twx_fft.h"

using namespace twx;

template <typename T> static double tol();
template <> double tol<float>() { return 2e-6; }
template <> double tol<double>() { return 1e-13; }

template <class P, typename T, bool INV, int W, int PADQ, int s>
struct RunStages {
    using TL = Tile<P, T, INV, W, PADQ>;
    static void run(std::vector<cpx<T>>& lds, const std::vector<cpx<T>>& tw, std::vector<cpx<T>>& out) {
        constexpr int R = P::radix(s);
        const int ntask = TL::template tasks<s>();
        std::vector<std::vector<cpx<T>>> regs(ntask, std::vector<cpx<T>>(R));
        // phase A: all threads read (+twiddle) + butterfly
        for (int t = 0; t < ntask; ++t) {
            int j = t / W, c = t % W;
            TL::template load_lds<s>(lds.data(), tw.data(), j, c, regs[t].data());
            TL::template bfly<s>(regs[t].data());
        }
        // barrier, phase B: all threads write
        if constexpr (s == P::S - 1) {
            for (int t = 0; t < ntask; ++t) {
                int j = t / W, c = t % W;
                for (int q = 0; q < R; ++q) out[TL::template out_pos<s>(j, q) * W + c] = regs[t][q];
            }
        } else {
            for (int t = 0; t < ntask; ++t) {
                int j = t / W, c = t % W;
                TL::template store_lds<s>(lds.data(), j, c, regs[t].data());
            }
            RunStages<P, T, INV, W, PADQ, (s + 1 < P::S ? s + 1 : s)>::run(lds, tw, out);
        }
    }
};

template <class P, typename T, bool INV, int W, int PADQ> static int check(const char* name) {
    using TL = Tile<P, T, INV, W, PADQ>;
    constexpr int L = P::L;
    std::vector<cpx<T>> x(L * W), out(L * W), lds(TL::lds_elems), tw(L);
    for (int m = 0; m < L; ++m) {
        double a = -2.0 * M_PI * m / L;
        tw[m] = mk<T>(T(std::cos(a)), T(std::sin(a)));
    }
    srand(1234);
    for (auto& v : x) v = mk<T>(T(rand() % 2001 - 1000) / 100, T(rand() % 2001 - 1000) / 100);
    // stage 0: "global" load straight into registers
    {
        constexpr int R = P::radix(0);
        const int ntask = TL::template tasks<0>();
        std::vector<std::vector<cpx<T>>> regs(ntask, std::vector<cpx<T>>(R));
        for (int t = 0; t < ntask; ++t) {
            int j = t / W, c = t % W;
            for (int r = 0; r < R; ++r) regs[t][r] = x[TL::template in_pos<0>(j, r) * W + c];
            TL::template bfly<0>(regs[t].data());
        }
        if (P::S == 1) {
            for (int t = 0; t < ntask; ++t) {
                int j = t / W, c = t % W;
                for (int q = 0; q < R; ++q) out[TL::template out_pos<0>(j, q) * W + c] = regs[t][q];
            }
        } else {
            for (int t = 0; t < ntask; ++t) TL::template store_lds<0>(lds.data(), t / W, t % W, regs[t].data());
            RunStages<P, T, INV, W, PADQ, (P::S > 1 ? 1 : 0)>::run(lds, tw, out);
        }
    }
    // naive DFT
    double maxerr = 0, maxref = 0;
    for (int c = 0; c < W; ++c)
        for (int k = 0; k < L; k += (L > 2000 ? 37 : 1)) {
            std::complex<double> acc = 0;
            for (int n = 0; n < L; ++n) {
                double a = (INV ? 2.0 : -2.0) * M_PI * double((long long)n * k % L) / L;
                acc += std::complex<double>(x[n * W + c].x, x[n * W + c].y) * std::complex<double>(std::cos(a), std::sin(a));
            }
            std::complex<double> got(out[k * W + c].x, out[k * W + c].y);
            maxerr = std::fmax(maxerr, std::abs(got - acc));
            maxref = std::fmax(maxref, std::abs(acc));
        }
    double rel = maxerr / maxref;
    bool ok = rel < tol<T>();
    printf("%-28s L=%5d W=%2d INV=%d %s rel=%.2e %s\n", name, L, W, int(INV), sizeof(T) == 4 ? "f32" : "f64", rel, ok ? "ok" : "FAIL");
    return ok ? 0 : 1;
}

#define CK(PLAN, W, PADQ)                                                             \
    fails += check<PLAN, float, false, W, PADQ>(#PLAN) + check<PLAN, float, true, W, PADQ>(#PLAN) + \
             check<PLAN, double, false, W, PADQ>(#PLAN) + check<PLAN, double, true, W, PADQ>(#PLAN)

int main() {
    int fails = 0;
    using P2 = Plan<2, 2>; using P3 = Plan<3, 3>; using P4 = Plan<4, 4>; using P5 = Plan<5, 5>;
    using P8 = Plan<8, 8>; using P10 = Plan<10, 10>; using P16 = Plan<16, 16>; using P20 = Plan<20, 20>;
    using P25 = Plan<25, 25>; using P6 = Plan<6, 6>; using P12 = Plan<12, 12>; using P15 = Plan<15, 15>;
    CK(P2, 1, 0); CK(P3, 1, 0); CK(P4, 1, 0); CK(P5, 2, 0); CK(P8, 1, 0); CK(P10, 1, 0); CK(P16, 1, 0);
    CK(P20, 1, 0); CK(P25, 4, 0); CK(P6, 1, 0); CK(P12, 1, 0); CK(P15, 1, 0);
    using P7 = Plan<7, 7>; using P14 = Plan<14, 14>; using P21 = Plan<21, 21>; using P490 = Plan<490, 7, 14, 5>;
    using P8750 = Plan<8750, 14, 25, 25>; using P7000 = Plan<7000, 14, 20, 25>;
    CK(P7, 1, 0); CK(P14, 2, 0); CK(P21, 1, 0); CK(P490, 2, 0); CK(P8750, 1, 20); CK(P7000, 1, 0);
    using P50 = Plan<50, 5, 10>; using P100 = Plan<100, 10, 10>; using P200 = Plan<200, 10, 20>;
    using P625 = Plan<625, 25, 25>; using P500 = Plan<500, 20, 25>; using P1000 = Plan<1000, 10, 10, 10>;
    using P8000 = Plan<8000, 20, 20, 20>; using P4000 = Plan<4000, 10, 20, 20>; using P400 = Plan<400, 20, 20>;
    using P10000 = Plan<10000, 10, 10, 10, 10>; using P240 = Plan<240, 3, 4, 20>;
    CK(P50, 4, 0); CK(P100, 8, 0); CK(P200, 1, 20); CK(P625, 16, 0); CK(P500, 8, 0); CK(P1000, 2, 0);
    CK(P400, 1, 20); CK(P4000, 1, 20); CK(P8000, 1, 20); CK(P10000, 1, 16); CK(P240, 2, 7);
    printf("%s\n", fails ? "FAILED" : "ALL OK");
    return fails ? 1 : 0;
}
