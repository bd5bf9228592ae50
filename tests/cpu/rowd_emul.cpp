// CPU emulation of the DIF/DIT row transform (RowD in csrc/twx_fft.h): forward vs naive DFT (bin
// order k_of), inverse(forward(x)) == L*x.  Phases separated like the device code (workgroup barrier /
// wave-local exchange); within a phase threads run in arbitrary (sequential) order.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../amaranth_twstft_amd/csrc/twx_fft.h"
using namespace twx;

template <class P, typename T> static int check(const char* name, double tol) {
    using D = RowD<P, T>;
    constexpr int L = D::L, R = D::R, R0 = D::R0, M = D::M;
    const int NT = D::NT_MIN;
    std::vector<cpx<T>> tabs(D::tab_total), lds(D::lds_elems), x(L), X(L), z(L);
    auto W = [](long long num, long long den) { double a = -2.0 * M_PI * double(num % den) / double(den); return mk<T>(T(std::cos(a)), T(std::sin(a))); };
    for (int q = 0; q < R; ++q) for (int xx = 0; xx < R; ++xx) tabs[D::tab_a + q * R + xx] = W((long long)xx * q, R * R);
    for (int q = 0; q < R0; ++q) for (int xx = 0; xx < R; ++xx) { tabs[D::tab_b + q * R + xx] = W((long long)xx * q, L); tabs[D::tab_c + q * R + xx] = W((long long)xx * q, L / R); }
    srand(7);
    for (auto& v : x) v = mk<T>(T(rand() % 2001 - 1000) / 100, T(rand() % 2001 - 1000) / 100);
    std::vector<std::vector<cpx<T>>> reg(NT, std::vector<cpx<T>>(R > R0 ? R : R0));
    // forward stage 0
    for (int t = 0; t < M; ++t) {
        for (int r = 0; r < R0; ++r) reg[t][r] = x[t + M * r];
        Bfly<T, R0, false>::run(reg[t].data());
        D::f0_twiddle_store(lds.data(), tabs.data(), t, reg[t].data());
    }
    // barrier; stage 1 (block-local), stage 2
    for (int tid = 0; tid < NT; ++tid) { int q0, i; if (D::blk_map(tid, q0, i)) D::f1(lds.data(), tabs.data(), q0, i, reg[tid].data()); }
    for (int tid = 0; tid < NT; ++tid) { int q0, i; if (D::blk_map(tid, q0, i)) D::f2(lds.data(), q0, i, reg[tid].data()); }
    double maxerr = 0, maxref = 0;
    for (int tid = 0; tid < NT; ++tid) {
        int q0, q1; if (!D::blk_map(tid, q0, q1)) continue;
        for (int q2 = 0; q2 < R; ++q2) {
            const int k = D::k_of(q0, q1, q2);
            X[k] = reg[tid][q2];
            if (k % (L > 1000 ? 53 : 1)) continue;
            std::complex<double> acc = 0;
            for (int n = 0; n < L; ++n) { double a = -2.0 * M_PI * double((long long)n * k % L) / L; acc += std::complex<double>(x[n].x, x[n].y) * std::complex<double>(std::cos(a), std::sin(a)); }
            maxerr = std::fmax(maxerr, std::abs(std::complex<double>(X[k].x, X[k].y) - acc)); maxref = std::fmax(maxref, std::abs(acc));
        }
    }
    const double fe = maxerr / maxref;
    // inverse: stage A from the same registers, B, barrier, C
    for (int tid = 0; tid < NT; ++tid) { int q0, q1; if (D::blk_map(tid, q0, q1)) D::iA(lds.data(), tabs.data(), q0, q1, reg[tid].data()); }
    for (int tid = 0; tid < NT; ++tid) { int q0, a; if (D::blk_map(tid, q0, a)) D::iB(lds.data(), tabs.data(), q0, a, reg[tid].data()); }
    double ie = 0, xm = 0;
    for (int t = 0; t < M; ++t) {
        D::iC(lds.data(), t, reg[t].data());
        for (int c = 0; c < R0; ++c) {
            const int n = t + M * c;
            ie = std::fmax(ie, std::hypot(double(reg[t][c].x) / L - x[n].x, double(reg[t][c].y) / L - x[n].y));
            xm = std::fmax(xm, std::hypot(double(x[n].x), double(x[n].y)));
        }
    }
    const bool ok = fe < tol && ie / xm < tol;
    printf("%-22s L=%5d R0=%2d R=%2d %s fwd rel=%.2e inv rel=%.2e %s\n", name, L, R0, R, sizeof(T) == 4 ? "f32" : "f64", fe, ie / xm, ok ? "ok" : "FAIL");
    return ok ? 0 : 1;
}

int main() {
    int fails = 0;
    using P8000 = Plan<8000, 20, 20, 20>; using P4000 = Plan<4000, 10, 20, 20>; using P400 = Plan<400, 20, 20>;
    using P2000 = Plan<2000, 5, 20, 20>; using P100 = Plan<100, 10, 10>; using P500 = Plan<500, 5, 10, 10>;
    fails += check<P8000, float>("P8000", 2e-6) + check<P8000, double>("P8000", 1e-13);
    fails += check<P4000, float>("P4000", 2e-6) + check<P4000, double>("P4000", 1e-13);
    fails += check<P400, float>("P400", 2e-6) + check<P400, double>("P400", 1e-13);
    fails += check<P2000, float>("P2000", 2e-6) + check<P100, double>("P100", 1e-13) + check<P500, float>("P500", 2e-6);
    printf("%s\n", fails ? "FAILED" : "ALL OK");
    return fails ? 1 : 0;
}
