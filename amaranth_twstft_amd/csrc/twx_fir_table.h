// Host side of k_fir_mfma (csrc/twx_aux.hip): geometry and the table of A fragments — plain C++ (with _Float16: clang), so that
// tests/cpu/fir_table_emul.cpp can run the kernel's arithmetic on the CPU against the direct sum.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>
#ifndef TWX_FM_HD
#if defined(__HIPCC__)
#define TWX_FM_HD __host__ __device__
#else
#define TWX_FM_HD
#endif
#endif
namespace twx_fm {
constexpr int FM_NT = 512, FM_OUT = 256, FM_MAXNPW = 6, FM_SP = 68, FM_NV = 3;      // (7 and 8 pairs per wave spill at 128 registers)
TWX_FM_HD inline int fm_phys(int q) { return q + 4 * (q >> 6); }
struct FirMfmaGeom { int A, KS, NPW, NQ, PS; size_t lds; bool ok; };
inline FirMfmaGeom fir_mfma_geom(int ntaps, int dec) {
    FirMfmaGeom g;
    g.A = (ntaps + dec - 1) / dec;
    g.KS = (g.A + 15 + 15) / 16;                        // Toeplitz columns A + 15 in steps of 16
    g.NPW = (dec * g.KS + 7) / 8;                       // (phase, step) pairs per wave
    g.NQ = FM_OUT + 16 * g.KS;                          // samples staged per phase
    int ps = fm_phys(g.NQ) + 4;
    ps = (ps + 3) & ~3;
    while ((ps & 7) != 4) ps += 4;                      // rows 16-byte aligned, PS / 4 odd: the staging writes of consecutive phases spread over the banks
    g.PS = ps;
    g.lds = (size_t)2 * dec * g.PS * 4 + (size_t)8 * 8 * FM_SP * 4;          // two spans + the partial sums [wave][8][FM_SP]
    g.ok = g.NPW >= 1 && g.NPW <= FM_MAXNPW && g.lds <= 80 * 1024 && (long long)g.NQ * dec <= 4ll * FM_NV * FM_NT;      // (the span in FM_NV vectors per thread)
    return g;
}
// table of A fragments [pair u][piece][lane] x 16 bytes, u = p * KS + ks padded to 8 * NPW pairs (zeros); returns 2^-s through inv_scale
inline std::vector<float> fir_mfma_table(const float* taps, int ntaps, int dec, const FirMfmaGeom& g, float* inv_scale) {
    double hmax = 0;
    for (int j = 0; j < ntaps; ++j) hmax = std::max(hmax, (double)fabsf(taps[j]));
    int s = 0;
    if (hmax > 0) { s = (int)floor(log2(127.0 / hmax)); s = std::max(-100, std::min(100, s)); }
    *inv_scale = (float)ldexp(1.0, -s);
    std::vector<_Float16> h1((size_t)ntaps), h2((size_t)ntaps);
    for (int j = 0; j < ntaps; ++j) {
        const double hs = ldexp((double)taps[j], s);
        h1[(size_t)j] = (_Float16)hs;
        h2[(size_t)j] = (_Float16)(hs - (double)h1[(size_t)j]);
    }
    const int npairs = 8 * g.NPW;
    std::vector<float> out((size_t)npairs * 2 * 64 * 4, 0.f);
    unsigned short* o = reinterpret_cast<unsigned short*>(out.data());
    for (int u = 0; u < dec * g.KS; ++u) {
        const int p = u / g.KS, ks = u % g.KS;
        for (int pc = 0; pc < 2; ++pc)
            for (int l = 0; l < 64; ++l) {
                const int i = l & 15, gq = l >> 4;
                for (int e = 0; e < 8; ++e) {
                    const int a = 16 * ks + 4 * gq + (e >> 1) - i;
                    const long long j = (long long)a * dec + p;
                    _Float16 v = (_Float16)0.0f;
                    if (a >= 0 && j < ntaps) {
                        const _Float16 h = pc ? h2[(size_t)j] : h1[(size_t)j];
                        v = (e & 1) ? h : (_Float16)((float)h * 256.0f);              // (xh, xl) pairs meet (256 h, h): exact power-of-two scaling
                    }
                    unsigned short bits; memcpy(&bits, &v, 2);
                    o[(((size_t)u * 2 + pc) * 64 + l) * 8 + e] = bits;
                }
            }
    }
    return out;
}
}  // namespace twx_fm
