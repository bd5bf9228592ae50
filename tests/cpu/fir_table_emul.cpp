// CPU emulation of k_fir_mfma's arithmetic from its host-built table (csrc/twx_fir_table.h): the Toeplitz product with (xh, xl) sample pairs
// against (256 h, h) tap pairs, fp16 operands, float accumulation, against the fp64 direct sum  y[m] = sum_j taps[j] x[m dec + j].
// Built with clang++ (_Float16) by tests/test_wideband_host.py.   usage: fir_table_emul ntaps dec seed
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "twx_fir_table.h"
using namespace twx_fm;

int main(int argc, char** argv) {
    const int ntaps = argc > 1 ? atoi(argv[1]) : 421, dec = argc > 2 ? atoi(argv[2]) : 14, seed = argc > 3 ? atoi(argv[3]) : 1;
    const FirMfmaGeom g = fir_mfma_geom(ntaps, dec);
    if (!g.ok) { printf("geometry does not fit (NPW %d)\n", g.NPW); return 3; }
    if ((g.PS & 7) != 4 || g.PS < fm_phys(g.NQ - 1) + 1 || g.lds > 80 * 1024) { printf("bad geometry: PS %d NQ %d lds %zu\n", g.PS, g.NQ, g.lds); return 2; }
    std::mt19937 rng((unsigned)seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-3.f, 0.f);
    std::vector<float> taps((size_t)ntaps);
    for (int j = 0; j < ntaps; ++j) taps[(size_t)j] = nd(rng) / sqrtf((float)ntaps) * powf(10.f, ud(rng));
    const int nout = FM_OUT + 37;                                       // one full trip and a ragged one
    const long long nin = (long long)(nout - 1) * dec + ntaps;
    std::vector<int16_t> xi((size_t)nin), xq((size_t)nin);
    for (long long e = 0; e < nin; ++e) {
        xi[(size_t)e] = (int16_t)std::max(-32768.f, std::min(32767.f, nd(rng) * 9000.f));
        xq[(size_t)e] = (int16_t)std::max(-32768.f, std::min(32767.f, nd(rng) * 9000.f));
    }
    xi[3] = -32768; xq[3] = 32767; xi[4] = -1; xq[4] = 255; xi[5] = 256; xq[5] = -256;
    float inv_scale = 0.f;
    const std::vector<float> tab = fir_mfma_table(taps.data(), ntaps, dec, g, &inv_scale);
    if (tab.size() != (size_t)8 * g.NPW * 2 * 64 * 4) { printf("table size\n"); return 2; }
    const _Float16* th = reinterpret_cast<const _Float16*>(tab.data());
    auto sample = [&](const std::vector<int16_t>& x, long long e) -> int { return e < nin ? (int)x[(size_t)e] : 0; };
    double maxref = 0, maxerr = 0;
    std::vector<double> refI((size_t)nout), refQ((size_t)nout);
    for (int m = 0; m < nout; ++m) {
        double sI = 0, sQ = 0;
        for (int j = 0; j < ntaps; ++j) { sI += (double)taps[(size_t)j] * xi[(size_t)((long long)m * dec + j)]; sQ += (double)taps[(size_t)j] * xq[(size_t)((long long)m * dec + j)]; }
        refI[(size_t)m] = sI; refQ[(size_t)m] = sQ;
        maxref = std::max(maxref, std::max(fabs(sI), fabs(sQ)));
    }
    const int ntrips = (nout + FM_OUT - 1) / FM_OUT;
    for (int trip = 0; trip < ntrips; ++trip) {
        const long long m0 = (long long)trip * FM_OUT;
        for (int n = 0; n < 16; ++n)
            for (int i = 0; i < 16; ++i) {
                const long long m = m0 + 16 * n + i;
                if (m >= nout) continue;
                for (int c = 0; c < 2; ++c) {
                    const std::vector<int16_t>& x = c ? xq : xi;
                    float acc8[8];
                    for (int w = 0; w < 8; ++w) {                               // the eight waves' partial sums, then their sum in wave order
                        float acc = 0.f;
                        for (int jp = 0; jp < g.NPW; ++jp) {
                            const int u = w + 8 * jp;
                            if (u >= dec * g.KS) continue;
                            const int p = u / g.KS, ks = u % g.KS;
                            for (int pc = 0; pc < 2; ++pc)
                                for (int gq = 0; gq < 4; ++gq)
                                    for (int e = 0; e < 8; ++e) {
                                        const _Float16 a = th[(((size_t)u * 2 + pc) * 64 + (i + 16 * gq)) * 8 + e];      // A[row i][k = 8 gq + e]
                                        const long long q = m0 + 16 * n + 16 * ks + 4 * gq + (e >> 1);                     // B[k][col n]: sample group of phase p
                                        const int v = sample(x, q * dec + p);
                                        const _Float16 b = (e & 1) ? (_Float16)(float)(v & 255) : (_Float16)(float)(v >> 8);  // (xh, xl) pairs
                                        acc += (float)a * (float)b;                                                         // exact product, float accumulation
                                    }
                        }
                        acc8[w] = acc;
                    }
                    float s = 0.f;
                    for (int w = 0; w < 8; ++w) s += acc8[w];
                    s *= inv_scale;
                    const double ref = c ? refQ[(size_t)m] : refI[(size_t)m];
                    maxerr = std::max(maxerr, fabs((double)s - ref));
                }
            }
    }
    const double gate = 2e-6 * maxref + 1e-3;
    printf("ntaps %d dec %d: A %d KS %d NPW %d PS %d lds %zu  inv_scale %g  max |err| %.3g  gate %.3g  max |ref| %.6g\n", ntaps, dec, g.A, g.KS, g.NPW, g.PS, g.lds,
           (double)inv_scale, maxerr, gate, maxref);
    return maxerr <= gate ? 0 : 1;
}
