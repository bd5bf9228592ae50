// twx_aux.hip — kernels beside the FFT chain: the direct sliding dot-product correlator for short
// codes (tracking stage of experiments/231001_DLL_PLL/rxcomplex.cpp:593-614) and the FIR
// decimating front end (no reference twin; parameters from experiments/2403/zmq_rx.py:208-215).
// Both are bounded by the fp32 vector rate, not by HBM (a12 ~50 flop/B; the 577-tap
// decimator 41 flop/B against a ridge of ~20), so they are laid out for the VALU: every LDS read
// feeds several FMAs, taps/replica reads are conflict-free, all 64 lanes of every wave work.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "twx_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Sliding dot product.  For code period p and lag index li (lag = li - nlag):
//   out[p][li] = (scale/nobs) * sum_i x[pt + p*nobs + i] * exp(-2 pi j (ff*(p*nobs+i) + phi)) * w[(i - lag) mod nobs]
// = downconv_trk (rxcomplex.cpp:1051-1061) + the cblas_dgemm against the PRN_mapping replicas
// (:605, :989-999), fused: the replica matrix is never materialised.
// One LANE per SAMPLE, all 2*NLAG+1 lag accumulators of that lane in registers: the mixed sample stays in
// registers, each replica value read from LDS (consecutive lanes, consecutive addresses) feeds two FMAs, and
// every lane of every wave is busy whatever the lag count.  Samples are streamed once (4 B/sample).
// grid = (chunks of SD_CH samples, ncodes), block = 256
// ---------------------------------------------------------------------------------------------
constexpr int SD_NT = 256, SD_S = 64, SD_CH = SD_NT * SD_S;      // 16384 samples per workgroup

template <int NLAG>
__global__ __launch_bounds__(SD_NT) void k_sliding_dot(const short2* __restrict__ x, int nch, long long pt, long long nobs, int nlag,
                                                       const float* __restrict__ w, double ff, double phi, float scale,
                                                       double* __restrict__ partial /*[ncodes][chunks][2*nlag+1][2]*/) {
    constexpr int NL = 2 * NLAG + 1;
    __shared__ float sw[SD_CH + 2 * NLAG + 2];                    // replica segment; reused by the final reduction
    static_assert(NL * SD_NT <= SD_CH + 2 * NLAG + 2, "reduction buffer must fit the replica segment");
    const int p = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const long long s0 = (long long)chunk * SD_CH;
    const int cnt = (int)min((long long)SD_CH, nobs - s0);
    const int tid = threadIdx.x;
    // entry u <-> w[(s0 - NLAG + u) mod nobs]
    for (int u = tid; u < cnt + 2 * NLAG; u += SD_NT) {
        long long k = (s0 - NLAG + u) % nobs; if (k < 0) k += nobs;
        sw[u] = w[k];
    }
    __syncthreads();
    float ar[NL], ai[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) { ar[l] = 0.f; ai[l] = 0.f; }
    for (int t = tid; t < cnt; t += SD_NT) {
        const long long i = (long long)p * nobs + s0 + t;         // the NCO runs over the whole block of codes
        const short2 s = x[(pt + i) * nch];
        double ph = ff * (double)i + phi;                          // fp64 phase reduction, fp32 sincos
        ph -= rint(ph);
        float sn, cs;
        sincospif(-2.0f * (float)ph, &sn, &cs);
        const float re = (float)s.x, im = (float)s.y;
        const float yr = scale * (re * cs - im * sn), yi = scale * (re * sn + im * cs);
        const float* c = sw + t + 2 * NLAG;                        // lag index l: replica index (s0+t) - (l - NLAG)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float cv = c[-l];
            ar[l] = fmaf(yr, cv, ar[l]);
            ai[l] = fmaf(yi, cv, ai[l]);
        }
    }
    // block reduction through LDS, one component at a time: buf[l][tid]; thread r = (l, quarter) sums 64 lanes,
    // starting at a lane-dependent rotation so that the 64 lanes of a wave hit 64 different banks
    float* buf = sw;
    const int r = tid, rl = r >> 2, rq = r & 3;
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < NL; ++l) buf[l * SD_NT + tid] = comp ? ai[l] : ar[l];
        __syncthreads();
        if (rl < NL) {
            float a = 0.f;
            const float* q = buf + rl * SD_NT + rq * 64;
            for (int j = 0; j < 64; ++j) a += q[(j + r) & 63];
            part[comp] = a;
        }
    }
    __syncthreads();
    float2* red = reinterpret_cast<float2*>(sw);
    if (rl < NL) red[r] = make_float2(part[0], part[1]);
    __syncthreads();
    if (tid < NL) {
        const int lag = tid - NLAG;
        if (lag >= -nlag && lag <= nlag) {
            double sr = 0, si = 0;
            for (int k = 0; k < 4; ++k) { sr += (double)red[tid * 4 + k].x; si += (double)red[tid * 4 + k].y; }
            double* o = partial + (((long long)p * nchunks + chunk) * (2 * nlag + 1) + (lag + nlag)) * 2;
            o[0] = sr; o[1] = si;
        }
    }
}

__global__ void k_sliding_reduce(const double* __restrict__ partial, int nchunks, int nl, double inv_nobs, double* __restrict__ out) {
    const int p = blockIdx.x, li = threadIdx.x;
    if (li >= nl) return;
    double sr = 0, si = 0;
    for (int c = 0; c < nchunks; ++c) {   // fixed order: bit-reproducible
        const double* q = partial + (((long long)p * nchunks + c) * nl + li) * 2;
        sr += q[0]; si += q[1];
    }
    out[((long long)p * nl + li) * 2] = sr * inv_nobs;
    out[((long long)p * nl + li) * 2 + 1] = si * inv_nobs;
}

int launch_sliding(hipStream_t st, const short2* dx, int nch, long long pt, long long nobs, int ncodes, int nlag, const float* dw,
                   double ff, double phi, double scale, double* dpart, double* dout) {
    const int nchunks = (int)((nobs + SD_CH - 1) / SD_CH);
    const dim3 grid(nchunks, ncodes), block(SD_NT);
#define SD_GO(NL_) hipLaunchKernelGGL((k_sliding_dot<NL_>), grid, block, 0, st, dx, nch, pt, nobs, nlag, dw, ff, phi, (float)scale, dpart)
    if (nlag <= 4) SD_GO(4); else if (nlag <= 8) SD_GO(8); else if (nlag <= 16) SD_GO(16); else if (nlag <= 28) SD_GO(28); else SD_GO(31);
#undef SD_GO
    if (hipGetLastError() != hipSuccess) return TWX_E_HIP;
    hipLaunchKernelGGL(k_sliding_reduce, dim3(ncodes), dim3(64), 0, st, dpart, nchunks, 2 * nlag + 1, 1.0 / (double)nobs, dout);
    return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
}
size_t sliding_part_bytes(long long nobs, int ncodes, int nlag) {
    return (size_t)ncodes * (size_t)((nobs + SD_CH - 1) / SD_CH) * (2 * nlag + 1) * 16;
}

// ---------------------------------------------------------------------------------------------
// FIR low-pass + decimation of interleaved int16 IQ:  y[m] = sum_j taps[j] * x[m*D + j]
// (BASELINE.json configs[4]: 70 Msps wideband capture -> 5 Msps; 577 taps, D = 14: 82 FMA per INPUT sample and
// component, i.e. bounded by the fp32 vector rate, not by the 4.6 B/sample it moves).
// Polyphase form  y[m] = sum_p sum_a h[a*D+p] * x_p[m+a],  x_p[q] = x[q*D+p]:  D stride-1 filters of A = ceil(ntaps/D) taps.
// A workgroup of FIR_NT threads produces FIR_NT*4 consecutive outputs, thread t the four outputs 4t..4t+3:
//   * its input span is converted to float once and staged in LDS phase by phase, "transposed" so that what the
//     64 lanes of a wave read together is contiguous: slot(p, q) = p*PS + (q mod 4)*QS + q div 4  (lane t, step s
//     reads q = 4t+s -> p*PS + (s mod 4)*QS + t + s div 4): conflict-free ds_read_b64;
//   * each value read feeds the four outputs (8 FMAs) with four different taps; the taps are wave-uniform, read with
//     scalar loads from a phase-major table hp[p][3 + a] that carries 3 zeros in front and zeros behind, so the
//     loop has no edge cases;
//   * outputs leave as one 16-B (int16 IQ) or two 16-B (float) stores per lane.
// grid = ceil(nout / (4*FIR_NT)), dynamic LDS = D*PS*8 bytes
// ---------------------------------------------------------------------------------------------
constexpr int FIR_NT = 128, FIR_K = 4, FIR_OUT = FIR_NT * FIR_K;
struct FirGeom { int A, SH, QS, PS, HROW; size_t lds; };
FirGeom fir_geom(int ntaps, int dec) {
    FirGeom g;
    g.A = (ntaps + dec - 1) / dec;
    g.SH = (g.A + FIR_K - 1 + 3) / 4;            // groups of four steps s = 0 .. A+K-2
    g.QS = (FIR_NT + g.SH) | 1;                  // odd: staging writes of one phase spread over the banks
    g.PS = FIR_K * g.QS + 1;
    g.HROW = 4 * g.SH + 4;                       // taps per phase incl. padding (7 are read per group of steps)
    g.lds = (size_t)dec * g.PS * sizeof(float2);
    return g;
}

__global__ __launch_bounds__(FIR_NT) void k_fir_poly(const short2* __restrict__ x, int nch, long long nin, const float* __restrict__ hp,
                                                     int D, int SH, int QS, int PS, int HROW, long long nout,
                                                     short2* __restrict__ y16, float2* __restrict__ yf) {
    extern __shared__ float2 X[];
    const int tid = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * FIR_OUT;
    const long long e0 = m0 * D;
    const int nq = FIR_K * (FIR_NT + SH);                          // q values staged per phase
    const int span = nq * D;
    for (int eb = 0; eb < span; eb += 8 * FIR_NT) {                // eight independent loads in flight per thread
        unsigned raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * FIR_NT + tid;
            const long long g = e0 + e;
            raw[u] = (e < span && g < nin) ? *reinterpret_cast<const unsigned*>(x + g * nch) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * FIR_NT + tid;
            if (e < span) {
                const int q = e / D, p = e - q * D;
                X[p * PS + (q & 3) * QS + (q >> 2)] = make_float2((float)(short)(raw[u] & 0xffffu), (float)(short)(raw[u] >> 16));
            }
        }
    }
    __syncthreads();
    float2 acc[FIR_K];
#pragma unroll
    for (int k = 0; k < FIR_K; ++k) acc[k] = make_float2(0.f, 0.f);
    for (int p = 0; p < D; ++p) {
        const float2* Xp = X + p * PS + tid;
        const float* h = hp + p * HROW;
        for (int sh = 0; sh < SH; ++sh) {
            float t[7];
#pragma unroll
            for (int j = 0; j < 7; ++j) t[j] = h[4 * sh + j];      // wave-uniform: scalar loads
            float2 v[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v[sl] = Xp[sl * QS + sh];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl)
#pragma unroll
                for (int k = 0; k < FIR_K; ++k) {                  // step s = 4sh+sl feeds output k with tap a = s-k (table index a+3)
                    acc[k].x = fmaf(v[sl].x, t[sl - k + 3], acc[k].x);
                    acc[k].y = fmaf(v[sl].y, t[sl - k + 3], acc[k].y);
                }
        }
    }
    const long long m = m0 + (long long)FIR_K * tid;
    if (m + FIR_K <= nout) {
        if (yf) {
            float4* o = reinterpret_cast<float4*>(yf + m);
            o[0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            o[1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        }
        if (y16) {
            unsigned w[FIR_K];
#pragma unroll
            for (int k = 0; k < FIR_K; ++k) {
                const float r = fminf(fmaxf(rintf(acc[k].x), -32768.f), 32767.f), q = fminf(fmaxf(rintf(acc[k].y), -32768.f), 32767.f);
                w[k] = ((unsigned)(unsigned short)(short)r) | ((unsigned)(unsigned short)(short)q << 16);
            }
            *reinterpret_cast<uint4*>(y16 + m) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int k = 0; k < FIR_K && m + k < nout; ++k) {
            if (yf) yf[m + k] = acc[k];
            if (y16) {
                const float r = fminf(fmaxf(rintf(acc[k].x), -32768.f), 32767.f), q = fminf(fmaxf(rintf(acc[k].y), -32768.f), 32767.f);
                y16[m + k] = make_short2((short)r, (short)q);
            }
        }
    }
}

// phase-major tap table: hp[p][3 + a] = taps[a*D + p], zeros elsewhere
std::vector<float> fir_phase_table(const float* taps, int ntaps, int dec, const FirGeom& g) {
    std::vector<float> hp((size_t)dec * g.HROW, 0.f);
    for (int j = 0; j < ntaps; ++j) hp[(size_t)(j % dec) * g.HROW + 3 + j / dec] = taps[j];
    return hp;
}
int launch_fir(hipStream_t st, const short2* dx, int nch, long long nin, const float* hp_dev, int ntaps, int dec, long long nout,
               short2* dy16, float2* dyf) {
    const FirGeom g = fir_geom(ntaps, dec);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fir_poly), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TWX_E_HIP;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((nout + FIR_OUT - 1) / FIR_OUT);
    hipLaunchKernelGGL(k_fir_poly, dim3(grid), dim3(FIR_NT), g.lds, st, dx, nch, nin, hp_dev, dec, g.SH, g.QS, g.PS, g.HROW, nout, dy16, dyf);
    return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
}

bool fir_args_ok(const void* iq, const float* taps, const int64_t* n_out, const void* o16, const void* of, int nch, int ch, int ntaps, int dec, int64_t n_in) {
    return iq && taps && n_out && (o16 || of) && nch >= 1 && ch >= 0 && ch < nch && ntaps >= 1 && ntaps <= 1024 && dec >= 1 && dec <= 16 && n_in >= ntaps;
}
bool sliding_args_ok(const void* iq, const void* rep, const void* out, int nch, int ch, int64_t nobs, int ncodes, int nlag, int64_t pt, int64_t n) {
    return iq && rep && out && nch >= 1 && ch >= 0 && ch < nch && nobs >= 1 && ncodes >= 1 && nlag >= 0 && nlag <= 31 && pt >= 0 && pt + nobs * ncodes <= n;
}

struct DevBuf {          // host-pointer entry points: temporaries of one call
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    bool alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)) == hipSuccess; }
};

}  // namespace

extern "C" {

// ---- device-resident, on the context's stream, context-owned work buffers ------------------------------------
int twx_sliding_dot_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                        int32_t ncodes, int32_t nlag, const float* replica_dev, double ff, double phi, double scale, double* out_dev) {
    if (!ctx) return TWX_E_ARG;
    if (!sliding_args_ok(iq_dev, replica_dev, out_dev, n_channels, channel, nobs, ncodes, nlag, pt, n_samples)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_sliding_dot_dev: bad argument");
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    double* dpart = static_cast<double*>(twx::ctx_scratch(ctx, 0, sliding_part_bytes(nobs, ncodes, nlag)));
    if (!dpart) return TWX_E_NOMEM;
    const int rc = launch_sliding(twx::ctx_stream(ctx), reinterpret_cast<const short2*>(iq_dev) + channel, n_channels, pt, nobs, ncodes, nlag,
                                  replica_dev, ff, phi, scale, dpart, out_dev);
    return rc ? twx::ctx_fail(ctx, rc, "twx_sliding_dot_dev: launch failed") : TWX_OK;
}

int twx_fir_decimate_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                         int32_t dec, void* out_i16_dev, void* out_f32_dev, int64_t* n_out) {
    if (!ctx) return TWX_E_ARG;
    if (!fir_args_ok(iq_dev, taps, n_out, out_i16_dev, out_f32_dev, n_channels, channel, ntaps, dec, n_in)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_fir_decimate_dev: bad argument");
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    const long long nout = (n_in - ntaps) / dec + 1;
    *n_out = nout;
    const FirGeom g = fir_geom(ntaps, dec);
    const std::vector<float> hp = fir_phase_table(taps, ntaps, dec, g);
    // the table lives in a context-owned buffer; the copy is ordered on the context's stream like the kernel
    float* hp_dev = static_cast<float*>(twx::ctx_scratch(ctx, 1, hp.size() * sizeof(float)));
    if (!hp_dev) return TWX_E_NOMEM;
    hipStream_t st = twx::ctx_stream(ctx);
    if (hipMemcpyAsync(hp_dev, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)            // hp is a local: the copy must have left it before we return
        return twx::ctx_fail(ctx, TWX_E_HIP, "twx_fir_decimate_dev: tap upload failed");
    const int rc = launch_fir(st, reinterpret_cast<const short2*>(iq_dev) + channel, n_channels, n_in, hp_dev, ntaps, dec, nout,
                              reinterpret_cast<short2*>(out_i16_dev), reinterpret_cast<float2*>(out_f32_dev));
    return rc ? twx::ctx_fail(ctx, rc, "twx_fir_decimate_dev: launch failed") : TWX_OK;
}

// ---- host-pointer convenience forms (no context): upload, run on the null stream, download -------------------
int twx_sliding_dot(const int16_t* iq, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                    int32_t ncodes, int32_t nlag, const float* replica, double ff, double phi, double scale, double* out) {
    if (!sliding_args_ok(iq, replica, out, n_channels, channel, nobs, ncodes, nlag, pt, n_samples)) return TWX_E_ARG;
    DevBuf dx, dw, dpart, dout;
    const size_t out_bytes = (size_t)ncodes * (2 * nlag + 1) * 16;
    if (!dx.alloc((size_t)n_samples * n_channels * 4) || !dw.alloc((size_t)nobs * 4) || !dpart.alloc(sliding_part_bytes(nobs, ncodes, nlag)) ||
        !dout.alloc(out_bytes)) return TWX_E_NOMEM;
    if (hipMemcpy(dx.p, iq, (size_t)n_samples * n_channels * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dw.p, replica, (size_t)nobs * 4, hipMemcpyHostToDevice) != hipSuccess) return TWX_E_HIP;
    if (int rc = launch_sliding(nullptr, static_cast<const short2*>(dx.p) + channel, n_channels, pt, nobs, ncodes, nlag, static_cast<const float*>(dw.p),
                                ff, phi, scale, static_cast<double*>(dpart.p), static_cast<double*>(dout.p))) return rc;
    return hipMemcpy(out, dout.p, out_bytes, hipMemcpyDeviceToHost) == hipSuccess ? TWX_OK : TWX_E_HIP;
}

int twx_fir_decimate(const int16_t* iq, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                     int32_t dec, int16_t* out_i16, float* out_f32, int64_t* n_out) {
    if (!fir_args_ok(iq, taps, n_out, out_i16, out_f32, n_channels, channel, ntaps, dec, n_in)) return TWX_E_ARG;
    const long long nout = (n_in - ntaps) / dec + 1;
    *n_out = nout;
    const FirGeom g = fir_geom(ntaps, dec);
    const std::vector<float> hp = fir_phase_table(taps, ntaps, dec, g);
    DevBuf dx, dh, dy16, dyf;
    if (!dx.alloc((size_t)n_in * n_channels * 4) || !dh.alloc(hp.size() * 4) || (out_i16 && !dy16.alloc((size_t)nout * 4)) ||
        (out_f32 && !dyf.alloc((size_t)nout * 8))) return TWX_E_NOMEM;
    if (hipMemcpy(dx.p, iq, (size_t)n_in * n_channels * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dh.p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return TWX_E_HIP;
    if (int rc = launch_fir(nullptr, static_cast<const short2*>(dx.p) + channel, n_channels, n_in, static_cast<const float*>(dh.p), ntaps, dec, nout,
                            static_cast<short2*>(dy16.p), static_cast<float2*>(dyf.p))) return rc;
    if (out_i16 && hipMemcpy(out_i16, dy16.p, (size_t)nout * 4, hipMemcpyDeviceToHost) != hipSuccess) return TWX_E_HIP;
    if (out_f32 && hipMemcpy(out_f32, dyf.p, (size_t)nout * 8, hipMemcpyDeviceToHost) != hipSuccess) return TWX_E_HIP;
    return TWX_OK;
}

}  // extern "C"
