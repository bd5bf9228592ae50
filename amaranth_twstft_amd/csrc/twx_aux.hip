// twx_aux.hip — kernels beside the FFT chain: the direct sliding dot-product correlator for short
// codes (tracking stage of experiments/231001_DLL_PLL/rxcomplex.cpp:593-614) and the FIR
// decimating front end (no reference twin; parameters from experiments/2403/zmq_rx.py:208-215).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "../../include/twstft_hip.h"

namespace {

#define AUXCHK(call) do { if ((call) != hipSuccess) { rc = TWX_E_HIP; goto done; } } while (0)

// ---------------------------------------------------------------------------------------------
// Sliding dot product.  For code period p and lag index li (lag = li - nlag):
//   out[p][li] = (scale/nobs) * sum_i x[pt + p*nobs + i] * exp(-2 pi j (ff*(p*nobs+i) + phi)) * w[(i - lag) mod nobs]
// = downconv_trk (rxcomplex.cpp:1051-1061) + the cblas_dgemm against the PRN_mapping replicas
// (:605, :989-999), fused: the replica matrix is never materialised (one LDS segment per chunk),
// one lane per lag (<= 64 lags per wave), samples streamed once (4 B/sample).
// grid = (chunks, ncodes), block = 256 (4 waves; wave w takes samples i = w mod 4)
// ---------------------------------------------------------------------------------------------
constexpr int SD_CH = 4096;      // samples per workgroup
constexpr int SD_MAXL = 64;      // lags per wave

__global__ __launch_bounds__(256) void k_sliding_dot(const short2* __restrict__ x, int nch, long long pt, long long nobs, int nlag,
                                                     const float* __restrict__ w, double ff, double phi, float scale,
                                                     double* __restrict__ partial /*[ncodes][chunks][nl][2]*/) {
    __shared__ float2 sy[SD_CH];
    __shared__ float sw[SD_CH + 2 * 32 + 2];
    __shared__ float2 red[4][SD_MAXL];
    const int p = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int nl = 2 * nlag + 1;
    const long long s0 = (long long)chunk * SD_CH;
    const int cnt = (int)min((long long)SD_CH, nobs - s0);
    const int tid = threadIdx.x;
    // mixed samples of this chunk (fp64 phase reduction, fp32 sincos)
    for (int t = tid; t < cnt; t += 256) {
        const long long i = (long long)p * nobs + s0 + t;        // sample index relative to pt (the NCO runs over the whole block)
        short2 s = x[(pt + i) * nch];
        double ph = ff * (double)i + phi;
        ph -= rint(ph);
        float sn, cs;
        sincospif(-2.0f * (float)ph, &sn, &cs);
        const float re = (float)s.x, im = (float)s.y;
        sy[t] = make_float2(scale * (re * cs - im * sn), scale * (re * sn + im * cs));
    }
    // replica segment: entry u ↔ w[(s0 - nlag + u) mod nobs], u = 0 .. cnt + 2*nlag - 1
    for (int u = tid; u < cnt + 2 * nlag; u += 256) {
        long long k = (s0 - nlag + u) % nobs; if (k < 0) k += nobs;
        sw[u] = w[k];
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    float ar = 0.f, ai = 0.f;
    if (lane < nl) {
        // sample t, lag index li=lane: replica index (s0+t) - (lane - nlag) → segment entry t + 2*nlag - lane
        for (int t = wv; t < cnt; t += 4) {
            const float2 y = sy[t];                   // wave-uniform address: LDS broadcast
            const float c = sw[t + 2 * nlag - lane];  // consecutive lanes, consecutive addresses
            ar = fmaf(y.x, c, ar);
            ai = fmaf(y.y, c, ai);
        }
        red[wv][lane] = make_float2(ar, ai);
    }
    __syncthreads();
    if (wv == 0 && lane < nl) {
        double sr = 0, si = 0;
        for (int k = 0; k < 4; ++k) { sr += red[k][lane].x; si += red[k][lane].y; }
        double* o = partial + (((long long)p * nchunks + chunk) * nl + lane) * 2;
        o[0] = sr; o[1] = si;
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

// ---------------------------------------------------------------------------------------------
// FIR low-pass + decimation of interleaved int16 IQ:  y[m] = sum_j taps[j] * x[m*dec + j]
// (BASELINE.json configs[4]: 70 Msps wideband capture → 5 Msps; ntaps ~ 577, dec = 14).
// One output per thread; the workgroup's input span (256*dec + ntaps samples) is staged in LDS with
// coalesced 4-B loads; taps in LDS; fp32 accumulate; output rounded to int16 IQ (round-half-even)
// or kept as float2.  grid = ceil(nout/256)
// ---------------------------------------------------------------------------------------------
constexpr int FIR_MAXSPAN = 256 * 16 + 1024;
__global__ __launch_bounds__(256) void k_fir_decimate(const short2* __restrict__ x, int nch, long long nin, const float* __restrict__ taps,
                                                      int ntaps, int dec, long long nout, short2* __restrict__ y16, float2* __restrict__ yf) {
    __shared__ short2 sx[FIR_MAXSPAN];
    __shared__ float st[1024];
    const long long m0 = (long long)blockIdx.x * 256;
    const int span = (int)min((long long)(255 * dec + ntaps), nin - m0 * dec);
    for (int t = threadIdx.x; t < span; t += 256) sx[t] = x[(m0 * dec + t) * nch];
    for (int t = threadIdx.x; t < ntaps; t += 256) st[t] = taps[t];
    __syncthreads();
    const long long m = m0 + threadIdx.x;
    if (m >= nout) return;
    const int base = threadIdx.x * dec;
    float ar = 0.f, ai = 0.f;
    for (int j = 0; j < ntaps; ++j) {
        const short2 s = sx[base + j];
        const float c = st[j];
        ar = fmaf((float)s.x, c, ar);
        ai = fmaf((float)s.y, c, ai);
    }
    if (yf) yf[m] = make_float2(ar, ai);
    if (y16) {
        const float r = fminf(fmaxf(rintf(ar), -32768.f), 32767.f), q = fminf(fmaxf(rintf(ai), -32768.f), 32767.f);
        y16[m] = make_short2((short)r, (short)q);
    }
}

}  // namespace

extern "C" {

int twx_sliding_dot(const int16_t* iq, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                    int32_t ncodes, int32_t nlag, const float* replica, double ff, double phi, double scale, double* out) {
    if (!iq || !replica || !out || n_channels < 1 || channel < 0 || channel >= n_channels || nobs < 1 || ncodes < 1 || nlag < 0 || nlag > 31 ||
        pt < 0 || pt + nobs * ncodes > n_samples)
        return TWX_E_ARG;
    int rc = TWX_OK;
    short2* dx = nullptr; float* dw = nullptr; double* dpart = nullptr; double* dout = nullptr;
    const int nl = 2 * nlag + 1;
    const int nchunks = (int)((nobs + SD_CH - 1) / SD_CH);
    AUXCHK(hipMalloc((void**)&dx, (size_t)n_samples * n_channels * 4));
    AUXCHK(hipMalloc((void**)&dw, (size_t)nobs * 4));
    AUXCHK(hipMalloc((void**)&dpart, (size_t)ncodes * nchunks * nl * 16));
    AUXCHK(hipMalloc((void**)&dout, (size_t)ncodes * nl * 16));
    AUXCHK(hipMemcpy(dx, iq, (size_t)n_samples * n_channels * 4, hipMemcpyHostToDevice));
    AUXCHK(hipMemcpy(dw, replica, (size_t)nobs * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sliding_dot, dim3(nchunks, ncodes), dim3(256), 0, 0, dx + channel, n_channels, (long long)pt, (long long)nobs, nlag, dw,
                       ff, phi, (float)scale, dpart);
    AUXCHK(hipGetLastError());
    hipLaunchKernelGGL(k_sliding_reduce, dim3(ncodes), dim3(64), 0, 0, dpart, nchunks, nl, 1.0 / (double)nobs, dout);
    AUXCHK(hipGetLastError());
    AUXCHK(hipMemcpy(out, dout, (size_t)ncodes * nl * 16, hipMemcpyDeviceToHost));
done:
    (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dpart); (void)hipFree(dout);
    return rc;
}

int twx_fir_decimate(const int16_t* iq, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                     int32_t dec, int16_t* out_i16, float* out_f32, int64_t* n_out) {
    if (!iq || !taps || !n_out || (!out_i16 && !out_f32) || n_channels < 1 || channel < 0 || channel >= n_channels || ntaps < 1 || ntaps > 1024 ||
        dec < 1 || dec > 16 || n_in < ntaps)
        return TWX_E_ARG;
    int rc = TWX_OK;
    const long long nout = (n_in - ntaps) / dec + 1;
    *n_out = nout;
    short2* dx = nullptr; float* dt = nullptr; short2* dy16 = nullptr; float2* dyf = nullptr;
    AUXCHK(hipMalloc((void**)&dx, (size_t)n_in * n_channels * 4));
    AUXCHK(hipMalloc((void**)&dt, (size_t)ntaps * 4));
    if (out_i16) AUXCHK(hipMalloc((void**)&dy16, (size_t)nout * 4));
    if (out_f32) AUXCHK(hipMalloc((void**)&dyf, (size_t)nout * 8));
    AUXCHK(hipMemcpy(dx, iq, (size_t)n_in * n_channels * 4, hipMemcpyHostToDevice));
    AUXCHK(hipMemcpy(dt, taps, (size_t)ntaps * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fir_decimate, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, 0, dx + channel, n_channels, (long long)n_in, dt, ntaps, dec,
                       nout, dy16, dyf);
    AUXCHK(hipGetLastError());
    if (out_i16) AUXCHK(hipMemcpy(out_i16, dy16, (size_t)nout * 4, hipMemcpyDeviceToHost));
    if (out_f32) AUXCHK(hipMemcpy(out_f32, dyf, (size_t)nout * 8, hipMemcpyDeviceToHost));
done:
    (void)hipFree(dx); (void)hipFree(dt); (void)hipFree(dy16); (void)hipFree(dyf);
    return rc;
}

}  // extern "C"
