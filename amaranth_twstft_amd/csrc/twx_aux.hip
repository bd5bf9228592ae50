// twx_aux.hip — kernels beside the FFT chain: the direct sliding dot-product correlator for short
// codes (tracking stage of experiments/231001_DLL_PLL/rxcomplex.cpp:593-614) and the FIR
// decimating front end (no reference twin; parameters from experiments/2403/zmq_rx.py:208-215).
// Both are bounded by the fp32 vector rate, not by HBM (a12 ~50 flop/B; the 577-tap
// decimator 41 flop/B against a ridge of ~20), so they are laid out for the VALU: every LDS read
// feeds several FMAs, taps/replica reads are conflict-free, all 64 lanes of every wave work.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <atomic>
#include <vector>
#include "twx_internal.h"
#include "twx_fir_table.h"
#include "twx_track_core.h"

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
constexpr int SD_T = 4;                                          // consecutive samples per lane and pass
constexpr int SD_NPASS = SD_CH / (SD_T * SD_NT);                 // passes of a full piece
struct SdRot { float2 r[SD_NPASS]; };                            // exp(-2 pi j ff 1024 k): rotation of pass k against pass 0 (host, fp64)

// Register tile: a lane owns SD_T = 4 CONSECUTIVE samples per pass, so the replica values it needs for all NL lags are the
// NL+3 consecutive entries c[4g-l+j] — every value read from LDS feeds four packed FMAs.  The segment lies in LDS as it
// lies in memory, and a lane fetches its entries as ceil((NL+3)/4) aligned ds_read_b128 (consecutive lanes, consecutive
// 16-byte words: conflict-free; 15 reads per pass at NLAG = 28, where the transposed layout of the first version needed 60
// ds_read_b32).  The NCO is evaluated once per lane and PIECE (fp64 phase reduction + sincospi); a pass multiplies it by
// the pass's rotation exp(-2 pi j ff 1024 k) from a 16-entry table in the kernel arguments (evaluated on the host in fp64), and the four samples of the pass
// are stepped by the fp32 rotation exp(-2 pi j ff): three roundings on top of the table's (3e-7 relative at most).
// XT = short2: interleaved int16 IQ of a capture; XT = float2: complex-float samples (the x2-interpolated stream the DLL/PLL
// receiver tracks on, rxcomplex.cpp:477,602: dev_smp)
__device__ __forceinline__ void sd_opaque(short2& v) { unsigned t = *reinterpret_cast<const unsigned*>(&v); asm volatile("" : "+v"(t)); v = *reinterpret_cast<const short2*>(&t); }
__device__ __forceinline__ void sd_opaque(float2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
// T = consecutive samples per lane and pass, CH = samples per LDS piece, WPC = workgroups per CU the register budget allows.
// (4, 16384, 2): the wide windows (114 accumulator registers at NLAG = 28).  (8, 8192, 4): windows of up to +-8 lags, where the
// accumulators are few and the kernel is a stream over the samples: eight samples per lane fetched as 16-byte loads, half the
// LDS per workgroup, twice the workgroups per CU — four times the bytes in flight (nch = 1 and nobs a multiple of 8 only: every
// group of eight is whole, the launcher falls back to the general form otherwise).
template <int NLAG, typename XT, int T = SD_T, int CH = SD_CH, int WPC = 2>
__global__ __launch_bounds__(SD_NT, WPC) void k_sliding_dot(const XT* __restrict__ x, int nch, long long pt, long long nobs, int nlag, int chunk_len,
                                                       const float* __restrict__ w, double ff, double phi, float scale, float rot_c, float rot_s,
                                                       SdRot rot, double* __restrict__ partial /*[ncodes][chunks][2*nlag+1][2]*/, int ch) {
    constexpr int NL = 2 * NLAG + 1;
    constexpr int NQ = (NL + T - 1 + 3) / 4;                    // 16-byte words a lane reads per pass
    constexpr int NE4 = CH / 4 + NQ;                               // 16-byte words of a full chunk's segment
    constexpr int WG = T / 4;                                      // 16-byte words per group of samples
    constexpr bool VEC = T == 8;                                   // whole groups, one channel: the samples of a group as 16-byte loads
    __shared__ float4 sw4[NE4];                                    // replica segment; reused by the final reduction
    float* sw = reinterpret_cast<float*>(sw4);
    static_assert(NL * SD_NT <= 4 * NE4, "reduction buffer must fit the replica segment");
    static_assert(T == 4 || T == 8, "whole 16-byte words per group of samples");
    static_assert(CH / (T * SD_NT) <= SD_NPASS, "pass rotation table too short");
    const int p = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int tid = threadIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[NL];                                                    // (re, im) pairs: one v_pk_fma_f32 per lag and sample
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = f2{0.f, 0.f};
    // a workgroup's chunk may be longer than the LDS segment: it is walked in pieces of SD_CH samples with ONE set of
    // accumulators, so the block reduction below runs once per workgroup
    const long long c0 = (long long)chunk * chunk_len;
    const long long c1 = min(c0 + (long long)chunk_len, nobs);
    for (long long s0 = c0; s0 < c1; s0 += CH) {
    const int cnt = (int)min((long long)CH, c1 - s0);
    const int ngrp = (cnt + T - 1) / T;
    if (s0 > c0) __syncthreads();                                  // the previous piece's replica segment is no longer read
    // entry u <-> w[(s0 - NLAG + u) mod nobs]; every 16-byte word a lane will read is written (the overhang with whatever
    // follows in the code: it meets samples that are masked to zero)
    // straight into LDS (global_load_lds_dword: lane i of a wave writes base + 4 i, no register in between), every entry
    // of the piece in flight at once: fetched through registers one or eight at a time, the 64-KB segment cost 8 of a full
    // piece's 40 microseconds
    {
        long long k = s0 - NLAG;                                   // 0 <= s0 < nobs
        if (k < 0) { k += nobs; if (k < 0) { k %= nobs; if (k < 0) k += nobs; } }      // the division only for periods shorter than the lag window
        k += tid;
        if (k >= nobs) { k -= nobs; if (k >= nobs) k %= nobs; }
        const long long step = SD_NT % nobs;
        const int nent = 4 * (WG * ngrp + NQ);
        float* wave_base = sw + (tid & ~63);
        for (int u0 = 0; u0 < nent; u0 += SD_NT) {
            if (u0 + tid < nent)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + k),
                                                 (__attribute__((address_space(3))) void*)(wave_base + u0), 4, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x8F70);                    // vmcnt(32): the counter has six bits
            k += step; if (k >= nobs) k -= nobs;
        }
    }
    // the samples of the NEXT pass are requested before this pass's arithmetic (two waves per SIMD do not hide a global
    // load round trip per pass by themselves); loads are unconditional with clamped indices, values masked below
    const long long ilast = (long long)p * nobs + s0 + cnt - 1;
    XT nx[T];
    auto load_group = [&](int g) {
        const long long i0 = (long long)p * nobs + s0 + (long long)T * min(g, ngrp - 1);
        if constexpr (VEC) {
            // 16-byte loads at the samples' own alignment (4 or 8 bytes: pt is any sample)
            typedef unsigned uvec4 __attribute__((ext_vector_type(4), aligned(4)));
            constexpr int NV = (int)(T * sizeof(XT) / 16);
            if (nch == 1) {
                const uvec4* src = reinterpret_cast<const uvec4*>(x + pt + i0);
                unsigned raw[NV * 4];
#pragma unroll
                for (int v = 0; v < NV; ++v) { const uvec4 q = src[v]; raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w; }
                __builtin_memcpy(nx, raw, sizeof raw);
            } else {
                // a channel of a two-channel capture ([a0 b0 a1 b1 ...], the reference's file format): the eight frames of the group
                // as 16-byte loads from the FRAME base (x points at this channel's sample of frame 0: x - ch is the frame), the
                // channel's words picked out — twice the bytes, which this kernel has to spare, and whole sectors either way
                constexpr int WPS = (int)(sizeof(XT) / 4);                 // 32-bit words per sample
                const uvec4* src = reinterpret_cast<const uvec4*>(x - ch + (pt + i0) * 2);
                unsigned raw[NV * 8];
#pragma unroll
                for (int v = 0; v < 2 * NV; ++v) { const uvec4 q = src[v]; raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w; }
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    unsigned wv[WPS];
#pragma unroll
                    for (int e = 0; e < WPS; ++e) wv[e] = ch ? raw[(2 * j + 1) * WPS + e] : raw[2 * j * WPS + e];
                    __builtin_memcpy(&nx[j], wv, sizeof(XT));
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) nx[j] = x[(pt + min(i0 + j, ilast)) * nch];
        }
    };
    load_group(tid);
    float bcs, bsn;                                                // the lane's NCO at its first sample of this piece
    {
        double ph = ff * (double)((long long)p * nobs + s0 + (long long)T * tid) + phi;   // fp64 phase reduction, fp32 sincos
        ph -= rint(ph);
        sincospif(-2.0f * (float)ph, &bsn, &bcs);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this thread's part of the segment has landed
    __syncthreads();
    int pass = 0;
    for (int g = tid; g < ngrp; g += SD_NT, ++pass) {
        const int t0 = T * g;
        XT sm[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            // opaque hand-over: otherwise the conversion to float moves up behind the load in the PREVIOUS pass (the loop then
            // carries floats) and that pass waits for the samples it asked for a few hundred cycles earlier
            sm[j] = nx[j];
            sd_opaque(sm[j]);
        }
        const float2 pr = rot.r[pass];                             // kernel argument, uniform index: a scalar load
        float cs = bcs * pr.x - bsn * pr.y, sn = bcs * pr.y + bsn * pr.x;
        f2 y[T];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const float re = (float)sm[j].x, im = (float)sm[j].y;
            const float m = (t0 + j < cnt) ? scale : 0.f;
            y[j] = f2{m * (re * cs - im * sn), m * (re * sn + im * cs)};
            const float c2 = cs * rot_c - sn * rot_s, s2 = cs * rot_s + sn * rot_c;
            cs = c2; sn = s2;
        }
        // sample t0+j, lag index l  <->  replica entry T*g + (j + 2*NLAG - l)
        // The next pass's samples are asked for HERE, after the mixing: their landing registers are single dwords, and next to
        // the mixing's temporaries they ended up as the unused halves of packed operands — the pass then waited for them a few
        // hundred cycles after asking (the packed instruction reads the pair).  In the accumulation below every packed operand
        // is a whole tuple (accumulator, y, the word read from LDS).
        __builtin_amdgcn_sched_barrier(0);
        load_group(g + SD_NT);
        __builtin_amdgcn_sched_barrier(0);
        // word by word, two words ahead of the arithmetic: the four entries of a word meet their (sample, lag) pairs and are dead
        // before the next word is needed.  The scheduling barriers keep that order — left alone, the scheduler asks for all
        // NQ words first (60 registers on top of the 114 accumulators: spills at the 256-register limit of two waves per SIMD)
        float4 q0 = sw4[WG * g], q1 = sw4[WG * g + 1];
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
            float4 q2 = q1;
            if (m + 2 < NQ) q2 = sw4[WG * g + m + 2];
            const float cw[4] = {q0.x, q0.y, q0.z, q0.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const int l = j + 2 * NLAG - (4 * m + e);
                    if (l >= 0 && l < NL) acc[l] = __builtin_elementwise_fma(y[j], f2{cw[e], cw[e]}, acc[l]);
                }
            }
            q0 = q1; q1 = q2;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    // block reduction through LDS, one component at a time: buf[l][tid]; thread r = (l, quarter) sums 64 lanes as sixteen
    // 16-byte words, starting at a lane-dependent rotation so that the eight lanes of a ds_read_b128 beat hit 32 different banks
    float* buf = sw;
    const int r = tid, rl = r >> 2, rq = r & 3;
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int comp = 0; comp < 2; ++comp) {
        __syncthreads();
#pragma unroll
        for (int l = 0; l < NL; ++l) buf[l * SD_NT + tid] = comp ? acc[l].y : acc[l].x;
        __syncthreads();
        if (rl < NL) {
            const float4* q = reinterpret_cast<const float4*>(buf + rl * SD_NT + rq * 64);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float4 t = q[(j + r) & 15]; a0 += t.x; a1 += t.y; a2 += t.z; a3 += t.w; }
            part[comp] = (a0 + a1) + (a2 + a3);
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
    for (int c0 = 0; c0 < nchunks; c0 += 16) {   // sixteen loads in flight, added in the fixed order: bit-reproducible
        double2 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            t[k] = *reinterpret_cast<const double2*>(partial + (((long long)p * nchunks + min(c0 + k, nchunks - 1)) * nl + li) * 2);
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (c0 + k < nchunks) { sr += t[k].x; si += t[k].y; } }
    }
    out[((long long)p * nl + li) * 2] = sr * inv_nobs;
    out[((long long)p * nl + li) * 2 + 1] = si * inv_nobs;
}

// ---------------------------------------------------------------------------------------------
// The sliding dot product on the MATRIX cores (round 5).  The reference evaluates this step as a GEMM — cblas_dgemm of the
// (2 nlag + 1) x nobs replica matrix against the nobs x 2(bps-1) matrix of mixed samples (rxcomplex.cpp:605,989-999) — and so does
// this kernel, on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, at the fp32 vector PEAK rate, which the
// packed-FMA form above cannot reach: v_pk_fma_f32 issues at half of it):
//     C[li][2p + c] = sum_i  W[li][i] * Y[i][2p + c],     W[li][i] = w[(i - (li - nlag)) mod nobs],   Y[i][2p + c] = Re/Im of the mixed sample
// W is Toeplitz: it is never materialised, lane (i = l & 15, k = l >> 4) of an A operand reads w_seg[kk + k - li + 63] from the
// replica segment in LDS (consecutive lanes, consecutive words).  Y is produced by the workgroup itself, 64 samples ("piece") of
// up to 32 codes at a time: coalesced int16 / float loads, NCO per (sample, code) = exp(-2 pi j (ff i + phi)) [fp64 phase reduction,
// one sincos per thread and piece] * exp(-2 pi j ff p nobs) [per code, once per workgroup], written column-major
// (YT[col][sample], row stride 68 words: the writes of consecutive samples and the B-operand reads lane (j = l & 15, k = l >> 4)
// -> YT[16 nt + j][kk + k] are both conflict-free).  Wave w owns the 16 lag rows of M tile (w mod MT) and every (4/MT)-th N
// tile: no reduction across waves.  Pieces are dealt round-robin to the workgroups of a code group (grid.x), two buffers.
// Every wave does both jobs in turn (matrix-core phase on piece q, then the mixing of piece q + G; four workgroups = 16 waves per CU,
// whose phases drift apart, keep both pipes busy).  Tried and measured slower: dedicated mixer / matrix-core waves in workgroups of
// eight (two workgroups per CU: 121 us against 95 for 96 codes — profiles/r05_sliding_mfma.txt has the whole series).
//   grid = (workgroups per code group, ceil(ncodes / 32)), block = 256;   partial: float [part = blockIdx.x][li][2 p + c]
// ---------------------------------------------------------------------------------------------
constexpr int SM_P = 64, SM_LD = 68, SM_CG = 32;                 // samples per piece, words per YT row, codes per group
template <typename XT, int MT, int NTW>
__global__ __launch_bounds__(256, 4) void k_sliding_mfma(const XT* __restrict__ x, int nch, long long pt, long long nobs, int ncodes, int nlag,
                                                        const float* __restrict__ w, double ff, double phi, float scale, int npieces,
                                                        double* __restrict__ partial /*as float [gridDim.x][2*nlag+1][2*ncodes]*/, int ablate) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    // MT = M tiles (16 lag rows each: 2 for windows of up to 31 lags, 4 otherwise), NTW = N tiles of the group (16 columns = 8 codes
    // each; the launcher instantiates the count of the fullest group, columns past a group's own are multiplied and never stored)
    __shared__ float yt[2][64 * SM_LD];
    __shared__ float wseg[2][SM_P + 64];
    __shared__ float2 ecode[SM_CG];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.y, p0 = cg * SM_CG, ncg = min(SM_CG, ncodes - p0);
    const int nl = 2 * nlag + 1;
    if (tid < ncg) {                                               // exp(-2 pi j ff p nobs): the code's place in the stream
        double ph = ff * (double)((long long)(p0 + tid) * nobs);
        ph -= rint(ph);
        float sn, cs;
        sincospif(-2.0f * (float)ph, &sn, &cs);
        ecode[tid] = make_float2(cs * scale, sn * scale);          // the scale rides on the per-code factor
    }
    // wave wv owns the 16 lag rows of M tile (wv mod MT) and every (4/MT)-th N tile: NTW accumulator tiles per wave, no sum across waves
    // (all tiles in every wave with the samples split over the waves needs 64 accumulator registers next to the three sets of samples
    // in flight: it spilled at the 128 registers of four workgroups per CU and ran 129 us where this form runs 95)
    constexpr int NG = 4 / MT;
    constexpr int NTL = (NTW + NG - 1) / NG;                       // N tiles per wave
    const int mt = wv % MT, ng = wv / MT;
    f4 acc[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    // mixing roles: thread (si = tid & 63, c0 = its wave) takes sample si of the piece for the codes c0, c0 + 4, ... of the group.
    // c0 is wave-uniform: the codes' stream bases are scalar (SGPR pair + 32-bit lane offset: no 64-bit vector address arithmetic
    // per load — with it, the address code of a piece was as long as its mixing) and the per-code NCO factors are scalar operands
    const int si = tid & 63, c0 = wv;
    constexpr int NJ = SM_CG / 4;
    // THREE pieces of samples on their way per mixer thread (three register sets, the loop below is unrolled by three): one piece
    // ahead, the bytes in flight per CU (16 KB) against the load latency under load (~4 us) were 1 TB/s for the whole chip — the
    // kernel waited for memory with idle matrix cores
    XT nxr[3][NJ];
    float wregr[3] = {0.f, 0.f, 0.f};
    auto load_piece = [&](int q, XT (&nx)[NJ]) {
        const long long i = (long long)q * SM_P + si;
        const unsigned voff = (unsigned)(min(i, nobs - 1) * nch) * (unsigned)sizeof(XT);      // clamped: the value is masked below
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pl = min(c0 + 4 * j, ncg - 1);
            const char* base = reinterpret_cast<const char*>(x + (pt + (long long)(p0 + pl) * nobs) * nch);      // uniform
            nx[j] = *reinterpret_cast<const XT*>(base + voff);
        }
    };
    // replica segment of a piece: entry u <-> w[(64 q + u - 63 + nlag) mod nobs], u < 128.  Asked for with the samples.
    auto load_w = [&](int q, float& wreg) {
        if (tid < SM_P + 64) {
            long long k = (long long)q * SM_P + tid - 63 + nlag;  // > -64, < nobs + 128
            if (nobs >= 256) { if (k < 0) k += nobs; else if (k >= nobs) k -= nobs; }
            else { k %= nobs; if (k < 0) k += nobs; }              // the division only for periods shorter than the segment
            wreg = w[k];
        }
    };
    // NCO of the thread's sample: exact (fp64 phase reduction, fp32 sincos) on every fourth of the workgroup's pieces, stepped by the
    // rotation between consecutive pieces of this workgroup exp(-2 pi j ff 64 gridDim.x) in between (three fp32 complex products at most)
    float rgc, rgs;
    {
        double ph = ff * (double)((long long)SM_P * gridDim.x);
        ph -= rint(ph);
        sincospif(-2.0f * (float)ph, &rgs, &rgc);
    }
    float ncs = 1.f, nsn = 0.f;
    float ecx[NJ], ecy[NJ];                                        // this wave's codes: uniform values (scalar registers), set below
    auto mix_piece = [&](int q, int buf, int it, const XT (&nx)[NJ], float wreg) {
        const long long i = (long long)q * SM_P + si;
        if ((it & 3) == 0) {
            double ph = ff * (double)i + phi;
            ph -= rint(ph);
            sincospif(-2.0f * (float)ph, &nsn, &ncs);
        } else {
            const float c2 = ncs * rgc - nsn * rgs, s2 = ncs * rgs + nsn * rgc;
            ncs = c2; nsn = s2;
        }
        const float m = i < nobs ? 1.f : 0.f;
        const float cs = ncs * m, sn = nsn * m;
        float* yrow = yt[buf] + (2 * c0) * SM_LD + si;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (c0 + 4 * j < ncg) {                                 // uniform
                const float ec = cs * ecx[j] - sn * ecy[j], es = cs * ecy[j] + sn * ecx[j];
                const float re = (float)nx[j].x, im = (float)nx[j].y;
                yrow[(8 * j) * SM_LD] = re * ec - im * es;
                yrow[(8 * j + 1) * SM_LD] = re * es + im * ec;
            }
        }
        if (tid < SM_P + 64) wseg[buf][tid] = wreg;
    };
    const int G = gridDim.x, qlast = npieces - 1;
    int q = blockIdx.x;
    // loads are unconditional (a piece past the end is the last one again; its mixing lands in a buffer nobody reads): behind a
    // branch the wait-count pass assumes the loads were skipped and every mixing waits with vmcnt(0) — for the set asked for one step
    // ago, i.e. a full memory latency per piece
#pragma unroll
    for (int k = 0; k < 3; ++k) { load_piece(min(q + k * G, qlast), nxr[k]); load_w(min(q + k * G, qlast), wregr[k]); }
    __syncthreads();                                               // ecode visible
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float2 e = ecode[min(c0 + 4 * j, ncg - 1)];
        ecx[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e.x)));
        ecy[j] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e.y)));
    }
    mix_piece(min(q, qlast), 0, 0, nxr[0], wregr[0]);              // piece 0 of this workgroup into buffer 0
    load_piece(min(q + 3 * G, qlast), nxr[0]); load_w(min(q + 3 * G, qlast), wregr[0]);
    // matrix-core phase: 16 K steps of 4 samples on the wave's tiles
    int buf = 0, it = 1;
#define SM_STEP(K_)                                                                                                       \
    {                                                                                                                     \
        __syncthreads();                                          /* piece q is in yt[buf] / wseg[buf]; the other is free */ \
        if (ablate != 1) {                                                                                                \
            const float* ws = wseg[buf] + 63 - 16 * mt - (lane & 15) + (lane >> 4);                                       \
            const float* yb = yt[buf] + ((lane & 15) + 16 * ng) * SM_LD + (lane >> 4);                                    \
            _Pragma("unroll") for (int kk = 0; kk < SM_P; kk += 4) {                                                      \
                const float a = ws[kk];                                                                                   \
                _Pragma("unroll") for (int i = 0; i < NTL; ++i)                                                           \
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, yb[NG * i * 16 * SM_LD + kk], acc[i], 0, 0, 0);      \
            }                                                                                                             \
        }                                                                                                                 \
        if (ablate != 2) {                                                                                                \
            mix_piece(min(q + G, qlast), buf ^ 1, it, nxr[K_], wregr[K_]);                                                \
            load_piece(min(q + 4 * G, qlast), nxr[K_]); load_w(min(q + 4 * G, qlast), wregr[K_]);                         \
        }                                                                                                                 \
        q += G; buf ^= 1; ++it;                                                                                           \
    }
    while (q < npieces) {
        SM_STEP(1)
        if (q >= npieces) break;
        SM_STEP(2)
        if (q >= npieces) break;
        SM_STEP(0)
    }
#undef SM_STEP
    // D of 16x16x4: column = lane & 15, rows 4 (lane >> 4) + r.  The workgroup's part leaves as floats (what the accumulators are),
    // [part][li][2 p + c]: the sixteen lanes of a row write 64 contiguous bytes
    float* po = reinterpret_cast<float*>(partial) + (long long)blockIdx.x * nl * 2 * ncodes;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int col = (ng + NG * i) * 16 + (lane & 15);
        if (col < 2 * ncg) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int li = 16 * mt + 4 * (lane >> 4) + r;
                if (li < nl) po[(long long)li * 2 * ncodes + 2 * p0 + col] = acc[i][r];
            }
        }
    }
}

// The sum over the parts of the matrix-core form, float [part][li][2 p + c] -> out[p][li][c] (double): block = 64 columns x 16 groups
// of parts; a group adds every sixteenth part in order (fp64), the sixteen sums are added in a fixed order — bit-reproducible.
// grid = (ceil(2 ncodes / 64), nl)
__global__ __launch_bounds__(1024) void k_sliding_reduce_wide(const float* __restrict__ partial, int nparts, int nl, int ncols, double inv_nobs, double* __restrict__ out) {
    __shared__ double sh[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), li = blockIdx.y, g = threadIdx.x >> 6;
    double s = 0;
    if (col < ncols) {
        const float* src = partial + (long long)li * ncols + col;
        const long long pstride = (long long)nl * ncols;
        for (int c0 = g; c0 < nparts; c0 += 16 * 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = src[(long long)min(c0 + 16 * k, nparts - 1) * pstride];
#pragma unroll
            for (int k = 0; k < 8; ++k) if (c0 + 16 * k < nparts) s += (double)t[k];
        }
    }
    sh[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && col < ncols) {
        double a = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[k][threadIdx.x & 63];
        out[((long long)(col >> 1) * nl + li) * 2 + (col & 1)] = a * inv_nobs;
    }
}

// samples per workgroup: at most SD_CH (the LDS segment), chosen so that the grid is a whole number of "rounds" of two
// workgroups per CU (a 600-workgroup grid on 512 slots runs a second, mostly empty round)
// the streaming form for narrow lag windows (see the kernel): whole groups of eight samples, one channel
bool sliding_narrow(long long nobs, int nlag, int nch) {
    static const bool off = getenv("TWX_SLIDING_NARROW") && atoi(getenv("TWX_SLIDING_NARROW")) == 0;      // A/B (profiles/r04_sliding_scan.txt)
    return !off && nlag <= 8 && (nch == 1 || nch == 2) && nobs % 8 == 0;
}
int sliding_chunk(long long nobs, int ncodes, bool narrow, bool wide8 = false) {
    static const int ncu = [] {                                    // one process drives one GPU (or GPUs of one kind)
        int n = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
        return 256;
    }();
    // one workgroup per slot (two per CU) where the work allows it: chunks of a code of nobs/ceil(...) samples, any length (the
    // kernel walks a chunk in LDS-sized pieces), at least 4096 samples so that the block reduction stays a small part
    const long long slots = (narrow ? 4ll : 2ll) * ncu;
    const long long per_code = std::max<long long>(1, slots / ncodes);                    // chunks per code in one round
    long long len = (nobs + per_code - 1) / per_code;
    const long long gran = SD_NT * ((narrow || wide8) ? 8 : SD_T);
    len = std::max<long long>(4096, ((len + gran - 1) / gran) * gran);
    return (int)std::min<long long>(len, 1ll << 24);
}
// The matrix-core form: wide lag windows (the narrow ones are a stream over the samples, bound by HBM) and enough codes to fill
// a 16-column tile to a useful degree.  TWX_SLIDING_MFMA=0 / 1 forces the choice (A/B runs, profiles/r05_sliding_mfma.txt).
bool sliding_mfma(long long nobs, int ncodes, int nlag) {
    static const int force = getenv("TWX_SLIDING_MFMA") ? atoi(getenv("TWX_SLIDING_MFMA")) : -1;
    if (force == 0) return false;
    if (force == 1) return true;
    // measured (profiles/r05_sliding_mfma.txt): +-28 lags x 96 codes 86.5 TFLOP/s against 68.3 of the packed-FMA form, x 24 codes
    // 53.6 against 55.0 — three N tiles leave the matrix cores a quarter idle and the fixed cost of a launch (12 us) weighs more
    return nlag > 8 && ncodes >= 40 && nobs >= 4096;
}
// workgroups per code group: between 2 and 4 per CU, the count that wastes the least of the last round of 64-sample pieces
int sliding_mfma_parts(long long nobs, int ncodes) {
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    const long long pieces = (nobs + SM_P - 1) / SM_P;
    const int groups = (ncodes + SM_CG - 1) / SM_CG;
    const long long lo = std::max<long long>(1, 3ll * ncu / groups), hi = std::max<long long>(lo, 4ll * ncu / groups);
    if (pieces <= hi) return (int)pieces;
    long long best = hi; double bw = 1e9;
    for (long long n = hi; n >= lo; --n) {
        const long long per = (pieces + n - 1) / n;
        const double waste = (double)(per * n) / (double)pieces;
        if (waste < bw - 1e-9) { bw = waste; best = n; }
    }
    return (int)best;
}
template <typename XT>
int launch_sliding(hipStream_t st, const XT* dx, int nch, long long pt, long long nobs, int ncodes, int nlag, const float* dw,
                   double ff, double phi, double scale, double* dpart, double* dout, int ch = 0) {       // dx: the channel's sample of frame 0; ch: its place in the frame
    if (sliding_mfma(nobs, ncodes, nlag)) {
        const int npieces = (int)((nobs + SM_P - 1) / SM_P);
        const int nparts = sliding_mfma_parts(nobs, ncodes);
        const dim3 grid(nparts, (ncodes + SM_CG - 1) / SM_CG), block(256);
        const int tiles = (2 * std::min(ncodes, SM_CG) + 15) / 16;                  // N tiles of the fullest code group
        static const int ablate = getenv("TWX_SM_ABLATE") ? atoi(getenv("TWX_SM_ABLATE")) : 0;        // diagnostic: 1 = no MFMAs, 2 = no mixing
#define SM_GO(MT_, NTW_) hipLaunchKernelGGL((k_sliding_mfma<XT, MT_, NTW_>), grid, block, 0, st, dx, nch, pt, nobs, ncodes, nlag, dw, ff, phi, (float)scale, npieces, dpart, ablate)
#define SM_MT(MT_) switch (tiles) { case 1: SM_GO(MT_, 1); break; case 2: SM_GO(MT_, 2); break; case 3: SM_GO(MT_, 3); break; default: SM_GO(MT_, 4); break; }
        if (2 * nlag + 1 <= 32) { SM_MT(2) } else { SM_MT(4) }
#undef SM_MT
#undef SM_GO
        if (hipGetLastError() != hipSuccess) return TWX_E_HIP;
        hipLaunchKernelGGL(k_sliding_reduce_wide, dim3((2 * ncodes + 63) / 64, 2 * nlag + 1), dim3(1024), 0, st, reinterpret_cast<const float*>(dpart), nparts,
                           2 * nlag + 1, 2 * ncodes, 1.0 / (double)nobs, dout);
        return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
    }
    const bool narrow = sliding_narrow(nobs, nlag, nch);
    // wide windows, eight samples per lane (whole groups of eight, one channel): the per-pass overhead (mixing, addresses, the
    // LDS words) is shared by twice the FMAs
    static const bool wide8_off = getenv("TWX_SLIDING_WIDE8") && atoi(getenv("TWX_SLIDING_WIDE8")) == 0;
    const bool wide8 = !narrow && !wide8_off && nlag > 8 && (nch == 1 || nch == 2) && nobs % 8 == 0;
    const int clen = sliding_chunk(nobs, ncodes, narrow, wide8);
    const int nchunks = (int)((nobs + clen - 1) / clen);
    const dim3 grid(nchunks, ncodes), block(SD_NT);
    const int T = (narrow || wide8) ? 8 : SD_T;
    const double two_pi = 6.283185307179586476925286766559;
    const float rot_c = (float)cos(two_pi * ff), rot_s = (float)(-sin(two_pi * ff));          // exp(-2 pi j ff)
    SdRot rot;
    for (int k = 0; k < SD_NPASS; ++k) {
        double a = ff * (double)(T * SD_NT * k);
        a -= rint(a);
        rot.r[k] = make_float2((float)cos(two_pi * a), (float)(-sin(two_pi * a)));
    }
#define SD_GO(NL_) hipLaunchKernelGGL((k_sliding_dot<NL_, XT>), grid, block, 0, st, dx, nch, pt, nobs, nlag, clen, dw, ff, phi, (float)scale, rot_c, rot_s, rot, dpart, ch)
#define SD_GO8(NL_) hipLaunchKernelGGL((k_sliding_dot<NL_, XT, 8, 8192, 4>), grid, block, 0, st, dx, nch, pt, nobs, nlag, clen, dw, ff, phi, (float)scale, rot_c, rot_s, rot, dpart, ch)
#define SD_GOW(NL_) hipLaunchKernelGGL((k_sliding_dot<NL_, XT, 8, 16384, 2>), grid, block, 0, st, dx, nch, pt, nobs, nlag, clen, dw, ff, phi, (float)scale, rot_c, rot_s, rot, dpart, ch)
    if (narrow) { if (nlag <= 4) SD_GO8(4); else SD_GO8(8); }
    else if (wide8) { if (nlag <= 16) SD_GOW(16); else if (nlag <= 28) SD_GOW(28); else SD_GOW(31); }
    else if (nlag <= 4) SD_GO(4); else if (nlag <= 8) SD_GO(8); else if (nlag <= 16) SD_GO(16); else if (nlag <= 28) SD_GO(28); else SD_GO(31);
#undef SD_GO
#undef SD_GO8
#undef SD_GOW
    if (hipGetLastError() != hipSuccess) return TWX_E_HIP;
    hipLaunchKernelGGL(k_sliding_reduce, dim3(ncodes), dim3(64), 0, st, dpart, nchunks, 2 * nlag + 1, 1.0 / (double)nobs, dout);
    return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
}
size_t sliding_part_bytes(long long nobs, int ncodes, int nlag) {
    const int clen = std::min(sliding_chunk(nobs, ncodes, false), sliding_chunk(nobs, ncodes, true));      // the form is chosen at launch (channel count); the wide8 chunks are no shorter
    const size_t parts = std::max<size_t>((size_t)((nobs + clen - 1) / clen), (size_t)sliding_mfma_parts(nobs, ncodes));
    return (size_t)ncodes * parts * (2 * nlag + 1) * 16;
}

// ---------------------------------------------------------------------------------------------
// FIR low-pass + decimation of interleaved int16 IQ:  y[m] = sum_j taps[j] * x[m*D + j]
// (BASELINE.json configs[4]: 70 Msps wideband capture -> 5 Msps; 577 taps, D = 14: 82 FMA per INPUT sample and
// component, i.e. bounded by the fp32 vector rate, not by the 4.6 B/sample it moves).
// Polyphase form  y[m] = sum_p sum_a h[a*D+p] * x_p[m+a],  x_p[q] = x[q*D+p]:  D stride-1 filters of A = ceil(ntaps/D) taps.
// A workgroup of FIR_NT threads produces FIR_NT*4 consecutive outputs, thread t the four outputs 4t..4t+3:
//   * its input span is staged in LDS as the raw int16 pairs (4 B per sample: 31 KB per workgroup, so five workgroups
//     share a CU and cover each other's latencies; as floats it was one wave per SIMD), phase-major: slot(p, q) =
//     p*PSQ + q.  Lane t, group sh reads x_p[4t + 4sh .. +3] as ONE ds_read_b128 (consecutive lanes, consecutive 16-byte
//     words: conflict-free), converted to float as it is read (two sign-extending converts per four packed FMAs);
//     PSQ/4 is odd, so the staging writes of consecutive phases spread over the banks;
//   * each value read feeds the four outputs (8 FMAs) with four different taps; the taps are wave-uniform, read with
//     scalar loads from a phase-major table hp[p][3 + a] that carries 3 zeros in front and zeros behind, so the
//     loop has no edge cases; the steps of the last group that meet only padding are skipped (uniform branch);
//   * outputs leave as one 16-B (int16 IQ) or two 16-B (float) stores per lane.
// The kernel is bound by vector issue slots (19 waves per SIMD x 21 k cycles of instructions = the measured time), so
// what counts is the instruction count: per group of four steps 1 LDS read, 8 converts, 16 packed FMAs.
// grid = ceil(nout / (4*FIR_NT)), dynamic LDS = D*PSQ*4 bytes
// ---------------------------------------------------------------------------------------------
constexpr int FIR_NT = 128, FIR_K = 4, FIR_OUT = FIR_NT * FIR_K;
// The EIGHT-output form (k_fir_poly8, round 4): thread t of a half-workgroup of FIR8_T threads owns the eight outputs 8t..8t+7, so a
// value read (and converted) feeds eight packed FMAs instead of four: per group of four steps 1 LDS read, 8 converts, 32 packed
// FMAs — 12 % fewer vector instructions per output.  Eight outputs per thread double the span a workgroup stages (60 KB for
// 1024 outputs: two workgroups per CU); to keep eight waves on the CU the workgroup has TWO halves of FIR8_T threads that share
// the staged span and split the D phases between them (half 0: phases 0..ceil(D/2)-1), and half 1 hands its partial sums to half
// 0 through LDS at the end.  Used when D >= 2 and the step count fits the unrolled instantiations; TWX_FIR_K=4 forces the old form.
constexpr int FIR8_T = 128, FIR8_K = 8, FIR8_NT = 2 * FIR8_T, FIR8_OUT = FIR8_T * FIR8_K;
struct FirGeom { int K, A, SH, PSQ, HROW, LASTN; size_t lds; };
FirGeom fir_geom(int ntaps, int dec) {
    FirGeom g;
    g.A = (ntaps + dec - 1) / dec;
    static const int force_k = [] { const char* e = getenv("TWX_FIR_K"); return e ? atoi(e) : 0; }();
    const int sh8 = std::max(4, (g.A + FIR8_K - 1 + 3) / 4);
    g.K = (dec >= 2 && sh8 <= 16 && force_k != 4) ? FIR8_K : FIR_K;
    g.SH = (g.A + g.K - 1 + 3) / 4;              // groups of four steps s = 0 .. A+K-2
    g.LASTN = g.A + g.K - 1 - 4 * (g.SH - 1);    // steps of the last group that can meet a tap (1..4)
    if (g.SH < 4) { g.SH = 4; g.LASTN = 4; }                  // counts 4..16 have unrolled kernels
    const int nq = g.K * (g.K == FIR8_K ? FIR8_T : FIR_NT) + 4 * g.SH;   // q values staged per phase
    g.PSQ = ((nq / 4) & 1) ? nq : nq + 4;        // a multiple of 4 (16-byte reads) with PSQ/4 odd (staging writes)
    g.HROW = 4 * g.SH + g.K;                     // taps per phase incl. padding (K-1 zeros in front; K+3 are read per group of steps)
    g.lds = (size_t)dec * g.PSQ * sizeof(unsigned);
    return g;
}

// Staging of a workgroup's input span into LDS, phase-major (NTH threads).
template <int NTH>
__device__ __forceinline__ void fir_stage(unsigned* __restrict__ X, const short2* __restrict__ x, int nch, long long nin, long long e0, int span, int D,
                                          int PSQ, int tid, int ch) {
    // sample e = q*D + p of the span goes to slot(p, q) = p*PSQ + q: along e the slot advances by PSQ, and by 1 - (D-1)*PSQ
    // where p wraps — no division and no multiplication per sample.
    // Loads are UNCONDITIONAL (clamped indices): a load inside a divergent branch gets its own s_waitcnt vmcnt(0), which
    // turns "eight loads in flight" into eight round trips.
    const int nvec = span >> 2;
    const int wrap = D * PSQ - 1;
    if (nch == 1 && ((reinterpret_cast<unsigned long long>(x + e0) & 15ull) == 0) && e0 + 4ll * nvec <= nin) {
        // interior workgroup, one channel, aligned: 16-B loads, four samples per lane
        const int4* xv = reinterpret_cast<const int4*>(x + e0);
        const int dq4 = (4 * NTH) / D, dp4 = 4 * NTH - dq4 * D;
        int q = (4 * tid) / D, p = 4 * tid - q * D;
        for (int jb = 0; jb < nvec; jb += 8 * NTH) {
            int4 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = xv[min(jb + u * NTH + tid, nvec - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jb + u * NTH + tid;
                const int w4[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
                int slot = p * PSQ + q, pp = p;
                if (j < nvec) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        X[slot] = (unsigned)w4[i];
                        slot += PSQ;
                        if (++pp >= D) { pp = 0; slot -= wrap; }
                    }
                }
                q += dq4; p += dp4;
                if (p >= D) { p -= D; ++q; }
            }
        }
    } else if (nch == 2 && ((reinterpret_cast<unsigned long long>(x - ch + 2 * e0) & 15ull) == 0) && e0 + 4ll * nvec <= nin) {
        // interior workgroup, a channel of a TWO-channel capture ([a0 b0 a1 b1 ...]; x points at this channel's sample of frame 0,
        // x - ch at the frame): 16-B loads of two frames, the channel's two samples picked out — with 4-byte loads of every second
        // word this path ran the kernel 60 % slower than the one-channel one (0.199 against 0.124 ms at 70 Msps, tools/fir_nch.py)
        const int4* xv = reinterpret_cast<const int4*>(x - ch + 2 * e0);
        const int nvec2 = span >> 1;
        const int dq2 = (2 * NTH) / D, dp2 = 2 * NTH - dq2 * D;
        int q = (2 * tid) / D, p = 2 * tid - q * D;
        for (int jb = 0; jb < nvec2; jb += 8 * NTH) {
            int4 raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) raw[u] = xv[min(jb + u * NTH + tid, nvec2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = jb + u * NTH + tid;
                const int w2[2] = {ch ? raw[u].y : raw[u].x, ch ? raw[u].w : raw[u].z};
                int slot = p * PSQ + q, pp = p;
                if (j < nvec2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        X[slot] = (unsigned)w2[i];
                        slot += PSQ;
                        if (++pp >= D) { pp = 0; slot -= wrap; }
                    }
                }
                q += dq2; p += dp2;
                if (p >= D) { p -= D; ++q; }
            }
        }
    } else {
        const int dq = NTH / D, dp = NTH - dq * D;
        int q = tid / D, p = tid - q * D;
        const unsigned* xs = reinterpret_cast<const unsigned*>(x);
        for (int eb = 0; eb < span; eb += 8 * NTH) {
            unsigned raw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long g = e0 + eb + u * NTH + tid;
                const unsigned v = xs[min(g, nin - 1) * nch];
                raw[u] = g < nin ? v : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = eb + u * NTH + tid;
                if (e < span) X[p * PSQ + q] = raw[u];
                q += dq; p += dp;
                if (p >= D) { p -= D; ++q; }
            }
        }
    }
}

// SHT > 0: the number of 4-step groups is a compile-time constant, so a whole phase is one straight-line block: all
// its taps are fetched with a few scalar loads up front (they then sit in SGPRs) and the LDS reads run ahead of the
// FMAs.  SHT == 0: generic loop for any tap count.
template <int SHT>
__global__ __launch_bounds__(FIR_NT) void k_fir_poly(const short2* __restrict__ x, int nch, long long nin, const float* __restrict__ hp,
                                                     int D, int SH_rt, int PSQ, int HROW, int lastn, long long nout,
                                                     short2* __restrict__ y16, float2* __restrict__ yf, int ch) {
    extern __shared__ uint4 X4[];                                   // raw int16 IQ pairs: 4 B per sample, converted when read
    unsigned* X = reinterpret_cast<unsigned*>(X4);
    const int SH = SHT > 0 ? SHT : SH_rt;
    const int tid = threadIdx.x;
    const long long m0 = (long long)blockIdx.x * FIR_OUT;
    const long long e0 = m0 * D;
    const int nq = FIR_K * (FIR_NT + SH);                          // q values staged per phase
    const int span = nq * D;                                       // a multiple of 4
    fir_stage<FIR_NT>(X, x, nch, nin, e0, span, D, PSQ, tid, ch);
    __syncthreads();
    // (I, Q) of an output as one 2-vector: one v_pk_fma_f32 per tap and output with the wave-uniform tap broadcast from its
    // scalar register.  Measured equal to the two v_fmac_f32 with a scalar operand it replaces (0.187 ms both: a packed fp32
    // instruction takes two passes, profiles/r02_valu_probe.txt): the kernel is bound by vector issue slots — per four LDS
    // reads 8 sign-extending converts and 16 packed FMAs, of which 86 % meet a real tap.
    typedef float pk2 __attribute__((ext_vector_type(2)));
    pk2 acc[FIR_K];
#pragma unroll
    for (int k = 0; k < FIR_K; ++k) acc[k] = pk2{0.f, 0.f};
    auto cvt = [](unsigned w) { return pk2{(float)(short)(w & 0xffffu), (float)(short)(w >> 16)}; };
    for (int p = 0; p < D; ++p) {
        const uint4* Xp = X4 + ((p * PSQ) >> 2) + tid;             // PSQ is a multiple of 4
        const float* h = hp + p * HROW;
        if constexpr (SHT > 0) {
            float t[4 * SHT + 4];
#pragma unroll
            for (int j = 0; j < 4 * SHT + 4; ++j) t[j] = h[j];     // wave-uniform: scalar loads, once per phase
#pragma unroll
            for (int sh = 0; sh < SHT; ++sh) {
                const uint4 w = Xp[sh];
                const unsigned wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    if (sh == SHT - 1 && sl >= lastn) break;        // only padding from here on (uniform)
                    const pk2 v = cvt(wv[sl]);
#pragma unroll
                    for (int k = 0; k < FIR_K; ++k) {              // step s = 4sh+sl feeds output k with tap a = s-k (table index a+3)
                        if (4 * sh + sl - k < 0) continue;         // the table's leading zeros (compile-time: six FMAs per phase)
                        const float tv = t[4 * sh + sl - k + 3];
                        acc[k] = __builtin_elementwise_fma(v, pk2{tv, tv}, acc[k]);
                    }
                }
            }
        } else {
            for (int sh = 0; sh < SH; ++sh) {
                float t[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) t[j] = h[4 * sh + j];
                const uint4 w = Xp[sh];
                const unsigned wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    const pk2 v = cvt(wv[sl]);
#pragma unroll
                    for (int k = 0; k < FIR_K; ++k) {
                        const float tv = t[sl - k + 3];
                        acc[k] = __builtin_elementwise_fma(v, pk2{tv, tv}, acc[k]);
                    }
                }
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
            if (yf) yf[m + k] = make_float2(acc[k].x, acc[k].y);
            if (y16) {
                const float r = fminf(fmaxf(rintf(acc[k].x), -32768.f), 32767.f), q = fminf(fmaxf(rintf(acc[k].y), -32768.f), 32767.f);
                y16[m + k] = make_short2((short)r, (short)q);
            }
        }
    }
}

// The eight-output form (see FIR8_* above).  Same staging, same table (7 zeros in front), same arithmetic per output except that
// the phases are summed as (0..D0-1) + (D0..D-1) instead of one after the other.  The step count AND the count of live steps of
// the last group are compile-time here, i.e. the taps per phase A = 4*SHT + LASTN - 11 are: the FMAs that would meet the table's
// zeros behind the taps (28 of 304 per phase at K = 8) are not emitted, and the phases that hold one tap less (p >= nfull: ntaps
// is rarely a multiple of D) run a body of their own with A-1 taps.
template <int SHT, int LASTN>
__global__ __launch_bounds__(FIR8_NT) void k_fir_poly8(const short2* __restrict__ x, int nch, long long nin, const float* __restrict__ hp,
                                                       int D, int PSQ, int HROW, int nfull, long long nout,
                                                       short2* __restrict__ y16, float2* __restrict__ yf, int ch) {
    extern __shared__ uint4 X4[];
    unsigned* X = reinterpret_cast<unsigned*>(X4);
    constexpr int K = FIR8_K, A = 4 * SHT + LASTN - (K + 3);
    static_assert(A >= 2, "at least two taps per phase");
    const int tid = threadIdx.x, t = tid & (FIR8_T - 1);
    const int half = __builtin_amdgcn_readfirstlane(tid / FIR8_T);  // wave-uniform: the tap loads below stay scalar
    const long long m0 = (long long)blockIdx.x * FIR8_OUT;
    const long long e0 = m0 * D;
    const int span = (FIR8_OUT + 4 * SHT) * D;                      // a multiple of 4
    fir_stage<FIR8_NT>(X, x, nch, nin, e0, span, D, PSQ, tid, ch);
    __syncthreads();
    typedef float pk2 __attribute__((ext_vector_type(2)));
    pk2 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = pk2{0.f, 0.f};
    auto cvt = [](unsigned w) { return pk2{(float)(short)(w & 0xffffu), (float)(short)(w >> 16)}; };
    auto phase = [&](auto at, int p) {
        constexpr int AT = decltype(at)::value;                     // taps of this phase
        const uint4* Xp = X4 + ((p * PSQ) >> 2) + 2 * t;            // lane t: x_p[8t + 4sh .. +3]
        const float* h = hp + p * HROW + (K - 1);
        float tp[AT];
#pragma unroll
        for (int j = 0; j < AT; ++j) tp[j] = h[j];                  // wave-uniform: scalar loads, once per phase
#pragma unroll
        for (int sh = 0; sh < SHT; ++sh) {
            if (4 * sh > AT + K - 2) break;
            const uint4 w = Xp[sh];
            const unsigned wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const int s = 4 * sh + sl;                          // step s feeds output k with tap a = s-k
                if (s > AT + K - 2) break;
                const pk2 v = cvt(wv[sl]);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (s - k < 0 || s - k > AT - 1) continue;
                    const float tv = tp[s - k];
                    acc[k] = __builtin_elementwise_fma(v, pk2{tv, tv}, acc[k]);
                }
            }
        }
    };
    const int D0 = (D + 1) >> 1;
    const int p_lo = half ? D0 : 0, p_hi = half ? D : D0;
    for (int p = p_lo; p < p_hi; ++p) {
        if (p < nfull) phase(std::integral_constant<int, A>{}, p);
        else phase(std::integral_constant<int, A - 1>{}, p);
    }
    __syncthreads();                                                // every read of the staged span is done: its first 8 KB carry half 1's sums
    float4* S = reinterpret_cast<float4*>(X4);
    if (half) {
#pragma unroll
        for (int k = 0; k < K; k += 2) S[(k >> 1) * FIR8_T + t] = make_float4(acc[k].x, acc[k].y, acc[k + 1].x, acc[k + 1].y);
    }
    __syncthreads();
    if (half) return;
#pragma unroll
    for (int k = 0; k < K; k += 2) {
        const float4 o = S[(k >> 1) * FIR8_T + t];
        acc[k] += pk2{o.x, o.y};
        acc[k + 1] += pk2{o.z, o.w};
    }
    const long long m = m0 + (long long)K * t;
    auto pack = [](pk2 a) {
        const float r = fminf(fmaxf(rintf(a.x), -32768.f), 32767.f), q = fminf(fmaxf(rintf(a.y), -32768.f), 32767.f);
        return ((unsigned)(unsigned short)(short)r) | ((unsigned)(unsigned short)(short)q << 16);
    };
    if (m + K <= nout) {
        if (yf) {
            float4* o = reinterpret_cast<float4*>(yf + m);
#pragma unroll
            for (int k = 0; k < K; k += 2) o[k >> 1] = make_float4(acc[k].x, acc[k].y, acc[k + 1].x, acc[k + 1].y);
        }
        if (y16) {
            uint4* o = reinterpret_cast<uint4*>(y16 + m);
            o[0] = make_uint4(pack(acc[0]), pack(acc[1]), pack(acc[2]), pack(acc[3]));
            o[1] = make_uint4(pack(acc[4]), pack(acc[5]), pack(acc[6]), pack(acc[7]));
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (m + k >= nout) break;
            if (yf) yf[m + k] = make_float2(acc[k].x, acc[k].y);
            if (y16) { const unsigned w = pack(acc[k]); y16[m + k] = make_short2((short)(w & 0xffffu), (short)(w >> 16)); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_fir_mfma (round 5): the decimating FIR on the fp16 matrix cores at fp32 accuracy.
//
// v_pk_fma_f32 ends at half the fp32 "peak" (two passes per instruction: k_fir_poly8 sits at 0.89 of what the vector unit can
// issue), the fp32 MFMA forms run at that same rate, and a banded Toeplitz operand wastes a third of them.  The fp16 forms are 16 x
// faster, and both operands of THIS product split into fp16 pieces without loss:
//   * a sample is an int16: x = 256 xh + xl with xh in -128..127 and xl in 0..255 — both exact in fp16;
//   * a tap (fp32) times 2^s (s: max |h| 2^s in [64, 128)) is h1 + h2 with two fp16 values to 2^-22 of the tap;
//   * fp16 x fp16 products are exact in the fp32 the matrix cores accumulate in.
// So  y[m] = 2^-s sum_j (h1 + h2)[j] (256 xh + xl)[m D + j]  is four fp16 products per tap, and a K = 32 step of
// v_mfma_f32_16x16x32_f16 takes 16 samples as (xh, xl) pairs against the column (256 h, h): the recombination of the sample happens
// inside the dot product.  Per phase p (x_p[q] = x[q D + p], h_p[a] = taps[a D + p]) the filter is the banded Toeplitz product
//   D[i][n] = sum_k A[i][k] B[k][n],   A[i][k] = h_p[k - i],   B[k][n] = x_p[m0 + 16 n + k]      (outputs m0 + 16 n + i)
// with KS = ceil((taps per phase + 15) / 16) steps of 16 samples.  The A fragments (taps, Toeplitz-shifted per lane, both pieces) are
// built on the host and live in registers for the life of the workgroup.  Error against the fp64 direct sum: that of an fp32
// accumulation over 4 x ntaps terms (simulated and measured: about half of the 2e-6 gate at 421 taps).
//
// Workgroup = 8 waves, one tile of 256 outputs per trip, persistent over its trips.  Wave w takes the (phase, step) pairs w, w + 8,
// ...; the eight partial sums meet in LDS.  The span of a trip is staged RAW (the int16 pair of a sample, one word), phase-major, rows
// padded by 16 bytes per 64 samples (the 16 lanes of a fragment read hit 16 different bank quads); a lane's B operand is ONE 16-byte
// read — four samples, both components — split into the (xh, xl) fragments of I and of Q by sixteen SDWA converts (vector work that
// runs beside the matrix cores; converted at staging time the span would take twice the LDS).  Two spans: trip i+1 is staged (from
// registers: its 16-byte loads were issued a trip earlier) while trip i is multiplied — two barriers per trip.
// ---------------------------------------------------------------------------------------------
#ifndef TWX_FM_ZERO_LDS
#define TWX_FM_ZERO_LDS 0
#endif
#ifndef TWX_FM_ABL
#define TWX_FM_ABL 0     // timing-only ablations of k_fir_mfma: 1 no matrix-core loop, 2 no staging after the first trips, 3 no reduction / stores
#endif
using namespace twx_fm;                 // FM_* constants, fm_phys, FirMfmaGeom, fir_mfma_geom, fir_mfma_table (twx_fir_table.h)
typedef _Float16 fm_h8 __attribute__((ext_vector_type(8)));
typedef float fm_f4 __attribute__((ext_vector_type(4)));
// one sample word (I | Q << 16, int16 each) -> its (xh, xl) fp16 pairs: four SDWA converts — the high byte of a component,
// sign-extended, is xh; the low byte, unsigned, xl — each written straight into its half of the word
__device__ __forceinline__ void fm_split(unsigned w, unsigned& wi, unsigned& wq) {
#if defined(TWX_FM_NOSDWA)
    // diagnostic: the same values without SDWA instructions (bit-field extracts, int -> f32 -> f16, pack)
    int ih, il, qh, ql; float fih, fil, fqh, fql; unsigned hih, hil, hqh, hql;
    asm volatile("v_bfe_i32 %0, %1, 8, 8" : "=v"(ih) : "v"(w));   asm volatile("v_bfe_u32 %0, %1, 0, 8" : "=v"(il) : "v"(w));
    asm volatile("v_ashrrev_i32 %0, 24, %1" : "=v"(qh) : "v"(w)); asm volatile("v_bfe_u32 %0, %1, 16, 8" : "=v"(ql) : "v"(w));
    asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fih) : "v"(ih)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fil) : "v"(il));
    asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fqh) : "v"(qh)); asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(fql) : "v"(ql));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hih) : "v"(fih)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hil) : "v"(fil));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hqh) : "v"(fqh)); asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hql) : "v"(fql));
    asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(wi) : "v"(hih), "v"(hil));
    asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(wq) : "v"(hqh), "v"(hql));
    return;
#endif
    asm volatile("v_cvt_f16_i16_sdwa %0, sext(%1) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(wi) : "v"(w));
    asm volatile("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(wi) : "v"(w));
    asm volatile("v_cvt_f16_i16_sdwa %0, sext(%1) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(wq) : "v"(w));
    asm volatile("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(wq) : "v"(w));
}
// four consecutive samples (one 16-byte vector, first sample at phase p of group q) into their slots of a span
__device__ __forceinline__ void fm_put4(unsigned* __restrict__ X, int4 v, int p, int q, int D, int PS) {
    const int w4[4] = {v.x, v.y, v.z, v.w};
    int slot = p * PS + fm_phys(q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        X[slot] = (unsigned)w4[i];
        if (++p >= D) { p = 0; ++q; slot += 1 - (D - 1) * PS + ((q & 63) == 0 ? 4 : 0); }       // next group: one word on, a pad every 64
        else slot += PS;
    }
}
// the general staging (a channel of a multi-channel capture, an unaligned base): 4-byte loads, unconditional (clamped), zeros past the end
__device__ __forceinline__ void fm_stage(unsigned* __restrict__ X, const short2* __restrict__ x, int nch, long long nin, long long e0, int span, int D,
                                         int PS, int tid) {
    const int dq = FM_NT / D, dp = FM_NT - dq * D;
    int q = tid / D, p = tid - q * D;
    const unsigned* xs = reinterpret_cast<const unsigned*>(x);
    for (int eb = 0; eb < span; eb += 8 * FM_NT) {
        unsigned raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long ge = e0 + eb + u * FM_NT + tid;
            const unsigned v = xs[min(ge, nin - 1) * nch];
            raw[u] = ge < nin ? v : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * FM_NT + tid;
            if (e < span) X[p * PS + fm_phys(q)] = raw[u];
            q += dq; p += dp;
            if (p >= D) { p -= D; ++q; }
        }
    }
}
// FAST: a one-channel capture on a 16-byte boundary — 16-byte loads, the next trip's asked for a trip ahead.  A vector index is clamped
// to the vector that holds the capture's last sample: what a clamped load returns only ever meets zero taps or outputs past the end
// (a 16-byte aligned vector that holds one valid byte lies inside that byte's page).
template <int NPW, bool FAST>
__global__ __launch_bounds__(FM_NT, 4) void k_fir_mfma(const short2* __restrict__ x, int nch, long long nin, const uint4* __restrict__ atab, int D, int KS,
                                                       int PS, float inv_scale, long long nout, int ntrips, short2* __restrict__ y16,
                                                       float2* __restrict__ yf) {
    extern __shared__ uint4 XM4[];
    unsigned* X0 = reinterpret_cast<unsigned*>(XM4);
    const int span_words = D * PS;
    float* S = reinterpret_cast<float*>(X0 + 2 * span_words);
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 15, g = l >> 4;
    const int NQ = FM_OUT + 16 * KS;
    // this wave's (phase, step) pairs: A fragments (both pieces) and the lane's word offset of its B operand
    uint4 af[NPW][2];
    int off[NPW];
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int u = w + 8 * j;
        af[j][0] = atab[((size_t)u * 2 + 0) * 64 + l];
        af[j][1] = atab[((size_t)u * 2 + 1) * 64 + l];
        const bool live = u < D * KS;
        const int p = live ? u / KS : 0, ks = live ? u - p * KS : 0;
        off[j] = p * PS + fm_phys(16 * n + 16 * ks + 4 * g);
    }
    const int span = NQ * D, nvec = span >> 2;                                   // (fir_mfma_geom: span <= FM_NV * 4 * FM_NT)
    const long long last_vec = (nin - 1) >> 2;
    const int dq4 = (4 * FM_NT) / D, dp4 = 4 * FM_NT - dq4 * D;
    const int q_first = (4 * tid) / D, p_first = 4 * tid - q_first * D;          // (group, phase) of the thread's first vector
    int4 nx[FAST ? FM_NV : 1];
    auto ask = [&](int trip) {
        const long long v0 = ((long long)trip * FM_OUT * D) >> 2;                // FM_OUT * D is a multiple of 4
        const int4* xv = reinterpret_cast<const int4*>(x);
#pragma unroll
        for (int u = 0; u < FM_NV; ++u) nx[u] = xv[min(v0 + min(u * FM_NT + tid, nvec - 1), last_vec)];
    };
    auto stage = [&](unsigned* X, int trip) {
        if constexpr (FAST) {
            int lt = tid, q = q_first, p = p_first;
            asm volatile("" : "+v"(lt), "+v"(q), "+v"(p));  // slot addresses per trip: hoisted out of the trip loop they hold a register each
#pragma unroll
            for (int u = 0; u < FM_NV; ++u) {
                if (u * FM_NT + lt < nvec) fm_put4(X, nx[u], p, q, D, PS);
                q += dq4; p += dp4;
                if (p >= D) { p -= D; ++q; }
            }
        } else fm_stage(X, x, nch, nin, (long long)trip * FM_OUT * D, span, D, PS, tid);
    };
    int trip = blockIdx.x;
    if (trip < ntrips) {
        if constexpr (FAST) ask(trip);
        stage(X0, trip);
        if constexpr (FAST) { if (trip + (int)gridDim.x < ntrips) ask(trip + (int)gridDim.x); }
    }
    __syncthreads();
    int b = 0;
    for (; trip < ntrips; trip += gridDim.x, b ^= 1) {
        const long long m0 = (long long)trip * FM_OUT;
        const unsigned* X = X0 + b * span_words;
        const int nt = trip + (int)gridDim.x;
        fm_f4 aI = {0.f, 0.f, 0.f, 0.f}, aQ = aI;
        // the next trip's span into the other buffer (its last readers passed the previous trip's first barrier), the one after asked for
        auto stage_next = [&]() {
            if (nt < ntrips && !(TWX_FM_ABL == 2)) {
                stage(X0 + (b ^ 1) * span_words, nt);
                if constexpr (FAST) { if (nt + (int)gridDim.x < ntrips) ask(nt + (int)gridDim.x); }
            }
        };
        auto multiply = [&]() {
#pragma unroll
            for (int j = 0; j < (TWX_FM_ABL == 1 ? 0 : NPW); ++j) {
                const uint4 raw = *reinterpret_cast<const uint4*>(X + off[j]);      // four samples of the step, I and Q
                uint4 bi, bq;
                if (TWX_FM_ABL == 5) { bi = raw; bq = raw; }          // no SDWA converts
                else { fm_split(raw.x, bi.x, bq.x); fm_split(raw.y, bi.y, bq.y); fm_split(raw.z, bi.z, bq.z); fm_split(raw.w, bi.w, bq.w); }
                const fm_h8 bI = __builtin_bit_cast(fm_h8, bi), bQ = __builtin_bit_cast(fm_h8, bq);
                const fm_h8 a1 = __builtin_bit_cast(fm_h8, af[j][0]), a2 = __builtin_bit_cast(fm_h8, af[j][1]);
                if (TWX_FM_ABL == 4) {                                  // no matrix-core instructions
                    aI[0] += (float)bI[0] * (float)a1[0]; aQ[0] += (float)bQ[1] * (float)a2[1];
                } else {
                aI = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bI, aI, 0, 0, 0);
                aQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bQ, aQ, 0, 0, 0);
                aI = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bI, aI, 0, 0, 0);
                aQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bQ, aQ, 0, 0, 0);
                }
#if defined(TWX_FM_THROTTLE)
                // diagnostic: idle issue slots after every group of four matrix-core instructions
                for (int z = 0; z < TWX_FM_THROTTLE; ++z) __builtin_amdgcn_s_sleep(1);
#endif
            }
        };
        // (the two jobs of a trip are independent; odd waves multiplying first and staging afterwards, so that every SIMD has vector
        // work and matrix-core work side by side, measured the same: 0.100-0.105 ms either way)
        stage_next();
        __builtin_amdgcn_sched_barrier(0);
        multiply();
        __syncthreads();                                   // this span is read, the next one is staged, the previous trip's sums are read
        if (TWX_FM_ABL == 3) { asm volatile("" :: "v"(aI), "v"(aQ)); continue; }
        {
            // S[wave][c * 4 + r][lane]: D[row 4 g + r][col n] of component c (planes of FM_SP = 68 floats: the finishing threads of a
            // wave — four columns n, sixteen rows 4 g + r — read banks n + 16 g + 4 r, all different)
            float* sw = S + (w * 8) * FM_SP + l;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sw[r * FM_SP] = aI[r]; sw[(4 + r) * FM_SP] = aQ[r]; }
        }
        __syncthreads();
        if (tid < FM_OUT) {
            // thread tid finishes output m0 + tid: block n' = tid >> 4, row i = tid & 15 = 4 g' + r
            const int nn = tid >> 4, i = tid & 15, gg = i >> 2, r = i & 3, ll = nn + 16 * gg;
            float sI = 0.f, sQ = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                sI += S[(ww * 8 + r) * FM_SP + ll];
                sQ += S[(ww * 8 + 4 + r) * FM_SP + ll];
            }
            sI *= inv_scale; sQ *= inv_scale;
            const long long m = m0 + tid;
            if (m < nout) {
                if (yf) yf[m] = make_float2(sI, sQ);
                if (y16) {
                    const float rI = fminf(fmaxf(rintf(sI), -32768.f), 32767.f), rQ = fminf(fmaxf(rintf(sQ), -32768.f), 32767.f);
                    y16[m] = make_short2((short)rI, (short)rQ);
                }
            }
        }
    }
#if TWX_FM_ZERO_LDS
    // diagnostic: leave the workgroup's LDS zeroed (does a kernel that follows on this CU read what this one left?)
    __syncthreads();
    for (int i = tid; i < 2 * span_words + 8 * 8 * FM_SP; i += FM_NT) X0[i] = 0u;
#endif
}
// Which form runs: the vector forms, unless TWX_FIR_MFMA=1 asks for the matrix-core form (taken wherever its geometry fits).  It is
// 20 % faster alone (0.100-0.105 against 0.124-0.131 ms per second of 70 Msps) and NOT the default: while its workgroups share CUs with
// k_rowd (the DIF/DIT row pass of a correlation running on another stream), a few rows of that pass come out wrong — whole rows k1 of
// the spectrum, 6-12 of 625 per call (tools/chain_corunner_map.py); the Stockham row pass, the column passes, a copy, a torch fp16 GEMM
// and the vector FIR beside the same chain are clean, the chain beside poisoned or busy LDS is clean, the FIR's own outputs are right
// and deterministic, and every part of the kernel removed in turn (matrix-core instructions, the SDWA converts, the staging) makes
// the effect disappear; a bare fp16 MFMA loop beside the chain reproduces it rarely (profiles/r05_fir_mfma.txt).  A diagnostic build of
// k_rowd (-DTWX_ROWD_CHECK) that runs ONE non-inlined butterfly function twice on the same register values gets two different results in
// such a wave, and only beside this kernel — also with arguments and results in registers only (sixteen v_pk_*_f32 instructions); a build
// of the library without packed-fp32 instructions is immune.  Not a race, not stale data, nothing either source shows: v_pk_*_f32 results
// of one wave go wrong while this kernel's waves are resident beside it.  So the form stays opt-in for callers that run nothing else on
// the GPU at the time.
bool fir_use_mfma(int ntaps, int dec, long long nout, int ctx_option = -1) {
    const char* fe = getenv("TWX_FIR_MFMA");          // read per call: tests switch it inside one process
    const int force = ctx_option >= 0 ? ctx_option : (fe ? atoi(fe) : -1);     // TWX_OPT_FIR_MFMA of the context wins over the environment
    if (force <= 0) return false;
    const FirMfmaGeom g = fir_mfma_geom(ntaps, dec);
    if (!g.ok) return false;
    (void)nout;
    return force == 1;
}

// phase-major tap table: hp[p][K-1 + a] = taps[a*D + p], zeros elsewhere
std::vector<float> fir_phase_table(const float* taps, int ntaps, int dec, const FirGeom& g) {
    std::vector<float> hp((size_t)dec * g.HROW, 0.f);
    for (int j = 0; j < ntaps; ++j) hp[(size_t)(j % dec) * g.HROW + (g.K - 1) + j / dec] = taps[j];
    return hp;
}
// the table a call uploads: the matrix-core form's A fragments or the vector forms' phase-major taps (FirTable::mfma says which)
struct FirTable { std::vector<float> words; bool mfma = false; float inv_scale = 1.f; };
FirTable fir_table(const float* taps, int ntaps, int dec, long long nout, int ctx_option = -1) {
    FirTable t;
    t.mfma = fir_use_mfma(ntaps, dec, nout, ctx_option);
    if (t.mfma) t.words = fir_mfma_table(taps, ntaps, dec, fir_mfma_geom(ntaps, dec), &t.inv_scale);
    else t.words = fir_phase_table(taps, ntaps, dec, fir_geom(ntaps, dec));
    return t;
}
int launch_fir_mfma(hipStream_t st, const short2* dx, int nch, long long nin, const float* tab_dev, int ntaps, int dec, float inv_scale, long long nout,
                    short2* dy16, float2* dyf, int ch) {
    const FirMfmaGeom g = fir_mfma_geom(ntaps, dec);
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    const unsigned long long dev_bit = 1ull << (cur_dev & 63);
    static std::atomic<int> ncu_of[64];
    int ncu = ncu_of[cur_dev & 63].load();
    if (ncu <= 0) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cur_dev) != hipSuccess || ncu <= 0) ncu = 256;
        ncu_of[cur_dev & 63].store(ncu);
    }
    const long long ntrips = (nout + FM_OUT - 1) / FM_OUT;
    const uint4* atab = reinterpret_cast<const uint4*>(tab_dev);
    const bool fast = nch == 1 && (reinterpret_cast<unsigned long long>(dx) & 15ull) == 0;
    static const int per_cu = [] { const char* e = getenv("TWX_FM_WGS_PER_CU"); return e ? atoi(e) : 2; }();     // experiments: 0 = one workgroup per trip
    const unsigned grid = (unsigned)(per_cu > 0 ? std::min<long long>(ntrips, (long long)per_cu * ncu) : ntrips);          // two workgroups per CU (registers), each walks its trips
    static const size_t lds_min = [] { const char* e = getenv("TWX_FM_LDS_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();     // experiments: reserve more LDS per workgroup
    const size_t lds_bytes = std::max(g.lds, lds_min);
    hipError_t attr = hipSuccess;
#define FM_GO1(NPW_, FAST_) do { static std::atomic<unsigned long long> set{0}; auto* fn = &k_fir_mfma<NPW_, FAST_>; \
        if (!(set.load() & dev_bit)) { attr = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr == hipSuccess) set.fetch_or(dev_bit); } \
        if (set.load() & dev_bit) hipLaunchKernelGGL(fn, dim3(grid), dim3(FM_NT), lds_bytes, st, dx, nch, nin, atab, dec, g.KS, g.PS, inv_scale, nout, (int)ntrips, dy16, dyf); } while (0)
#define FM_GO(NPW_) do { if (fast) FM_GO1(NPW_, true); else FM_GO1(NPW_, false); } while (0)
    switch (g.NPW) {
        case 1: FM_GO(1); break; case 2: FM_GO(2); break; case 3: FM_GO(3); break; case 4: FM_GO(4); break;
        case 5: FM_GO(5); break; case 6: FM_GO(6); break;
        default: return TWX_E_STATE;
    }
#undef FM_GO1
#undef FM_GO
    if (attr != hipSuccess) return TWX_E_HIP;
    return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
}
int launch_fir(hipStream_t st, const short2* dx, int nch, long long nin, const float* hp_dev, int ntaps, int dec, long long nout,
               short2* dy16, float2* dyf, int ch = 0, bool mfma = false, float inv_scale = 1.f) {            // dx: the channel's sample of frame 0; ch: its place in the frame
    if (mfma) {
        // never beside other work of this process on the device (twx_internal.h): ordered behind everything enqueued so far on every
        // other stream of the library, and everything enqueued later waits for it
        int cur = 0;
        (void)hipGetDevice(&cur);
        twx::FenceExclusive fence(cur, st);
        return launch_fir_mfma(st, dx, nch, nin, hp_dev, ntaps, dec, inv_scale, nout, dy16, dyf, ch);
    }
    const FirGeom g = fir_geom(ntaps, dec);
    // The dynamic-LDS limit is an attribute of the function ON A DEVICE: one bit per device (contexts on several devices and host
    // threads — twx_multi — may get here together; the attribute call is idempotent)
    static std::atomic<unsigned long long> attr_devs{0};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    const unsigned long long dev_bit = 1ull << (cur_dev & 63);
    if (!(attr_devs.load() & dev_bit)) {
        const void* fns[] = {(const void*)&k_fir_poly<0>, (const void*)&k_fir_poly<4>, (const void*)&k_fir_poly<5>, (const void*)&k_fir_poly<6>,
                             (const void*)&k_fir_poly<7>, (const void*)&k_fir_poly<8>, (const void*)&k_fir_poly<9>, (const void*)&k_fir_poly<10>,
                             (const void*)&k_fir_poly<11>, (const void*)&k_fir_poly<12>, (const void*)&k_fir_poly<13>, (const void*)&k_fir_poly<14>,
                             (const void*)&k_fir_poly<15>, (const void*)&k_fir_poly<16>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TWX_E_HIP;
        attr_devs.fetch_or(dev_bit);
    }
    if (g.K == FIR8_K) {
        const unsigned grid8 = (unsigned)((nout + FIR8_OUT - 1) / FIR8_OUT);
        const size_t lds8 = std::max<size_t>(g.lds, (size_t)FIR8_OUT * 8);
        // phases that hold all A taps: ntaps = (A-1)*D + nfull; a geometry padded up to the smallest instantiation has no short phases
        const int a_t = 4 * g.SH + (g.SH == 4 ? 4 : g.LASTN) - (FIR8_K + 3);      // taps per phase of the instantiation chosen below
        const int nfull = a_t == g.A ? ntaps - (g.A - 1) * dec : dec;
        hipError_t attr = hipSuccess;
#define FIR8_GO(SHT_, LN_) do { static std::atomic<unsigned long long> set{0}; auto* fn = &k_fir_poly8<SHT_, LN_>; \
            if (!(set.load() & dev_bit)) { attr = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); if (attr == hipSuccess) set.fetch_or(dev_bit); } \
            if (set.load() & dev_bit) hipLaunchKernelGGL(fn, dim3(grid8), dim3(FIR8_NT), lds8, st, dx, nch, nin, hp_dev, dec, g.PSQ, g.HROW, nfull, nout, dy16, dyf, ch); } while (0)
#define FIR8_LN(SHT_) switch (g.LASTN) { case 1: FIR8_GO(SHT_, 1); break; case 2: FIR8_GO(SHT_, 2); break; case 3: FIR8_GO(SHT_, 3); break; default: FIR8_GO(SHT_, 4); break; }
        switch (g.SH) {
            case 4: FIR8_GO(4, 4); break;                            // A <= 9 (padded up): one instantiation
            case 5: FIR8_LN(5); break;   case 6: FIR8_LN(6); break;   case 7: FIR8_LN(7); break;   case 8: FIR8_LN(8); break;
            case 9: FIR8_LN(9); break;   case 10: FIR8_LN(10); break; case 11: FIR8_LN(11); break; case 12: FIR8_LN(12); break;
            case 13: FIR8_LN(13); break; case 14: FIR8_LN(14); break; case 15: FIR8_LN(15); break; case 16: FIR8_LN(16); break;
            default: return TWX_E_STATE;
        }
#undef FIR8_LN
#undef FIR8_GO
        if (attr != hipSuccess) return TWX_E_HIP;
        return hipGetLastError() == hipSuccess ? TWX_OK : TWX_E_HIP;
    }
    const unsigned grid = (unsigned)((nout + FIR_OUT - 1) / FIR_OUT);
#define FIR_GO(SHT_) hipLaunchKernelGGL((k_fir_poly<SHT_>), dim3(grid), dim3(FIR_NT), g.lds, st, dx, nch, nin, hp_dev, dec, g.SH, g.PSQ, g.HROW, g.LASTN, nout, dy16, dyf, ch)
    switch (g.SH) {                                       // unrolled instantiations for the usual tap counts, generic loop otherwise
        case 4: FIR_GO(4); break;   case 5: FIR_GO(5); break;   case 6: FIR_GO(6); break;   case 7: FIR_GO(7); break;
        case 8: FIR_GO(8); break;   case 9: FIR_GO(9); break;   case 10: FIR_GO(10); break; case 11: FIR_GO(11); break;
        case 12: FIR_GO(12); break; case 13: FIR_GO(13); break; case 14: FIR_GO(14); break; case 15: FIR_GO(15); break;
        case 16: FIR_GO(16); break; default: FIR_GO(0); break;
    }
#undef FIR_GO
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

// bodies of the C entry points (std::vector, the context's scratch bookkeeping can throw: see aux_guard below)

// ---- device-resident, on the context's stream, context-owned work buffers ------------------------------------
static int twx_sliding_dot_dev_impl(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                        int32_t ncodes, int32_t nlag, const float* replica_dev, double ff, double phi, double scale, double* out_dev) {
    if (!ctx) return TWX_E_ARG;
    if (!sliding_args_ok(iq_dev, replica_dev, out_dev, n_channels, channel, nobs, ncodes, nlag, pt, n_samples)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_sliding_dot_dev: bad argument");
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    double* dpart = static_cast<double*>(twx::ctx_scratch(ctx, 0, sliding_part_bytes(nobs, ncodes, nlag)));
    if (!dpart) return TWX_E_NOMEM;
    const int rc = launch_sliding(twx::ctx_stream(ctx), reinterpret_cast<const short2*>(iq_dev) + channel, n_channels, pt, nobs, ncodes, nlag,
                                  replica_dev, ff, phi, scale, dpart, out_dev, channel);
    return rc ? twx::ctx_fail(ctx, rc, "twx_sliding_dot_dev: launch failed") : TWX_OK;
}

static int twx_sliding_dot_cdev_impl(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t pt, int64_t nobs, int32_t ncodes, int32_t nlag,
                                     const float* replica_dev, double ff, double phi, double scale, double* out_dev) {
    if (!ctx) return TWX_E_ARG;
    if (!sliding_args_ok(smp_dev, replica_dev, out_dev, 1, 0, nobs, ncodes, nlag, pt, n_samples)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_sliding_dot_cdev: bad argument");
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    double* dpart = static_cast<double*>(twx::ctx_scratch(ctx, 0, sliding_part_bytes(nobs, ncodes, nlag)));
    if (!dpart) return TWX_E_NOMEM;
    const int rc = launch_sliding(twx::ctx_stream(ctx), reinterpret_cast<const float2*>(smp_dev), 1, pt, nobs, ncodes, nlag, replica_dev, ff, phi, scale, dpart, out_dev);
    return rc ? twx::ctx_fail(ctx, rc, "twx_sliding_dot_cdev: launch failed") : TWX_OK;
}

static int twx_fir_decimate_dev_impl(twx_ctx* ctx, const void* iq_dev, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                         int32_t dec, void* out_i16_dev, void* out_f32_dev, int64_t* n_out) {
    if (!ctx) return TWX_E_ARG;
    if (!fir_args_ok(iq_dev, taps, n_out, out_i16_dev, out_f32_dev, n_channels, channel, ntaps, dec, n_in)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_fir_decimate_dev: bad argument");
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    const long long nout = (n_in - ntaps) / dec + 1;
    *n_out = nout;
    const FirTable ft = fir_table(taps, ntaps, dec, nout, twx::ctx_fir_mfma_option(ctx));
    const std::vector<float>& hp = ft.words;
    // the table lives in a context-owned buffer; the copy is ordered on the context's stream like the kernel
    float* hp_dev = static_cast<float*>(twx::ctx_scratch(ctx, 1, hp.size() * sizeof(float)));
    if (!hp_dev) return TWX_E_NOMEM;
    hipStream_t st = twx::ctx_stream(ctx);
    std::vector<unsigned char>& shadow = twx::ctx_scratch_shadow(ctx, 1);
    const size_t hp_bytes = hp.size() * sizeof(float);
    if (shadow.size() != hp_bytes || memcmp(shadow.data(), hp.data(), hp_bytes) != 0) {      // same filter as last time: nothing to upload
        if (hipMemcpyAsync(hp_dev, hp.data(), hp_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)        // hp is a local: the copy must have left it before we return
            return twx::ctx_fail(ctx, TWX_E_HIP, "twx_fir_decimate_dev: tap upload failed");
        shadow.assign(reinterpret_cast<const unsigned char*>(hp.data()), reinterpret_cast<const unsigned char*>(hp.data()) + hp_bytes);
    }
    const int rc = launch_fir(st, reinterpret_cast<const short2*>(iq_dev) + channel, n_channels, n_in, hp_dev, ntaps, dec, nout,
                              reinterpret_cast<short2*>(out_i16_dev), reinterpret_cast<float2*>(out_f32_dev), channel, ft.mfma, ft.inv_scale);
    return rc ? twx::ctx_fail(ctx, rc, "twx_fir_decimate_dev: launch failed") : TWX_OK;
}

// ---- host-pointer convenience forms (no context): upload, run on the null stream, download -------------------
static int twx_sliding_dot_impl(const int16_t* iq, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                    int32_t ncodes, int32_t nlag, const float* replica, double ff, double phi, double scale, double* out) {
    if (!sliding_args_ok(iq, replica, out, n_channels, channel, nobs, ncodes, nlag, pt, n_samples)) return TWX_E_ARG;
    DevBuf dx, dw, dpart, dout;
    const size_t out_bytes = (size_t)ncodes * (2 * nlag + 1) * 16;
    if (!dx.alloc((size_t)n_samples * n_channels * 4) || !dw.alloc((size_t)nobs * 4) || !dpart.alloc(sliding_part_bytes(nobs, ncodes, nlag)) ||
        !dout.alloc(out_bytes)) return TWX_E_NOMEM;
    if (hipMemcpy(dx.p, iq, (size_t)n_samples * n_channels * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dw.p, replica, (size_t)nobs * 4, hipMemcpyHostToDevice) != hipSuccess) return TWX_E_HIP;
    if (int rc = launch_sliding(nullptr, static_cast<const short2*>(dx.p) + channel, n_channels, pt, nobs, ncodes, nlag, static_cast<const float*>(dw.p),
                                ff, phi, scale, static_cast<double*>(dpart.p), static_cast<double*>(dout.p), channel)) return rc;
    return hipMemcpy(out, dout.p, out_bytes, hipMemcpyDeviceToHost) == hipSuccess ? TWX_OK : TWX_E_HIP;
}

static int twx_fir_decimate_impl(const int16_t* iq, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                     int32_t dec, int16_t* out_i16, float* out_f32, int64_t* n_out) {
    if (!fir_args_ok(iq, taps, n_out, out_i16, out_f32, n_channels, channel, ntaps, dec, n_in)) return TWX_E_ARG;
    const long long nout = (n_in - ntaps) / dec + 1;
    *n_out = nout;
    const FirTable ft = fir_table(taps, ntaps, dec, nout);
    const std::vector<float>& hp = ft.words;
    DevBuf dx, dh, dy16, dyf;
    if (!dx.alloc((size_t)n_in * n_channels * 4) || !dh.alloc(hp.size() * 4) || (out_i16 && !dy16.alloc((size_t)nout * 4)) ||
        (out_f32 && !dyf.alloc((size_t)nout * 8))) return TWX_E_NOMEM;
    if (hipMemcpy(dx.p, iq, (size_t)n_in * n_channels * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dh.p, hp.data(), hp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return TWX_E_HIP;
    if (int rc = launch_fir(nullptr, static_cast<const short2*>(dx.p) + channel, n_channels, n_in, static_cast<const float*>(dh.p), ntaps, dec, nout,
                            static_cast<short2*>(dy16.p), static_cast<float2*>(dyf.p), channel, ft.mfma, ft.inv_scale)) return rc;
    if (out_i16 && hipMemcpy(out_i16, dy16.p, (size_t)nout * 4, hipMemcpyDeviceToHost) != hipSuccess) return TWX_E_HIP;
    if (out_f32 && hipMemcpy(out_f32, dyf.p, (size_t)nout * 8, hipMemcpyDeviceToHost) != hipSuccess) return TWX_E_HIP;
    return TWX_OK;
}

// ---- tracking epoch (rxcomplex.cpp:620-745): host arithmetic on the handful of correlation results: twx_track_core.h --------
using twx_track::track_update_impl;

static int twx_track_epoch_dev_impl(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t nobs,
                                    int32_t bps, int32_t nlag, const float* replica_dev, double scale, twx_track_state* st, twx_track_result* out,
                                    bool complex_float = false, const twx_track_mai* mai = nullptr) {
    if (!ctx) return TWX_E_ARG;
    if (!st || !out || bps < 2 || nlag < 2 || nlag > 31 || !(st->fs > 0)) return twx::ctx_fail(ctx, TWX_E_ARG, "twx_track_epoch_dev: bad argument");
    const int ncodes = bps - 1, nl = 2 * nlag + 1;
    if (int rc = twx::ctx_set_device(ctx)) return rc;
    double* res_dev = static_cast<double*>(twx::ctx_scratch(ctx, 6, (size_t)ncodes * nl * 16));
    if (!res_dev) return TWX_E_NOMEM;
    const double ph0 = fmod((double)st->pt * st->fc / st->fs, 1.0);                      // :594
    if (int rc = complex_float ? twx_sliding_dot_cdev_impl(ctx, iq_dev, n_samples, st->pt, nobs, ncodes, nlag, replica_dev, st->fc / st->fs, ph0, scale, res_dev)
                               : twx_sliding_dot_dev_impl(ctx, iq_dev, n_samples, n_channels, channel, st->pt, nobs, ncodes, nlag, replica_dev, st->fc / st->fs, ph0, scale, res_dev)) return rc;
    std::vector<double> res((size_t)ncodes * nl * 2), cor((size_t)ncodes * nl), ph((size_t)ncodes * nl);
    hipStream_t s = twx::ctx_stream(ctx);
    if (hipMemcpyAsync(res.data(), res_dev, res.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return twx::ctx_fail(ctx, TWX_E_HIP, "twx_track_epoch_dev: D2H copy failed");
    const double two_pi = 2.0 * 3.141592653589793;
    for (size_t i = 0; i < cor.size(); ++i) {                                            // get_cor_and_phi :1063-1072
        const double re = res[2 * i], im = res[2 * i + 1];
        cor[i] = re * re + im * im;
        ph[i] = atan2(im, re) / two_pi;
    }
    const int rc = track_update_impl(cor.data(), ph.data(), bps, nlag, st, out, nobs, mai);
    return rc ? twx::ctx_fail(ctx, rc, "twx_track_epoch_dev: bad state") : TWX_OK;
}

// No exception may cross the C boundary.
template <class F> static int aux_guard(F f, twx_ctx* ctx = nullptr) noexcept {
    // the launches of this library are checked with hipGetLastError(): an error another library left behind on this thread
    // (RCCL and PyTorch probe pointers and peers and do not clear what those probes set) must not be taken for ours
    (void)hipGetLastError();
    try {
        if (ctx) { twx::FenceShared fence(twx::ctx_device(ctx), twx::ctx_stream(ctx)); return f(); }     // never beside a matrix-core FIR (twx_internal.h)
        return f();
    }
    catch (const std::bad_alloc&) { return TWX_E_NOMEM; }
    catch (...) { return TWX_E_STATE; }
}

extern "C" {
int twx_sliding_dot_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                        int32_t ncodes, int32_t nlag, const float* replica_dev, double ff, double phi, double scale, double* out_dev) {
    return aux_guard([&]() { return twx_sliding_dot_dev_impl(ctx, iq_dev, n_samples, n_channels, channel, pt, nobs, ncodes, nlag, replica_dev, ff, phi, scale, out_dev); }, ctx);
}
int twx_sliding_dot_cdev(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t pt, int64_t nobs, int32_t ncodes, int32_t nlag,
                         const float* replica_dev, double ff, double phi, double scale, double* out_dev) {
    return aux_guard([&]() { return twx_sliding_dot_cdev_impl(ctx, smp_dev, n_samples, pt, nobs, ncodes, nlag, replica_dev, ff, phi, scale, out_dev); }, ctx);
}
int twx_track_epoch_cdev(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t nobs, int32_t bps, int32_t nlag, const float* replica_dev,
                         double scale, twx_track_state* state, twx_track_result* out) {
    return aux_guard([&]() { return twx_track_epoch_dev_impl(ctx, smp_dev, n_samples, 1, 0, nobs, bps, nlag, replica_dev, scale, state, out, true); }, ctx);
}
int twx_track_epoch_cdev_mai(twx_ctx* ctx, const void* smp_dev, int64_t n_samples, int64_t nobs, int32_t bps, int32_t nlag, const float* replica_dev,
                             double scale, twx_track_state* state, twx_track_result* out, const twx_track_mai* mai) {
    return aux_guard([&]() { return twx_track_epoch_dev_impl(ctx, smp_dev, n_samples, 1, 0, nobs, bps, nlag, replica_dev, scale, state, out, true, mai); }, ctx);
}
int twx_track_update_mai(const double* cor, const double* phi, int32_t bps, int32_t nlag, int64_t nobs, twx_track_state* state, twx_track_result* out,
                         const twx_track_mai* mai) {
    return aux_guard([&]() { return track_update_impl(cor, phi, bps, nlag, state, out, nobs, mai); });
}
int twx_fir_decimate_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                         int32_t dec, void* out_i16_dev, void* out_f32_dev, int64_t* n_out) {
    return aux_guard([&]() { return twx_fir_decimate_dev_impl(ctx, iq_dev, n_in, n_channels, channel, taps, ntaps, dec, out_i16_dev, out_f32_dev, n_out); }, ctx);
}
int twx_track_update(const double* cor, const double* phi, int32_t bps, int32_t nlag, twx_track_state* state, twx_track_result* out) {
    return aux_guard([&]() { return track_update_impl(cor, phi, bps, nlag, state, out); });
}
int twx_track_epoch_dev(twx_ctx* ctx, const void* iq_dev, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t nobs,
                        int32_t bps, int32_t nlag, const float* replica_dev, double scale, twx_track_state* state, twx_track_result* out) {
    return aux_guard([&]() { return twx_track_epoch_dev_impl(ctx, iq_dev, n_samples, n_channels, channel, nobs, bps, nlag, replica_dev, scale, state, out); }, ctx);
}
int twx_sliding_dot(const int16_t* iq, int64_t n_samples, int32_t n_channels, int32_t channel, int64_t pt, int64_t nobs,
                    int32_t ncodes, int32_t nlag, const float* replica, double ff, double phi, double scale, double* out) {
    return aux_guard([&]() { return twx_sliding_dot_impl(iq, n_samples, n_channels, channel, pt, nobs, ncodes, nlag, replica, ff, phi, scale, out); });
}
int twx_fir_decimate(const int16_t* iq, int64_t n_in, int32_t n_channels, int32_t channel, const float* taps, int32_t ntaps,
                     int32_t dec, int16_t* out_i16, float* out_f32, int64_t* n_out) {
    return aux_guard([&]() { return twx_fir_decimate_impl(iq, n_in, n_channels, channel, taps, ntaps, dec, out_i16, out_f32, n_out); });
}
}  // extern "C"
