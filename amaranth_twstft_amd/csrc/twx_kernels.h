// twx_kernels.h — gfx950 kernels of the per-window correlator.
//
// One channel-window of processing/Octave/godual_ranging.m:12-49 (≡ experiments/221219_twoway/
// processing/godual_ranging.py:18-65) is three HBM round trips over N = N1*N2 samples:
//
//   k_col_fwd   int16 IQ → (x-mean)·NCO (or squared) → length-N1 column FFTs → ·W_N^{k1 n2} → A[k1][n2]
//   k_row       A[k1][:] → length-N2 FFT  (spectrum index k = k1 + N1*k2, stored [k1][k2])
//                 BAND : |.|^2 arg-max over the carrier search band           (godual_ranging.m:14-15)
//                 MID  : · conj(FFT(code)) → R=(2*Nint+1) phase-ramped inverse FFTs → ·W_N^{-k1 q2} → Bz[rho][k1][q2]
//                 STORE: spectrum out (code spectrum at context creation, FFT test entry point)
//   k_col_inv   Bz[rho][:][q2] → length-N1 inverse FFTs → |z|^2 arg-max with index R*(q1*N2+q2)+rho
//
// The zero-padded 3N-point inverse FFT of godual_ranging.m:27-28 is evaluated as R interleaved
// N-point inverse FFTs of the spectrum times exp(2*pi*i*k*rho/(R*N)) (polyphase form, exact),
// so no 3N transform and no radix-3 pass exist; k_peak re-evaluates the samples around the peak
// in fp64 from Bz and derives the code-wipe-off SNR (godual_ranging.m:38-48) from them
// (identity in DESIGN.md §SNR; tests/test_oracle_golden.py::test_snr_identity_used_by_device_path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "twx_fft.h"
#include "../../include/twstft_hip.h"

// build-time switches (defaults = product build; the non-zero ablation values are timing-only
// diagnostic builds whose results are wrong)
#ifndef TWX_ABL
#define TWX_ABL 0       // k_row<MID>: 1 no Bz stores, 2 no code-spectrum load, 3 one phase only, 4 no A load, 5 no inverse transforms
#endif
#ifndef TWX_ABLC
#define TWX_ABLC 0      // k_col_inv: 1 no transforms, 2 no global loads
#endif
#ifndef TWX_ABLR
#define TWX_ABLR 0      // k_rowd<MID> (diagnostic variants): 1 no Bz stores (loads + arithmetic), 2 no arithmetic (row in, nphase copies out)
#endif
#ifndef TWX_ABLF
#define TWX_ABLF 0      // k_col_fwd: 1 no transform, 2 no loads, 3 no stores, 4 stores only, 5 loads only
#endif

#ifndef TWX_NT_A
#define TWX_NT_A 1      // k_rowd: non-temporal loads of the column-pass output A (read exactly once)
#endif
#ifndef TWX_MID_PERSIST
#define TWX_MID_PERSIST 1  // k_rowd<MID>: resident workgroups loop over the rows; the next row's A is loaded during the last phase
#endif
#ifndef TWX_MID_FOLD
#define TWX_MID_FOLD 1   // k_rowd<MID>: output twiddle W_N^{-k1 t} and phase ramp folded into stage B's factors (one product per output less)
#endif
#ifndef TWX_MID_SGPR
#define TWX_MID_SGPR 1   // k_rowd<MID>: wave-uniform factors (stage C's W_N^{-k1 c M}, the ramp's q2 part) through scalar loads, not LDS reads
#endif
#ifndef TWX_MID_ST16
#define TWX_MID_ST16 0  // k_rowd<MID>: Bz stores as 16-byte stores (lane pairs swap half of their outputs through DPP); A/B: profiles/r04_rowd_st16.txt
#endif
#ifndef TWX_NT_INV
#define TWX_NT_INV 1    // k_col_inv3: non-temporal loads of Bz (read exactly once)
#endif
#ifndef TWX_NT_BZ
#define TWX_NT_BZ 1     // k_rowd<MID>: non-temporal stores of Bz (written once, read once by k_col_inv 1 GB later)
#endif
#ifndef TWX_MID_PERSIST64
#define TWX_MID_PERSIST64 1  // k_rowd<MID> for complex double in the row-walking form as well (round 5: one 128-KB row per CU, so the wait for the
                             // next row was fully exposed; 0.461 -> 0.392 ms per 4 windows, profiles/r05_f64_rowwalk.txt); 0 = one row per workgroup
#endif
#ifndef TWX_FWD3_LDS_SQ
#define TWX_FWD3_LDS_SQ 3   // k_col_fwd3<SQUARE>: table values of the last stage through LDS, asked for ahead of the samples: bit 0 stage twiddles, bit 1 output twiddle
                            // (round 6, profiles/r06_colinv.txt: 0.126 -> 0.098 ms per 8 windows with both; 0.108 with the output twiddle alone)
#endif
#ifndef TWX_FWD3_LDS_MIX
#define TWX_FWD3_LDS_MIX 2  // the same for k_col_fwd3<MIX>: 0.124 -> 0.112 ms with the output twiddle; the stage twiddles through LDS cost it 12 % (0.139), both 0.170
#endif
#ifndef TWX_INV3_TWLDS
#define TWX_INV3_TWLDS 1  // k_col_inv3: last-stage twiddles parked in LDS (round 6: 0.227 -> 0.182 ms per 8 windows, profiles/r06_colinv.txt)
#endif
#ifndef TWX_BZ16
#define TWX_BZ16 0      // EXPERIMENT (round 5, profiles/r05_bz16.txt): fp32 contexts keep Bz as fp16 pairs (4 instead of 8 bytes per element) in
                        // k_rowd<MID> (store), k_col_inv3 (load) and k_peak (load) — the upper bound of what a compact Bz can buy; the
                        // candidate / exact-mode machinery that would make the lag decision rigorous is NOT in this build
#endif

namespace twx {
#if TWX_BZ16
__device__ __forceinline__ unsigned bz16_pack(float x, float y) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)x, (_Float16)y};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float2 bz16_unpack(unsigned u) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, u);
    return make_float2((float)v.x, (float)v.y);
}
#endif

// ------------------------------------------------------------------------------------------
// device-side per-window records
// ------------------------------------------------------------------------------------------
struct WinSums {            // exact integer sums over the raw int16 window
    long long sI, sQ;
    unsigned long long sP;  // sum(I^2+Q^2)
    double fP;              // complex-double input (twx_process_complex): sum |d|^2 in fp64, valid when is_f != 0
    int is_f, pad;
};
// k_sums / k_sums_deint2: one partial per workgroup, added up by k_sums_final; no atomics and nothing to clear beforehand, so
// the grid can be sized for the chip whatever the batch
struct SumPart { long long sI, sQ; unsigned long long sP; long long pad; };
#define TWX_SUMS_MAXCHUNKS 1024
template <typename T> struct ArgPart { T val; unsigned int idx; };
template <> struct ArgPart<double> { double val; unsigned int idx; unsigned int pad; };

// XCD-aware block remap: consecutive logical ids land on the same XCD (blocks b and b+8 share
// one under round-robin dispatch; MI355X_MICROARCH.md §Workgroup dispatch). Bijective for any n.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, rem = nblk & 7u, xcd = bid & 7u, pos = bid >> 3;
    return xcd < rem ? xcd * (q + 1) + pos : rem * (q + 1) + (xcd - rem) * q + pos;
}

template <typename T> __device__ __forceinline__ T shfl_down_t(T v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ long long shfl_down_ll(long long v, int d) {
    int lo = __shfl_down((int)(v & 0xffffffffll), d, 64);
    int hi = __shfl_down((int)(v >> 32), d, 64);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// (value, index) max with "first index wins" ties — Octave max / np.argmax semantics
// (godual_ranging.m:15,29).
template <typename T> struct Best {
    T val; unsigned int idx;
    __device__ __forceinline__ void take(T v, unsigned int i) {
        const bool better = (v > val) | ((v == val) & (i < idx));   // branch-free: v_cmp + v_cndmask, no exec-mask juggling
        val = better ? v : val;
        idx = better ? i : idx;
    }
};
template <typename T> __device__ __forceinline__ Best<T> wave_best(Best<T> b) {
    for (int d = 32; d >= 1; d >>= 1) {
        T ov = __shfl_down(b.val, d, 64);
        unsigned int oi = __shfl_down(b.idx, d, 64);
        b.take(ov, oi);
    }
    return b;
}
// max of two finite values as ONE instruction (v_max_f32 / v_max_f64); `a > b ? a : b` compiles to compare + select
__device__ __forceinline__ float tmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double tmax(double a, double b) { return __builtin_fmax(a, b); }
// block-wide arg-max; result valid in thread 0. scratch: >= 2*nwaves words of T/uint
template <typename T, int NT> __device__ __forceinline__ Best<T> block_best(Best<T> b, void* scratch) {
    constexpr int NW = (NT + 63) / 64;                   // a partial last wave counts
    T* sv = reinterpret_cast<T*>(scratch);
    unsigned int* si = reinterpret_cast<unsigned int*>(sv + NW);
    b = wave_best(b);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sv[wv] = b.val; si[wv] = b.idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NW; ++w) b.take(sv[w], si[w]);
    }
    return b;
}

// Global access at (wave-uniform base, 32-bit lane byte offset).  The base is pinned into an SGPR pair with
// readfirstlane (which also stops the optimiser from folding base and lane parts into one 64-bit VALU address per
// access) and re-typed as a GLOBAL pointer, so the access compiles to the saddr form
// `global_load/store v, v_off, s[base:base+1]`.
#define TWX_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ unsigned long long sgpr_u64(unsigned long long u) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <typename V> __device__ __forceinline__ V ld_su(const void* ubase, unsigned lane_bytes) {
    const TWX_GLOBAL char* g = (const TWX_GLOBAL char*)sgpr_u64(reinterpret_cast<unsigned long long>(ubase));
    return *(const TWX_GLOBAL V*)(g + lane_bytes);
}
template <typename V> __device__ __forceinline__ void st_su(void* ubase, unsigned lane_bytes, V val) {
    TWX_GLOBAL char* g = (TWX_GLOBAL char*)sgpr_u64(reinterpret_cast<unsigned long long>(ubase));
    *(TWX_GLOBAL V*)(g + lane_bytes) = val;
}
// Load from a read-only table at a WAVE-UNIFORM index through the constant address space: compiles to s_load_* (the value
// arrives in SGPRs and is counted by lgkmcnt, so it neither occupies a VGPR nor waits behind earlier vector stores).
#define TWX_CONSTAS __attribute__((address_space(4)))
template <typename V> __device__ __forceinline__ V ld_uniform(const V* table, int idx) {
    return ((const TWX_CONSTAS V*)(unsigned long long)table)[idx];
}
// With a base that is ALREADY pinned in SGPRs (sgpr_u64 once, then scalar arithmetic on it): base + uniform_bytes + lane_bytes
template <typename V, bool NT> __device__ __forceinline__ V ld_pin(unsigned long long pinned, unsigned long long uniform_bytes, unsigned lane_bytes) {
    const TWX_GLOBAL char* g = (const TWX_GLOBAL char*)(pinned + uniform_bytes);
    if constexpr (NT) return __builtin_nontemporal_load((const TWX_GLOBAL V*)(g + lane_bytes));
    else return *(const TWX_GLOBAL V*)(g + lane_bytes);
}
template <typename V, bool NT> __device__ __forceinline__ void st_pin(unsigned long long pinned, unsigned long long uniform_bytes, unsigned lane_bytes, V val) {
    TWX_GLOBAL char* g = (TWX_GLOBAL char*)(pinned + uniform_bytes);
    if constexpr (NT) __builtin_nontemporal_store(val, (TWX_GLOBAL V*)(g + lane_bytes));
    else *(TWX_GLOBAL V*)(g + lane_bytes) = val;
}
// the same with the non-temporal hint (V must be a plain vector type)
template <typename V> __device__ __forceinline__ V ld_su_nt(const void* ubase, unsigned lane_bytes) {
    const TWX_GLOBAL char* g = (const TWX_GLOBAL char*)sgpr_u64(reinterpret_cast<unsigned long long>(ubase));
    return __builtin_nontemporal_load((const TWX_GLOBAL V*)(g + lane_bytes));
}
template <typename V> __device__ __forceinline__ void st_su_nt(void* ubase, unsigned lane_bytes, V val) {
    TWX_GLOBAL char* g = (TWX_GLOBAL char*)sgpr_u64(reinterpret_cast<unsigned long long>(ubase));
    __builtin_nontemporal_store(val, (TWX_GLOBAL V*)(g + lane_bytes));
}

// ndw dwords global -> LDS without a register in between (global_load_lds_dword: lane i of a wave writes its wave's base + 4 i); the
// caller's next vmcnt(0) + workgroup barrier make them readable
template <int NT> __device__ __forceinline__ void lds_fill_dwords(float* lds, const float* g, int ndw) {
    float* wave_base = lds + (threadIdx.x & ~63u);
    for (int u0 = 0; u0 < ndw; u0 += NT)
        if (u0 + (int)threadIdx.x < ndw)
            __builtin_amdgcn_global_load_lds((const TWX_GLOBAL void*)(g + u0 + threadIdx.x), (__attribute__((address_space(3))) void*)(wave_base + u0), 4, 0, 0);
}

// ------------------------------------------------------------------------------------------
// input loaders (sample n of the current window → complex T)
// ------------------------------------------------------------------------------------------
struct InI16 {   // interleaved int16 IQ, nch channels per sample (godual_ranging.m:76-79)
    const short2* p; int nch;
    __device__ __forceinline__ void advance(long long e) { p += e; }
    template <typename T> __device__ __forceinline__ cpx<T> load(long long n) const {
        short2 s = p[n * nch];
        return mk<T>((T)s.x, (T)s.y);
    }
    // sample (ubase + lane): wave-uniform part and 32-bit lane part kept apart (saddr-form loads)
    template <typename T> __device__ __forceinline__ cpx<T> load2(long long ubase, unsigned lane) const {
        const short2* q = p + ubase * nch;
        short2 s = q[lane * (unsigned)nch];
        return mk<T>((T)s.x, (T)s.y);
    }
    // the same sample as one raw 32-bit word: wave-uniform byte base (SGPR pair) + 32-bit lane BYTE offset, which the
    // compiler turns into `global_load_dword v, v_off, s[base]` — no per-load 64-bit VALU address arithmetic, so all
    // the loads of a butterfly can be in flight at once
    static constexpr bool has_raw = true;
    __device__ __forceinline__ unsigned load_raw(long long ubase, unsigned lane) const {
        return ld_su<unsigned>(reinterpret_cast<const char*>(p) + (unsigned long long)ubase * (unsigned long long)nch * 4ull, lane * (unsigned)nch * 4u);
    }
    template <typename T> static __device__ __forceinline__ cpx<T> unpack(unsigned w) {
        return mk<T>((T)(short)(w & 0xffffu), (T)(short)(w >> 16));
    }
};
// The velocity-compensated window (experiments/220706_TWSTFT/godual_ranging_OP_vitesse.m:40-43): the MIXED window y = (d - mean) .* lo is
// resampled linearly on the stretched axis x_q(n) = n / (1 - vitesse) + t0 before the transform.  x_q(n) - n = n c + t0 with
// c = vitesse / (1 - vitesse) stays inside (-2, 2) (twx_set_resample checks it), so sample n needs d[n + m] and d[n + m + 1], m = floor(n c + t0)
// — the neighbours' own loads — and lo[n + m] = lo[n] * rot^m with rot = exp(-2 pi i df / fs) = the NCO table entry of sample 1:
//   yi[n] = lo[n] * rot^m * ((1 - f) (d[i0] - mean) + f (d[i0 + 1] - mean) rot),   i0 = n + m,  f = n c + t0 - m
// (mix first, then interp1, as the script does).  The edge rule of :42-43 — yi(end) = yi(end-1) and yi(1) = yi(2) where the query left the
// window — is the per-window word `edge` (bit 0 / bit 1), decided on the host with the script's own expression.
struct InI16Resample {
    const short2* p; int nch;
    const double* t0; const int* edge; double c;
    __device__ __forceinline__ void advance(long long e) { p += e; }
    template <typename T> __device__ __forceinline__ cpx<T> load(long long n) const { short2 s = p[n * nch]; return mk<T>((T)s.x, (T)s.y); }
    static constexpr bool has_raw = false;
};
template <class In> struct InTraits { static constexpr bool resample = false; };
template <> struct InTraits<InI16Resample> { static constexpr bool resample = true; };
// yi[n] / lo[n] of the resampled window (see InI16Resample): ((1 - f)(d[i0] - mean) + f (d[i0+1] - mean) rot) rot^m
template <typename T> __device__ __forceinline__ cpx<T> rs_sample(const short2* p, int nch, long long n, long long N, double c, double t0, int edge, T mx, T my,
                                                                  cpx<T> rot) {
    long long ne = n;
    if (n == 0 && (edge & 1)) ne = 1;                              // yi(1) = yi(2)        (godual_ranging_OP_vitesse.m:43)
    if (n == N - 1 && (edge & 2)) ne = N - 2;                      // yi(end) = yi(end-1)  (:42)
    const double dl = fma((double)ne, c, t0);
    const double fl = floor(dl);
    const T f = (T)(dl - fl);
    const int m = (int)fl + (int)(ne - n);                         // i0 - n
    long long i0 = n + m;
    i0 = i0 < 0 ? 0 : (i0 > N - 1 ? N - 1 : i0);
    const long long i1 = i0 + 1 > N - 1 ? N - 1 : i0 + 1;         // (x_q = N-1 exactly: f = 0, the second sample is not used)
    const short2 s0 = p[i0 * nch], s1 = p[i1 * nch];
    cpx<T> d0 = mk<T>((T)s0.x - mx, (T)s0.y - my), d1 = mk<T>((T)s1.x - mx, (T)s1.y - my);
    const cpx<T> d1r = cmul(d1, rot);
    const cpx<T> z = mk<T>(d0.x + (d1r.x - d0.x) * f, d0.y + (d1r.y - d0.y) * f);
    cpx<T> pw = mk<T>(1, 0);
    const cpx<T> st = m > 0 ? rot : cconj(rot);
    for (int q = (m > 0 ? m : -m); q > 0; --q) pw = cmul(pw, st); // |m| <= 2
    return cmul(z, pw);
}
template <typename T> __device__ __attribute__((noinline)) cpx<T> rs_sample_call(const short2* p, int nch, long long n, long long N, double c, double t0, int edge,
                                                                                 T mx, T my, cpx<T> rot) {
    return rs_sample<T>(p, nch, n, N, c, t0, edge, mx, my, rot);
}
struct InChips {  // code replica: chips {0,1} held sps samples, value 2c-1 (godual_ranging.m:63-65)
    const unsigned char* p; int sps;
    __device__ __forceinline__ void advance(long long e) { p += e; }
    template <typename T> __device__ __forceinline__ cpx<T> load(long long n) const {
        return mk<T>((T)(2 * (int)p[n / sps] - 1), (T)0);
    }
    template <typename T> __device__ __forceinline__ cpx<T> load2(long long ubase, unsigned lane) const { return load<T>(ubase + lane); }
    static constexpr bool has_raw = false;
};
template <typename S> struct InCplx {  // complex float/double samples (FFT test entry)
    const cpx<S>* p;
    __device__ __forceinline__ void advance(long long e) { p += e; }
    template <typename T> __device__ __forceinline__ cpx<T> load(long long n) const {
        cpx<S> s = p[n];
        return mk<T>((T)s.x, (T)s.y);
    }
    template <typename T> __device__ __forceinline__ cpx<T> load2(long long ubase, unsigned lane) const {
        const cpx<S>* q = p + ubase;
        cpx<S> s = q[lane];
        return mk<T>((T)s.x, (T)s.y);
    }
    static constexpr bool has_raw = false;
};

// complex double samples as MATLAB/Octave hold them: d(n) = re[n*stride] + j*im[n*stride]  (stride 1: separate real
// and imaginary arrays, mxGetPr/mxGetPi; stride 2 with im = re+1: interleaved, mxGetComplexDoubles) — the `d` of
// processing(d,k), godual_ranging.m:12
struct InCplxSplit {
    const double* re; const double* im; int stride;
    __device__ __forceinline__ void advance(long long e) { re += e; im += e; }       // e in doubles
    template <typename T> __device__ __forceinline__ cpx<T> load(long long n) const {
        return mk<T>((T)re[n * stride], (T)im[n * stride]);
    }
    template <typename T> __device__ __forceinline__ cpx<T> load2(long long ubase, unsigned lane) const {
        const long long n = (ubase + lane) * stride;
        return mk<T>((T)re[n], (T)im[n]);
    }
    static constexpr bool has_raw = false;
};

// sum |d|^2 of complex-double windows, deterministic two-step reduction (no atomics): grid = (64, windows), then (windows)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sums_c64(InCplxSplit in, long long win_stride, long long n, double* __restrict__ partial) {
    in.advance((long long)blockIdx.y * win_stride);
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(n, lo + per);
    double acc = 0;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const double x = in.re[i * in.stride], y = in.im[i * in.stride];
        acc += x * x + y * y;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
template <int UNUSED = 0>
__global__ void k_sums_c64_final(const double* __restrict__ partial, int nparts, WinSums* __restrict__ sums) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a = 0;
    for (int i = 0; i < nparts; ++i) a += partial[(long long)b * nparts + i];
    WinSums s; s.sI = 0; s.sQ = 0; s.sP = 0; s.fP = a; s.is_f = 1; s.pad = 0;
    sums[b] = s;
}

// Block-wide finish of NC (channels) x {sum I, sum Q, sum I^2+Q^2}: every thread brings its own partial sums; thread 0 ends up
// with the block's totals.
template <int NC>
__device__ __forceinline__ void sums_block_total(long long (&v)[NC][3]) {
    __shared__ long long sh[NC * 3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            for (int d = 32; d >= 1; d >>= 1) v[c][k] += shfl_down_ll(v[c][k], d);
            if (lane == 0) sh[c * 3 + k][wv] = v[c][k];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[c][k] = (sh[c * 3 + k][0] + sh[c * 3 + k][1]) + (sh[c * 3 + k][2] + sh[c * 3 + k][3]);
    }
}
// Second step of k_sums / k_sums_deint2: the per-workgroup partials of a window -> its WinSums (integer sums: the order cannot
// change the result).  A launch of its own instead of atomics on the window's three words: same-address device-scope atomics
// cost ~0.4 us each on this chip (256 workgroups per window on one ticket word: 0.127 ms for what reads in 0.03,
// profiles/r04_ksums.txt), and a grid sized for the chip means hundreds of them.   grid = (windows, NC), block = 256
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sums_final(const SumPart* __restrict__ parts, int nparts, long long chan_stride /*SumPart units*/,
                                                    WinSums* __restrict__ sums0, WinSums* __restrict__ sums1) {
    const int b = blockIdx.x, c = blockIdx.y;
    const SumPart* p = parts + (long long)c * chan_stride + (long long)b * nparts;
    long long v[1][3] = {{0, 0, 0}};
    for (int i = threadIdx.x; i < nparts; i += 256) { const SumPart q = p[i]; v[0][0] += q.sI; v[0][1] += q.sQ; v[0][2] += (long long)q.sP; }
    sums_block_total<1>(v);
    if (threadIdx.x == 0) {
        WinSums s; s.sI = v[0][0]; s.sQ = v[0][1]; s.sP = (unsigned long long)v[0][2]; s.fP = 0.0; s.is_f = 0; s.pad = 0;
        (c == 0 ? sums0 : sums1)[b] = s;
    }
}

// ------------------------------------------------------------------------------------------
// k_sums: exact integer statistics of a raw int16 window (mean removal godual_ranging.m:80,
// power terms of :46).  grid = (chunks, windows)
// ------------------------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sums(const short2* __restrict__ in, long long win_stride /*short2 units*/,
                                              int nch, long long n, SumPart* __restrict__ parts /*[windows][chunks]*/) {
    const int b = blockIdx.y;
    const short2* p = in + (long long)b * win_stride;
    long long sI = 0, sQ = 0;
    unsigned long long sP = 0;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(n, lo + per);
    auto acc = [&](short2 s) {
        sI += s.x; sQ += s.y;
        // through unsigned: (-32768)^2 * 2 = 2^31 does not fit an int (a clipped capture holds such samples), like acc4 below
        sP += (unsigned long long)((unsigned int)((int)s.x * (int)s.x) + (unsigned int)((int)s.y * (int)s.y));
    };
    if (nch == 1 && ((reinterpret_cast<unsigned long long>(p) & 15) == 0)) {
        // 16 B per lane: four [I Q] samples per load (coalesced 1 KiB per wave-instruction)
        long long i = lo;
        for (; i < hi && (i & 3); ++i) if (threadIdx.x == 0) acc(p[i]);
        const long long nv = (hi - i) >> 2;
        const int4* pv = reinterpret_cast<const int4*>(p + i);
        // four independent 16-B loads in flight per thread; 32-bit partial sums per group (|I|,|Q| <= 2^15:
        // 16 samples fit), widened once per group
        auto acc4 = [&](int4 q, int& aI, int& aQ, unsigned long long& aP) {
            const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int xi = (short)(w[t] & 0xffff), xq = w[t] >> 16;
                aI += xi; aQ += xq;
                aP += (unsigned long long)(unsigned int)(xi * xi + xq * xq);
            }
        };
        long long k = threadIdx.x;
        for (; k + 7 * 256 < nv; k += 8 * 256) {
            int4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = pv[k + u * 256];
            int aI = 0, aQ = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc4(q[u], aI, aQ, sP);
            sI += aI; sQ += aQ;
        }
        for (; k < nv; k += 256) {
            int aI = 0, aQ = 0;
            acc4(pv[k], aI, aQ, sP);
            sI += aI; sQ += aQ;
        }
        for (long long t = i + (nv << 2) + threadIdx.x; t < hi; t += 256) acc(p[t]);
    } else {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) acc(p[i * nch]);
    }
    long long blk[1][3] = {{sI, sQ, (long long)sP}};
    sums_block_total<1>(blk);
    if (threadIdx.x == 0) { SumPart q; q.sI = blk[0][0]; q.sQ = blk[0][1]; q.sP = (unsigned long long)blk[0][2]; q.pad = 0; parts[(long long)b * gridDim.x + blockIdx.x] = q; }
}

// ------------------------------------------------------------------------------------------
// k_sums_deint2: two-channel captures ([I1 Q1 I2 Q2] per sample, the B210-era files of godual_ranging.m:76-79) in
// all-channel mode.  The three input passes of the chain (k_sums, 2 x k_col_fwd) would each fetch the 8-byte frames to
// use 4 bytes of them, once per channel: 24 B of traffic per channel-sample.  This pass reads the frames ONCE, takes
// the integer statistics of both channels and writes two planar [I Q] copies (8 B per channel-sample in all); the
// column passes then read 4 B each.  grid = (chunks, windows); frames must be 16-byte aligned (two per load).
// ------------------------------------------------------------------------------------------
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sums_deint2(const int4* __restrict__ in /*two frames per element*/, long long win_stride /*frames*/,
                                                     long long n, short2* __restrict__ p0, short2* __restrict__ p1,
                                                     SumPart* __restrict__ parts0, SumPart* __restrict__ parts1) {
    const int b = blockIdx.y;
    const int4* p = in + ((long long)b * win_stride >> 1);
    int2* o0 = reinterpret_cast<int2*>(p0 + (long long)b * n);
    int2* o1 = reinterpret_cast<int2*>(p1 + (long long)b * n);
    const long long nv = n >> 1;                                      // n is even (window lengths are)
    const long long per = (nv + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per, hi = min(nv, lo + per);
    long long sI[2] = {0, 0}, sQ[2] = {0, 0};
    unsigned long long sP[2] = {0, 0};
    auto take = [&](int w, int& aI, int& aQ, unsigned long long& aP) {
        const int xi = (short)(w & 0xffff), xq = w >> 16;
        aI += xi; aQ += xq;
        aP += (unsigned long long)(unsigned int)(xi * xi + xq * xq);
    };
    long long k = lo + threadIdx.x;
    for (; k + 3 * 256 < hi; k += 4 * 256) {
        typedef int i4v __attribute__((ext_vector_type(4)));            // the nontemporal builtin wants a plain vector type
        i4v q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = __builtin_nontemporal_load(reinterpret_cast<const i4v*>(p + k + u * 256));   // read exactly once
        int aI[2] = {0, 0}, aQ[2] = {0, 0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            take(q[u].x, aI[0], aQ[0], sP[0]); take(q[u].z, aI[0], aQ[0], sP[0]);
            take(q[u].y, aI[1], aQ[1], sP[1]); take(q[u].w, aI[1], aQ[1], sP[1]);
            o0[k + u * 256] = make_int2(q[u].x, q[u].z);
            o1[k + u * 256] = make_int2(q[u].y, q[u].w);
        }
        sI[0] += aI[0]; sQ[0] += aQ[0]; sI[1] += aI[1]; sQ[1] += aQ[1];
    }
    for (; k < hi; k += 256) {
        const int4 q = p[k];
        int aI[2] = {0, 0}, aQ[2] = {0, 0};
        take(q.x, aI[0], aQ[0], sP[0]); take(q.z, aI[0], aQ[0], sP[0]);
        take(q.y, aI[1], aQ[1], sP[1]); take(q.w, aI[1], aQ[1], sP[1]);
        o0[k] = make_int2(q.x, q.z);
        o1[k] = make_int2(q.y, q.w);
        sI[0] += aI[0]; sQ[0] += aQ[0]; sI[1] += aI[1]; sQ[1] += aQ[1];
    }
    long long blk[2][3] = {{sI[0], sQ[0], (long long)sP[0]}, {sI[1], sQ[1], (long long)sP[1]}};
    sums_block_total<2>(blk);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            SumPart q; q.sI = blk[c][0]; q.sQ = blk[c][1]; q.sP = (unsigned long long)blk[c][2]; q.pad = 0;
            (c == 0 ? parts0 : parts1)[(long long)b * gridDim.x + blockIdx.x] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_df_tables: (optionally) finish the coarse carrier estimate, then build the per-window NCO
// tables  E1[n1] = exp(-2*pi*i*df*n1*N2/fs),  E2[n2] = exp(-2*pi*i*df*n2/fs)
// (lo=exp(-j*2*pi*df*temps), godual_ranging.m:17 with temps=[0:N-1]/fs :72).  grid = (windows, slices): every slice finishes the
// estimate for itself (a few hundred records) and fills its share of the tables — one workgroup per window spent 9 us on the
// 8 625 fp64 sincospi of a 1-s window, which a one-window call (the per-second loops, MEX form A) waits for in full
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_df_tables(int estimate, const ArgPart<T>* __restrict__ part, int nparts,
                                                   double* __restrict__ dfv, long long* __restrict__ dfidx,
                                                   double fs, long long n, int n1, int n2,
                                                   cpx<T>* __restrict__ e1, cpx<T>* __restrict__ e2, double* __restrict__ sqmax = nullptr) {
    const int b = blockIdx.x;
    const bool lead = blockIdx.y == 0;
    __shared__ double s_df;
    __shared__ char scratch[64];
    if (estimate == 1) {
        Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
        for (int i = threadIdx.x; i < nparts; i += 256) best.take(part[(long long)b * nparts + i].val, part[(long long)b * nparts + i].idx);
        best = block_best<T, 256>(best, scratch);
        if (threadIdx.x == 0) {
            // freq=linspace(-fs/2,fs/2,N); df=freq(idx)/2  (godual_ranging.m:15,73; numpy linspace
            // arithmetic: start + i*step, endpoint exact) — separate mul/add, no FMA contraction
            const long long i = best.idx;
            const double step = fs / (double)(n - 1);
            double f = (i == n - 1) ? fs / 2 : __dadd_rn(__dmul_rn((double)i, step), -fs / 2);
            s_df = f / 2;
            if (lead) { dfv[b] = s_df; dfidx[b] = i; if (sqmax) sqmax[b] = sqrt((double)best.val); }     // valmax_square (process_OP.m:95)
        }
    } else if (threadIdx.x == 0) {
        s_df = dfv[b];
        if (estimate == 0 && lead) dfidx[b] = -1;     // 2: tables only (after the fine-frequency step)
    }
    __syncthreads();
    const double fn = s_df / fs;   // cycles per sample
    for (int i = threadIdx.x + 256 * blockIdx.y; i < n1 + n2; i += 256 * gridDim.y) {
        double ph;
        if (i < n1) ph = fn * ((double)i * (double)n2); else ph = fn * (double)(i - n1);
        ph -= rint(ph);
        double s, c;
        sincospi(-2.0 * ph, &s, &c);
        if (i < n1) e1[(long long)b * n1 + i] = mk<T>((T)c, (T)s);
        else e2[(long long)b * n2 + (i - n1)] = mk<T>((T)c, (T)s);
    }
}

// ------------------------------------------------------------------------------------------
// Fine carrier offset from the phase drift (optional; experiments/221219_twoway/processing/
// godual_ranging.py:26-30, Octave twin 221219…/godual_ranging.m:19-24, commented out in
// processing/Octave/godual_ranging.m:19-24):
//   a = polyfit([1:10:fs/3]/fs, conv(angle(y(1:10:fs/3)), ones(100,1)/100)(50:end-50), 1); df += a(1)/2/pi
// k_fine_angle: u[m] = angle((d[10m]-mean) * exp(-2 pi j df 10m/fs)), m < M  (fp64).  grid = (chunks, windows)
// k_fine_fit:   100-tap moving average with the reference's edge handling, straight-line LSQ slope,
//               df += slope/2pi.  grid = windows
// ------------------------------------------------------------------------------------------
template <class In>
__global__ __launch_bounds__(256) void k_fine_angle(In in, long long win_stride, int remove_mean, long long n,
                                                    const WinSums* __restrict__ sums, const double* __restrict__ dfv, double fs,
                                                    int M, double* __restrict__ u) {
    const int b = blockIdx.y;
    in.advance((long long)b * win_stride);
    const WinSums s = sums[b];
    const double mI = remove_mean ? (double)s.sI / (double)n : 0.0, mQ = remove_mean ? (double)s.sQ / (double)n : 0.0;
    const double fn = dfv[b] / fs;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const long long i = 10ll * m;
        const cpx<double> x = in.template load<double>(i);
        double ph = fn * (double)i; ph -= rint(ph);
        double sn, cs;
        sincospi(-2.0 * ph, &sn, &cs);
        const double re = x.x - mI, im = x.y - mQ;
        u[(long long)b * M + m] = atan2(re * sn + im * cs, re * cs - im * sn);
    }
}

__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < 16; ++w) t += sh[w];
    return t;
}

template <int UNUSED = 0>
__global__ __launch_bounds__(1024) void k_fine_fit(const double* __restrict__ u, int M, double fs, double* __restrict__ dfv) {
    const int b = blockIdx.x;
    const double* ub = u + (long long)b * M;
    __shared__ double sh[16];
    // pass 1: means of t and v
    double sv = 0, st = 0;
    for (int i = threadIdx.x; i < M; i += 1024) {
        const int lo = max(0, i - 50), hi = min(i + 49, M - 1);
        double a = 0;
        for (int j = lo; j <= hi; ++j) a += ub[j];
        sv += a * 0.01;
        st += (1.0 + 10.0 * (double)i) / fs;
    }
    const double vbar = block_sum_1024(sv, sh) / (double)M;
    const double tbar = block_sum_1024(st, sh) / (double)M;
    double sxy = 0, sxx = 0;
    for (int i = threadIdx.x; i < M; i += 1024) {
        const int lo = max(0, i - 50), hi = min(i + 49, M - 1);
        double a = 0;
        for (int j = lo; j <= hi; ++j) a += ub[j];
        const double dt = (1.0 + 10.0 * (double)i) / fs - tbar;
        sxy += dt * (a * 0.01 - vbar);
        sxx += dt * dt;
    }
    const double txy = block_sum_1024(sxy, sh);
    const double txx = block_sum_1024(sxx, sh);
    if (threadIdx.x == 0) dfv[b] += txy / txx / (2.0 * 3.14159265358979323846);
}

// ------------------------------------------------------------------------------------------
// k_col_fwd: first pass of the forward transform
// ------------------------------------------------------------------------------------------
enum { COL_MIX = 0, COL_SQUARE = 1, COL_PLAIN = 2 };

template <typename T> struct ColFwdArgs {
    long long in_win_stride;        // elements of the input type between windows
    const WinSums* sums;            // per window (MIX/SQUARE with mean removal) or nullptr
    int remove_mean;
    long long n;                    // N = n1*n2
    int n2, ntiles, nwin;
    const cpx<T>* e1; const cpx<T>* e2;   // per-window NCO tables (MIX)
    const cpx<T>* tw1;              // exp(-2 pi i m/N1)
    const cpx<T>* ta; const cpx<T>* tb; int tshift;  // exp(-2 pi i m/N) two-level: m = a<<tshift | b
    const cpx<T>* tc;               // [N1][W]: exp(-2 pi i k1 c/N), c < W  (coalesced part of the output twiddle)
    cpx<T>* out;                    // A[b][k1][n2]
};

template <class P1, typename T, int W, int MODE, class In, int NT>
__global__ __launch_bounds__(NT) void k_col_fwd(In in, ColFwdArgs<T> a) {
    using TL = Tile<P1, T, false, W, 0>;
    using C = cpx<T>;
    constexpr int S = P1::S;
    __shared__ C lds[TL::lds_elems + 32];                    // + the wave-uniform output twiddles (s_wq)
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / a.ntiles, tile = logical % a.ntiles;
    const int c0 = tile * W;
    const int tid = threadIdx.x;
    In win = in; win.advance((long long)b * a.in_win_stride);
    T mx = 0, my = 0;
    if (a.remove_mean) {
        WinSums s = a.sums[b];
        mx = (T)((double)s.sI / (double)a.n);
        my = (T)((double)s.sQ / (double)a.n);
    }
    C v[P1::rmax()];
    // ---- stage 0: global → registers
    {
        constexpr int R = P1::radix(0);
        if (tid < TL::template tasks<0>()) {
            const int j = tid / W, c = tid % W;
            const unsigned lane_in = (unsigned)j * (unsigned)a.n2 + (unsigned)(c0 + c);
            unsigned raw[In::has_raw ? R : 1];
            if constexpr (In::has_raw) {            // every load of the butterfly issued before anything waits
                TWX_UNROLL
                for (int r = 0; r < R; ++r) raw[r] = win.load_raw((long long)(r * (P1::L / R)) * a.n2, lane_in);
            }
            // NCO factor exp(-j 2 pi df n/fs), n = (j + r*T)*N2 + n2:  E1[j]·E2[n2] per thread, E1[r*T] wave-uniform
            C ejc = mk<T>(1, 0);
            if (MODE == COL_MIX) ejc = cmul(a.e1[(long long)b * P1::L + j], a.e2[(long long)b * a.n2 + c0 + c]);
            C rs_rot = mk<T>(1, 0);
            double rs_t0 = 0; int rs_edge = 0;
            if constexpr (InTraits<In>::resample) {
                static_assert(!InTraits<In>::resample || MODE == COL_MIX, "the resampled window is the mixed one");
                rs_rot = a.e2[(long long)b * a.n2 + 1];           // exp(-2 pi i df / fs): the NCO's step from one sample to the next
                rs_t0 = in.t0[b]; rs_edge = in.edge[b];
            }
            TWX_UNROLL
            for (int r = 0; r < R; ++r) {
                C x;
                if constexpr (InTraits<In>::resample) {
                    const long long n = (long long)(j + r * (P1::L / R)) * a.n2 + (c0 + c);
                    // (complex double: a real call per sample — inlined 25 times the body's address arithmetic and scalar loads were hoisted
                    // to the top of the unrolled loop and spilled)
                    if constexpr (sizeof(T) == 8) x = rs_sample_call<T>(win.p, win.nch, n, a.n, in.c, rs_t0, rs_edge, mx, my, rs_rot);
                    else x = rs_sample<T>(win.p, win.nch, n, a.n, in.c, rs_t0, rs_edge, mx, my, rs_rot);
                    C e = (r == 0) ? ejc : cmul(a.e1[(long long)b * P1::L + r * (P1::L / R)], ejc);
                    v[r] = cmul(x, e);
                    continue;
                }
                if constexpr (In::has_raw) x = (TWX_ABLF == 2 || TWX_ABLF == 4) ? mk<T>((T)(tid + r), (T)(r - tid)) : In::template unpack<T>(raw[r]);
                else if constexpr (!InTraits<In>::resample) x = win.template load2<T>((long long)(r * (P1::L / R)) * a.n2, lane_in);
                x.x -= mx; x.y -= my;
                if (MODE == COL_MIX) {
                    C e = (r == 0) ? ejc : cmul(a.e1[(long long)b * P1::L + r * (P1::L / R)], ejc);   // scalar load × per-thread constant
                    x = cmul(x, e);
                } else if (MODE == COL_SQUARE) {
                    x = mk<T>(x.x * x.x - x.y * x.y, T(2) * x.x * x.y);
                }
                v[r] = x;
            }
            if (TWX_ABLF != 1 && TWX_ABLF < 4) TL::template bfly<0>(v);
            if (S > 1 && TWX_ABLF != 1 && TWX_ABLF < 4) TL::template store_lds<0>(lds, j, c, v);
        }
    }
    // S == 2: everything of the last stage that does not depend on the data is done before the barrier (v is dead
    // until then): the stage-twiddle gather, the per-thread part W_N^{j c0} of the output twiddle folded into those
    // twiddles (the butterfly is linear), and the wave-uniform parts W_N^{q QS c0} computed once per workgroup by
    // RL threads into LDS instead of 2*RL dependent scalar loads per wave.
    constexpr bool TWPRE = (S == 2);
    constexpr int RLc = P1::radix(S - 1);
    C twr[TWPRE ? RLc - 1 : 1];
    C wj0 = mk<T>(1, 0);
    C* s_wq = lds + TL::lds_elems;
    if constexpr (TWPRE) {
        const unsigned mask = (1u << a.tshift) - 1u;
        if (tid < TL::template tasks<S - 1>()) {
            const int j = tid / W;
            TL::template load_tw<S - 1>(a.tw1, j, twr);
            const unsigned mj = (unsigned)j * (unsigned)c0;
            wj0 = cmul(a.ta[mj >> a.tshift], a.tb[mj & mask]);
            TWX_UNROLL
            for (int r = 1; r < RLc; ++r) twr[r - 1] = cmul(twr[r - 1], wj0);
        }
        if (tid < RLc) {
            const unsigned mq = (unsigned)(tid * (P1::L / RLc)) * (unsigned)c0;
            s_wq[tid] = cmul(a.ta[mq >> a.tshift], a.tb[mq & mask]);
        }
    }
    if (S > 1) __syncthreads();
    if constexpr (S > 2) {
        if (tid < TL::template tasks<1>()) { TL::template load_lds<1>(lds, a.tw1, tid / W, tid % W, v); TL::template bfly<1>(v); }
        __syncthreads();
        if (tid < TL::template tasks<1>()) TL::template store_lds<1>(lds, tid / W, tid % W, v);
        __syncthreads();
    }
    if constexpr (S > 3) {
        if (tid < TL::template tasks<2>()) { TL::template load_lds<2>(lds, a.tw1, tid / W, tid % W, v); TL::template bfly<2>(v); }
        __syncthreads();
        if (tid < TL::template tasks<2>()) TL::template store_lds<2>(lds, tid / W, tid % W, v);
        __syncthreads();
    }
    // ---- last stage: registers → ·W_N^{k1 n2} → global
    {
        constexpr int s = S - 1;
        constexpr int R = P1::radix(s);
        if (tid < TL::template tasks<s>()) {
            const int j = tid / W, c = tid % W;
            if (S > 1 && TWX_ABLF != 1 && TWX_ABLF < 4) {
                if constexpr (TWPRE) { TL::template load_lds_tw<s>(lds, twr, j, c, v); v[0] = cmul(v[0], wj0); }
                else TL::template load_lds<s>(lds, a.tw1, j, c, v);
                TL::template bfly<s>(v);
            }
            const unsigned n2i = c0 + c;
            const unsigned mask = (1u << a.tshift) - 1u;
            // A is stored tile-blocked, A[b][tile][k1][c]: a workgroup writes ONE contiguous N1*W*8-byte block (strided
            // 128-B row pieces were the slowest part of this kernel: 3.1 TB/s for the stores alone); the row pass
            // gathers its row from the N2/W blocks in 128-B pieces, which reads handle well (a_index below)
            char* out = reinterpret_cast<char*>(a.out + (long long)b * a.n + (long long)tile * (P1::L * W));
            const unsigned lane_out = ((unsigned)TL::template out_pos<s>(j, 0) * W + c) * (unsigned)sizeof(C);
            constexpr int QS = (S == 1) ? 1 : P1::L / R;     // row step between a thread's outputs
            // W_N^{k1 n2}, k1 = j + q*QS, n2 = c0 + c  =  W_N^{j c0} (per thread) · W_N^{q QS c0} (wave-uniform)
            //                                            · W_N^{k1 c} (one coalesced 8-B load from tc[k1][c])
            const unsigned mj = (S == 1 || TWPRE) ? 0u : (unsigned)j * (unsigned)c0;
            const C wj = TWPRE ? mk<T>(1, 0) : cmul(a.ta[mj >> a.tshift], a.tb[mj & mask]);
            const char* tcb = reinterpret_cast<const char*>(a.tc);
            const unsigned tcl = ((unsigned)TL::template out_pos<s>(j, 0) * W + c) * (unsigned)sizeof(C);
            constexpr unsigned ostride = QS * W * sizeof(C);
            // All twiddle loads first, all stores last: on gfx9 loads and stores share vmcnt, so a table load issued
            // after a store can only be waited for together with that store — interleaving them made the epilogue a
            // chain of 25 load + store-acknowledge round trips (12 us per workgroup).
            TWX_UNROLL
            for (int q = 0; q < R; ++q) {
                if constexpr (TWPRE) v[q] = cmul3(v[q], s_wq[q], ld_su<C>(tcb + (q * QS * W * (int)sizeof(C)), tcl));
                else {
                    const unsigned mq = (unsigned)(q * QS) * (unsigned)c0;              // wave-uniform: scalar loads
                    const C wq = cmul(a.ta[mq >> a.tshift], a.tb[mq & mask]);
                    v[q] = cmul(v[q], cmul3(wq, wj, ld_su<C>(tcb + (q * QS * W * (int)sizeof(C)), tcl)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            TWX_UNROLL
            for (int q = 0; q < R; ++q) {
                if (TWX_ABLF == 3 || TWX_ABLF == 5) { asm volatile("" ::"v"(v[q])); }
                else st_su<C>(out + q * ostride, lane_out, v[q]);                         // uniform base + 32-bit lane offset
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_col_fwd3: k_col_fwd for two-stage fp32 column plans with the stage exchange done one component at a time
// (40 KB of LDS instead of 80, ~70 registers instead of 100: three to four workgroups per CU instead of two; see
// k_col_inv3).  Same arithmetic, same order of operations on every value as k_col_fwd's two-stage path.
// ------------------------------------------------------------------------------------------
template <class P1, typename T, int W, int MODE, class In, int NT>
__global__ __launch_bounds__(NT, 6) void k_col_fwd3(In in, ColFwdArgs<T> a) {
    static_assert(P1::S == 2 && sizeof(T) == 4, "two-stage fp32 plans only");
    using TL = Tile<P1, T, false, W, 0>;
    using C = cpx<T>;
    constexpr int L = P1::L, R0 = P1::radix(0), R1 = P1::radix(1);
    __shared__ T lf[L * W];
    __shared__ C s_wq[32];
    // (per mode: bit 0 the stage twiddles, bit 1 the output twiddle)
    // (the long column plans, 500 and 625 points, where it was measured; the 250-point plan's MIX form ran out of its 80 registers with it)
    constexpr int LDSM = L < 500 ? 0 : (MODE == COL_SQUARE ? TWX_FWD3_LDS_SQ : (MODE == COL_MIX ? TWX_FWD3_LDS_MIX : 0));
    // Every table value the last stage needs is asked for BEFORE the tile's samples and waits in LDS (round 6, after k_col_inv3): the stage
    // twiddles W_L^{j r} straight into LDS (global_load_lds: no register in between), and the output twiddle W_N^{k1 n2}, k1 = j + q QS,
    // n2 = c0 + c, as (W_N^{j c0} W_N^{j c}) — one value per thread, folded into the stage twiddles — times s_w2[q][c] = W_N^{q QS (c0 + c)},
    // R1 x W values per workgroup.  No load is issued after the exchange: 25 + 24 + 2 late L2 round trips per lane leave the critical path.
    __shared__ C s_tw[(LDSM & 1) ? L : 1];
    __shared__ C s_w2[(LDSM & 2) ? R1 * W : 1];
    if constexpr (LDSM & 1) lds_fill_dwords<NT>(reinterpret_cast<float*>(s_tw), reinterpret_cast<const float*>(a.tw1), 2 * L);
    static_assert(R1 <= 32, "output-twiddle table");
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / a.ntiles, tile = logical % a.ntiles;
    const int c0 = tile * W;
    const int tid = threadIdx.x;
    const int j = tid / W, c = tid % W;
    const bool p0 = tid < TL::template tasks<0>(), p1 = tid < TL::template tasks<1>();
    const unsigned mask = (1u << a.tshift) - 1u;
    In win = in; win.advance((long long)b * a.in_win_stride);
    T mx = 0, my = 0;
    if (a.remove_mean) {
        WinSums s = a.sums[b];
        mx = (T)((double)s.sI / (double)a.n);
        my = (T)((double)s.sQ / (double)a.n);
    }
    C wjx = mk<T>(1, 0);
    if constexpr (LDSM & 2) {
        const int jj = min(j, L / R1 - 1);                       // (idle lanes of the last wave: a valid address)
        const unsigned mj = (unsigned)jj * (unsigned)c0;
        wjx = cmul3(a.ta[mj >> a.tshift], a.tb[mj & mask], a.tc[jj * W + c]);          // W_N^{j c0} W_N^{j c}
        const int q = min(tid / W, R1 - 1), cc = tid % W;
        const unsigned mq = (unsigned)(q * (L / R1)) * (unsigned)c0;
        const C w2 = cmul3(a.ta[mq >> a.tshift], a.tb[mq & mask], a.tc[(q * (L / R1)) * W + cc]);      // W_N^{q QS c0} W_N^{q QS c}
        if (tid < R1 * W) s_w2[tid] = w2;
    }
    C v[R0];
    if (p0) {
        const unsigned lane_in = (unsigned)j * (unsigned)a.n2 + (unsigned)(c0 + c);
        unsigned raw[In::has_raw ? R0 : 1];
        if constexpr (In::has_raw) {
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) raw[r] = win.load_raw((long long)(r * (L / R0)) * a.n2, lane_in);
        }
        C ejc = mk<T>(1, 0);
        if (MODE == COL_MIX) ejc = cmul(a.e1[(long long)b * L + j], a.e2[(long long)b * a.n2 + c0 + c]);
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) {
            C x;
            if constexpr (In::has_raw) x = In::template unpack<T>(raw[r]);
            else x = win.template load2<T>((long long)(r * (L / R0)) * a.n2, lane_in);
            x.x -= mx; x.y -= my;
            if (MODE == COL_MIX) {
                C e = (r == 0) ? ejc : cmul(a.e1[(long long)b * L + r * (L / R0)], ejc);
                x = cmul(x, e);
            } else if (MODE == COL_SQUARE) {
                x = mk<T>(x.x * x.x - x.y * x.y, T(2) * x.x * x.y);
            }
            v[r] = x;
        }
        TL::template bfly<0>(v);
        const int ob = TL::template out_base<0>(j);
        TWX_UNROLL
        for (int q = 0; q < R0; ++q) lf[TL::template out_idx<0>(ob, j, q) * W + c] = v[q].x;
    }
    constexpr int QS = L / R1;                                   // row step between a thread's outputs
    if (!(LDSM & 2) && tid < R1) {                               // wave-uniform parts W_N^{q QS c0} of the output twiddle
        const unsigned mq = (unsigned)(tid * QS) * (unsigned)c0;
        s_wq[tid] = cmul(a.ta[mq >> a.tshift], a.tb[mq & mask]);
    }
    __syncthreads();
    C u[R1];
    const int ib = TL::template in_base<1>(j);
    if (p1) {
        TWX_UNROLL
        for (int r = 0; r < R1; ++r) u[r].x = lf[TL::template in_idx<1>(ib, j, r) * W + c];
    }
    __syncthreads();
    if (p0) {
        const int ob = TL::template out_base<0>(j);
        TWX_UNROLL
        for (int q = 0; q < R0; ++q) lf[TL::template out_idx<0>(ob, j, q) * W + c] = v[q].y;
    }
    __syncthreads();
    if (p1) {
        TWX_UNROLL
        for (int r = 0; r < R1; ++r) u[r].y = lf[TL::template in_idx<1>(ib, j, r) * W + c];
        // stage twiddles W_L^{j r} with the per-thread part W_N^{j c0} of the output twiddle folded in (the butterfly is linear)
        C wj0 = wjx;
        if constexpr (!(LDSM & 2)) {
            const unsigned mj = (unsigned)j * (unsigned)c0;
            wj0 = cmul(a.ta[mj >> a.tshift], a.tb[mj & mask]);
        }
        u[0] = cmul(u[0], wj0);
        TWX_UNROLL
        for (int r = 1; r < R1; ++r) {
            if constexpr (LDSM & 1) u[r] = cmul(u[r], cmul(s_tw[j * r], wj0));
            else u[r] = cmul(u[r], cmul(tw_load(a.tw1, (unsigned)(j * r)), wj0));
        }
        TL::template bfly<1>(u);
        // W_N^{k1 n2}, k1 = j + q*QS, n2 = c0 + c: remaining factors W_N^{q QS c0} (LDS) and W_N^{k1 c} (one coalesced 8-B load)
        char* out = reinterpret_cast<char*>(a.out + (long long)b * a.n + (long long)tile * (L * W));
        const unsigned lane_out = ((unsigned)TL::template out_pos<1>(j, 0) * W + c) * (unsigned)sizeof(C);
        const char* tcb = reinterpret_cast<const char*>(a.tc);
        constexpr unsigned ostride = QS * W * sizeof(C);
        if constexpr (LDSM & 2) {
            TWX_UNROLL
            for (int q = 0; q < R1; ++q) u[q] = cmul(u[q], s_w2[q * W + c]);
        } else {
            TWX_UNROLL
            for (int q = 0; q < R1; ++q) u[q] = cmul3(u[q], s_wq[q], ld_su<C>(tcb + (q * QS * W * (int)sizeof(C)), lane_out));
            __builtin_amdgcn_sched_barrier(0);                   // all table loads before the first store (shared vmcnt)
        }
        TWX_UNROLL
        for (int q = 0; q < R1; ++q) st_su<C>(out + q * ostride, lane_out, u[q]);
    }
}

// ------------------------------------------------------------------------------------------
// k_row: second pass of the forward transform with a fused epilogue
// ------------------------------------------------------------------------------------------
enum { ROW_STORE = 0, ROW_BAND = 1, ROW_MID = 2 };
// element (k1, n2) of the tile-blocked column-pass output A[tile][k1][c], tile = n2 >> wshift, c = n2 & (W-1)
__device__ __forceinline__ unsigned a_index(unsigned n2, unsigned k1, unsigned n1, int wshift) {
    return ((((n2 >> wshift) * n1 + k1) << wshift) | (n2 & ((1u << wshift) - 1u)));
}
#define TWX_MAX_PHASE 5
#ifndef TWX_ROW_WAVES
#define TWX_ROW_WAVES 1
#endif

template <typename T> struct RowArgs {
    long long n; int n1, nwin;
    const cpx<T>* A;                 // [b][tile][k1][c], tile = n2 >> wshift (see a_index)
    int wshift;                      // log2 of the column-pass tile width
    const cpx<T>* stab_f;            // StageTabs<P2> entries (forward sign)
    const cpx<T>* stab_i;            // StageTabs<Rev<P2>> entries (forward sign; used conjugated)
    // STORE
    cpx<T>* spec_out;                // [b][k1][k2]
    int conj_out; int hamming;       // code-spectrum options (main.cpp:717-719 window)
    // BAND
    long long band_lo, band_hi;      // inclusive, indices of the fftshifted spectrum
    ArgPart<T>* part;                // [b][n1]
    // MID
    const cpx<T>* cspec;             // conj(FFT(code)) in [k1][k2] layout
    // interpolation ramp exp(+2 pi i rho k2s/(R N2)), k2 = j + Ns*r, split as ea[rho][j]*eb[rho][wrap][r]
    const cpx<T>* ea;                // [R][Ns], Ns = N2 / (last forward radix)
    const cpx<T>* eb;                // [R][2][RL]
    const cpx<T>* ramp1;             // [R][N1]: exp(+2 pi i rho k1/(R N))
    int nphase;                      // R = 2*Nint+1
    T scale;                         // power of two applied to the product (range safety)
    const cpx<T>* ta; const cpx<T>* tb; int tshift;   // exp(-2 pi i m/N), m = a<<tshift | b
    cpx<T>* Bz;                      // [b][rho][k1][q2]
    cpx<T>* dc;                      // [b]  X[0] of the window (mean(y) for puissance, :46)
    unsigned long long* stamps;      // diagnostic builds (TWX_STAMPS) only
    int pf_stride;                   // k_rowd<MID>: workgroups in the launch (a few per CU, two resident at a time); 0: one per row.  The launch
                                     // has min(rows, pf_stride) workgroups, workgroup g takes the rows g, g + grid, g + 2 grid, ... (same XCD)
};

// reverse of a plan (inverse transform consumes the forward's last-stage register layout)
template <class P> struct Rev;
template <int L, int R0, int R1, int R2, int R3> struct Rev<Plan<L, R0, R1, R2, R3>> {
    using type = typename std::conditional<R1 == 1, Plan<L, R0>,
                 typename std::conditional<R2 == 1, Plan<L, R1, R0>,
                 typename std::conditional<R3 == 1, Plan<L, R2, R1, R0>, Plan<L, R3, R2, R1, R0>>::type>::type>::type;
};

// middle stages (1 .. S-2) of an in-LDS row transform, registers v, in place
template <class TL, class P, typename T, int s> struct MidStages {
    static __device__ __forceinline__ void run(cpx<T>* lds, const cpx<T>* tabs, cpx<T>* v, int tid) {
        if constexpr (s < P::S - 1) {
            if (tid < TL::template tasks<s>()) { TL::template load_lds_tab<s>(lds, tabs, tid, v); TL::template bfly<s>(v); }
            __syncthreads();
            if (tid < TL::template tasks<s>()) TL::template store_lds<s>(lds, tid, 0, v);
            __syncthreads();
            MidStages<TL, P, T, s + 1>::run(lds, tabs, v, tid);
        }
    }
};

#ifdef TWX_STAMPS   // diagnostic build only: s_memtime at segment boundaries of one wave per workgroup
#define TWX_STAMP(i) do { if (MODE == ROW_MID && a.stamps && (threadIdx.x & 63) == 0) { \
        unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        a.stamps[((long long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 32 + (i)] = t_; } } while (0)
#else
#define TWX_STAMP(i) do { } while (0)
#endif

template <class P2, typename T, int MODE, int PADQ, int NT>
__global__ __launch_bounds__(NT, TWX_ROW_WAVES) void k_row(RowArgs<T> a) {
    using C = cpx<T>;
    using TF = RowTile<P2, T, false, PADQ>;
    using PR = typename Rev<P2>::type;
    using TI = RowTile<PR, T, true, PADQ>;
    constexpr int S = P2::S;
    constexpr int N2 = P2::L;
    constexpr int RL = P2::radix(S - 1);          // forward last radix == inverse first radix
    constexpr int NSL = N2 / RL;                  // tasks of those stages
    constexpr bool PAL = std::is_same<P2, PR>::value;
    constexpr int NTF = StageTabs<P2>::total, NTI = (MODE == ROW_MID && !PAL) ? StageTabs<PR>::total : 0;
    constexpr int NEB = MODE == ROW_MID ? TWX_MAX_PHASE * 2 * RL : 0;
    constexpr int NVC = MODE == ROW_MID ? PR::radix(S - 1) : 0;
    static_assert(S >= 2, "row plans need >= 2 stages");
    __shared__ C lds[TF::lds_elems + NTF + NTI + NEB + NVC + 16];
    C* tab_f = lds + TF::lds_elems;
    C* tab_i = PAL ? tab_f : tab_f + NTF;
    C* s_eb = tab_f + NTF + NTI;
    C* s_vc = s_eb + NEB;
    void* red = (void*)(s_vc + NVC);
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int k1 = logical / a.nwin, b = logical % a.nwin;   // same k1 of all windows adjacent → shared code-spectrum row
    const int tid = threadIdx.x;
    const C* Ab = a.A + (long long)b * a.n;
    C v[P2::rmax()];
    C csr[MODE == ROW_MID ? RL : 1];     // code-spectrum row, requested at kernel start (latency hidden by the forward FFT)
    if constexpr (MODE == ROW_MID) {
        if (tid < NSL) {
            const C* cs = a.cspec + (long long)k1 * N2;
            TWX_UNROLL
            for (int q = 0; q < RL; ++q) csr[q] = (cs + q * NSL)[(unsigned)tid];
        }
    }
    TWX_STAMP(0);
    // ---- forward stage 0 (global → regs → LDS); small tables global → LDS
    {
        constexpr int R = P2::radix(0);
        if (tid < TF::template tasks<0>()) {
            TWX_UNROLL
            for (int r = 0; r < R; ++r) v[r] = (TWX_ABL == 4 && MODE == ROW_MID) ? mk<T>((T)(tid + r), (T)r) : Ab[a_index((unsigned)(tid + r * (N2 / R)), (unsigned)k1, (unsigned)a.n1, a.wshift)];
        }
        for (int i = tid; i < NTF; i += NT) tab_f[i] = a.stab_f[i];
        if constexpr (MODE == ROW_MID) {
            if constexpr (!PAL) for (int i = tid; i < NTI; i += NT) tab_i[i] = a.stab_i[i];
            for (int i = tid; i < a.nphase * 2 * RL; i += NT) s_eb[i] = a.eb[i];
            // Vc[q] = conj(W_N^{k1 * q * NSLi}), NSLi = N2 / (last inverse radix)
            constexpr int RIL = PR::radix(S - 1);
            if (tid < RIL) {
                const unsigned m = (unsigned)k1 * (unsigned)tid * (unsigned)(N2 / RIL);
                s_vc[tid] = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & ((1u << a.tshift) - 1u)]));
            }
        }
        if (tid < TF::template tasks<0>()) {
            TF::template bfly<0>(v);
            TF::template store_lds<0>(lds, tid, 0, v);
        }
    }
    TWX_STAMP(1);
    __syncthreads();
    TWX_STAMP(2);
    MidStages<TF, P2, T, 1>::run(lds, tab_f, v, tid);
    TWX_STAMP(3);
    // ---- forward last stage: v[q] = X[k1 + N1*k2], k2 = tid + q*NSL
    const bool act = tid < NSL;
    if (act) { TF::template load_lds_tab<S - 1>(lds, tab_f, tid, v); TF::template bfly<S - 1>(v); }

    if constexpr (MODE == ROW_STORE) {
        if (act) {
            C* out = a.spec_out + (long long)b * a.n + (long long)k1 * N2;
            TWX_UNROLL
            for (int q = 0; q < RL; ++q) {
                const int k2 = tid + q * NSL;
                C x = v[q];
                if (a.conj_out) x = cconj(x);
                if (a.scale != T(0)) x = cscale(x, a.scale);   // code spectrum is stored pre-scaled (0 = leave as is)
                if (a.hamming) {
                    const double k = (double)k1 + (double)a.n1 * (double)k2;
                    x = cscale(x, (T)(0.54 - 0.46 * cospi(2.0 * k / (double)(a.n - 1))));
                }
                out[k2] = x;
            }
        }
    } else if constexpr (MODE == ROW_BAND) {
        // d2=fftshift(abs(fft(d.^2))); [~,df]=max(d2(k))   (godual_ranging.m:14-15)
        Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
        if (act) {
            const long long half = a.n / 2;   // shifted index i ↔ bin k = (i - floor(N/2)) mod N
            TWX_UNROLL
            for (int q = 0; q < RL; ++q) {
                const int k2 = tid + q * NSL;
                long long k = (long long)k1 + (long long)a.n1 * k2;
                long long i = k - (a.n - half); if (i < 0) i += a.n;
                if (i >= a.band_lo && i <= a.band_hi) best.take(cnorm(v[q]), (unsigned int)i);
            }
        }
        best = block_best<T, NT>(best, red);
        if (tid == 0) { ArgPart<T> p; p.val = best.val; p.idx = best.idx; a.part[(long long)b * a.n1 + k1] = p; }
    } else {
        // ---- MID: product with the code spectrum, R phase-ramped inverse transforms
        constexpr int RIL = PR::radix(S - 1);
        constexpr int NSI = N2 / RIL;
        C pr[RL];
        if (act) {
            if (k1 == 0 && tid == 0) a.dc[b] = v[0];
            TWX_UNROLL
            for (int q = 0; q < RL; ++q)
                pr[q] = cmul(v[q], TWX_ABL == 2 ? v[(q + 1) % RL] : csr[q]);   // ffty.*fcode (godual_ranging.m:26); fcode carries the range scale
        }
        // per-thread output twiddle base conj(W_N^{k1*j}), j = output task index of the inverse's last stage
        C ub = mk<T>(1, 0);
        if (tid < NSI) {
            const unsigned m = (unsigned)k1 * (unsigned)tid;
            ub = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & ((1u << a.tshift) - 1u)]));
        }
        TWX_STAMP(4);
        const int nph = TWX_ABL == 3 ? 1 : TWX_ABL == 5 ? 0 : a.nphase;
        // software-pipelined per-phase scalars: the ramp factor of the NEXT phase is requested one
        // phase ahead so its global-load latency never sits on the critical path
        C ea_cur = mk<T>(1, 0), ea_n = mk<T>(1, 0);
        if (nph > 1 && tid < NSL) ea_n = a.ea[NSL + tid];
        C r1_cur = a.ramp1[k1];
        for (int rho = 0; rho < nph; ++rho) {
            __syncthreads();   // previous transform's LDS reads are done
            // launder the thread index: stops LICM from hoisting ~100 loop-invariant LDS/table
            // addresses out of the rho loop (they were being spilled to scratch)
            int lt = tid;
            asm volatile("" : "+v"(lt));
            TWX_STAMP(5 + rho * 6);
            const C eaj = ea_cur;          // exp(+2 pi i rho j/(R N2)) of THIS phase (unused for rho = 0)
            const C r1 = r1_cur;
            ea_cur = ea_n;
            if (rho + 2 < nph && lt < NSL) ea_n = a.ea[(rho + 2) * NSL + lt];
            if (rho + 1 < nph) r1_cur = a.ramp1[(long long)(rho + 1) * a.n1 + k1];
            if (lt < NSL) {
                if (rho == 0) {
                    TWX_UNROLL
                    for (int r = 0; r < RL; ++r) v[r] = pr[r];
                } else {
                    TWX_UNROLL
                    for (int r = 0; r < RL; ++r) {
                        C e;
                        if constexpr (RL % 2 == 0) e = a.eb[(rho * 2 + (r >= RL / 2 ? 1 : 0)) * RL + r];   // wave-uniform → scalar load
                        else e = s_eb[(rho * 2 + ((2 * (lt + r * NSL) >= N2) ? 1 : 0)) * RL + r];
                        v[r] = cmul3(pr[r], eaj, e);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                TI::template bfly<0>(v);
                TI::template store_lds<0>(lds, lt, 0, v);
            }
            TWX_STAMP(6 + rho * 6);
            __syncthreads();
            TWX_STAMP(7 + rho * 6);
            MidStages<TI, PR, T, 1>::run(lds, tab_i, v, lt);
            TWX_STAMP(8 + rho * 6);
            if (lt < NSI) {
                TI::template load_lds_tab<S - 1>(lds, tab_i, lt, v);
                TI::template bfly<S - 1>(v);
                __builtin_amdgcn_sched_barrier(0);
                TWX_STAMP(9 + rho * 6);
                const C u = cmul(ub, r1);
                C* out = a.Bz + ((long long)b * a.nphase + rho) * a.n + (long long)k1 * N2;
                TWX_UNROLL
                for (int q = 0; q < RIL; ++q) {
                    C o = cmul3(v[q], u, s_vc[q]);                     // · W_N^{-k1 q2} · ramp1
                    if (TWX_ABL == 1) { asm volatile("" ::"v"(o)); } else (out + q * NSI)[(unsigned)lt] = o;
                }
            }
            TWX_STAMP(10 + rho * 6);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_rowd: the row pass on the DIF/DIT transform (RowD, twx_fft.h): same epilogues as k_row<BAND> and
// k_row<MID>, but only the stride-(N2/R0) stages synchronise the workgroup — 7 workgroup barriers
// per row for the full middle pass instead of 21.  The code spectrum is read in "block-thread"
// order (cspec_perm[k1][q2][q0*R+q1] = conj(FFT(code))[k1 + N1*(q0 + R0 q1 + R0 R q2)]).
// ------------------------------------------------------------------------------------------
template <typename T> struct RowDArgs {
    RowArgs<T> r;                    // shared fields (A, band, part, ramp1, ta/tb, Bz, dc, nphase, ...)
    const cpx<T>* dtabs;             // RowD tables: ta | tb | tc
    const cpx<T>* cspec_perm;        // [k1][q2][u]
    const cpx<T>* ea_d;              // [rho][u]      exp(+2 pi i rho (q0 + R0 q1)/(Rint N2)), u = q0*R + q1
    const cpx<T>* eb_d;              // [rho][2][q2]  exp(+2 pi i rho (R0 R q2 - wrap N2)/(Rint N2))
    // ROW_BAND with a narrow search band: only the bins k2 = q0 + R0 q1 + R0 R q2 of a few (q1, q2) pairs can lie inside it
    // (k = k1 + N1 k2; +-20 kHz of a 5-Msps second = |k2| <= 32 of 8000: 4 pairs).  nprune > 0: the last stage is replaced
    // by one R-term sum per (q0, pair) instead of a full radix-R butterfly in every lane.
    const cpx<T>* vc;                // [k1][c]  exp(+2 pi i k1 c M/N), c < R0: the per-row output twiddle of stage C, read with scalar loads
    const cpx<T>* vw;                // [k1][2][R]  exp(+2 pi i k1 a/N), exp(+2 pi i k1 R b/N): the folded output twiddle of k_rowd<MID>
    const cpx<T>* wr;                // [R] exp(-2 pi i j / R)
    const cpx<T>* wm;                // [R R] exp(-2 pi i j / (R R)): k_rowd_bandsum
    int nprune;                      // number of (q1, q2) pairs, 0 = full last stage
    unsigned long long pr_q1, pr_q2; // pair p in byte p
    unsigned total_rows;             // N1 * windows (the grid is smaller when k_rowd<MID> runs with resident workgroups)
    // TWX_OPT_SELFCHECK (the CHK instantiation of k_rowd<MID>): per window a status word, context-wide statistics, the relative tolerance
    float* chk_rows;                 // [b][k1][TWX_CHK_SLOTS]: the row's energy sums, compared by k_chk_verdict
    int chk_fault;                   // TWX_OPT_DEBUG_FAULT (CHK instantiation only): 0 none; 2*(row + 1) + which damages one value of that row —
                                     // which 0: between the forward stages, 1: between the inverse stages of the last phase
};

// TWX_OPT_SELFCHECK: energy sums of one row, reduced over the wave (DPP-free butterfly on shuffles) and added to the workgroup's LDS slot
// (DPP inside the rows of 16 lanes — no LDS traffic, no address registers: the kernel has none to spare — then four v_readlane)
#define TWX_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true)))
__device__ __forceinline__ void chk_add(float* slot, float v) {
    v = TWX_DPP_ADD(v, 0xB1);       // quad_perm [1,0,3,2]
    v = TWX_DPP_ADD(v, 0x4E);       // quad_perm [2,3,0,1]
    v = TWX_DPP_ADD(v, 0x141);      // row_half_mirror
    v = TWX_DPP_ADD(v, 0x140);      // row_mirror: every lane of a row of 16 holds the row's sum
    const int iv = __builtin_bit_cast(int, v);
    const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, t);
}
#ifndef TWX_CHK_PARTS
#define TWX_CHK_PARTS 7      // diagnostic: bit 0 the |A|^2 sum, bit 1 spectrum + product, bit 2 the Bz sums
#endif
#define TWX_CHK_SLOTS (3 + TWX_MAX_PHASE)     // |A row|^2, |spectrum row|^2, |product row|^2, |Bz row|^2 of every phase

#ifdef TWX_ROWD_CHECK
template <typename T, int R> __device__ __attribute__((noinline)) void rowd_check_bfly(const cpx<T>* in, cpx<T>* out) {
    cpx<T> t[R];
    for (int r = 0; r < R; ++r) t[r] = in[r];
    Bfly<T, R, false>::run(t);
    for (int r = 0; r < R; ++r) out[r] = t[r];
}
// the same question with nothing but registers: arguments and results of a non-inlined function travel in VGPRs (no scratch memory)
struct RowdCheck4 { cpx<float> a, b, c, d; };
__device__ __attribute__((noinline)) RowdCheck4 rowd_check4(cpx<float> a, cpx<float> b, cpx<float> c, cpx<float> d, int tag) {
    asm volatile("" :: "v"(tag) : "memory");          // not a pure function: two calls with the same values stay two calls
    cpx<float> t[4] = {a, b, c, d};
    Bfly<float, 4, false>::run(t);
    t[1] = cmul(t[1], t[3]); t[2] = cmul(t[2], t[0]);
    return RowdCheck4{t[0], t[1], t[2], t[3]};
}
#endif
template <class P> constexpr int N2_of_rowd() { return P::L; }
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// k_rowd<MID> with resident workgroups (see the kernel): the launcher sizes the grid with the same predicate
template <class P2, typename T> constexpr bool rowd_mid_resident() {
    using D = RowD<P2, T>;
    // complex double: only where the row, the tables and the copy of the forward table tc still fit the 160 KB of one CU
    constexpr bool fits64 = (size_t)(D::lds_elems + D::tab_total + D::R0 * D::R + 32) * sizeof(cpx<double>) + 1024 <= 160 * 1024;
    return TWX_MID_PERSIST && TWX_MID_FOLD && D::R0 > 1 && (sizeof(T) == 4 || (TWX_MID_PERSIST64 && fits64)) && D::M % 16 == 0;
}

// CHK (TWX_OPT_SELFCHECK, MID only): every row is checked against Parseval's identity in both directions — sum |A row|^2 * N2 = sum |spectrum
// row|^2 for the forward transform, sum |product row|^2 * N2 = sum |Bz row|^2 for every phase of the inverse (all twiddles, ramps and folded
// factors have modulus one) — from values the passes hold in registers anyway: a wave reduction and one LDS atomic per sum and wave, compared
// by one thread a barrier later.  A row outside the tolerance sets bit 0 of its window's status word (k_peak copies it into
// twx_result.status).  This is the detector for the fault class of profiles/r05_fir_mfma.txt: whole rows of this pass going wrong while
// some other kernel's waves are resident beside it, with nothing in either source to show for it.
template <class P2, typename T, int MODE, int NT, bool CHK = false>
__global__ __launch_bounds__(NT, (MODE == ROW_MID && sizeof(T) == 4 ? 4 : 1)) void k_rowd(RowDArgs<T> ad) {
    using C = cpx<T>;
    using D = RowD<P2, T>;
    static_assert(!CHK || MODE == ROW_MID, "the self-check belongs to the middle pass");
    const RowArgs<T>& a = ad.r;
    constexpr int N2 = D::L, R = D::R, R0 = D::R0, M = D::M, NU = R0 * R;
    constexpr int RMAX = R > R0 ? R : R0;
    // MID with the folded output twiddle and the uniform factors in scalar registers reads neither s_vc nor s_eb
    constexpr bool NEED_SVC = MODE == ROW_MID && !(TWX_MID_SGPR && TWX_MID_FOLD && R0 > 1);
    constexpr bool NEED_SEB = MODE == ROW_MID && !(TWX_MID_SGPR && R % 2 == 0);
    constexpr int NEB = NEED_SEB ? TWX_MAX_PHASE * 2 * R : 0;
    constexpr int NVC = NEED_SVC ? R0 : 0;
    static_assert(NT >= D::NT_MIN, "not enough threads for RowD");
    // Row-walking workgroups (MID, fp32, folded tables): the launch has a few workgroups per CU and each walks the rows
    // g, g + grid, g + 2 grid, ... — the tables go to LDS once per workgroup, and the next row's A is loaded into the registers
    // of the product pr[], dead during the last phase, while that phase runs: the wait for the row (30 % of a one-row
    // workgroup's life, tools/stamps_rowd.py) is gone.  What remains is the kernel's two balanced halves: its memory traffic
    // alone takes 0.26 ms, its loads + arithmetic without the stores 0.245 ms, together 0.326 ms (profiles/r03_rowd_resident.txt).
    constexpr bool PERSIST = MODE == ROW_MID && rowd_mid_resident<P2, T>();
    // (The same walk for the BAND pass — resident workgroups, the next row asked for as soon as the current one has left its landing
    // registers — was built in round 6 and lost: 0.1026 against 0.0911 ms per 8 windows, profiles/r06_colinv.txt.)
    constexpr int NTC = PERSIST ? R0 * R : 0;
    // data and tables are SEPARATE shared arrays: with one array the compiler must assume that a table read may
    // alias an earlier data write and serialises read -> wait -> multiply -> write for every element of a stage
    __shared__ C lds[D::lds_elems];
    __shared__ C tabs[D::tab_total + NEB + NVC + NTC + 16];
    C* s_eb = tabs + D::tab_total;
    C* s_vc = s_eb + NEB;
    C* s_tc = s_vc + NVC;                   // PERSIST: the forward table tc as loaded (tabs[tab_c..] is folded per row and restored from here)
    void* red = (void*)(s_tc + NTC);
    __shared__ float s_chk[CHK ? 2 * TWX_CHK_SLOTS : 1];     // [row parity][sum]: a row's sums are compared (and cleared) one barrier into the next row
    const unsigned total = PERSIST ? ad.total_rows : gridDim.x;
    const int tid0 = threadIdx.x;
    if constexpr (CHK) { if (tid0 < 2 * TWX_CHK_SLOTS) s_chk[tid0] = 0.f; }
    int chk_par = 0, chk_prev_row = -1;
    // a row's sums leave for global memory one workgroup barrier into the next row (every wave reached that barrier behind its atomics);
    // k_chk_verdict compares them — nothing of the comparison costs this kernel a register
    auto chk_flush = [&](int par, int row) {
        if constexpr (CHK) {
            if (tid0 < TWX_CHK_SLOTS) {
                ad.chk_rows[(long long)row * TWX_CHK_SLOTS + tid0] = s_chk[par * TWX_CHK_SLOTS + tid0];
                s_chk[par * TWX_CHK_SLOTS + tid0] = 0.f;
            }
        }
    };
    TWX_STAMP(30);
    C v[RMAX];
    C csr[MODE == ROW_MID ? R : 1];
    C pr[MODE == ROW_MID ? RMAX : 1];
    // the row loads are unconditional (idle lanes of the last wave re-read a valid element): inside divergent branches
    // the wait-count pass has to assume the branch was skipped and waits for far more than the tables
    auto load_row = [&](unsigned vb, int tl, C* dst) {
        const unsigned lg = xcd_remap(vb, total);
        const int rk1 = lg / a.nwin, rb = lg % a.nwin;
        const C* Ab = a.A + (long long)rb * a.n;
        if constexpr (M % 16 == 0) {
            // M is a multiple of every tile width (W <= 16): element tl + r*M of the row sits r*M*N1 elements after element
            // tl — one a_index per thread (its integer multiply runs at quarter rate) instead of one per element, the
            // r-dependent part is scalar: SGPR base + scalar offset + 32-bit lane offset
            const unsigned lb = a_index((unsigned)tl, (unsigned)rk1, (unsigned)a.n1, a.wshift) * (unsigned)sizeof(C);
            const unsigned long long ab = sgpr_u64(reinterpret_cast<unsigned long long>(Ab));
            const unsigned long long rstep = sgpr_u64((unsigned long long)M * (unsigned long long)a.n1 * sizeof(C));
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) dst[r] = ld_pin<C, TWX_NT_A != 0>(ab, r * rstep, lb);       // A is read exactly once
        } else {
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) {
                const C* q = Ab + a_index((unsigned)(tl + r * M), (unsigned)rk1, (unsigned)a.n1, a.wshift);
                dst[r] = TWX_NT_A ? __builtin_nontemporal_load(q) : *q;
            }
        }
    };
    {
        // small tables first (they are needed first and loads return in order), then the row, then the code spectrum.
        // (Computing stage 0's tables in the kernel instead of loading them was tried in round 3 and changed nothing: what the
        // first 18 k cycles of a workgroup wait for is its 64-KB row at the CU's share of the HBM bandwidth, not a round trip.)
        constexpr int NTAB = (D::tab_total + NT - 1) / NT;
        C treg[NTAB];
        TWX_UNROLL
        for (int k = 0; k < NTAB; ++k) treg[k] = ad.dtabs[min(tid0 + k * NT, D::tab_total - 1)];     // clamped: no branch, no select
        C tc0 = mk<T>(1, 0);
        if constexpr (PERSIST) tc0 = ad.dtabs[D::tab_c + min(tid0, R0 * R - 1)];
        C ebreg = mk<T>(0, 0);
        if constexpr (NEED_SEB) {
            static_assert(!NEED_SEB || TWX_MAX_PHASE * 2 * R <= NT, "phase-ramp table larger than the workgroup");
            ebreg = ad.eb_d[min(tid0, a.nphase * 2 * R - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the table loads first in program order (loads return in order)
        if constexpr (PERSIST) load_row(blockIdx.x, min(tid0, M - 1), pr);      // the first row arrives where every later one does
        else load_row(blockIdx.x, min(tid0, M - 1), v);
        __builtin_amdgcn_sched_barrier(0);
        TWX_UNROLL
        for (int k = 0; k < NTAB; ++k) { const int i = tid0 + k * NT; if (i < D::tab_total) tabs[i] = treg[k]; }
        if constexpr (NEED_SEB) { if (tid0 < a.nphase * 2 * R) s_eb[tid0] = ebreg; }
        if constexpr (PERSIST) {
            if (tid0 < R0 * R) s_tc[tid0] = tc0;
            // The first row is waited for HERE, with a wait the compiler sees: entering the row loop with these loads pending,
            // the wait-count pass merges "the 20 youngest memory operations" with the steady state "20 loads, then the last
            // phase's 20 stores" and makes every row's first butterfly wait for the previous row's stores (vmcnt(4..0)).
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt and lgkmcnt untouched
        }
    }

    unsigned vb = blockIdx.x;
    do {                                                   // one trip unless PERSIST
    TWX_STAMP(0);
    int tid = tid0;
    if constexpr (PERSIST) asm volatile("" : "+v"(tid));   // per-row address arithmetic stays inside the row loop (hoisted, it is spilled)
    int q0, qi;
    const bool act = D::blk_map(tid, q0, qi);
    const int u = q0 * R + qi;
    const unsigned mask = (1u << a.tshift) - 1u;
    const unsigned logical = xcd_remap(vb, total);
    const int k1 = logical / a.nwin, b = logical % a.nwin;
    const bool has_next = PERSIST && vb + gridDim.x < total;
    C vca = mk<T>(1, 0), vcb = mk<T>(1, 0);
    C fa = mk<T>(1, 0), fb = mk<T>(1, 0);
    if constexpr (PERSIST) {
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) v[r] = pr[r];
    }
    if constexpr (CHK && (TWX_CHK_PARTS & 1)) {
        float sa = 0.f;
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) sa += (float)cnorm(v[r]);
        chk_add(s_chk + chk_par * TWX_CHK_SLOTS + 0, tid < M ? sa : 0.f);      // (the idle lanes of the last wave hold a copy of element M-1)
    }
    // the code spectrum of the row: needed after the forward transform.  One trip: issued with the row's start.  Resident
    // workgroups: after stage 0 has put its outputs into LDS, when v[] is free — at the row's start the butterfly's temporaries,
    // v[] and 40 landing registers do not fit into 128 (the L2 round trip is covered by the barrier and two stages either way)
    auto load_cspec = [&]() {
        const C* cs = ad.cspec_perm + (long long)k1 * N2;
        const unsigned ulb = (unsigned)min(u, NU - 1) * (unsigned)sizeof(C);
        const unsigned long long csb = sgpr_u64(reinterpret_cast<unsigned long long>(cs));
        TWX_UNROLL
        for (int q2 = 0; q2 < (MODE == ROW_MID ? R : 0); ++q2) csr[q2] = ld_pin<C, false>(csb, (unsigned long long)q2 * NU * sizeof(C), ulb);
    };
    if constexpr (MODE == ROW_MID) {
        if constexpr (NEED_SVC) {
            const unsigned m = (unsigned)k1 * (unsigned)min(tid, R0 - 1) * (unsigned)M;  // conj(W_N^{k1 * c * M})
            vca = a.ta[m >> a.tshift]; vcb = a.tb[m & mask];
        }
        if constexpr (TWX_MID_FOLD && R0 > 1) {
            // the output twiddle W_N^{-k1 t}, t = a + R b, split into the part of stage B's thread (a = qi) and the part
            // that goes into the table of stage B (b = tid mod R): one load each from the per-row table, with the other
            // loads of the row's start
            fb = ad.vw[((long long)k1 * 2 + 1) * R + tid % R];
        }
        if constexpr (!PERSIST) load_cspec();
        if constexpr (NEED_SVC) {
            __syncthreads();                               // (previous row's stage C has read s_vc)
            if (tid < R0) s_vc[tid] = cconj(cmul(vca, vcb));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PERSIST && TWX_ABLR == 2) {              // diagnostic: the kernel's memory traffic without its arithmetic
        for (int rho = 0; rho < a.nphase; ++rho) {
            if (has_next && rho == a.nphase - 1) load_row(vb + gridDim.x, min(tid, M - 1), pr);
            C* out = a.Bz + ((long long)b * a.nphase + rho) * a.n + (long long)k1 * N2;
            const unsigned ltb = (unsigned)min(tid, M - 1) * (unsigned)sizeof(C);
            const unsigned long long ob = sgpr_u64(reinterpret_cast<unsigned long long>(out));
            TWX_UNROLL
            for (int c = 0; c < R0; ++c) st_pin<C, TWX_NT_BZ != 0>(ob, (unsigned long long)c * M * sizeof(C), ltb, v[c]);
        }
        continue;
    }
    if constexpr (PERSIST) { if (tid < M) Bfly<T, R0, false>::run(v); }     // register work ahead of the barrier
    TWX_STAMP(1);
    // first row: tables visible (the row loads are in flight meanwhile); later rows: the previous row's stage C has read every
    // block, and the restored table tc is visible
    __syncthreads();
    if constexpr (CHK) { if (chk_prev_row >= 0) chk_flush(chk_par ^ 1, chk_prev_row); }
    TWX_STAMP(2);
    if (tid < M) {
        if constexpr (!PERSIST) Bfly<T, R0, false>::run(v);
        D::f0_twiddle_store(lds, tabs, tid, v);
    }
    if constexpr (PERSIST) load_cspec();
    TWX_STAMP(3);
    __syncthreads();                                   // all-to-all exchange of the stride-M stage
    TWX_STAMP(4);
    if constexpr (MODE == ROW_MID && TWX_MID_FOLD && R0 > 1) {
        // stage 0 was the last reader of the forward table tc: turn it into stage B's table, conj(tc[q0][b]) * W_N^{-k1 R b}
        // (visible to every wave after the barrier at the top of the first phase)
        if (tid < R0 * R) tabs[D::tab_c + tid] = cmulc(fb, PERSIST ? s_tc[tid] : tabs[D::tab_c + tid]);
    }
#ifdef TWX_ROWD_CHECK
    // diagnostic (profiles/r05_fir_mfma.txt): ONE wave runs the SAME instructions (a function that is not inlined) twice on the same inputs:
    // do the results agree?  And does it read the same LDS words twice with the same result?
    if constexpr (MODE == ROW_MID && sizeof(T) == 4) {
        C va[R], vb[R], oa[R], ob[R];
        int m1 = 0, m2 = 0;
        if (act) { TWX_UNROLL for (int r = 0; r < R; ++r) va[r] = lds[D::phys(q0, r, qi)]; }
        wave_sync_lds();
        if (act) {
            TWX_UNROLL for (int r = 0; r < R; ++r) vb[r] = lds[D::phys(q0, r, qi)];
            TWX_UNROLL for (int r = 0; r < R; ++r) m1 += (va[r].x != vb[r].x) | (va[r].y != vb[r].y);
            rowd_check_bfly<T, R>(va, oa);
            rowd_check_bfly<T, R>(va, ob);
            TWX_UNROLL for (int r = 0; r < R; ++r) m2 += (__float_as_uint(oa[r].x) != __float_as_uint(ob[r].x)) | (__float_as_uint(oa[r].y) != __float_as_uint(ob[r].y));
            int m3 = 0;
            TWX_UNROLL for (int r = 0; r + 3 < R; r += 4) {
                const RowdCheck4 x = rowd_check4(va[r], va[r + 1], va[r + 2], va[r + 3], r);
                const RowdCheck4 y = rowd_check4(va[r], va[r + 1], va[r + 2], va[r + 3], r + 1);
                m3 += (__float_as_uint(x.a.x) != __float_as_uint(y.a.x)) | (__float_as_uint(x.a.y) != __float_as_uint(y.a.y)) | (__float_as_uint(x.b.x) != __float_as_uint(y.b.x)) |
                      (__float_as_uint(x.b.y) != __float_as_uint(y.b.y)) | (__float_as_uint(x.c.x) != __float_as_uint(y.c.x)) | (__float_as_uint(x.c.y) != __float_as_uint(y.c.y)) |
                      (__float_as_uint(x.d.x) != __float_as_uint(y.d.x)) | (__float_as_uint(x.d.y) != __float_as_uint(y.d.y));
            }
            if (m1 | m2 | m3) printf("k_rowd check: k1 %d q0 %d lane-row %d: %d words read differently twice, %d butterfly outputs differ between two calls of one function, %d register-only calls differ\n", k1, q0, qi, m1, m2, m3);
        }
    }
#endif
    if (act) D::f1(lds, tabs, q0, qi, v);
    if constexpr (CHK) {       // diagnostic fault: one element of the row between the forward stages, as a wrong butterfly output would leave it
        if (ad.chk_fault == 2 * ((int)logical + 1) && tid == 0) { wave_sync_lds(); lds[D::phys(0, 1, 0)] = cscale(lds[D::phys(0, 1, 0)], T(1.5)); }
    }
    TWX_STAMP(5);
    bool pruned = false;
    if constexpr (MODE == ROW_BAND) pruned = ad.nprune > 0;
    if (!pruned) {
        wave_sync_lds();
        if (act) D::f2(lds, q0, qi, v);                // v[q2] = X[k1 + N1*(q0 + R0 qi + R0 R q2)]
    }

    if constexpr (MODE == ROW_BAND) {
        Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
        if (pruned) {
            __syncthreads();                           // stage f1 of every block visible to the whole workgroup
            if (tid < R0 * ad.nprune) {
                const int pq0 = tid % R0, p = tid / R0;
                const int q1 = (int)((ad.pr_q1 >> (8 * p)) & 0xffu), q2 = (int)((ad.pr_q2 >> (8 * p)) & 0xffu);
                C acc = lds[D::phys(pq0, q1, 0)];
                int e = 0;
                TWX_UNROLL
                for (int i2 = 1; i2 < R; ++i2) {
                    e += q2; if (e >= R) e -= R;       // (i2 * q2) mod R
                    acc = acc + cmul(lds[D::phys(pq0, q1, i2)], ad.wr[e]);
                }
                const long long half = a.n / 2;
                const long long k = (long long)k1 + (long long)a.n1 * D::k_of(pq0, q1, q2);
                long long i = k - (a.n - half); if (i < 0) i += a.n;
                if (i >= a.band_lo && i <= a.band_hi) best.take(cnorm(acc), (unsigned int)i);
            }
        } else
        if (act) {
            const long long half = a.n / 2;
            TWX_UNROLL
            for (int q2 = 0; q2 < R; ++q2) {
                const long long k = (long long)k1 + (long long)a.n1 * D::k_of(q0, qi, q2);
                long long i = k - (a.n - half); if (i < 0) i += a.n;
                if (i >= a.band_lo && i <= a.band_hi) best.take(cnorm(v[q2]), (unsigned int)i);
            }
        }
        best = block_best<T, NT>(best, red);
        if (tid == 0) { ArgPart<T> p; p.val = best.val; p.idx = best.idx; a.part[(long long)b * a.n1 + k1] = p; }
    } else {
        constexpr bool FOLD = TWX_MID_FOLD && R0 > 1;
        constexpr bool USGPR = TWX_MID_SGPR != 0;
        C ub = mk<T>(1, 0);
        C wa0 = mk<T>(1, 0);
        if constexpr (!FOLD) {
            // in EVERY lane, the idle ones of the last wave with lane M-1's factor: they repeat that lane's store in stage C (below),
            // and with a factor of their own they raced it with a different value — element M-1 of every row was wrong in the one
            // instantiation that takes this path (R0 = 1 in fp64, e.g. N2 = 400: found by test_randomised_option_sweep)
            const unsigned m = (unsigned)k1 * (unsigned)min(tid, M - 1);
            ub = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & mask]));
        }
        const C* vcrow = ad.vc + (long long)k1 * R0;                              // wave-uniform address: scalar loads
        if (act && k1 == 0 && u == 0) a.dc[b] = v[0];
        // in every lane (the idle ones of the last wave multiply leftovers): a conditional definition would keep the previous
        // contents of pr[] — the row loaded ahead — alive through the whole forward part
        float chk_sx = 0.f, chk_sp = 0.f;
        TWX_UNROLL
        for (int q2 = 0; q2 < R; ++q2) {
            if constexpr (CHK && (TWX_CHK_PARTS & 2)) chk_sx += (float)cnorm(v[q2]);
            pr[q2] = cmul(v[q2], csr[q2]);      // ffty.*fcode (godual_ranging.m:26)
            if constexpr (CHK && (TWX_CHK_PARTS & 2)) chk_sp += (float)cnorm(pr[q2]);
        }
        if constexpr (CHK && (TWX_CHK_PARTS & 2)) {
            chk_add(s_chk + chk_par * TWX_CHK_SLOTS + 1, act ? chk_sx : 0.f);
            chk_add(s_chk + chk_par * TWX_CHK_SLOTS + 2, act ? chk_sp : 0.f);
        }
        // the thread's part of the folded output twiddle: asked for here, where csr[] has just left its registers, and met
        // after the first inverse butterfly (held from the row's start it was spilled)
        if constexpr (FOLD) fa = ad.vw[(long long)k1 * 2 * R + min(qi, R - 1)];
        // Phase rho's register work (ramp, first inverse butterfly, its twiddles) is done BEFORE the barrier
        // that ends phase rho-1, so waves that finish stage C early spend the wait on arithmetic.
        C r1_cur = USGPR ? ld_uniform(a.ramp1, k1) : a.ramp1[k1];                  // uniform per row: no vector load, no vmcnt
        if (act) {
            TWX_UNROLL
            for (int q2 = 0; q2 < R; ++q2) v[q2] = pr[q2];
            D::iA_pre(tabs, qi, v);
            // stage B's per-thread factor without the phase ramp: conj(W_L^{q0 a}) * W_N^{-k1 a}
            if constexpr (FOLD) wa0 = cmulc(fa, tabs[D::tab_b + q0 * R + qi]);
        }
        TWX_STAMP(6);
        for (int rho = 0; rho < a.nphase; ++rho) {
            int lt = tid;
            asm volatile("" : "+v"(lt));               // keep address arithmetic inside the loop (see k_row)
            int lq0, lqi;
            const bool lact = D::blk_map(lt, lq0, lqi);
            const C r1 = r1_cur;
            C eaj = mk<T>(1, 0);
            if (rho + 1 < a.nphase) {                  // next phase's ramp factors: in flight during this phase
                r1_cur = USGPR ? ld_uniform(a.ramp1, (rho + 1) * a.n1 + k1) : a.ramp1[(long long)(rho + 1) * a.n1 + k1];
                eaj = ad.ea_d[(rho + 1) * NU + min(lq0 * R + lqi, NU - 1)];      // every lane (see the stores of stage C below)
            }
            if (rho > 0 || FOLD) __syncthreads();      // previous phase's stage C has read every block (rho = 0: the folded table is complete)
            TWX_STAMP(7 + rho * 6);
            if constexpr (PERSIST) {
                // the product pr[] was read for the last time when this phase's inputs were formed: its registers take the
                // next row's A, in flight for the whole phase and issued BEFORE this phase's stores (loads and stores share
                // vmcnt and return in order: the wait at the top of the next row does not include a store round trip)
                if (has_next && rho == a.nphase - 1) load_row(vb + gridDim.x, min(lt, M - 1), pr);
            }
            if (lact) D::iA_store(lds, lq0, lqi, v);
            wave_sync_lds();
            TWX_STAMP(8 + rho * 6);
            if constexpr (FOLD) { if (lact) D::iB_folded(lds, tabs, lq0, lqi, cmul(wa0, r1), v); }
            else if (lact) D::iB(lds, tabs, lq0, lqi, v);
            if constexpr (CHK) {
                if (ad.chk_fault == 2 * ((int)logical + 1) + 1 && rho == a.nphase - 1 && lt == 0) { wave_sync_lds(); lds[D::phys(0, 1, 0)] = cscale(lds[D::phys(0, 1, 0)], T(1.5)); }
            }
            TWX_STAMP(9 + rho * 6);
            __syncthreads();
            TWX_STAMP(10 + rho * 6);
            if constexpr (PERSIST) {
                // every wave is past stage B, the last reader of the folded table: back to the forward table for the next row
                if (has_next && rho == a.nphase - 1 && lt < R0 * R) tabs[D::tab_c + lt] = s_tc[lt];
            }
            {
                // In EVERY lane: the 48 idle lanes of the last wave repeat the work and the stores of lane M-1 (same values to the
                // same addresses).  Inside a divergent branch the wait-count pass must assume the stores were skipped, and every
                // later wait for a load issued before them (the next phase's ramp factor, the next row) becomes vmcnt(0): a store
                // round trip per phase.  Unconditional, they are counted: vmcnt(20).
                constexpr bool ST16 = TWX_MID_ST16 && FOLD && sizeof(T) == 4 && R0 % 2 == 0 && M % 2 == 0;
                static_assert(!(CHK && ST16), "the self-check sums the plain store path");
                // ST16: the idle lanes repeat the PAIR (M-2, M-1), so that every lane pair exchanges like a live one
                const int ltc = ST16 ? (lt < M ? lt : M - 2 + (lt & 1)) : min(lt, M - 1);
                D::iC(lds, ltc, v);
                if constexpr (CHK && (TWX_CHK_PARTS & 4)) {
                    // the phase's outputs before their last factor (modulus one): |Bz row|^2 without touching the store loop's registers
                    float sb = 0.f;
                    TWX_UNROLL
                    for (int c = 0; c < R0; ++c) sb += (float)cnorm(v[c]);
                    chk_add(s_chk + chk_par * TWX_CHK_SLOTS + 3 + rho, lt < M ? sb : 0.f);     // (idle lanes repeat lane M-1's work)
                }
                const C uu = cmul(ub, r1);
                C* out = a.Bz + ((long long)b * a.nphase + rho) * a.n + (long long)k1 * N2;
                const unsigned ltb = (unsigned)ltc * (unsigned)sizeof(C);
                const unsigned long long ob = sgpr_u64(reinterpret_cast<unsigned long long>(out));
                if constexpr (ST16) {
                    // Lane t (even) and t+1 hold z[t + M c], z[t+1 + M c] for every c.  For the row pair (c, c+1) the even lane takes
                    // its neighbour's value of row c and stores {z[t], z[t+1]} of row c as ONE 16-byte store, the odd lane the same
                    // for row c+1: half the store instructions, each a full 128-byte line per eight lanes.  The exchange is a DPP
                    // quad_perm [1,0,3,2] move (full-rate VALU, no LDS): each lane sends what its neighbour stores.
                    const bool odd = (ltc & 1) != 0;
                    const unsigned ltb16 = (unsigned)(ltc & ~1) * (unsigned)sizeof(C);
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    TWX_UNROLL
                    for (int c = 0; c < R0; c += 2) {
                        const C oe = USGPR ? cmul_us(v[c], ld_uniform(vcrow, c)) : cmul(v[c], s_vc[c]);
                        const C oo = USGPR ? cmul_us(v[c + 1], ld_uniform(vcrow, c + 1)) : cmul(v[c + 1], s_vc[c + 1]);
                        const float sx = odd ? (float)oe.x : (float)oo.x, sy = odd ? (float)oe.y : (float)oo.y;
                        const float rx = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sx), 0xB1, 0xF, 0xF, true));
                        const float ry = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sy), 0xB1, 0xF, 0xF, true));
                        f4 w;
                        w.x = odd ? rx : (float)oe.x; w.y = odd ? ry : (float)oe.y;
                        w.z = odd ? (float)oo.x : rx; w.w = odd ? (float)oo.y : ry;
                        // even lane: row c; odd lane: row c + 1 (the row offset is a per-lane quantity here: one VALU add)
                        const unsigned rowoff = (unsigned)((c + (odd ? 1 : 0)) * M) * (unsigned)sizeof(C);
                        if (TWX_ABLR != 1) st_pin<f4, TWX_NT_BZ != 0>(ob, 0ull, ltb16 + rowoff, w);
                    }
                } else {
                TWX_UNROLL
                for (int c = 0; c < R0; ++c) {
                    C o;                                                                       // · W_N^{-k1 (t + c M)} · ramp1
                    if constexpr (FOLD) o = USGPR ? cmul_us(v[c], ld_uniform(vcrow, c)) : cmul(v[c], s_vc[c]);
                    else o = USGPR ? cmul3_us(v[c], uu, ld_uniform(vcrow, c)) : cmul3(v[c], uu, s_vc[c]);
#if TWX_BZ16
                    if constexpr (sizeof(T) == 4) {
                        // the same element index in a buffer of 4-byte elements: base, row offset and lane offset all halve
                        const unsigned long long ob16 = sgpr_u64(reinterpret_cast<unsigned long long>(a.Bz) +
                                                                 (unsigned long long)((((long long)b * a.nphase + rho) * a.n + (long long)k1 * N2) * 4));
                        st_pin<unsigned, TWX_NT_BZ != 0>(ob16, (unsigned long long)c * M * 4, ltb / 2, bz16_pack((float)o.x, (float)o.y));
                    } else
#endif
                    if (TWX_ABLR != 1) st_pin<C, TWX_NT_BZ != 0>(ob, (unsigned long long)c * M * sizeof(C), ltb, o);     // SGPR base + lane offset
                }
                }
            }
            TWX_STAMP(11 + rho * 6);
            if (rho + 1 < a.nphase && lact) {
                const int rn = rho + 1;
                TWX_UNROLL
                for (int q2 = 0; q2 < R; ++q2) {
                    // from the LDS copy (or, uniform, with scalar loads), not with vector loads from global memory: a vector load
                    // issued after this phase's Bz stores can only be waited for together with them (loads and stores share
                    // vmcnt) — a store round trip per phase
                    if constexpr (R % 2 == 0) {
                        if constexpr (USGPR) v[q2] = cmul3_us(pr[q2], eaj, ld_uniform(ad.eb_d, (rn * 2 + (q2 >= R / 2 ? 1 : 0)) * R + q2));
                        else v[q2] = cmul3(pr[q2], eaj, s_eb[(rn * 2 + (q2 >= R / 2 ? 1 : 0)) * R + q2]);      // uniform address: LDS broadcast
                    } else v[q2] = cmul3(pr[q2], eaj, s_eb[(rn * 2 + ((2 * D::k_of(lq0, lqi, q2) >= N2) ? 1 : 0)) * R + q2]);
                }
                D::iA_pre(tabs, lqi, v);
            }
            TWX_STAMP(12 + rho * 6);
        }
    }
    if constexpr (CHK) { chk_prev_row = b * a.n1 + k1; chk_par ^= 1; }
    } while (PERSIST && (vb += gridDim.x) < total);        // rows of this workgroup
    if constexpr (CHK) {
        __syncthreads();                                   // the last row's sums are complete
        if (chk_prev_row >= 0) chk_flush(chk_par ^ 1, chk_prev_row);
    }
    TWX_STAMP(31);
}

// TWX_OPT_SELFCHECK, second half: Parseval's identity per row from the sums k_rowd<MID, CHK> left — |A row|^2 N2 = |spectrum row|^2, and
// |product row|^2 N2 = |Bz row|^2 for every phase.  A row outside the tolerance (or not a number) sets bit 0 of its window's flag word;
// stat[0] keeps the largest relative deviation the context has seen (float bits), stat[1] counts the rows flagged.   grid = windows
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_chk_verdict(const float* __restrict__ rows, int n1, int n2, int nphase, float tol, int* __restrict__ flag,
                                                    unsigned* __restrict__ stat) {
    const int b = blockIdx.x;
    float worst = 0.f; int nbad = 0;
    for (int k1 = threadIdx.x; k1 < n1; k1 += 256) {
        const float* c = rows + ((long long)b * n1 + k1) * TWX_CHK_SLOTS;
        const float fn = (float)n2;
        float dev;
        if (c[0] == 0.f && c[1] == 0.f && c[2] == 0.f) {           // an all-zero row stays zero
            dev = 0.f;
            for (int rho = 0; rho < nphase; ++rho) dev = fmaxf(dev, c[3 + rho]);
        } else {
            dev = fabsf(c[1] - fn * c[0]) / (fn * c[0]);
            for (int rho = 0; rho < nphase; ++rho) { const float d = fabsf(c[3 + rho] - fn * c[2]) / (fn * c[2]); dev = (d > dev || !(d == d)) ? d : dev; }
        }
        if (!(dev <= tol)) ++nbad;
        if (!(dev == dev)) dev = __uint_as_float(0x7f800000u);
        worst = fmaxf(worst, dev);
    }
    __shared__ float s_w[256]; __shared__ int s_n[256];
    s_w[threadIdx.x] = worst; s_n[threadIdx.x] = nbad;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { s_w[threadIdx.x] = fmaxf(s_w[threadIdx.x], s_w[threadIdx.x + d]); s_n[threadIdx.x] += s_n[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        flag[b] = s_n[0] ? 1 : 0;
        atomicMax(stat, __float_as_uint(s_w[0]));
        if (s_n[0]) atomicAdd(stat + 1, (unsigned)s_n[0]);
    }
}

// ------------------------------------------------------------------------------------------
// k_rowd_bandsum: the row pass of the carrier search (fft(d.^2) restricted to the search band, godual_ranging.m:13-16) where the band touches
// only a few (q1, q2) digit pairs of the row bins k2 = q0 + R0 (q1 + R q2) — the pruned case of k_rowd<BAND>: +-20 kHz of a 5-Msps second is
// |k2| <= 32 of 8000, four pairs — WITHOUT the row in LDS.  After the radix-R0 butterfly and its twiddle, thread t holds Y_t[q0], and
//     X[q0 + R0 kappa] = sum_t Y_t[q0] W_M^{t kappa},   kappa = q1 + R q2,   M = R R,
// is a sum over the workgroup's threads for each of the few kappa: R0 products per thread and kappa, summed over the wave by a
// reduce-scatter on v_permlane32_swap / v_permlane16_swap (each halves the values a lane carries) and DPP inside the rows of 16 lanes,
// over the waves through 9 KB of LDS.  k_rowd<BAND> holds the 64-KB row in LDS (two workgroups per CU, each waiting for its row, then
// computing: 3.5 TB/s); this form keeps 15 KB: with two tasks per thread (the default: 256 threads, 124 registers) four workgroups per CU have
// their rows in flight (4.6 TB/s; one task per thread: 71 registers, three workgroups, slower — the reduction is the larger half of the arithmetic).
// fp32, even R0; everything else takes k_rowd<BAND>.  (Complex double was built and dropped: a thread's 20 values are 80 registers before the
// butterfly's temporaries — one task per thread spilled 136 registers at the 128 that two workgroups per CU allow, two tasks 149 at 256 — and
// with ONE workgroup per CU, which is what k_rowd<BAND, double> has, the row's wait is not hidden either way; profiles/r06_colinv.txt item 7.)
// ------------------------------------------------------------------------------------------
#define TWX_BANDSUM_MAXP 8
#ifndef TWX_BANDSUM_ABL
#define TWX_BANDSUM_ABL 0
#endif
__device__ __forceinline__ float twx_row16_sum(float v) {      // every lane of a row of 16 ends with the row's sum
    v = TWX_DPP_ADD(v, 0xB1);       // quad_perm [1,0,3,2]
    v = TWX_DPP_ADD(v, 0x4E);       // quad_perm [2,3,0,1]
    v = TWX_DPP_ADD(v, 0x141);      // row_half_mirror
    v = TWX_DPP_ADD(v, 0x140);      // row_mirror
    return v;
}
// CH complex values per lane -> CH/2 floats per lane: rows 0 / 1 of 16 lanes hold the wave's sums of re[i] / re[i + CH/2], rows 2 / 3 of im[i] / im[i + CH/2]
template <int CH> __device__ __forceinline__ void wave_sum_scatter(const cpx<float>* z, float* out) {
    static_assert(CH % 2 == 0, "pairs of values");
    float s[CH];
    TWX_UNROLL
    for (int i = 0; i < CH; ++i) {
        // rows 2, 3 of the first operand <-> rows 0, 1 of the second: lanes < 32 then add re of lanes l and l + 32, lanes >= 32 the same for im
        // (operands and results go through named scalars: this clang evaluates __builtin_bit_cast of a vector ELEMENT expression — z[i].y, r[1] —
        // as a cast of element 0)
        const float zx = z[i].x, zy = z[i].y;
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, zx), __builtin_bit_cast(unsigned, zy), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        s[i] = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    }
    TWX_UNROLL
    for (int i = 0; i < CH / 2; ++i) {
        // odd rows of the first operand <-> even rows of the second: row 0 = s[i] rows 0 + 1, row 1 = s[i + CH/2] rows 0 + 1, rows 2, 3 the same of rows 2 + 3
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s[i]), __builtin_bit_cast(unsigned, s[i + CH / 2]), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        out[i] = twx_row16_sum(__builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1));
    }
}
template <int TILES, int TPT> struct BandsumDeal {
    static constexpr int cnt(int j) { return TILES / TPT + (j < TILES % TPT ? 1 : 0); }
    static constexpr int off(int j) { int o = 0; for (int i = 0; i < j; ++i) o += cnt(i); return o; }
    static constexpr int threads = ((cnt(0) * 16 + 63) / 64) * 64;
};
#ifndef TWX_BANDSUM_TPT
#define TWX_BANDSUM_TPT 2
#endif
template <class P2> constexpr bool rowd_bandsum_ok() {      // (asked of plans that have a RowD form)
    using D = RowD<P2, float>;
    return P2::S == 3 && D::R0 % 2 == 0 && D::M % 16 == 0 && D::M / 16 >= TWX_BANDSUM_TPT;
}
template <class P2, typename T, int TPT>
__global__ __launch_bounds__((BandsumDeal<RowD<P2, T>::M / 16, TPT>::threads), (TPT == 1 ? 6 : TPT == 2 ? 4 : 2)) void k_rowd_bandsum(RowDArgs<T> ad) {
    using C = cpx<T>;
    using D = RowD<P2, T>;
    static_assert(sizeof(T) == 4 && D::R0 % 2 == 0, "fp32, even first radix");
    const RowArgs<T>& a = ad.r;
    constexpr int R = D::R, R0 = D::R0, M = D::M;
    using Deal = BandsumDeal<M / 16, TPT>;
    constexpr int NT = Deal::threads, NW = NT / 64;
    constexpr int CH = (R0 % 4 == 0) ? R0 / 2 : R0, NCH = R0 / CH;      // values per reduce-scatter round (20 registers at R0 = 20)
    constexpr int NTW = 2 * R0 * R, NLT = (NTW + NT - 1) / NT, NLM = (M + NT - 1) / NT;
    static_assert(M % 16 == 0, "whole tiles of 16 tasks");
    __shared__ C s_tw[NTW];                                        // tb | tc: W_L^{q0 i2}, W_L^{q0 R i1}
    __shared__ C s_wm[M];                                          // W_M^j
    __shared__ float s_part[NW * TWX_BANDSUM_MAXP * R0 * 2];       // [wave][pair][q0][re, im]
    __shared__ unsigned long long s_red[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int k1 = logical / a.nwin, b = logical % a.nwin;
    // tables first (loads return in order), then the row
    C treg[NLT], mreg[NLM];
    TWX_UNROLL
    for (int k = 0; k < NLT; ++k) treg[k] = ad.dtabs[D::tab_b + min(tid + k * NT, NTW - 1)];
    TWX_UNROLL
    for (int k = 0; k < NLM; ++k) mreg[k] = ad.wm[min(tid + k * NT, M - 1)];
    __builtin_amdgcn_sched_barrier(0);
    C v[TPT][R0];
    int tj[TPT];
    bool live[TPT];
    {
        const C* Ab = a.A + (long long)b * a.n;
        const unsigned long long ab = sgpr_u64(reinterpret_cast<unsigned long long>(Ab));
        const unsigned long long rstep = sgpr_u64((unsigned long long)M * (unsigned long long)a.n1 * sizeof(C));
        TWX_UNROLL
        for (int j = 0; j < TPT; ++j) {
            live[j] = tid < 16 * Deal::cnt(j);
            tj[j] = 16 * Deal::off(j) + min(tid, 16 * Deal::cnt(j) - 1);      // idle lanes copy the slot's last task (their factor is zero)
            const unsigned lb = a_index((unsigned)tj[j], (unsigned)k1, (unsigned)a.n1, a.wshift) * (unsigned)sizeof(C);
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) v[j][r] = ld_pin<C, TWX_NT_A != 0>(ab, r * rstep, lb);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    TWX_UNROLL
    for (int k = 0; k < NLT; ++k) s_tw[min(tid + k * NT, NTW - 1)] = treg[k];      // (the clamped duplicates write the value that is there)
    TWX_UNROLL
    for (int k = 0; k < NLM; ++k) s_wm[min(tid + k * NT, M - 1)] = mreg[k];
    __syncthreads();
    TWX_UNROLL
    for (int j = 0; j < TPT; ++j) {
        Bfly<T, R0, false>::run(v[j]);
        const int i1 = tj[j] / R, i2 = tj[j] % R;
        TWX_UNROLL
        for (int q0 = 1; q0 < R0; ++q0) v[j][q0] = cmul3(v[j][q0], s_tw[q0 * R + i2], s_tw[R0 * R + q0 * R + i1]);
    }
    const int np = TWX_BANDSUM_ABL == 2 ? 1 : ad.nprune;      // (diagnostic 2: one pair only)
    for (int p = 0; p < np; ++p) {
        const unsigned kap = (unsigned)((ad.pr_q1 >> (8 * p)) & 0xffu) + (unsigned)R * (unsigned)((ad.pr_q2 >> (8 * p)) & 0xffu);
        C g[TPT];
        TWX_UNROLL
        for (int j = 0; j < TPT; ++j) {
            g[j] = s_wm[((unsigned)tj[j] * kap) % (unsigned)M];
            if (!live[j]) g[j] = mk<T>(0, 0);
        }
        TWX_UNROLL
        for (int c = 0; c < NCH; ++c) {
            C z[CH];
            float o[CH / 2];
            TWX_UNROLL
            for (int i = 0; i < CH; ++i) {
                z[i] = cmul(v[0][c * CH + i], g[0]);
                TWX_UNROLL
                for (int j = 1; j < TPT; ++j) z[i] = z[i] + cmul(v[j][c * CH + i], g[j]);
            }
#if TWX_BANDSUM_ABL == 1      // diagnostic: the products without the reduction over the wave
            TWX_UNROLL
            for (int i = 0; i < CH / 2; ++i) o[i] = z[i].x + z[i + CH / 2].y;
#else
            wave_sum_scatter<CH>(z, o);
#endif
            if ((lane & 15) == 0) {
                const int row = lane >> 4;
                float* dst = s_part + (((wv * TWX_BANDSUM_MAXP + p) * R0 + c * CH + (row & 1) * (CH / 2)) * 2 + (row >> 1));
                TWX_UNROLL
                for (int i = 0; i < CH / 2; ++i) dst[2 * i] = o[i];
            }
        }
    }
    __syncthreads();
    Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
    for (int e = tid; e < R0 * np; e += NT) {
        const int q0 = e % R0, p = e / R0;
        const int q1 = (int)((ad.pr_q1 >> (8 * p)) & 0xffu), q2 = (int)((ad.pr_q2 >> (8 * p)) & 0xffu);
        C acc = mk<T>(0, 0);
        TWX_UNROLL
        for (int w = 0; w < NW; ++w) {
            const float* src = s_part + ((w * TWX_BANDSUM_MAXP + p) * R0 + q0) * 2;
            acc.x += src[0]; acc.y += src[1];
        }
        const long long half = a.n / 2;
        const long long k = (long long)k1 + (long long)a.n1 * D::k_of(q0, q1, q2);
        long long i = k - (a.n - half); if (i < 0) i += a.n;
        if (i >= a.band_lo && i <= a.band_hi) best.take(cnorm(acc), (unsigned int)i);
    }
    best = block_best<T, NT>(best, (void*)s_red);
    if (tid == 0) { ArgPart<T> pp; pp.val = best.val; pp.idx = best.idx; a.part[(long long)b * a.n1 + k1] = pp; }
}

// ------------------------------------------------------------------------------------------
// k_rowd_small: the row pass for SHORT rows N2 = R*R (R0 = 1, e.g. 400 = 20*20: the 5 k ... 25 k chip codes).
// A row is one block of RowD, i.e. the work of R lanes, so one workgroup takes G = (NT/64)*(64/R) rows — one per
// R-lane group — instead of leaving all but R lanes of 448 idle.  After the load phase nothing crosses a wave:
// one workgroup barrier in total.  Same epilogues and the same arithmetic as k_rowd.
// grid = ceil(rows / G), rows = N1 * windows
// ------------------------------------------------------------------------------------------
template <class P2, typename T, int MODE, int NT>
__global__ __launch_bounds__(NT) void k_rowd_small(RowDArgs<T> ad, unsigned int total_rows) {
    using C = cpx<T>;
    using D = RowD<P2, T>;
    static_assert(D::R0 == 1, "k_rowd_small is the R0 == 1 form");
    const RowArgs<T>& a = ad.r;
    constexpr int N2 = D::L, R = D::R, M = D::M, BPW = D::BPW, G = (NT / 64) * BPW;
    constexpr int NEB = MODE == ROW_MID ? TWX_MAX_PHASE * 2 * R : 0;
    __shared__ C lds[G * D::BK];
    static_assert(sizeof(ArgPart<T>) <= sizeof(C), "arg-max records share the per-row scratch");
    __shared__ C tabs[R * R + NEB + G * R];                       // tab_a | phase ramp (MID) | per-row scratch [G][R]
    C* s_eb = tabs + R * R;
    C* s_vcr = s_eb + NEB;                                        // MID: W_N^{-k1 R b} per row
    ArgPart<T>* s_red = reinterpret_cast<ArgPart<T>*>(s_vcr);     // BAND: per-lane arg-max records
    const int tid = threadIdx.x;
    const int wv = tid >> 6, l = tid & 63;
    const int g = wv * BPW + l / R, q1 = l % R;                   // row slot of this lane, position inside the row
    const unsigned row_id = blockIdx.x * G + (unsigned)g;
    const bool act = (l < BPW * R) && row_id < total_rows;
    const unsigned rid = act ? row_id : (blockIdx.x * G);         // a valid row for address arithmetic of idle lanes
    const int k1 = rid / a.nwin, b = rid % a.nwin;
    const unsigned mask = (1u << a.tshift) - 1u;
    // ---- tables, then the rows: every thread copies elements t = tid, tid+NT, ... of the G*M element group
    for (int i = tid; i < R * R; i += NT) tabs[i] = ad.dtabs[i];
    if constexpr (MODE == ROW_MID) {
        for (int i = tid; i < a.nphase * 2 * R; i += NT) s_eb[i] = ad.eb_d[i];
        if (act) {
            const unsigned m = (unsigned)k1 * (unsigned)(q1 * R);                  // conj(W_N^{k1 R b}), b = q1
            s_vcr[g * R + q1] = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & mask]));
        }
    }
    constexpr int NLD = (G * M + NT - 1) / NT;
    C ld[NLD];
    TWX_UNROLL
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + k * NT;
        const unsigned rr = blockIdx.x * G + (unsigned)(idx / M);
        const int t = idx % M;
        const bool ok = idx < G * M && rr < total_rows;
        const unsigned r2 = ok ? rr : blockIdx.x * G;
        ld[k] = (a.A + (long long)(r2 % a.nwin) * a.n)[a_index((unsigned)t, r2 / a.nwin, (unsigned)a.n1, a.wshift)];
    }
    C csr[MODE == ROW_MID ? R : 1];
    if constexpr (MODE == ROW_MID) {
        const C* cs = ad.cspec_perm + (long long)k1 * N2;
        TWX_UNROLL
        for (int q2 = 0; q2 < R; ++q2) csr[q2] = (cs + q2 * R)[(unsigned)q1];
    }
    TWX_UNROLL
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + k * NT;
        if (idx < G * M) { const int t = idx % M; lds[D::phys(idx / M, t / R, t % R)] = ld[k]; }
    }
    __syncthreads();
    C v[R];
    if (act) D::f1(lds, tabs, g, q1, v);
    wave_sync_lds();
    if (act) D::f2(lds, g, q1, v);                                 // v[q2] = X[k1 + N1*(q1 + R q2)]

    if constexpr (MODE == ROW_BAND) {
        Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
        if (act) {
            const long long half = a.n / 2;
            TWX_UNROLL
            for (int q2 = 0; q2 < R; ++q2) {
                const long long k = (long long)k1 + (long long)a.n1 * (q1 + R * q2);
                long long i = k - (a.n - half); if (i < 0) i += a.n;
                if (i >= a.band_lo && i <= a.band_hi) best.take(cnorm(v[q2]), (unsigned int)i);
            }
            ArgPart<T> p; p.val = best.val; p.idx = best.idx;
            s_red[g * R + q1] = p;
        }
        wave_sync_lds();
        if (act && q1 == 0) {
            TWX_UNROLL
            for (int j = 1; j < R; ++j) { const ArgPart<T> p = s_red[g * R + j]; best.take(p.val, p.idx); }
            ArgPart<T> p; p.val = best.val; p.idx = best.idx;
            a.part[(long long)b * a.n1 + k1] = p;
        }
    } else {
        C pr[R];
        C ua = mk<T>(1, 0);
        if (act) {
            const unsigned m = (unsigned)k1 * (unsigned)q1;                        // W_N^{-k1 a}, a = q1
            ua = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & mask]));
            if (k1 == 0 && q1 == 0) a.dc[b] = v[0];
            TWX_UNROLL
            for (int q2 = 0; q2 < R; ++q2) pr[q2] = cmul(v[q2], csr[q2]);          // ffty.*fcode (godual_ranging.m:26)
        }
        for (int rho = 0; rho < a.nphase; ++rho) {
            int lq = q1;
            asm volatile("" : "+v"(lq));                                           // keep address arithmetic inside the loop
            if (act) {
                if (rho == 0) {
                    TWX_UNROLL
                    for (int q2 = 0; q2 < R; ++q2) v[q2] = pr[q2];
                } else {
                    const C eaj = ad.ea_d[rho * R + lq];
                    TWX_UNROLL
                    for (int q2 = 0; q2 < R; ++q2) {
                        C e;
                        if constexpr (R % 2 == 0) e = s_eb[(rho * 2 + (q2 >= R / 2 ? 1 : 0)) * R + q2];
                        else e = s_eb[(rho * 2 + ((2 * (lq + R * q2) >= N2) ? 1 : 0)) * R + q2];
                        v[q2] = cmul3(pr[q2], eaj, e);
                    }
                }
                D::iA_pre(tabs, lq, v);
            }
            wave_sync_lds();                       // this wave's reads of the previous phase are done (blocks are wave-local)
            if (act) D::iA_store(lds, g, lq, v);
            wave_sync_lds();
            if (act) {
                D::iB_keep(lds, g, lq, v);                                         // v[bq] = z[lq + R bq]
                const C uu = cmul(ua, a.ramp1[(long long)rho * a.n1 + k1]);
                C* out = a.Bz + ((long long)b * a.nphase + rho) * a.n + (long long)k1 * N2;
                TWX_UNROLL
                for (int bq = 0; bq < R; ++bq) out[bq * R + lq] = cmul3(v[bq], uu, s_vcr[g * R + bq]);   // · W_N^{-k1 q2} · ramp1
            }
        }
    }
}

// code spectrum in block-thread order for k_rowd<MID>
template <typename T>
__global__ void k_cspec_perm(const cpx<T>* __restrict__ nat, cpx<T>* __restrict__ perm, int n1, int n2, int r0, int r) {
    const int k1 = blockIdx.x;
    const int nu = r0 * r;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int q2 = i / nu, uu = i % nu, q0 = uu / r, q1 = uu % r;
        perm[(long long)k1 * n2 + i] = nat[(long long)k1 * n2 + q0 + r0 * q1 + r0 * r * q2];
    }
}

// ------------------------------------------------------------------------------------------
// k_row_caf: delay x Doppler cross-ambiguity on the integer-bin Doppler grid f = kappa*fs/N.
// Mixing by exp(-2 pi i kappa n/N) is a circular shift of FFT(y) by kappa bins,
// so one forward transform Y serves every bin: row k1 of the shifted spectrum is row
// (k1+kappa) mod N1 of Y rotated by floor((k1+kappa)/N1) along k2.  Per bin this kernel does
// Y_shift .* conj(FFT(code)) → inverse row FFT → ·W_N^{-k1 q2}  (the loop body of
// experiments/231001_DLL_PLL/rxcomplex.cpp:543-551 without the per-bin forward FFT).
// grid = N1 * nbins (bins of equal k1 adjacent → the code-spectrum row is shared in L2)
// ------------------------------------------------------------------------------------------
template <typename T> struct CafArgs {
    long long n; int n1, nbins; long long kappa0;
    const cpx<T>* Y;        // FFT of the mean-removed window, [k1][k2]
    const cpx<T>* cspec;    // conj(FFT(code)), [k1][k2]
    const cpx<T>* stab_i;
    const cpx<T>* ta; const cpx<T>* tb; int tshift;
    T scale;
    cpx<T>* Bz;             // [bin][k1][q2]
    // DIF/DIT form (k_rowd_caf): Y and the code spectrum in block-thread order, RowD tables, bins per workgroup
    const cpx<T>* Yperm;    // [k1][q2][u] = Y[k1][q0 + R0 q1 + R0 R q2], u = q0*R + q1 (k_cspec_perm applied to Y); nullptr: Stockham form
    const cpx<T>* cspec_perm;
    const cpx<T>* dtabs;
    const cpx<T>* vc;       // [k1][c] exp(+2 pi i k1 c M/N) (k_rowd_caf: stage C's output twiddle, scalar loads)
    int bpw;
    int nt;                 // 1: non-temporal bin-buffer stores (a buffer far larger than the caches); 0: the few bins of a launch
                            // are meant to stay in L2 / Infinity Cache until the last pass reads them
    int rotate;             // k_rowd_caf: walk the workgroup's bins in the order rotated by k1 (L2 reuse of the Y rows, see the kernel)
    int lds_pad;            // k_rowd_caf: bytes of dynamic LDS the launch reserves without using them (experiment: one row-pass workgroup
                            // per CU, so that the column pass of the other stream's launch can share the CU; profiles/r04_caf_overlap_rotate.txt)
};

template <class P2, typename T, int PADQ, int NT>
__global__ __launch_bounds__(NT) void k_row_caf(CafArgs<T> a) {
    using C = cpx<T>;
    using PR = typename Rev<P2>::type;
    using TI = RowTile<PR, T, true, PADQ>;
    constexpr int S = P2::S, N2 = P2::L;
    constexpr int R0 = PR::radix(0), NS0 = N2 / R0, RIL = PR::radix(S - 1), NSI = N2 / RIL;
    constexpr int NTI = StageTabs<PR>::total;
    __shared__ C lds[TI::lds_elems + NTI + RIL];
    C* tab_i = lds + TI::lds_elems;
    C* s_vc = tab_i + NTI;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int k1 = logical / a.nbins, bin = logical % a.nbins;
    const int tid = threadIdx.x;
    const long long sft = (long long)k1 + a.kappa0 + bin;
    long long k1s = sft % a.n1; if (k1s < 0) k1s += a.n1;
    long long cr = ((sft - k1s) / a.n1) % N2; if (cr < 0) cr += N2;
    const C* yrow = a.Y + k1s * N2;
    const C* cs = a.cspec + (long long)k1 * N2;
    C v[PR::rmax()];
    if (tid < NS0) {
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) {
            int k2 = tid + r * NS0;
            int ks = k2 + (int)cr; if (ks >= N2) ks -= N2;
            v[r] = cmul(yrow[ks], cs[k2]);
        }
    }
    for (int i = tid; i < NTI; i += NT) tab_i[i] = a.stab_i[i];
    if (tid < RIL) {
        const unsigned m = (unsigned)k1 * (unsigned)tid * (unsigned)NSI;
        s_vc[tid] = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & ((1u << a.tshift) - 1u)]));
    }
    if (tid < NS0) { TI::template bfly<0>(v); TI::template store_lds<0>(lds, tid, 0, v); }
    __syncthreads();
    MidStages<TI, PR, T, 1>::run(lds, tab_i, v, tid);
    if (tid < NSI) {
        TI::template load_lds_tab<S - 1>(lds, tab_i, tid, v);
        TI::template bfly<S - 1>(v);
        const unsigned m = (unsigned)k1 * (unsigned)tid;
        const C ub = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & ((1u << a.tshift) - 1u)]));
        C* out = a.Bz + (long long)bin * a.n + (long long)k1 * N2;
        TWX_UNROLL
        for (int q = 0; q < RIL; ++q) (out + q * NSI)[(unsigned)tid] = cmul(cmul(v[q], ub), s_vc[q]);
    }
}

// The same on the DIF/DIT row transform (RowD): one workgroup takes row k1 of `bpw` consecutive bins, so the tables, the
// W_N^{-k1 q2} factors and the block-thread mapping are set up once, and each bin costs the inverse transform alone
// (2 workgroup barriers instead of 5).  The k2 rotation of the shifted row becomes an address computation on the
// block-thread layout: k2' = k2 + cr = s_lo + R0 R ((q2 + s_hi) mod R) with s = q0 + R0 q1 + cr.
// grid = N1 * ceil(nbins / bpw)
template <class P2, typename T, int NT>
__global__ __launch_bounds__(NT, (sizeof(T) == 4 ? 4 : 1)) void k_rowd_caf(CafArgs<T> a) {
    using C = cpx<T>;
    using D = RowD<P2, T>;
    constexpr int N2 = D::L, R = D::R, R0 = D::R0, M = D::M, NU = R0 * R;
    constexpr int RMAX = R > R0 ? R : R0;
    static_assert(NT >= D::NT_MIN, "not enough threads for RowD");
    __shared__ C lds[D::lds_elems];
    __shared__ C tabs[D::tab_total + R0];
    C* s_vc = tabs + D::tab_total;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    // k1 fastest: neighbouring workgroups (one XCD takes a contiguous range of logical ids) read rows k1+kappa .. +bpw-1
    // of Y shifted by one, so all but one of them are L2 hits; the code-spectrum row is private to the workgroup and
    // stays in registers for all its bins
    const int grp = logical / a.n1, k1 = logical % a.n1;
    const int tid = threadIdx.x;
    int q0, qi;
    const bool act = D::blk_map(tid, q0, qi);
    const int q0c = min(q0, R0 - 1);                       // idle lanes compute valid addresses (their loads are unconditional)
    const unsigned mask = (1u << a.tshift) - 1u;
    // As in k_rowd<MID>: the output twiddle W_N^{-k1 t}, t = a + R b, goes into stage B's factors (the table tc becomes
    // conj(tc[q0][b]) * W_N^{-k1 R b}, the thread's scalar conj(W_L^{q0 a}) * W_N^{-k1 a}), and stage C's W_N^{-k1 c M} is read
    // with scalar loads from the per-row table vc: one product and one LDS read per output less.
    constexpr bool FOLD = TWX_MID_FOLD && R0 > 1;
    for (int i = tid; i < D::tab_total; i += NT) {
        C t = a.dtabs[i];
        if (FOLD && i >= D::tab_c && i < D::tab_c + R0 * R) {
            const unsigned mb = (unsigned)k1 * (unsigned)R * (unsigned)((i - D::tab_c) % R);
            t = cmulc(cconj(cmul(a.ta[mb >> a.tshift], a.tb[mb & mask])), t);
        }
        tabs[i] = t;
    }
    if (tid < R0) {
        const unsigned m = (unsigned)k1 * (unsigned)tid * (unsigned)M;
        s_vc[tid] = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & mask]));
    }
    C ub = mk<T>(1, 0);
    C wa0 = mk<T>(1, 0);
    if constexpr (FOLD) {
        const unsigned ma = (unsigned)k1 * (unsigned)qi;
        wa0 = cmulc(cconj(cmul(a.ta[ma >> a.tshift], a.tb[ma & mask])), a.dtabs[D::tab_b + q0c * R + qi]);
    } else if (tid < M) {
        const unsigned m = (unsigned)k1 * (unsigned)tid;
        ub = cconj(cmul(a.ta[m >> a.tshift], a.tb[m & mask]));
    }
    const C* vcrow = a.vc + (long long)k1 * R0;
    constexpr bool HOLD = sizeof(T) == 4;                  // complex double: 2 x 80 registers would spill, re-read the row per bin
    C v[RMAX], csr[HOLD ? R : 1];
    const C* cs = a.cspec_perm + (long long)k1 * N2 + (q0c * R + qi);
    if constexpr (HOLD) {
        TWX_UNROLL
        for (int q2 = 0; q2 < R; ++q2) csr[q2] = cs[q2 * NU];
    }
    __syncthreads();
    const int bin_end = min(a.nbins, (grp + 1) * a.bpw);
    // The workgroup walks its bins in an order ROTATED by k1: bin = bin0 + (s - k1) mod nbg at step s.  Row k1 + kappa of Y is what
    // it reads for bin kappa, so the nbg workgroups of consecutive k1 that run side by side on an XCD (k1 is the fast index of the
    // launch) ask for the SAME one or two rows of Y at the same step — walked in plain order they ask for nbg different rows per
    // step, 64 rows = 4 MB = the whole L2 of the XCD for its 64 resident workgroups, and every row is evicted between its uses
    // (PMC: 1.5 GB of Y fetched per 64-bin launch for the 40 MB that Y holds, profiles/r04_caf_*).
    const int bin0 = grp * a.bpw, nbg = bin_end - bin0;
    const int rot = a.rotate ? k1 % max(nbg, 1) : 0;
    for (int s = 0; s < nbg; ++s) {
        int bo = s - rot; if (bo < 0) bo += nbg;
        const int bin = bin0 + bo;
        const long long sft = (long long)k1 + a.kappa0 + bin;
        long long k1s = sft % a.n1; if (k1s < 0) k1s += a.n1;
        long long cr = ((sft - k1s) / a.n1) % N2; if (cr < 0) cr += N2;
        {
            const int sh = q0c + R0 * qi + (int)cr;        // < NU + N2
            const int s_hi = sh / NU, s_lo = sh - s_hi * NU;
            const int up = (s_lo % R0) * R + s_lo / R0;
            const C* yrow = a.Yperm + k1s * N2 + up;
            TWX_UNROLL
            for (int q2 = 0; q2 < R; ++q2) {
                int qq = q2 + s_hi; if (qq >= R) qq -= R;
                if constexpr (HOLD) v[q2] = yrow[qq * NU]; else v[q2] = cmul(yrow[qq * NU], cs[q2 * NU]);
            }
        }
        if (act) {
            if constexpr (HOLD) {
                TWX_UNROLL
                for (int q2 = 0; q2 < R; ++q2) v[q2] = cmul(v[q2], csr[q2]);
            }
            D::iA_pre(tabs, qi, v);
        }
        if (s > 0) __syncthreads();                        // the previous bin's stage C has read every block
        if (act) D::iA_store(lds, q0, qi, v);
        wave_sync_lds();
        if constexpr (FOLD) { if (act) D::iB_folded(lds, tabs, q0, qi, wa0, v); }
        else if (act) D::iB(lds, tabs, q0, qi, v);
        __syncthreads();
        if (tid < M) {
            D::iC(lds, tid, v);
            C* out = a.Bz + (long long)bin * a.n + (long long)k1 * N2;
            TWX_UNROLL
            for (int c = 0; c < R0; ++c) {
                C o;
                if constexpr (FOLD) o = TWX_MID_SGPR ? cmul_us(v[c], ld_uniform(vcrow, c)) : cmul(v[c], s_vc[c]);
                else o = cmul3(v[c], ub, s_vc[c]);
                if (TWX_NT_BZ && a.nt) __builtin_nontemporal_store(o, out + c * M + (unsigned)tid); else (out + c * M)[(unsigned)tid] = o;
            }
        }
    }
}

// per-bin (peak magnitude, lag) from the column-pass partial records.  grid = nbins
template <typename T>
__global__ __launch_bounds__(256) void k_caf_reduce(const ArgPart<T>* __restrict__ part, int nparts, double mag_scale,
                                                    double* __restrict__ pk, long long* __restrict__ lag) {
    const int b = blockIdx.x;
    __shared__ char scratch[64];
    Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
    for (int i = threadIdx.x; i < nparts; i += 256) best.take(part[(long long)b * nparts + i].val, part[(long long)b * nparts + i].idx);
    best = block_best<T, 256>(best, scratch);
    if (threadIdx.x == 0) { pk[b] = sqrt((double)best.val) * mag_scale; lag[b] = best.idx; }
}

// ------------------------------------------------------------------------------------------
// Long squared spectra for the acquisition stage (search_df and the per-chunk carrier update of
// acquisition/claudio_aligned_code_ranging_separate.m:27-31,162-163): d2 = abs(fft(d.^2)) over a
// chunk of L samples, L not tied to the code length.
//   k_sq_dft_bins:    a handful of bins by direct summation (exact integer phase reduction).
//   k_sqspec_combine: a band of bins of an L = M*N point transform from the M decimated N-point
//                     spectra F_r (r = n mod M): X[k] = sum_r W_L^{r k} F_r[k mod N].
// ------------------------------------------------------------------------------------------
// One pass over the samples for up to SQ_NB bins at a time.  A thread walks the samples n0, n0 + stride, ...: its twiddle
// exp(-2 pi i k n / L) starts from an exact integer phase reduction (one 64-bit modulo and one fp64 sincospi per bin and thread) and
// advances by the constant factor exp(-2 pi i k stride / L) — a complex multiply per sample and bin where the first version paid
// a modulo and a sincospi; re-seeded exactly every 128 steps (fp64: 128 roundings of 1.1e-16; at 32 the re-seeding — eight 64-bit
// modulos and fp64 sincospi per thread — was half of the kernel's instructions when a thread walks ~80 samples).  Per-workgroup partials, added up in a fixed order by
// k_sq_dft_final: the first version's atomicAdd on the bins' 14 words (2 048 workgroups on each) was most of its 1.15 ms per
// 10^7-sample chunk (same-address device atomics: ~0.4 us each, profiles/r04_tracked_rate.txt) and made the sums order-dependent.
// grid = workgroups (any), partial[gridDim.x][nb][2]
#define TWX_SQ_NB 8
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sq_dft_bins(const short2* __restrict__ in, int nch, long long L,
                                                     const long long* __restrict__ bins, int nb, double* __restrict__ partial) {
    __shared__ double red[4][2 * TWX_SQ_NB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n0 = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    for (int b0 = 0; b0 < nb; b0 += TWX_SQ_NB) {
        const int cnt = min(TWX_SQ_NB, nb - b0);
        double zc[TWX_SQ_NB], zs[TWX_SQ_NB], wc[TWX_SQ_NB], ws[TWX_SQ_NB], sx[TWX_SQ_NB], sy[TWX_SQ_NB];
        unsigned long long kk[TWX_SQ_NB];
#pragma unroll
        for (int j = 0; j < TWX_SQ_NB; ++j) {
            kk[j] = (unsigned long long)bins[b0 + min(j, cnt - 1)];          // already reduced to 0..L-1
            const unsigned long long ts = ((unsigned long long)stride % (unsigned long long)L * kk[j]) % (unsigned long long)L;
            sincospi(2.0 * (double)ts / (double)L, &ws[j], &wc[j]);
            sx[j] = 0; sy[j] = 0; zc[j] = 1; zs[j] = 0;
        }
        int step = 0;
        for (long long n = n0; n < L; n += stride, ++step) {
            if ((step & 127) == 0) {
#pragma unroll
                for (int j = 0; j < TWX_SQ_NB; ++j) {
                    const unsigned long long t = ((unsigned long long)n * kk[j]) % (unsigned long long)L;     // n < L < 2^32 in every use: no overflow
                    sincospi(2.0 * (double)t / (double)L, &zs[j], &zc[j]);
                }
            }
            const short2 s = in[n * nch];
            const double I = (double)s.x, Q = (double)s.y;
            const double re = I * I - Q * Q, im = 2.0 * I * Q;               // d^2, exact
#pragma unroll
            for (int j = 0; j < TWX_SQ_NB; ++j) {
                sx[j] += re * zc[j] + im * zs[j];                            // d^2 * exp(-i phi)
                sy[j] += im * zc[j] - re * zs[j];
                const double c2 = zc[j] * wc[j] - zs[j] * ws[j], s2 = zc[j] * ws[j] + zs[j] * wc[j];
                zc[j] = c2; zs[j] = s2;
            }
        }
#pragma unroll
        for (int j = 0; j < TWX_SQ_NB; ++j) {
            for (int d = 32; d >= 1; d >>= 1) { sx[j] += __shfl_down(sx[j], d, 64); sy[j] += __shfl_down(sy[j], d, 64); }
            if (lane == 0) { red[wave][2 * j] = sx[j]; red[wave][2 * j + 1] = sy[j]; }
        }
        __syncthreads();
        if (threadIdx.x < 2 * cnt) {
            const int q = threadIdx.x;
            partial[((long long)blockIdx.x * nb + b0) * 2 + q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        }
        __syncthreads();
    }
}
// acc[q] = sum over the workgroups' partials, fixed order.  grid = nvals (one workgroup per value), block = 256
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_sq_dft_final(const double* __restrict__ partial, int nparts, int nvals, double* __restrict__ acc) {
    __shared__ double sh[256];
    const int q = blockIdx.x;
    double a = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += partial[(long long)i * nvals + q];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h]; __syncthreads(); }
    if (threadIdx.x == 0) acc[q] = sh[0];
}

// the twiddle W_L^{r k} advances by the constant factor exp(-2 pi i k / L) from r to r + 1 (one fp64 sincospi per bin instead of
// one per term; M <= 4096 steps of fp64 rounding)
template <typename T>
__global__ __launch_bounds__(256) void k_sqspec_combine(const cpx<T>* __restrict__ spec /*[M][k1][k2]*/, int M, long long N, int N1, int N2,
                                                        long long k_lo, long long nk, double* __restrict__ mag) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nk) return;
    const long long L = (long long)M * N;
    long long kk = (k_lo + i) % L; if (kk < 0) kk += L;
    const long long j = kk % N;
    const int k1 = (int)(j % N1), k2 = (int)(j / N1);
    double ws, wc;
    sincospi(2.0 * (double)kk / (double)L, &ws, &wc);
    double zc = 1, zs = 0, sx = 0, sy = 0;
    for (int r = 0; r < M; ++r) {
        if ((r & 63) == 0 && r) {                                            // exact re-seed
            const unsigned long long t = ((unsigned long long)r * (unsigned long long)kk) % (unsigned long long)L;
            sincospi(2.0 * (double)t / (double)L, &zs, &zc);
        }
        const cpx<T> f = spec[(long long)r * N + (long long)k1 * N2 + k2];
        sx += (double)f.x * zc + (double)f.y * zs;
        sy += (double)f.y * zc - (double)f.x * zs;
        const double c2 = zc * wc - zs * ws, s2 = zc * ws + zs * wc;
        zc = c2; zs = s2;
    }
    mag[i] = sqrt(sx * sx + sy * sy);
}

// ------------------------------------------------------------------------------------------
// k_col_inv: last pass of the inverse transform fused with [~,indice]=max(abs(prnmap))
// (godual_ranging.m:29).  grid = ntiles * R * windows
// ------------------------------------------------------------------------------------------
template <typename T> struct ColInvArgs {
    long long n; int n2, ntiles, nphase, nwin;
    const cpx<T>* Bz;        // [b][rho][k1][q2]
    const cpx<T>* tw1;
    ArgPart<T>* part;        // [b][rho*ntiles + tile]
    cpx<T>* zout;            // optional full output [b][R*N] (natural interleaved order), or nullptr
    int norm1;               // arg-max of (|re|+|im|)^2 instead of |z|^2: cblas_izamax (rxcomplex.cpp:553) — selects the NORM1 instantiation
    T zscale;                // factor on the values written to zout (the ifft normalisation of twx_xcorr_map_dev: no separate pass)
};

template <class P1R, typename T, int W, int NT, int NORM1 = 0>
__global__ __launch_bounds__(NT) void k_col_inv(ColInvArgs<T> a) {
    using TL = Tile<P1R, T, true, W, 0>;
    using C = cpx<T>;
    constexpr int S = P1R::S;
    __shared__ C lds[TL::lds_elems];
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = logical % a.ntiles;
    const int rho = (logical / a.ntiles) % a.nphase;
    const int b = logical / (a.ntiles * a.nphase);
    const int c0 = tile * W, tid = threadIdx.x;
    const C* src = a.Bz + ((long long)b * a.nphase + rho) * a.n;
    C v[P1R::rmax()];
    {
        constexpr int R = P1R::radix(0);
        if (tid < TL::template tasks<0>()) {
            const int j = tid / W, c = tid % W;
            // address = wave-uniform row base (SGPR pair) + 32-bit lane offset → saddr-form global loads,
            // no per-element 64-bit VALU address arithmetic
            const unsigned int loff = ((unsigned int)j * (unsigned int)a.n2 + (unsigned int)(c0 + c)) * (unsigned int)sizeof(C);
            const char* cbase = reinterpret_cast<const char*>(src);
            const unsigned long long rstride = (unsigned long long)(P1R::L / R) * (unsigned long long)a.n2 * sizeof(C);
            TWX_UNROLL
            for (int r = 0; r < R; ++r)
                v[r] = (TWX_ABLC == 2) ? mk<T>((T)(tid + r), (T)(r - tid)) : *reinterpret_cast<const C*>(cbase + r * rstride + loff);
            if (TWX_ABLC != 1) TL::template bfly<0>(v);
            if (S > 1 && TWX_ABLC != 1) TL::template store_lds<0>(lds, j, c, v);
        }
    }
    constexpr bool TWPRE = (S == 2);      // last stage's twiddle gather issued before the barrier (see k_col_fwd)
    C twr[TWPRE ? P1R::radix(S - 1) - 1 : 1];
    if constexpr (TWPRE) {
        if (tid < TL::template tasks<S - 1>()) TL::template load_tw<S - 1>(a.tw1, tid / W, twr);
    }
    if (S > 1) __syncthreads();
    if constexpr (S > 2) {
        if (tid < TL::template tasks<1>()) { TL::template load_lds<1>(lds, a.tw1, tid / W, tid % W, v); TL::template bfly<1>(v); }
        __syncthreads();
        if (tid < TL::template tasks<1>()) TL::template store_lds<1>(lds, tid / W, tid % W, v);
        __syncthreads();
    }
    if constexpr (S > 3) {
        if (tid < TL::template tasks<2>()) { TL::template load_lds<2>(lds, a.tw1, tid / W, tid % W, v); TL::template bfly<2>(v); }
        __syncthreads();
        if (tid < TL::template tasks<2>()) TL::template store_lds<2>(lds, tid / W, tid % W, v);
        __syncthreads();
    }
    Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
    {
        constexpr int s = S - 1;
        constexpr int R = P1R::radix(s);
        if (tid < TL::template tasks<s>()) {
            const int j = tid / W, c = tid % W;
            if (S > 1 && TWX_ABLC != 1) {
                if constexpr (TWPRE) TL::template load_lds_tw<s>(lds, twr, j, c, v);
                else TL::template load_lds<s>(lds, a.tw1, j, c, v);
                TL::template bfly<s>(v);
            }
            const unsigned int mbase = (unsigned int)(c0 + c) * (unsigned int)a.nphase + (unsigned int)rho;
            const unsigned int mstep = (unsigned int)a.n2 * (unsigned int)a.nphase;
            // per-thread arg-max in two cheap sweeps: max value (1 op per sample), then the first q that
            // attains it (the lag index grows with q, so the lowest q is the lowest index)
            T nv[R];
            T bv = T(-1);
            TWX_UNROLL
            for (int q = 0; q < R; ++q) {
                if constexpr (NORM1) { const T s1 = (v[q].x < 0 ? -v[q].x : v[q].x) + (v[q].y < 0 ? -v[q].y : v[q].y); nv[q] = s1 * s1; }
                else nv[q] = cnorm(v[q]);
                bv = tmax(nv[q], bv);
            }
            int bq = 0;
            TWX_UNROLL
            for (int q = R - 1; q >= 0; --q) bq = (nv[q] == bv) ? q : bq;
            best.val = bv;
            best.idx = (unsigned int)(j + bq * (P1R::L / R)) * mstep + mbase;                         // R*(q1*N2+q2)+rho < 2^32
            if (a.zout) {       // test/inspection path only (twx_xcorr_map)
                TWX_UNROLL
                for (int q = 0; q < R; ++q) {
                    const unsigned int m = (unsigned int)TL::template out_pos<s>(j, q) * mstep + mbase;
                    a.zout[(long long)b * a.n * a.nphase + m] = cscale(v[q], a.zscale);
                }
            }
        }
    }
    __syncthreads();   // LDS reads finished before it is reused as reduction scratch
    best = block_best<T, NT>(best, lds);
    if (tid == 0) {
        ArgPart<T> p; p.val = best.val; p.idx = best.idx;
        a.part[(long long)b * (a.ntiles * a.nphase) + rho * a.ntiles + tile] = p;
    }
}

// ------------------------------------------------------------------------------------------
// k_col_inv3: k_col_inv for two-stage column plans in fp32 with THREE workgroups resident per CU.
// The column passes are latency-bound, not bandwidth-bound: a workgroup loads its tile, waits, transforms, reduces —
// strictly in that order — and with the 80-KB tile exchange only two of them fit a CU, so HBM idles while both
// compute (tools/bw_probe: the same read pattern alone sustains 5.9 TB/s, 6.6 with non-temporal loads, the kernel
// 3.7).  Here the stage-0 -> stage-1 exchange goes through LDS one COMPONENT at a time (real parts, then imaginary
// parts: the registers holding the written half are free before the other half is read, so the register peak stays
// at one tile column per thread), which halves the LDS footprint to L*W*4 bytes (40 KB), and the last-stage
// twiddles are fetched after the exchange instead of being parked in 48 registers, so that 3 x 7 waves fit.
// Bz is read exactly once: non-temporal loads.   grid = ntiles * R * windows
// ------------------------------------------------------------------------------------------
template <class P1R, typename T, int W, int NT, int NORM1 = 0>
__global__ __launch_bounds__(NT, 6) void k_col_inv3(ColInvArgs<T> a) {
    static_assert(P1R::S == 2 && sizeof(T) == 4, "two-stage fp32 plans only");
    using TL = Tile<P1R, T, true, W, 0>;
    using C = cpx<T>;
    constexpr int L = P1R::L, R0 = P1R::radix(0), R1 = P1R::radix(1);
    __shared__ T lf[L * W];
#if TWX_INV3_TWLDS
    // The last stage's twiddles through LDS (round 6): asked for with the tile's own loads and parked in LDS instead of registers, so that
    // the L2 round trip of the twiddle gather — 24 dependent-free but LATE loads per lane, issued after the exchange because 48 registers
    // to hold them from the start do not exist — leaves the workgroup's critical path.  5 KB more LDS (three resident workgroups, not
    // four) for 20 % less time: 4.24 -> 5.25 TB/s, profiles/r06_colinv.txt — what the cache counters had said (requests served faster
    // than the probe's, fewer of them in flight: the kernel, not the memory system, was waiting).  The other reading of those counters —
    // resident workgroups that walk the tiles with the next tile loaded ahead, 128 registers, two per CU — was built first and lost
    // (0.238 - 0.263 against 0.226 ms): with two workgroups a CU has nothing to run while both wait at their barriers.
    __shared__ C s_tw[L];
    lds_fill_dwords<NT>(reinterpret_cast<float*>(s_tw), reinterpret_cast<const float*>(a.tw1), 2 * L);      // (global_load_lds: no register in between)
#endif
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = logical % a.ntiles;
    const int rho = (logical / a.ntiles) % a.nphase;
    const int b = logical / (a.ntiles * a.nphase);
    const int c0 = tile * W, tid = threadIdx.x;
    const int j = tid / W, c = tid % W;
    const bool p0 = tid < TL::template tasks<0>(), p1 = tid < TL::template tasks<1>();
    const C* src = a.Bz + ((long long)b * a.nphase + rho) * a.n;
    C v[R0];
    if (p0) {
        const unsigned int loff = ((unsigned int)j * (unsigned int)a.n2 + (unsigned int)(c0 + c)) * (unsigned int)sizeof(C);
        const char* cbase = reinterpret_cast<const char*>(src);
        const unsigned long long rstride = (unsigned long long)(L / R0) * (unsigned long long)a.n2 * sizeof(C);
        // base and row stride pinned in SGPRs, the lane part a 32-bit offset: no 64-bit VALU address per load
        const unsigned long long cb = sgpr_u64(reinterpret_cast<unsigned long long>(cbase)), rs = sgpr_u64(rstride);
        (void)cb; (void)rs;
#if TWX_BZ16
        {
            const unsigned long long cb16 = sgpr_u64(reinterpret_cast<unsigned long long>(a.Bz) + (unsigned long long)((((long long)b * a.nphase + rho) * a.n) * 4));
            const unsigned long long rs16 = sgpr_u64(rstride / 2);
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) { const float2 f = bz16_unpack(ld_pin<unsigned, TWX_NT_INV != 0>(cb16, r * rs16, loff / 2)); v[r] = mk<T>((T)f.x, (T)f.y); }
        }
#else
        TWX_UNROLL
        for (int r = 0; r < R0; ++r) v[r] = (TWX_ABLC == 2) ? mk<T>((T)(tid + r), (T)(r - tid)) : ld_pin<C, TWX_NT_INV != 0>(cb, r * rs, loff);
#endif
        if (TWX_ABLC == 1) {                 // timing-only build: loads, no arithmetic, no exchange
            TWX_UNROLL
            for (int r = 0; r < R0; ++r) asm volatile("" ::"v"(v[r]));
            return;
        }
        TL::template bfly<0>(v);
        const int ob = TL::template out_base<0>(j);
        TWX_UNROLL
        for (int q = 0; q < R0; ++q) lf[TL::template out_idx<0>(ob, j, q) * W + c] = v[q].x;
    }
    __syncthreads();
    C u[R1];
    const int ib = TL::template in_base<1>(j);
    if (p1) {
        TWX_UNROLL
        for (int r = 0; r < R1; ++r) u[r].x = lf[TL::template in_idx<1>(ib, j, r) * W + c];
    }
    __syncthreads();
    if (p0) {
        const int ob = TL::template out_base<0>(j);
        TWX_UNROLL
        for (int q = 0; q < R0; ++q) lf[TL::template out_idx<0>(ob, j, q) * W + c] = v[q].y;
    }
    __syncthreads();
    Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
    if (p1) {
        TWX_UNROLL
        for (int r = 0; r < R1; ++r) u[r].y = lf[TL::template in_idx<1>(ib, j, r) * W + c];
        constexpr int step = L / (P1R::ns(1) * R1);             // == 1 for a two-stage plan; jm == j
        TWX_UNROLL
#if TWX_INV3_TWLDS
        for (int r = 1; r < R1; ++r) u[r] = cmulc(u[r], s_tw[j * r * step]);
#else
        for (int r = 1; r < R1; ++r) u[r] = cmulc(u[r], tw_load(a.tw1, (unsigned)(j * r * step)));
#endif
        TL::template bfly<1>(u);
        const unsigned int mbase = (unsigned int)(c0 + c) * (unsigned int)a.nphase + (unsigned int)rho;
        const unsigned int mstep = (unsigned int)a.n2 * (unsigned int)a.nphase;
        T nv[R1];
        T bv = T(-1);
        TWX_UNROLL
        for (int q = 0; q < R1; ++q) {
            if constexpr (NORM1) { const T s1 = (u[q].x < 0 ? -u[q].x : u[q].x) + (u[q].y < 0 ? -u[q].y : u[q].y); nv[q] = s1 * s1; }
            else nv[q] = cnorm(u[q]);
            bv = tmax(nv[q], bv);
        }
        int bq = 0;
        TWX_UNROLL
        for (int q = R1 - 1; q >= 0; --q) bq = (nv[q] == bv) ? q : bq;
        best.val = bv;
        best.idx = (unsigned int)(j + bq * (L / R1)) * mstep + mbase;
        if (a.zout) {       // test/inspection path only (twx_xcorr_map)
            TWX_UNROLL
            for (int q = 0; q < R1; ++q) {
                const unsigned int m = (unsigned int)TL::template out_pos<1>(j, q) * mstep + mbase;
                a.zout[(long long)b * a.n * a.nphase + m] = cscale(u[q], a.zscale);
            }
        }
    }
    __syncthreads();
    best = block_best<T, NT>(best, lf);
    if (tid == 0) {
        ArgPart<T> p; p.val = best.val; p.idx = best.idx;
        a.part[(long long)b * (a.ntiles * a.nphase) + rho * a.ntiles + tile] = p;
    }
}

// ------------------------------------------------------------------------------------------
// The other SNR estimators the reference compares (experiments/220830_OP/process_OP.m:94-97,119-121,138; 221127_SNR/simu_snr.m),
// optional outputs (TWX_OPT_BRUIT_LEN / TWX_OPT_NOISE_SQUARE_LEN) next to the wipe-off SNR of k_peak:
//   bruit         = var(prnmap(indice+20 : indice+20+L-1))   off-peak variance of the correlation map, from Bz (fp64 re-evaluation
//                   lag by lag, as k_peak does for the twelve lags around the peak) — the map itself is never written
//   valmax_square = max(d22(freqindex)), noise_square = var(d22(tmpdf+20 : tmpdf+20+L-1)), d22 = fftshift(abs(fft(d1.^2))): the bins
//                   behind the carrier peak of the squared signal, from the column-pass output A of the SQUARE pass (still in its
//                   buffer between k_df_tables and the MIX column pass): per row k1 the handful of bins k = k1 + N1 k2 of the range
//                   as a pruned DFT over n2 (one twiddle per element, then a rotation per bin)
// Partial sums per workgroup in fixed order (no atomics), k_extra_final combines them.
// ------------------------------------------------------------------------------------------
struct ExtraArgs {
    long long n; int n1, n2, nphase;
    int bruit_len, sq_len;
    double* part_b;          // [b][nblk_b][3]: sum re, sum im, sum |z|^2 of the lags a workgroup evaluated
    int nblk_b;
    double* part_s;          // [b][n1][2]: sum |S|, sum |S|^2 of the bins row k1 holds
    const double* sqmax;     // [b]: |fft(d.^2)| at the carrier arg-max (k_df_tables), or nullptr when the carrier was supplied
    twx_extra* out; int out_stride;
};

template <typename T>
__global__ __launch_bounds__(256) void k_offpeak(ExtraArgs a, const cpx<T>* __restrict__ Bz, const cpx<double>* __restrict__ tw1d, const twx_result* __restrict__ res,
                                                 int res_stride, int convention, double inv_scale) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const long long M = a.n * a.nphase;
    const long long ind = res[(long long)b * res_stride].indice0;
    const long long li = (long long)blockIdx.x * 256 + tid;            // lag number inside the range
    double zr = 0, zi = 0, zz = 0;
    if (li < a.bruit_len && ind + 20 + a.bruit_len < M) {              // the guard of process_OP.m:119 (1-based indice + 1020 < length)
        long long m = ind + 20 + li;
        if (convention == TWX_CONV_CLAUDIO) m = (M - m) % M;           // prnmap_c[m] = conj(prnmap_g[(M - m) mod M]): same modulus, same variance
        const int rho = (int)(m % a.nphase);
        const long long q = m / a.nphase;
        const int q1 = (int)(q / a.n2), q2 = (int)(q % a.n2);
        const cpx<T>* src = Bz + ((long long)b * a.nphase + rho) * a.n + q2;
        double sx = 0, sy = 0;
        int ti = 0;
        for (int k1 = 0; k1 < a.n1; ++k1) {
            const cpx<T> v = src[(long long)k1 * a.n2];
            const cpx<double> w = tw1d[ti];                             // conj -> inverse
            sx += (double)v.x * w.x + (double)v.y * w.y;
            sy += (double)v.y * w.x - (double)v.x * w.y;
            ti += q1; if (ti >= a.n1) ti -= a.n1;
        }
        zr = sx * inv_scale / (double)M; zi = sy * inv_scale / (double)M;
        zz = zr * zr + zi * zi;
    }
    __shared__ double sh[3][256];
    sh[0][tid] = zr; sh[1][tid] = zi; sh[2][tid] = zz;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) { sh[0][tid] += sh[0][tid + d]; sh[1][tid] += sh[1][tid + d]; sh[2][tid] += sh[2][tid + d]; }
        __syncthreads();
    }
    if (tid < 3) a.part_b[((long long)b * a.nblk_b + blockIdx.x) * 3 + tid] = sh[tid][0];
}

#define TWX_SQN_EMAX 40      // row elements per thread of k_sq_noise: rows up to 10 240 points
#define TWX_SQN_GROUP 16     // bins of a row evaluated together
template <typename T>
__global__ __launch_bounds__(256) void k_sq_noise(ExtraArgs a, const cpx<T>* __restrict__ A, int wshift, const long long* __restrict__ dfidx) {
    const int k1 = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const long long i0 = dfidx[b];
    const long long lo = i0 + 20, hi = i0 + 20 + a.sq_len - 1;         // shifted indices tmpdf+20 .. tmpdf+20+L-1 (0-based)
    double s1 = 0, s2 = 0;
    __shared__ double sh[2][256];
    if (i0 >= 0 && hi <= a.n - 1) {
        // natural bin of shifted index i: k = (i + N - N/2) mod N; the bins of this row are k = k1 + N1 k2
        const long long sh0 = a.n - a.n / 2;
        const long long klo = (lo + sh0) % a.n;                          // first natural bin of the range (the range may wrap past N)
        // k2 of the first bin of the row at or after klo, going round: members are k2 = k2a, k2a + 1, ... while inside the range
        long long first = klo - k1; first = first <= 0 ? 0 : (first + a.n1 - 1) / a.n1;          // smallest k2 with k1 + N1 k2 >= klo
        cpx<T> x[TWX_SQN_EMAX];
        const cpx<T>* Ab = A + (long long)b * a.n;
        TWX_UNROLL
        for (int e = 0; e < TWX_SQN_EMAX; ++e) {
            const int n2i = tid + 256 * e;
            x[e] = n2i < a.n2 ? Ab[a_index((unsigned)n2i, (unsigned)k1, (unsigned)a.n1, wshift)] : mk<T>(0, 0);
        }
        // the row's bins of the range are k2 = first, first + 1, ... (going round past the end of the spectrum): sixteen at a time, per
        // element ONE twiddle exp(-2 pi i n2 k2 / N2) for the group's first bin and the step exp(-2 pi i n2 / N2), then a rotation per bin
        const long long cnt_max = (a.sq_len + a.n1 - 1) / a.n1 + 1;
        for (long long j0 = 0; j0 < cnt_max; j0 += TWX_SQN_GROUP) {
            double sr[TWX_SQN_GROUP], si[TWX_SQN_GROUP];
            TWX_UNROLL
            for (int g = 0; g < TWX_SQN_GROUP; ++g) { sr[g] = 0; si[g] = 0; }
            const long long k2g = (first + j0) % a.n2;
            TWX_UNROLL
            for (int e = 0; e < TWX_SQN_EMAX; ++e) {
                const int n2i = tid + 256 * e;
                if (n2i < a.n2) {
                    double ws, wc, ts, tc;
                    sincospi(-2.0 * (double)n2i / (double)a.n2, &ws, &wc);
                    sincospi(-2.0 * (double)(((long long)n2i * k2g) % a.n2) / (double)a.n2, &ts, &tc);
                    const double xr = (double)x[e].x, xi = (double)x[e].y;
                    double pr = xr * tc - xi * ts, pi = xr * ts + xi * tc;            // x * t
                    TWX_UNROLL
                    for (int g = 0; g < TWX_SQN_GROUP; ++g) {
                        sr[g] += pr; si[g] += pi;
                        const double nr = pr * wc - pi * ws; pi = pr * ws + pi * wc; pr = nr;
                    }
                }
            }
            TWX_UNROLL
            for (int g = 0; g < TWX_SQN_GROUP; ++g) {
                const long long j = j0 + g;
                long long k2 = first + j;
                long long k = (long long)k1 + (long long)a.n1 * k2;
                if (k >= a.n) k -= a.n;                                            // past the end of the spectrum: the range continues at bin 0
                long long i = k - sh0; if (i < 0) i += a.n;
                const bool in = j < cnt_max && k2 < 2ll * a.n2 && i >= lo && i <= hi;      // (uniform over the workgroup)
                sh[0][tid] = sr[g]; sh[1][tid] = si[g];
                __syncthreads();
                for (int d = 128; d >= 1; d >>= 1) {
                    if (tid < d) { sh[0][tid] += sh[0][tid + d]; sh[1][tid] += sh[1][tid + d]; }
                    __syncthreads();
                }
                const double mag = sqrt(sh[0][0] * sh[0][0] + sh[1][0] * sh[1][0]);
                __syncthreads();
                if (in) { s1 += mag; s2 += mag * mag; }
            }
        }
    }
    if (tid == 0) { a.part_s[((long long)b * a.n1 + k1) * 2] = s1; a.part_s[((long long)b * a.n1 + k1) * 2 + 1] = s2; }
}

template <int UNUSED = 0>
__global__ __launch_bounds__(256) void k_extra_final(ExtraArgs a, const twx_result* __restrict__ res, int res_stride, const long long* __restrict__ dfidx) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ double sh[5][256];
    double v[5] = {0, 0, 0, 0, 0};
    if (a.bruit_len > 0) for (int i = tid; i < a.nblk_b; i += 256) { const double* p = a.part_b + ((long long)b * a.nblk_b + i) * 3; v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; }
    if (a.sq_len > 0 && a.sqmax) for (int i = tid; i < a.n1; i += 256) { const double* p = a.part_s + ((long long)b * a.n1 + i) * 2; v[3] += p[0]; v[4] += p[1]; }
    for (int j = 0; j < 5; ++j) sh[j][tid] = v[j];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) for (int j = 0; j < 5; ++j) sh[j][tid] += sh[j][tid + d];
        __syncthreads();
    }
    if (tid == 0) {
        twx_extra e;
        const double qn = nan("");
        e.bruit = e.valmax_square = e.noise_square = qn; e.reserved = 0;
        const long long M = a.n * a.nphase;
        if (a.bruit_len > 1 && res[(long long)b * res_stride].indice0 + 20 + a.bruit_len < M) {
            const double L = (double)a.bruit_len;
            e.bruit = (sh[2][0] - (sh[0][0] * sh[0][0] + sh[1][0] * sh[1][0]) / L) / (L - 1.0);       // Octave var of a complex vector: sum |z - mean|^2 / (L - 1)
        }
        if (a.sqmax) {
            e.valmax_square = a.sqmax[b];
            const long long i0 = dfidx[b];
            if (a.sq_len > 1 && i0 >= 0 && i0 + 20 + a.sq_len - 1 <= a.n - 1) {
                const double L = (double)a.sq_len;
                e.noise_square = (sh[4][0] - sh[3][0] * sh[3][0] / L) / (L - 1.0);
            }
        }
        a.out[(long long)b * a.out_stride] = e;
    }
}

// ------------------------------------------------------------------------------------------
// k_peak: finish the arg-max, re-evaluate prnmap around the peak in fp64 from Bz, parabolic
// correction (godual_ranging.m:29-33) and the wipe-off statistics (:38-48).  grid = windows
// ------------------------------------------------------------------------------------------
#define TWX_PEAK_LO (-4)
#define TWX_PEAK_HI (7)
#define TWX_PEAK_NP (TWX_PEAK_HI - TWX_PEAK_LO + 1)

template <typename T> struct PeakArgs {
    long long n; int n1, n2, nphase, nparts;
    const ArgPart<T>* part;
    const cpx<T>* Bz;
    const cpx<double>* tw1d;     // exp(-2 pi i m/N1) in fp64
    const WinSums* sums; int remove_mean;
    const cpx<T>* dc;
    const double* dfv; const long long* dfidx;
    double inv_scale;            // undoes RowArgs::scale
    int var_ddof, snr_rot;
    int convention;              // TWX_CONV_*
    int snr_valid;               // 0: replica is not a +-1 code, the wipe-off statistics are undefined
    twx_result* res;             // record of window b at res[b * res_stride]
    int res_stride;              // 1, or the channel count when all channels of a window are interleaved in the output
    const int* chk_flag;         // TWX_OPT_SELFCHECK: the window's status word of k_chk_verdict (nullptr: off)
    const int* rs_dt; const int* rs_edge;   // twx_set_resample: the window's carried dt (-> twx_result.dt) and edge word (bit 2: the whole map is NaN)
    int snr_only;                // 1: second call of a Hamming-window context — Bz now holds the correlation with the UNWINDOWED
                                 // replica; the peak is the one already in the record, only the wipe-off statistics are written
};

template <typename T>
__global__ __launch_bounds__(1024) void k_peak(PeakArgs<T> a) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ char scratch[256];
    __shared__ unsigned int s_idx;
    __shared__ double s_z[TWX_PEAK_NP][2];
    Best<T> best; best.val = T(-1); best.idx = 0xffffffffu;
    for (int i = tid; i < a.nparts; i += 1024) {
        ArgPart<T> p = a.part[(long long)b * a.nparts + i];
        best.take(p.val, p.idx);
    }
    best = block_best<T, 1024>(best, scratch);
    const long long M = a.n * a.nphase;
    if (tid == 0) {
        s_idx = best.idx;
        if (a.snr_only) {
            const long long i0 = a.res[(long long)b * a.res_stride].indice0;
            s_idx = (unsigned int)(a.convention == TWX_CONV_CLAUDIO ? (M - i0) % M : i0);
        }
    }
    __syncthreads();
    const long long mstar = s_idx;
    const int lane = tid & 63, wv = tid >> 6;
    for (int pt = wv; pt < TWX_PEAK_NP; pt += 16) {
        long long m = (mstar + TWX_PEAK_LO + pt) % M; if (m < 0) m += M;
        const int rho = (int)(m % a.nphase);
        const long long q = m / a.nphase;
        const int q1 = (int)(q / a.n2), q2 = (int)(q % a.n2);
        const cpx<T>* src = a.Bz + ((long long)b * a.nphase + rho) * a.n + q2;
        double sx = 0, sy = 0;
        for (int k1 = lane; k1 < a.n1; k1 += 64) {
#if TWX_BZ16
            cpx<T> v;
            if constexpr (sizeof(T) == 4) {
                const unsigned* s16 = reinterpret_cast<const unsigned*>(a.Bz) + ((long long)b * a.nphase + rho) * a.n + q2;
                const float2 f = bz16_unpack(s16[(long long)k1 * a.n2]);
                v = mk<T>((T)f.x, (T)f.y);
            } else v = src[(long long)k1 * a.n2];
#else
            cpx<T> v = src[(long long)k1 * a.n2];
#endif
            cpx<double> w = a.tw1d[(int)(((long long)k1 * q1) % a.n1)];   // conj → inverse
            sx += (double)v.x * w.x + (double)v.y * w.y;
            sy += (double)v.y * w.x - (double)v.x * w.y;
        }
        for (int d = 32; d >= 1; d >>= 1) { sx += __shfl_down(sx, d, 64); sy += __shfl_down(sy, d, 64); }
        if (lane == 0) { s_z[pt][0] = sx * a.inv_scale / (double)M; s_z[pt][1] = sy * a.inv_scale / (double)M; }
    }
    __syncthreads();
    if (tid == 0) {
        twx_result r;
        const int c = -TWX_PEAK_LO;
        int rot = a.snr_rot;
        if (a.convention == TWX_CONV_CLAUDIO) {
            // prnmap_c[m] = conj(prnmap_g[(M - m) mod M])  (fcode.*conj(ffty), claudio…separate.m:59):
            // mirrored index, conjugated samples, m1/p1 swapped; the code-rotating wipe-off of :90-94
            // sums prnmap_g[m*+1 .. m*+R] (absolute 0..R-1 when the 1-based peak index is <= 2)
            const long long ic = (M - mstar) % M;
            r.indice0 = ic;
            for (int i = 0; i < 7; ++i) { r.zwin[i][0] = s_z[c + 3 - i][0]; r.zwin[i][1] = -s_z[c + 3 - i][1]; }
            rot = (ic + 1 > 2) ? 1 : (int)((M - mstar) % M);   // ic in {0,1}: absolute indices 0.. ⇒ offset ic from m*
        } else {
            r.indice0 = mstar;
            for (int i = 0; i < 7; ++i) { r.zwin[i][0] = s_z[c - 3 + i][0]; r.zwin[i][1] = s_z[c - 3 + i][1]; }
        }
        r.xval[0] = r.zwin[3][0]; r.xval[1] = r.zwin[3][1];
        r.xvalm1[0] = r.zwin[2][0]; r.xvalm1[1] = r.zwin[2][1];
        r.xvalp1[0] = r.zwin[4][0]; r.xvalp1[1] = r.zwin[4][1];
        const double am = hypot(r.xvalm1[0], r.xvalm1[1]), a0 = hypot(r.xval[0], r.xval[1]), ap = hypot(r.xvalp1[0], r.xvalp1[1]);
        r.correction = (am - ap) / (am + ap - 2 * a0) / 2;                 // godual_ranging.m:33
        r.df = a.dfv[b];
        r.df_index = a.dfidx[b];
        // wipe-off mean = sum_{i<R} prnmap[indice+rot+i] / M   
        double mr = 0, mi = 0;
        int ok = 1;
        for (int i = 0; i < a.nphase; ++i) {
            const int o = rot + i;
            if (o < TWX_PEAK_LO || o > TWX_PEAK_HI) { ok = 0; break; }
            mr += s_z[c + o][0]; mi += s_z[c + o][1];
        }
        mr /= (double)M; mi /= (double)M;
        const double N = (double)a.n;
        WinSums s = a.sums[b];
        double mI = 0, mQ = 0;
        if (a.remove_mean) { mI = (double)s.sI / N; mQ = (double)s.sQ / N; }
        // sum|y|^2 = sum|d-mean|^2 (|lo|=1);   mean(y) = X[0]/N
        const double sumsq = s.is_f ? s.fP : (double)s.sP - 2.0 * (mI * (double)s.sI + mQ * (double)s.sQ) + N * (mI * mI + mQ * mQ);
        const double ybx = (double)a.dc[b].x / N, yby = (double)a.dc[b].y / N;
        r.puissance = (sumsq - N * (ybx * ybx + yby * yby)) / (N - a.var_ddof);          // var(y), :46
        const double R2 = (double)a.nphase * (double)a.nphase;
        const double sum_yint = (double)M * (sumsq / N) / R2;                             // sum|yint|^2
        const double var = (sum_yint - (double)M * (mr * mr + mi * mi)) / ((double)M - a.var_ddof);
        if (!a.snr_valid) ok = 0;
        r.SNRr = ok ? mr * mr / var : nan("");
        r.SNRi = ok ? mi * mi / var : nan("");
        r.puissancecode = ok ? mr * mr + mi * mi : nan("");
        r.puissancenoise = ok ? var : nan("");
        r.status = (a.chk_flag && a.chk_flag[b]) ? TWX_STATUS_SELFCHECK : 0; r.dt = a.rs_dt ? a.rs_dt[b] : 0;
        if (a.rs_edge && (a.rs_edge[b] & 4)) {
            // more than the two edge samples of the resampled window fell outside it: interp1 leaves NaN in yi, the whole map is NaN and
            // Octave's max returns index 1 (godual_ranging_OP_vitesse.m:40-48)
            const double qn = nan("");
            r.indice0 = 0; r.correction = qn;
            r.xval[0] = r.xval[1] = r.xvalm1[0] = r.xvalm1[1] = r.xvalp1[0] = r.xvalp1[1] = qn;
            for (int i = 0; i < 7; ++i) r.zwin[i][0] = r.zwin[i][1] = qn;
            r.SNRr = r.SNRi = r.puissancecode = r.puissancenoise = qn;
            r.status |= TWX_STATUS_RESAMPLE_NAN;
        }
        if (a.snr_only) {                                                  // everything but the wipe-off statistics stays as the first call left it
            twx_result o = a.res[(long long)b * a.res_stride];
            o.SNRr = r.SNRr; o.SNRi = r.SNRi; o.puissancecode = r.puissancecode; o.puissancenoise = r.puissancenoise;
            o.status |= r.status;
            r = o;
        }
        a.res[(long long)b * a.res_stride] = r;
    }
}

}  // namespace twx
