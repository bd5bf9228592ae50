// bw_probe.hip — what HBM sustains on this MI355X for the access shapes the correlator's kernels use.
// Hand-written read / write / copy sweeps (no library calls), buffers far larger than the 256 MiB Infinity Cache.
//   build: make -C amaranth_twstft_amd/csrc probe      run: tools/bin/bw_probe [GiB]
// Patterns
//   read16        16 B per lane, consecutive lanes consecutive addresses, U loads in flight per lane
//   read8         8 B per lane (one complex fp32), same sweep
//   write16       16 B per lane stores
//   copy16        read16 + write16 (bytes counted both ways)
//   colinv        k_col_inv's read: a workgroup of 448 threads owns 16 adjacent columns of a [625][8000] complex matrix:
//                 625 pieces of 128 B at a 64 000-B stride, 8 B per lane, 25 loads in flight per lane
//   colinv32      the same with 32 columns per workgroup (256-B pieces, 16 B per lane)
//   rowmid        k_rowd<MID>'s traffic: gather one 64 000-B row from 500 tile blocks (128-B pieces at an 80 000-B
//                 stride), write three contiguous 64 000-B rows
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float2 v, float2* p) { v2f t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<v2f*>(p)); }
__device__ __forceinline__ void nt_store(float4 v, float4* p) { v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p)); }
__device__ __forceinline__ float2 nt_load(const float2* p) { v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)); return make_float2(t.x, t.y); }
__device__ __forceinline__ float4 nt_load(const float4* p) { v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(t.x, t.y, t.z, t.w); }
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename V, int U>
__global__ __launch_bounds__(256) void k_read(const V* __restrict__ p, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// each block sweeps its own contiguous chunk (what a tiled kernel does) instead of a grid-stride sweep
template <typename V, int U>
__global__ __launch_bounds__(256) void k_read_chunk(const V* __restrict__ p, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t per = n / gridDim.x;
    const V* q = p + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = q[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_write16(float4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) p[i + u * stride] = v;
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
    }
}
template <int U, int NT>
__global__ __launch_bounds__(256) void k_write8(float2* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float2 v = make_float2((float)i, 1.f);
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) nt_store(v, p + i + u * stride); else p[i + u * stride] = v; }
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_write16nt(float4* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float4 v = make_float4((float)i, 1.f, 2.f, 3.f);
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) nt_store(v, p + i + u * stride);
    }
}
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, rem = nblk & 7u, xcd = bid & 7u, pos = bid >> 3;
    return xcd < rem ? xcd * (q + 1) + pos : rem * (q + 1) + (xcd - rem) * q + pos;
}
// matrices [N1=625][N2=8000] of V-sized... elements are 8 B; W columns per workgroup, VW elements per lane-load
template <int W, typename V, int REMAP, int LDSB = 0, int NTL = 0>
__global__ __launch_bounds__(448) void k_colinv(const float2* __restrict__ m, int nmat, float* __restrict__ sink) {
    constexpr int N1 = 625, N2 = 8000, EPL = sizeof(V) / 8, LW = W / EPL;     // lanes per 128/256-B piece
    __shared__ float dummy[LDSB / 4 + 1];                                      // LDSB > 0: limits the workgroups per CU like the real kernel
    if (LDSB && threadIdx.x == 447) dummy[(blockIdx.x * 7) % (LDSB / 4 + 1)] = 1.f;
    constexpr int ROWS_PER_PASS = 400 / LW;                                    // 25 (W=16) — rows j, j+25, ...
    const unsigned logical = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int ntiles = N2 / W;
    const int tile = logical % ntiles, mat = logical / ntiles;
    const float2* src = m + (size_t)mat * N1 * N2 + (size_t)tile * W;
    const int tid = threadIdx.x;
    float acc = 0.f;
    if (tid < ROWS_PER_PASS * LW) {
        const int j = tid / LW, c = tid % LW;
        V v[N1 / ROWS_PER_PASS];
#pragma unroll
        for (int r = 0; r < N1 / ROWS_PER_PASS; ++r) {
            const V* q = reinterpret_cast<const V*>(src + (size_t)(j + r * ROWS_PER_PASS) * N2 + c * EPL);
            v[r] = NTL ? nt_load(q) : *q;
        }
#pragma unroll
        for (int r = 0; r < N1 / ROWS_PER_PASS; ++r) acc += v[r].x;
    }
    if (LDSB) { __syncthreads(); acc += dummy[threadIdx.x % (LDSB / 4 + 1)]; }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int NTS, int LDSB>
__global__ __launch_bounds__(448) void k_rowmid(const float2* __restrict__ A, float2* __restrict__ Bz, int nwin, float* __restrict__ sink) {
    constexpr int N1 = 625, N2 = 8000, M = 400, R0 = 20;
    __shared__ float dummy[LDSB / 4 + 1];
    if (LDSB && threadIdx.x == 447) dummy[(blockIdx.x * 7) % (LDSB / 4 + 1)] = 1.f;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int k1 = logical / nwin, b = logical % nwin;
    const float2* Ab = A + (size_t)b * N1 * N2;
    const int tid = threadIdx.x;
    if (tid >= M) return;
    float2 v[R0];
#pragma unroll
    for (int r = 0; r < R0; ++r) {
        const unsigned n2 = tid + r * M;
        v[r] = Ab[(((n2 >> 4) * N1 + k1) << 4) | (n2 & 15u)];
    }
    for (int rho = 0; rho < 3; ++rho) {
        float2* out = Bz + ((size_t)b * 3 + rho) * N1 * N2 + (size_t)k1 * N2;
#pragma unroll
        for (int r = 0; r < R0; ++r) {
            const float2 o = make_float2(v[r].x + rho, v[r].y);
            if (NTS) nt_store(o, out + r * M + tid); else (out + r * M)[tid] = o;
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_random(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
        p[i] = 0x3F000000u | ((unsigned)z & 0x007FFFFFu) | ((unsigned)(z >> 40) & 0x80000000u);     // floats of magnitude 0.5..1, random mantissa and sign
    }
}
template <class F> static double time_ms(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms = 0; CHK(hipEventElapsedTime(&ms, a, b));
    CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
    return ms / reps;
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095;
    void *a = nullptr, *b = nullptr; float* sink = nullptr;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    // argv[2] = "rand": the source buffer holds pseudo-random floats (what a kernel reads in production: every bit of the bus toggles)
    // instead of one repeated byte — the difference is what constant test data flatter a memory system by
    const bool randomised = argc > 2 && strcmp(argv[2], "rand") == 0;
    if (randomised) { hipLaunchKernelGGL(k_fill_random, dim3(8192), dim3(256), 0, 0, (unsigned*)a, bytes / 4); CHK(hipDeviceSynchronize()); }
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, buffers 2 x %.2f GiB%s\n", prop.name, prop.multiProcessorCount, gib, randomised ? ", source: random floats" : ", source: one repeated byte");
    const int reps = 10;
    auto report = [&](const char* name, double ms, double moved) { printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, moved / ms / 1e6); fflush(stdout); };
    const size_t n16 = bytes / 16, n8 = bytes / 8;
    for (int blocks : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read16 grid-stride U=4 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<float4, 4>), dim3(blocks), dim3(256), 0, 0, (const float4*)a, n16, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "read16 grid-stride U=8 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<float4, 8>), dim3(blocks), dim3(256), 0, 0, (const float4*)a, n16, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "read16 chunked     U=8 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read_chunk<float4, 8>), dim3(blocks), dim3(256), 0, 0, (const float4*)a, n16, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "read8  grid-stride U=8 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<float2, 8>), dim3(blocks), dim3(256), 0, 0, (const float2*)a, n8, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "read8  grid-stride U=16 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<float2, 16>), dim3(blocks), dim3(256), 0, 0, (const float2*)a, n8, sink); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "write16 U=4 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_write16<4>), dim3(blocks), dim3(256), 0, 0, (float4*)b, n16); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "copy16 U=4 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_copy16<4>), dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, n16); }, reps), 2.0 * bytes);
    }
    // column-pass gather: matrices of 625 x 8000 complex fp32 = 40 MB each
    const int nmat = (int)(bytes / (625ull * 8000 * 8));
    const double mbytes = (double)nmat * 625 * 8000 * 8;
    report("colinv  W=16 8B/lane  (k_col_inv shape) remap", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 1>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=16 8B/lane  no remap", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 0>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=32 16B/lane (256-B pieces) remap", time_ms([&] { hipLaunchKernelGGL((k_colinv<32, float4, 1>), dim3(nmat * 250), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=16 8B/lane 80 KB LDS (2 WG/CU)", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 1, 80000>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=16 8B/lane 52 KB LDS (3 WG/CU)", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 1, 52000>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=16 8B/lane nt loads", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 1, 0, 1>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    report("colinv  W=16 8B/lane nt loads 80 KB LDS", time_ms([&] { hipLaunchKernelGGL((k_colinv<16, float2, 1, 80000, 1>), dim3(nmat * 500), dim3(448), 0, 0, (const float2*)a, nmat, sink); }, reps), mbytes);
    {   // row-mid: nwin windows of A (40 MB) in, 3x out
        const int nwin = (int)(bytes / (3ull * 625 * 8000 * 8));
        const double moved = (double)nwin * 625 * 8000 * 8 * 4;
        report("rowmid  gather 1 row + write 3 rows", time_ms([&] { hipLaunchKernelGGL((k_rowmid<0, 0>), dim3(625 * nwin), dim3(448), 0, 0, (const float2*)a, (float2*)b, nwin, sink); }, reps), moved);
        report("rowmid  ... nt stores", time_ms([&] { hipLaunchKernelGGL((k_rowmid<1, 0>), dim3(625 * nwin), dim3(448), 0, 0, (const float2*)a, (float2*)b, nwin, sink); }, reps), moved);
        report("rowmid  ... 78 KB LDS (2 WG/CU)", time_ms([&] { hipLaunchKernelGGL((k_rowmid<0, 78000>), dim3(625 * nwin), dim3(448), 0, 0, (const float2*)a, (float2*)b, nwin, sink); }, reps), moved);
        report("rowmid  ... 78 KB LDS nt stores", time_ms([&] { hipLaunchKernelGGL((k_rowmid<1, 78000>), dim3(625 * nwin), dim3(448), 0, 0, (const float2*)a, (float2*)b, nwin, sink); }, reps), moved);
        report("rowmid  ... 52 KB LDS (3 WG/CU)", time_ms([&] { hipLaunchKernelGGL((k_rowmid<0, 52000>), dim3(625 * nwin), dim3(448), 0, 0, (const float2*)a, (float2*)b, nwin, sink); }, reps), moved);
    }
    for (int blocks : {8192, 32768}) {
        char nm[96];
        snprintf(nm, sizeof nm, "write8  U=8 blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_write8<8, 0>), dim3(blocks), dim3(256), 0, 0, (float2*)b, n8); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "write8  U=8 nt blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_write8<8, 1>), dim3(blocks), dim3(256), 0, 0, (float2*)b, n8); }, reps), (double)bytes);
        snprintf(nm, sizeof nm, "write16 U=4 nt blocks=%d", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_write16nt<4>), dim3(blocks), dim3(256), 0, 0, (float4*)b, n16); }, reps), (double)bytes);
    }
    CHK(hipFree(a)); CHK(hipFree(b)); CHK(hipFree(sink));
    return 0;
}
