// power_probe.hip — sustained synthetic loads for reading board power / clocks with rocm-smi (tools/history/gpu_power2.sh):
//   power_probe read|copy|valu|pkvalu|lds|mix  [seconds]
// read: 16-B streaming reads of a 4 GiB buffer; copy: read + write; valu: v_fma_f32 on registers, every CU, 4 waves per SIMD;
// pkvalu: v_pk_fma_f32; lds: ds_read_b64 / ds_write_b64 round trips; mix: read stream and packed FMAs in the same waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 7 * stride < n; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[i + u * stride] = v[u];
    }
}
template <int PK>
__global__ __launch_bounds__(256) void k_valu(float* out, float seed, int iters) {
    float a[16]; f2 p[16];
    const float b = seed + threadIdx.x * 1e-6f, c = 1.0f - seed;
    const f2 pb = {b, b * 0.5f}, pc = {c, c * 0.25f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed * i; p[i] = f2{seed * i, seed + i}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
    __shared__ f2 buf[4096];
    f2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = f2{(float)threadIdx.x, (float)u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) buf[(threadIdx.x + u * 256 + it) & 4095] = v[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf[(threadIdx.x * 3 + u * 257 + it) & 4095];
        __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u].x;
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ p, size_t n, float* sink, int fma_per_load) {
    f2 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = f2{0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 7 * stride < n; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
        for (int k = 0; k < fma_per_load; ++k) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { f2 x = {v[u].x, v[u].y}, y = {v[u].z, v[u].w}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(x), "v"(y)); }
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += acc[u].x + acc[u].y;
    if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "read";
    const double secs = argc > 2 ? atof(argv[2]) : 4.0;
    const int arg3 = argc > 3 ? atoi(argv[3]) : 4;
    const size_t bytes = 4ull << 30;
    void *a, *b; float* sink;
    CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(a, 1, bytes)); CHK(hipMemset(b, 2, bytes));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const size_t n16 = bytes / 16;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0; double unit = 0; const char* uname = "";
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int r = 0; r < 8; ++r) {
            if (!strcmp(mode, "read")) { hipLaunchKernelGGL(k_read, dim3(16384), dim3(256), 0, 0, (const float4*)a, n16, sink); unit = bytes / 1e12; uname = "TB"; }
            else if (!strcmp(mode, "copy")) { hipLaunchKernelGGL(k_copy, dim3(16384), dim3(256), 0, 0, (const float4*)a, (float4*)b, n16); unit = 2.0 * bytes / 1e12; uname = "TB"; }
            else if (!strcmp(mode, "valu")) { hipLaunchKernelGGL((k_valu<0>), dim3(ncu * 4), dim3(256), 0, 0, sink, 0.5f, 20000); unit = (double)ncu * 4 * 256 * 20000 * 16 * 2 / 1e12; uname = "TFLOP"; }
            else if (!strcmp(mode, "pkvalu")) { hipLaunchKernelGGL((k_valu<1>), dim3(ncu * 4), dim3(256), 0, 0, sink, 0.5f, 20000); unit = (double)ncu * 4 * 256 * 20000 * 16 * 4 / 1e12; uname = "TFLOP"; }
            else if (!strcmp(mode, "lds")) { hipLaunchKernelGGL(k_lds, dim3(ncu * 4), dim3(256), 0, 0, sink, 4000); unit = (double)ncu * 4 * 256 * 4000 * 16 * 8 / 1e12; uname = "TB(LDS)"; }
            else { hipLaunchKernelGGL(k_mix, dim3(16384), dim3(256), 0, 0, (const float4*)a, n16, sink, arg3); unit = bytes / 1e12; uname = "TB"; }
            ++launches;
        }
        CHK(hipDeviceSynchronize());
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s: %.2f s, %ld launches, %.2f %s/s\n", mode, dt, launches, launches * unit / dt, uname);
    return 0;
}
