// Diagnostic (tools/chain_corunner.py): fill the LDS of every CU with a bit pattern and leave — what does a kernel that follows read
// from LDS it has not written?   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_poison.hip -o tools/bin/liblds_poison.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void k_poison(unsigned pattern, int words, int spin) {
    extern __shared__ unsigned L[];
    for (int i = threadIdx.x; i < words; i += 256) L[i] = pattern;
    __syncthreads();
    unsigned acc = 0;
    for (int s = 0; s < spin; ++s) { for (int i = threadIdx.x; i < words; i += 256) acc += L[i]; __builtin_amdgcn_s_sleep(20); }
    if (acc == 0x12345u) L[0] = acc;          // keep the loop
}
extern "C" int lds_poison(void* stream, unsigned pattern, int kbytes, int blocks, int spin) {
    static bool set = false;
    if (!set) { if (hipFuncSetAttribute((const void*)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; set = true; }
    hipLaunchKernelGGL(k_poison, dim3(blocks), dim3(256), (size_t)kbytes * 1024, (hipStream_t)stream, pattern, kbytes * 256, spin);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
// a matrix-core burner that shares CUs with whatever else runs: 256 threads, little LDS, v_mfma_f32_16x16x32_f16 on register operands
typedef _Float16 bh8 __attribute__((ext_vector_type(8)));
typedef float bf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_burn(unsigned abits, unsigned bbits, int iters, float* sink) {
    uint4 ua = {abits, abits, abits, abits}, ub = {bbits + threadIdx.x % 3, bbits, bbits, bbits};
    bh8 a = __builtin_bit_cast(bh8, ua), b = __builtin_bit_cast(bh8, ub);
    bf4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_burn32(float av, float bv, int iters, float* sink) {
    bf4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const float a = av + (threadIdx.x % 3) * 0.f, b = bv;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
// the same with operands that CHANGE every iteration (normal fp16 numbers in [1, 2) from a running counter): live data
__global__ __launch_bounds__(256) void k_burn_live(int iters, float* sink) {
    unsigned s0 = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x, s1 = 0x85EBCA6Bu * (threadIdx.x + 7);
    bf4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        s0 = s0 * 1664525u + 1013904223u; s1 = s1 * 22695477u + 1u;
        uint4 ua = {(s0 & 0x03FF03FFu) | 0x3C003C00u, ((s0 >> 3) & 0x03FF03FFu) | 0x3C003C00u, ((s0 >> 5) & 0x03FF03FFu) | 0xBC003C00u, ((s0 >> 7) & 0x03FF03FFu) | 0x3C00BC00u};
        uint4 ub = {(s1 & 0x03FF03FFu) | 0x3C003C00u, ((s1 >> 3) & 0x03FF03FFu) | 0xBC003C00u, ((s1 >> 5) & 0x03FF03FFu) | 0x3C003C00u, ((s1 >> 7) & 0x03FF03FFu) | 0x3C00BC00u};
        const bh8 a = __builtin_bit_cast(bh8, ua), b = __builtin_bit_cast(bh8, ub);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}
// closer to k_fir_mfma: 512 threads, 52 KB of LDS; every iteration a wave writes words into LDS (ds_write_b32), reads 16-byte operands back
// (ds_read_b128) and feeds them to the matrix cores; optionally 16-byte global loads feed the writes
__global__ __launch_bounds__(512) void k_burn_lds(int iters, const uint4* src, int nsrc, float* sink) {
    extern __shared__ unsigned L[];
    const int tid = threadIdx.x, words = 13 * 1024;
    unsigned s0 = 0x9E3779B9u * (tid + 1) + blockIdx.x;
    bf4 c0 = {0, 0, 0, 0}, c1 = c0;
    for (int i = tid; i < words; i += 512) L[i] = 0x3C003C00u;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        uint4 g = {0, 0, 0, 0};
        if (src) g = src[(blockIdx.x * 512 + tid + i * 7919) % nsrc];
        s0 = s0 * 1664525u + 1013904223u + g.x;
        for (int k = 0; k < 4; ++k) L[(tid * 4 + k * 2053 + i) % words] = ((s0 >> k) & 0x03FF03FFu) | 0x3C003C00u;
        __syncthreads();
        const uint4 ua = *reinterpret_cast<const uint4*>(L + ((tid * 4 + i * 4) % (words - 4) & ~3));
        const uint4 ub = *reinterpret_cast<const uint4*>(L + ((tid * 4 + 2048 + i * 8) % (words - 4) & ~3));
        const bh8 a = __builtin_bit_cast(bh8, ua), b = __builtin_bit_cast(bh8, ub);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, b, c1, 0, 0, 0);
        __syncthreads();
    }
    if (c0[0] + c1[1] == 12345.f) sink[0] = 1.f;
}
extern "C" int mfma_burn_lds(void* stream, int blocks, int iters, const void* src, int nsrc, float* sink) {
    static bool set = false;
    if (!set) { if (hipFuncSetAttribute((const void*)k_burn_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1; set = true; }
    hipLaunchKernelGGL(k_burn_lds, dim3(blocks), dim3(512), (size_t)52 * 1024, (hipStream_t)stream, iters, (const uint4*)src, nsrc, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mfma_burn_live(void* stream, int blocks, int iters, float* sink) {
    hipLaunchKernelGGL(k_burn_live, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mfma_burn32(void* stream, float a, float b, int blocks, int iters, float* sink) {
    hipLaunchKernelGGL(k_burn32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mfma_burn(void* stream, unsigned abits, unsigned bbits, int blocks, int iters, float* sink) {
    hipLaunchKernelGGL(k_burn, dim3(blocks), dim3(256), 0, (hipStream_t)stream, abits, bbits, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
