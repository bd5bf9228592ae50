// valu_probe.hip — issue rate of the fp32 vector instructions the FFT butterflies are made of, on this MI355X:
// plain (v_fma_f32, v_add_f32, v_mul_f32) against packed (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32), at 1, 2 and 4
// waves per SIMD.  Every kernel runs ITER x 32 independent instructions per wave on register operands only.
//   build: make -C amaranth_twstft_amd/csrc probe      run: tools/bin/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k_valu(float* out, float seed) {
    float a[16]; f2 p[16];
    const float b = seed + threadIdx.x * 1e-6f, c = 1.0f - seed;
    const f2 pb = {b, b * 0.5f}, pc = {c, c * 0.25f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed * i; p[i] = f2{seed * i, seed + i}; }
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 3) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 4) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 5) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 6) {   // the 2-instruction complex multiply of twx_fft.h (op_sel / neg modifiers)
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X)
#undef X
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X)
#undef X
        } else if (OP == 7) {   // dependent chain of v_fma_f32 (latency)
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 8) {   // dependent chain of v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(pb), "v"(pc));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 9) {   // alternating plain fma / packed add (mixed stream)
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_pk_add_f32 %1, %1, %4" : "+v"(a[i]), "+v"(p[i]) : "v"(b), "v"(c), "v"(pb));
            REP16(X)
#undef X
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int OP> static void run(const char* name, int flop_per_lane_instr, float* out) {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    for (int wps : {1, 2, 4}) {                  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
        const int blocks = ncu * wps;
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        hipLaunchKernelGGL((k_valu<OP>), dim3(blocks), dim3(256), 0, 0, out, 0.5f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a));
        hipLaunchKernelGGL((k_valu<OP>), dim3(blocks), dim3(256), 0, 0, out, 0.5f);
        CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_simd = (double)ITER * 32 * wps;             // wave-instructions issued by one SIMD
        const double clk = ms * 1e-3 * 2.4e9;                              // at the nominal 2.4 GHz
        printf("%-34s waves/SIMD %d  %7.3f ms  %6.2f clk per wave-instruction (2.4 GHz)  %7.1f TFLOP/s\n", name, wps, ms,
               clk / instr_per_simd, instr_per_simd * 4 * ncu * 64.0 * flop_per_lane_instr / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float* out; CHK(hipMalloc(&out, 64));
    run<0>("v_fma_f32", 2, out);
    run<1>("v_add_f32", 1, out);
    run<2>("v_mul_f32", 1, out);
    run<3>("v_pk_fma_f32", 4, out);
    run<4>("v_pk_add_f32", 2, out);
    run<5>("v_pk_mul_f32", 2, out);
    run<6>("cmul = v_pk_mul + v_pk_fma (op_sel)", 3, out);
    run<7>("v_fma_f32 dependent chain", 2, out);
    run<8>("v_pk_fma_f32 dependent chain", 4, out);
    run<9>("v_fma_f32 + v_pk_add_f32 pairs", 2, out);
    return 0;
}
