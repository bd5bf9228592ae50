// Diagnostic (round 6): what v_permlane32_swap / v_permlane16_swap + DPP row sums leave where — the reduce-scatter of k_rowd_bandsum
// (twx_kernels.h: wave_sum_scatter) against plain sums.   hipcc --offload-arch=gfx950 -O3 -I amaranth_twstft_amd/csrc tools/permlane_probe.hip -o tools/bin/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "twx_kernels.h"
using namespace twx;
constexpr int CH = 10;
__global__ void k(const float* in, float* out) {
    cpx<float> z[CH];
    for (int i = 0; i < CH; ++i) { z[i].x = in[(threadIdx.x * CH + i) * 2]; z[i].y = in[(threadIdx.x * CH + i) * 2 + 1]; }
    float o[CH / 2];
    wave_sum_scatter<CH>(z, o);
    for (int i = 0; i < CH / 2; ++i) out[threadIdx.x * (CH / 2) + i] = o[i];
}
int main() {
    float h[64 * CH * 2], *d, *o, ho[64 * CH / 2];
    for (int l = 0; l < 64; ++l) for (int i = 0; i < CH; ++i) { h[(l * CH + i) * 2] = (float)((l * 7 + i * 3) % 11 - 5); h[(l * CH + i) * 2 + 1] = (float)((l * 5 + i) % 13 - 6); }
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int row = 0; row < 4; ++row) for (int i = 0; i < CH / 2; ++i) {
        const int q = i + (row & 1) * (CH / 2), comp = row >> 1;
        float want = 0; for (int l = 0; l < 64; ++l) want += h[(l * CH + q) * 2 + comp];
        for (int l = row * 16; l < row * 16 + 16; ++l) if (ho[l * (CH / 2) + i] != want) { if (bad < 8) printf("row %d i %d lane %d: got %g want %g\n", row, i, l, ho[l * (CH / 2) + i], want); ++bad; }
    }
    printf("wave_sum_scatter<%d>: %d mismatches of %d\n", CH, bad, 64 * CH / 2);
    return bad != 0;
}
