// Measurement aid (not part of the library): the fp32-MFMA rate the chip SUSTAINS, as a function of how long the
// matrix pipes are kept busy.  A kernel of back-to-back independent v_mfma_f32_32x32x2_f32 (4 accumulators per wave,
// `waves` waves per SIMD on every CU), launched for ~0.1 ms ... ~50 ms; prints TFLOP/s per duration.  The datasheet
// peak (157.3 TFLOP/s) assumes 2.4 GHz; under a sustained fp32-matrix load the chip clocks to its power budget, and
// THAT rate is what a dense product can be held against.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/mfma_peak && tools/probe/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = seed + threadIdx.x * 1e-3f, y = seed - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 2; ++wps) {
        for (int iters : {64, 256, 1024, 4096, 16384, 65536}) {
            const int grid = 256 * wps;                     // 256-thread workgroups: one wave per SIMD each
            hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, 16, 0.5f);
            CK(hipDeviceSynchronize());
            const int reps = iters >= 16384 ? 3 : 10;
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = (double)reps * grid * 4.0 * iters * 32.0 * 4096.0;     // waves x MFMAs x flop
            printf("waves/SIMD %d  %6d iterations: %8.3f ms per launch  %7.1f TFLOP/s  (%.0f%% of 157.3)\n", wps, iters,
                   ms / reps, flops / (ms * 1e-3) / 1e12, 100.0 * flops / (ms * 1e-3) / 1e12 / 157.3);
        }
    }
    return 0;
}
