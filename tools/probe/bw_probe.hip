// Read-bandwidth floor for the Sinkhorn row pass shapes: 64 MiB (4096 x 4096 f32), one wave per row.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int WAVES, int STAGE>
__global__ __launch_bounds__(64 * WAVES) void k_oneshot(const float* __restrict__ M, int n, const double* __restrict__ v,
                                                        float* __restrict__ out) {
    extern __shared__ double vs[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * WAVES + wv;
    const float* row = M + (size_t)r * n;
    float4 c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = *reinterpret_cast<const float4*>(row + lane * 4 + 256 * k);
    float acc = 0.f;
    if (STAGE) {
        for (int j = threadIdx.x * 2; j < n; j += 128 * WAVES)
            *reinterpret_cast<double2*>(vs + j) = *reinterpret_cast<const double2*>(v + j);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double4 v4 = *reinterpret_cast<const double4*>(vs + lane * 4 + 256 * k);
            acc += c[k].x * (float)v4.x + c[k].y * (float)v4.y + c[k].z * (float)v4.z + c[k].w * (float)v4.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += c[k].x + c[k].y + c[k].z + c[k].w;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[r] = acc;
}

// column-pass shape: lane owns 4 columns, wave streams 8 rows at a time over a strip
__global__ __launch_bounds__(256) void k_strip(const float* __restrict__ M, int n, int rows_per_chunk, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 256 + lane * 4;
    const int r_beg = blockIdx.y * rows_per_chunk, r_end = r_beg + rows_per_chunk;
    float4 a = make_float4(0, 0, 0, 0);
    for (int r0 = r_beg + wv * 8; r0 < r_end; r0 += 32) {
        float4 c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = *reinterpret_cast<const float4*>(M + (size_t)(r0 + k) * n + j);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a.x += c[k].x; a.y += c[k].y; a.z += c[k].z; a.w += c[k].w; }
    }
    out[(size_t)(blockIdx.y * 4 + wv) * n + j] = a.x + a.y + a.z + a.w;
}

// persistent streaming: G workgroups of 1024 threads, wave <-> rows strided, next row requested while reducing
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_stream(const float* __restrict__ M, int n, int rows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nw = gridDim.x * WAVES;
    int r = wv * gridDim.x + blockIdx.x;
    float4 c[16];
    if (r < rows) {
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = *reinterpret_cast<const float4*>(M + (size_t)r * n + lane * 4 + 256 * k);
    }
    while (r < rows) {
        const int rn = r + nw;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc += c[k].x + c[k].y + c[k].z + c[k].w;
            if (rn < rows) c[k] = *reinterpret_cast<const float4*>(M + (size_t)rn * n + lane * 4 + 256 * k);
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) out[r] = acc;
        r = rn;
    }
}

template <typename F>
static float time_it(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    const int n = 4096;
    float *M, *out; double* v;
    CK(hipMalloc(&M, (size_t)n * n * 4)); CK(hipMalloc(&out, (size_t)n * 256 * 4)); CK(hipMalloc(&v, n * 8));
    CK(hipMemset(M, 0, (size_t)n * n * 4)); CK(hipMemset(v, 0, n * 8));
    CK(hipFuncSetAttribute((const void*)k_oneshot<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_oneshot<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    CK(hipFuncSetAttribute((const void*)k_oneshot<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const double mb = (double)n * n * 4 / 1e6;
    auto rep = [&](const char* name, float us) { printf("%-34s %7.2f us  %6.0f GB/s\n", name, us, mb / us * 1e3); };
    // back-to-back launches of the same kernel overlap their ramps a little; alternate two kernels like the solver does
    rep("oneshot 4 waves, no v", time_it([&] { hipLaunchKernelGGL((k_oneshot<4, 0>), dim3(n / 4), dim3(256), 0, 0, M, n, v, out); }, 100));
    rep("oneshot 8 waves, no v", time_it([&] { hipLaunchKernelGGL((k_oneshot<8, 0>), dim3(n / 8), dim3(512), 0, 0, M, n, v, out); }, 100));
    rep("oneshot 16 waves, no v", time_it([&] { hipLaunchKernelGGL((k_oneshot<16, 0>), dim3(n / 16), dim3(1024), 0, 0, M, n, v, out); }, 100));
    rep("oneshot 4 waves, v staged", time_it([&] { hipLaunchKernelGGL((k_oneshot<4, 1>), dim3(n / 4), dim3(256), n * 8, 0, M, n, v, out); }, 100));
    rep("oneshot 8 waves, v staged", time_it([&] { hipLaunchKernelGGL((k_oneshot<8, 1>), dim3(n / 8), dim3(512), n * 8, 0, M, n, v, out); }, 100));
    rep("oneshot 16 waves, v staged", time_it([&] { hipLaunchKernelGGL((k_oneshot<16, 1>), dim3(n / 16), dim3(1024), n * 8, 0, M, n, v, out); }, 100));
    rep("strip 16x64 (col-pass shape)", time_it([&] { hipLaunchKernelGGL(k_strip, dim3(n / 256, 64), dim3(256), 0, 0, M, n, n / 64, out); }, 100));
    rep("strip 16x32", time_it([&] { hipLaunchKernelGGL(k_strip, dim3(n / 256, 32), dim3(256), 0, 0, M, n, n / 32, out); }, 100));
    rep("stream 256 x 16 waves", time_it([&] { hipLaunchKernelGGL((k_stream<16>), dim3(256), dim3(1024), 0, 0, M, n, n, out); }, 100));
    rep("stream 256 x 8 waves", time_it([&] { hipLaunchKernelGGL((k_stream<8>), dim3(256), dim3(512), 0, 0, M, n, n, out); }, 100));
    rep("stream 512 x 4 waves", time_it([&] { hipLaunchKernelGGL((k_stream<4>), dim3(512), dim3(256), 0, 0, M, n, n, out); }, 100));
    rep("stream 768 x 4 waves", time_it([&] { hipLaunchKernelGGL((k_stream<4>), dim3(768), dim3(256), 0, 0, M, n, n, out); }, 100));
    rep("alternating oneshot4+v / strip", time_it([&] {
        hipLaunchKernelGGL((k_oneshot<4, 1>), dim3(n / 4), dim3(256), n * 8, 0, M, n, v, out);
        hipLaunchKernelGGL(k_strip, dim3(n / 256, 64), dim3(256), 0, 0, M, n, n / 64, out); }, 100) / 2);
    return 0;
}
