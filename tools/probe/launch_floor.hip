// Measurement aid (not part of the library): cost of one dependent kernel boundary inside a hipGraph,
// by grid shape, block size, dynamic LDS and whether the kernel reads a word that the previous one wrote.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/launch_floor.hip -o tools/probe/launch_floor && tools/probe/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(int* p) { }
__global__ void k_read(int* p) { if (p[0] == 12345) p[1] = 1; }                                  // one dependent scalar read
__global__ void k_chain(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + 1; }    // read-modify-write by one thread
__global__ void k_lds(int* p) { extern __shared__ int s[]; if (p[0] == 12345) { s[threadIdx.x] = 1; p[1] = s[0]; } }
__global__ void k_ticket(int* p) {            // every workgroup draws a device-scope ticket, the last one writes
    __shared__ int last;
    if (threadIdx.x == 0) { int t = __hip_atomic_fetch_add(&p[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = (t == (int)gridDim.x - 1); if (last) __hip_atomic_store(&p[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (last && threadIdx.x == 0) p[0] = p[0] + 1;
}

template <typename F>
static float run(hipStream_t s, int reps, int len, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < len; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (reps * len);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    int* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int reps = 20, len = 200;
    struct Cfg { int grid, block; } cfgs[] = {{1, 64}, {1, 1024}, {256, 256}, {256, 512}, {256, 1024}, {512, 1024}, {1024, 256}, {64, 1024}};
    printf("us per dependent launch inside a hipGraph (%d launches per graph)\n", len);
    for (auto c : cfgs) {
        float e = run(s, reps, len, [&] { hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, p); });
        float r = run(s, reps, len, [&] { hipLaunchKernelGGL(k_read, dim3(c.grid), dim3(c.block), 0, s, p); });
        float ch = run(s, reps, len, [&] { hipLaunchKernelGGL(k_chain, dim3(c.grid), dim3(c.block), 0, s, p); });
        float l48 = run(s, reps, len, [&] { hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), 48 * 1024, s, p); });
        float l140 = run(s, reps, len, [&] { hipLaunchKernelGGL(k_lds, dim3(c.grid), dim3(c.block), 140 * 1024, s, p); });
        float tk = run(s, reps, len, [&] { hipLaunchKernelGGL(k_ticket, dim3(c.grid), dim3(c.block), 0, s, p); });
        printf("grid %4d x %4d thr: empty %.2f  read %.2f  rmw-chain %.2f  lds48K %.2f  lds140K %.2f  ticket+last-writer %.2f\n",
               c.grid, c.block, e, r, ch, l48, l140, tk);
    }
    // plain launches (no graph), same stream
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, s, p);
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_read, dim3(256), dim3(1024), 0, s, p);
        hipEventRecord(b, s); hipStreamSynchronize(s);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("plain launches 256 x 1024 read: %.2f us each\n", ms * 1e3f / 2000);
    }
    return 0;
}
