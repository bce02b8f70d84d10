// Measurement aid (not part of the library; starting point of the next round): where does the direct-to-LDS fp32-MFMA
// engine (csrc/gemm_glds.h) spend the 30 % of the matrix pipe it leaves idle at 110 TFLOP/s?
//
// Round 4 priced the pieces by leaving them out (no loads: +13 %; no LDS stores: +8.5 %; fragment reads, barrier: free)
// and found both engines on the same plateau.  What was never measured: (a) what the memory path ALONE delivers for the
// engine's access pattern (DMA + waits, no MFMA), (b) where a wave actually waits — for its DMAs (s_waitcnt vmcnt) or
// for the slowest wave of its workgroup (s_barrier), (c) whether a deeper pipeline (3 / 4 stages in LDS, one workgroup
// per CU) hides what two stages do not.  This program answers the three in one run (~10 s):
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probe/glds_probe.hip -o tools/probe/glds_probe
//   tools/probe/glds_probe [M N K]          (default 4096 4096 3136: the asymptotic shape of profiles/r4_gemm_probes.txt)
//   tools/probe/glds_probe ab [rounds [M N K]]   interleaved A/B of every uninstrumented loop, with each launch's core clock
//   tools/probe/glds_probe padded           the header's second-generation loop (gl_run_padded) on padded operands, checked
//
// Per variant: time, TFLOP/s, and (instrumented builds) the share of a wave's lifetime spent in the vmcnt wait and in the
// barrier, from s_memtime around them.  MODE 0 results are checked against a float64 host product on sampled entries.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <functional>
#include "../../conditional-flow-matching_amd/csrc/gemm_glds.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long pr_clk() { return __builtin_readcyclecounter(); }      // s_memtime
__device__ __forceinline__ unsigned long long pr_rt() { return wall_clock64(); }                     // s_memrealtime: 100 MHz

// one lane of the launch stamps both clocks at the start and at the end of its K loop: the effective core clock of THIS launch
// (s_memtime ticks per 10 ns of s_memrealtime) without instrumenting the loop
__device__ __forceinline__ void pr_stamp(unsigned long long* clk, int which) {
    if (clk && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { clk[2 * which] = pr_clk(); clk[2 * which + 1] = pr_rt(); }
}

template <int NST> __device__ __forceinline__ void pr_wait_vm(bool later_stages_in_flight) {
    // DMAs complete in order: with NST - 2 younger stages (8 DMAs per wave each) behind the one needed, vmcnt may stay
    // at 8 (NST - 2); in the tail (no younger stage issued) everything outstanding is needed
    if (NST == 2 || !later_stages_in_flight) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (NST == 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
}

// The DMAs as inline assembly in the SGPR-base + 32-bit-lane-offset form.  Two reasons (found with this probe, round 4):
//   * the compiler treats __builtin_amdgcn_global_load_lds as a write to LDS that any later ds_read may alias and puts an
//     s_waitcnt vmcnt(0) in front of the first fragment read of EVERY K step — i.e. behind the DMAs of the NEXT stage that
//     were issued a moment earlier: the engine of gemm_glds.h never overlaps a stage's loads with its own MFMAs, only
//     with those of the other workgroup on the CU.  Assembly DMAs are invisible to that pass; the explicit
//     s_waitcnt vmcnt(n) + s_barrier at the top of the K step is the ordering.
//   * per DMA the builtin path spends ~12 instructions (64-bit add, zero-row selects, v_readfirstlane + s_mov m0 of an LDS
//     address that is wave uniform anyway); here a stage advances by ONE scalar add per operand.
// Requires M, N multiples of 128 and K a multiple of 32 (no zero rows / k tail): a probe, not the product engine.
struct FastDma {
    unsigned offa[4], offb[4];          // byte offset of this lane's 16 bytes from the operand base, at k0 = 0
    unsigned la[4], lb[4];              // LDS byte address of the wave's piece q inside stage 0 (wave uniform)
    __device__ __forceinline__ void bind(int lda, int row0, int ldb, int col0, float* lds) {
        const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 8 * (wv + 4 * q) + (lane >> 3);
            const int kq = (lane & 7) ^ ((r >> 1) & 7);
            offa[q] = (unsigned)(((size_t)(row0 + r) * lda + 4 * kq) * 4);
            offb[q] = (unsigned)(((size_t)(col0 + r) * ldb + 4 * kq) * 4);
            la[q] = base + (unsigned)((wv + 4 * q) * 256 * 4);
            lb[q] = la[q] + GL_BM * GL_BK * 4;
        }
    }
    static __device__ __forceinline__ void dma(unsigned off, const float* base, unsigned ldsaddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(ldsaddr) : "memory");
    }
    __device__ __forceinline__ void issue(const float* Ak, const float* Bk, int st) const {
        const unsigned so = (unsigned)st * (GL_STAGE_FLOATS * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { dma(offa[q], Ak, la[q] + so); dma(offb[q], Bk, lb[q] + so); }
    }
};

// MODE 0: the product loop     1: DMA + waits only (no MFMA, no fragment reads)     2: no DMA inside the loop
template <int MODE, int NST, int WPC, bool STATS, bool FAST = false, bool FAIR = false>
__global__ __launch_bounds__(256, WPC) void glds_probe(const float* __restrict__ A, const float* __restrict__ B, int M, int N,
                                                      int K, const float* __restrict__ zeros, float* __restrict__ C,
                                                      unsigned long long* __restrict__ stats, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {   // the 8 x 8 super-tile order of cost_gemm
        const int G = 8;
        const int per_band = G * tiles_n;
        const int band = lid / per_band, r = lid - band * per_band;
        const int rows_in_band = min(G, tiles_m - band * G);
        const int fgt = rows_in_band * G;
        const int gcol = r / fgt;
        const int rr = r - gcol * fgt;
        const int cols_in_group = min(G, tiles_n - gcol * G);
        tm = band * G + rr / cols_in_group;
        tn = gcol * G + rr % cols_in_group;
    }
    const int row0 = tm * GL_BM, col0 = tn * GL_BN;
    GldsCore g;
    g.zero();
    FastDma fd;
    if (FAST) fd.bind(K, row0, K, col0, lds); else g.bind(A, K, row0, M, B, K, col0, N);
    unsigned long long t_vm = 0, t_bar = 0, t_start = 0, r_start = 0;
    unsigned long long* const clk = STATS ? nullptr : stats + (size_t)gridDim.x * 32;      // (behind the per-wave records)
    if (STATS) { t_start = pr_clk(); r_start = pr_rt(); }
    pr_stamp(clk, 0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s * GL_BK < K) { if (FAST) fd.issue(A + s * GL_BK, B + s * GL_BK, s); else g.issue(lds, s, s * GL_BK, K, zeros); }
    int st = 0, nx = NST - 1;                       // stage being consumed, stage buffer the next DMA goes to
    for (int k0 = 0; k0 < K; k0 += GL_BK) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if (STATS) t0 = pr_clk();
        pr_wait_vm<NST>(k0 + (NST - 2) * GL_BK < K && NST > 2);
        if (STATS) t1 = pr_clk();
        asm volatile("s_barrier" ::: "memory");
        if (STATS) { t2 = pr_clk(); t_vm += t1 - t0; t_bar += t2 - t1; }
        if (FAIR) {         // the two workgroups of a CU take turns at the higher issue priority (see the timeline of the plain loop)
            if (((blockIdx.x >> 8) ^ (unsigned)(k0 >> 5)) & 1u) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0");
        }
        if (MODE != 2 && k0 + (NST - 1) * GL_BK < K) {
            const int kn = k0 + (NST - 1) * GL_BK;
            if (FAST) fd.issue(A + kn, B + kn, nx); else g.issue(lds, nx, kn, K, zeros);
        }
        if (MODE != 1) g.compute(lds, st);
        st = st + 1 == NST ? 0 : st + 1;
        nx = nx + 1 == NST ? 0 : nx + 1;
    }
    pr_stamp(clk, 1);
    if (STATS) {
        const unsigned long long t_all = pr_clk() - t_start, r_all = pr_rt() - r_start;
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* s = stats + 8 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6));
            s[0] = t_all; s[1] = t_vm; s[2] = t_bar; s[3] = r_all; s[4] = r_start; s[5] = r_start + r_all;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + GldsCore::row_of(m, r);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gc = col0 + GldsCore::col_of(u);
                if (gr < M && gc < N) C[(size_t)gr * N + gc] = g.acc[m][u][r];
            }
        }
}

// PIPELINED K-step boundary (next round's candidate; compiled and ISA-checked at the end of round 4, never run):
// the product loop drains the matrix pipe at every K step — wait, barrier, DMA issue, then eight fragment reads nothing
// covers (a lone wave keeps the pipe 74 % busy even with assembly DMAs).  Here the LAST k-quad of stage s is computed
// AFTER the barrier of step s + 1: its fragments are already in registers (the barrier only protects the LDS buffer), and
// its eight MFMAs (512 cycles) cover the first fragment reads of stage s + 1 and — ILV — the issue of the eight DMAs of
// stage s + 2, placed one behind each MFMA.  Same k order per output as every other loop here.
struct PipeFrag { float4 a0, a1, b0, b1; };
template <int WPC, bool ILV, bool FAIR>
__global__ __launch_bounds__(256, WPC) void glds_pipe(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int K,
                                                     float* __restrict__ C, int tiles_m, int tiles_n, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {
        const int G = 8;
        const int per_band = G * tiles_n;
        const int band = lid / per_band, r = lid - band * per_band;
        const int rows_in_band = min(G, tiles_m - band * G);
        const int fgt = rows_in_band * G;
        const int gcol = r / fgt;
        const int rr = r - gcol * fgt;
        const int cols_in_group = min(G, tiles_n - gcol * G);
        tm = band * G + rr / cols_in_group;
        tn = gcol * G + rr % cols_in_group;
    }
    const int row0 = tm * GL_BM, col0 = tn * GL_BN;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wm = wv >> 1, wn = wv & 1;
    const bool hi = lane >= 32;
    const int fr = lane & 31, sw = (fr >> 1) & 7;
    GldsCore g;
    g.zero();
    FastDma fd;
    fd.bind(K, row0, K, col0, lds);
    const int nsteps = K / GL_BK;
    auto rd = [&](int st, int q) {
        const float* As = lds + st * GL_STAGE_FLOATS + (wm * 64 + fr) * GL_BK;
        const float* Bs = lds + st * GL_STAGE_FLOATS + GL_BM * GL_BK + (wn * 64 + fr) * GL_BK;
        const int so = 4 * (q ^ sw);
        PipeFrag f;
        f.a0 = *reinterpret_cast<const float4*>(As + so); f.a1 = *reinterpret_cast<const float4*>(As + 32 * GL_BK + so);
        f.b0 = *reinterpret_cast<const float4*>(Bs + so); f.b1 = *reinterpret_cast<const float4*>(Bs + 32 * GL_BK + so);
        return f;
    };
    // the 8 MFMAs of a k-quad; dma(i): hook run behind MFMA i (the interleaved DMA issue)
    auto mm = [&](const PipeFrag& t, auto dma) {
        const float a00 = hi ? t.a0.y : t.a0.x, a01 = hi ? t.a1.y : t.a1.x, b00 = hi ? t.b0.y : t.b0.x, b01 = hi ? t.b1.y : t.b1.x;
        const float a10 = hi ? t.a0.w : t.a0.z, a11 = hi ? t.a1.w : t.a1.z, b10 = hi ? t.b0.w : t.b0.z, b11 = hi ? t.b1.w : t.b1.z;
        g.acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b00, g.acc[0][0], 0, 0, 0); dma(0);
        g.acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b01, g.acc[0][1], 0, 0, 0); dma(1);
        g.acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b00, g.acc[1][0], 0, 0, 0); dma(2);
        g.acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01, g.acc[1][1], 0, 0, 0); dma(3);
        g.acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b10, g.acc[0][0], 0, 0, 0); dma(4);
        g.acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b11, g.acc[0][1], 0, 0, 0); dma(5);
        g.acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b10, g.acc[1][0], 0, 0, 0); dma(6);
        g.acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b11, g.acc[1][1], 0, 0, 0); dma(7);
    };
    auto none = [](int) {};
    pr_stamp(clk, 0);
    fd.issue(A, B, 0);
    if (nsteps > 1) { fd.issue(A + GL_BK, B + GL_BK, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    PipeFrag t = rd(0, 0);
    // ONE body in the loop, the last stage peeled behind it, no branch around an MFMA: accumulators that reach a K step
    // over two paths are copied (round 4; the first form of this loop moved 64 registers per step).
    // The DMAs of the loop's last trip have no stage left to fetch: they fetch the last stage once more into the buffer
    // nobody reads again (drained by the vmcnt(0) behind the loop).
    for (int s = 0; s + 1 < nsteps; ++s) {
        const int st = s & 1;
        if (FAIR) { if (((blockIdx.x >> 8) ^ (unsigned)s) & 1u) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0"); }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const PipeFrag n = rd(st, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(t, none);
            t = n;
        }
        // stage s + 1 has landed (issued a whole step ago); this wave's reads of stage s are complete (lgkmcnt); behind
        // the barrier nobody reads stage s any more: its buffer takes stage s + 2
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int sn = min(s + 2, nsteps - 1);
        const float* An = A + (size_t)sn * GL_BK; const float* Bn = B + (size_t)sn * GL_BK;
        if (!ILV) fd.issue(An, Bn, st);
        const PipeFrag n = rd(st ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ILV) {
            const unsigned so = (unsigned)st * (GL_STAGE_FLOATS * 4);
            mm(t, [&](int i) {
                __builtin_amdgcn_sched_barrier(0);
                if (i & 1) FastDma::dma(fd.offb[i >> 1], Bn, fd.lb[i >> 1] + so); else FastDma::dma(fd.offa[i >> 1], An, fd.la[i >> 1] + so);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else mm(t, none);
        t = n;
    }
    {
        const int st = (nsteps - 1) & 1;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const PipeFrag n = rd(st, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(t, none);
            t = n;
        }
        mm(t, none);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may outlive the workgroup's LDS allocation
    pr_stamp(clk, 1);
    if (FAIR) asm volatile("s_setprio 0");
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + GldsCore::row_of(m, r);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gc = col0 + GldsCore::col_of(u);
                if (gr < M && gc < N) C[(size_t)gr * N + gc] = g.acc[m][u][r];
            }
        }
}

// the matrix pipe alone (registers only), operands constant (what tools/probe/mfma_peak measures) or changing from one
// MFMA to the next (16 different register pairs per lane): does the sustained rate depend on the data?
template <bool VARY>
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters, const float* __restrict__ src, unsigned long long* stats) {
    gl_f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = src[(threadIdx.x * 16 + 2 * u) & 4095]; y[u] = src[(threadIdx.x * 16 + 2 * u + 1) & 4095]; }
    const unsigned long long t0 = pr_clk(), r0 = pr_rt();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float p = VARY ? x[u] : x[0], q = VARY ? y[u] : y[0];
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, q, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(q, p, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(p, p, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(q, q, a3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = pr_clk() - t0, r1 = pr_rt() - r0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = t1; stats[1] = r1; }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) out[0] = s;
}

// the header's second-generation loop (gl_run_padded) on padded operands: any M, N, K
template <bool FAIR, bool PIPE>
__global__ __launch_bounds__(256, 2) void glds_padded(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int Kp,
                                                      float* __restrict__ C, int tiles_m, int tiles_n, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {
        const int G = 8;
        const int per_band = G * tiles_n;
        const int band = lid / per_band, r = lid - band * per_band;
        const int rows_in_band = min(G, tiles_m - band * G);
        const int fgt = rows_in_band * G;
        const int gcol = r / fgt;
        const int rr = r - gcol * fgt;
        const int cols_in_group = min(G, tiles_n - gcol * G);
        tm = band * G + rr / cols_in_group;
        tn = gcol * G + rr % cols_in_group;
    }
    const int row0 = tm * GL_BM, col0 = tn * GL_BN;
    GldsCore g;
    g.zero();
    pr_stamp(clk, 0);
    if (PIPE) gl_run_padded_pipe(g, lds, A, Kp, row0, M, B, Kp, col0, N, Kp, FAIR ? (int)((blockIdx.x >> 8) & 1u) : -1);
    else gl_run_padded(g, lds, A, Kp, row0, M, B, Kp, col0, N, Kp, FAIR ? (int)((blockIdx.x >> 8) & 1u) : -1);
    pr_stamp(clk, 1);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + GldsCore::row_of(m, r);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int gc = col0 + GldsCore::col_of(u);
                if (gr < M && gc < N) C[(size_t)gr * N + gc] = g.acc[m][u][r];
            }
        }
}

// pads, runs and checks one shape on glds_padded
static int run_padded(int M, int N, int K) {
    const int Kp = (K + GL_BK - 1) / GL_BK * GL_BK, tm = (M + GL_BM - 1) / GL_BM, tn = (N + GL_BN - 1) / GL_BN;
    std::vector<float> hA((size_t)(M + 1) * Kp, 0.f), hB((size_t)(N + 1) * Kp, 0.f);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int i = 0; i < M; ++i) for (int k = 0; k < K; ++k) hA[(size_t)i * Kp + k] = rnd();
    for (int j = 0; j < N; ++j) for (int k = 0; k < K; ++k) hB[(size_t)j * Kp + k] = rnd();
    float *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    int rc = 0;
    CK(hipFuncSetAttribute((const void*)glds_padded<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)glds_padded<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)glds_padded<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)glds_padded<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS_BYTES));
    for (int var = 0; var < 4; ++var) {
        const int fair = var & 1, pipe = var >> 1;
        auto launch = [&]() {
            if (pipe) { if (fair) hipLaunchKernelGGL((glds_padded<true, true>), dim3(tm * tn), dim3(256), GL_LDS_BYTES, 0, A, B, M, N, Kp, C, tm, tn, (unsigned long long*)nullptr);
                        else hipLaunchKernelGGL((glds_padded<false, true>), dim3(tm * tn), dim3(256), GL_LDS_BYTES, 0, A, B, M, N, Kp, C, tm, tn, (unsigned long long*)nullptr); }
            else { if (fair) hipLaunchKernelGGL((glds_padded<true, false>), dim3(tm * tn), dim3(256), GL_LDS_BYTES, 0, A, B, M, N, Kp, C, tm, tn, (unsigned long long*)nullptr);
                   else hipLaunchKernelGGL((glds_padded<false, false>), dim3(tm * tn), dim3(256), GL_LDS_BYTES, 0, A, B, M, N, Kp, C, tm, tn, (unsigned long long*)nullptr); }
        };
        CK(hipMemset(C, 0xff, (size_t)M * N * 4));
        launch(); launch(); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 10;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 4096; ++t) {
            // the last rows / columns (edge tiles) are sampled as densely as the interior
            const int i = t & 1 ? M - 1 - (int)((t * 2654435761u) % 200u) % M : (int)((t * 2654435761u) % (unsigned)M);
            const int j = t & 2 ? N - 1 - (int)((t * 40503u + 17u) % 200u) % N : (int)((t * 40503u + 17u) % (unsigned)N);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * Kp + k] * (double)hB[(size_t)j * Kp + k];
            const double err = fabs(ref - hC[(size_t)i * N + j]) / (fabs(ref) + 1.0);
            if (!(err <= worst)) worst = err;
        }
        printf("%s %5d x %5d x %5d (K padded to %d)%s: %8.1f us  %6.1f TFLOP/s   worst relative error of 4096 samples %.2e %s\n",
               pipe ? "gl_run_padded_pipe" : "gl_run_padded     ", M, N, K, Kp, fair ? ", alternating s_setprio" : "                        ", ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, worst,
               worst < 1e-4 ? "ok" : "WRONG");
        if (!(worst < 1e-4)) rc = 2;
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    return rc;
}

struct Ctx { float *A, *B, *C, *zeros; unsigned long long* stats; int M, N, K, tm, tn; std::vector<float> hA, hB; };

template <int MODE, int NST, int WPC, bool STATS, bool FAST = false, bool FAIR = false>
static int run(Ctx& c, const char* label, bool check) {
    // one workgroup per CU is enforced through the LDS request (96 KiB of 160): registers alone would let two in
    size_t ldsb = (size_t)NST * GL_STAGE_FLOATS * 4;
    if (WPC == 1 && ldsb < 96 * 1024) ldsb = 96 * 1024;
    auto kern = glds_probe<MODE, NST, WPC, STATS, FAST, FAIR>;
    if (FAST && (c.M % GL_BM || c.N % GL_BN || c.K % GL_BK)) { printf("%s: skipped (M, N, K not multiples of the tile)\n", label); return 0; }
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int grid = c.tm * c.tn;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, c.A, c.B, c.M, c.N, c.K, c.zeros, c.C, c.stats, c.tm, c.tn);
    CK(hipDeviceSynchronize());
    const int reps = 6;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, c.A, c.B, c.M, c.N, c.K, c.zeros, c.C, c.stats, c.tm, c.tn);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double tf = 2.0 * c.M * c.N * c.K / (ms * 1e-3) / 1e12;
    const double gb = (double)grid * (GL_BM + GL_BN) * c.K * 4.0 / 1e9;          // operand bytes the tiles request
    printf("%-46s %8.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)  operand requests %5.2f TB/s", label, ms * 1e3, tf, tf / 1.573, gb / ms);
    if (STATS) {
        std::vector<unsigned long long> h((size_t)grid * 32);
        CK(hipMemcpy(h.data(), c.stats, h.size() * 8, hipMemcpyDeviceToHost));
        double all = 0, vm = 0, bar = 0, rt = 0;
        unsigned long long first = ~0ull, last = 0;
        std::vector<double> life, start, end;
        for (size_t w = 0; w < (size_t)grid * 4; ++w) {
            const unsigned long long* q = &h[8 * w];
            all += q[0]; vm += q[1]; bar += q[2]; rt += q[3];
            if (q[4] < first) first = q[4];
            if (q[5] > last) last = q[5];
        }
        for (size_t w = 0; w < (size_t)grid * 4; w += 4) {      // wave 0 of every workgroup
            const unsigned long long* q = &h[8 * w];
            life.push_back(q[3] * 0.01); start.push_back((q[4] - first) * 0.01); end.push_back((q[5] - first) * 0.01);
        }
        std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end()); std::sort(end.begin(), end.end());
        const double span = (last - first) * 0.01;
        printf("   wave lifetime %7.0f s_memtime ticks = %6.1f us (s_memtime runs at %5.3f GHz): vmcnt wait %4.1f %%, barrier %4.1f %%\n",
               all / (grid * 4.0), rt / (grid * 4.0) * 0.01, all / (rt * 10.0), 100 * vm / all, 100 * bar / all);
        auto pc = [](const std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
        printf("      K loops span %7.1f us of the %7.1f us launch; lifetimes min / 10 %% / median / 90 %% / max: %.0f / %.0f / %.0f / %.0f / %.0f us;"
               " mean K loops alive %.0f of %d slots\n", span, ms * 1e3, life.front(), pc(life, .1), pc(life, .5), pc(life, .9), life.back(),
               rt * 0.01 / 4.0 / span, WPC * 256);
        printf("      K-loop starts at 25 / 50 / 51 / 75 / 100 %% of the workgroups: %.0f / %.0f / %.0f / %.0f / %.0f us;  ends at 25 / 50 / 75 / 100 %%: %.0f / %.0f / %.0f / %.0f us",
               pc(start, .25), pc(start, .5), pc(start, .51), pc(start, .75), start.back(), pc(end, .25), pc(end, .5), pc(end, .75), end.back());
    }
    printf("\n");
    if (check) {
        std::vector<float> hC((size_t)c.M * c.N);
        CK(hipMemcpy(hC.data(), c.C, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int s = 0; s < 2048; ++s) {
            const int i = (int)((s * 2654435761u) % (unsigned)c.M), j = (int)((s * 40503u + 17u) % (unsigned)c.N);
            double ref = 0;
            for (int k = 0; k < c.K; ++k) ref += (double)c.hA[(size_t)i * c.K + k] * (double)c.hB[(size_t)j * c.K + k];
            const double err = fabs(ref - hC[(size_t)i * c.N + j]) / (fabs(ref) + 1.0);
            if (err > worst) worst = err;
        }
        printf("    check (2048 sampled entries vs float64): worst relative error %.2e %s\n", worst, worst < 1e-4 ? "ok" : "WRONG");
        if (!(worst < 1e-4)) return 2;
    }
    return 0;
}

template <int WPC, bool ILV, bool FAIR>
static int run_pipe(Ctx& c, const char* label) {
    if (c.M % GL_BM || c.N % GL_BN || c.K % GL_BK) { printf("%s: skipped (M, N, K not multiples of the tile)\n", label); return 0; }
    size_t ldsb = GL_LDS_BYTES;
    if (WPC == 1) ldsb = 96 * 1024;
    auto kern = glds_pipe<WPC, ILV, FAIR>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int grid = c.tm * c.tn;
    auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsb, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, (unsigned long long*)nullptr); };
    CK(hipMemset(c.C, 0xff, (size_t)c.M * c.N * 4));
    launch(); launch(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 6;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double tf = 2.0 * c.M * c.N * c.K / (ms * 1e-3) / 1e12;
    std::vector<float> hC((size_t)c.M * c.N);
    CK(hipMemcpy(hC.data(), c.C, hC.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 2048; ++t) {
        const int i = (int)((t * 2654435761u) % (unsigned)c.M), j = (int)((t * 40503u + 17u) % (unsigned)c.N);
        double ref = 0;
        for (int k = 0; k < c.K; ++k) ref += (double)c.hA[(size_t)i * c.K + k] * (double)c.hB[(size_t)j * c.K + k];
        const double err = fabs(ref - hC[(size_t)i * c.N + j]) / (fabs(ref) + 1.0);
        if (!(err <= worst)) worst = err;
    }
    printf("%-58s %8.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)   check %.2e %s\n", label, ms * 1e3, tf, tf / 1.573, worst, worst < 1e-4 ? "ok" : "WRONG");
    return worst < 1e-4 ? 0 : 2;
}

// Interleaved A/B of the uninstrumented loops: every variant is timed `rounds` times, round-robin, so that the chip's power
// state (its clock moves by 10 % with what ran before: run F of profiles/r4_glds_probe.txt) hits all of them alike.
// Per variant: median / min / max of the launch time, TFLOP/s of the median, median core clock of the stamped launches.
struct AbVar { const char* label; std::function<void()> launch; std::vector<double> us, ghz; };
static int run_ab(Ctx& c, int rounds) {
    if (c.M % GL_BM || c.N % GL_BN || c.K % GL_BK) { printf("ab: M, N, K must be multiples of the tile\n"); return 1; }
    const int grid = c.tm * c.tn;
    unsigned long long* clk = c.stats + (size_t)grid * 32;
    const size_t l2 = GL_LDS_BYTES;
    std::vector<AbVar> v;
#define AB_ATTR(K_) CK(hipFuncSetAttribute((const void*)(K_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2))
    {
        auto k0 = glds_probe<0, 2, 2, false>;             AB_ATTR(k0);
        auto k1 = glds_probe<0, 2, 2, false, true>;       AB_ATTR(k1);
        auto k2 = glds_probe<0, 2, 2, false, true, true>; AB_ATTR(k2);
        auto k3 = glds_pipe<2, false, false>;             AB_ATTR(k3);
        auto k4 = glds_pipe<2, false, true>;              AB_ATTR(k4);
        auto k5 = glds_pipe<2, true, false>;              AB_ATTR(k5);
        auto k6 = glds_padded<false, false>;              AB_ATTR(k6);
        auto k7 = glds_padded<true, false>;               AB_ATTR(k7);
        auto k8 = glds_padded<false, true>;               AB_ATTR(k8);
        auto k9 = glds_padded<true, true>;                AB_ATTR(k9);
        v.push_back({"builtin DMAs (the engine as the product has it)", [=, &c]() { hipLaunchKernelGGL(k0, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.zeros, c.C, c.stats, c.tm, c.tn); }});
        v.push_back({"assembly DMAs", [=, &c]() { hipLaunchKernelGGL(k1, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.zeros, c.C, c.stats, c.tm, c.tn); }});
        v.push_back({"assembly DMAs + alternating s_setprio", [=, &c]() { hipLaunchKernelGGL(k2, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.zeros, c.C, c.stats, c.tm, c.tn); }});
        v.push_back({"pipelined boundary (probe form)", [=, &c]() { hipLaunchKernelGGL(k3, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"pipelined boundary + alternating s_setprio", [=, &c]() { hipLaunchKernelGGL(k4, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"pipelined boundary, DMAs between the MFMAs", [=, &c]() { hipLaunchKernelGGL(k5, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"header: gl_run_padded", [=, &c]() { hipLaunchKernelGGL(k6, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"header: gl_run_padded + alternating s_setprio", [=, &c]() { hipLaunchKernelGGL(k7, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"header: gl_run_padded_pipe", [=, &c]() { hipLaunchKernelGGL(k8, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
        v.push_back({"header: gl_run_padded_pipe + alternating s_setprio", [=, &c]() { hipLaunchKernelGGL(k9, dim3(grid), dim3(256), l2, 0, c.A, c.B, c.M, c.N, c.K, c.C, c.tm, c.tn, clk); }});
    }
#undef AB_ATTR
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& x : v) { x.launch(); x.launch(); }
    CK(hipDeviceSynchronize());
    const int reps = 4;
    for (int r = 0; r < rounds; ++r)
        for (auto& x : v) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) x.launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[4]; CK(hipMemcpy(h, clk, 32, hipMemcpyDeviceToHost));
            x.us.push_back(ms / reps * 1e3);
            x.ghz.push_back((double)(h[2] - h[0]) / ((double)(h[3] - h[1]) * 10.0));
        }
    printf("interleaved A/B, %d rounds of %d launches per variant (%d x %d x %d):\n", rounds, reps, c.M, c.N, c.K);
    for (auto& x : v) {
        std::sort(x.us.begin(), x.us.end()); std::sort(x.ghz.begin(), x.ghz.end());
        const double med = x.us[x.us.size() / 2];
        printf("  %-52s median %7.1f us (%7.1f .. %7.1f)  %6.1f TFLOP/s   clock %5.3f GHz (%5.3f .. %5.3f)\n", x.label, med, x.us.front(), x.us.back(),
               2.0 * c.M * c.N * c.K / (med * 1e-6) / 1e12, x.ghz[x.ghz.size() / 2], x.ghz.front(), x.ghz.back());
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "padded")) {      // glds_probe padded: the header's gl_run_padded on three shapes
        int rc = run_padded(4096, 4096, 784);          // C3
        rc |= run_padded(4000, 4090, 777);             // edge tiles in both directions, a k tail inside the padding
        rc |= run_padded(4096, 4096, 3136);
        rc |= run_padded(130, 300, 36);
        rc |= run_padded(128, 128, 32);              // one K step: the pipelined loop's degenerate case
        rc |= run_padded(256, 384, 64);              // two K steps
        return rc;
    }
    Ctx c;
    const bool ab = argc > 1 && !strcmp(argv[1], "ab");              // glds_probe ab [rounds [M N K]]
    const int ab_rounds = (ab && argc > 2) ? atoi(argv[2]) : 7;
    if (ab) { const int shift = argc > 2 ? 2 : 1; argc -= shift; argv += shift; }      // what follows: [M N K]
    c.M = argc > 3 ? atoi(argv[1]) : 4096; c.N = argc > 3 ? atoi(argv[2]) : 4096; c.K = argc > 3 ? atoi(argv[3]) : 3136;
    if (c.K % 4) { printf("K must be a multiple of 4 (16-byte DMA pieces)\n"); return 1; }
    c.tm = (c.M + GL_BM - 1) / GL_BM; c.tn = (c.N + GL_BN - 1) / GL_BN;
    c.hA.assign((size_t)(c.M + 1) * c.K, 0.f); c.hB.assign((size_t)(c.N + 1) * c.K, 0.f);      // (one more row of zeros: the padded layout)
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (size_t i = 0; i < (size_t)c.M * c.K; ++i) c.hA[i] = rnd();
    for (size_t i = 0; i < (size_t)c.N * c.K; ++i) c.hB[i] = rnd();
    CK(hipMalloc(&c.A, c.hA.size() * 4)); CK(hipMalloc(&c.B, c.hB.size() * 4)); CK(hipMalloc(&c.C, (size_t)c.M * c.N * 4));
    CK(hipMalloc(&c.zeros, 256)); CK(hipMemset(c.zeros, 0, 256));
    CK(hipMalloc(&c.stats, ((size_t)c.tm * c.tn * 32 + 8) * 8));
    CK(hipMemcpy(c.A, c.hA.data(), c.hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c.B, c.hB.data(), c.hB.size() * 4, hipMemcpyHostToDevice));
    printf("glds_probe: C[%d x %d] = A[%d x %d] . B[%d x %d]^T, %d tiles of 128 x 128, K steps of %d\n", c.M, c.N, c.M, c.K, c.N, c.K, c.tm * c.tn, GL_BK);
    if (ab) return run_ab(c, ab_rounds);
    int rc = 0;
    rc |= run<0, 2, 2, false>(c, "product loop, 2 stages, 2 workgroups / CU", true);
    rc |= run<0, 2, 2, true>(c, "  the same, instrumented", false);
    rc |= run<0, 2, 2, false, true>(c, "assembly DMAs, 2 stages, 2 / CU", true);
    rc |= run<0, 2, 2, true, true>(c, "  the same, instrumented", false);
    rc |= run<0, 2, 2, false, true, true>(c, "assembly DMAs + alternating s_setprio, 2 / CU", true);
    rc |= run<0, 2, 2, true, true, true>(c, "  the same, instrumented", false);
    rc |= run<0, 2, 2, true, false, true>(c, "builtin DMAs + alternating s_setprio, 2 / CU", false);
    rc |= run<0, 2, 1, true, true>(c, "assembly DMAs, 2 stages, 1 / CU", false);
    rc |= run<0, 3, 1, false, true>(c, "assembly DMAs, 3 stages, 1 / CU", true);
    rc |= run<0, 3, 1, true, true>(c, "  the same, instrumented", false);
    rc |= run_pipe<2, false, false>(c, "pipelined K-step boundary, assembly DMAs, 2 / CU");
    rc |= run_pipe<2, true, false>(c, "  + the next stage's DMAs between the MFMAs, 2 / CU");
    rc |= run_pipe<2, true, true>(c, "  + alternating s_setprio, 2 / CU");
    rc |= run_pipe<1, false, false>(c, "pipelined K-step boundary, 1 / CU");
    rc |= run_pipe<1, true, false>(c, "  + the next stage's DMAs between the MFMAs, 1 / CU");
    rc |= run<1, 2, 2, true>(c, "DMA + waits only (no MFMA), 2 stages, 2 / CU", false);
    rc |= run<2, 2, 2, true>(c, "no DMA in the loop (MFMA + LDS reads only)", false);
    rc |= run<0, 2, 1, true>(c, "product loop, 2 stages, 1 workgroup / CU", false);
    // the same product on operands that are all zero: if the rate depends on the DATA (power), this one is faster
    CK(hipMemset(c.A, 0, c.hA.size() * 4)); CK(hipMemset(c.B, 0, c.hB.size() * 4));
    rc |= run<0, 2, 2, true>(c, "product loop, 2 stages, 2 / CU, ALL-ZERO operands", false);
    rc |= run<0, 2, 2, true, true>(c, "assembly DMAs, 2 stages, 2 / CU, ALL-ZERO operands", false);
    rc |= run<0, 2, 1, true>(c, "product loop, 2 stages, 1 / CU, ALL-ZERO operands", false);
    CK(hipMemcpy(c.A, c.hA.data(), 4096 * 4, hipMemcpyHostToDevice));
    for (int vary = 0; vary < 2; ++vary)
        for (int wps = 1; wps <= 2; ++wps) {
            const int iters = 16384, grid = 256 * wps;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&]() {
                if (vary) hipLaunchKernelGGL(mfma_only<true>, dim3(grid), dim3(256), 0, 0, c.C, iters, c.A, c.stats);
                else hipLaunchKernelGGL(mfma_only<false>, dim3(grid), dim3(256), 0, 0, c.C, iters, c.A, c.stats);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); launch(); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 2;
            unsigned long long h[2]; CK(hipMemcpy(h, c.stats, 16, hipMemcpyDeviceToHost));
            const double tf = (double)grid * 4 * iters * 32.0 * 4096.0 / (ms * 1e-3) / 1e12;
            printf("MFMA only, %s operands, %d wave(s) / SIMD: %8.2f ms  %6.1f TFLOP/s   s_memtime at %5.3f GHz\n",
                   vary ? "16 different" : "constant    ", wps, ms, tf, h[0] / (h[1] * 10.0));
        }
    return rc;
}
