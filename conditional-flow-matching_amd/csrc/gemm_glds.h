// gemm_glds.h — the fp32-MFMA tile engine with DIRECT-TO-LDS operand loads (gfx950 global_load_lds_dwordx4), for
// products whose two operands are both K-contiguous in memory ([rows][k]: point clouds, activations, nn.Linear weights).
//
//   C tile [128 x 128] += sum_k A(i, k) . B(j, k)   on v_mfma_f32_32x32x2_f32, ascending k: bitwise the same fmaf
//   chain per output as gemm_core.h (MFMA j of a K step covers k = 2 j on lanes 0-31 and 2 j + 1 on lanes 32-63).
//
// Why a second engine (round 4, profiles/r4_gemm_probes.txt): in gemm_core.h an operand travels global -> registers ->
// (transposing ds_write_b32 x 4) -> K-major LDS.  Probes priced that path at 21.5 % of the asymptotic time (the loads
// 13 %, the stores 8.5 %) while fragment reads and the barrier were free; re-placing, re-addressing or deepening it
// moved nothing.  Here nothing of it is left:
//   * a stage of an operand is ROW-major in LDS, 32 floats (128 bytes = one cache line) per row, written by the memory
//     pipeline itself: one global_load_lds_dwordx4 per wave moves 8 rows x 128 B (lane l: row l >> 3, 16-byte slot
//     l & 7) — no staging registers, no LDS store instructions, no per-element VALU, full-line requests;
//   * the LDS image is lane-linear (the hardware writes base + 16 x lane), so the bank swizzle lives on the SOURCE side:
//     the lane that fills slot s of row r fetches k-quad s ^ g(r), g(r) = (r >> 1) & 7, and a fragment read of k-quad q
//     of row r looks at slot q ^ g(r): the 16 lanes of every ds_read_b128 group then hit 16 distinct 16-byte columns;
//   * a lane's ds_read_b128 holds k = 4 q .. 4 q + 3 of its row: MFMA 2 q takes (x | y) and MFMA 2 q + 1 (z | w), the
//     half-waves selecting their component with one v_cndmask per operand — the k order of gemm_core.h exactly;
//   * rows beyond the operand and k-quads beyond K are fetched from a 16-byte block of zeros (`zeros`), so edges need
//     no masks anywhere;
//   * two stages, one barrier per K step: wait for the stage's DMA (issued a whole step earlier), barrier, issue the next
//     stage's DMA into the buffer everybody has just finished reading, compute.
#pragma once
#include "cfm_common.h"

typedef float gl_f32x16 __attribute__((ext_vector_type(16)));

#define GL_BM 128
#define GL_BN 128
#define GL_BK 32
#define GL_STAGE_FLOATS ((GL_BM + GL_BN) * GL_BK)            // 8192 floats = 32 KiB per stage
#define GL_LDS_BYTES (2 * GL_STAGE_FLOATS * 4)               // two stages: 64 KiB

__device__ __forceinline__ void gl_dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// wait for this wave's DMAs, then the workgroup barrier: orders LDS-DMA writes against the ds_reads behind it
__device__ __forceinline__ void gl_wait_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GldsCore {
    gl_f32x16 acc[2][2];                 // wave (wm, wn): rows wm * 64 + m * 32 + rho, columns wn * 64 + n * 32 + (lane & 31)
    const float* pa[4]; const float* pb[4];   // this lane's source of each of its 4 + 4 DMA pieces at k0 = 0 (or `zeros`)
    int kqa[4], kqb[4];                  // first k of the lane's 16 bytes inside a stage (swizzled k-quad x 4)
    bool za[4], zb[4];                   // piece lies in a row beyond the operand: always zeros

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

    __device__ __forceinline__ void bind(const float* __restrict__ A, int lda, int row0, int M,
                                         const float* __restrict__ B, int ldb, int col0, int N) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 8 * (wv + 4 * q) + (lane >> 3);          // row of the tile this lane fills in piece q
            const int kq = (lane & 7) ^ ((r >> 1) & 7);            // the k-quad that belongs into its slot
            kqa[q] = 4 * kq; kqb[q] = 4 * kq;
            za[q] = row0 + r >= M; zb[q] = col0 + r >= N;
            pa[q] = A + (size_t)(za[q] ? 0 : row0 + r) * lda + 4 * kq;
            pb[q] = B + (size_t)(zb[q] ? 0 : col0 + r) * ldb + 4 * kq;
        }
    }

    // DMA of the stage [k0, k0 + 32) into LDS stage `st` (8 pieces of 1 KiB per wave)
    __device__ __forceinline__ void issue(float* __restrict__ lds, int st, int k0, int K, const float* __restrict__ zeros) {
        const int wv = threadIdx.x >> 6;
        float* As = lds + st * GL_STAGE_FLOATS;
        float* Bs = As + GL_BM * GL_BK;
        const bool tail = k0 + GL_BK > K;                           // (uniform) some k-quads lie beyond K
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool oa = za[q] || (tail && k0 + kqa[q] >= K);
            const bool ob = zb[q] || (tail && k0 + kqb[q] >= K);
            gl_dma16(oa ? zeros : pa[q] + k0, As + (wv + 4 * q) * 256);
            gl_dma16(ob ? zeros : pb[q] + k0, Bs + (wv + 4 * q) * 256);
        }
    }

    // the 64 MFMAs of a stage
    __device__ __forceinline__ void compute(const float* __restrict__ lds, int st) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int wm = wv >> 1, wn = wv & 1;
        const bool hi = lane >= 32;
        const int fr = lane & 31, sw = (fr >> 1) & 7;
        const float* As = lds + st * GL_STAGE_FLOATS + (wm * 64 + fr) * GL_BK;
        const float* Bs = lds + st * GL_STAGE_FLOATS + GL_BM * GL_BK + (wn * 64 + fr) * GL_BK;
        // fragments of k-quad q + 1 are requested before the MFMAs of k-quad q are issued (two register sets)
        float4 ta0, ta1, tb0, tb1, na0, na1, nb0, nb1;
        {
            const int so = 4 * sw;
            ta0 = *reinterpret_cast<const float4*>(As + so); ta1 = *reinterpret_cast<const float4*>(As + 32 * GL_BK + so);
            tb0 = *reinterpret_cast<const float4*>(Bs + so); tb1 = *reinterpret_cast<const float4*>(Bs + 32 * GL_BK + so);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q < 7) {
                const int so = 4 * ((q + 1) ^ sw);                   // swizzled slot of the next k-quad in this lane's rows
                na0 = *reinterpret_cast<const float4*>(As + so); na1 = *reinterpret_cast<const float4*>(As + 32 * GL_BK + so);
                nb0 = *reinterpret_cast<const float4*>(Bs + so); nb1 = *reinterpret_cast<const float4*>(Bs + 32 * GL_BK + so);
            }
            __builtin_amdgcn_sched_barrier(0);       // (the scheduler otherwise sinks these reads behind the MFMAs and waits for them at once)
            const float a00 = hi ? ta0.y : ta0.x, a01 = hi ? ta1.y : ta1.x;      // k = 4 q (+ 1 on the upper half-wave)
            const float b00 = hi ? tb0.y : tb0.x, b01 = hi ? tb1.y : tb1.x;
            const float a10 = hi ? ta0.w : ta0.z, a11 = hi ? ta1.w : ta1.z;      // k = 4 q + 2 (+ 1)
            const float b10 = hi ? tb0.w : tb0.z, b11 = hi ? tb1.w : tb1.z;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b00, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b01, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b00, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b10, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b11, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b10, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b11, acc[1][1], 0, 0, 0);
            if (q < 7) { ta0 = na0; ta1 = na1; tb0 = nb0; tb1 = nb1; }
        }
    }

    // Main loop over [0, K).  lds: GL_LDS_BYTES of dynamic LDS; zeros: >= 16 bytes of zeros in global memory.
    __device__ __forceinline__ void run(float* __restrict__ lds, const float* __restrict__ A, int lda, int row0, int M,
                                        const float* __restrict__ B, int ldb, int col0, int N, int K,
                                        const float* __restrict__ zeros) {
        bind(A, lda, row0, M, B, ldb, col0, N);
        issue(lds, 0, 0, K, zeros);
        int st = 0;
        for (int k0 = 0; k0 < K; k0 += GL_BK) {
            gl_wait_barrier();                                      // stage `st` has landed; everybody is done with the other one
#ifndef GL_DBG
#define GL_DBG 0      // timing probes (results WRONG): bit 0: no DMA inside the loop
#endif
            if (!(GL_DBG & 1) && k0 + GL_BK < K) issue(lds, st ^ 1, k0 + GL_BK, K, zeros);
            compute(lds, st);
            st ^= 1;
        }
    }

    // epilogue geometry: C/D layout of the MFMA — column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    __device__ __forceinline__ static int row_of(int m, int r) {
        const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
        return wm * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ static int col_of(int n) {
        const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
        return wn * 64 + n * 32 + (lane & 31);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Second generation of the stage loads (measured with tools/probe/glds_probe.hip at the end of round 4, not yet used by a
// product kernel — profiles/r4_glds_probe.txt):
//   * the compiler treats __builtin_amdgcn_global_load_lds as an LDS write that any later ds_read may alias and puts an
//     s_waitcnt vmcnt(0) in front of the first fragment read of EVERY K step — behind the DMAs of the NEXT stage, issued
//     a moment earlier.  GldsCore::run therefore never overlaps a stage's loads with its own wave's MFMAs (a lone wave
//     keeps the matrix pipe 61 % busy).  DMAs written as inline assembly are invisible to that pass; the ordering is the
//     explicit s_waitcnt vmcnt + s_barrier at the top of the K step.  Lone wave: 74 %; 2 workgroups per CU, K = 3136:
//     103 -> 114 TFLOP/s.
//   * SGPR base + 32-bit lane offset addressing: a stage advances by ONE scalar add per operand, and the LDS address of a
//     piece is wave uniform — 4 scalar instructions per DMA instead of ~12 (64-bit vector add, zero-row selects,
//     v_readfirstlane + s_mov m0).
//   * PADDED operands make every DMA unconditional: the operand has one more row than it has rows (index `nrows`, all
//     zeros) and its rows are zero filled up to a multiple of GL_BK floats (row pitch >= that): lanes of rows beyond the
//     operand read the zero row, and there is no k tail.
//   * the two workgroups of a CU are served oldest first: the older one runs ~25 % faster, and with two rounds of tiles
//     per CU half of the slots sit empty for the last 15 % of a launch.  Alternating the issue priority per K step
//     (s_setprio, parity from the caller) lets both finish together: 114 -> 120 TFLOP/s.
struct GldsDma {
    unsigned offa[4], offb[4];          // byte offset of this lane's 16 bytes from the operand base at k0 = 0
    unsigned la[4], lb[4];              // LDS byte address of the wave's piece q inside stage 0 (wave uniform)

    // A: [M + 1][lda], B: [N + 1][ldb], zero row at index M / N, rows zero filled to a multiple of GL_BK
    __device__ __forceinline__ void bind_padded(int lda, int row0, int M, int ldb, int col0, int N, float* lds) {
        const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 8 * (wv + 4 * q) + (lane >> 3);
            const int kq = (lane & 7) ^ ((r >> 1) & 7);
            offa[q] = (unsigned)(((size_t)min(row0 + r, M) * lda + 4 * kq) * 4);
            offb[q] = (unsigned)(((size_t)min(col0 + r, N) * ldb + 4 * kq) * 4);
            la[q] = base + (unsigned)((wv + 4 * q) * 256 * 4);
            lb[q] = la[q] + GL_BM * GL_BK * 4;
        }
    }
    static __device__ __forceinline__ void dma(unsigned off, const float* base, unsigned ldsaddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(ldsaddr) : "memory");
    }
    // stage `st` <- [k0, k0 + GL_BK) of both operands; Ak = A + k0, Bk = B + k0 (uniform)
    __device__ __forceinline__ void issue(const float* Ak, const float* Bk, int st) const {
        const unsigned so = (unsigned)st * (GL_STAGE_FLOATS * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { dma(offa[q], Ak, la[q] + so); dma(offb[q], Bk, lb[q] + so); }
    }
};

// Main loop over [0, Kp) on padded operands (Kp % GL_BK == 0).  prio_phase: 0 / 1 for the two workgroups that share a CU
// (alternating s_setprio), -1: leave the priority alone.  lds: GL_LDS_BYTES of dynamic LDS.
__device__ __forceinline__ void gl_run_padded(GldsCore& g, float* __restrict__ lds, const float* __restrict__ A, int lda, int row0,
                                              int M, const float* __restrict__ B, int ldb, int col0, int N, int Kp, int prio_phase) {
    GldsDma d;
    d.bind_padded(lda, row0, M, ldb, col0, N, lds);
    d.issue(A, B, 0);
    int st = 0;
    for (int k0 = 0; k0 < Kp; k0 += GL_BK) {
        gl_wait_barrier();
        if (prio_phase >= 0) {
            if ((prio_phase ^ (k0 / GL_BK)) & 1) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0");
        }
        if (k0 + GL_BK < Kp) d.issue(A + k0 + GL_BK, B + k0 + GL_BK, st ^ 1);
        g.compute(lds, st);
        st ^= 1;
    }
    if (prio_phase >= 0) asm volatile("s_setprio 0");
}

// The same loop with a PIPELINED K-step boundary (tools/probe/glds_probe.hip: glds_pipe; 4096 x 4096 x 3136, two workgroups
// per CU: 112.6 -> 125.5 TFLOP/s in the probe's own form, 114.5 -> 113.9 in this one — same loop body, another process; the
// chip's clock moves by 10 % with what ran before, so the gain is NOT established yet): gl_run_padded drains the matrix pipe at every K step — wait, barrier, DMA
// issue, then eight fragment reads nothing covers.  Here the LAST k-quad of stage s is computed AFTER the barrier of step
// s + 1: its fragments are in registers by then (the barrier only protects the LDS buffer), and its eight MFMAs cover the
// first fragment reads of stage s + 1.  (Issuing the DMAs of stage s + 2 one behind each of those MFMAs was slower with two
// workgroups per CU — 112.9 — and faster with one.)  ONE body in the loop, the last stage peeled behind it, no branch
// around an MFMA: accumulators that reach a K step over two paths get copied.  The DMAs of the
// loop's last trip have no stage left to fetch: they fetch the last stage once more into the buffer nobody reads again
// and are drained before the function returns (no DMA may outlive the workgroup's LDS allocation).  Same k order per
// output: bitwise the results of GldsCore::run.
struct GlFrag { float4 a0, a1, b0, b1; };
__device__ __forceinline__ GlFrag gl_read_frag(const float* __restrict__ lds, int st, int q) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wm = wv >> 1, wn = wv & 1;
    const int fr = lane & 31, so = 4 * (q ^ ((fr >> 1) & 7));
    const float* As = lds + st * GL_STAGE_FLOATS + (wm * 64 + fr) * GL_BK;
    const float* Bs = lds + st * GL_STAGE_FLOATS + GL_BM * GL_BK + (wn * 64 + fr) * GL_BK;
    GlFrag f;
    f.a0 = *reinterpret_cast<const float4*>(As + so); f.a1 = *reinterpret_cast<const float4*>(As + 32 * GL_BK + so);
    f.b0 = *reinterpret_cast<const float4*>(Bs + so); f.b1 = *reinterpret_cast<const float4*>(Bs + 32 * GL_BK + so);
    return f;
}
__device__ __forceinline__ void gl_mfma8(GldsCore& g, const GlFrag& t) {
    const bool hi = (threadIdx.x & 63) >= 32;
    const float a00 = hi ? t.a0.y : t.a0.x, a01 = hi ? t.a1.y : t.a1.x, b00 = hi ? t.b0.y : t.b0.x, b01 = hi ? t.b1.y : t.b1.x;
    const float a10 = hi ? t.a0.w : t.a0.z, a11 = hi ? t.a1.w : t.a1.z, b10 = hi ? t.b0.w : t.b0.z, b11 = hi ? t.b1.w : t.b1.z;
    g.acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b00, g.acc[0][0], 0, 0, 0);
    g.acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, b01, g.acc[0][1], 0, 0, 0);
    g.acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b00, g.acc[1][0], 0, 0, 0);
    g.acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01, g.acc[1][1], 0, 0, 0);
    g.acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b10, g.acc[0][0], 0, 0, 0);
    g.acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, b11, g.acc[0][1], 0, 0, 0);
    g.acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b10, g.acc[1][0], 0, 0, 0);
    g.acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, b11, g.acc[1][1], 0, 0, 0);
}
__device__ __forceinline__ void gl_run_padded_pipe(GldsCore& g, float* __restrict__ lds, const float* __restrict__ A, int lda, int row0,
                                                   int M, const float* __restrict__ B, int ldb, int col0, int N, int Kp, int prio_phase) {
    GldsDma d;
    d.bind_padded(lda, row0, M, ldb, col0, N, lds);
    const int nsteps = Kp / GL_BK;
    if (nsteps <= 0) return;
    d.issue(A, B, 0);
    if (nsteps > 1) { d.issue(A + GL_BK, B + GL_BK, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    GlFrag t = gl_read_frag(lds, 0, 0);
    for (int s = 0; s + 1 < nsteps; ++s) {
        const int st = s & 1;
        if (prio_phase >= 0) { if ((prio_phase ^ s) & 1) asm volatile("s_setprio 1"); else asm volatile("s_setprio 0"); }
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const GlFrag n = gl_read_frag(lds, st, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            gl_mfma8(g, t);
            t = n;
        }
        // stage s + 1 has landed (issued a whole step ago); this wave's reads of stage s are complete (lgkmcnt); behind
        // the barrier nobody reads stage s any more: its buffer takes stage s + 2
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int sn = min(s + 2, nsteps - 1);
        d.issue(A + (size_t)sn * GL_BK, B + (size_t)sn * GL_BK, st);
        const GlFrag n = gl_read_frag(lds, st ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        gl_mfma8(g, t);                              // the last k-quad of stage s, behind the barrier of step s + 1
        t = n;
    }
    {
        const int st = (nsteps - 1) & 1;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const GlFrag n = gl_read_frag(lds, st, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            gl_mfma8(g, t);
            t = n;
        }
        gl_mfma8(g, t);
    }
    if (prio_phase >= 0) asm volatile("s_setprio 0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
