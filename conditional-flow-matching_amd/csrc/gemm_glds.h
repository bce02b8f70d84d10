// gemm_glds.h — the fp32-MFMA tile engine with DIRECT-TO-LDS operand loads (gfx950 global_load_lds_dwordx4), for
// products whose two operands are both K-contiguous in memory ([rows][k]: point clouds, activations, nn.Linear weights).
//
//   C tile [128 x 128] += sum_k A(i, k) . B(j, k)   on v_mfma_f32_32x32x2_f32, ascending k: bitwise the same fmaf
//   chain per output as gemm_core.h (MFMA j of a K step covers k = 2 j on lanes 0-31 and 2 j + 1 on lanes 32-63).
//
// Why a second engine (round 4, profiles/r4_gemm_probes.txt): in gemm_core.h an operand travels global -> registers ->
// (transposing ds_write_b32 x 4) -> K-major LDS.  Probes priced that path at 21.5 % of the asymptotic time (the loads
// 13 %, the stores 8.5 %) while fragment reads and the barrier were free.  Here nothing of it is left:
//   * a stage of an operand is ROW-major in LDS, 32 floats (128 bytes = one cache line) per row, written by the memory
//     pipeline itself: one global_load_lds_dwordx4 per wave moves 8 rows x 128 B (lane l: row l >> 3, 16-byte slot
//     l & 7) — no staging registers, no LDS store instructions, no per-element VALU, full-line requests;
//   * the LDS image is lane-linear (the hardware writes base + 16 x lane), so the bank swizzle lives on the SOURCE side:
//     the lane that fills slot s of row r fetches k-quad s ^ g(r), g(r) = (r >> 1) & 7, and a fragment read of k-quad q
//     of row r looks at slot q ^ g(r): the 16 lanes of every ds_read_b128 group then hit 16 distinct 16-byte columns;
//   * a lane's ds_read_b128 holds k = 4 q .. 4 q + 3 of its row: MFMA 2 q takes (x | y) and MFMA 2 q + 1 (z | w), the
//     half-waves selecting their component with one v_cndmask per operand — the k order of gemm_core.h exactly;
//   * the DMAs are INLINE ASSEMBLY: the compiler treats __builtin_amdgcn_global_load_lds as an LDS write that any later
//     ds_read may alias and puts an s_waitcnt vmcnt(0) in front of the first fragment read of every K step — behind the
//     DMAs of the NEXT stage, issued a moment earlier — so a wave never overlaps a stage's loads with its own MFMAs
//     (measured: 116 -> 129 TFLOP/s at 4096 x 4096 x 3136, 112 -> 125 at K = 800; profiles/r5_experiments.txt).  The
//     ordering is the explicit s_waitcnt vmcnt + s_barrier at the K-step boundary;
//   * SGPR base + 32-bit lane offset addressing: a stage advances by ONE scalar add per operand, and the LDS address of
//     a piece is wave uniform — 4 scalar instructions per DMA;
//   * PADDED operands make every DMA unconditional: the operand has one more row than it has rows (index `nrows`, all
//     zeros) and its rows are zero filled up to a multiple of GL_BK floats: lanes of rows beyond the operand read the
//     zero row, and there is no k tail;
//   * PIPELINED K-step boundary: the LAST k-quad of stage s is computed AFTER the barrier of step s + 1 — its fragments
//     are in registers by then (the barrier only protects the LDS buffer) and its eight MFMAs cover the first fragment
//     reads of stage s + 1.
// Tried and removed (round 5, interleaved A/B with per-launch clocks, profiles/r5_experiments.txt): the compiler-builtin
// DMAs (above); alternating s_setprio between the two workgroups of a CU (no gain beyond run-to-run spread in the probe,
// 252.8 vs 246.9 us on the cost matrix); the un-pipelined boundary (256.7 vs 246.9 us).
#pragma once
#include "cfm_common.h"

typedef float gl_f32x16 __attribute__((ext_vector_type(16)));

#define GL_BM 128
#define GL_BN 128
#define GL_BK 32
#define GL_STAGE_FLOATS ((GL_BM + GL_BN) * GL_BK)            // 8192 floats = 32 KiB per stage
#define GL_LDS_BYTES (2 * GL_STAGE_FLOATS * 4)               // two stages: 64 KiB

// wait for this wave's DMAs, then the workgroup barrier: orders LDS-DMA writes against the ds_reads behind it
__device__ __forceinline__ void gl_wait_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct GldsCore {
    gl_f32x16 acc[2][2];                 // wave (wm, wn): rows wm * 64 + m * 32 + rho, columns wn * 64 + n * 32 + (lane & 31)

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
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
    // (m0: the AMDGPU backend RESERVES m0 — it is never allocated to a value, only written immediately in front of the few
    //  instructions that consume it (LDS-direct loads, movrel, sendmsg, GWS) — and rejects it on a clobber list with
    //  "reserved registers ... may lead to undefined behaviour"; the kernels built on this engine contain none of those
    //  consumers: tools/isa_report.py lists every m0 access of the code object, all of them are this asm's)
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
// Main loop over [0, Kp) on padded operands (A: [M + 1][lda], B: [N + 1][ldb], Kp % GL_BK == 0).  lds: GL_LDS_BYTES of
// dynamic LDS.  ONE body in the loop, the last stage peeled behind it, no branch around an MFMA: accumulators that reach a
// K step over two paths get copied.  The DMAs of the loop's last trip have no stage left to fetch: they
// fetch the last stage once more into the buffer nobody reads again and are drained before the function returns (no
// DMA may outlive the workgroup's LDS allocation).
__device__ __forceinline__ void gl_run_padded_pipe(GldsCore& g, float* __restrict__ lds, const float* __restrict__ A, int lda, int row0,
                                                   int M, const float* __restrict__ B, int ldb, int col0, int N, int Kp) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
