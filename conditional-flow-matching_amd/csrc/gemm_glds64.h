// gemm_glds64.h — the direct-to-LDS fp32-MFMA engine (gemm_glds.h) in the 64 x 64 form the MLP layers need (round 6).
//
//   C tile [64 x 64] += sum_k A(i, k) . B(j, k),  both operands K-contiguous in memory ([rows][k]: activations and
//   nn.Linear weights — the forward of torchcfm/models/models.py:10-21), on v_mfma_f32_16x16x4_f32.
//
// Why a third form.  At the C3 layer shapes (4096 x 512 outputs) a 128 x 128 tile gives 128 workgroups for 256 CUs; the
// layers therefore ran on the register-staged 64 x 64 core of gemm_core.h (global -> registers -> four transposing
// ds_write_b32 -> K-major LDS), the path round 4 priced at 21.5 % of the asymptotic time.  Here:
//   * a stage of an operand is ROW-major in LDS, 32 floats (128 B) per row, written by global_load_lds_dwordx4: one DMA
//     per wave moves 8 rows x 128 B, no staging registers, no LDS store instructions;
//   * the bank swizzle lives on the SOURCE side, as in gemm_glds.h: the lane that fills slot s of LDS row L fetches
//     k-quad s ^ g(L), g(L) = (L >> 1) & 7;
//   * a lane's fragment is ONE ds_read_b128 per 16 x 16 block and 16 k: lane (r = lane & 15, f = lane >> 4) reads k-quad
//     4 j + f of its row, i.e. k = 16 j + 4 f + {0, 1, 2, 3}; MFMA t of the group takes component t of every lane, so it
//     covers k = 16 j + t + {0, 4, 8, 12}.  Four reads feed SIXTEEN MFMAs (2 x 2 blocks x 4 components): 16 LDS cycles per
//     512 matrix-pipe cycles, no selects, and the four accumulators of a wave rotate (a 16x16x4 MFMA is 8 passes; its
//     accumulator comes round again after four of them).  Every ds_read_b128 lane group sees 16 distinct 16-byte bank
//     slots (checked exhaustively for the layout below);
//   * the k order per output is therefore: K steps ascending, inside a step the 16-blocks ascending, inside a block
//     t = 0 .. 3, inside an MFMA the hardware's order over f.  It is a fixed order (results are deterministic and the
//     same for every tile position), but NOT the plain ascending chain of gemm_core.h: the forward of a layer differs
//     from the round-5 bits in the last ulp (tests/test_gpu_train.py compares against the fp64 oracle at 1e-5);
//   * the B rows of a tile are stored PERMUTED in LDS (LDS row wn * 32 + u * 16 + c holds tile column wn * 32 + 2 c + u):
//     a lane then owns two ADJACENT output columns (blocks u = 0, 1) — 8-byte stores — while the fragment read of a
//     block still walks 16 consecutive LDS rows (conflict free).  The permutation costs nothing: it is the DMA's source
//     address;
//   * NST stages, ONE barrier per K step, DMAs issued NST - 1 steps ahead (they hold no registers, so depth is free);
//   * edges: rows beyond the operand are clamped to its last row (the epilogue masks them); K % 16 == 0 is required
//     (512, 784: every C3 layer) and a last step of 16 k fetches k-quads 0 .. 3 twice instead of reading past the row.
#pragma once
#include "cfm_common.h"
#include "gemm_glds.h"

typedef float g6_f32x4 __attribute__((ext_vector_type(4)));

#define G6_BM 64
#define G6_BN 64
#define G6_BK 32
#define G6_STAGE_FLOATS ((G6_BM + G6_BN) * G6_BK)            // 4096 floats = 16 KiB per stage

template <int NST>
struct Glds64 {
    static constexpr int EM = 2, EU = 2, ER = 4;             // epilogue: at(m, u, r)
    static constexpr int LDS_BYTES = NST * G6_STAGE_FLOATS * 4;
    g6_f32x4 acc[2][2];
    unsigned offa[2], offb[2], offa_t[2], offb_t[2];        // this lane's 16 bytes at k0 = 0 (…_t: the 16-k tail step)
    unsigned la[2], lb[2];                                   // LDS byte address of the wave's pieces in stage 0 (uniform)

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    }
    __device__ __forceinline__ float at(int m, int u, int r) const { return acc[m][u][r]; }
    // C/D layout of the 16x16x4 MFMA: column = lane & 15, row = 4 (lane >> 4) + r
    __device__ __forceinline__ static int row_of(int m, int r) {
        const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
        return wm * 32 + m * 16 + 4 * (lane >> 4) + r;
    }
    __device__ __forceinline__ static int col_lo() {
        const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
        return wn * 32 + 2 * (lane & 15);
    }

    // A: [M][lda], B: [N][ldb]; rows clamped to the last one
    __device__ __forceinline__ void bind(int lda, int row0, int M, int ldb, int col0, int N, float* lds) {
        const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int L = 8 * (wv + 4 * q) + (lane >> 3);                 // LDS row of the operand's stage
            const int kq = (lane & 7) ^ ((L >> 1) & 7);
            const int ra = min(row0 + L, M - 1);
            const int pc = (L & 32) + 2 * (L & 15) + ((L >> 4) & 1);      // tile column stored in LDS row L
            const int rb = min(col0 + pc, N - 1);
            offa[q] = (unsigned)(((size_t)ra * lda + 4 * kq) * 4);
            offb[q] = (unsigned)(((size_t)rb * ldb + 4 * kq) * 4);
            offa_t[q] = (unsigned)(((size_t)ra * lda + 4 * (kq & 3)) * 4);
            offb_t[q] = (unsigned)(((size_t)rb * ldb + 4 * (kq & 3)) * 4);
            la[q] = base + (unsigned)((wv + 4 * q) * 256 * 4);
            lb[q] = la[q] + G6_BM * G6_BK * 4;
        }
    }
    // stage buffer `st` <- [k0, k0 + 32) of both operands; Ak = A + k0, Bk = B + k0 (uniform); tail: only 16 k are valid
    __device__ __forceinline__ void issue(const float* Ak, const float* Bk, int st, bool tail) const {
        const unsigned so = (unsigned)st * (G6_STAGE_FLOATS * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            GldsDma::dma(tail ? offa_t[q] : offa[q], Ak, la[q] + so);
            GldsDma::dma(tail ? offb_t[q] : offb[q], Bk, lb[q] + so);
        }
    }

    struct Frag { float4 a[2], b[2]; };
    // fragments of the 16-k group j of stage buffer `st`
    __device__ __forceinline__ static Frag read_frag(const float* __restrict__ lds, int st, int j) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wm = wv >> 1, wn = wv & 1;
        const int r = lane & 15, f = lane >> 4;
        const int so = 4 * ((4 * j + f) ^ ((r >> 1) & 7));
        const float* As = lds + st * G6_STAGE_FLOATS + (wm * 32 + r) * G6_BK + so;
        const float* Bs = lds + st * G6_STAGE_FLOATS + G6_BM * G6_BK + (wn * 32 + r) * G6_BK + so;
        Frag t;
        t.a[0] = *reinterpret_cast<const float4*>(As); t.a[1] = *reinterpret_cast<const float4*>(As + 16 * G6_BK);
        t.b[0] = *reinterpret_cast<const float4*>(Bs); t.b[1] = *reinterpret_cast<const float4*>(Bs + 16 * G6_BK);
        return t;
    }
    // The MFMAs are INLINE ASSEMBLY with the accumulator tied to its own VGPRs ("+v"): with the builtin the register
    // allocator un-ties the destination of a step's last MFMAs and rotates the accumulators back at the loop head — 24
    // v_accvgpr read / mov / write per K step, each waiting for the matrix pipe to drain (tools/isa_report.py).  What the
    // compiler's hazard recogniser no longer sees is covered by hand: an accumulator comes round again after three other
    // 8-pass MFMAs (no wait states needed), fragments come from ds_reads (s_waitcnt is placed on the asm's operands as on any
    // instruction), and run() ends with s_nop 15 x 2 in front of the first VALU read of an accumulator.
    __device__ __forceinline__ void mfma16(const Frag& t) {
#define G6_MFMA(ACC_, A_, B_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC_) : "v"(A_), "v"(B_))
#define G6_MM(C_) \
        G6_MFMA(acc[0][0], t.a[0].C_, t.b[0].C_); G6_MFMA(acc[0][1], t.a[0].C_, t.b[1].C_); \
        G6_MFMA(acc[1][0], t.a[1].C_, t.b[0].C_); G6_MFMA(acc[1][1], t.a[1].C_, t.b[1].C_);
        G6_MM(x) G6_MM(y) G6_MM(z) G6_MM(w)
#undef G6_MM
#undef G6_MFMA
    }

    // Main loop over [0, K), K % 16 == 0, K >= 16.  lds: LDS_BYTES of dynamic LDS.
    __device__ __forceinline__ void run(float* __restrict__ lds, const float* __restrict__ A, int lda, int row0, int M,
                                        const float* __restrict__ B, int ldb, int col0, int N, int K) {
        bind(lda, row0, M, ldb, col0, N, lds);
        const int nsteps = (K + G6_BK - 1) / G6_BK;
        const bool has_tail = (K % G6_BK) != 0;
        // the first NST - 1 stages (steps beyond the last one fetch the last one again: every DMA is unconditional and
        // the wait counts below are constants)
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) {
            const int sn = min(s, nsteps - 1);
            issue(A + (size_t)sn * G6_BK, B + (size_t)sn * G6_BK, s, has_tail && sn == nsteps - 1);
        }
        int st = 0, sf = NST - 1;                                  // buffer of step s, buffer the step's DMAs fill
        // stage s has landed (the NST - 2 younger stages may still be in flight); behind the barrier nobody reads stage
        // s - 1 any more: its buffer takes stage s + NST - 1.  ONE body in the loop (no branch around an MFMA: accumulators
        // that reach a step over two paths get copied, gemm_core.h); the 16-k tail step is peeled behind it.
#define G6_STEP_HEAD(S_) \
        if (NST == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else if (NST == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        { const int sn = min((S_) + NST - 1, nsteps - 1); \
          issue(A + (size_t)sn * G6_BK, B + (size_t)sn * G6_BK, sf, has_tail && sn == nsteps - 1); }
        // Pipelined K-step boundary: the second 16-k group of step s is computed BEHIND the barrier of step s + 1 — its
        // fragments are in registers by then (the barrier protects the LDS buffer only) and its sixteen MFMAs cover the
        // first fragment reads of stage s + 1.
        const int nfull = K / G6_BK;
        G6_STEP_HEAD(0)
        Frag t0 = read_frag(lds, st, 0);
        for (int s = 0; s + 1 < nfull; ++s) {
            const Frag t1 = read_frag(lds, st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(t0);
            st = (st + 1 == NST) ? 0 : st + 1;
            sf = (sf + 1 == NST) ? 0 : sf + 1;
            G6_STEP_HEAD(s + 1)
            t0 = read_frag(lds, st, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(t1);
        }
        if (nfull > 0) {                                           // the last full step (t0 holds its first group)
            const Frag t1 = read_frag(lds, st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(t0);
            if (has_tail) {                                        // (uniform) 16 more k: one group of the next stage
                st = (st + 1 == NST) ? 0 : st + 1;
                sf = (sf + 1 == NST) ? 0 : sf + 1;
                G6_STEP_HEAD(nfull)
                t0 = read_frag(lds, st, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(t1);
                mfma16(t0);
            } else {
                mfma16(t1);
            }
        } else {
            mfma16(t0);                                            // K = 16: the tail is the only step
        }
#undef G6_STEP_HEAD
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");    // no DMA may outlive the workgroup's LDS allocation; MFMA -> VALU read
    }
};
