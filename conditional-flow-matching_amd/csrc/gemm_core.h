// gemm_core.h — the fp32-MFMA tile engine shared by every dense product of the path (gfx950):
//   mlp_layer (MLP forward, mlp.hip), gemm_f32_mfma (dgrad / wgrad, mlp_train.hip).  (The cost matrix runs on the
//   direct-to-LDS engine, gemm_glds.h.)
//
//   C tile [BM x BN] += sum_k A(i, k) . B(k, j)   on v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an ascending-k
//   fmaf chain per output, whatever the tile shape — the k order below never changes).
//
// Design (256 threads = 2 x 2 waves, one 32x32 accumulator block per 32 x 32 outputs):
//   * LDS tiles are K-MAJOR for both operands: T[k][row], row stride LD (even).  The fragment of a 32x32x2 MFMA is
//     lane -> (row = lane & 31, k = lane >> 5): a wave reads 32 consecutive floats of two k rows — conflict free.
//     With two MFMA row blocks per wave the wave's 64 rows are INTERLEAVED (block t owns rows 2 j + t), so both
//     fragments of a lane are adjacent: one ds_read_b64 feeds two MFMAs (4 MFMAs per 2 LDS reads at 128 x 128).
//     The accumulator layout follows: a lane holds outputs of adjacent columns (2 j, 2 j + 1) -> 8-byte stores.
//   * two LDS stages, ONE barrier per K step: global loads for step s + 2 are issued right after the barrier of
//     step s and have a whole compute phase (BK / 2 x MT x NT MFMAs = 4096 cycles at 128 x 128 x 32) to land
//     before they are written to the stage that step s + 1 has finished reading.
//   * operand storage in memory, per operand: K-contiguous ([row][k]: activations, nn.Linear weights, the point
//     clouds) is loaded as float4 along k and transposed on the way into LDS (4 ds_write_b32, row stride
//     LD = rows + 2: 2-way bank conflicts, free on ds_write_b32); K-major ([k][row]: W as the B of dgrad, both
//     operands of wgrad) is loaded as float4 along the row and stored with ds_write_b128 (LD = rows + 4).
//     Unaligned operands (row pitch or base not a multiple of 16 bytes: the 785-wide first layer) take the
//     same path with scalar loads.
//   * fragments are double buffered in registers: the reads of k-pair s + 1 are in flight under the MFMAs of s.
#pragma once
#include "cfm_common.h"

// PIPELINED K-step boundary (adopted in round 5: C3 model step 428 -> 417 us, cost matrix 266 -> 260 us on this engine,
// profiles/r5_experiments.txt; same bits).  A K step ends with a barrier and the next one would start with fragment reads
// nothing covers — the matrix pipe drains once per step.  The MFMAs of a step's LAST fragment group are therefore
// deferred: its fragments are in registers before the barrier, the MFMAs run right behind the next step's first fragment
// reads.  Per output the k order is unchanged.  (Tried and removed: alternating s_setprio of the two workgroups of a CU
// — 266 -> 275 us on the cost matrix; 128 x 64 layer tiles — model step 428 -> 501 us; earlier issue points of the
// global loads — no difference.)
typedef float gc_f32x16 __attribute__((ext_vector_type(16)));
typedef float gc_f32x2 __attribute__((ext_vector_type(2)));
typedef float gc_f32x4 __attribute__((ext_vector_type(4)));

template <int ROWS, bool KMAJOR> struct GcLd { static constexpr int value = KMAJOR ? ROWS + 4 : ROWS + 2; };

// One operand of a stage: ROWS x BK floats, element (row, k).  `src(row, k)` addressing:
//   KMAJOR = false: p[row * ld + k]       KMAJOR = true: p[k * ld + row]
// rows >= nrows and k >= kend read as zero.  VEC: 16-byte loads (preconditions at fetch()).
template <int ROWS, int BK, bool KMAJOR, bool VEC>
struct GcOperand {
    static constexpr int LD = GcLd<ROWS, KMAJOR>::value;
    static constexpr int NV = ROWS * BK / (256 * 4);      // float4 per thread and stage
    static constexpr int NS = ROWS * BK / 256;            // floats per thread and stage (scalar path)
    float4 v[VEC ? NV : 1];
    float s[VEC ? 1 : NS];
    unsigned okm;            // bit q: element q of this thread lies inside the operand (applied when the registers are
                             // consumed — stash — so that nothing waits for a load right after issuing it)

    // Loads are unconditional from a clamped (always valid) address and zeroed afterwards: no branches in the main
    // loop.  VEC preconditions (the launchers check them): K-contiguous: ld % 4 == 0, kend % 4 == 0, 16-byte aligned
    // base; K-major: ld % 4 == 0, nrows % 4 == 0, aligned base — so a vector never straddles the valid range.
    // Per-thread base pointers of the stage at k0 = 0 and the in-range mask of their rows, formed ONCE (bind): a stage
    // whose K range lies inside the operand then costs one 64-bit add of a uniform offset per 16-byte load.  Round 3
    // re-derived every address per stage — two v_mad_u64_u32, selects and compares per load, ~60 VALU instructions
    // per K step — and the timing probes of round 4 priced the global loads at 13 % of the asymptotic rate.
    const float* base[VEC ? NV : 1];
    unsigned rowm = 0u;
    bool bound = false;
    __device__ __forceinline__ void bind(const float* __restrict__ p, int ld, int row0, int nrows) {
        if (!VEC) return;
        const int tid = threadIdx.x;
        rowm = 0u;
#pragma unroll
        for (int q = 0; q < (VEC ? NV : 1); ++q) {
            int r, k;
            if (!KMAJOR) { k = 4 * (tid % (BK / 4)); r = tid / (BK / 4) + q * (1024 / BK); }
            else { r = 4 * (tid % (ROWS / 4)); k = tid / (ROWS / 4) + q * (1024 / ROWS); }
            const int gr = row0 + r;
            const bool ok = gr < nrows;
            const int cr = ok ? gr : 0;
            base[q] = KMAJOR ? p + (size_t)k * ld + cr : p + (size_t)cr * ld + k;
            rowm |= ok ? (1u << q) : 0u;
        }
        bound = true;
    }

    __device__ __forceinline__ void fetch(const float* __restrict__ p, int ld, int row0, int nrows, int k0, int kend) {
        const int tid = threadIdx.x;
        if (VEC && bound && k0 + BK <= kend) {                 // (uniform) the whole K range of the stage is inside
            const size_t koff = KMAJOR ? (size_t)k0 * (size_t)ld : (size_t)k0;
#pragma unroll
            for (int q = 0; q < (VEC ? NV : 1); ++q) v[q] = *reinterpret_cast<const float4*>(base[q] + koff);
            okm = rowm;
            return;
        }
        okm = 0u;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                int r, k;
                if (!KMAJOR) { k = 4 * (tid % (BK / 4)); r = tid / (BK / 4) + q * (1024 / BK); }
                else { r = 4 * (tid % (ROWS / 4)); k = tid / (ROWS / 4) + q * (1024 / ROWS); }
                const int gr = row0 + r, gk = k0 + k;
                const bool ok = gr < nrows && gk < kend;
                const int cr = ok ? gr : 0, ck = ok ? gk : 0;
                v[q] = *reinterpret_cast<const float4*>(KMAJOR ? p + (size_t)ck * ld + cr : p + (size_t)cr * ld + ck);
                okm |= ok ? (1u << q) : 0u;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                int r, k;
                if (!KMAJOR) { k = tid % BK; r = tid / BK + q * (256 / BK); }
                else { r = tid % ROWS; k = tid / ROWS + q * (256 / ROWS); }
                const int gr = row0 + r, gk = k0 + k;
                const bool ok = gr < nrows && gk < kend;
                const int cr = ok ? gr : 0, ck = ok ? gk : 0;
                s[q] = KMAJOR ? p[(size_t)ck * ld + cr] : p[(size_t)cr * ld + ck];
                okm |= ok ? (1u << q) : 0u;
            }
        }
    }

    __device__ __forceinline__ void stash(float* __restrict__ T) const {      // T: [BK][LD]
        const int tid = threadIdx.x;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                float4 x = v[q];
                if (!((okm >> q) & 1u)) x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!KMAJOR) {
                    const int k = 4 * (tid % (BK / 4)), r = tid / (BK / 4) + q * (1024 / BK);
                    T[(k + 0) * LD + r] = x.x; T[(k + 1) * LD + r] = x.y;
                    T[(k + 2) * LD + r] = x.z; T[(k + 3) * LD + r] = x.w;
                } else {
                    const int r = 4 * (tid % (ROWS / 4)), k = tid / (ROWS / 4) + q * (1024 / ROWS);
                    *reinterpret_cast<float4*>(T + k * LD + r) = x;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                int r, k;
                if (!KMAJOR) { k = tid % BK; r = tid / BK + q * (256 / BK); }
                else { r = tid % ROWS; k = tid / ROWS + q * (256 / ROWS); }
                float x = s[q];
                T[k * LD + r] = ((okm >> q) & 1u) ? x : 0.f;
            }
        }
    }
};

template <int BM, int BN, int BK, bool A_KMAJOR, bool B_KMAJOR, bool VEC_A, bool VEC_B>
struct GemmCore {
    static constexpr int MT = BM / 64, NT = BN / 64;                 // 32-row MFMA blocks per wave along M / N
    static constexpr int WM = BM / 2, WN = BN / 2;
    static constexpr int LDA = GcLd<BM, A_KMAJOR>::value, LDB = GcLd<BN, B_KMAJOR>::value;
    static constexpr int STAGE_A = BK * LDA, STAGE_B = BK * LDB;
    static constexpr int LDS_FLOATS = 2 * (STAGE_A + STAGE_B);
    static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "tile");
    static_assert(BK % 8 == 0 && (BM * BK) % 1024 == 0 && (BN * BK) % 1024 == 0, "stage");

    // 64 x 64 tiles (one 32 x 32 block per wave) run on v_mfma_f32_16x16x4_f32 instead: a 32 x 32 block is then FOUR
    // independent 16 x 16 accumulators in rotation.  With a single 32x32x2 accumulator every MFMA depends on the
    // previous one and anything issued between two of them stretches the chain (measured: MFMA-busy 37 %, 51 % of the
    // wave cycles waiting to issue); same k order, same bits.
    static constexpr bool M16 = (BM == 64 && BN == 64);
    static constexpr int EM = M16 ? 2 : MT, EU = M16 ? 2 : NT, ER = M16 ? 4 : 16;   // epilogue: at(m, u, r), m < EM, u < EU, r < ER

    gc_f32x16 acc[M16 ? 1 : MT][M16 ? 1 : NT];
    gc_f32x4 acc16[2][2];
    float pend_a[2][2], pend_b[2][2];       // fragments of the deferred group (two k-pairs / k-quads x two blocks)

    __device__ __forceinline__ void zero() {
        if (M16) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc16[a][b][r] = 0.f;
        } else {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
    }
    __device__ __forceinline__ float at(int m, int u, int r) const { return M16 ? acc16[m][u][r] : acc[M16 ? 0 : m][M16 ? 0 : u][r]; }

    // One K step: the MFMAs of the current stage with, woven between them IN SOURCE ORDER (LDS stores cannot be
    // moved across LDS loads by the scheduler, so the source order is the issue order), the stores of the next
    // stage's registers into the other LDS stage and the global loads of the stage after next.  A wave alone on its
    // SIMD (one 128 x 64 tile per CU at the C3 layer shapes) then spends the step issuing MFMAs back to back: the
    // stores / loads go out in the 64-cycle shadows of the matrix pipe.  Fragment reads run two k-pair groups ahead
    // (a group = two k-pairs = one ds_read2 per operand = 2 MT NT MFMAs).
    // PEND_IN: the deferred group of the PREVIOUS step is computed first, behind this step's first fragment reads;
    // PEND_OUT: this step's last group is deferred (its fragments go to pend_a / pend_b).  Both false: the plain step.
    template <bool NEXT, bool PEND_IN, bool PEND_OUT, typename OA, typename OB, typename FetchA, typename FetchB>
    __device__ __forceinline__ void step(const float* __restrict__ As, const float* __restrict__ Bs,
                                         float* __restrict__ An, float* __restrict__ Bn, OA& oa, OB& ob, FetchA fetch_a,
                                         FetchB fetch_b) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int wm = wv >> 1, wn = wv & 1;
        if (M16) {
            // fragment of a 16x16x4 MFMA: lane -> (row = lane & 15, k = lane >> 4); the wave's 32 rows are interleaved
            // over the two row blocks (block t owns local rows 2 j + t): one ds_read_b64 per operand and k-quad
            const int fr = lane & 15, fq = lane >> 4;
            const float* pa = As + fq * LDA + wm * WM + 2 * fr;
            const float* pb = Bs + fq * LDB + wn * WN + 2 * fr;
            constexpr int NQ = BK / 4, NG = NQ / 2;
            float a[NQ][2], b[NQ][2];
            auto rd = [&](int q) {
                const gc_f32x2 ta = *reinterpret_cast<const gc_f32x2*>(pa + 4 * q * LDA); a[q][0] = ta.x; a[q][1] = ta.y;
                const gc_f32x2 tb = *reinterpret_cast<const gc_f32x2*>(pb + 4 * q * LDB); b[q][0] = tb.x; b[q][1] = tb.y;
            };
            auto mm = [&](int q) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
                        acc16[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][m], b[q][nn], acc16[m][nn], 0, 0, 0);
            };
            rd(0); rd(1);
            if (NG > 1) { rd(2); rd(3); }
            __builtin_amdgcn_sched_barrier(0);
            if (PEND_IN) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn)
                            acc16[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(pend_a[h][m], pend_b[h][nn], acc16[m][nn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool defer = PEND_OUT && g == NG - 1;
                if (!defer) mm(2 * g);
                if (NEXT && g == 0) oa.stash(An);
                if (NEXT && g == 1) ob.stash(Bn);
                if (NEXT && NG == 1 && g == 0) ob.stash(Bn);
                if (NEXT && g == (NG > 2 ? 2 : NG - 1)) { fetch_a(); fetch_b(); }       // the stage after next: behind the third group
                if (!defer) mm(2 * g + 1);
                else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) { pend_a[h][0] = a[2 * g + h][0]; pend_a[h][1] = a[2 * g + h][1]; pend_b[h][0] = b[2 * g + h][0]; pend_b[h][1] = b[2 * g + h][1]; }
                }
                if (g + 2 < NG) { rd(2 * g + 4); rd(2 * g + 5); }
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        const int fr = lane & 31, fk = lane >> 5;
        const float* pa = As + fk * LDA + wm * WM + (MT == 2 ? 2 * fr : fr);
        const float* pb = Bs + fk * LDB + wn * WN + (NT == 2 ? 2 * fr : fr);
        constexpr int NP = BK / 2, NG = NP / 2;
        float a[NP][2], b[NP][2];
        auto rd = [&](int q) {
            if (MT == 2) { const gc_f32x2 t = *reinterpret_cast<const gc_f32x2*>(pa + 2 * q * LDA); a[q][0] = t.x; a[q][1] = t.y; }
            else { a[q][0] = pa[2 * q * LDA]; a[q][1] = 0.f; }
            if (NT == 2) { const gc_f32x2 t = *reinterpret_cast<const gc_f32x2*>(pb + 2 * q * LDB); b[q][0] = t.x; b[q][1] = t.y; }
            else { b[q][0] = pb[2 * q * LDB]; b[q][1] = 0.f; }
        };
        auto mm = [&](int q) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][m], b[q][nn], acc[m][nn], 0, 0, 0);
        };
        rd(0); rd(1);
        if (NG > 1) { rd(2); rd(3); }
        __builtin_amdgcn_sched_barrier(0);
        if (PEND_IN) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn)
                        acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(pend_a[h][m], pend_b[h][nn], acc[m][nn], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (PEND_OUT && g == NG - 1) {       // deferred: the fragments wait in registers for the next step (nothing else is scheduled in the last group when NG >= 4)
#pragma unroll
                for (int h = 0; h < 2; ++h) { pend_a[h][0] = a[2 * g + h][0]; pend_a[h][1] = a[2 * g + h][1]; pend_b[h][0] = b[2 * g + h][0]; pend_b[h][1] = b[2 * g + h][1]; }
                if (NEXT && g == 0) oa.stash(An);
                if (NEXT && g == 1) ob.stash(Bn);
                if (NEXT && NG == 1 && g == 0) ob.stash(Bn);
                if (NEXT && g == (NG > 2 ? 2 : NG - 1)) { fetch_a(); fetch_b(); }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            mm(2 * g);
            if (NEXT && g == 0) oa.stash(An);
            if (NEXT && g == 1) ob.stash(Bn);
            if (NEXT && NG == 1 && g == 0) ob.stash(Bn);
            if (NEXT && g == (NG > 2 ? 2 : NG - 1)) { fetch_a(); fetch_b(); }           // the stage after next: behind the third group
            mm(2 * g + 1);
            if (g + 2 < NG) { rd(2 * g + 4); rd(2 * g + 5); }
            __builtin_amdgcn_sched_barrier(0);       // the reads of group g + 2 stay in front of the MFMAs of group g + 1
        }
    }

    // Main loop over [k_begin, k_end).  post(As_stage, k0): hook run once per stage after its barrier, before the MFMAs
    // (wgrad's bias sums).
    template <typename Post>
    __device__ __forceinline__ void run(float* __restrict__ lds, const float* __restrict__ A, int lda, int row0, int M,
                                        const float* __restrict__ B, int ldb, int col0, int N, int k_begin, int k_end,
                                        Post post) {
        GcOperand<BM, BK, A_KMAJOR, VEC_A> oa;
        GcOperand<BN, BK, B_KMAJOR, VEC_B> ob;
        float* As = lds; float* Bs = lds + 2 * STAGE_A;
        if (k_begin >= k_end) return;
        oa.bind(A, lda, row0, M); ob.bind(B, ldb, col0, N);
        oa.fetch(A, lda, row0, M, k_begin, k_end); ob.fetch(B, ldb, col0, N, k_begin, k_end);
        oa.stash(As); ob.stash(Bs);
        __syncthreads();
        int st = 0;
        if (k_begin + BK < k_end) { oa.fetch(A, lda, row0, M, k_begin + BK, k_end); ob.fetch(B, ldb, col0, N, k_begin + BK, k_end); }
        // ONE body in the loop, the last K step peeled behind it: with both step<> forms inside the loop (round 3) the
        // accumulators were loop-carried through a phi the register allocator resolved with a full copy of the
        // accumulator file on entry AND exit of every K step (64 + 64 v_accvgpr moves per step at 128 x 128:
        // tools/isa_report.py), a quarter of the step's issue slots with the matrix pipe idle behind them.
        int k0 = k_begin;
        // the deferred group needs a fragment group to itself (NG >= 2 groups per step: every tile shape in use)
        constexpr bool PIPE_BODY = ((M16 ? BK / 8 : BK / 4) >= 2);
        if (PIPE_BODY) {
            if (k0 + BK < k_end) {                   // the first step: nothing deferred comes in, its last group goes out
                float* Ac = As + st * STAGE_A; float* Bc = Bs + st * STAGE_B;
                post(Ac, k0);
                step<true, false, true>(Ac, Bc, As + (st ^ 1) * STAGE_A, Bs + (st ^ 1) * STAGE_B, oa, ob,
                           [&]() { if (k0 + 2 * BK < k_end) { oa.fetch(A, lda, row0, M, k0 + 2 * BK, k_end); } },
                           [&]() { if (k0 + 2 * BK < k_end) { ob.fetch(B, ldb, col0, N, k0 + 2 * BK, k_end); } });
                __syncthreads();
                st ^= 1; k0 += BK;
            } else {                                 // a single step: the deferred group that comes in is all zeros (adds + 0 to + 0)
#pragma unroll
                for (int h = 0; h < 2; ++h) { pend_a[h][0] = pend_a[h][1] = 0.f; pend_b[h][0] = pend_b[h][1] = 0.f; }
            }
        }
        for (; k0 + BK < k_end; k0 += BK) {
            float* Ac = As + st * STAGE_A; float* Bc = Bs + st * STAGE_B;
            post(Ac, k0);
            step<true, PIPE_BODY, PIPE_BODY>(Ac, Bc, As + (st ^ 1) * STAGE_A, Bs + (st ^ 1) * STAGE_B, oa, ob,
                       [&]() { if (k0 + 2 * BK < k_end) { oa.fetch(A, lda, row0, M, k0 + 2 * BK, k_end); } },
                       [&]() { if (k0 + 2 * BK < k_end) { ob.fetch(B, ldb, col0, N, k0 + 2 * BK, k_end); } });
            __syncthreads();
            st ^= 1;
        }
        {
            float* Ac = As + st * STAGE_A; float* Bc = Bs + st * STAGE_B;
            post(Ac, k0);
            step<false, PIPE_BODY, false>(Ac, Bc, nullptr, nullptr, oa, ob, []() {}, []() {});
        }
    }

    // Epilogue geometry.  C/D layout of the MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // A lane owns local columns col_lo() (+ 1 when NT == 2: acc[m][0][r], acc[m][1][r] are ADJACENT columns, one
    // 8-byte store) and, for accumulator register r of row block m, local row row_of(m, r).
    // (16x16x4 mode: col = lane & 15, row = 4 (lane >> 4) + r, both interleaved over the two blocks.)
    __device__ __forceinline__ static int col_lo() {
        const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
        if (M16) return wn * WN + 2 * (lane & 15);
        return wn * WN + (NT == 2 ? 2 * (lane & 31) : (lane & 31));
    }
    __device__ __forceinline__ static int row_of(int m, int r) {
        const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
        if (M16) return wm * WM + 2 * (4 * (lane >> 4) + r) + m;
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        return wm * WM + (MT == 2 ? 2 * rho + m : rho);
    }
};

struct GcNoPost { __device__ __forceinline__ void operator()(const float*, int) const {} };
