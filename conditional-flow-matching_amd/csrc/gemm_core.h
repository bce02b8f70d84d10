// gemm_core.h — the fp32-MFMA tile engine shared by every dense product of the path (gfx950):
//   mlp_layer (MLP forward, mlp.hip), gemm_f32_mfma (dgrad / wgrad, mlp_train.hip), cost_gemm (cost.hip).
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

typedef float gc_f32x16 __attribute__((ext_vector_type(16)));
typedef float gc_f32x2 __attribute__((ext_vector_type(2)));

template <int ROWS, bool KMAJOR> struct GcLd { static constexpr int value = KMAJOR ? ROWS + 4 : ROWS + 2; };

// One operand of a stage: ROWS x BK floats, element (row, k).  `src(row, k)` addressing:
//   KMAJOR = false: p[row * ld + k]       KMAJOR = true: p[k * ld + row]
// rows >= nrows and k >= kend read as zero.  VEC: 16-byte loads (ld % 4 == 0, base 16-byte aligned, and the
// vector never straddles the valid range because nrows / kend are then multiples of 4 or the tail is masked per
// element below).
template <int ROWS, int BK, bool KMAJOR, bool VEC>
struct GcOperand {
    static constexpr int LD = GcLd<ROWS, KMAJOR>::value;
    static constexpr int NV = ROWS * BK / (256 * 4);      // float4 per thread and stage
    static constexpr int NS = ROWS * BK / 256;            // floats per thread and stage (scalar path)
    float4 v[VEC ? NV : 1];
    float s[VEC ? 1 : NS];

    __device__ __forceinline__ void fetch(const float* __restrict__ p, int ld, int row0, int nrows, int k0, int kend) {
        const int tid = threadIdx.x;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                int r, k;
                if (!KMAJOR) { k = 4 * (tid % (BK / 4)); r = tid / (BK / 4) + q * (1024 / BK); }
                else { r = 4 * (tid % (ROWS / 4)); k = tid / (ROWS / 4) + q * (1024 / ROWS); }
                const int gr = row0 + r, gk = k0 + k;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!KMAJOR) {
                    if (gr < nrows && gk < kend) {
                        const float* a = p + (size_t)gr * ld + gk;
                        if (gk + 3 < kend) x = *reinterpret_cast<const float4*>(a);
                        else { x.x = a[0]; if (gk + 1 < kend) x.y = a[1]; if (gk + 2 < kend) x.z = a[2]; }
                    }
                } else {
                    if (gk < kend && gr < nrows) {
                        const float* a = p + (size_t)gk * ld + gr;
                        if (gr + 3 < nrows) x = *reinterpret_cast<const float4*>(a);
                        else { x.x = a[0]; if (gr + 1 < nrows) x.y = a[1]; if (gr + 2 < nrows) x.z = a[2]; }
                    }
                }
                v[q] = x;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                int r, k;
                if (!KMAJOR) { k = tid % BK; r = tid / BK + q * (256 / BK); }
                else { r = tid % ROWS; k = tid / ROWS + q * (256 / ROWS); }
                const int gr = row0 + r, gk = k0 + k;
                s[q] = (gr < nrows && gk < kend) ? (KMAJOR ? p[(size_t)gk * ld + gr] : p[(size_t)gr * ld + gk]) : 0.f;
            }
        }
    }

    // optional per-k offset (cost_gemm subtracts the common centre mu[k] on the way in)
    __device__ __forceinline__ void sub_k(const float* __restrict__ mu, int k0, int kend) {
        const int tid = threadIdx.x;
        if (VEC) {
            static_assert(!KMAJOR, "sub_k: K-contiguous operands only");
            const int gk = k0 + 4 * (tid % (BK / 4));
            float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < kend) { m.x = mu[gk]; if (gk + 1 < kend) m.y = mu[gk + 1]; if (gk + 2 < kend) m.z = mu[gk + 2]; if (gk + 3 < kend) m.w = mu[gk + 3]; }
#pragma unroll
            for (int q = 0; q < NV; ++q) { v[q].x -= m.x; v[q].y -= m.y; v[q].z -= m.z; v[q].w -= m.w; }
        } else {
            const int gk = k0 + tid % BK;
            const float m = gk < kend ? mu[gk] : 0.f;
#pragma unroll
            for (int q = 0; q < NS; ++q) s[q] -= m;
        }
    }

    __device__ __forceinline__ void stash(float* __restrict__ T) const {      // T: [BK][LD]
        const int tid = threadIdx.x;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (!KMAJOR) {
                    const int k = 4 * (tid % (BK / 4)), r = tid / (BK / 4) + q * (1024 / BK);
                    T[(k + 0) * LD + r] = v[q].x; T[(k + 1) * LD + r] = v[q].y;
                    T[(k + 2) * LD + r] = v[q].z; T[(k + 3) * LD + r] = v[q].w;
                } else {
                    const int r = 4 * (tid % (ROWS / 4)), k = tid / (ROWS / 4) + q * (1024 / ROWS);
                    *reinterpret_cast<float4*>(T + k * LD + r) = v[q];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                int r, k;
                if (!KMAJOR) { k = tid % BK; r = tid / BK + q * (256 / BK); }
                else { r = tid % ROWS; k = tid / ROWS + q * (256 / ROWS); }
                T[k * LD + r] = s[q];
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

    gc_f32x16 acc[MT][NT];

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

    // the MFMAs of one stage
    __device__ __forceinline__ void compute(const float* __restrict__ As, const float* __restrict__ Bs) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int wm = wv >> 1, wn = wv & 1;
        const int fr = lane & 31, fk = lane >> 5;
        const float* pa = As + fk * LDA + wm * WM + (MT == 2 ? 2 * fr : fr);
        const float* pb = Bs + fk * LDB + wn * WN + (NT == 2 ? 2 * fr : fr);
        float a[2][2], b[2][2];
        auto rd = [&](int kk, int buf) {
            if (MT == 2) { const gc_f32x2 t = *reinterpret_cast<const gc_f32x2*>(pa + kk * LDA); a[buf][0] = t.x; a[buf][1] = t.y; }
            else a[buf][0] = pa[kk * LDA];
            if (NT == 2) { const gc_f32x2 t = *reinterpret_cast<const gc_f32x2*>(pb + kk * LDB); b[buf][0] = t.x; b[buf][1] = t.y; }
            else b[buf][0] = pb[kk * LDB];
        };
        rd(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int cur = (kk >> 1) & 1;
            if (kk + 2 < BK) rd(kk + 2, cur ^ 1);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m], b[cur][nn], acc[m][nn], 0, 0, 0);
        }
    }

    // Main loop over [k_begin, k_end).  pre(opA, opB, k0): hook applied to freshly fetched registers (cost_gemm's
    // centring); post(As_stage): hook run once per stage after its barrier, before the MFMAs (wgrad's bias sums).
    template <typename Pre, typename Post>
    __device__ __forceinline__ void run(float* __restrict__ lds, const float* __restrict__ A, int lda, int row0, int M,
                                        const float* __restrict__ B, int ldb, int col0, int N, int k_begin, int k_end,
                                        Pre pre, Post post) {
        GcOperand<BM, BK, A_KMAJOR, VEC_A> oa;
        GcOperand<BN, BK, B_KMAJOR, VEC_B> ob;
        float* As = lds; float* Bs = lds + 2 * STAGE_A;
        if (k_begin >= k_end) return;
        oa.fetch(A, lda, row0, M, k_begin, k_end); ob.fetch(B, ldb, col0, N, k_begin, k_end);
        pre(oa, ob, k_begin);
        oa.stash(As); ob.stash(Bs);
        __syncthreads();
        int st = 0;
        const bool more = k_begin + BK < k_end;
        if (more) { oa.fetch(A, lda, row0, M, k_begin + BK, k_end); ob.fetch(B, ldb, col0, N, k_begin + BK, k_end); pre(oa, ob, k_begin + BK); }
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            post(As + st * STAGE_A);
            compute(As + st * STAGE_A, Bs + st * STAGE_B);
            if (k0 + BK < k_end) {
                oa.stash(As + (st ^ 1) * STAGE_A); ob.stash(Bs + (st ^ 1) * STAGE_B);
                __syncthreads();
                if (k0 + 2 * BK < k_end) {
                    oa.fetch(A, lda, row0, M, k0 + 2 * BK, k_end); ob.fetch(B, ldb, col0, N, k0 + 2 * BK, k_end);
                    pre(oa, ob, k0 + 2 * BK);
                }
                st ^= 1;
            }
        }
    }

    // Epilogue geometry.  C/D layout of the MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // A lane owns local columns col_lo() (+ 1 when NT == 2: acc[m][0][r], acc[m][1][r] are ADJACENT columns, one
    // 8-byte store) and, for accumulator register r of row block m, local row row_of(m, r).
    __device__ __forceinline__ static int col_lo() {
        const int lane = threadIdx.x & 63, wn = (threadIdx.x >> 6) & 1;
        return wn * WN + (NT == 2 ? 2 * (lane & 31) : (lane & 31));
    }
    __device__ __forceinline__ static int row_of(int m, int r) {
        const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        return wm * WM + (MT == 2 ? 2 * rho + m : rho);
    }
};

struct GcNoPre { template <typename OA, typename OB> __device__ __forceinline__ void operator()(OA&, OB&, int) const {} };
struct GcNoPost { __device__ __forceinline__ void operator()(const float*) const {} };
