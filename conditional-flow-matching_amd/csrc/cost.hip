// cost.hip — K1: squared-Euclidean cost matrix for gfx950.
//
// Replaces torch.cdist(x0, x1) ** 2 (torchcfm/optimal_transport.py:84).
// Direct-difference form  M[i,j] = sum_k (x0[i,k] - x1[j,k])^2  in fp32: no
// sqrt/square round trip and no ||x||^2+||y||^2-2xy cancellation — and, for d >= 64 (with
// scratch: cfm_sqeuclid_cost_ws_f32), the centred Gram form on the matrix cores with every
// cancelling entry recomputed in the direct form.
//
// Three kernels:
//   cost_gemm_glds : d >= 64, B0, B1 >= 256.  v_mfma_f32_32x32x2_f32, 128x128 tile per workgroup, operands DMA'd
//                  straight into LDS (gemm_glds.h); measured max relative error 2-5e-7 against fp64 (direct
//                  form: 0.5-2e-6), 247 us at B = 4096, d = 784 (direct: 574 us).  See the section below.
//   cost_small_d : d <= 8.  Output-write bound (4*B0*B1 bytes): each lane owns
//                  four consecutive columns whose x1 rows live in registers,
//                  loops over a strip of rows (x0 row is wave-uniform) and
//                  stores one float4 per row -> 1 KiB contiguous per wave store.
//   cost_tiled   : general d.  VALU bound (3*B0*B1*d flop).  128x128 output
//                  tile per 256-thread workgroup, 8x8 micro-tile per lane,
//                  k-major LDS tiles (conflict-free ds_write_b32 on the
//                  transposing store, conflict-free ds_read_b128 on the read),
//                  8x8 super-tiles of workgroups per XCD so both operand panels
//                  stay in that XCD's 4 MiB L2.
#include "cfm_common.h"
#include "gemm_glds.h"
#include <stdlib.h>

// ---------------------------------------------------------------- small d ----
template <int D>
__global__ __launch_bounds__(256) void cost_small_d(const float* __restrict__ x0,
                                                    const float* __restrict__ x1, int B0,
                                                    int B1, float* __restrict__ M,
                                                    int rows_per_block) {
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int i_beg = blockIdx.y * rows_per_block;
    const int i_end = min(B0, i_beg + rows_per_block);
    float y[4][D];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < D; ++k) y[c][k] = (j0 + c < B1) ? x1[(size_t)(j0 + c) * D + k] : 0.f;
    if (j0 >= B1) return;
    const bool full = (j0 + 3 < B1) && ((B1 & 3) == 0);
    for (int i = i_beg; i < i_end; ++i) {
        float xr[D];
#pragma unroll
        for (int k = 0; k < D; ++k) xr[k] = x0[(size_t)i * D + k];  // wave-uniform address
        float acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                float df = xr[k] - y[c][k];
                s = fmaf(df, df, s);
            }
            acc[c] = s;
        }
        float* dst = M + (size_t)i * B1 + j0;
        if (full) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j0 + c < B1) dst[c] = acc[c];
        }
    }
}

// ------------------------------------------------------------------ tiled ----
// BM = BN = 16*TM.  Lane (tx,ty) owns rows {ty*4..+3} + h*(BM/2) and columns
// {tx*4..+3} + h*(BN/2) (h = 0..TM/4-1): every ds_read_b128 of a wave touches
// one contiguous 256-byte bank row (B operand) or 4 broadcast slots (A operand),
// and every output store of 16 lanes is 256 contiguous bytes.
template <int TM, bool VEC>
__global__ __launch_bounds__(256) void cost_tiled(const float* __restrict__ x0,
                                                  const float* __restrict__ x1, int B0, int B1,
                                                  int d, float* __restrict__ M, int tiles_m,
                                                  int tiles_n) {
    constexpr int BM = 16 * TM, BN = 16 * TM, BK = 32, LD = BM + 4, H = TM / 4;
    __shared__ __attribute__((aligned(16))) float As[BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LD];

    // XCD-aware super-tile order: remap so each XCD walks 8x8 groups of tiles.
    unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    {
        const int G = 8;
        const int per_band = G * tiles_n;        // tiles in a full band of G tile-rows
        const int band = lid / per_band, r = lid - band * per_band;
        const int rows_in_band = min(G, tiles_m - band * G);
        const int fgt = rows_in_band * G;        // tiles in a full-width group of this band
        const int gcol = r / fgt;                // groups left of the last one are full width
        const int rr = r - gcol * fgt;
        const int cols_in_group = min(G, tiles_n - gcol * G);
        tm = band * G + rr / cols_in_group;
        tn = gcol * G + rr % cols_in_group;
    }
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, wv = tid >> 6;

    float acc[TM][TM];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = 0.f;

    for (int k0 = 0; k0 < d; k0 += BK) {
        // ---- stage: lane -> row (conflict-free transposing LDS store) ----
        // pass p covers rows lane + 64*p (p < BM/64) and k chunk (wv + 4*q)*4, q = 0..1
#pragma unroll
        for (int p = 0; p < BM / 64; ++p) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = lane + 64 * p, kc = (wv + 4 * q) * 4;
                const int ga = row0 + r, gb = col0 + r, gk = k0 + kc;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                if (VEC) {
                    if (ga < B0 && gk < d) va = *reinterpret_cast<const float4*>(x0 + (size_t)ga * d + gk);
                    if (gb < B1 && gk < d) vb = *reinterpret_cast<const float4*>(x1 + (size_t)gb * d + gk);
                } else {
                    if (ga < B0) {
                        const float* s = x0 + (size_t)ga * d;
                        if (gk + 0 < d) va.x = s[gk + 0];
                        if (gk + 1 < d) va.y = s[gk + 1];
                        if (gk + 2 < d) va.z = s[gk + 2];
                        if (gk + 3 < d) va.w = s[gk + 3];
                    }
                    if (gb < B1) {
                        const float* s = x1 + (size_t)gb * d;
                        if (gk + 0 < d) vb.x = s[gk + 0];
                        if (gk + 1 < d) vb.y = s[gk + 1];
                        if (gk + 2 < d) vb.z = s[gk + 2];
                        if (gk + 3 < d) vb.w = s[gk + 3];
                    }
                }
                As[(kc + 0) * LD + r] = va.x; As[(kc + 1) * LD + r] = va.y;
                As[(kc + 2) * LD + r] = va.z; As[(kc + 3) * LD + r] = va.w;
                Bs[(kc + 0) * LD + r] = vb.x; Bs[(kc + 1) * LD + r] = vb.y;
                Bs[(kc + 2) * LD + r] = vb.z; Bs[(kc + 3) * LD + r] = vb.w;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < BK; ++k) {
            float a[TM], b[TM];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                float4 t4 = *reinterpret_cast<const float4*>(&As[k * LD + h * (BM / 2) + ty * 4]);
                a[h * 4 + 0] = t4.x; a[h * 4 + 1] = t4.y; a[h * 4 + 2] = t4.z; a[h * 4 + 3] = t4.w;
                float4 u4 = *reinterpret_cast<const float4*>(&Bs[k * LD + h * (BN / 2) + tx * 4]);
                b[h * 4 + 0] = u4.x; b[h * 4 + 1] = u4.y; b[h * 4 + 2] = u4.z; b[h * 4 + 3] = u4.w;
            }
#pragma unroll
            for (int ia = 0; ia < TM; ++ia)
#pragma unroll
                for (int ib = 0; ib < TM; ++ib) {
                    float df = a[ia] - b[ib];
                    acc[ia][ib] = fmaf(df, df, acc[ia][ib]);
                }
        }
        __syncthreads();
    }
    // ---- epilogue: float4 stores, 256 contiguous bytes per 16 lanes ----
    const bool vec_out = ((B1 & 3) == 0);
#pragma unroll
    for (int ia = 0; ia < TM; ++ia) {
        const int gi = row0 + (ia >> 2) * (BM / 2) + ty * 4 + (ia & 3);
        if (gi >= B0) continue;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int gj = col0 + h * (BN / 2) + tx * 4;
            float* dst = M + (size_t)gi * B1 + gj;
            if (vec_out && gj + 3 < B1) {
                *reinterpret_cast<float4*>(dst) = make_float4(acc[ia][h * 4 + 0], acc[ia][h * 4 + 1],
                                                              acc[ia][h * 4 + 2], acc[ia][h * 4 + 3]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (gj + c < B1) dst[c] = acc[ia][h * 4 + c];
            }
        }
    }
}

// ------------------------------------------------------------------- MFMA ----
// d >= 64: the Gram form on the matrix cores.  With every point centred on mu (the mean of 128
// strided sample rows — any mu is valid, a central one keeps the norms at the scale of the
// distances),  a_i = fl(x0_i - mu),  b_j = fl(x1_j - mu):
//     M_ij = |a_i|^2 + |b_j|^2 - 2 <a_i, b_j>,        <a, b> on v_mfma_f32_32x32x2_f32
// (exact fp32 products, fp32 accumulation).  The form cancels when a pair is much closer than the
// cloud is wide, so every entry that comes out below 1/8 of |a_i|^2 + |b_j|^2 is recomputed in
// the direct-difference form by its wave (64 lanes split the k loop): the relative error of what
// is kept from the Gram form stays within 8x the direct form's, duplicates and x-vs-x diagonals
// get the direct value.  2 flop per (i, j, k) at the 157 TFLOP/s matrix peak against 3 on the
// packed vector pipes: 0.17 ms vs 0.33 ms floor at B = 4096, d = 784.
typedef float cost_f32x16 __attribute__((ext_vector_type(16)));

// (round 4: both kernels request their loads ahead of the dependent adds — the sums run in the same order as before,
//  so mu and the norms keep their bits: 25 + 20 us of dependent-latency chains stood in front of every cost matrix;
//  tools/gemm_quick.py, scratch/gemm_kscan.py: time(d) = 58 us + 0.29 us x d before the change)
__global__ __launch_bounds__(256) void cost_center(const float* __restrict__ x0, const float* __restrict__ x1,
                                                   int B0, int B1, int d, float* __restrict__ mu) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int n0 = min(B0, 64), n1 = min(B1, 64);
    float s = 0.f;
    if (c < d) {
        // sample rows g, g + 4, ...: at most 16 per cloud and thread, all in flight together
        float v0[16], v1[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int q = g + 4 * t;
            v0[t] = (q < n0) ? x0[(size_t)((long long)q * B0 / n0) * d + c] : 0.f;
            v1[t] = (q < n1) ? x1[(size_t)((long long)q * B1 / n1) * d + c] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) if (g + 4 * t < n0) s += v0[t];
#pragma unroll
        for (int t = 0; t < 16; ++t) if (g + 4 * t < n1) s += v1[t];
    }
    part[g][lane] = s;
    __syncthreads();
    if (g == 0 && c < d) mu[c] = (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]) / (float)(n0 + n1);
}

// nrm[i] = |fl(x0_i - mu)|^2 (i < B0), nrm[B0 + j] = |fl(x1_j - mu)|^2: one wave per row; and the centred rows
// fl(x - mu) themselves in the PADDED layout the direct-to-LDS product wants (gemm_glds.h): xc0 [(B0 + 1)][dp],
// xc1 [(B1 + 1)][dp] behind it, dp = d rounded up to GL_BK, the k padding and the extra row of each zero — so that every
// DMA of the product is unconditional (any d, any alignment of the clouds).
__global__ __launch_bounds__(256) void cost_norms(const float* __restrict__ x0, const float* __restrict__ x1,
                                                  int B0, int B1, int d, const float* __restrict__ mu,
                                                  float* __restrict__ nrm, float* __restrict__ xc) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int dp = (d + GL_BK - 1) / GL_BK * GL_BK;
    if (row >= B0 + B1 && row < B0 + B1 + 2) {                // the two zero rows
        float* z = xc + (size_t)(row == B0 + B1 ? B0 : B0 + 1 + B1) * dp;
        for (int k = lane; k < dp; k += 64) z[k] = 0.f;
    }
    if (row >= B0 + B1) return;
    const float* p = row < B0 ? x0 + (size_t)row * d : x1 + (size_t)(row - B0) * d;
    float* pc = xc + (size_t)(row < B0 ? row : row + 1) * dp;
    for (int k = d + lane; k < dp; k += 64) pc[k] = 0.f;
    float s = 0.f;
    int k = lane;
    for (; k + 64 * 7 < d; k += 64 * 8) {                 // 8 trips' loads in flight, the chain in k order as before
        float xv[8], mv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { xv[t] = p[k + 64 * t]; mv[t] = mu[k + 64 * t]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) { const float tt = xv[t] - mv[t]; s = fmaf(tt, tt, s); pc[k + 64 * t] = tt; }
    }
    for (; k < d; k += 64) {
        const float t = p[k] - mu[k];
        s = fmaf(t, t, s);
        pc[k] = t;
    }
    s = wave_sum_f(s);
    if (lane == 0) nrm[row] = s;
}

// 128 x 128 output tile per workgroup on the direct-to-LDS engine (gemm_glds.h): operands = the centred, padded clouds
// cost_norms wrote (xc0, xc1).  Epilogue: |a|^2 + |b|^2 - 2 a.b, clamped; entries that cancel (below 1/8 of the norm
// sum: duplicated / nearly identical points) are recomputed in the direct form by their wave from the ORIGINAL clouds
// x0 / x1.  64 KiB of dynamic LDS, two workgroups per CU.  C3 (4096 x 4096 x 784): 247 us = 106 TFLOP/s (round 4's
// register-staged engine: 266 us; profiles/r5_experiments.txt).
__global__ __launch_bounds__(256, 2) void cost_gemm_glds(const float* __restrict__ xc0, const float* __restrict__ xc1,
                                                         const float* __restrict__ x0, const float* __restrict__ x1,
                                                         int B0, int B1, int d, const float* __restrict__ nrm,
                                                         float* __restrict__ M, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float glds_lds[];
    unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
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
    const int tid = threadIdx.x, lane = tid & 63;
    GldsCore g;
    g.zero();
    {
        const int dp = (d + GL_BK - 1) / GL_BK * GL_BK;      // xc0 / xc1 are padded: [B + 1][dp]
        gl_run_padded_pipe(g, glds_lds, xc0, dp, row0, B0, xc1, dp, col0, B1, dp);
    }
    gl_wait_barrier();                                   // every wave is done with the last stage
    float* An = glds_lds; float* Bn = glds_lds + GL_BM;
    if (tid < GL_BM) An[tid] = (row0 + tid < B0) ? nrm[row0 + tid] : 0.f;
    if (tid < GL_BN) Bn[tid] = (col0 + tid < B1) ? nrm[B0 + col0 + tid] : 0.f;
    __syncthreads();
    const float ny[2] = {Bn[GldsCore::col_of(0)], Bn[GldsCore::col_of(1)]};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        float v[16][2];
        bool anybad = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = GldsCore::row_of(m, r);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float sum = An[rl] + ny[u];
                const float x = fmaxf(fmaf(-2.f, g.acc[m][u][r], sum), 0.f);
                anybad |= (row0 + rl < B0 && col0 + GldsCore::col_of(u) < B1 && x < 0.125f * sum);
                v[r][u] = x;
            }
        }
        if (__ballot(anybad) != 0ull) {                       // wave uniform, rare: recompute the cancelling entries directly
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = row0 + GldsCore::row_of(m, r);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int gc = col0 + GldsCore::col_of(u);
                    const float sum = An[GldsCore::row_of(m, r)] + ny[u];
                    unsigned long long mask = __ballot(gr < B0 && gc < B1 && v[r][u] < 0.125f * sum);
                    while (mask) {
                        const int l = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int gi = __shfl(gr, l, 64), gj = __shfl(gc, l, 64);
                        const float* pa = x0 + (size_t)gi * d;
                        const float* pb = x1 + (size_t)gj * d;
                        float p = 0.f;
                        for (int k = lane; k < d; k += 64) {
                            const float t = pa[k] - pb[k];
                            p = fmaf(t, t, p);
                        }
                        p = wave_sum_f(p);
                        if (lane == l) v[r][u] = p;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + GldsCore::row_of(m, r);
            if (gr < B0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int gc = col0 + GldsCore::col_of(u);
                    if (gc < B1) M[(size_t)gr * B1 + gc] = v[r][u];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void max_reduce_f32(const float* __restrict__ M, size_t n,
                                                      unsigned* __restrict__ out_bits) {
    float m = 0.f;  // costs are >= 0
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        m = fmaxf(m, M[i]);
    m = wave_max_f(m);
    __shared__ float s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        atomicMax(out_bits, __float_as_uint(m));  // non-negative floats order like uints
    }
}

__global__ __launch_bounds__(256) void scale_inv_f32(float* __restrict__ M, size_t n,
                                                     const float* __restrict__ maxval) {
    const float mx = *maxval;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        M[i] = M[i] / mx;  // IEEE division, same as torch's M / M.max()
}

__global__ __launch_bounds__(256) void sqrt_inplace_f32(float* __restrict__ M, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        M[i] = sqrtf(M[i]);
}

// ------------------------------------------------------------------- C ABI ---
template <int D>
static void launch_small(const float* x0, const float* x1, int B0, int B1, float* M,
                         hipStream_t st) {
    const int rows_per_block = 16;
    dim3 grid((B1 + 1023) / 1024, (B0 + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(cost_small_d<D>, grid, dim3(256), 0, st, x0, x1, B0, B1, M, rows_per_block);
}

// Gram form on the matrix cores for d >= 64 and at least a 2 x 2 grid of tiles; 0 = not taken.
// ws: mu [d padded to 64] | norms [B0 + B1, padded to 4] | centred, padded clouds [(B0 + 1) + (B1 + 1)][dp]
static inline size_t cost_ws_head_floats(int B0, int B1, int d) {
    return (size_t)((d + 63) & ~63) + (((size_t)B0 + (size_t)B1 + 3) & ~(size_t)3);
}
static inline size_t cost_dp(int d) { return (size_t)((d + GL_BK - 1) / GL_BK * GL_BK); }
extern "C" size_t cfm_cost_ws_bytes_internal(int B0, int B1, int d) {
    const size_t centred = ((size_t)B0 + (size_t)B1 + 2) * cost_dp(d);
    return sizeof(float) * (cost_ws_head_floats(B0, B1, d) + centred) + 256;
}
static bool cost_use_mfma(int B0, int B1, int d) {
    // (the engine addresses an operand with 32-bit byte offsets: clouds beyond 4 GiB take the VALU tiles)
    return d >= 64 && B0 >= 256 && B1 >= 256 && ((size_t)(B0 > B1 ? B0 : B1) + 1) * cost_dp(d) * 4 < (1ull << 32);
}
static int cost_mfma(const float* x0, const float* x1, int B0, int B1, int d, float* M, void* ws,
                     hipStream_t st) {
    float* mu = reinterpret_cast<float*>(ws);
    float* nrm = mu + ((d + 63) & ~63);
    float* xc = nrm + (((size_t)B0 + (size_t)B1 + 3) & ~(size_t)3);
    hipLaunchKernelGGL(cost_center, dim3((d + 63) / 64), dim3(256), 0, st, x0, x1, B0, B1, d, mu);
    hipLaunchKernelGGL(cost_norms, dim3((B0 + B1 + 2 + 3) / 4), dim3(256), 0, st, x0, x1, B0, B1, d, mu, nrm, xc);
    // (256 x 128 tiles — 8 MFMA tiles per wave, one resident round at B = 4096 — were measured at
    //  549 us against 368 us for 128 x 128: the accumulators leave two waves per SIMD no room.)
    const int tm = (B0 + 127) / 128, tn = (B1 + 127) / 128;
    hipLaunchKernelGGL(cost_gemm_glds, dim3(tm * tn), dim3(256), GL_LDS_BYTES, st, xc, xc + (size_t)(B0 + 1) * cost_dp(d), x0, x1,
                       B0, B1, d, nrm, M, tm, tn);
    return cfm_status();
}

static int cost_dispatch(const float* x0, const float* x1, int B0, int B1, int d, float* M,
                         float* opt_max, void* ws, void* stream);

extern "C" int cfm_sqeuclid_cost_f32(const float* x0, const float* x1, int B0, int B1, int d,
                                     float* M, float* opt_max, void* stream) {
    return cost_dispatch(x0, x1, B0, B1, d, M, opt_max, nullptr, stream);
}

extern "C" int cfm_sqeuclid_cost_ws_f32(const float* x0, const float* x1, int B0, int B1, int d,
                                        float* M, float* opt_max, void* ws, void* stream) {
    if (!ws || ((uintptr_t)ws & 15) != 0) return ws ? CFM_EALIGN : CFM_EINVAL;
    return cost_dispatch(x0, x1, B0, B1, d, M, opt_max, ws, stream);
}

static int cost_dispatch(const float* x0, const float* x1, int B0, int B1, int d, float* M,
                         float* opt_max, void* ws, void* stream) {
    if (!x0 || !x1 || !M || B0 < 0 || B1 < 0 || d <= 0) return CFM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B0 == 0 || B1 == 0) return 0;
    bool small = true;
    if (ws && cost_use_mfma(B0, B1, d)) {
        int rc = cost_mfma(x0, x1, B0, B1, d, M, ws, st);
        if (rc) return rc;
        small = true;      // done
        goto finish;
    }
    switch (d) {
        case 1: launch_small<1>(x0, x1, B0, B1, M, st); break;
        case 2: launch_small<2>(x0, x1, B0, B1, M, st); break;
        case 3: launch_small<3>(x0, x1, B0, B1, M, st); break;
        case 4: launch_small<4>(x0, x1, B0, B1, M, st); break;
        case 5: launch_small<5>(x0, x1, B0, B1, M, st); break;
        case 6: launch_small<6>(x0, x1, B0, B1, M, st); break;
        case 7: launch_small<7>(x0, x1, B0, B1, M, st); break;
        case 8: launch_small<8>(x0, x1, B0, B1, M, st); break;
        default: small = false;
    }
    if (!small) {
        const bool vec = (d % 4 == 0) && (((uintptr_t)x0 & 15) == 0) && (((uintptr_t)x1 & 15) == 0);
        const bool big = (B0 >= 1024 && B1 >= 1024);
        if (big) {
            int tm = (B0 + 127) / 128, tn = (B1 + 127) / 128;
            if (vec) hipLaunchKernelGGL((cost_tiled<8, true>), dim3(tm * tn), dim3(256), 0, st, x0, x1, B0, B1, d, M, tm, tn);
            else     hipLaunchKernelGGL((cost_tiled<8, false>), dim3(tm * tn), dim3(256), 0, st, x0, x1, B0, B1, d, M, tm, tn);
        } else {
            int tm = (B0 + 63) / 64, tn = (B1 + 63) / 64;
            if (vec) hipLaunchKernelGGL((cost_tiled<4, true>), dim3(tm * tn), dim3(256), 0, st, x0, x1, B0, B1, d, M, tm, tn);
            else     hipLaunchKernelGGL((cost_tiled<4, false>), dim3(tm * tn), dim3(256), 0, st, x0, x1, B0, B1, d, M, tm, tn);
        }
    }
finish:
    int rc = cfm_status();
    if (rc) return rc;
    if (opt_max) {
        rc = cfm_hip(hipMemsetAsync(opt_max, 0, sizeof(float), st));
        if (rc) return rc;
        size_t n = (size_t)B0 * B1;
        int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(max_reduce_f32, dim3(blocks), dim3(256), 0, st, M, n, (unsigned*)opt_max);
        rc = cfm_status();
    }
    return rc;
}

extern "C" int cfm_scale_inv_f32(float* M, size_t n, const float* maxval, void* stream) {
    if (!M || !maxval) return CFM_EINVAL;
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_inv_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, n, maxval);
    return cfm_status();
}

extern "C" int cfm_sqrt_inplace_f32(float* M, size_t n, void* stream) {
    if (!M) return CFM_EINVAL;
    if (n == 0) return 0;
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(sqrt_inplace_f32, dim3(blocks), dim3(256), 0, (hipStream_t)stream, M, n);
    return cfm_status();
}
