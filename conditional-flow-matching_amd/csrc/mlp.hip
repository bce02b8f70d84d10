// mlp.hip — K10: MLP vector field on fp32 MFMA for gfx950.
//
// Replaces MLP.forward (torchcfm/models/models.py:10-21: Linear-SELU x3 +
// Linear) called through torch_wrapper.forward (torchcfm/utils.py:51-52:
// cat([x, t.repeat(B)[:, None]], 1)).
//
// Each layer is one fused kernel  out = act(X W^T + b [+ t * W[:, d]]):
//   * the time column is never concatenated: its rank-1 contribution is added
//     in the epilogue (scalar t inside an ODE solve, per-row t in training);
//   * GEMM on v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise an fmaf chain), so
//     1e-5 parity with the fp32 reference is natural; 157 TFLOP/s peak;
//   * both operands are K-contiguous in memory (torch.nn.Linear keeps W as
//     [out, in]) -> "NT" GEMM on the shared tile engine (gemm_core.h): K-major LDS
//     tiles, two stages and one barrier per K step, ds_read_b64 fragments;
//   * bias + time column + SELU fused into the accumulator epilogue.
#include "cfm_common.h"
#include "gemm_core.h"
#include "gemm_glds64.h"
#include <stdlib.h>

#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f

__device__ __forceinline__ float selu_f(float x) {
    return x > 0.f ? SELU_SCALE * x : (SELU_SCALE * SELU_ALPHA) * expm1f(x);
}

// X [B,K] (row stride lda), W [N, *] (row stride ldw, first K columns used),
// out [B,N].  tcol: column index of the time weight inside W rows (ldw > K) or -1.
// The product runs on the shared tile engine (gemm_core.h): both operands K-contiguous, K-major LDS tiles,
// double-buffered stages, ds_read_b64 fragments; VEC = 16-byte global loads (rows 16-byte aligned).
// Epilogue of a layer tile: bias + time column + (pre-activation) + SELU (+ the MSE seed of the regression step); a lane
// owns EU adjacent columns.  G: the tile engine's accumulators (at(m, u, r), row_of(m, r), col_lo()).
struct LayerEpi {
    const float* W; int ldw; const float* bias; const float* tptr; float tval; int t_per_row, tcol, B, N;
    float* out; float* zout; const float* mse_u; float mse_scale, mse_inv_n; float* mse_partial;
};
template <bool ACT, typename G>
__device__ __forceinline__ void layer_epilogue(const G& g, int row0, int col0, const LayerEpi& E) {
    constexpr int EU = G::EU, EM = G::EM, ER = G::ER;
    const float* __restrict__ W = E.W; const float* __restrict__ tptr = E.tptr; const float* __restrict__ mse_u = E.mse_u;
    float* __restrict__ out = E.out; float* __restrict__ zout = E.zout;
    const int tcol = E.tcol, t_per_row = E.t_per_row, B = E.B, N = E.N, ldw = E.ldw;
    const float tsc = (tcol >= 0 && !t_per_row) ? (tptr ? tptr[0] : E.tval) : 0.f;
    const int gc = col0 + G::col_lo();
    float bv[2] = {0.f, 0.f}, wt[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < EU; ++u) {
        if (gc + u < N) {
            bv[u] = E.bias ? E.bias[gc + u] : 0.f;
            wt[u] = (tcol >= 0) ? W[(size_t)(gc + u) * ldw + tcol] : 0.f;
        }
    }
    const bool pair = EU == 2 && (N & 1) == 0 && gc + 1 < N;       // 8-byte aligned store of both columns
    float lsum = 0.f;
#pragma unroll
    for (int m = 0; m < EM; ++m) {
#pragma unroll
        for (int r = 0; r < ER; ++r) {
            const int gr = row0 + G::row_of(m, r);
            if (gr >= B || gc >= N) continue;
            const float tv = (tcol >= 0) ? (t_per_row ? tptr[gr] : tsc) : 0.f;
            float v[2], z[2];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                float x = g.at(m, u, r) + bv[u];
                if (tcol >= 0) x = fmaf(tv, wt[u], x);
                z[u] = x;
                v[u] = ACT ? selu_f(x) : x;
                if (!ACT && mse_u != nullptr && gc + u < N) {
                    const float dlt = v[u] - mse_u[(size_t)gr * N + gc + u];
                    lsum = fmaf(dlt, dlt, lsum);
                    v[u] = dlt * E.mse_scale;
                }
            }
            float* po = out + (size_t)gr * N + gc;
            if (pair) {
                *reinterpret_cast<float2*>(po) = make_float2(v[0], v[EU - 1]);
                if (zout) *reinterpret_cast<float2*>(zout + (size_t)gr * N + gc) = make_float2(z[0], z[EU - 1]);   // training: SELU' needs exp(z), not h
            } else {
#pragma unroll
                for (int u = 0; u < EU; ++u)
                    if (gc + u < N) { po[u] = v[u]; if (zout) zout[(size_t)gr * N + gc + u] = z[u]; }
            }
        }
    }
    if (!ACT && E.mse_partial != nullptr) {        // (uniform) this workgroup's share of the loss, in a fixed order
        __shared__ float msh[4];
        lsum = wave_sum_f(lsum);
        if ((threadIdx.x & 63) == 0) msh[threadIdx.x >> 6] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) E.mse_partial[blockIdx.x] = ((msh[0] + msh[1]) + (msh[2] + msh[3])) * E.mse_inv_n;
    }
}

template <int BM, int BN, int BK, bool ACT, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void mlp_layer(const float* __restrict__ X, int lda,
                                                 const float* __restrict__ W, int ldw,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ tptr, float tval,
                                                 int t_per_row, int tcol, int B, int K, int N,
                                                 float* __restrict__ out, int tiles_n,
                                                 float* __restrict__ zout = nullptr,
                                                 const float* __restrict__ mse_u = nullptr,      // regression step, last layer: out = (2 / n) (v - u) and
                                                 float mse_scale = 0.f, float mse_inv_n = 0.f,   // mse_partial[workgroup] = sum (v - u)^2 / n  (round 6: no
                                                 float* __restrict__ mse_partial = nullptr) {    // separate MSE launch, no second pass over v)
    using Core = GemmCore<BM, BN, BK, false, false, VECA, VECB>;
    __shared__ __attribute__((aligned(16))) float lds[Core::LDS_FLOATS];
    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    Core g;
    g.zero();
    g.run(lds, X, lda, row0, B, W, ldw, col0, N, 0, K, GcNoPost());
    const LayerEpi E = {W, ldw, bias, tptr, tval, t_per_row, tcol, B, N, out, zout, mse_u, mse_scale, mse_inv_n, mse_partial};
    layer_epilogue<ACT>(g, row0, col0, E);
}

// The same layer on the direct-to-LDS engine (gemm_glds64.h): 64 x 64 tiles, operands DMA'd row-major into LDS, one
// ds_read_b128 per 16 x 16 block and 16 k.  Preconditions (launch_layer): K % 16 == 0, both operands' rows 16-byte
// aligned, every byte offset below 4 GiB.
#define MLP_GLDS_NST 3
template <bool ACT>
__global__ __launch_bounds__(256) void mlp_layer_glds(const float* __restrict__ X, int lda, const float* __restrict__ W, int ldw,
                                                      int K, int tiles_n, LayerEpi E) {
    extern __shared__ __attribute__((aligned(16))) float glds[];
    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * G6_BM, col0 = tn * G6_BN;
    Glds64<MLP_GLDS_NST> g;
    g.zero();
    g.run(glds, X, lda, row0, E.B, W, ldw, col0, E.N, K);
    layer_epilogue<ACT>(g, row0, col0, E);
}

extern "C" size_t cfm_mlp_ws_bytes_internal(int B, int width) {
    return 2 * sizeof(float) * (size_t)B * (size_t)width + 256;
}

// Tile choice of the dense products (shared with mlp_train.hip): 0 = 128 x 128 x 16, 2 = 64 x 64 x 32 — the largest
// tile that still gives the chip about two workgroups per CU (same bits whatever the tile).
// Measured at the C3 layer shapes: 128 x 128 needs >= 2 workgroups per CU to pay (1024 tiles at 4096 x 4096); below
// that the 64 x 64 tile (four 16x16x4 accumulators per wave, 2+ workgroups per CU) wins.  128 x 64 at one workgroup
// per CU was measured twice (round 3; round 5 with the pipelined K-step boundary: C3 model step 501 vs 428 us,
// profiles/r5_experiments.txt) and removed.
int cfm_gemm_pick_tile(long M, long N, long splits) {
    const long t0 = ((M + 127) / 128) * ((N + 127) / 128) * splits;
    return t0 >= 512 ? 0 : 2;
}

// 0: the register-staged core for every layer; 1: 64 x 64 layers with 16-byte aligned rows take the direct-to-LDS engine;
// 2 (default): rows that are only 4-byte aligned too (the 785-wide first layer of a time-varying field: a
// global_load_lds_dwordx4 needs dword alignment only).  Measured at C3 (tools/probe/glds64_probe.py): forward 143.5 /
// 138.9 / 128.7 us, model step 377.9 / 379.0 / 368.6 us for modes 0 / 1 / 2.
static int g_mlp_glds = -1;
static int mlp_glds_mode() {
    if (g_mlp_glds < 0) { const char* e = getenv("CFM_MLP_GLDS"); g_mlp_glds = e ? atoi(e) : 2; if (g_mlp_glds < 0 || g_mlp_glds > 2) g_mlp_glds = 2; }
    return g_mlp_glds;
}
extern "C" void cfm_mlp_set_glds(int mode) { g_mlp_glds = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
extern "C" int cfm_mlp_get_glds(void) { return mlp_glds_mode(); }

template <bool ACT, bool VECA, bool VECB>
static void launch_layer_t(int tile, const float* X, int lda, const float* W, int ldw, const float* bias,
                           const float* t, float tval, int t_per_row, int tcol, int B, int K, int N, float* out,
                           hipStream_t s, float* zout, const float* mse_u = nullptr, float mse_scale = 0.f, float mse_inv_n = 0.f,
                           float* mse_partial = nullptr) {
    if (tile == 0) {
        const int tm = (B + 127) / 128, tn = (N + 127) / 128;
        hipLaunchKernelGGL((mlp_layer<128, 128, 16, ACT, VECA, VECB>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout, mse_u, mse_scale, mse_inv_n, mse_partial);
    } else {
        const int tm = (B + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((mlp_layer<64, 64, 32, ACT, VECA, VECB>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout, mse_u, mse_scale, mse_inv_n, mse_partial);
    }
}

// one layer launch; picks the tile so the grid covers the chip.  16-byte loads per operand when its rows allow it
// (K % 4 == 0, row pitch % 4 == 0, aligned base): the 785-wide rows of a time-varying first layer do not.
static int launch_layer(const float* X, int lda, const float* W, int ldw, const float* bias,
                        const float* t, float tval, int t_per_row, int tcol, int B, int K, int N, float* out,
                        bool act, hipStream_t s, float* zout = nullptr, const float* mse_u = nullptr, float mse_scale = 0.f,
                        float mse_inv_n = 0.f, float* mse_partial = nullptr, int* mse_blocks = nullptr) {
    const int tile = cfm_gemm_pick_tile(B, N, 1);
    if (mse_blocks) *mse_blocks = (tile == 0) ? ((B + 127) / 128) * ((N + 127) / 128) : ((B + 63) / 64) * ((N + 63) / 64);
    const bool va = (K % 4 == 0) && (lda % 4 == 0) && ((uintptr_t)X & 15) == 0;
    const bool vb = (K % 4 == 0) && (ldw % 4 == 0) && ((uintptr_t)W & 15) == 0;
    // the direct-to-LDS form of the 64 x 64 tile (round 6)
    const int gmode = mlp_glds_mode();
    const bool ga = gmode == 2 ? (((uintptr_t)X & 3) == 0) : va, gb = gmode == 2 ? (((uintptr_t)W & 3) == 0) : vb;
    if (gmode && tile == 2 && K >= 16 && K % 16 == 0 && ga && gb &&
        (size_t)B * (size_t)lda * 4 < 0xffff0000ull && (size_t)N * (size_t)ldw * 4 < 0xffff0000ull) {
        const int tm = (B + 63) / 64, tn = (N + 63) / 64;
        const LayerEpi E = {W, ldw, bias, t, tval, t_per_row, tcol, B, N, out, zout, mse_u, mse_scale, mse_inv_n, mse_partial};
        constexpr int lds_bytes = Glds64<MLP_GLDS_NST>::LDS_BYTES;
        if (act) hipLaunchKernelGGL(mlp_layer_glds<true>, dim3(tm * tn), dim3(256), lds_bytes, s, X, lda, W, ldw, K, tn, E);
        else hipLaunchKernelGGL(mlp_layer_glds<false>, dim3(tm * tn), dim3(256), lds_bytes, s, X, lda, W, ldw, K, tn, E);
        return cfm_status();
    }
#define CFM_LL(ACT_, VA_, VB_) launch_layer_t<ACT_, VA_, VB_>(tile, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, s, zout, mse_u, mse_scale, mse_inv_n, mse_partial)
    if (act) { if (va) { if (vb) CFM_LL(true, true, true); else CFM_LL(true, true, false); }
               else    { if (vb) CFM_LL(true, false, true); else CFM_LL(true, false, false); } }
    else     { if (va) { if (vb) CFM_LL(false, true, true); else CFM_LL(false, true, false); }
               else    { if (vb) CFM_LL(false, false, true); else CFM_LL(false, false, false); } }
#undef CFM_LL
    return cfm_status();
}

// (mlp_train.hip: the fused regression step runs its forward through the same launcher)
int cfm_mlp_launch_layer(const float* X, int lda, const float* W, int ldw, const float* bias, const float* t,
                         int t_per_row, int tcol, int B, int K, int N, float* out, bool act, hipStream_t s, float* zout) {
    return launch_layer(X, lda, W, ldw, bias, t, 0.f, t_per_row, tcol, B, K, N, out, act, s, zout);
}
// the last layer of the fused regression step: out = (2 / n) (v - u), one loss partial per workgroup (*n_partials of them)
int cfm_mlp_launch_layer_mse(const float* X, int lda, const float* W, int ldw, const float* bias, const float* t,
                             int t_per_row, int tcol, int B, int K, int N, float* out, hipStream_t s, const float* u,
                             float scale, float inv_n, float* partial, int* n_partials) {
    return launch_layer(X, lda, W, ldw, bias, t, 0.f, t_per_row, tcol, B, K, N, out, false, s, nullptr, u, scale, inv_n, partial, n_partials);
}

// Forward through all layers.  dims[0] counts the time column when the net is
// time varying (has_t).  Time comes from `t` (device: scalar or [B]) or, when t is
// NULL, from the by-value `tval` (ODE drivers).  ws: two [B, maxw] activations.
int cfm_mlp_forward_impl(const float* x, const float* t, float tval, int has_t, int t_per_row,
                         const float* const* W, const float* const* b, const int* dims,
                         int n_layers, int B, float* out, void* ws, hipStream_t s) {
    int maxw = 0;
    for (int l = 1; l < n_layers; ++l) maxw = dims[l] > maxw ? dims[l] : maxw;
    float* buf[2] = {(float*)ws, (float*)ws + (size_t)B * maxw};
    const float* cur = x;
    for (int l = 0; l < n_layers; ++l) {
        const int in = dims[l], on = dims[l + 1];
        const bool first = (l == 0);
        const int K = (first && has_t) ? in - 1 : in;
        const int tcol = (first && has_t) ? K : -1;
        float* dst = (l == n_layers - 1) ? out : buf[l & 1];
        int rc = launch_layer(cur, K, W[l], in, b[l], t, tval, t_per_row, tcol, B, K, on, dst,
                              l != n_layers - 1, s);
        if (rc) return rc;
        cur = dst;
    }
    return 0;
}

extern "C" int cfm_mlp_forward_f32(const float* x, const float* t, int t_per_row,
                                   const float* const* W, const float* const* b, const int* dims,
                                   int n_layers, int B, float* out, void* ws, void* stream) {
    if (!x || !W || !b || !dims || !out || n_layers < 1 || B < 0) return CFM_EINVAL;
    if (n_layers > 1 && !ws) return CFM_EINVAL;
    if (B == 0) return 0;
    return cfm_mlp_forward_impl(x, t, 0.f, t != nullptr, t_per_row, W, b, dims, n_layers, B, out, ws,
                                (hipStream_t)stream);
}

// Training forward: the same layer kernels, every hidden layer's activation h_l = selu(z_l) AND
// pre-activation z_l kept in caller buffers (hidden[l], preact[l] : [B, dims[l + 1]], l = 0 .. n_layers - 2)
// for cfm_mlp_backward_f32 (mlp_train.hip): h is the next layer's wgrad operand, z gives
// selu'(z) = scale * alpha * exp(z) without the cancellation of h + scale * alpha.
// x already holds every input column (the reference concatenates the time itself:
// net(torch.cat([xt, t[:, None]], -1)), examples/images/cifar10/train_cifar10.py:147).
extern "C" int cfm_mlp_forward_train_f32(const float* x, const float* const* W, const float* const* b,
                                         const int* dims, int n_layers, int B, float* const* hidden,
                                         float* const* preact, float* out, void* stream) {
    if (!x || !W || !b || !dims || !out || n_layers < 1 || B < 0) return CFM_EINVAL;
    if (n_layers > 1 && (!hidden || !preact)) return CFM_EINVAL;
    if (B == 0) return 0;
    const float* cur = x;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float* dst = last ? out : hidden[l];
        int rc = launch_layer(cur, dims[l], W[l], dims[l], b[l], nullptr, 0.f, 0, -1, B, dims[l], dims[l + 1], dst,
                              !last, (hipStream_t)stream, last ? nullptr : preact[l]);
        if (rc) return rc;
        cur = dst;
    }
    return 0;
}
