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
//     [out, in]) -> "NT" GEMM, k-contiguous LDS rows with odd stride
//     (conflict-free ds_read_b32 fragments: lane&31 -> row, lane>>5 -> k);
//   * next K-tile is fetched into registers while the current one feeds the
//     matrix pipe (issue-early / write-late staging);
//   * bias + time column + SELU fused into the accumulator epilogue.
#include "cfm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f

__device__ __forceinline__ float selu_f(float x) {
    return x > 0.f ? SELU_SCALE * x : (SELU_SCALE * SELU_ALPHA) * expm1f(x);
}

// X [B,K] (row stride lda), W [N, *] (row stride ldw, first K columns used),
// out [B,N].  tcol: column index of the time weight inside W rows (ldw > K) or -1.
template <int BM, int BN, bool ACT>
__global__ __launch_bounds__(256) void mlp_layer(const float* __restrict__ X, int lda,
                                                 const float* __restrict__ W, int ldw,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ tptr, float tval,
                                                 int t_per_row, int tcol, int B, int K, int N,
                                                 float* __restrict__ out, int tiles_n,
                                                 float* __restrict__ zout = nullptr) {
    constexpr int BK = 32, LD = BK + 1;
    constexpr int WM = BM / 2, WN = BN / 2;          // per-wave tile (2x2 waves)
    constexpr int MT = WM / 32, NT = WN / 32;        // 32x32 MFMA tiles per wave
    constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;   // floats per thread per stage
    __shared__ float As[BM * LD];
    __shared__ float Bs[BN * LD];

    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[A_PER], rb[B_PER];
    // element e of a [R x BK] stage: row = e / BK, k = e % BK; thread owns
    // e = tid + 256*q (consecutive lanes -> consecutive k: coalesced 128-B rows)
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            const int e = tid + 256 * q, r = e / BK, k = e % BK;
            const int gr = row0 + r, gk = k0 + k;
            ra[q] = (gr < B && gk < K) ? X[(size_t)gr * lda + gk] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            const int e = tid + 256 * q, r = e / BK, k = e % BK;
            const int gr = col0 + r, gk = k0 + k;
            rb[q] = (gr < N && gk < K) ? W[(size_t)gr * ldw + gk] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < A_PER; ++q) { const int e = tid + 256 * q; As[(e / BK) * LD + (e % BK)] = ra[q]; }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) { const int e = tid + 256 * q; Bs[(e / BK) * LD + (e % BK)] = rb[q]; }
    };

    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        stash();
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);            // in flight while the MFMAs run
        const int fr = lane & 31, fk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = As[(wm * WM + m * 32 + fr) * LD + kk + fk];
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) b[nn] = Bs[(wn * WN + nn * 32 + fr) * LD + kk + fk];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nn], acc[m][nn], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const float tsc = (tcol >= 0 && !t_per_row) ? (tptr ? tptr[0] : tval) : 0.f;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        const int gc = col0 + wn * WN + nn * 32 + (lane & 31);
        if (gc >= N) continue;
        const float bv = bias ? bias[gc] : 0.f;
        const float wt = (tcol >= 0) ? W[(size_t)gc * ldw + tcol] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = row0 + wm * WM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gr >= B) continue;
                float v = acc[m][nn][r] + bv;
                if (tcol >= 0) v = fmaf(t_per_row ? tptr[gr] : tsc, wt, v);
                if (zout) zout[(size_t)gr * N + gc] = v;      // training: the pre-activation (SELU' needs exp(z), not h)
                if (ACT) v = selu_f(v);
                out[(size_t)gr * N + gc] = v;
            }
        }
    }
}

extern "C" size_t cfm_mlp_ws_bytes_internal(int B, int width) {
    return 2 * sizeof(float) * (size_t)B * (size_t)width + 256;
}

// one layer launch; picks the tile so the grid covers the chip
static int launch_layer(const float* X, int lda, const float* W, int ldw, const float* bias,
                        const float* t, float tval, int t_per_row, int tcol, int B, int K, int N, float* out,
                        bool act, hipStream_t s, float* zout = nullptr) {
    const long tiles128 = (long)((B + 127) / 128) * ((N + 127) / 128);
    if (tiles128 >= 512) {
        const int tm = (B + 127) / 128, tn = (N + 127) / 128;
        if (act) hipLaunchKernelGGL((mlp_layer<128, 128, true>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout);
        else     hipLaunchKernelGGL((mlp_layer<128, 128, false>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout);
    } else {
        const int tm = (B + 63) / 64, tn = (N + 63) / 64;
        if (act) hipLaunchKernelGGL((mlp_layer<64, 64, true>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout);
        else     hipLaunchKernelGGL((mlp_layer<64, 64, false>), dim3(tm * tn), dim3(256), 0, s, X, lda, W, ldw, bias, t, tval, t_per_row, tcol, B, K, N, out, tn, zout);
    }
    return cfm_status();
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
