// mlp_train.hip — training step of the MLP vector field on fp32 MFMA (gfx950): backward GEMMs with
// fused SELU' epilogue, bias gradients, fused multi-tensor Adam.
//
// Replaces, for torchcfm.models.MLP (torchcfm/models/models.py:10-21), what autograd + torch.optim.Adam
// execute in the reference's training loop (examples/images/cifar10/train_cifar10.py:141-151:
// vt = net(...); loss = mean((vt - ut)^2); loss.backward(); optim.step()):
//
//   forward (training)   the same mlp_layer kernels as inference (mlp.hip), keeping h_l = selu(z_l) (the
//                        next wgrad's operand) and z_l (selu'(z) = scale * alpha * exp(z): recovering it from
//                        h as h + scale * alpha cancels catastrophically for saturated units — measured
//                        1.7e-3 relative error on the first layer's gradient at d = 784)
//   dgrad                dz_{l-1} = (dz_l . W_l) * selu'(z_{l-1})     [B,N] x [N,K]  ("NN")
//   wgrad                dW_l = dz_l^T . h_{l-1}                      [N,B] x [B,K]  ("TN"), the
//                        contraction runs over the batch: split over S batch chunks so that a 512 x 512
//                        gradient still fills the chip, partial sums reduced in a fixed order
//                        (deterministic, no atomics)
//   bias grad            db_l = column sums of dz_l, accumulated by the wgrad workgroups of the first tile column
//                        from the dz tile they stage anyway; every split-K partial of the pass (weights and
//                        biases of all layers) is reduced by ONE table-driven launch
//   Adam                 one launch for every parameter tensor of the model (pointer table), torch.optim.Adam
//                        arithmetic (lerp / addcmul / addcdiv order, bias corrections as Python doubles)
//
// All GEMMs are v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak) on the shared tile engine of gemm_core.h:
// K-major LDS tiles whatever the operand's storage order (k-contiguous operands are transposed on the way in),
// two stages, ds_read_b64 fragments.
#include "cfm_common.h"
#include "gemm_core.h"

#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f

// selu'(z)
__device__ __forceinline__ float selu_grad(float z) {
    return z > 0.f ? SELU_SCALE : (SELU_SCALE * SELU_ALPHA) * expf(z);
}

enum { EPI_PLAIN = 0, EPI_SELU_GRAD = 1 };

// C[M,N] (+ epilogue) = A[M,Kc] . B[Kc,N], contraction over [k_begin, k_end) of this workgroup's split, on the
// shared tile engine (gemm_core.h).
//   A_KMAJOR = false: A(i,k) = A[i * lda + k]      true: A(i,k) = A[k * lda + i]
//   B_KMAJOR = false: B(k,j) = Bm[j * ldb + k]     true: B(k,j) = Bm[k * ldb + j]
// grid: x = tiles_m * tiles_n (XCD-remapped), y = split index s; split s writes C + s * split_stride.
// the arguments of one product (a kernel argument of the single and of the paired launch)
struct GemmArgs {
    const float* A; int lda; const float* Bm; int ldb; float* C; int ldc; size_t split_stride;
    const float* H;            // EPI_SELU_GRAD: pre-activations [M, ldc]
    int M, N, Kc, k_chunk, tiles_n;
    float* colsum;             // A_KMAJOR: sum_k A(i, k) per split -> colsum[s * M + i]
    const float* tvec;         // A_KMAJOR: weights t[k] of a second column sum
    float* tsum;               //   sum_k A(i, k) t[k] per split -> tsum[s * M + i]
};

template <int BM, int BN, int BK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VECA, bool VECB>
__device__ __forceinline__ void gemm_tile(float* __restrict__ lds, unsigned lid, int split, const GemmArgs& G) {
    using Core = GemmCore<BM, BN, BK, A_KMAJOR, B_KMAJOR, VECA, VECB>;
    const float* __restrict__ A = G.A; const float* __restrict__ Bm = G.Bm; float* __restrict__ C = G.C;
    const float* __restrict__ H = G.H; float* __restrict__ colsum = G.colsum; const float* __restrict__ tvec = G.tvec;
    float* __restrict__ tsum = G.tsum;
    const int lda = G.lda, ldb = G.ldb, ldc = G.ldc, M = G.M, N = G.N, Kc = G.Kc, k_chunk = G.k_chunk, tiles_n = G.tiles_n;
    const size_t split_stride = G.split_stride;
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x;
    const int k_begin = split * k_chunk;
    const int k_end = (k_begin + k_chunk < Kc) ? k_begin + k_chunk : Kc;

    // bias gradient rides along: the workgroups of the first tile column add up their A tile (dz) over k, from the
    // K-major LDS stage every step (same ascending-k order as before)
    // (the time column of a time-varying first layer is not part of the GEMM operand: its gradient
    //  dW[:, d] = sum_b dz[b, :] t[b] is a second, weighted column sum of the same tile)
    const bool do_colsum = A_KMAJOR && colsum != nullptr && tn == 0 && tid < BM;
    const bool do_tsum = do_colsum && tvec != nullptr && tsum != nullptr;
    float csum = 0.f, wsum = 0.f;
    auto post = [&](const float* As, int k0) {
        if (A_KMAJOR && do_colsum) {
            if (do_tsum) {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) {
                    const float a = As[kk * Core::LDA + tid];
                    csum += a;
                    wsum = fmaf(a, (k0 + kk < k_end) ? tvec[k0 + kk] : 0.f, wsum);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BK; ++kk) csum += As[kk * Core::LDA + tid];
            }
        }
    };
    Core g;
    g.zero();
    g.run(lds, A, lda, row0, M, Bm, ldb, col0, N, k_begin, k_end, post);
    if (A_KMAJOR && do_colsum && row0 + tid < M) {
        colsum[(size_t)split * M + row0 + tid] = csum;
        if (do_tsum) tsum[(size_t)split * M + row0 + tid] = wsum;
    }

    constexpr int EU = Core::EU, EM = Core::EM, ER = Core::ER;
    float* Cs = C + (size_t)split * split_stride;
    const int gc = col0 + Core::col_lo();
    const bool pair = EU == 2 && (ldc & 1) == 0 && gc + 1 < N;
#pragma unroll
    for (int m = 0; m < EM; ++m) {
#pragma unroll
        for (int r = 0; r < ER; ++r) {
            const int gr = row0 + Core::row_of(m, r);
            if (gr >= M || gc >= N) continue;
            float v[2];
#pragma unroll
            for (int u = 0; u < EU; ++u) v[u] = g.at(m, u, r);
            if (EPI == EPI_SELU_GRAD) {
                if (pair) {
                    const float2 h = *reinterpret_cast<const float2*>(H + (size_t)gr * ldc + gc);
                    v[0] *= selu_grad(h.x); v[EU - 1] *= selu_grad(h.y);
                } else {
#pragma unroll
                    for (int u = 0; u < EU; ++u) if (gc + u < N) v[u] *= selu_grad(H[(size_t)gr * ldc + gc + u]);
                }
            }
            float* po = Cs + (size_t)gr * ldc + gc;
            if (pair) *reinterpret_cast<float2*>(po) = make_float2(v[0], v[EU - 1]);
            else {
#pragma unroll
                for (int u = 0; u < EU; ++u) if (gc + u < N) po[u] = v[u];
            }
        }
    }
}


// grid: x = tiles_m * tiles_n (XCD-remapped), y = split index s; split s writes C + s * split_stride.
template <int BM, int BN, int BK, bool A_KMAJOR, bool B_KMAJOR, int EPI, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void gemm_f32_mfma(GemmArgs G) {
    using Core = GemmCore<BM, BN, BK, A_KMAJOR, B_KMAJOR, VECA, VECB>;
    __shared__ __attribute__((aligned(16))) float lds[Core::LDS_FLOATS];
    gemm_tile<BM, BN, BK, A_KMAJOR, B_KMAJOR, EPI, VECA, VECB>(lds, cfm_xcd_remap(blockIdx.x, gridDim.x), (int)blockIdx.y, G);
}

// dgrad and wgrad of ONE layer in one launch (round 6): both read dz_l and nothing of each other.  A 1-D grid: the
// dgrad tiles first (their contraction is the long one: 512 - 784 against 256 per wgrad split), then tiles x splits of
// wgrad; 64 x 64 x 32 tiles and 16-byte loads on both (the launcher falls back to two launches otherwise).  Same tile
// routine, same bits.  14 -> 11 launches per C3 model step.
__global__ __launch_bounds__(256) void gemm_pair_f32_mfma(GemmArgs D, int tiles_d, GemmArgs Wg, int tiles_w, int splits_w) {
    using CoreD = GemmCore<64, 64, 32, false, true, true, true>;
    using CoreW = GemmCore<64, 64, 32, true, true, true, true>;
    constexpr int LDSF = CoreD::LDS_FLOATS > CoreW::LDS_FLOATS ? CoreD::LDS_FLOATS : CoreW::LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int b = (int)blockIdx.x;
    if (b < tiles_d) {
        gemm_tile<64, 64, 32, false, true, EPI_SELU_GRAD, true, true>(lds, cfm_xcd_remap((unsigned)b, (unsigned)tiles_d), 0, D);
    } else {
        const int q = b - tiles_d;
        gemm_tile<64, 64, 32, true, true, EPI_PLAIN, true, true>(lds, cfm_xcd_remap((unsigned)(q % tiles_w), (unsigned)tiles_w), q / tiles_w, Wg);
    }
}

// out[e] = sum_s partial[s * stride + e] in split order (deterministic), for every tensor of the table
// (all weight and bias gradients of a backward pass in ONE launch)
struct ReduceJob { const float* partial; float* out; unsigned long long stride, n; int S, cols, ld_out, pad; };   // cols > 0: out[(e / cols) * ld_out + e % cols]
#define MLP_MAX_LAYERS 16
struct ReduceTable { ReduceJob job[2 * MLP_MAX_LAYERS + 2]; int count; };

__global__ __launch_bounds__(256) void reduce_splits_multi(ReduceTable T) {
    for (int q = blockIdx.y; q < T.count; q += gridDim.y) {
        const ReduceJob J = T.job[q];
        if (J.pad == 1) {
            // ONE number out of S partials (the loss: one partial per workgroup of the last layer, ~ 800 of them): lane l
            // adds partials l, l + 64, ... in order, then a fixed tree over the wave — deterministic, and not 800
            // dependent loads in one thread
            if (blockIdx.x == 0 && threadIdx.x < 64) {
                float v = 0.f;
                for (int s2 = threadIdx.x; s2 < J.S; s2 += 64) v += J.partial[s2];
                v = wave_sum_f(v);
                if (threadIdx.x == 0) J.out[0] = v;
            }
            continue;
        }
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < J.n; e += (size_t)gridDim.x * 256) {
            float v = J.partial[e];
            // (8 partials in flight per thread, added in split order: the same sums as one load at a time)
            for (int s0 = 1; s0 < J.S; s0 += 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = (s0 + u < J.S) ? J.partial[(size_t)(s0 + u) * J.stride + e] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) if (s0 + u < J.S) v += t[u];
            }
            if (J.cols > 0) J.out[(e / (unsigned)J.cols) * (size_t)J.ld_out + (e % (unsigned)J.cols)] = v;
            else J.out[e] = v;
        }
    }
}

#define MLP_MAX_SPLITS 32
#define MLP_LOSS_PARTIALS 4096
extern "C" size_t cfm_mlp_train_ws_bytes_internal(int B, int maxw, int max_params) {
    // two [B, maxw] gradient buffers + per-layer split-K partials (weights and biases; <= MLP_MAX_LAYERS layers
    // are sized here by the largest one: callers pass the largest dims[l] * dims[l+1])
    // (+ MLP_LOSS_PARTIALS loss partials for cfm_mlp_regression_step_f32: one per workgroup of the last layer)
    return sizeof(float) * ((size_t)2 * B * maxw + (size_t)MLP_MAX_SPLITS * ((size_t)max_params + maxw) * 4) + 4 * MLP_LOSS_PARTIALS;
}

int cfm_gemm_pick_tile(long M, long N, long splits);      // mlp.hip
int cfm_mlp_launch_layer(const float* X, int lda, const float* W, int ldw, const float* bias, const float* t,
                         int t_per_row, int tcol, int B, int K, int N, float* out, bool act, hipStream_t s, float* zout);
int cfm_mlp_launch_layer_mse(const float* X, int lda, const float* W, int ldw, const float* bias, const float* t,
                             int t_per_row, int tcol, int B, int K, int N, float* out, hipStream_t s, const float* u,
                             float scale, float inv_n, float* partial, int* n_partials);

static GemmArgs gemm_args(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, size_t split_stride,
                          const float* H, int M, int N, int Kc, int k_chunk, int tiles_n, float* colsum, const float* tvec,
                          float* tsum) {
    GemmArgs G;
    G.A = A; G.lda = lda; G.Bm = Bm; G.ldb = ldb; G.C = C; G.ldc = ldc; G.split_stride = split_stride; G.H = H;
    G.M = M; G.N = N; G.Kc = Kc; G.k_chunk = k_chunk; G.tiles_n = tiles_n; G.colsum = colsum; G.tvec = tvec; G.tsum = tsum;
    return G;
}

template <bool AK, bool BK_, int EPI, bool VA, bool VB>
static void launch_gemm_t(int tile, const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, size_t split_stride,
                          const float* H, int M, int N, int Kc, int k_chunk, int S, hipStream_t s, float* colsum,
                          const float* tvec, float* tsum) {
    if (tile == 0) {
        const int tm = (M + 127) / 128, tn = (N + 127) / 128;
        hipLaunchKernelGGL((gemm_f32_mfma<128, 128, 16, AK, BK_, EPI, VA, VB>), dim3(tm * tn, S), dim3(256), 0, s,
                           gemm_args(A, lda, Bm, ldb, C, ldc, split_stride, H, M, N, Kc, k_chunk, tn, colsum, tvec, tsum));
    } else {
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((gemm_f32_mfma<64, 64, 32, AK, BK_, EPI, VA, VB>), dim3(tm * tn, S), dim3(256), 0, s,
                           gemm_args(A, lda, Bm, ldb, C, ldc, split_stride, H, M, N, Kc, k_chunk, tn, colsum, tvec, tsum));
    }
}

static bool gemm_vec_ok(const float* p, int ld, int extent) { return (ld % 4 == 0) && ((uintptr_t)p & 15) == 0 && (extent % 4 == 0); }

template <bool AK, bool BK_, int EPI>
static int launch_gemm(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, size_t split_stride,
                       const float* H, int M, int N, int Kc, int S, hipStream_t s, float* colsum = nullptr,
                       const float* tvec = nullptr, float* tsum = nullptr) {
    int k_chunk = (Kc + S - 1) / S;
    k_chunk = (k_chunk + 31) / 32 * 32;
    const int tile = cfm_gemm_pick_tile(M, N, S);
    // 16-byte loads per operand: K-contiguous needs Kc % 4 == 0 (k_chunk is a multiple of 32), K-major needs the row
    // extent % 4 == 0; both need the pitch % 4 == 0 and an aligned base
    const bool va = gemm_vec_ok(A, lda, AK ? M : Kc);
    const bool vb = gemm_vec_ok(Bm, ldb, BK_ ? N : Kc);
#define CFM_LG(VA_, VB_) launch_gemm_t<AK, BK_, EPI, VA_, VB_>(tile, A, lda, Bm, ldb, C, ldc, split_stride, H, M, N, Kc, k_chunk, S, s, colsum, tvec, tsum)
    if (va) { if (vb) CFM_LG(true, true); else CFM_LG(true, false); }
    else    { if (vb) CFM_LG(false, true); else CFM_LG(false, false); }
#undef CFM_LG
    return cfm_status();
}

// Backward through all layers.  acts[l] = h_l (l = 0: the network input [B, dims[0]]; l = 1 .. n-1: the
// saved hidden activations), preact[l] = z_l for l = 1 .. n-1 (preact[0] unused); dout [B, dims[n]].
// Writes dW[l] ([dims[l+1], dims[l]]), db[l] and, if dx is not NULL, the input gradient [B, dims[0]].
// Launches: per layer ONE launch for wgrad (bias column sums ride along) + dgrad (round 6; two where the shapes do not allow
// the pair), then ONE reduction of every split-K partial (weights and biases of all layers).
// tvec != NULL: the network input is [acts[0] (B x dims[0] - 1, pitch dims[0] - 1), tvec (B)] — the time column is
// kept apart (the fused regression step never concatenates it); its weight gradient is the weighted column sum.
// extra: one more job for the final reduction (the loss partials of the fused step), or NULL.
static int mlp_backward_impl(const float* const* acts, const float* const* preact, const float* const* W,
                             const int* dims, int n_layers, int B, const float* dout, float* const* dW,
                             float* const* db, float* dx, void* ws, hipStream_t s, const float* tvec,
                             const ReduceJob* extra, void* const* layer_done = nullptr) {
    int maxw = 0; size_t maxp = 0;
    for (int l = 0; l <= n_layers; ++l) maxw = dims[l] > maxw ? dims[l] : maxw;
    for (int l = 0; l < n_layers; ++l) { const size_t p = (size_t)dims[l] * dims[l + 1]; maxp = p > maxp ? p : maxp; }
    float* gbuf[2] = {(float*)ws, (float*)ws + (size_t)B * maxw};
    float* pool = gbuf[1] + (size_t)B * maxw;
    const size_t pool_floats = (size_t)MLP_MAX_SPLITS * (maxp + maxw) * 4;
    size_t used = 0;
    ReduceTable T; T.count = 0;
    const float* dz = dout;
    for (int l = n_layers - 1; l >= 0; --l) {
        const bool split_t = (l == 0 && tvec != nullptr);
        const int Kfull = dims[l], N = dims[l + 1];
        const int K = split_t ? Kfull - 1 : Kfull;            // columns of the GEMM operand
        // wgrad: dW[N,K] = dz^T[N,B] . h[B,K], contraction over the batch, S splits; db partials ride along
        const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
        int S = 1;
        while (S < MLP_MAX_SPLITS && tiles * S < 256 && B / (2 * S) >= 64) S *= 2;
        // (round 6, forced split counts at C3: S = 4 / 8 / 16 (this rule) / 32: forward + MSE + backward 444 / 413 / 410 / 477 us —
        //  profiles/r6_experiments.txt)
        const size_t np = (size_t)N * K;
        if (used + (size_t)S * (np + 2 * (size_t)N) > pool_floats) return CFM_EINVAL;      // more layers than the workspace was sized for
        float* part = pool + used; used += (size_t)S * np;
        float* bpart = pool + used; used += (size_t)S * N;
        float* tpart = nullptr;
        if (split_t) { tpart = pool + used; used += (size_t)S * N; }
        // dgrad of the same layer: dz_prev[B,K] = (dz[B,N] . W[N,K]) * selu'(z_prev) — reads dz_l like wgrad and nothing of it:
        // ONE launch for both when both run on 64 x 64 tiles with 16-byte loads (every hidden layer at C3)
        float* dprev = (l > 0) ? gbuf[l & 1] : nullptr;
        int rc = 0;
        bool paired = false;
        if (l > 0 && cfm_gemm_pick_tile(N, K, S) == 2 && cfm_gemm_pick_tile(B, K, 1) == 2 &&
            gemm_vec_ok(dz, N, N) && gemm_vec_ok(acts[l], K, K) && gemm_vec_ok(W[l], K, K) && gemm_vec_ok(dz, N, N)) {
            int k_chunk_w = (B + S - 1) / S; k_chunk_w = (k_chunk_w + 31) / 32 * 32;
            const int k_chunk_d = (N + 31) / 32 * 32;
            const int tnw = (K + 63) / 64, tiles_w = ((N + 63) / 64) * tnw;
            const int tnd = (K + 63) / 64, tiles_d = ((B + 63) / 64) * tnd;
            const GemmArgs Ga = gemm_args(dz, N, W[l], K, dprev, K, 0, preact[l], B, K, N, k_chunk_d, tnd, nullptr, nullptr, nullptr);
            const GemmArgs Gw = gemm_args(dz, N, acts[l], K, part, K, np, nullptr, N, K, B, k_chunk_w, tnw, bpart, nullptr, nullptr);
            hipLaunchKernelGGL(gemm_pair_f32_mfma, dim3(tiles_d + tiles_w * S), dim3(256), 0, s, Ga, tiles_d, Gw, tiles_w, S);
            rc = cfm_status();
            paired = true;
        } else {
            rc = launch_gemm<true, true, EPI_PLAIN>(dz, N, acts[l], K, part, K, np, nullptr, N, K, B, S, s, bpart,
                                                    split_t ? tvec : nullptr, tpart);
        }
        if (rc) return rc;
        T.job[T.count++] = ReduceJob{part, dW[l], np, np, S, split_t ? K : 0, Kfull, 0};
        T.job[T.count++] = ReduceJob{bpart, db[l], (unsigned long long)N, (unsigned long long)N, S, 0, 0, 0};
        if (split_t) T.job[T.count++] = ReduceJob{tpart, dW[l] + K, (unsigned long long)N, (unsigned long long)N, S, 1, Kfull, 0};
        if (layer_done) {
            // bucketed form (data parallel): this layer's gradients are final as soon as its own reduction has run —
            // the caller's communication stream waits for layer_done[l] and all-reduces them while the remaining
            // layers' products run.  Same jobs, same order of the partial sums: bit-equal to the one-reduction form.
            if (l == 0 && extra) T.job[T.count++] = *extra;
            hipLaunchKernelGGL(reduce_splits_multi, dim3(256, T.count), dim3(256), 0, s, T);
            T.count = 0;
            if (layer_done[l]) { const hipError_t e = hipEventRecord((hipEvent_t)layer_done[l], s); if (e != hipSuccess) return (int)e; }
        }
        // dgrad: dz_prev[B,K] = (dz[B,N] . W[N,K]) * selu'(z_prev)
        if (l > 0) {
            if (!paired) {
                rc = launch_gemm<false, true, EPI_SELU_GRAD>(dz, N, W[l], K, dprev, K, 0, preact[l], B, K, N, 1, s);
                if (rc) return rc;
            }
            dz = dprev;
        } else if (dx) {
            rc = launch_gemm<false, true, EPI_PLAIN>(dz, N, W[0], Kfull, dx, Kfull, 0, nullptr, B, Kfull, N, 1, s);
            if (rc) return rc;
        }
    }
    if (layer_done) return cfm_status();
    if (extra) T.job[T.count++] = *extra;
    hipLaunchKernelGGL(reduce_splits_multi, dim3(256, T.count), dim3(256), 0, s, T);
    return cfm_status();
}

extern "C" int cfm_mlp_backward_f32(const float* const* acts, const float* const* preact, const float* const* W,
                                    const int* dims, int n_layers, int B, const float* dout, float* const* dW,
                                    float* const* db, float* dx, void* ws, void* stream) {
    if (!acts || !W || !dims || !dout || !dW || !db || n_layers < 1 || n_layers > MLP_MAX_LAYERS - 1 || B < 0 || !ws) return CFM_EINVAL;
    if (n_layers > 1 && !preact) return CFM_EINVAL;
    if (B == 0) return 0;
    return mlp_backward_impl(acts, preact, W, dims, n_layers, B, dout, dW, db, dx, ws, (hipStream_t)stream, nullptr, nullptr);
}

// ------------------------------------------------------- fused regression step ----
// g = (2 / n) (v - u) in place of v, and the partial sums of (v - u)^2 (one per workgroup, summed in block order by
// the backward's final reduction: deterministic)
#define MSE_BLOCKS 256
__global__ __launch_bounds__(256) void mse_grad(float* __restrict__ v, const float* __restrict__ u, size_t n, float scale,
                                                float inv_n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    const size_t n4 = n / 4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        float4 a = reinterpret_cast<float4*>(v)[e];
        const float4 b = reinterpret_cast<const float4*>(u)[e];
        a.x -= b.x; a.y -= b.y; a.z -= b.z; a.w -= b.w;
        acc = fmaf(a.x, a.x, acc); acc = fmaf(a.y, a.y, acc); acc = fmaf(a.z, a.z, acc); acc = fmaf(a.w, a.w, acc);
        a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
        reinterpret_cast<float4*>(v)[e] = a;
    }
    if (blockIdx.x == 0)
        for (size_t e = n4 * 4 + threadIdx.x; e < n; e += 256) { const float d = v[e] - u[e]; acc = fmaf(d, d, acc); v[e] = d * scale; }
    acc = wave_sum_f(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * inv_n;
}

// One regression step of the vector field on a coupled batch, everything but the optimizer update:
//     v = net([xt, t]);  loss = mean((v - ut)^2);  dW, db = d loss / d parameters
// — what `vt = model(torch.cat([xt, t[:, None]], -1)); loss = torch.mean((vt - ut) ** 2); loss.backward()` does in the
// reference's loops (examples/images/cifar10/train_cifar10.py:147-149, every 2D tutorial).  t == NULL: the net is
// not time varying (dims[0] = columns of xt); otherwise dims[0] = columns of xt + 1 and the time column is never
// materialised: forward adds its rank-1 term in the first layer's epilogue, backward takes its weight gradient
// as a weighted column sum.  g [B, dims[n]] receives d loss / d v (the caller may ignore it); *loss a device float.
// 10 launches for the 4-layer field (round 6: dgrad + wgrad of a layer share one, the MSE rides in the last layer's
// epilogue), all kernels of this library (no eager elementwise ops in between).
extern "C" int cfm_mlp_regression_step_f32(const float* xt, const float* t, const float* ut,
                                           const float* const* W, const float* const* b, const int* dims, int n_layers,
                                           int B, float* const* hidden, float* const* preact, float* g,
                                           float* const* dW, float* const* db, float* loss, void* const* layer_done,
                                           void* ws, void* stream) {
    if (!xt || !ut || !W || !b || !dims || !g || !dW || !db || !loss || !ws || n_layers < 1 || n_layers > MLP_MAX_LAYERS - 1 || B < 1)
        return CFM_EINVAL;
    if (n_layers > 1 && (!hidden || !preact)) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int has_t = t != nullptr;
    if (has_t && dims[0] < 2) return CFM_EINVAL;
    int maxw = 0; size_t maxp = 0;
    for (int l = 0; l <= n_layers; ++l) maxw = dims[l] > maxw ? dims[l] : maxw;
    for (int l = 0; l < n_layers; ++l) { const size_t p = (size_t)dims[l] * dims[l + 1]; maxp = p > maxp ? p : maxp; }
    float* lpart = (float*)((char*)ws + cfm_mlp_train_ws_bytes_internal(B, maxw, (int)maxp) - 4 * MLP_LOSS_PARTIALS);
    const size_t nel = (size_t)B * dims[n_layers];
    const float inv_n = 1.0f / (float)nel;
    // forward, keeping h_l and z_l; the LAST layer's epilogue forms the loss gradient seed g = (2 / n) (v - u) and one
    // loss partial per workgroup (round 6: the MSE was a launch of its own and a second pass over v)
    const float* cur = xt;
    int n_lpart = MSE_BLOCKS;
    bool mse_fused = false;
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        const bool first_t = (l == 0 && has_t);
        const int K = first_t ? dims[0] - 1 : dims[l];
        float* dst = last ? g : hidden[l];
        int rc;
        const long last_wgs = (long)((B + 63) / 64) * ((dims[l + 1] + 63) / 64);      // (an upper bound: the 128 x 128 form has fewer)
        if (last && last_wgs <= MLP_LOSS_PARTIALS) {
            rc = cfm_mlp_launch_layer_mse(cur, K, W[l], dims[l], b[l], first_t ? t : nullptr, first_t ? 1 : 0, first_t ? K : -1,
                                          B, K, dims[l + 1], dst, s, ut, 2.0f * inv_n, inv_n, lpart, &n_lpart);
            mse_fused = true;
        } else {
            rc = cfm_mlp_launch_layer(cur, K, W[l], dims[l], b[l], first_t ? t : nullptr, first_t ? 1 : 0, first_t ? K : -1,
                                      B, K, dims[l + 1], dst, !last, s, last ? nullptr : preact[l]);
        }
        if (rc) return rc;
        cur = dst;
    }
    if (!mse_fused) {
        hipLaunchKernelGGL(mse_grad, dim3(MSE_BLOCKS), dim3(256), 0, s, g, ut, nel, 2.0f * inv_n, inv_n, lpart);
        const int rc = cfm_status();
        if (rc) return rc;
    }
    // backward (+ the loss partials in its final reduction)
    const float* acts[MLP_MAX_LAYERS]; const float* zs[MLP_MAX_LAYERS];
    acts[0] = xt; zs[0] = nullptr;
    for (int l = 1; l < n_layers; ++l) { acts[l] = hidden[l - 1]; zs[l] = preact[l - 1]; }
    const ReduceJob lj = ReduceJob{lpart, loss, 1ull, 1ull, n_lpart, 0, 0, 1};      // (pad = 1: one number out of n_lpart partials)
    return mlp_backward_impl(acts, zs, W, dims, n_layers, B, g, dW, db, nullptr, ws, s, has_t ? t : nullptr, &lj, layer_done);
}

// ------------------------------------------------------------------- Adam ----
// torch.optim.Adam (amsgrad = False, maximize = False), single step on every tensor of the table:
//   g      = grad (+ weight_decay * p)
//   m      = lerp(m, g, 1 - beta1)            = m + (1 - beta1) * (g - m)
//   v      = v * beta2 + (1 - beta2) * g * g
//   denom  = sqrt(v) / sqrt(bias_correction2) + eps
//   p      = p - (lr / bias_correction1) * m / denom
// The scalar factors arrive as the fp32 roundings of the Python doubles torch computes them with.
struct AdamTable { float* p; const float* g; float* m; float* v; unsigned long long n; };

// (the contraction pattern below is the one that is bit-equal to torch's foreach kernels on ROCm 7 /
//  torch 2.10, found by probing the alternatives: tools/probe/adam_probe.py)
__global__ __launch_bounds__(256) void adam_multi(const AdamTable* __restrict__ tab, int n_tensors,
                                                  float w1, float beta2, float w2, float bc2_sqrt, float eps,
                                                  float step_size, float weight_decay, float grad_scale) {
    for (int q = blockIdx.y; q < n_tensors; q += gridDim.y) {
        const AdamTable T = tab[q];
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < T.n; e += (size_t)gridDim.x * 256) {
            float g = T.g[e];
            if (grad_scale != 1.f) {              // data parallel: the all-reduced SUM -> the mean (`grad.mul_(1 / world)`),
                g = __fmul_rn(g, grad_scale);     // folded into this launch; .grad is left holding the mean, as DDP leaves it
                const_cast<float*>(T.g)[e] = g;
            }
            const float p = T.p[e];
            if (weight_decay != 0.f) g = fmaf(weight_decay, p, g);
            float m = T.m[e], v = T.v[e];
            m = fmaf(w1, g - m, m);                                   // lerp
            v = fmaf(w2, __fmul_rn(g, g), __fmul_rn(v, beta2));       // mul_, then addcmul_
            const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
            T.m[e] = m; T.v[e] = v;
            T.p[e] = fmaf(-step_size, __fdiv_rn(m, denom), p);        // addcdiv_
        }
    }
}

// table: device array of n_tensors AdamTable records {param, grad, exp_avg, exp_avg_sq, numel}
extern "C" int cfm_adam_step_f32(const void* table, int n_tensors, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, int step, double grad_scale, void* stream) {
    if (!table || n_tensors < 0 || step < 1 || !(grad_scale > 0.0)) return CFM_EINVAL;
    if (n_tensors == 0) return 0;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1, bc2_sqrt = sqrt(bc2);
    hipLaunchKernelGGL(adam_multi, dim3(256, n_tensors < 16 ? n_tensors : 16), dim3(256), 0, (hipStream_t)stream,
                       (const AdamTable*)table, n_tensors, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)bc2_sqrt, (float)eps, (float)step_size, (float)weight_decay, (float)grad_scale);
    return cfm_status();
}
