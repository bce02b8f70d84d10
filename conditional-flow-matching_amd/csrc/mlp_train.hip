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
// All GEMMs are v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak), LDS tiles laid out so that
// every fragment read is a conflict-free ds_read_b32 whatever the operand's storage order:
//   operand stored k-contiguous ([row][k], e.g. activations as the A of dgrad): tile [rows][BK + 1]
//   operand stored k-major      ([k][row], e.g. W as the B of dgrad, both operands of wgrad): tile [BK][rows]
#include "cfm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SELU_SCALE 1.0507009873554805f
#define SELU_ALPHA 1.6732632423543772f

// selu'(z)
__device__ __forceinline__ float selu_grad(float z) {
    return z > 0.f ? SELU_SCALE : (SELU_SCALE * SELU_ALPHA) * expf(z);
}

enum { EPI_PLAIN = 0, EPI_SELU_GRAD = 1 };

// C[M,N] (+ epilogue) = A[M,Kc] . B[Kc,N], contraction over [k_begin, k_end) of this workgroup's split.
//   A_KMAJOR = false: A(i,k) = A[i * lda + k]      true: A(i,k) = A[k * lda + i]
//   B_KMAJOR = false: B(k,j) = Bm[j * ldb + k]     true: B(k,j) = Bm[k * ldb + j]
// grid: x = tiles_m * tiles_n (XCD-remapped), y = split index s; split s writes C + s * split_stride.
template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const float* __restrict__ A, int lda,
                                                     const float* __restrict__ Bm, int ldb,
                                                     float* __restrict__ C, int ldc, size_t split_stride,
                                                     const float* __restrict__ H,     // EPI_SELU_GRAD: pre-activations [M, ldc]
                                                     int M, int N, int Kc, int k_chunk, int tiles_n,
                                                     float* __restrict__ colsum = nullptr) {   // A_KMAJOR: sum_k A(i, k) per split -> colsum[s * M + i]
    constexpr int BK = 32;
    constexpr int LDA = A_KMAJOR ? BM : BK + 1, LDB = B_KMAJOR ? BN : BK + 1;
    constexpr int WM = BM / 2, WN = BN / 2;          // per-wave tile (2x2 waves)
    constexpr int MT = WM / 32, NT = WN / 32;        // 32x32 MFMA tiles per wave
    constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;
    __shared__ float As[A_KMAJOR ? BK * LDA : BM * LDA];
    __shared__ float Bs[B_KMAJOR ? BK * LDB : BN * LDB];

    const unsigned lid = cfm_xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const int k_begin = blockIdx.y * k_chunk;
    const int k_end = (k_begin + k_chunk < Kc) ? k_begin + k_chunk : Kc;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[A_PER], rb[B_PER];
    // thread owns elements e = tid + 256 q of a stage; consecutive lanes walk the operand's contiguous
    // dimension (coalesced global rows) and consecutive LDS addresses
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            const int e = tid + 256 * q;
            const int r = A_KMAJOR ? e % BM : e / BK, k = A_KMAJOR ? e / BM : e % BK;
            const int gr = row0 + r, gk = k0 + k;
            ra[q] = (gr < M && gk < k_end) ? (A_KMAJOR ? A[(size_t)gk * lda + gr] : A[(size_t)gr * lda + gk]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            const int e = tid + 256 * q;
            const int r = B_KMAJOR ? e % BN : e / BK, k = B_KMAJOR ? e / BN : e % BK;
            const int gc = col0 + r, gk = k0 + k;
            rb[q] = (gc < N && gk < k_end) ? (B_KMAJOR ? Bm[(size_t)gk * ldb + gc] : Bm[(size_t)gc * ldb + gk]) : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            const int e = tid + 256 * q;
            if (A_KMAJOR) As[(e / BM) * LDA + (e % BM)] = ra[q]; else As[(e / BK) * LDA + (e % BK)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            const int e = tid + 256 * q;
            if (B_KMAJOR) Bs[(e / BN) * LDB + (e % BN)] = rb[q]; else Bs[(e / BK) * LDB + (e % BK)] = rb[q];
        }
    };

    // bias gradient rides along: the workgroups of the first tile column add up their A tile (dz) over k
    const bool do_colsum = A_KMAJOR && colsum != nullptr && tn == 0 && tid < BM;
    float csum = 0.f;
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        stash();
        __syncthreads();
        if (k0 + BK < k_end) fetch(k0 + BK);            // in flight while the MFMAs run
        if (A_KMAJOR && do_colsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += As[kk * LDA + tid];
        }
        const int fr = lane & 31, fk = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[MT], b[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int i = wm * WM + m * 32 + fr;
                a[m] = A_KMAJOR ? As[(kk + fk) * LDA + i] : As[i * LDA + kk + fk];
            }
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                const int j = wn * WN + nn * 32 + fr;
                b[nn] = B_KMAJOR ? Bs[(kk + fk) * LDB + j] : Bs[j * LDB + kk + fk];
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    acc[m][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[nn], acc[m][nn], 0, 0, 0);
        }
        __syncthreads();
    }
    if (A_KMAJOR && do_colsum && row0 + tid < M) colsum[(size_t)blockIdx.y * M + row0 + tid] = csum;
    // epilogue.  C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cs = C + (size_t)blockIdx.y * split_stride;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        const int gc = col0 + wn * WN + nn * 32 + (lane & 31);
        if (gc >= N) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gr = row0 + wm * WM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gr >= M) continue;
                float v = acc[m][nn][r];
                if (EPI == EPI_SELU_GRAD) v *= selu_grad(H[(size_t)gr * ldc + gc]);
                Cs[(size_t)gr * ldc + gc] = v;
            }
        }
    }
}

// out[e] = sum_s partial[s * stride + e] in split order (deterministic), for every tensor of the table
// (all weight and bias gradients of a backward pass in ONE launch)
struct ReduceJob { const float* partial; float* out; unsigned long long stride, n; int S, pad; };
#define MLP_MAX_LAYERS 16
struct ReduceTable { ReduceJob job[2 * MLP_MAX_LAYERS]; int count; };

__global__ __launch_bounds__(256) void reduce_splits_multi(ReduceTable T) {
    for (int q = blockIdx.y; q < T.count; q += gridDim.y) {
        const ReduceJob J = T.job[q];
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < J.n; e += (size_t)gridDim.x * 256) {
            float v = J.partial[e];
            for (int s = 1; s < J.S; ++s) v += J.partial[(size_t)s * J.stride + e];
            J.out[e] = v;
        }
    }
}

#define MLP_MAX_SPLITS 32
extern "C" size_t cfm_mlp_train_ws_bytes_internal(int B, int maxw, int max_params) {
    // two [B, maxw] gradient buffers + per-layer split-K partials (weights and biases; <= MLP_MAX_LAYERS layers
    // are sized here by the largest one: callers pass the largest dims[l] * dims[l+1])
    return sizeof(float) * ((size_t)2 * B * maxw + (size_t)MLP_MAX_SPLITS * ((size_t)max_params + maxw) * 4) + 1024;
}

template <bool AK, bool BK_, int EPI>
static int launch_gemm(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, size_t split_stride,
                       const float* H, int M, int N, int Kc, int S, hipStream_t s, float* colsum = nullptr) {
    int k_chunk = (Kc + S - 1) / S;
    k_chunk = (k_chunk + 31) / 32 * 32;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    if (t128 * S >= 192) {
        const int tm = (M + 127) / 128, tn = (N + 127) / 128;
        hipLaunchKernelGGL((gemm_f32_mfma<128, 128, AK, BK_, EPI>), dim3(tm * tn, S), dim3(256), 0, s, A, lda, Bm, ldb, C, ldc,
                           split_stride, H, M, N, Kc, k_chunk, tn, colsum);
    } else {
        const int tm = (M + 63) / 64, tn = (N + 63) / 64;
        hipLaunchKernelGGL((gemm_f32_mfma<64, 64, AK, BK_, EPI>), dim3(tm * tn, S), dim3(256), 0, s, A, lda, Bm, ldb, C, ldc,
                           split_stride, H, M, N, Kc, k_chunk, tn, colsum);
    }
    return cfm_status();
}

// Backward through all layers.  acts[l] = h_l (l = 0: the network input [B, dims[0]]; l = 1 .. n-1: the
// saved hidden activations), preact[l] = z_l for l = 1 .. n-1 (preact[0] unused); dout [B, dims[n]].
// Writes dW[l] ([dims[l+1], dims[l]]), db[l] and, if dx is not NULL, the input gradient [B, dims[0]].
// Launches: per layer one wgrad (bias column sums ride along) and one dgrad, then ONE reduction of every
// split-K partial (weights and biases of all layers).
extern "C" int cfm_mlp_backward_f32(const float* const* acts, const float* const* preact, const float* const* W,
                                    const int* dims, int n_layers, int B, const float* dout, float* const* dW,
                                    float* const* db, float* dx, void* ws, void* stream) {
    if (!acts || !W || !dims || !dout || !dW || !db || n_layers < 1 || n_layers > MLP_MAX_LAYERS || B < 0 || !ws) return CFM_EINVAL;
    if (n_layers > 1 && !preact) return CFM_EINVAL;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
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
        const int K = dims[l], N = dims[l + 1];
        // wgrad: dW[N,K] = dz^T[N,B] . h[B,K], contraction over the batch, S splits; db partials ride along
        const long tiles = (long)((N + 127) / 128) * ((K + 127) / 128);
        int S = 1;
        while (S < MLP_MAX_SPLITS && tiles * S < 256 && B / (2 * S) >= 64) S *= 2;
        const size_t np = (size_t)N * K;
        if (used + (size_t)S * (np + N) > pool_floats) return CFM_EINVAL;      // more layers than the workspace was sized for
        float* part = pool + used; used += (size_t)S * np;
        float* bpart = pool + used; used += (size_t)S * N;
        int rc = launch_gemm<true, true, EPI_PLAIN>(dz, N, acts[l], K, part, K, np, nullptr, N, K, B, S, s, bpart);
        if (rc) return rc;
        T.job[T.count++] = ReduceJob{part, dW[l], np, np, S, 0};
        T.job[T.count++] = ReduceJob{bpart, db[l], (unsigned long long)N, (unsigned long long)N, S, 0};
        // dgrad: dz_prev[B,K] = (dz[B,N] . W[N,K]) * selu'(z_prev)
        if (l > 0) {
            float* dst = gbuf[l & 1];
            rc = launch_gemm<false, true, EPI_SELU_GRAD>(dz, N, W[l], K, dst, K, 0, preact[l], B, K, N, 1, s);
            if (rc) return rc;
            dz = dst;
        } else if (dx) {
            rc = launch_gemm<false, true, EPI_PLAIN>(dz, N, W[0], K, dx, K, 0, nullptr, B, K, N, 1, s);
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(reduce_splits_multi, dim3(256, T.count), dim3(256), 0, s, T);
    return cfm_status();
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
                                                  float step_size, float weight_decay) {
    for (int q = blockIdx.y; q < n_tensors; q += gridDim.y) {
        const AdamTable T = tab[q];
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < T.n; e += (size_t)gridDim.x * 256) {
            float g = T.g[e];
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
                                 double weight_decay, int step, void* stream) {
    if (!table || n_tensors < 0 || step < 1) return CFM_EINVAL;
    if (n_tensors == 0) return 0;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1, bc2_sqrt = sqrt(bc2);
    hipLaunchKernelGGL(adam_multi, dim3(256, n_tensors < 16 ? n_tensors : 16), dim3(256), 0, (hipStream_t)stream,
                       (const AdamTable*)table, n_tensors, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)bc2_sqrt, (float)eps, (float)step_size, (float)weight_decay);
    return cfm_status();
}
