// sample.hip — K6: draw (i, j) index pairs from the OT plan on the device.
//
// Replaces OTPlanSampler.sample_map (torchcfm/optimal_transport.py:116-121):
//   p = pi.flatten(); p = p / p.sum();
//   choices = np.random.choice(B0*B1, p=p, size=n)   # cdf = cumsum(p); cdf /= cdf[-1]
//   i, j = divmod(choices, B1)                        # searchsorted(cdf, u, side="right")
// np.random.choice consumes exactly n doubles from the legacy global RNG; the
// host draws them (np.random.random_sample) and hands them over as `u01`, so
// the RNG stream and the sampled indices match the reference.
//
//  * permutation plan (exact OT): cdf steps at k/B -> i = floor(u*B), j = perm[i]
//    (exact for power-of-two B; for other B identical unless u lies within a few
//    ulp of a step, probability < 1e-12 per draw).
//  * dense plan (Sinkhorn): never materialised.  fp64 row sums of
//    exp(u_i + v_j - M_ij/reg) (one wave per row), block scan over rows, then one
//    wave per draw: binary search of the row, segmented in-row scan, walk.
#include "cfm_common.h"

extern "C" size_t cfm_sk_ws_bytes_internal(int B0, int B1);

__global__ __launch_bounds__(256) void sample_perm_kernel(const int* __restrict__ perm,
                                                          const double* __restrict__ u01, int B,
                                                          int n, int64_t* __restrict__ oi,
                                                          int64_t* __restrict__ oj) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double u = u01[k];
    long long ii = (long long)floor(u * (double)B);
    if (ii < 0) ii = 0;
    if (ii > B - 1) ii = B - 1;
    // first index m with cdf_m = (m+1)/B > u
    if (ii < B - 1 && u >= (double)(ii + 1) / (double)B) ++ii;
    else if (ii > 0 && u < (double)ii / (double)B) --ii;
    oi[k] = ii;
    oj[k] = (int64_t)perm[ii];
}

extern "C" int cfm_plan_sample_perm(const int* perm, const double* u01, int B, int n, int64_t* i,
                                    int64_t* j, void* stream) {
    if (!perm || !u01 || !i || !j || B <= 0 || n < 0) return CFM_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sample_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       perm, u01, B, n, i, j);
    return cfm_status();
}

// ------------------------------------------------------------- dense plan ----
struct PlanFromPotentials {
    const float* M; const double* u; const double* v; double inv_reg; int B1;
    __device__ __forceinline__ double operator()(int i, int j) const {
        return exp(u[i] + v[j] - (double)M[(size_t)i * B1 + j] * inv_reg);
    }
};
struct PlanFromPi {
    const double* pi; int B1;
    __device__ __forceinline__ double operator()(int i, int j) const { return pi[(size_t)i * B1 + j]; }
};

template <class P>
__global__ __launch_bounds__(256) void sd_rowsum(P plan, int B0, int B1, double* __restrict__ rs) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B0) return;
    double acc = 0.0;
    for (int j = lane; j < B1; j += 64) acc += plan(i, j);
    acc = wave_sum_d(acc);
    if (lane == 0) rs[i] = acc;
}

// inclusive scan of rs[0..B0) in place (single workgroup, 1024 threads)
__global__ __launch_bounds__(1024) void sd_rowscan(double* __restrict__ rs, int B0) {
    __shared__ double wsum[17];
    __shared__ double carry;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (int i0 = 0; i0 < B0; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        double x = (i < B0) ? rs[i] : 0.0;
        double inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            double t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double run = carry;
            for (int q = 0; q < 16; ++q) { double t = wsum[q]; wsum[q] = run; run += t; }
            wsum[16] = run;
        }
        __syncthreads();
        if (i < B0) rs[i] = inc + wsum[wv];
        __syncthreads();
        if (threadIdx.x == 0) carry = wsum[16];
        __syncthreads();
    }
}

template <class P>
__global__ __launch_bounds__(256) void sd_sample(P plan, int B0, int B1,
                                                 const double* __restrict__ rinc,
                                                 const double* __restrict__ u01, int n,
                                                 int64_t* __restrict__ oi,
                                                 int64_t* __restrict__ oj) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    const double T = rinc[B0 - 1];
    const double target = u01[k] * T;
    // first row whose inclusive prefix exceeds target (uniform across the wave)
    int lo = 0, hi = B0 - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rinc[mid] > target) hi = mid; else lo = mid + 1;
    }
    const int i = lo;
    const double base = (i > 0) ? rinc[i - 1] : 0.0;
    // segmented scan: lane owns columns [lane*L, lane*L + L)
    const int L = (B1 + 63) / 64;
    const int jb = lane * L, je = min(B1, jb + L);
    double seg = 0.0;
    for (int j = jb; j < je; ++j) seg += plan(i, j);
    double inc = seg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const bool hit = (base + inc) > target;
    const unsigned long long mask = __ballot(hit);
    int jout = B1 - 1;
    if (mask != 0ull) {
        const int src = __ffsll((long long)mask) - 1;
        if (lane == src) {
            double run = base + (inc - seg);
            int jj = je - 1;
            for (int j = jb; j < je; ++j) {
                run += plan(i, j);
                if (run > target) { jj = j; break; }
            }
            jout = jj;
        }
        jout = __shfl(jout, src, 64);
    } else {
        // rounding put the target past this row's mass: last column carrying mass
        int jl = -1;
        for (int j = je - 1; j >= jb; --j) if (plan(i, j) > 0.0) { jl = j; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) jl = max(jl, __shfl_xor(jl, o, 64));
        jout = jl >= 0 ? jl : B1 - 1;
    }
    if (lane == 0) { oi[k] = i; oj[k] = jout; }
}

extern "C" size_t cfm_sd_ws_bytes_internal(int B0, int B1) {
    return sizeof(double) * ((size_t)B0 + 8 + (size_t)B1) + 64;
}

template <class P>
static int sd_run(P plan, int B0, int B1, const double* u01, int n, int64_t* i, int64_t* j, void* ws,
                  hipStream_t s) {
    double* rs = (double*)ws;
    hipLaunchKernelGGL(sd_rowsum<P>, dim3((B0 + 3) / 4), dim3(256), 0, s, plan, B0, B1, rs);
    hipLaunchKernelGGL(sd_rowscan, dim3(1), dim3(1024), 0, s, rs, B0);
    if (n > 0)
        hipLaunchKernelGGL(sd_sample<P>, dim3((n + 3) / 4), dim3(256), 0, s, plan, B0, B1, rs, u01, n, i, j);
    return cfm_status();
}

// layout of the Sinkhorn workspace (must match sinkhorn.hip)
struct SkStateView { int done, iters_done, vfinal, precise; double err2[2]; double last_err; };

__global__ void sd_pick_v(const SkStateView* st, const double* v0, const double* v1, int B1,
                          double* vout) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < B1) vout[j] = st->vfinal ? v1[j] : v0[j];
}

extern "C" int cfm_plan_sample_dense(const float* M, int B0, int B1, double reg, const void* sk_ws,
                                     const double* u01, int n, int64_t* i, int64_t* j, void* ws,
                                     void* stream) {
    if (!M || !sk_ws || !ws || B0 <= 0 || B1 <= 0 || n < 0 || !(reg > 0.0)) return CFM_EINVAL;
    if (n > 0 && (!u01 || !i || !j)) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // ws: [row sums B0][v copy B1]
    double* rs = (double*)ws;
    double* vsel = rs + ((size_t)B0 + 8);
    const char* q = (const char*)sk_ws;
    const SkStateView* st = (const SkStateView*)q;
    const double* u = (const double*)(q + 256);
    const double* v0 = u + B0;
    const double* v1 = v0 + B1;
    hipLaunchKernelGGL(sd_pick_v, dim3((B1 + 255) / 256), dim3(256), 0, s, st, v0, v1, B1, vsel);
    PlanFromPotentials plan{M, u, vsel, 1.0 / reg, B1};
    return sd_run(plan, B0, B1, u01, n, i, j, ws, s);
}

extern "C" int cfm_plan_sample_pi_f64(const double* pi, int B0, int B1, const double* u01, int n,
                                      int64_t* i, int64_t* j, void* ws, void* stream) {
    if (!pi || !ws || B0 <= 0 || B1 <= 0 || n < 0) return CFM_EINVAL;
    if (n > 0 && (!u01 || !i || !j)) return CFM_EINVAL;
    PlanFromPi plan{pi, B1};
    return sd_run(plan, B0, B1, u01, n, i, j, ws, (hipStream_t)stream);
}

// ------------------------------------------------- per-row draws (sample_trajectory) ----
// Replaces the inner loop of OTPlanSampler.sample_trajectory (torchcfm/optimal_transport.py:237-246):
//     for i in indices[-1]:  j.append(np.random.choice(pi.shape[1], p=pi[i] / pi[i].sum()))
// one draw per entry of `rows` from the CONDITIONAL distribution of that plan row: np.random.choice builds
// cdf = cumsum(p), cdf /= cdf[-1] and returns searchsorted(cdf, u, side="right") — the first column whose running sum
// exceeds u x (row total).  One wave per draw, reading the plan row where it lies (potentials + cost row, or an fp64
// plan): no [m, B1] sub-plan, no row-normalised copy, and the chain of slices keeps its indices on the device.
template <class P>
__global__ __launch_bounds__(256) void sd_sample_rows(P plan, int B0, int B1, const int64_t* __restrict__ rows,
                                                      const double* __restrict__ u01, int n,
                                                      int64_t* __restrict__ oj) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n) return;
    long long r = rows[k];
    if (r < 0) r = 0;
    if (r > B0 - 1) r = B0 - 1;
    const int i = (int)r;
    const int L = (B1 + 63) / 64;
    const int jb = lane * L, je = min(B1, jb + L);
    double seg = 0.0;
    for (int j = jb; j < je; ++j) seg += plan(i, j);
    double inc = seg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const double T = __shfl(inc, 63, 64);                 // the row total, as the scan sees it
    const double target = u01[k] * T;
    const bool hit = inc > target;
    const unsigned long long mask = __ballot(hit);
    int jout = B1 - 1;
    if (mask != 0ull) {
        const int src = __ffsll((long long)mask) - 1;
        if (lane == src) {
            double run = inc - seg;
            int jj = je - 1;
            for (int j = jb; j < je; ++j) {
                run += plan(i, j);
                if (run > target) { jj = j; break; }
            }
            jout = jj;
        }
        jout = __shfl(jout, src, 64);
    } else {
        // (a row without mass, or a NaN total: numpy raises there; return the last column carrying mass)
        int jl = -1;
        for (int j = je - 1; j >= jb; --j) if (plan(i, j) > 0.0) { jl = j; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) jl = max(jl, __shfl_xor(jl, o, 64));
        jout = jl >= 0 ? jl : B1 - 1;
    }
    if (lane == 0) oj[k] = jout;
}

extern "C" int cfm_plan_sample_rows_dense(const float* M, int B0, int B1, double reg, const void* sk_ws,
                                          const int64_t* rows, const double* u01, int n, int64_t* j, void* ws,
                                          void* stream) {
    if (!M || !sk_ws || !ws || B0 <= 0 || B1 <= 0 || n < 0 || !(reg > 0.0)) return CFM_EINVAL;
    if (n == 0) return 0;
    if (!rows || !u01 || !j) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double* vsel = (double*)ws + ((size_t)B0 + 8);        // same carving as cfm_plan_sample_dense
    const char* q = (const char*)sk_ws;
    const SkStateView* st = (const SkStateView*)q;
    const double* u = (const double*)(q + 256);
    const double* v0 = u + B0;
    const double* v1 = v0 + B1;
    hipLaunchKernelGGL(sd_pick_v, dim3((B1 + 255) / 256), dim3(256), 0, s, st, v0, v1, B1, vsel);
    PlanFromPotentials plan{M, u, vsel, 1.0 / reg, B1};
    hipLaunchKernelGGL(sd_sample_rows<PlanFromPotentials>, dim3((n + 3) / 4), dim3(256), 0, s, plan, B0, B1, rows, u01, n, j);
    return cfm_status();
}

extern "C" int cfm_plan_sample_rows_pi_f64(const double* pi, int B0, int B1, const int64_t* rows, const double* u01,
                                           int n, int64_t* j, void* stream) {
    if (!pi || B0 <= 0 || B1 <= 0 || n < 0) return CFM_EINVAL;
    if (n == 0) return 0;
    if (!rows || !u01 || !j) return CFM_EINVAL;
    PlanFromPi plan{pi, B1};
    hipLaunchKernelGGL(sd_sample_rows<PlanFromPi>, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, plan, B0, B1,
                       rows, u01, n, j);
    return cfm_status();
}

// zero entries of a flat fp64 plan (replace=False bookkeeping of np.random.choice)
__global__ void zero_flat_kernel(double* pi, const int64_t* flat, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) pi[flat[k]] = 0.0;
}
extern "C" int cfm_plan_zero_entries_f64(double* pi, const int64_t* flat, int n, void* stream) {
    if (!pi || (n > 0 && !flat) || n < 0) return CFM_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(zero_flat_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pi, flat, n);
    return cfm_status();
}
