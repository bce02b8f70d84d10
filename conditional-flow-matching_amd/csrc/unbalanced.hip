// unbalanced.hip — OTPlanSampler(method="unbalanced" | "partial") on gfx950.
//
// Replaces, for uniform marginals a = 1/B0, b = 1/B1 (torchcfm/optimal_transport.py:79),
//   pot.unbalanced.sinkhorn_knopp_unbalanced(a, b, M, reg, reg_m)     optimal_transport.py:52-53,87
//   pot.partial.entropic_partial_wasserstein(a, b, M, reg)            optimal_transport.py:54-55,87
// Both are kernel-space (not log-domain) float64 algorithms in POT, and the reference's wrapper
// semantics depend on exactly that (underflow of exp(-M/reg) -> "numerical errors" -> previous
// iterate / uniform-plan fallback, :88-96), so they are kept in kernel space here: the Gibbs
// kernel K = exp(M / -reg) is built ONCE in fp64 (128 MiB at B = 4096: nothing on a 288 GB
// part) and every iteration is two streaming fp64 passes over it,
//     rowdot:  y_i = sum_j K_ij x_j      one wave per row, 16 B loads
//     coldot:  z_j = sum_i K_ij w_i      lane <-> column strips + finalize
// i.e. HBM-bound at 2 * 8 * B0 * B1 bytes per iteration.
//
// Unbalanced follows the loop of the in-repo statement of POT's algorithm
// (runner/src/models/components/sinkhorn_knopp_unbalanced.py:134-186 with reg_m_1 = reg_m_2):
//     u = (a / K v)^fi ; v = (b / K^T u)^fi ; fi = reg_m / (reg_m + reg)
//     numerical error (K^T u == 0, nan / inf) -> previous (u, v), stop
//     every 10th iteration: err = (|u-u'|_inf / max(|u|_inf,|u'|_inf,1) + same for v) / 2 <= stopThr
// Partial is POT's Dykstra loop; its three B0 x B1 correction matrices q1, q2, q3 are constant
// along rows / columns / everywhere (q1 <- q1 * Kprev / K1 = 1 / r_i, ...), so the iteration is
// carried on two scaling vectors and a scalar: K = diag(alpha) K0 diag(beta).
#include "cfm_common.h"

#define UB_NCHUNK_MAX 64

struct UbState {
    int done, iters, status, final_idx;   // status: 0 ok, 1 numerical error (previous iterate returned)
    int flag_bad, n_zero, n_nonfinite, pad;
    double err;
    double sumK;                // sum of the Gibbs kernel (partial: K *= m / sumK)
    double sigma;               // partial: q3
    double total;               // partial: sum_j beta2_j * coldot_j
    unsigned long long mx[6];   // ordered-double maxima: |u-u'|, |u|, |u'|, |v-v'|, |v|, |v'|
    double err2;                // partial: || Kprev - K ||_F^2
    double fold[2];             // partial: scalar factor sigma * s that alpha (U[0] / U[1]) still has to be multiplied by
    unsigned long long amax;    // partial: ordered-double max |alpha2| of the iteration (finiteness of alpha2 * fold)
    unsigned tiles_done;        // fused column kernel: arrival word
    unsigned pad2;
};
static_assert(sizeof(UbState) <= 512, "UbState must fit the 512 bytes carved for it");

struct UbWs {
    UbState* st;
    double* U[2];     // unbalanced: u ping-pong   | partial: alpha, alpha2
    double* V[2];     // unbalanced: v ping-pong   | partial: beta, beta2
    double* rho;      // partial q1 (rows)
    double* kappa;    // partial q2 (cols)
    double* rowacc;   // rowdot result
    double* colacc;   // coldot result
    double* part;     // [nchunk][B1]
};

static inline int ub_nchunk(int B0, int B1) {
    int tiles = (B1 + 255) / 256;
    int n = (1024 + tiles - 1) / tiles;
    if (n > UB_NCHUNK_MAX) n = UB_NCHUNK_MAX;
    int maxn = (B0 + 31) / 32;
    if (n > maxn) n = maxn;
    return n < 1 ? 1 : n;
}

static inline size_t ub_ws_bytes(int B0, int B1) {
    size_t nchunk = ub_nchunk(B0, B1);
    return 512 + sizeof(double) * (4 * (size_t)B0 + 5 * (size_t)B1 + nchunk * (size_t)B1) + 256;
}

static inline UbWs ub_carve(void* ws, int B0, int B1) {
    UbWs w; char* q = (char*)ws;
    w.st = (UbState*)q; q += 512;
    w.U[0] = (double*)q; q += 8 * (size_t)B0;
    w.U[1] = (double*)q; q += 8 * (size_t)B0;
    w.rho = (double*)q; q += 8 * (size_t)B0;
    w.rowacc = (double*)q; q += 8 * (size_t)B0;
    w.V[0] = (double*)q; q += 8 * (size_t)B1;
    w.V[1] = (double*)q; q += 8 * (size_t)B1;
    w.kappa = (double*)q; q += 8 * (size_t)B1;
    w.colacc = (double*)q; q += 8 * (size_t)B1;
    w.part = (double*)q;
    return w;
}

extern "C" size_t cfm_ub_ws_bytes_internal(int B0, int B1) { return ub_ws_bytes(B0, B1); }

// ---------------------------------------------------------------- Gibbs kernel
// K = exp(M / -reg) in fp64 (np.divide(M, -reg, out=K); np.exp(K, out=K), runner/...:139-141),
// its sum, and how many entries underflowed to 0 / are not finite.
__global__ __launch_bounds__(256) void ub_gibbs(const float* __restrict__ M, size_t n, double neg_reg,
                                                double* __restrict__ K, UbState* st) {
    double acc = 0.0; int nz = 0, nf = 0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const double v = exp((double)M[k] / neg_reg);
        K[k] = v;
        acc += v;
        nz += (v == 0.0) ? 1 : 0;
        nf += (isfinite(v)) ? 0 : 1;
    }
    acc = wave_sum_d(acc); nz = wave_sum_i(nz); nf = wave_sum_i(nf);
    __shared__ double sd[4]; __shared__ int sz[4], sf[4];
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = acc; sz[threadIdx.x >> 6] = nz; sf[threadIdx.x >> 6] = nf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&st->sumK, sd[0] + sd[1] + sd[2] + sd[3]);
        const int z = sz[0] + sz[1] + sz[2] + sz[3], f = sf[0] + sf[1] + sf[2] + sf[3];
        if (z) atomicAdd(&st->n_zero, z);
        if (f) atomicAdd(&st->n_nonfinite, f);
    }
}

__global__ void ub_state_init(UbState* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->done = 0; st->iters = 0; st->status = 0; st->final_idx = 0;
        st->flag_bad = 0; st->n_zero = 0; st->n_nonfinite = 0;
        st->err = 1.0; st->sumK = 0.0; st->sigma = 1.0; st->total = 0.0; st->err2 = 0.0;
        st->fold[0] = 1.0; st->fold[1] = 1.0; st->amax = 0ull; st->tiles_done = 0u;
        for (int k = 0; k < 6; ++k) st->mx[k] = 0ull;
    }
}


// partial column sums over a strip of rows: part[chunk][j] = sum_{i in strip} w_i K_ij
__global__ __launch_bounds__(256) void ub_coldot(const double* __restrict__ K, int B0, int B1,
                                                 const UbState* __restrict__ st,
                                                 const double* __restrict__ w,
                                                 double* __restrict__ part, int rows_per_chunk) {
    if (st->done) return;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(B0, r0 + rows_per_chunk);
    double acc = 0.0;
    if (j < B1) {
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
            double k8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) k8[q] = K[(size_t)(r + q) * B1 + j];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += k8[q] * w[r + q];
        }
        for (; r < r1; ++r) acc += K[(size_t)r * B1 + j] * w[r];
        part[(size_t)blockIdx.y * B1 + j] = acc;
    }
}

__device__ __forceinline__ void ub_atomic_max_abs(unsigned long long* slot, double v) {
    // |v| >= 0: the bit pattern orders like the value; NaN never wins (flagged separately)
    atomicMax(slot, (unsigned long long)__double_as_longlong(fabs(v)));
}




// plan = u_i K_ij v_j  (in place)
__global__ __launch_bounds__(256) void ub_plan(double* __restrict__ K, int B0, int B1,
                                               const UbState* __restrict__ st,
                                               const double* __restrict__ u0, const double* __restrict__ u1,
                                               const double* __restrict__ v0, const double* __restrict__ v1,
                                               int nan_if_zero) {
    const double* u = st->final_idx ? u1 : u0;
    const double* v = st->final_idx ? v1 : v0;
    const bool poison = nan_if_zero && (st->n_zero > 0 || st->n_nonfinite > 0);
    const size_t n = (size_t)B0 * B1;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const int i = (int)(k / B1), j = (int)(k - (size_t)i * B1);
        K[k] = poison ? __longlong_as_double(0x7ff8000000000000ll) : u[i] * K[k] * v[j];
    }
}

__global__ void ub_fill(double* x, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

__global__ void ub_info(const UbState* st, int* info) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && info) {
        info[0] = st->iters; info[1] = st->status; info[2] = st->n_zero; info[3] = st->n_nonfinite;
    }
}

// ---------------------------------------------------------------- fused iteration (round 4)
// Round 3 ran an iteration as a chain of launches: two streaming passes and, around them, one small launch per
// elementwise update and per control decision — five ~4.8 us launches per partial iteration that did nothing wide
// (24 of 78 us), and a column update whose 16 workgroups summed 64 strip partials one dependent load at a time (21 us).
// Here an iteration is THREE launches:
//   ub_row_fused:  y_i = sum_j K_ij x_j and, by the wave that owns the row, the row update itself;
//   ub_coldot:     the strip partials of z_j = sum_i K_ij w_i (unchanged);
//   ub_col_fused:  per 64-column workgroup the sum of the strip partials (same order, all loads in flight) and the
//                  column update; the LAST workgroup to arrive (one device-scope ticket; everything it needs from the
//                  others — the running total, the flags, the maxima — travels through device-scope atomics, so no
//                  fence is paid) takes the scalar step and the loop's control decision.
// (The strip partials are NOT merged inside ub_coldot by a last-strip-done ticket: 1024 workgroups each writing back
//  and invalidating their L2 measured 124 us against 39 us for the same structure in the Sinkhorn column pass.)
// Same arithmetic in the same order as the launch-per-step loop (kept behind -DUB_FUSED=0).
__device__ __forceinline__ int ub_ld_i32(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ub_ld_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ub_ld_f64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ double ub_row_dot(const double* __restrict__ row, const double* __restrict__ x, int B1, int lane) {
    double acc = 0.0;
    if ((B1 & 1) == 0) {
        for (int j = lane * 2; j < B1; j += 128 * 4) {
            double2 k2[4], x2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = j + 128 * q;
                k2[q] = (jj < B1) ? *reinterpret_cast<const double2*>(row + jj) : make_double2(0.0, 0.0);
                x2[q] = (jj < B1) ? *reinterpret_cast<const double2*>(x + jj) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += k2[q].x * x2[q].x + k2[q].y * x2[q].y;
        }
    } else {
        for (int j = lane; j < B1; j += 64) acc += row[j] * x[j];
    }
    return wave_sum_d(acc);
}

// MODE 0 (unbalanced): u_new = (a / (K v)_i)^fi          MODE 1 (partial): the Dykstra row step
template <int MODE>
__global__ __launch_bounds__(256) void ub_row_fused(const double* __restrict__ K, int B0, int B1, UbState* st,
                                                    const double* __restrict__ x, double a, double fi,
                                                    const double* __restrict__ uprev, double* __restrict__ unew,
                                                    double* __restrict__ rho, int check, int p) {
    if (st->done) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= B0) return;
    // (the operands of the row update are requested before the row is streamed: no dependent load behind the dot)
    const double up = (MODE == 1) ? uprev[r] : 0.0;
    const double rh = (MODE == 1) ? rho[r] : 0.0;
    const double fo = (MODE == 1) ? st->fold[p] : 1.0;
    const double kv = ub_row_dot(K + (size_t)r * B1, x, B1, lane);
    if (lane != 0) return;
    if (MODE == 0) {
        const double u = pow(a / kv, fi);
        unew[r] = u;
        if (isnan(u) || isinf(u)) atomicOr(&st->flag_bad, 1);
        // (the maxima |u - u'|, |u|, |u'| of a check iteration are taken by the deciding wave of ub_col_fused from the
        //  two arrays: one device-scope atomic per ROW — 4096 waves, one lane each, on one address — cost 11 ns apiece
        //  at the memory side, 135 us per check launch; the launch-per-step kernel issued them 64 lanes at a time)
    } else {
        // alpha arrives with its scalar factor still pending (fold[p]: one multiplication, exactly the one pt_fold did)
        const double al = up * fo, a1 = al * rh;
        const double rowsum = a1 * kv;
        const double rr = fmin(a / rowsum, 1.0);
        const double a2 = rr * a1;
        unew[r] = a2;
        rho[r] = rh * al / a2;
        if (!isfinite(a2)) atomicOr(&st->flag_bad, 1);         // (max |alpha2|: by the deciding wave of ub_col_fused)
    }
}

// MODE 0 (unbalanced): v_new = (b / (K^T u)_j)^fi, then ub_decide      MODE 1 (partial): the Dykstra column step, the
// scalar step, the finiteness flags and (unless an error pass follows: defer_decide) pt_decide
#define UB_COLW 64       // columns per workgroup of the fused column kernel (one wave: B1 / 64 workgroups spread over the chip)
template <int MODE>
__global__ __launch_bounds__(UB_COLW) void ub_col_fused(int B1, int nchunk, UbState* st, const double* __restrict__ part,
                                                        double b, double fi, const double* __restrict__ vprev,
                                                        double* __restrict__ vnew, double* __restrict__ kappa, int check,
                                                        int cpt, int max_iter, double stop_thr, double m, int defer_decide,
                                                        const double* __restrict__ errbuf, int B0,
                                                        const double* __restrict__ uold, const double* __restrict__ unew) {
    if (st->done) return;
    const int j = blockIdx.x * UB_COLW + threadIdx.x;
    double contrib = 0.0;
    double cm0 = 0.0, cm1 = 0.0, cm2 = 0.0;                 // unbalanced, check iterations: this wave's column maxima
    if (j < B1) {
        // every strip partial of the column is requested at once (nchunk <= UB_NCHUNK_MAX = 64), then summed in strip
        // order as before: round 3 walked them one dependent load at a time in 16 workgroups (21 us per iteration)
        const double be = vprev[j];
        const double ka = (MODE == 1) ? kappa[j] : 0.0;
        double pv[UB_NCHUNK_MAX];
#pragma unroll
        for (int c = 0; c < UB_NCHUNK_MAX; ++c) pv[c] = part[(size_t)min(c, nchunk - 1) * B1 + j];     // (clamped: back-to-back requests)
        double kt = 0.0;
#pragma unroll
        for (int c = 0; c < UB_NCHUNK_MAX; ++c) if (c < nchunk) kt += pv[c];
        if (MODE == 0) {
            const double v = pow(b / kt, fi);
            vnew[j] = v;
            if (kt == 0.0 || isnan(v) || isinf(v)) atomicOr(&st->flag_bad, 1);
            if (check) { cm0 = fabs(v - be); cm1 = fabs(v); cm2 = fabs(be); }
        } else {
            const double b1 = be * ka;
            const double colsum = b1 * kt;
            const double cc = fmin(b / colsum, 1.0);
            const double b2 = cc * b1;
            vnew[j] = b2;
            kappa[j] = ka * be / b2;
            contrib = b2 * kt;
            if (!isfinite(b2)) atomicOr(&st->flag_bad, 1);
        }
    }
    if (MODE == 1) {
        contrib = wave_sum_d(contrib);
        if (threadIdx.x == 0 && contrib != 0.0) atomicAdd(&st->total, contrib);
    } else if (check) {                                      // one atomic per wave and maximum, not one per lane
        cm0 = wave_max_d(cm0); cm1 = wave_max_d(cm1); cm2 = wave_max_d(cm2);
        if (threadIdx.x == 0) { ub_atomic_max_abs(&st->mx[3], cm0); ub_atomic_max_abs(&st->mx[4], cm1); ub_atomic_max_abs(&st->mx[5], cm2); }
    }
    // arrival: the last workgroup (one wave) takes the control decision of the iteration.  What it needs from the other
    // workgroups of THIS launch travels through device-scope atomics (total, flags, column maxima); the row side it
    // reads from the arrays the row kernel wrote one launch earlier.
    // (the ticket is an acquire-release operation at device scope: the atomics above are ordered before it and the
    //  deciding wave's reads behind it by the memory model, not by the fields happening to share one 256-byte block)
    unsigned tt = 0u;
    if (threadIdx.x == 0) tt = __hip_atomic_fetch_add(&st->tiles_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    tt = (unsigned)__shfl((int)tt, 0, 64);
    if (tt != gridDim.x - 1u) return;
    const int lane = threadIdx.x;
    const int next = cpt + 1;
    if (MODE == 0) {
        double du = 0.0, mu = 0.0, mup = 0.0;
        if (check) {
            for (int i0 = lane; i0 < B0; i0 += 64 * 16) {    // 2 x 16 loads per lane in flight
                double uu[16], pp[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {              // (clamped index, no predicated load: the requests go out back to back;
                    const int i = min(i0 + 64 * q, B0 - 1);  //  an element read twice changes no maximum)
                    uu[q] = unew[i]; pp[q] = uold[i];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) { du = fmax(du, fabs(uu[q] - pp[q])); mu = fmax(mu, fabs(uu[q])); mup = fmax(mup, fabs(pp[q])); }
            }
            du = wave_max_d(du); mu = wave_max_d(mu); mup = wave_max_d(mup);
        }
        if (lane != 0) return;
        st->tiles_done = 0u;
        if (ub_ld_i32(&st->flag_bad)) {                      // u, v = uprev, vprev; break
            st->status = 1; st->final_idx = cpt & 1; st->iters = cpt; st->done = 1;
            return;
        }
        if (check) {
            const double dv = __longlong_as_double((long long)ub_ld_u64(&st->mx[3])), mv = __longlong_as_double((long long)ub_ld_u64(&st->mx[4])),
                         mvp = __longlong_as_double((long long)ub_ld_u64(&st->mx[5]));
            const double err_u = du / fmax(fmax(mu, mup), 1.0);
            const double err_v = dv / fmax(fmax(mv, mvp), 1.0);
            st->err = 0.5 * (err_u + err_v);
            for (int k = 0; k < 6; ++k) st->mx[k] = 0ull;
        }
        st->iters = next; st->final_idx = next & 1;
        if (!(st->err > stop_thr) || next >= max_iter) st->done = 1;
    } else {
        double am = 0.0;                                     // largest |alpha2| of the iteration (non-finite ones are flagged)
        for (int i0 = lane; i0 < B0; i0 += 64 * 32) {        // 32 loads per lane in flight
            double uu[32];
#pragma unroll
            for (int q = 0; q < 32; ++q) uu[q] = unew[min(i0 + 64 * q, B0 - 1)];     // (clamped: back-to-back requests)
#pragma unroll
            for (int q = 0; q < 32; ++q) am = fmax(am, fabs(uu[q]));
        }
        am = wave_max_d(am);
        if (lane != 0) return;
        st->tiles_done = 0u;
        // pt_scalar: K = K2 * q3 * (m / sum(K2 * q3)); q3 <- 1 / s; the factor sigma * s stays pending on alpha2 = U[q]
        const double S = st->sigma * ub_ld_f64(&st->total);
        const double sc = m / S;
        const double fold = st->sigma * sc;
        st->fold[next & 1] = fold;
        st->sigma = 1.0 / sc;
        st->total = 0.0;
        // pt_flags: alpha2 * fold and beta2 finite (the largest |alpha2| times the common factor decides for every row)
        if (!isfinite(am * fold)) atomicOr(&st->flag_bad, 1);
        if (defer_decide) return;                            // an error pass follows, then pt_decide
        if (ub_ld_i32(&st->flag_bad)) { st->status = 1; st->final_idx = next & 1; st->iters = cpt; st->done = 1; return; }
        st->iters = next; st->final_idx = next & 1;
        if (!(*errbuf > stop_thr) || next >= max_iter) st->done = 1;
    }
}

// apply the scalar factors still pending on alpha (partial: before the plan is written)
__global__ void pt_fold_final(int B0, const UbState* st, double* __restrict__ u0, double* __restrict__ u1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B0) { u0[i] *= st->fold[0]; u1[i] *= st->fold[1]; }
}

extern "C" int cfm_unbalanced_sinkhorn_f64(const float* M, int B0, int B1, double reg, double reg_m,
                                           int max_iter, double stop_thr, double* plan, int* info,
                                           void* ws, void* stream) {
    if (!M || !plan || !ws || B0 <= 0 || B1 <= 0 || !(reg > 0.0) || !(reg_m > 0.0) || max_iter < 0)
        return CFM_EINVAL;
    if (((uintptr_t)ws & 15) != 0 || ((uintptr_t)plan & 15) != 0) return CFM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    UbWs w = ub_carve(ws, B0, B1);
    const size_t n = (size_t)B0 * B1;
    const int nchunk = ub_nchunk(B0, B1);
    const int rows_per_chunk = (B0 + nchunk - 1) / nchunk;
    const double a = 1.0 / B0, b = 1.0 / B1, fi = reg_m / (reg_m + reg);
    hipLaunchKernelGGL(ub_state_init, dim3(1), dim3(64), 0, s, w.st);
    hipLaunchKernelGGL(ub_gibbs, dim3(2048), dim3(256), 0, s, M, n, -reg, plan, w.st);
    hipLaunchKernelGGL(ub_fill, dim3((B0 + 255) / 256), dim3(256), 0, s, w.U[0], B0, 1.0 / B0);
    hipLaunchKernelGGL(ub_fill, dim3((B1 + 255) / 256), dim3(256), 0, s, w.V[0], B1, 1.0 / B1);
    for (int cpt = 0; cpt < max_iter; ++cpt) {
        const int check = (cpt % 10 == 0) ? 1 : 0;
        const int p = cpt & 1, q = p ^ 1;
        hipLaunchKernelGGL(ub_row_fused<0>, dim3((B0 + 3) / 4), dim3(256), 0, s, plan, B0, B1, w.st, w.V[p], a, fi,
                           w.U[p], w.U[q], (double*)nullptr, check, p);
        hipLaunchKernelGGL(ub_coldot, dim3((B1 + 255) / 256, nchunk), dim3(256), 0, s, plan, B0, B1, w.st,
                           w.U[q], w.part, rows_per_chunk);
        hipLaunchKernelGGL(ub_col_fused<0>, dim3((B1 + UB_COLW - 1) / UB_COLW), dim3(UB_COLW), 0, s, B1, nchunk, w.st, w.part, b, fi,
                           w.V[p], w.V[q], (double*)nullptr, check, cpt, max_iter, stop_thr, 0.0, 0, (const double*)nullptr,
                           B0, (const double*)w.U[p], (const double*)w.U[q]);
    }
    hipLaunchKernelGGL(ub_plan, dim3(2048), dim3(256), 0, s, plan, B0, B1, w.st, w.U[0], w.U[1], w.V[0], w.V[1], 0);
    hipLaunchKernelGGL(ub_info, dim3(1), dim3(64), 0, s, w.st, info);
    return cfm_status();
}

// ---------------------------------------------------------------- partial (Dykstra)
// K0 <- K * (m / sum(K));  alpha = beta = rho = kappa = 1
__global__ __launch_bounds__(256) void pt_scale(double* __restrict__ K, size_t n, double m, const UbState* st) {
    const double f = m / st->sumK;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) K[k] *= f;
}



// || diag(alpha) K0 diag(beta) - diag(alpha2) K0 diag(beta2) ||_F^2 (alpha2 already carries sigma * s)
__global__ __launch_bounds__(256) void pt_err(const double* __restrict__ K, int B0, int B1, UbState* st,
                                              const double* __restrict__ alpha, const double* __restrict__ beta,
                                              const double* __restrict__ alpha2, const double* __restrict__ beta2, int p) {
    if (st->done) return;
    const size_t n = (size_t)B0 * B1;
    // fused loop: the scalar factors are still pending on alpha = U[p] and alpha2 = U[p ^ 1] (one multiplication each,
    // the one pt_fold did before the value was stored); launch-per-step loop (p < 0): already applied
    const double fa = p >= 0 ? st->fold[p] : 1.0, fb = p >= 0 ? st->fold[p ^ 1] : 1.0;
    double acc = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const int i = (int)(k / B1), j = (int)(k - (size_t)i * B1);
        const double al = p >= 0 ? alpha[i] * fa : alpha[i], al2 = p >= 0 ? alpha2[i] * fb : alpha2[i];
        const double d = K[k] * (al * beta[j] - al2 * beta2[j]);
        acc += d * d;
    }
    acc = wave_sum_d(acc);
    __shared__ double sd[4];
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&st->err2, sd[0] + sd[1] + sd[2] + sd[3]);
}


__global__ void pt_fold(int B0, const UbState* st, double* __restrict__ alpha2) {
    if (st->done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B0) {
        alpha2[i] *= st->fold[0];
    }
}


// loop control (POT entropic_partial_wasserstein): nan / inf in K -> warning, break (K kept);
// cpt % 10 == 0: err = ||Kprev - K||_F; `while err > stopThr and cpt < numItermax`.
__global__ void pt_decide(UbState* st, int cpt, int max_iter, double stop_thr, int check, double* errbuf) {
    if (st->done) return;
    const int next = cpt + 1;
    if (st->flag_bad) { st->status = 1; st->final_idx = next & 1; st->iters = cpt; st->done = 1; return; }
    double err = *errbuf;
    if (check) { err = sqrt(st->err2); *errbuf = err; st->err2 = 0.0; }
    st->iters = next; st->final_idx = next & 1;
    if (!(err > stop_thr) || next >= max_iter) st->done = 1;
}

extern "C" int cfm_partial_entropic_f64(const float* M, int B0, int B1, double reg, double m,
                                        int max_iter, double stop_thr, double* plan, int* info,
                                        void* ws, void* stream) {
    if (!M || !plan || !ws || B0 <= 0 || B1 <= 0 || !(reg > 0.0) || max_iter < 0) return CFM_EINVAL;
    // POT: m must lie in [0, min(sum a, sum b)] = [0, 1]
    if (!(m >= 0.0) || m > 1.0 + 1e-15) return CFM_EINVAL;
    if (((uintptr_t)ws & 15) != 0 || ((uintptr_t)plan & 15) != 0) return CFM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    UbWs w = ub_carve(ws, B0, B1);
    const size_t n = (size_t)B0 * B1;
    const int nchunk = ub_nchunk(B0, B1);
    const int rows_per_chunk = (B0 + nchunk - 1) / nchunk;
    const double a = 1.0 / B0, b = 1.0 / B1;
    const int nmax = B0 > B1 ? B0 : B1;
    hipLaunchKernelGGL(ub_state_init, dim3(1), dim3(64), 0, s, w.st);
    hipLaunchKernelGGL(ub_gibbs, dim3(2048), dim3(256), 0, s, M, n, -reg, plan, w.st);
    hipLaunchKernelGGL(pt_scale, dim3(2048), dim3(256), 0, s, plan, n, m, w.st);
    hipLaunchKernelGGL(ub_fill, dim3((B0 + 255) / 256), dim3(256), 0, s, w.U[0], B0, 1.0);
    hipLaunchKernelGGL(ub_fill, dim3((B1 + 255) / 256), dim3(256), 0, s, w.V[0], B1, 1.0);
    hipLaunchKernelGGL(ub_fill, dim3((B0 + 255) / 256), dim3(256), 0, s, w.rho, B0, 1.0);
    hipLaunchKernelGGL(ub_fill, dim3((B1 + 255) / 256), dim3(256), 0, s, w.kappa, B1, 1.0);
    // err lives in colacc[0] between checks (initial value 1: the loop always starts)
    hipLaunchKernelGGL(ub_fill, dim3(1), dim3(64), 0, s, w.colacc, 1, 1.0);
    for (int cpt = 0; cpt < max_iter; ++cpt) {
        const int check = (cpt % 10 == 0) ? 1 : 0;
        const int p = cpt & 1, q = p ^ 1;    // (alpha, beta) = (U[p], V[p]) -> (U[q], V[q])
        hipLaunchKernelGGL(ub_row_fused<1>, dim3((B0 + 3) / 4), dim3(256), 0, s, plan, B0, B1, w.st, w.V[p], a, 0.0,
                           w.U[p], w.U[q], w.rho, 0, p);
        hipLaunchKernelGGL(ub_coldot, dim3((B1 + 255) / 256, nchunk), dim3(256), 0, s, plan, B0, B1, w.st,
                           w.U[q], w.part, rows_per_chunk);
        hipLaunchKernelGGL(ub_col_fused<1>, dim3((B1 + UB_COLW - 1) / UB_COLW), dim3(UB_COLW), 0, s, B1, nchunk, w.st, w.part, b, 0.0,
                           w.V[p], w.V[q], w.kappa, check, cpt, max_iter, stop_thr, m, check, (const double*)w.colacc,
                           B0, (const double*)w.U[p], (const double*)w.U[q]);
        if (check) {
            hipLaunchKernelGGL(pt_err, dim3(2048), dim3(256), 0, s, plan, B0, B1, w.st, w.U[p], w.V[p], w.U[q], w.V[q], p);
            hipLaunchKernelGGL(pt_decide, dim3(1), dim3(1), 0, s, w.st, cpt, max_iter, stop_thr, check, w.colacc);
        }
    }
    hipLaunchKernelGGL(pt_fold_final, dim3((B0 + 255) / 256), dim3(256), 0, s, B0, w.st, w.U[0], w.U[1]);
    // POT keeps the NaN matrix when K had zeros (0 / 0 in the q updates poisons every entry within
    // two iterations): reproduce it so the caller's diagnostics (optimal_transport.py:88-92) fire
    hipLaunchKernelGGL(ub_plan, dim3(2048), dim3(256), 0, s, plan, B0, B1, w.st, w.U[0], w.U[1], w.V[0], w.V[1], 1);
    hipLaunchKernelGGL(ub_info, dim3(1), dim3(64), 0, s, w.st, info);
    return cfm_status();
}
