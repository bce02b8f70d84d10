// sinkhorn.hip — K5: log-domain Sinkhorn iteration for gfx950.
//
// Replaces pot.sinkhorn(a, b, M, reg) (torchcfm/optimal_transport.py:51,87) in
// its stable log-domain form, POT loop semantics (u0 = 0; column update then
// row update; marginal check every `check_every` iterations; stopThr on the L2
// violation of the column marginal; numItermax).
//
// One iteration = two streaming passes over M (2*4*B0*B1 algorithmic bytes):
//   sk_col_pass      lane owns 4 adjacent columns, walks a strip of rows with an
//                    online (max,sum) LSE -> no cross-lane traffic, 1 KiB
//                    contiguous per wave load; strips combined by
//   sk_col_finalize  (per-column merge of the strip partials, new v, and — for
//                    free — the marginal error of the PREVIOUS iteration:
//                    colsum_j = b_j * exp(v_old_j - v_new_j));
//   sk_row_stream    persistent grid, one 8-wave workgroup per CU: v staged in LDS (fp64) once per
//                    workgroup, every wave streams its rows in 1024-column units (registers
//                    re-requested for the next unit as each trip consumes them), wave
//                    reduction of the per-lane (max,sum) pairs per row;
//   sk_row_pass      the one-shot / generic form (ragged widths, v too wide for LDS).
// The exponent (u_i + v_j - M_ij/reg) cancels three O(100/reg) numbers to O(1):
// it is formed in fp64 (the chip is HBM-bound here; fp64 VALU is free), only
// exp() itself runs in fp32.  Log-scalings u, v live in fp64 in the workspace.
//
// Convergence is decided on the device (no host sync): every kernel of the
// pre-enqueued sequence reads the state block and exits if `done` is set.
#include "cfm_common.h"
#include <mutex>
#include <stdlib.h>
#include <type_traits>

#define SK_NEG (-1.0e300)
#define SK_NCHUNK_MAX 64

struct SkState {
    int done;        // set once converged
    int iters_done;  // POT's ii+1 at break, or max_iter
    int vfinal;      // which v buffer holds the final v
    int precise;     // 1: fp64 exp/accumulate (near convergence)
    double err2[2];  // sum of squared marginal violations (ping-pong)
    double last_err;
};

struct SkWs {
    SkState* st;
    double* u;
    double* v[2];
    double* pm;  // [nchunk][B1]
    double* ps;  // [nchunk][B1]
};

static inline int sk_nchunk(int B0, int B1) {
    int tiles = (B1 + 255) / 256;
    int n = (1024 + tiles - 1) / tiles;  // aim for ~1024 workgroups
    if (n > SK_NCHUNK_MAX) n = SK_NCHUNK_MAX;
    int maxn = (B0 + 31) / 32;           // at least 32 rows per strip
    if (n > maxn) n = maxn;
    if (n < 1) n = 1;
    return n;
}

static inline size_t sk_ws_bytes(int B0, int B1) {
    size_t nchunk = sk_nchunk(B0, B1);
    return 256 + sizeof(double) * ((size_t)B0 + 2 * (size_t)B1 + 2 * nchunk * (size_t)B1) + 64;
}

static inline SkWs sk_carve(void* ws, int B0, int B1) {
    SkWs w;
    char* p = (char*)ws;
    w.st = (SkState*)p; p += 256;
    w.u = (double*)p; p += sizeof(double) * (size_t)B0;
    w.v[0] = (double*)p; p += sizeof(double) * (size_t)B1;
    w.v[1] = (double*)p; p += sizeof(double) * (size_t)B1;
    size_t nchunk = sk_nchunk(B0, B1);
    w.pm = (double*)p; p += sizeof(double) * nchunk * (size_t)B1;
    w.ps = (double*)p;
    return w;
}

extern "C" size_t cfm_sk_ws_bytes_internal(int B0, int B1) { return sk_ws_bytes(B0, B1); }

__global__ void sk_init(SkState* st, double* u, double* v0, double* v1, int B0, int B1,
                        int max_iter) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B0) u[i] = 0.0;
    if (i < B1) { v0[i] = 0.0; v1[i] = 0.0; }
    if (i == 0) {
        st->done = 0; st->iters_done = max_iter; st->vfinal = (max_iter - 1) & 1; st->precise = 0;
        st->err2[0] = 0.0; st->err2[1] = 0.0; st->last_err = 1.0;
    }
}

__device__ __forceinline__ float4 sk_load4(const float* __restrict__ row, int j, int B1, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(row + j);
    float4 r;
    r.x = (j + 0 < B1) ? row[j + 0] : 0.f;
    r.y = (j + 1 < B1) ? row[j + 1] : 0.f;
    r.z = (j + 2 < B1) ? row[j + 2] : 0.f;
    r.w = (j + 3 < B1) ? row[j + 3] : 0.f;
    return r;
}

#define SK_COL_U 8    // rows per wave and trip of the column pass

// Column pass: partial LSE_i(u_i - M_ij/reg) over a strip of rows.
// precise == 0: exponent formed in fp64, exp() in fp32 (fast, HBM-bound).
// precise != 0: exp() and the running sums in fp64 (engaged near convergence,
// where the fp32 exp noise floor ~1e-7 would hide a 1e-9 marginal violation).
// The state block (done / precise) is read AFTER the first trip's rows have been requested: a
// kernel starts cold, and the state would otherwise be a dependent hop in front of the matrix.
__global__ __launch_bounds__(256) void sk_col_pass(const float* __restrict__ M, int B0, int B1,
                                                   double inv_reg, const SkState* __restrict__ st,
                                                   const double* __restrict__ u,
                                                   double* __restrict__ pm,
                                                   double* __restrict__ ps, int rows_per_chunk,
                                                   int vec) {
    __shared__ double sm[4][256];
    __shared__ double ss[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 256 + lane * 4;
    const int chunk = blockIdx.y;
    const int r_beg = chunk * rows_per_chunk;
    const int r_end = min(B0, r_beg + rows_per_chunk);
    const bool active = j < B1;
    const bool v4 = vec && (j + 3 < B1);

    double m[4] = {SK_NEG, SK_NEG, SK_NEG, SK_NEG};
    double s[4] = {0, 0, 0, 0};     // the fp32 path keeps its sums in float: the round trip is exact
    int precise = 0;

    constexpr int U = SK_COL_U;
    for (int r0 = r_beg + wv * U, first = 1; r0 < r_end || first; r0 += 4 * U) {
        float4 c[U];
        double ui[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int r = r0 + k;
            const bool ok = active && r < r_end;
            c[k] = ok ? sk_load4(M + (size_t)r * B1, j, B1, v4) : make_float4(0.f, 0.f, 0.f, 0.f);
            ui[k] = (r < r_end) ? u[r] : SK_NEG;  // wave-uniform
        }
        if (first) {
            first = 0;
            if (st->done) return;
            precise = st->precise;
            if (r0 >= r_end) break;            // a wave without rows still joins the merge below
        }
        double x[U][4];
        double mx[4] = {m[0], m[1], m[2], m[3]};
#pragma unroll
        for (int k = 0; k < U; ++k) {
            x[k][0] = fma(-(double)c[k].x, inv_reg, ui[k]);
            x[k][1] = fma(-(double)c[k].y, inv_reg, ui[k]);
            x[k][2] = fma(-(double)c[k].z, inv_reg, ui[k]);
            x[k][3] = fma(-(double)c[k].w, inv_reg, ui[k]);
#pragma unroll
            for (int q = 0; q < 4; ++q) mx[q] = fmax(mx[q], x[k][q]);
        }
        if (precise) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double acc = s[q] * exp(m[q] - mx[q]);
#pragma unroll
                for (int k = 0; k < U; ++k) acc += exp(x[k][q] - mx[q]);
                s[q] = acc;
                m[q] = mx[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float acc = (float)s[q] * __expf((float)(m[q] - mx[q]));
#pragma unroll
                for (int k = 0; k < U; ++k) acc += __expf((float)(x[k][q] - mx[q]));
                s[q] = (double)acc;
                m[q] = mx[q];
            }
        }
    }
    // merge the 4 waves of the workgroup (same columns, different rows)
#pragma unroll
    for (int q = 0; q < 4; ++q) { sm[wv][lane * 4 + q] = m[q]; ss[wv][lane * 4 + q] = s[q]; }
    __syncthreads();
    const int tc = threadIdx.x;  // one column per thread
    const int jc = blockIdx.x * 256 + tc;
    if (jc < B1) {
        double mm = fmax(fmax(sm[0][tc], sm[1][tc]), fmax(sm[2][tc], sm[3][tc]));
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += ss[w][tc] * exp(sm[w][tc] - mm);
        pm[(size_t)chunk * B1 + jc] = mm;
        ps[(size_t)chunk * B1 + jc] = tot;
    }
}

// Merge strip partials -> v_new; accumulate the previous iteration's marginal
// error when that iteration was a check iteration.  Workgroup = 64 columns x 4
// groups of strips (lane <-> column: every partial load is a coalesced 512 B
// wave load, all of a thread's loads are in flight together), LDS merge.
#define SK_FIN_MAXPER 16   // strips per thread: SK_NCHUNK_MAX / 4
__global__ __launch_bounds__(256) void sk_col_finalize(int B1, int nchunk, double logb, double b,
                                                       SkState* __restrict__ st,
                                                       const double* __restrict__ pm,
                                                       const double* __restrict__ ps,
                                                       const double* __restrict__ v_old,
                                                       double* __restrict__ v_new, int check,
                                                       int slot) {
    __shared__ double sm[4][64];
    __shared__ double ss[4][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    double m[SK_FIN_MAXPER], sv[SK_FIN_MAXPER];
#pragma unroll
    for (int q = 0; q < SK_FIN_MAXPER; ++q) {     // requested before the state block is looked at
        const int c = part + 4 * q;
        const bool ok = (j < B1) && (c < nchunk);
        m[q] = ok ? pm[(size_t)c * B1 + j] : SK_NEG;
        sv[q] = ok ? ps[(size_t)c * B1 + j] : 0.0;
    }
    const double vo = (check && j < B1) ? v_old[j] : 0.0;
    if (st->done) return;
    double mm = SK_NEG;
#pragma unroll
    for (int q = 0; q < SK_FIN_MAXPER; ++q) mm = fmax(mm, m[q]);
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < SK_FIN_MAXPER; ++q) tot += sv[q] * exp(m[q] - mm);
    sm[part][lane] = mm; ss[part][lane] = tot;
    __syncthreads();
    double e2 = 0.0;
    if (part == 0) {
        if (j < B1) {
            const double m4 = fmax(fmax(sm[0][lane], sm[1][lane]), fmax(sm[2][lane], sm[3][lane]));
            double t4 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) t4 += ss[w][lane] * exp(sm[w][lane] - m4);
            const double vn = logb - (m4 + log(t4));
            v_new[j] = vn;
            if (check) {
                const double e = b * exp(vo - vn) - b;
                e2 = e * e;
            }
        }
        if (check) {
            e2 = wave_sum_d(e2);
            if (lane == 0) atomicAdd(&st->err2[slot], e2);
        }
    }
}

template <bool PRECISE>
__device__ __forceinline__ void sk_row_body_generic(const float* __restrict__ M, int B0, int B1,
                                            double inv_reg, double loga,
                                            const double* __restrict__ vv, double* __restrict__ u,
                                            int rows_per_wg, int vec) {
    typedef typename std::conditional<PRECISE, double, float>::type acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r_beg = blockIdx.x * rows_per_wg;
    const int r_end = min(B0, r_beg + rows_per_wg);
    for (int r = r_beg + wv; r < r_end; r += 4) {
        const float* row = M + (size_t)r * B1;
        double m = SK_NEG;
        acc_t s = 0;
        constexpr int U = 4;
        for (int j0 = lane * 4; j0 < B1; j0 += 256 * U) {
            float4 c[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int j = j0 + 256 * k;
                c[k] = (j < B1) ? sk_load4(row, j, B1, vec && (j + 3 < B1)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            double x[U][4];
            double mx = m;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int j = j0 + 256 * k;
                const float cc[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x[k][q] = (j + q < B1) ? fma(-(double)cc[q], inv_reg, vv[j + q]) : SK_NEG;
                    mx = fmax(mx, x[k][q]);
                }
            }
            if (PRECISE) {
                double acc = (double)s * exp(m - mx);
#pragma unroll
                for (int k = 0; k < U; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc += exp(x[k][q] - mx);
                s = (acc_t)acc;
            } else {
                float acc = (float)s * __expf((float)(m - mx));
#pragma unroll
                for (int k = 0; k < U; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc += __expf((float)(x[k][q] - mx));
                s = (acc_t)acc;
            }
            m = mx;
        }
        const double mm = wave_max_d(m);
        const double tot = wave_sum_d((double)s * exp(m - mm));
        if (lane == 0) u[r] = loga - (mm + log(tot));
    }
}


// ---- fast row pass: B1 % 1024 == 0, 16-byte aligned rows -------------------
// One trip of the row LSE: U float4 per lane (4 * U columns), online (max, sum).
template <bool PRECISE, int U>
__device__ __forceinline__ void sk_row_trip(const float4 (&c)[U], int j0, double inv_reg,
                                            const double* __restrict__ vv, double& m,
                                            typename std::conditional<PRECISE, double, float>::type& s) {
    typedef typename std::conditional<PRECISE, double, float>::type acc_t;
    double x[U][4];
    double mx = m;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int j = j0 + 256 * k;
        const double4 v4 = *reinterpret_cast<const double4*>(vv + j);
        x[k][0] = fma(-(double)c[k].x, inv_reg, v4.x);
        x[k][1] = fma(-(double)c[k].y, inv_reg, v4.y);
        x[k][2] = fma(-(double)c[k].z, inv_reg, v4.z);
        x[k][3] = fma(-(double)c[k].w, inv_reg, v4.w);
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmax(mx, x[k][q]);
    }
    if (PRECISE) {
        double acc = (double)s * exp(m - mx);
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += exp(x[k][q] - mx);
        s = (acc_t)acc;
    } else {
        float acc = (float)s * __expf((float)(m - mx));
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += __expf((float)(x[k][q] - mx));
        s = (acc_t)acc;
    }
    m = mx;
}

#define SK_ROW_PRE 16   // float4 per lane in flight per 4096-column segment
#define SK_ROW_WAVES 4   // waves per workgroup of the fast row pass
#define SK_RPW 4         // rows per workgroup of the fast row pass (a multiple of SK_ROW_WAVES)
#define SK_ROW_OCC 3     // waves per SIMD the fast row pass is compiled for
#define SK_ROW_THREADS (64 * SK_ROW_WAVES)

// One wave per row; the row's first segment was loaded BEFORE v was staged into LDS (the two
// latencies overlap), later segments of a longer row are loaded 16 float4 at a time.
template <bool PRECISE>
__device__ __forceinline__ void sk_row_fast(const float* __restrict__ row, int B1, double inv_reg,
                                            double loga, const double* __restrict__ vv,
                                            double* __restrict__ u_out, float4 (&pre)[SK_ROW_PRE]) {
    typedef typename std::conditional<PRECISE, double, float>::type acc_t;
    const int lane = threadIdx.x & 63;
    double m = SK_NEG;
    acc_t s = 0;
    constexpr int UT = PRECISE ? 1 : 2;      // float4 per trip (the fp64-exp path is register hungry)
    for (int seg = 0; seg < B1; seg += 4096) {
        if (seg > 0) {
#pragma unroll
            for (int k = 0; k < SK_ROW_PRE; ++k)
                pre[k] = *reinterpret_cast<const float4*>(row + seg + lane * 4 + 256 * k);
        }
        const int ntrip = min(4096, B1 - seg) / (256 * UT);     // B1 % 1024 == 0
#pragma unroll
        for (int t = 0; t < SK_ROW_PRE / UT; ++t) {
            if (t < ntrip) {
                float4 c4[UT];
#pragma unroll
                for (int k = 0; k < UT; ++k) c4[k] = pre[UT * t + k];
                sk_row_trip<PRECISE, UT>(c4, seg + lane * 4 + 256 * UT * t, inv_reg, vv, m, s);
            }
            __builtin_amdgcn_sched_barrier(0);   // one trip at a time: bounds the live registers
        }
    }
    const double mm = wave_max_d(m);
    const double tot = wave_sum_d((double)s * exp(m - mm));
    if (lane == 0) *u_out = loga - (mm + log(tot));
}

// Row pass: u_i = log a - LSE_j(v_j - M_ij/reg); also the convergence decision
// for the previous iteration (every workgroup derives it from the same data).
template <bool FAST>
__global__ __launch_bounds__(FAST ? SK_ROW_THREADS : 256, FAST ? SK_ROW_OCC : 4) void sk_row_pass(const float* __restrict__ M, int B0, int B1,
                                                      double inv_reg, double loga,
                                                      SkState* __restrict__ st,
                                                      const double* __restrict__ v,
                                                      double* __restrict__ u, int rows_per_wg,
                                                      int check, int slot, double stop_thr, int ii,
                                                      int vec, int v_in_lds, double precise_below) {
    if (st->done) return;
    if (check) {
        const double err = sqrt(st->err2[slot]);
        if (err < stop_thr) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                st->last_err = err;
                st->iters_done = ii;          // iterations 0..ii-1 ran; POT broke at ii-1
                st->vfinal = (ii - 1) & 1;
                __threadfence();
                st->done = 1;
            }
            return;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->last_err = err;
            // fp32 exp leaves ~1e-7 relative noise per column sum -> ~1e-7/sqrt(B1) in the L2
            // violation; go precise well above that floor when the caller asks for less.
            if (!st->precise && err < precise_below && stop_thr < precise_below) st->precise = 1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->err2[slot ^ 1] = 0.0;  // next accumulation slot
    const int precise = st->precise;

    extern __shared__ __attribute__((aligned(16))) double vs[];
    if (FAST) {
        // SK_RPW rows per workgroup, SK_RPW / 4 consecutive rows per wave, v staged in LDS once per
        // workgroup.  The first row is requested BEFORE v is staged (the two latencies overlap); with
        // more than one row per wave the next row is requested before the current one is reduced.
        constexpr int RPWV = SK_RPW / SK_ROW_WAVES;
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar row base
        const int r0 = blockIdx.x * SK_RPW + wv * RPWV;
        float4 preA[SK_ROW_PRE];
        {
            const float* row = M + (size_t)(r0 < B0 ? r0 : 0) * B1;
#pragma unroll
            for (int k = 0; k < SK_ROW_PRE; ++k) {
                const int j = lane * 4 + 256 * k;
                preA[k] = (j < B1) ? *reinterpret_cast<const float4*>(row + j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int j = threadIdx.x * 2; j < B1; j += 2 * SK_ROW_THREADS)
            *reinterpret_cast<double2*>(vs + j) = *reinterpret_cast<const double2*>(v + j);
        __syncthreads();
        if (RPWV == 1) {
            if (r0 < B0) {
                const float* row = M + (size_t)r0 * B1;
                if (precise) sk_row_fast<true>(row, B1, inv_reg, loga, vs, u + r0, preA);
                else         sk_row_fast<false>(row, B1, inv_reg, loga, vs, u + r0, preA);
            }
        } else {
            float4 preB[SK_ROW_PRE];
#pragma unroll
            for (int rr = 0; rr < RPWV; rr += 2) {
                const int ra = r0 + rr, rb = r0 + rr + 1, rc = r0 + rr + 2;
                if (rr + 1 < RPWV && rb < B0) {
                    const float* row = M + (size_t)rb * B1;
#pragma unroll
                    for (int k = 0; k < SK_ROW_PRE; ++k) {
                        const int j = lane * 4 + 256 * k;
                        preB[k] = (j < B1) ? *reinterpret_cast<const float4*>(row + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (ra < B0) {
                    const float* row = M + (size_t)ra * B1;
                    if (precise) sk_row_fast<true>(row, B1, inv_reg, loga, vs, u + ra, preA);
                    else         sk_row_fast<false>(row, B1, inv_reg, loga, vs, u + ra, preA);
                }
                if (rr + 2 < RPWV && rc < B0) {
                    const float* row = M + (size_t)rc * B1;
#pragma unroll
                    for (int k = 0; k < SK_ROW_PRE; ++k) {
                        const int j = lane * 4 + 256 * k;
                        preA[k] = (j < B1) ? *reinterpret_cast<const float4*>(row + j) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (rr + 1 < RPWV && rb < B0) {
                    const float* row = M + (size_t)rb * B1;
                    if (precise) sk_row_fast<true>(row, B1, inv_reg, loga, vs, u + rb, preB);
                    else         sk_row_fast<false>(row, B1, inv_reg, loga, vs, u + rb, preB);
                }
            }
        }
    } else {
        if (v_in_lds) {
            for (int j = threadIdx.x; j < B1; j += 256) vs[j] = v[j];
            __syncthreads();
        }
        const double* vv = v_in_lds ? vs : v;
        if (precise) sk_row_body_generic<true>(M, B0, B1, inv_reg, loga, vv, u, rows_per_wg, vec);
        else         sk_row_body_generic<false>(M, B0, B1, inv_reg, loga, vv, u, rows_per_wg, vec);
    }
}

// ---- streaming row pass ------------------------------------------------------
// A persistent grid (about two workgroups per CU): v is staged into LDS once per workgroup, every
// wave walks its rows in 4096-column units and re-requests each pair of float4 registers for the
// NEXT unit as soon as the current unit's trip has consumed them, so the loads of one unit are in
// flight while the previous one is reduced.  Measured on the 64 MiB matrix (scratch/probe): a
// one-shot "request the row, stage v, reduce" workgroup shape takes 18 us where the bare reads take
// 10.7 us — all of the exp/LDS work lands behind the last load — a streaming shape 11.7 us.
#define SK_STREAM_WAVES 8
#define SK_STREAM_UNIT 4         // float4 per lane and unit at most (256 columns each); 8, 16: 2 % and 9 % slower
#define SK_STREAM_OCC 2          // waves per SIMD the streaming row pass is compiled for
#define SK_STREAM_THREADS (64 * SK_STREAM_WAVES)

// One 256*NF4-column unit: reduce the registers trip by trip; with NEXT, each trip's registers are
// re-requested for the next unit right after they were consumed.  No branch in here: a conditional
// load makes the compiler wait for ALL outstanding loads (vmcnt(0)) at every join, which serialises
// the stream.
template <bool PRECISE, int NF4, bool NEXT>
__device__ __forceinline__ void sk_stream_unit(float4 (&c)[SK_ROW_PRE], const float* __restrict__ nxt, int col0,
                                               double inv_reg, const double* __restrict__ vs, double& m,
                                               typename std::conditional<PRECISE, double, float>::type& s) {
    constexpr int UT = PRECISE ? 1 : 2;
#pragma unroll
    for (int t = 0; t < NF4 / UT; ++t) {
        float4 c4[UT];
#pragma unroll
        for (int k = 0; k < UT; ++k) c4[k] = c[UT * t + k];
        sk_row_trip<PRECISE, UT>(c4, col0 + 256 * UT * t, inv_reg, vs, m, s);
        if (NEXT) {
#pragma unroll
            for (int k = 0; k < UT; ++k) c[UT * t + k] = *reinterpret_cast<const float4*>(nxt + 256 * (UT * t + k));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool PRECISE, int NF4>
__device__ __forceinline__ void sk_stream_rows(const float* __restrict__ M, int B0, int B1, double inv_reg,
                                               double loga, const double* __restrict__ vs,
                                               double* __restrict__ u, float4 (&c)[SK_ROW_PRE],
                                               int r, int r_stride) {
    typedef typename std::conditional<PRECISE, double, float>::type acc_t;
    const int lane = threadIdx.x & 63;
    constexpr int UW = 256 * NF4;      // columns per unit; B1 is a multiple of it
    int seg = 0;
    double m = SK_NEG;
    acc_t s = 0;
    if (r >= B0) return;
    for (;;) {
        int rn = r, segn = seg + UW;
        if (segn >= B1) { segn = 0; rn = r + r_stride; }
        const bool has_next = rn < B0;
        if (has_next)
            sk_stream_unit<PRECISE, NF4, true>(c, M + (size_t)rn * B1 + segn + lane * 4, seg + lane * 4, inv_reg, vs, m, s);
        else
            sk_stream_unit<PRECISE, NF4, false>(c, M, seg + lane * 4, inv_reg, vs, m, s);
        if (segn == 0) {                       // the row is complete
            const double mm = wave_max_d(m);
            const double tot = wave_sum_d((double)s * exp(m - mm));
            if (lane == 0) u[r] = loga - (mm + log(tot));
            m = SK_NEG; s = 0;
        }
        if (!has_next) break;
        r = rn; seg = segn;
    }
}

template <int NF4>
__global__ __launch_bounds__(SK_STREAM_THREADS, SK_STREAM_OCC) void sk_row_stream(const float* __restrict__ M, int B0, int B1,
                                                      double inv_reg, double loga,
                                                      SkState* __restrict__ st,
                                                      const double* __restrict__ v,
                                                      double* __restrict__ u,
                                                      int check, int slot, double stop_thr, int ii,
                                                      double precise_below) {
    extern __shared__ __attribute__((aligned(16))) double vs[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // consecutive rows go to different workgroups
    const int r0 = wv * gridDim.x + blockIdx.x;
    const int r_stride = gridDim.x * SK_STREAM_WAVES;
    // the first unit is requested before anything is known about the state: a kernel starts cold,
    // and the state block would otherwise be a dependent hop in front of the matrix
    float4 c[SK_ROW_PRE];
    {
        const float* row = M + (size_t)(r0 < B0 ? r0 : 0) * B1 + lane * 4;
#pragma unroll
        for (int k = 0; k < NF4; ++k) c[k] = *reinterpret_cast<const float4*>(row + 256 * k);
    }
    if (st->done) return;
    if (check) {
        const double err = sqrt(st->err2[slot]);
        if (err < stop_thr) {
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                st->last_err = err;
                st->iters_done = ii;
                st->vfinal = (ii - 1) & 1;
                __threadfence();
                st->done = 1;
            }
            return;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->last_err = err;
            if (!st->precise && err < precise_below && stop_thr < precise_below) st->precise = 1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) st->err2[slot ^ 1] = 0.0;
    const int precise = st->precise;
    for (int j = threadIdx.x * 2; j < B1; j += 2 * SK_STREAM_THREADS)
        *reinterpret_cast<double2*>(vs + j) = *reinterpret_cast<const double2*>(v + j);
    __syncthreads();
    if (precise) sk_stream_rows<true, NF4>(M, B0, B1, inv_reg, loga, vs, u, c, r0, r_stride);
    else         sk_stream_rows<false, NF4>(M, B0, B1, inv_reg, loga, vs, u, c, r0, r_stride);
}

template <int NF4>
static void sk_launch_stream(int grid, size_t lds, hipStream_t s, const float* M, int B0, int B1, double inv_reg,
                             double loga, SkState* st, const double* v, double* u, int check, int slot,
                             double stop_thr, int ii, double precise_below) {
    hipLaunchKernelGGL(sk_row_stream<NF4>, dim3(grid), dim3(SK_STREAM_THREADS), lds, s, M, B0, B1, inv_reg, loga,
                       st, v, u, check, slot, stop_thr, ii, precise_below);
}

__global__ void sk_finish(SkState* st, const double* u, const double* v0, const double* v1,
                          int B0, int B1, double reg, float* f, float* g, int* iters_done,
                          float* last_err, int pending_check, int slot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double* v = st->vfinal ? v1 : v0;
    if (i < B0 && f) f[i] = (float)(reg * u[i]);
    if (i < B1 && g) g[i] = (float)(reg * v[i]);
    if (i == 0) {
        double le = st->last_err;
        if (!st->done && pending_check) { le = sqrt(st->err2[slot]); st->last_err = le; }
        if (iters_done) *iters_done = st->iters_done;
        if (last_err) *last_err = (float)le;
    }
}

extern "C" int cfm_sinkhorn_log_f32(const float* M, int B0, int B1, double reg, int max_iter,
                                    double stop_thr, int check_every, float* f, float* g,
                                    int* iters_done, float* last_err, void* ws, void* stream) {
    if (!M || !ws || B0 <= 0 || B1 <= 0 || !(reg > 0.0) || max_iter < 0 || check_every <= 0)
        return CFM_EINVAL;
    if (((uintptr_t)ws & 15) != 0) return CFM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    SkWs w = sk_carve(ws, B0, B1);
    const int nchunk = sk_nchunk(B0, B1);
    const int rows_per_chunk = (B0 + nchunk - 1) / nchunk;
    const int col_tiles = (B1 + 255) / 256;
    const int vec = ((B1 & 3) == 0) && (((uintptr_t)M & 15) == 0);
    const double inv_reg = 1.0 / reg;
    const double a = 1.0 / B0, b = 1.0 / B1;
    const double precise_below = 1e-4 / sqrt((double)B1);
    const double loga = log(a), logb = log(b);
    // fast row pass: whole 1024-column trips, 16-byte aligned rows, v fits in LDS
    const bool row_fast = vec && (B1 % 1024 == 0) && ((size_t)B1 * 8 <= 128 * 1024);
    const int rows_per_wg = row_fast ? SK_RPW : 8;
    const int row_wgs = (B0 + rows_per_wg - 1) / rows_per_wg;
    int v_in_lds = ((size_t)B1 * 8 <= 128 * 1024) ? 1 : 0;
    if (v_in_lds && (size_t)B1 * 8 > 48 * 1024) {
        static int raised_d[CFM_MAX_DEVICES];   // dynamic LDS above the 64 KiB default needs the attribute (per device)
        static std::once_flag once_d[CFM_MAX_DEVICES];
        int& raised = raised_d[cfm_device_index()];
        std::call_once(once_d[cfm_device_index()], [&raised] {
            hipError_t e = hipFuncSetAttribute((const void*)sk_row_pass<true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipError_t e2 = hipFuncSetAttribute((const void*)sk_row_pass<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipGetLastError();
            raised = (e == hipSuccess && e2 == hipSuccess) ? 1 : -1;
        });
        if (raised < 0) v_in_lds = 0;
    }
    const size_t lds = v_in_lds ? (size_t)B1 * 8 : 0;
    // streaming row pass: a persistent grid of SK_STREAM_WAVES-wave workgroups
    int stream_grid = 0, stream_nf4 = 0;
    if (row_fast && v_in_lds) {
        static int per_cu_d[CFM_MAX_DEVICES], cus_d[CFM_MAX_DEVICES];
        static std::once_flag once_stream_d[CFM_MAX_DEVICES];
        const int dvi = cfm_device_index();
        int& per_cu = per_cu_d[dvi]; int& cus = cus_d[dvi];
        std::call_once(once_stream_d[dvi], [&per_cu, &cus] {
            const char* e = getenv("CFM_SK_STREAM");       // workgroups per CU; 0 = one-shot row pass
            int pc = e ? atoi(e) : 1;
            int dev = 0, c = 0;
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
            const void* fns[4] = {(const void*)sk_row_stream<4>, (const void*)sk_row_stream<8>,
                                  (const void*)sk_row_stream<12>, (const void*)sk_row_stream<16>};
            for (int q = 0; q < 4; ++q)
                if (hipFuncSetAttribute(fns[q], hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
                    pc = 0;
            (void)hipGetLastError();
            cus = c; per_cu = pc < 0 ? 0 : pc;
        });
        // units of 256 * nf4 columns: the largest of 4, 8, 12, 16 float4 per lane that divides the row
        for (int q = 4; q <= SK_STREAM_UNIT; q += 4)
            if ((B1 / 256) % q == 0) stream_nf4 = q;
        const int want = (B0 + SK_STREAM_WAVES - 1) / SK_STREAM_WAVES;
        stream_grid = (per_cu > 0 && stream_nf4 > 0) ? (want < cus * per_cu ? want : cus * per_cu) : 0;
    }

    int n = B0 > B1 ? B0 : B1;
    hipLaunchKernelGGL(sk_init, dim3((n + 255) / 256), dim3(256), 0, s, w.st, w.u, w.v[0], w.v[1],
                       B0, B1, max_iter);
    // iteration ii: v^(ii) -> v[ii&1];  error of iteration ii-1 is measured by
    // the column pass of iteration ii and decided at the start of its row pass.
    // Short runs are enqueued in one go (no host sync); for very long caps
    // (wasserstein() passes numItermax=1e7) the host polls `done` every 512
    // iterations so the queue stays bounded.
    const bool poll = max_iter > 4096;
    int host_done = 0;
    for (int ii = 0; ii <= max_iter; ++ii) {
        const int check = (ii >= 1) && (((ii - 1) % check_every) == 0);
        const int slot = ii & 1;
        const bool trailing = (ii == max_iter);
        if (trailing && !check) break;  // nothing left to measure
        hipLaunchKernelGGL(sk_col_pass, dim3(col_tiles, nchunk), dim3(256), 0, s, M, B0, B1, inv_reg,
                           w.st, w.u, w.pm, w.ps, rows_per_chunk, vec);
        hipLaunchKernelGGL(sk_col_finalize, dim3((B1 + 63) / 64), dim3(256), 0, s, B1, nchunk, logb, b,
                           w.st, w.pm, w.ps, w.v[(ii + 1) & 1], w.v[ii & 1], check, slot);
        if (trailing) break;
        if (row_fast && v_in_lds && stream_grid > 0)
            switch (stream_nf4) {
            case 4:  sk_launch_stream<4>(stream_grid, lds, s, M, B0, B1, inv_reg, loga, w.st, w.v[ii & 1], w.u, check, slot, stop_thr, ii, precise_below); break;
            case 8:  sk_launch_stream<8>(stream_grid, lds, s, M, B0, B1, inv_reg, loga, w.st, w.v[ii & 1], w.u, check, slot, stop_thr, ii, precise_below); break;
            case 12: sk_launch_stream<12>(stream_grid, lds, s, M, B0, B1, inv_reg, loga, w.st, w.v[ii & 1], w.u, check, slot, stop_thr, ii, precise_below); break;
            default: sk_launch_stream<16>(stream_grid, lds, s, M, B0, B1, inv_reg, loga, w.st, w.v[ii & 1], w.u, check, slot, stop_thr, ii, precise_below); break;
            }
        else if (row_fast && v_in_lds)
            hipLaunchKernelGGL(sk_row_pass<true>, dim3(row_wgs), dim3(SK_ROW_THREADS), lds, s, M, B0, B1, inv_reg, loga,
                               w.st, w.v[ii & 1], w.u, rows_per_wg, check, slot, stop_thr, ii,
                               vec, v_in_lds, precise_below);
        else
            hipLaunchKernelGGL(sk_row_pass<false>, dim3(row_wgs), dim3(256), lds, s, M, B0, B1, inv_reg, loga,
                               w.st, w.v[ii & 1], w.u, rows_per_wg, check, slot, stop_thr, ii,
                               vec, v_in_lds, precise_below);
        if (poll && (ii & 511) == 511) {
            int rc = cfm_hip(hipMemcpyAsync(&host_done, &w.st->done, sizeof(int), hipMemcpyDeviceToHost, s));
            if (rc) return rc;
            rc = cfm_hip(hipStreamSynchronize(s));
            if (rc) return rc;
            if (host_done) break;
        }
    }
    const int pending = (max_iter >= 1) && (((max_iter - 1) % check_every) == 0);
    hipLaunchKernelGGL(sk_finish, dim3((n + 255) / 256), dim3(256), 0, s, w.st, w.u, w.v[0], w.v[1],
                       B0, B1, reg, f, g, iters_done, last_err, pending, max_iter & 1);
    return cfm_status();
}

__global__ void sk_copy_potentials(const SkState* st, const double* u, const double* v0,
                                   const double* v1, int B0, int B1, double* uo, double* vo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double* v = st->vfinal ? v1 : v0;
    if (i < B0 && uo) uo[i] = u[i];
    if (i < B1 && vo) vo[i] = v[i];
}

extern "C" int cfm_sinkhorn_potentials_f64(const void* ws, int B0, int B1, double* u, double* v,
                                           void* stream) {
    if (!ws || B0 <= 0 || B1 <= 0) return CFM_EINVAL;
    SkWs w = sk_carve((void*)ws, B0, B1);
    int n = B0 > B1 ? B0 : B1;
    hipLaunchKernelGGL(sk_copy_potentials, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       w.st, w.u, w.v[0], w.v[1], B0, B1, u, v);
    return cfm_status();
}

// Dense fp64 plan and <pi, M>.
__global__ __launch_bounds__(256) void sk_plan_f64(const float* __restrict__ M, int B0, int B1,
                                                   double inv_reg, const SkState* st,
                                                   const double* __restrict__ u,
                                                   const double* __restrict__ v0,
                                                   const double* __restrict__ v1,
                                                   double* __restrict__ pi,
                                                   double* __restrict__ cost_out) {
    const double* v = st->vfinal ? v1 : v0;
    const size_t n = (size_t)B0 * B1;
    double acc = 0.0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
        const int i = (int)(k / B1), j = (int)(k - (size_t)i * B1);
        const double c = (double)M[k];
        const double p = exp(u[i] + v[j] - c * inv_reg);
        if (pi) pi[k] = p;
        acc += p * c;
    }
    if (cost_out) {
        acc = wave_sum_d(acc);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(cost_out, red[0] + red[1] + red[2] + red[3]);
    }
}

extern "C" int cfm_sinkhorn_plan_f64(const float* M, int B0, int B1, double reg, const void* ws,
                                     double* pi, void* stream) {
    if (!M || !ws || !pi || B0 <= 0 || B1 <= 0 || !(reg > 0.0)) return CFM_EINVAL;
    SkWs w = sk_carve((void*)ws, B0, B1);
    hipLaunchKernelGGL(sk_plan_f64, dim3(2048), dim3(256), 0, (hipStream_t)stream, M, B0, B1,
                       1.0 / reg, w.st, w.u, w.v[0], w.v[1], pi, (double*)nullptr);
    return cfm_status();
}

extern "C" int cfm_sinkhorn_cost_f64(const float* M, int B0, int B1, double reg, const void* ws,
                                     double* out, void* stream) {
    if (!M || !ws || !out || B0 <= 0 || B1 <= 0 || !(reg > 0.0)) return CFM_EINVAL;
    SkWs w = sk_carve((void*)ws, B0, B1);
    int rc = cfm_hip(hipMemsetAsync(out, 0, sizeof(double), (hipStream_t)stream));
    if (rc) return rc;
    hipLaunchKernelGGL(sk_plan_f64, dim3(2048), dim3(256), 0, (hipStream_t)stream, M, B0, B1,
                       1.0 / reg, w.st, w.u, w.v[0], w.v[1], (double*)nullptr, out);
    return cfm_status();
}
