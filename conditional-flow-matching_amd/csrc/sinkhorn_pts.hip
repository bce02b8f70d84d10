// sinkhorn_pts.hip — K5, variant B: log-domain Sinkhorn for low-dimensional point clouds (d <= 8) with
// the cost recomputed on the fly.
//
// Same loop as sinkhorn.hip (pot.sinkhorn through torchcfm/optimal_transport.py:51,84,87 — POT order:
// column update, row update, marginal check every `check_every` iterations, stopThr, numItermax), same
// state block and workspace layout (so the potentials feed cfm_plan_sample_dense / cfm_sinkhorn_plan_f64 /
// cfm_sinkhorn_cost_f64 unchanged), but no pass ever reads the B0 x B1 matrix: at d = 2 an entry is two
// subtractions and two FMAs away from the 32 KiB of coordinates, against 4 bytes of a 64 MiB matrix that
// two passes per iteration would stream (134 MB of algorithmic traffic per iteration at B = 4096).
// The entry is formed EXACTLY as cost_small_d forms it (s = fmaf(x0_k - x1_k, x0_k - x1_k, s) in k order),
// so the potentials belong to the same fp32 matrix the sampling kernels read.
//
// One kernel per half-iteration (both directions are the same kernel with the clouds swapped), no
// separate merge launch: a 16-wave workgroup owns 16 points of the updated side; lane = (sub, own):
// 4 sub-groups x 16 own points; wave w and sub-group s take every 64th point of the other side
// (staged in LDS with its fp64 potential, read as 4-address broadcasts), an online fp64 (max, sum) LSE
// with fp32 exp per lane (fp64 exp near convergence, as in sinkhorn.hip), two lane exchanges across the
// sub-groups, an LDS merge across the waves, and the 16 new potentials written by wave 0 — together
// with their share of the marginal violation on check iterations.
#include "cfm_common.h"
#include <type_traits>
#include <mutex>

#define SK_NEG (-1.0e300)

// (identical to sinkhorn.hip: the two files share the workspace)
struct SkState {
    int done; int iters_done; int vfinal; int precise;
    double err2[2];
    double last_err;
};
extern "C" size_t cfm_sk_ws_bytes_internal(int B0, int B1);

#define PTS_T 1024
#define PTS_NW (PTS_T / 64)
#define PTS_OWN 16           // own points per workgroup
#define PTS_U 8              // other points per trip and lane
#define PTS_STRIDE (PTS_NW * 4)   // lane-groups (wave, sub) interleave the other side's points
// other points per trip and lane, by dimension: a trip keeps U x D coordinates + 2 U doubles in registers, and the
// 1024-thread workgroup leaves 128 per lane — at U = 8, d = 6 / 7 / 8 spilled 48 / 100 / 248 bytes per lane (round 3);
// trips of 4 for d >= 6 fit (tools/isa_report.py: 0 bytes of scratch for every d).  The staged chunk stays padded to
// whole trips of PTS_U = 8, which is a whole number of the shorter trips too.
template <int D> struct PtsTrip { static constexpr int U = (D <= 5) ? PTS_U : PTS_U / 2; };

template <int D, bool PRECISE>
__device__ __forceinline__ void pts_accumulate(const float (&own)[D], const float* __restrict__ pts_lds,
                                               const double* __restrict__ pot_lds, int first, int n_stage,
                                               double inv_reg, double& m, double& s_acc) {
    // this lane's points of the staged chunk: first, first + PTS_STRIDE, ...   (PTS_U of them per trip).  The chunk is
    // padded to a whole number of trips with potential = -inf entries, so the trip is branch free: its
    // 2 x PTS_U LDS reads go out back to back (a per-point bounds test made every read wait for the previous one)
    constexpr int U = PtsTrip<D>::U;
    for (int t0 = first; t0 < n_stage; t0 += PTS_STRIDE * U) {
        float pc[U][D]; double pp[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int t = t0 + PTS_STRIDE * k;
            pp[k] = pot_lds[t];
#pragma unroll
            for (int q = 0; q < D; ++q) pc[k][q] = pts_lds[t * D + q];
        }
        double x[U];
        double mx = m;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            float c = 0.f;
#pragma unroll
            for (int q = 0; q < D; ++q) { const float df = own[q] - pc[k][q]; c = fmaf(df, df, c); }
            x[k] = fma(-(double)c, inv_reg, pp[k]);
            mx = fmax(mx, x[k]);
        }
        if (PRECISE) {
            double acc = s_acc * exp(m - mx);
#pragma unroll
            for (int k = 0; k < U; ++k) acc += exp(x[k] - mx);
            s_acc = acc;
        } else {
            float acc = (float)s_acc * __expf((float)(m - mx));
#pragma unroll
            for (int k = 0; k < U; ++k) acc += __expf((float)(x[k] - mx));
            s_acc = (double)acc;
        }
        m = mx;
    }
}

// merge two (max, sum) pairs; the rescaling exp runs in fp32 unless the solve is in its precise phase
// (the sums were accumulated with fp32 exps anyway)
__device__ __forceinline__ void pts_merge(double& m, double& s, double m2, double s2, bool precise) {
    const double mm = fmax(m, m2);
    if (precise) s = s * exp(m - mm) + s2 * exp(m2 - mm);
    else s = (double)((float)s * __expf((float)(m - mm)) + (float)s2 * __expf((float)(m2 - mm)));
    m = mm;
}

// new_pot[o] = logw - LSE_t(pot_other[t] - |own_o - other_t|^2 / reg) for the 16 own points of the workgroup.
// For the row update own = x0, for the column update own = x1: the difference has the other sign, its
// square is bit-identical.
#define PTS_PRE 4            // points per thread of the first staged chunk that are requested in the prologue (d <= 5; 2 beyond)
template <int D>
__global__ __launch_bounds__(PTS_T) void sk_pts_pass(const float* __restrict__ own_pts, const float* __restrict__ other_pts,
                                                     int n_own, int n_other, double inv_reg, double logw, double wgt,
                                                     SkState* __restrict__ st, const double* __restrict__ pot_other,
                                                     const double* __restrict__ pot_old, double* __restrict__ pot_new,
                                                     int row_update, int check, int slot, double stop_thr, int ii,
                                                     double precise_below, int stage_cap) {
    extern __shared__ __attribute__((aligned(16))) char pts_lds_raw[];
    double* pot_lds = reinterpret_cast<double*>(pts_lds_raw);                       // [stage_cap]
    float* pts_lds = reinterpret_cast<float*>(pts_lds_raw + (size_t)stage_cap * 8); // [stage_cap * D]
    __shared__ double sm[PTS_NW][PTS_OWN];
    __shared__ double ss[PTS_NW][PTS_OWN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 4, ol = lane & 15;
    const int o = blockIdx.x * PTS_OWN + ol;
    // A launch starts cold: everything the first chunk needs — the own point, this thread's share of the
    // other cloud and of its potentials — is requested before the state block is looked at.
    float own[D];
#pragma unroll
    for (int q = 0; q < D; ++q) own[q] = (o < n_own) ? own_pts[(size_t)o * D + q] : 0.f;
    const int n_stage0 = n_other < stage_cap ? n_other : stage_cap;
    constexpr int PRE = (D <= 5) ? PTS_PRE : PTS_PRE / 2;
    double pre_pot[PRE]; float pre_pts[PRE][D];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const int t = threadIdx.x + PTS_T * k;
        const bool ok = t < n_stage0;
        pre_pot[k] = ok ? pot_other[t] : SK_NEG;
#pragma unroll
        for (int q = 0; q < D; ++q) pre_pts[k][q] = ok ? other_pts[(size_t)t * D + q] : 0.f;
    }
    const double pold = (!row_update && check && o < n_own) ? pot_old[o] : 0.0;
    if (st->done) return;
    if (row_update) {
        // the convergence decision for the previous iteration: every workgroup derives it from the same data
        if (check) {
            const double err = sqrt(st->err2[slot]);
            if (err < stop_thr) {
                if (blockIdx.x == 0 && threadIdx.x == 0) {
                    st->last_err = err; st->iters_done = ii; st->vfinal = (ii - 1) & 1;
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
    }
    const int precise = st->precise;
    double m = SK_NEG, s_acc = 0.0;
    for (int c0 = 0; c0 < n_other; c0 += stage_cap) {
        const int n_stage = (n_other - c0 < stage_cap) ? n_other - c0 : stage_cap;
        const int n_pad = (n_stage + PTS_STRIDE * PTS_U - 1) / (PTS_STRIDE * PTS_U) * (PTS_STRIDE * PTS_U);     // <= stage_cap (a multiple of it)
        __syncthreads();
        if (c0 == 0) {
#pragma unroll
            for (int k = 0; k < PRE; ++k) {
                const int t = threadIdx.x + PTS_T * k;
                if (t < n_pad) {
                    pot_lds[t] = pre_pot[k];
#pragma unroll
                    for (int q = 0; q < D; ++q) pts_lds[t * D + q] = pre_pts[k][q];
                }
            }
            for (int t = threadIdx.x + PTS_T * PRE; t < n_pad; t += PTS_T) {
                pot_lds[t] = (t < n_stage) ? pot_other[t] : SK_NEG;
#pragma unroll
                for (int q = 0; q < D; ++q) pts_lds[t * D + q] = (t < n_stage) ? other_pts[(size_t)t * D + q] : 0.f;
            }
        } else {
            for (int t = threadIdx.x; t < n_pad; t += PTS_T) {
                pot_lds[t] = (t < n_stage) ? pot_other[c0 + t] : SK_NEG;
#pragma unroll
                for (int q = 0; q < D; ++q) pts_lds[t * D + q] = (t < n_stage) ? other_pts[(size_t)(c0 + t) * D + q] : 0.f;
            }
        }
        __syncthreads();
        const int first = wv * 4 + sub;          // lane-group (wave, sub) takes points first, first + PTS_STRIDE, ...
        if (precise) pts_accumulate<D, true>(own, pts_lds, pot_lds, first, n_stage, inv_reg, m, s_acc);
        else         pts_accumulate<D, false>(own, pts_lds, pot_lds, first, n_stage, inv_reg, m, s_acc);
    }
    // merge the 4 sub-groups of the wave (lanes l, l ^ 16, l ^ 32, l ^ 48) ...
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1)
        pts_merge(m, s_acc, __shfl_xor(m, off, 64), __shfl_xor(s_acc, off, 64), precise != 0);
    if (sub == 0) { sm[wv][ol] = m; ss[wv][ol] = s_acc; }
    __syncthreads();
    // ... then the 16 waves: lane (sub, ol) of wave 0 folds waves 4 sub .. 4 sub + 3, two more lane exchanges
    if (wv == 0) {
        m = sm[sub * (PTS_NW / 4)][ol]; s_acc = ss[sub * (PTS_NW / 4)][ol];
#pragma unroll
        for (int w = 1; w < PTS_NW / 4; ++w) pts_merge(m, s_acc, sm[sub * (PTS_NW / 4) + w][ol], ss[sub * (PTS_NW / 4) + w][ol], precise != 0);
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1)
            pts_merge(m, s_acc, __shfl_xor(m, off, 64), __shfl_xor(s_acc, off, 64), precise != 0);
        double e2 = 0.0;
        if (sub == 0 && o < n_own) {
            const double pn = logw - (m + log(s_acc));
            if (!row_update && check) {
                const double e = wgt * exp(pold - pn) - wgt;      // column marginal of the previous iterate
                e2 = e * e;
            }
            pot_new[o] = pn;
        }
        if (!row_update && check) {
            e2 = wave_sum_d(e2);
            if (lane == 0) atomicAdd(&st->err2[slot], e2);
        }
    }
}

// (kernels of sinkhorn.hip, reused through their C entry points would need a header; the three small ones
//  are restated here as they touch only the shared state layout)
__global__ void sk_pts_init(SkState* st, double* u, double* v0, double* v1, int B0, int B1, int max_iter) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B0) u[i] = 0.0;
    if (i < B1) { v0[i] = 0.0; v1[i] = 0.0; }
    if (i == 0) {
        st->done = 0; st->iters_done = max_iter; st->vfinal = (max_iter - 1) & 1; st->precise = 0;
        st->err2[0] = 0.0; st->err2[1] = 0.0; st->last_err = 1.0;
    }
}
__global__ void sk_pts_finish(SkState* st, const double* u, const double* v0, const double* v1, int B0, int B1,
                              double reg, float* f, float* g, int* iters_done, float* last_err, int pending_check,
                              int slot) {
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

template <int D>
static void pts_launch(int grid, size_t lds, hipStream_t s, const float* own, const float* other, int n_own, int n_other,
                       double inv_reg, double logw, double wgt, SkState* st, const double* pot_other,
                       const double* pot_old, double* pot_new, int row_update, int check, int slot, double stop_thr,
                       int ii, double precise_below, int stage_cap) {
    hipLaunchKernelGGL(sk_pts_pass<D>, dim3(grid), dim3(PTS_T), lds, s, own, other, n_own, n_other, inv_reg, logw, wgt, st,
                       pot_other, pot_old, pot_new, row_update, check, slot, stop_thr, ii, precise_below, stage_cap);
}

#define PTS_DISPATCH(D_, ...) \
    switch (D_) { case 1: pts_launch<1>(__VA_ARGS__); break; case 2: pts_launch<2>(__VA_ARGS__); break; \
                  case 3: pts_launch<3>(__VA_ARGS__); break; case 4: pts_launch<4>(__VA_ARGS__); break; \
                  case 5: pts_launch<5>(__VA_ARGS__); break; case 6: pts_launch<6>(__VA_ARGS__); break; \
                  case 7: pts_launch<7>(__VA_ARGS__); break; default: pts_launch<8>(__VA_ARGS__); break; }

extern "C" int cfm_sinkhorn_log_points_f32(const float* x0, const float* x1, int B0, int B1, int d, double reg,
                                           int max_iter, double stop_thr, int check_every, float* f, float* g,
                                           int* iters_done, float* last_err, void* ws, void* stream) {
    if (!x0 || !x1 || !ws || B0 <= 0 || B1 <= 0 || d < 1 || d > 8 || !(reg > 0.0) || max_iter < 0 || check_every <= 0)
        return CFM_EINVAL;
    if (((uintptr_t)ws & 15) != 0) return CFM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    // workspace layout of sinkhorn.hip: state (256 B) | u [B0] | v[0] [B1] | v[1] [B1] | strip partials (unused here)
    char* p = (char*)ws;
    SkState* st = (SkState*)p; p += 256;
    double* u = (double*)p; p += sizeof(double) * (size_t)B0;
    double* v[2]; v[0] = (double*)p; p += sizeof(double) * (size_t)B1; v[1] = (double*)p;
    const double inv_reg = 1.0 / reg;
    const double a = 1.0 / B0, b = 1.0 / B1;
    const double precise_below = 1e-4 / sqrt((double)B1);
    const double loga = log(a), logb = log(b);
    // stage as much of the other cloud as fits 128 KiB of LDS (8 B potential + 4 d B coordinates per point)
    const int n_max = B0 > B1 ? B0 : B1;
    int stage_cap = (128 * 1024) / (8 + 4 * d);
    stage_cap = stage_cap / (PTS_STRIDE * PTS_U) * (PTS_STRIDE * PTS_U);          // whole trips
    if (stage_cap > n_max) stage_cap = (n_max + PTS_STRIDE * PTS_U - 1) / (PTS_STRIDE * PTS_U) * (PTS_STRIDE * PTS_U);
    const size_t lds = (size_t)stage_cap * (8 + 4 * d);
    {
        static std::once_flag once_d[CFM_MAX_DEVICES];          // the attribute is per device
        std::call_once(once_d[cfm_device_index()], [] {
            const void* fns[8] = {(const void*)sk_pts_pass<1>, (const void*)sk_pts_pass<2>, (const void*)sk_pts_pass<3>,
                                  (const void*)sk_pts_pass<4>, (const void*)sk_pts_pass<5>, (const void*)sk_pts_pass<6>,
                                  (const void*)sk_pts_pass<7>, (const void*)sk_pts_pass<8>};
            for (int q = 0; q < 8; ++q) (void)hipFuncSetAttribute(fns[q], hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
            (void)hipGetLastError();
        });
    }
    const int n = n_max;
    hipLaunchKernelGGL(sk_pts_init, dim3((n + 255) / 256), dim3(256), 0, s, st, u, v[0], v[1], B0, B1, max_iter);
    const int col_grid = (B1 + PTS_OWN - 1) / PTS_OWN, row_grid = (B0 + PTS_OWN - 1) / PTS_OWN;
    const bool poll = max_iter > 4096;
    int host_done = 0;
    for (int ii = 0; ii <= max_iter; ++ii) {
        const int check = (ii >= 1) && (((ii - 1) % check_every) == 0);
        const int slot = ii & 1;
        const bool trailing = (ii == max_iter);
        if (trailing && !check) break;
        // column update: own = x1 (v), other = x0 with u
        PTS_DISPATCH(d, col_grid, lds, s, x1, x0, B1, B0, inv_reg, logb, b, st, u, v[(ii + 1) & 1], v[ii & 1], 0, check, slot,
                     stop_thr, ii, precise_below, stage_cap);
        if (trailing) break;
        // row update: own = x0 (u), other = x1 with the new v
        PTS_DISPATCH(d, row_grid, lds, s, x0, x1, B0, B1, inv_reg, loga, a, st, v[ii & 1], u, u, 1, check, slot,
                     stop_thr, ii, precise_below, stage_cap);
        if (poll && (ii & 511) == 511) {
            int rc = cfm_hip(hipMemcpyAsync(&host_done, &st->done, sizeof(int), hipMemcpyDeviceToHost, s));
            if (rc) return rc;
            rc = cfm_hip(hipStreamSynchronize(s));
            if (rc) return rc;
            if (host_done) break;
        }
    }
    const int pending = (max_iter >= 1) && (((max_iter - 1) % check_every) == 0);
    hipLaunchKernelGGL(sk_pts_finish, dim3((n + 255) / 256), dim3(256), 0, s, st, u, v[0], v[1], B0, B1, reg, f, g,
                       iters_done, last_err, pending, max_iter & 1);
    return cfm_status();
}
