// transport.hip — K4r: exact optimal transport between uniform marginals of DIFFERENT sizes.
//
// Replaces pot.emd(a, b, M) (torchcfm/optimal_transport.py:49,79,87) for x0.shape[0] != x1.shape[0]: masses 1 / B0 on
// the rows, 1 / B1 on the columns.  In integer units (g = gcd(B0, B1)): every row supplies p = B1 / g units, every
// column takes q = B0 / g, one unit = 1 / lcm(B0, B1) of mass — the TRANSPORTATION problem on the B0 x B1 matrix itself,
// not the lcm x lcm assignment problem (optimal_transport.py's exact_plan_rect takes that route up to lcm = 8192 only).
//
// Method (round 6; rounds 3-5 ran one shortest augmenting path per search on ONE wavefront: 914 searches and 96 ms at
// 127 x 128): the primal-dual method with a TREE PUSH per phase, one workgroup of 16 waves, all node state in LDS.
// Orientation: R <= C (the smaller side supplies; a transposed copy is made when B0 > B1).  State: integer flows x on
// the R x C grid, row duals u, column duals v with reduced costs c - u - v >= 0 and flow only on tight entries; rows
// with supply left ("sources"), columns with demand left ("sinks").  A phase:
//   distances  d(.) = the shortest residual distance of every node TO the nearest sink, by label-correcting sweeps
//              (Bellman-Ford, everything in parallel): a row relaxes over all columns (d_i = min_j rc_ij + d_j, a wave
//              per row), a non-sink column takes the smallest label of the rows it carries flow from (backward entries
//              cost 0; over an arc list of the support, a thread per arc, 64-bit LDS atomic minima).  Queue form: a row
//              only looks at the columns whose label moved in the previous sweep (a thread per row while that list is
//              short, a wave per row beyond), a column only at rows that moved.  A label only ever
//              moves on a STRICT decrease, so the next-hop pointers (row -> column nh, column -> row nr) form a forest
//              into the sinks, zero-cost cycles included (the classical predecessor-subgraph argument, which holds for
//              relaxations on stale values).
//   duals      u_i += min(d_i, D), v_j -= min(d_j, D), D = the largest source label: feasibility is kept, every forest
//              arc below D becomes tight.
//   push       EVERY source sends ALL its supply down the forest at once: rows forward everything (their arcs are
//              uncapacitated), a column forwards up to the flow of its backward entry, a sink absorbs up to its demand;
//              one hop per round, integer atomics.  What is stuck in front of a saturated entry or a full sink goes
//              back ONE hop to the rows that sent it (they keep it as supply; a row in the middle of a path may become
//              a source that way — the pseudo-flow stays complementary-slack).  At least one unit per sink with a
//              source in its tree arrives: the phases terminate.
// 127 x 128 from the greedy start: 56 phases / 1.3 k sweeps (numpy prototype, scratch/tp_proto.py; equal to HiGHS on
// every case tried) instead of 914 searches / 55 k row relaxations.
//   warm start (optional, `sigma`): an optimal assignment of the R rows to distinct columns (the square solver on the
//              matrix padded with C - R zero rows: cfm_assign_exact_f32) — its duals are recovered by label correcting
//              (v_j <= v_sigma(i) + c_ij - c_i,sigma(i)), every row fills its column, and what is left is p - q units per
//              row against the C - R open columns: for R = C - 1 (127 vs 128, 511 vs 512: the sizes the lcm route cannot
//              take) ONE phase — a single shortest-path forest into the one open column carries every row's last unit.
//   finish     a chip-wide pass writes the plan (units / lcm) and checks in fp64: c - u - v >= -tol everywhere, = 0 on
//              the support, row sums p, column sums q.
// Exactness never depends on the warm start (an invalid or non-optimal sigma is detected and ignored).
#include "cfm_common.h"
#include <mutex>
#include <stdlib.h>

#define TP_NMAX 2048            // B0 + B1
#define TP_T 1024
#define TP_NW (TP_T / 64)
#define TP_ARC_PER_NODE 6       // capacity of the support's arc list (a basic solution has < R + C arcs; pushes add a few)
#define TP_INF_BITS 0x7ff0000000000000ull

struct TpArgs {
    const float* M; int R, C, p, q;      // oriented: R <= C, row-major R x C
    int* x;                              // flows [R * C] (global: the export pass reads them; the solver works there unless staged)
    double* u; double* v;                // duals, exported for the certificate
    float* cmax;                         // max |M|
    const int* sigma;                    // optional warm start (row -> distinct column), or nullptr
    int* info; int stage; int max_phases;
    int verify;                          // debugging aid (CFM_TP_VERIFY=1): re-derive every label densely after each phase's sweeps
};

struct TpL {
    double *u, *dr, *v;
    unsigned long long* dkey;            // bits of the column's label d_j (>= +0: orders like the value)
    int *e, *nh, *pushed, *rchg, *rq, *f, *t, *tin, *nr, *nrk, *chg, *clist;
    unsigned* arcs; int* misc;
    float* Ms; int* xs;                  // staged matrix / flows (row stride C / C + 1)
};
// misc slots
#define TP_M_CHANGED 0
#define TP_M_NARC 1
#define TP_M_MOVED 2
#define TP_M_ERR 3
#define TP_M_ANY 4
#define TP_M_NCL 5       // columns whose label moved in the last sweep (clist)
#define TP_M_NRQ 6       // rows in the queue (warm start's dual recovery)

static inline size_t tp_state_bytes(int R, int C) {
    const size_t ecap = (size_t)TP_ARC_PER_NODE * (R + C);
    return (size_t)R * (8 + 8 + 4 * 5) + (size_t)C * (8 + 8 + 4 * 7) + ecap * 4 + 64 + 64;
}
static inline size_t tp_stage_bytes(int R, int C) { return (size_t)R * C * 4 + (size_t)R * (C + 1) * 4 + 32; }

__device__ __forceinline__ TpL tp_carve(char* z, int R, int C, bool stage) {
    TpL L;
    L.u = (double*)z; z += 8 * (size_t)R;
    L.dr = (double*)z; z += 8 * (size_t)R;
    L.v = (double*)z; z += 8 * (size_t)C;
    L.dkey = (unsigned long long*)z; z += 8 * (size_t)C;
    L.e = (int*)z; z += 4 * (size_t)R;
    L.nh = (int*)z; z += 4 * (size_t)R;
    L.pushed = (int*)z; z += 4 * (size_t)R;
    L.rchg = (int*)z; z += 4 * (size_t)R;
    L.rq = (int*)z; z += 4 * (size_t)R;
    L.f = (int*)z; z += 4 * (size_t)C;
    L.t = (int*)z; z += 4 * (size_t)C;
    L.tin = (int*)z; z += 4 * (size_t)C;
    L.nr = (int*)z; z += 4 * (size_t)C;
    L.nrk = (int*)z; z += 4 * (size_t)C;
    L.chg = (int*)z; z += 4 * (size_t)C;
    L.clist = (int*)z; z += 4 * (size_t)C;
    L.misc = (int*)z; z += 64;
    L.arcs = (unsigned*)z; z += 4 * (size_t)TP_ARC_PER_NODE * (R + C);
    z = (char*)(((uintptr_t)z + 15) & ~(uintptr_t)15);
    L.Ms = nullptr; L.xs = nullptr;
    if (stage) { L.Ms = (float*)z; z += 4 * (size_t)R * C; L.xs = (int*)z; }
    return L;
}

// wave64 DPP reductions (row_shr within 16-lane rows, then row_bcast 15 / 31; the result is read from lane 63): the
// shuffle form costs 18 dependent LDS-crossbar round trips per arg-min (measured in this kernel's first version: 10 us
// per sweep at 127 x 128, the reductions of 8 rows per wave one after the other)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int tp_dpp_i(int oldv, int v) { return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROWMASK, 0xf, false); }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double tp_dpp_d(double oldv, double v) {
    const int lo = tp_dpp_i<CTRL, ROWMASK>(__double2loint(oldv), __double2loint(v));
    const int hi = tp_dpp_i<CTRL, ROWMASK>(__double2hiint(oldv), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tp_wave_min_d(double v) {
    v = fmin(v, tp_dpp_d<0x111, 0xf>(INFINITY, v));
    v = fmin(v, tp_dpp_d<0x112, 0xf>(INFINITY, v));
    v = fmin(v, tp_dpp_d<0x114, 0xf>(INFINITY, v));
    v = fmin(v, tp_dpp_d<0x118, 0xf>(INFINITY, v));
    v = fmin(v, tp_dpp_d<0x142, 0xa>(INFINITY, v));
    v = fmin(v, tp_dpp_d<0x143, 0xc>(INFINITY, v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tp_wave_min_i(int v) {
    v = min(v, tp_dpp_i<0x111, 0xf>(0x7fffffff, v));
    v = min(v, tp_dpp_i<0x112, 0xf>(0x7fffffff, v));
    v = min(v, tp_dpp_i<0x114, 0xf>(0x7fffffff, v));
    v = min(v, tp_dpp_i<0x118, 0xf>(0x7fffffff, v));
    v = min(v, tp_dpp_i<0x142, 0xa>(0x7fffffff, v));
    v = min(v, tp_dpp_i<0x143, 0xc>(0x7fffffff, v));
    return __builtin_amdgcn_readlane(v, 63);
}
// wave arg-min of (value, index), ties to the smaller index; result uniform
__device__ __forceinline__ void tp_argmin(double& d, int& j) {
    const double dm = tp_wave_min_d(d);
    j = tp_wave_min_i(d == dm ? j : 0x7fffffff);
    d = dm;
}
// inclusive prefix sum over the wave
__device__ __forceinline__ int tp_scan(int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if ((int)(threadIdx.x & 63) >= o) v += t; }
    return v;
}

// The matrix and the flows.  STAGE: both in LDS.  Otherwise the flows live in global memory and every access is a
// device-scope atomic (relaxed): the pushes ADD to an entry with an atomic (performed at the L2) while other waves of the
// workgroup read it a step later — a plain load could be served from a stale L1 line (the first version did: units were
// lost now and then beyond the LDS-staged sizes).
template <bool STAGE>
struct TpMat {
    const float* M; int* x; int C, xs;       // xs: row stride of the flows
    __device__ __forceinline__ float c(int i, int j) const { return M[(size_t)i * C + j]; }
    __device__ __forceinline__ int* xp(int i, int j) const { return x + (size_t)i * xs + j; }
    __device__ __forceinline__ int xld(int i, int j) const {
        return STAGE ? *xp(i, j) : __hip_atomic_load(xp(i, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void xst(int i, int j, int v) const {
        if (STAGE) *xp(i, j) = v; else __hip_atomic_store(xp(i, j), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ int xadd(int i, int j, int m) const {          // returns the old value
        return STAGE ? atomicAdd(xp(i, j), m) : __hip_atomic_fetch_add(xp(i, j), m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

__device__ __forceinline__ void tp_arc_append(const TpL& L, int ecap, int i, int j) {
    const int k = atomicAdd(&L.misc[TP_M_NARC], 1);
    if (k < ecap) L.arcs[k] = ((unsigned)i << 16) | (unsigned)j;
    // (past the capacity: the list is rebuilt from the flows before it is read again — tp_arcs_rebuild)
}

// d_i = min over the listed columns of rc_ij + d_j for ONE row by one wave (lanes over the list; dense: the list is every
// column).  Returns the best (uniform); ties to the smaller column.
template <bool STAGE, bool DENSE>
__device__ __forceinline__ void tp_row_relax_wave(const TpL& L, const TpMat<STAGE>& A, int i, int C, int ncl, int lane, double& bd, int& bj) {
    const double ui = L.u[i];
    bd = INFINITY; bj = 0x7fffffff;
    const int cnt = DENSE ? C : ncl;
    for (int k0 = 0; k0 < cnt; k0 += 64 * 8) {
        int jj[8]; float c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                  // 8 matrix entries in flight per lane
            const int k = k0 + 64 * q + lane;
            jj[q] = (k < cnt) ? (DENSE ? k : L.clist[k]) : -1;
            c[q] = (jj[q] >= 0) ? A.c(i, jj[q]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (jj[q] < 0) continue;
            const unsigned long long kj = L.dkey[jj[q]];
            if (kj >= TP_INF_BITS) continue;
            const double cand = fmax(((double)c[q] - ui) - L.v[jj[q]], 0.0) + __longlong_as_double((long long)kj);
            if (cand < bd || (cand == bd && jj[q] < bj)) { bd = cand; bj = jj[q]; }
        }
    }
    tp_argmin(bd, bj);
}

template <bool STAGE>
__global__ __launch_bounds__(TP_T) void tp_pd_solve(TpArgs P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int R = P.R, C = P.C, p = P.p, q = P.q;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ double red[TP_NW];
    const TpL L = tp_carve(lds, R, C, STAGE);
    const int ecap = TP_ARC_PER_NODE * (R + C);
    TpMat<STAGE> A;
    A.C = C;
    if (STAGE) {
        for (size_t k = tid; k < (size_t)R * C; k += TP_T) L.Ms[k] = P.M[k];
        A.M = L.Ms; A.x = L.xs; A.xs = C + 1;          // (odd stride: a column of the flows is read without bank conflicts)
        for (size_t k = tid; k < (size_t)R * (C + 1); k += TP_T) L.xs[k] = 0;
    } else {
        A.M = P.M; A.x = P.x; A.xs = C;
        for (size_t k = tid; k < (size_t)R * C; k += TP_T) A.xst((int)(k / C), (int)(k % C), 0);
    }
    if (tid < 16) L.misc[tid] = 0;
    for (int j = tid; j < C; j += TP_T) { L.v[j] = 0.0; L.f[j] = q; L.t[j] = 0; L.tin[j] = 0; L.nr[j] = -1; L.nrk[j] = 0x7fffffff; L.chg[j] = 0; }
    for (int i = tid; i < R; i += TP_T) { L.e[i] = p; L.pushed[i] = 0; L.nh[i] = -1; L.rchg[i] = 0; }
    __syncthreads();
    // ---- max |M| (the certificate's tolerance), row minima
    float cm = 0.f;
    for (int i = wv; i < R; i += TP_NW) {
        double bd = INFINITY; int bj = 0x7fffffff;
        for (int j0 = 0; j0 < C; j0 += 64 * 8) {
            float c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u + lane; c[u] = (j < C) ? A.c(i, j) : INFINITY; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 64 * u + lane;
                if (j < C) { cm = fmaxf(cm, fabsf(c[u])); if ((double)c[u] < bd) { bd = (double)c[u]; bj = j; } }
            }
        }
        tp_argmin(bd, bj);
        if (lane == 0) { L.u[i] = bd; L.nh[i] = bj; }        // (nh: the cheapest column, for the greedy start)
    }
    cm = wave_max_f(cm);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(&L.misc[8]), __float_as_uint(cm));     // (>= 0: orders like the bits)
    __syncthreads();
    const float cmax_abs = __uint_as_float((unsigned)L.misc[8]);
    // ---- warm start: sigma must map the rows to DISTINCT columns; its duals come from label correcting
    bool warm = false;
    if (P.sigma != nullptr) {
        for (int i = tid; i < R; i += TP_T) {
            const int s = P.sigma[i];
            if (s < 0 || s >= C || atomicAdd(&L.chg[s], 1) != 0) L.misc[TP_M_ERR] = 1;
        }
        __syncthreads();
        warm = (L.misc[TP_M_ERR] == 0);
        __syncthreads();
        for (int j = tid; j < C; j += TP_T) { L.chg[j] = 0; L.dkey[j] = d2ord(0.0); }
        for (int i = tid; i < R; i += TP_T) L.rq[i] = i;
        if (tid == 0) { L.misc[TP_M_ERR] = 0; L.misc[TP_M_NRQ] = R; }
        __syncthreads();
        if (warm) {
            // v_j = min(0, min_i (v_sigma(i) - c_i,sigma(i) + c_ij)), queue form: a row relaxes all columns again only when
            // the price of ITS column moved.  No negative cycle iff sigma is an optimal assignment of the rows to its own
            // columns — more than C + 2 rounds: it is not, and is dropped.
            int rounds = 0;
            for (;;) {
                const int nrq = L.misc[TP_M_NRQ];
                if (nrq == 0) break;
                if (++rounds > C + 2) { warm = false; break; }
                for (int k = wv; k < nrq; k += TP_NW) {
                    const int i = L.rq[k], s = P.sigma[i];
                    const double base = ord2d(L.dkey[s]) - (double)A.c(i, s);
                    for (int j0 = 0; j0 < C; j0 += 64 * 8) {
                        float c[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u + lane; c[u] = (j < C) ? A.c(i, j) : 0.f; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = j0 + 64 * u + lane;
                            if (j >= C) continue;
                            const unsigned long long kk = d2ord(base + (double)c[u]);
                            if (kk < L.dkey[j]) { atomicMin(&L.dkey[j], kk); L.chg[j] = 1; }
                        }
                    }
                }
                __syncthreads();
                if (tid == 0) L.misc[TP_M_NRQ] = 0;
                __syncthreads();
                for (int i = tid; i < R; i += TP_T) if (L.chg[P.sigma[i]]) L.rq[atomicAdd(&L.misc[TP_M_NRQ], 1)] = i;
                __syncthreads();
                for (int j = tid; j < C; j += TP_T) L.chg[j] = 0;
                __syncthreads();
            }
            __syncthreads();
        }
        if (warm) {
            for (int j = tid; j < C; j += TP_T) L.v[j] = ord2d(L.dkey[j]);
            __syncthreads();
            for (int i = tid; i < R; i += TP_T) {
                const int s = P.sigma[i];
                L.u[i] = (double)A.c(i, s) - L.v[s];
                const int m = min(p, q);
                A.xst(i, s, m); L.e[i] = p - m; L.f[s] = q - m;
            }
        }
        for (int j = tid; j < C; j += TP_T) L.chg[j] = 0;
        __syncthreads();
    }
    if (!warm) {
        // greedy start: u_i = min_j c_ij, v = 0; the rows, in order, push what their cheapest column still takes
        for (int j = tid; j < C; j += TP_T) L.v[j] = 0.0;
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < R; ++i) {
                const int j = L.nh[i];
                const int m = min(L.e[i], L.f[j]);
                if (m > 0) { A.xst(i, j, m); L.e[i] -= m; L.f[j] -= m; }
            }
        }
        __syncthreads();
    }
    // the arc list of the support
    auto arcs_rebuild = [&]() {
        if (tid == 0) L.misc[TP_M_NARC] = 0;
        __syncthreads();
        for (size_t k = tid; k < (size_t)R * C; k += TP_T) {
            const int i = (int)(k / C), j = (int)(k % C);
            if (A.xld(i, j) > 0) tp_arc_append(L, ecap, i, j);
        }
        __syncthreads();
        if (tid == 0 && L.misc[TP_M_NARC] > ecap) L.misc[TP_M_ERR] = -10;      // the support itself does not fit
        __syncthreads();
    };
    arcs_rebuild();
    int phases = 0, sweeps_tot = 0, status = 1;
    // ---- phases
    for (;;) {
        // supply left?  (and the books: supply left == demand left, or units were lost)
        int se = 0, sf = 0;
        for (int i = tid; i < R; i += TP_T) se += L.e[i];
        for (int j = tid; j < C; j += TP_T) sf += L.f[j];
        se = wave_sum_i(se); sf = wave_sum_i(sf);
        if (tid == 0) { L.misc[10] = 0; L.misc[11] = 0; }
        __syncthreads();
        if (lane == 0) { atomicAdd(&L.misc[10], se); atomicAdd(&L.misc[11], sf); }
        __syncthreads();
        const int supply = L.misc[10], demand = L.misc[11];
        const int narc_now = L.misc[TP_M_NARC], err_now = L.misc[TP_M_ERR];      // (read in front of the barrier: the rebuild below resets the
        __syncthreads();                                                          //  counter, and a thread that looked later would skip its barriers)
        if (supply != demand) { status = -12; break; }
        if (supply == 0) break;
        if (err_now < 0) { status = err_now; break; }
        if (++phases > P.max_phases) { status = -5; break; }
        if (narc_now > ecap - (R + C)) {
            arcs_rebuild();
            const int e2 = L.misc[TP_M_ERR];
            __syncthreads();
            if (e2 < 0) { status = e2; break; }
        }
        // labels: sinks 0, everything else unreached; the first sweep's list = the sinks
        if (tid == 0) L.misc[TP_M_NCL] = 0;
        for (int i = tid; i < R; i += TP_T) { L.dr[i] = INFINITY; L.nh[i] = -1; L.pushed[i] = 0; L.rchg[i] = 0; }
        __syncthreads();
        for (int j = tid; j < C; j += TP_T) {
            const bool sink = L.f[j] > 0;
            L.dkey[j] = sink ? 0ull : TP_INF_BITS; L.nr[j] = -1; L.nrk[j] = 0x7fffffff; L.chg[j] = 0; L.t[j] = 0; L.tin[j] = 0;
            if (sink) L.clist[atomicAdd(&L.misc[TP_M_NCL], 1)] = j;
        }
        __syncthreads();
        int guard = 0;
        for (;;) {
            const int ncl = L.misc[TP_M_NCL];
            if (ncl == 0) break;                      // no label moved in the last sweep: converged
            if (++guard > 4 * (R + C) + 16) { status = -6; break; }
            ++sweeps_tot;
            // rows from the columns that moved
            if (ncl <= 16) {
                // a thread per row, the short list in a loop: no reduction at all
                for (int i = tid; i < R; i += TP_T) {
                    const double ui = L.u[i];
                    double bd = INFINITY; int bj = 0x7fffffff;
                    for (int k = 0; k < ncl; ++k) {
                        const int j = L.clist[k];
                        const unsigned long long kj = L.dkey[j];
                        if (kj >= TP_INF_BITS) continue;
                        const double cand = fmax(((double)A.c(i, j) - ui) - L.v[j], 0.0) + __longlong_as_double((long long)kj);
                        if (cand < bd || (cand == bd && j < bj)) { bd = cand; bj = j; }
                    }
                    const bool imp = bd < L.dr[i];
                    if (imp) { L.dr[i] = bd; L.nh[i] = bj; }
                    L.rchg[i] = imp ? 1 : 0;
                }
            } else {
                const bool dense = 4 * ncl >= C;
                for (int i = wv; i < R; i += TP_NW) {
                    double bd; int bj;
                    if (dense) tp_row_relax_wave<STAGE, true>(L, A, i, C, ncl, lane, bd, bj);
                    else tp_row_relax_wave<STAGE, false>(L, A, i, C, ncl, lane, bd, bj);
                    if (lane == 0) {
                        const bool imp = bd < L.dr[i];
                        if (imp) { L.dr[i] = bd; L.nh[i] = bj; }
                        L.rchg[i] = imp ? 1 : 0;
                    }
                }
            }
            if (tid == 0) L.misc[TP_M_ANY] = 0;
            __syncthreads();
            if (tid == 0) L.misc[TP_M_NCL] = 0;       // (behind the barrier: every thread has read this sweep's count)
            // columns from the rows they carry flow from (the arc list; lazily deleted entries are skipped)
            const int narc = min(L.misc[TP_M_NARC], ecap);
            for (int k = tid; k < narc; k += TP_T) {
                const unsigned a = L.arcs[k]; const int i = (int)(a >> 16), j = (int)(a & 0xffffu);
                if (!L.rchg[i] || L.f[j] > 0 || A.xld(i, j) <= 0) continue;
                const unsigned long long nb = (unsigned long long)__double_as_longlong(L.dr[i]);
                if (nb < L.dkey[j]) {
                    const unsigned long long old = atomicMin(&L.dkey[j], nb);
                    if (old > nb) { L.chg[j] = 1; L.misc[TP_M_ANY] = 1; }
                }
            }
            __syncthreads();
            if (L.misc[TP_M_ANY]) {
                // the row behind each lowered label (ties: the lowest row), and the next sweep's list
                for (int k = tid; k < narc; k += TP_T) {
                    const unsigned a = L.arcs[k]; const int i = (int)(a >> 16), j = (int)(a & 0xffffu);
                    if (!L.chg[j] || !L.rchg[i] || A.xld(i, j) <= 0) continue;
                    if ((unsigned long long)__double_as_longlong(L.dr[i]) == L.dkey[j]) atomicMin(&L.nrk[j], i);
                }
                __syncthreads();
                for (int j = tid; j < C; j += TP_T)
                    if (L.chg[j]) { L.nr[j] = L.nrk[j]; L.nrk[j] = 0x7fffffff; L.chg[j] = 0; L.clist[atomicAdd(&L.misc[TP_M_NCL], 1)] = j; }
            }
            __syncthreads();
        }
        if (status != 1) break;
        if (P.verify) {
            // every row label against the dense minimum, every column label against its flow rows (fixed point?)
            int mism = 0;
            for (int i = wv; i < R; i += TP_NW) {
                double bd; int bj;
                tp_row_relax_wave<STAGE, true>(L, A, i, C, C, lane, bd, bj);
                if (lane == 0 && !(bd == L.dr[i])) ++mism;
            }
            for (int j = tid; j < C; j += TP_T) {
                if (L.f[j] > 0) continue;
                double m = INFINITY;
                for (int i = 0; i < R; ++i) if (A.xld(i, j) > 0) m = fmin(m, L.dr[i]);
                if (!(m == __longlong_as_double((long long)L.dkey[j]))) ++mism;
                if (L.nr[j] >= 0 && !(L.dr[L.nr[j]] == m)) ++mism;
                if (L.nr[j] >= 0 && L.nh[L.nr[j]] == j) ++mism;          // 2-cycle
            }
            if (mism) atomicAdd(&L.misc[12], mism);
            __syncthreads();
            if (L.misc[12] && L.misc[13] == 0) { if (tid == 0) L.misc[13] = phases; }
            __syncthreads();
        }
        // ---- duals: D = the largest source label
        double dmax = 0.0; int bad = 0;
        for (int i = tid; i < R; i += TP_T) if (L.e[i] > 0) { const double d = L.dr[i]; if (!(d < INFINITY)) bad = 1; else dmax = fmax(dmax, d); }
        dmax = wave_max_d(dmax);
        if (lane == 0) red[wv] = dmax;
        bad = __syncthreads_or(bad);
        if (bad) { status = -7; break; }
        double D = red[0];
#pragma unroll
        for (int w2 = 1; w2 < TP_NW; ++w2) D = fmax(D, red[w2]);
        for (int i = tid; i < R; i += TP_T) L.u[i] += fmin(L.dr[i], D);
        for (int j = tid; j < C; j += TP_T) { const double dj = __longlong_as_double((long long)L.dkey[j]); L.v[j] -= fmin(dj, D); }
        __syncthreads();
        // ---- push: every source sends all it has down the forest
        for (int i = tid; i < R; i += TP_T) {
            const int m = L.e[i];
            if (m > 0) {
                const int j = L.nh[i];
                const int old = A.xadd(i, j, m);
                if (old == 0) tp_arc_append(L, ecap, i, j);
                atomicAdd(&L.t[j], m); L.pushed[i] = m; L.e[i] = 0;
            }
        }
        __syncthreads();
        for (int round = 0; round <= R + C + 2; ++round) {
            int moved = 0;
            for (int j = tid; j < C; j += TP_T) {
                const int tj = L.t[j];
                if (tj <= 0) continue;
                if (L.f[j] > 0) {
                    const int a = min(tj, L.f[j]);
                    L.f[j] -= a; L.t[j] = tj - a; moved = 1;
                } else if (L.nr[j] >= 0) {
                    const int i2 = L.nr[j];
                    const int cap = A.xld(i2, j), m = min(tj, cap);
                    if (m > 0) {
                        A.xadd(i2, j, -m); L.t[j] = tj - m;              // (nobody else touches (nr_j, j) in this round: nh[nr_j] != j in a forest)
                        const int j2 = L.nh[i2];
                        const int old = A.xadd(i2, j2, m);
                        if (old == 0) tp_arc_append(L, ecap, i2, j2);
                        atomicAdd(&L.pushed[i2], m); atomicAdd(&L.tin[j2], m);
                        moved = 1;
                    }
                }
            }
            moved = __syncthreads_or(moved);
            for (int j = tid; j < C; j += TP_T) { const int a = L.tin[j]; if (a) { L.t[j] += a; L.tin[j] = 0; } }
            __syncthreads();
            if (!moved) break;
        }
        // ---- what is stuck in front of a saturated entry / a full sink goes back one hop, rows in index order
        for (int j = wv; j < C; j += TP_NW) {
            int rem = L.t[j];
            if (rem <= 0) continue;                                      // (uniform)
            for (int base = 0; base < R && rem > 0; base += 64) {
                const int i = base + lane;
                const int amt = (i < R && L.nh[i] == j) ? L.pushed[i] : 0;
                const int incl = tp_scan(amt), excl = incl - amt;
                const int take = max(0, min(amt, rem - excl));
                if (take > 0) { A.xadd(i, j, -take); L.e[i] += take; L.pushed[i] -= take; }
                rem -= min(rem, __shfl(incl, 63, 64));
            }
            if (lane == 0) { L.t[j] = 0; if (rem != 0) L.misc[TP_M_ERR] = -8; }
        }
        __syncthreads();
    }
    if (status == 1 && L.misc[TP_M_ERR] < 0) status = L.misc[TP_M_ERR];
    // ---- export: duals, flows (staged), bookkeeping; the plan / certificate pass follows
    for (int j = tid; j < C; j += TP_T) { P.v[j] = L.v[j]; if (L.f[j] != 0 && status == 1) L.misc[TP_M_ERR] = -9; }
    for (int i = tid; i < R; i += TP_T) P.u[i] = L.u[i];
    if (STAGE) for (size_t k = tid; k < (size_t)R * C; k += TP_T) P.x[k] = L.xs[(k / C) * (size_t)(C + 1) + (k % C)];
    __syncthreads();
    if (tid == 0) {
        if (status == 1 && L.misc[TP_M_ERR] < 0) status = L.misc[TP_M_ERR];
        *P.cmax = cmax_abs;
        P.info[0] = status; P.info[1] = phases; P.info[2] = sweeps_tot; P.info[5] = p; P.info[6] = q;
        P.info[7] = (STAGE ? 1 : 0) | (warm ? 2 : 0) | (P.verify ? (L.misc[12] << 8) | (L.misc[13] << 20) : 0);
    }
}

// Plan + certificate: one wave per oriented row.  plan is [B0, B1]; the oriented problem is its transpose when B0 > B1.
// The cost is summed in a fixed order (rows of a wave in order, waves of a workgroup in order, workgroups in order by the
// last one to arrive): the same bits every run.
__global__ __launch_bounds__(256) void tp_export(const float* M, const int* x, const double* u, const double* v, const float* cmax,
                                                 int R, int C, int p, int q, int transposed, double* plan, double* total_cost,
                                                 int* colsum, double* costpart, int* info, unsigned* ticket) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nw = gridDim.x * 4, w0 = blockIdx.x * 4 + wv;
    const double tol = 1e-10 * fmax((double)*cmax, 1e-30);
    const double L = (double)R * (double)p;
    const bool ok = info[0] == 1;
    __shared__ double wsum[4];
    __shared__ int last;
    int bad = 0, nsup = 0; double tot = 0.0;
    for (int i = w0; i < R; i += nw) {
        const double ui = u[i];
        int rs = 0; double rt = 0.0;
        for (int j = lane; j < C; j += 64) {
            const double c = (double)M[(size_t)i * C + j];
            const int xv = x[(size_t)i * C + j];
            const double rc = (c - ui) - v[j];
            if (rc < -tol || xv < 0) ++bad;
            if (xv > 0) { if (fabs(rc) > tol) ++bad; rt += c * (double)xv; ++nsup; atomicAdd(&colsum[j], xv); }
            rs += xv;
            const size_t o = transposed ? (size_t)j * R + i : (size_t)i * C + j;
            plan[o] = ok ? (double)xv / L : 0.0;
        }
        rs = wave_sum_i(rs);
        if (rs != p) ++bad;
        tot += wave_sum_d(rt);
    }
    bad = wave_sum_i(bad); nsup = wave_sum_i(nsup);
    if (lane == 0) {
        wsum[wv] = tot;
        if (bad) atomicAdd(&info[4], bad);
        if (nsup) atomicAdd(&info[3], nsup);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        costpart[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
        __threadfence();
        last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (last) {
        // the last workgroup: cost in block order, column sums, final status
        __threadfence();
        int b2 = 0;
        for (int j = threadIdx.x; j < C; j += 256) if (__hip_atomic_load(&colsum[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != q) ++b2;
        b2 = __syncthreads_or(b2);
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (unsigned b = 0; b < gridDim.x; ++b)
                t += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(&costpart[b]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            *total_cost = t / L;
            const int viol = __hip_atomic_load(&info[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (b2 ? 1 : 0);
            info[4] = viol;
            if (info[0] == 1 && viol) info[0] = -11;
        }
    }
}

__global__ void tp_transpose(const float* M, int B0, int B1, float* Mt) {      // Mt[j][i] = M[i][j]
    __shared__ float tile[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < B0 && j < B1) ? M[(size_t)i * B1 + j] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + threadIdx.x;
        if (i < B0 && j < B1) Mt[(size_t)j * B0 + i] = tile[threadIdx.x][r];
    }
}

static int tp_gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// workspace: [Mt: R*C floats (B0 > B1 only)] [x: R*C ints] [u: R doubles] [v: C doubles] [colsum: C ints] [cmax, ticket: 256 B]
//            [costpart: 512 doubles]
extern "C" size_t cfm_tp_ws_bytes_internal(int B0, int B1) {
    const size_t N = (size_t)B0 * B1;
    const int R = B0 < B1 ? B0 : B1, C = B0 < B1 ? B1 : B0;
    return cfm_align_up(4 * N, 256) + cfm_align_up(4 * N, 256) + cfm_align_up(8 * (size_t)R, 256) + cfm_align_up(8 * (size_t)C, 256)
           + cfm_align_up(4 * (size_t)C, 256) + 256 + 4096;
}

extern "C" int cfm_transport_exact_f32(const float* M, int B0, int B1, const int* sigma, double* plan, double* total_cost,
                                       int* info, void* ws, void* stream) {
    if (!M || !plan || !info || !total_cost || !ws || B0 < 1 || B1 < 1) return CFM_EINVAL;
    if (B0 + B1 > TP_NMAX) return CFM_EINVAL;
    if (((uintptr_t)ws & 15) != 0) return CFM_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    static int raised_d[CFM_MAX_DEVICES];
    static std::once_flag once_d[CFM_MAX_DEVICES];
    const int dvi = cfm_device_index();
    int& raised = raised_d[dvi];
    std::call_once(once_d[dvi], [&raised] {
        hipError_t e = hipFuncSetAttribute((const void*)tp_pd_solve<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tp_pd_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        (void)hipGetLastError();
        raised = (e == hipSuccess) ? 1 : -1;
    });
    if (raised < 0) return CFM_EINVAL;
    const int transposed = B0 > B1 ? 1 : 0;
    const int R = transposed ? B1 : B0, C = transposed ? B0 : B1;
    const int g = tp_gcd(R, C);
    const size_t N = (size_t)R * C;
    char* z = (char*)ws;
    float* Mt = (float*)z; z += cfm_align_up(4 * N, 256);
    int* x = (int*)z; z += cfm_align_up(4 * N, 256);
    double* u = (double*)z; z += cfm_align_up(8 * (size_t)R, 256);
    double* v = (double*)z; z += cfm_align_up(8 * (size_t)C, 256);
    int* colsum = (int*)z; z += cfm_align_up(4 * (size_t)C, 256);
    float* cmax = (float*)z; unsigned* ticket = (unsigned*)(z + 16); double* costpart = (double*)(z + 256);
    int rc = cfm_hip(hipMemsetAsync(info, 0, 8 * sizeof(int), s)); if (rc) return rc;
    rc = cfm_hip(hipMemsetAsync(total_cost, 0, sizeof(double), s)); if (rc) return rc;
    rc = cfm_hip(hipMemsetAsync(colsum, 0, cfm_align_up(4 * (size_t)C, 256) + 256, s)); if (rc) return rc;
    const float* Mo = M;
    if (transposed) {
        hipLaunchKernelGGL(tp_transpose, dim3((B1 + 31) / 32, (B0 + 31) / 32), dim3(32, 8), 0, s, M, B0, B1, Mt);
        Mo = Mt;
    }
    TpArgs A;
    A.M = Mo; A.R = R; A.C = C; A.p = C / g; A.q = R / g; A.x = x; A.u = u; A.v = v; A.cmax = cmax; A.sigma = sigma; A.info = info;
    A.max_phases = 64 * (R + C) + 1024;
    { static const int v = [] { const char* e = getenv("CFM_TP_VERIFY"); return (e && e[0] == '1') ? 1 : 0; }(); A.verify = v; }
    const size_t state = tp_state_bytes(R, C), budget = 158 * 1024;
    if (state > budget) return CFM_EINVAL;
    A.stage = (state + tp_stage_bytes(R, C) <= budget) ? 1 : 0;
    const size_t lds = state + (A.stage ? tp_stage_bytes(R, C) : 0);
    if (A.stage) hipLaunchKernelGGL(tp_pd_solve<true>, dim3(1), dim3(TP_T), lds, s, A);
    else hipLaunchKernelGGL(tp_pd_solve<false>, dim3(1), dim3(TP_T), lds, s, A);
    int grid = (R + 3) / 4; if (grid > 512) grid = 512;
    hipLaunchKernelGGL(tp_export, dim3(grid), dim3(256), 0, s, Mo, x, u, v, cmax, R, C, A.p, A.q, transposed, plan, total_cost, colsum, costpart, info, ticket);
    return cfm_status();
}
