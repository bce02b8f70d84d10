// assign.hip — K4: exact optimal assignment (uniform, equal-size marginals).
//
// Replaces pot.emd(a, b, M) (torchcfm/optimal_transport.py:49,87) and
// scipy.optimize.linear_sum_assignment (:179) for the minibatch-OT coupling.
// With uniform marginals and B0 == B1 the optimal plan is a permutation / B, so
// the problem is the linear assignment problem on the fp32 cost matrix.
//
// MI355X design.  The 64 MiB (B=4096) cost matrix is Infinity-Cache resident, a
// full sweep costs ~15-25 us, kernel boundaries ~1.5 us: the algorithm is built
// from wide, cheap sweeps and a tiny sequential control step, all device
// resident.  The host only pumps a fixed (asg_wide, asg_ctrl) kernel pair — replayed
// as a hipGraph, the kernels take nothing but the workspace — and polls 64 bytes.
// Every kernel of the chain starts cold, so each is organised as two dependent global
// hops (everything a hop needs is requested together, speculatively if need be).
//
//   init     Jonker-Volgenant row + column reduction: u_i = min_j c_ij,
//            p_j = max_i (u_i - c_ij) (every column tight for some row) — the auction
//            then starts at eps = 8e-3 of the cost range instead of 0.2.
//   phase A  epsilon-scaling forward auction, Jacobi rounds (wave <-> row, a row bids
//            iff it is unmatched; prices staged in LDS, the bidder's 16 float4 per
//            lane in flight before the barrier, branch-free fp64 top-2, DPP wave
//            reduction, one 64-bit atomicMax per bid).
//            Each epsilon phase is cut when <= 2 % of the rows are unassigned —
//            the phases only have to produce good prices.
//   phase B  the same rounds with epsilon = 0 (Jonker-Volgenant "augmenting row
//            reduction"): every kept pair is exactly tight, duals are exactly
//            feasible (up to fp64 rounding).
//   phase C  shortest augmenting paths for the remaining free rows.  First the free
//            columns are "column reduced" (their stale auction prices are lowered until
//            each is tight for some row: a pure dual ascent step).  Then MULTI-SOURCE
//            rounds: one batched Bellman-Ford label-correcting search is grown from ALL
//            free rows at once (a shortest-path forest, one tree per free row; every
//            round relaxes all dirty rows, one lane per column: single writer, no
//            atomics), and ONE path per tree that reached a free column is augmented
//            (the trees are vertex disjoint; the dual update with the radius D = the
//            longest accepted path makes every accepted path tight).  A phase costs the
//            hop depth of one search but retires many free rows.  The last few (hard)
//            rows go to the one-workgroup candidate-list solver (n <= 4096).
//   phase D  fp64 certificate: dual feasibility + complementary slackness over
//            the whole matrix, total cost.
//
// Exactness comes from phases B-D (fp64 on exactly the fp32 costs the caller
// passed); phase A is a heuristic warm start.  Any winner among simultaneous
// bidders is a valid Gauss-Seidel order, so the fp32-rounded bid in the atomic
// key only affects speed.
#include "cfm_common.h"
#include <stdlib.h>
#include <string.h>

enum { MODE_INIT = 0, MODE_AUCTION = 1, MODE_ARR = 2, MODE_SAP = 3, MODE_CERT = 4, MODE_DONE = 5,
       MODE_BUILD = 6, MODE_SAP1 = 7, MODE_SAP1_DONE = 8,
       MODE_UMIN = 9,      // u_i = min_k (c_ik + p_k) for every row (before the column reduction)
       MODE_COLRED = 10,   // lower the price of every free column until it is tight
       MODE_ROOTMIN = 11,  // u_r for the free rows of the next multi-source phase
       MODE_UMIN0 = 12,    // initial dual: u_i = min_j c_ij
       MODE_INITRED = 13 };// initial prices: p_j = max_i (u_i - c_ij)   (row + column reduction)

struct AsgParams {
    double theta;          // epsilon reduction factor
    double eps0_frac;      // first epsilon  = eps0_frac  * (cmax - cmin)
    double eps_last_frac;  // last epsilon  >= eps_last_frac * (cmax - cmin)
    double stop_frac;      // cut a phase when unassigned <= stop_frac * n
    int round_cap;         // max rounds per epsilon phase
    int arr_cap;           // max epsilon = 0 rounds
    int chunk;             // kernel pairs per host poll
    int max_pairs;         // safety cap on kernel pairs
    int sparse;            // 1: the last free rows go to the one-workgroup candidate-list solver (n <= 4096)
    int handoff;           // ... once at most this many free rows are left
    double ms_q;           // radius of a multi-source phase: quantile of the free-column labels
    double stop_early;     // stop_frac of every epsilon phase but the last (0 = same as stop_frac; a looser
                           // cut saves auction rounds but measured slower overall: 5.9 vs 4.8 ms at 0.05)
};

static AsgParams g_params = {5.0, 8e-3, 1e-6, 0.02, 4000, 15, 64, 400000, 1, 6, 1.0, 0.0};

extern "C" void cfm_assign_set_params(double theta, double eps0_frac, double eps_last_frac,
                                      double stop_frac, int round_cap, int arr_cap, int chunk) {
    if (theta > 1.0) g_params.theta = theta;
    if (eps0_frac > 0) g_params.eps0_frac = eps0_frac;
    if (eps_last_frac > 0) g_params.eps_last_frac = eps_last_frac;
    if (stop_frac >= 0) g_params.stop_frac = stop_frac;
    if (round_cap > 0) g_params.round_cap = round_cap;
    if (arr_cap >= 0) g_params.arr_cap = arr_cap;
    if (chunk > 0) g_params.chunk = chunk;
}

extern "C" void cfm_assign_set_mode(int sparse) { g_params.sparse = sparse ? 1 : 0; }
// Upper bound on the workgroups of asg_wide (0 = none).  The kernel runs one 1024-thread workgroup
// per CU (128 VGPRs), so a 256-workgroup launch needs the whole chip; with several couplings in
// flight on different streams a smaller grid lets their kernels run side by side.
static int g_wide_blocks_cap = 0;
extern "C" void cfm_assign_set_wide_blocks(int cap) { g_wide_blocks_cap = cap > 0 ? cap : 0; }
extern "C" void cfm_assign_set_handoff(int handoff) { if (handoff >= 0) g_params.handoff = handoff; }
extern "C" void cfm_assign_set_stop_early(double f) { if (f >= 0.0 && f < 1.0) g_params.stop_early = f; }
extern "C" void cfm_assign_set_ms_quantile(double q) { if (q > 0.0 && q <= 1.0) g_params.ms_q = q; }
// (cfm_assign_debug_times, below the state definition: microseconds per state-machine mode of the
//  last solve on a workspace)

struct AsgState {
    int mode, n, phase, round;
    // [offset 16] the cost matrix: kernels take it from here (same cache line as `mode`) so that their
    // launch arguments depend on the workspace only and the kernel pairs can be replayed as a hipGraph
    const float* Mptr;
    int nU, stop, arr_round, error;
    int nF, fidx, i0, nS;
    int jfree, certified, cert_bad, nN;
    // stats
    int st_auction_rounds, st_arr_rounds, st_free_after_arr, st_sap_batches;
    int st_sap_row_scans, st_total_row_scans, st_steps, cur;
    double eps, eps_last, theta, stop_frac;
    double cmin, cmax, dfree, total_cost;
    unsigned long long minslack_ord, pad2;
    unsigned cmin_bits, cmax_bits;  // ordered-float atomics
    int round_cap, arr_cap;
    int nFC, sparse;
    int st_dense_fallbacks, handoff;
    int st_ms_phases, st_ms_augmented;
    double ms_q;
    double stop_early;
    int wide_blocks, pad5;    // grid of asg_wide
    // time accounting (100 MHz device clock): every controller launch books the time since the
    // previous one on the mode that pair of launches ran in
    long long t_prev;
    long long t_acc[16];
    int t_ctrl[16];       // ... and the controller's own share of it
};
static_assert(sizeof(AsgState) <= 512, "AsgState has 512 bytes at the head of the workspace");

// Tuning aid, not part of the ABI: microseconds the last solve on `ws` spent in each mode (slots
// 0-15, index = MODE_*: wide launch + controller launch + gaps) and, of that, inside the controller
// kernel itself (slots 16-31).  Blocking.
extern "C" int cfm_assign_debug_times(const void* ws, double* us32) {
    if (!ws || !us32) return CFM_EINVAL;
    AsgState h;
    int rc = cfm_hip(hipMemcpy(&h, ws, sizeof(h), hipMemcpyDeviceToHost));
    if (rc) return rc;
    for (int q = 0; q < 16; ++q) { us32[q] = (double)h.t_acc[q] * 0.01; us32[16 + q] = (double)h.t_ctrl[q] * 0.01; }
    return 0;
}

// SAP scan list entry arrays (two copies: current / next)
struct SList {
    int* col;       // column j
    int* row;       // owner[j]
    double* base;   // dist[j] when it was listed
    double* rj;     // c[row,j] + p[j]  (= u_row: matched edge is tight)
    int* root;      // tree of the entry (index of its free row in listF)
};

struct AsgWs {
    AsgState* st;
    double* p;        // prices (= -v)
    double* bidval;   // per bidder
    double* dist;     // SAP labels
    unsigned long long* packed;  // per object: (fp32 bid bits << 32) | (row+1)
    int* a;           // row -> col (or -1)
    int* owner;       // col -> row (or -1)
    int* bidcol;
    int* listA;       // unassigned rows (current)
    int* listF;       // free rows snapshot for SAP
    int* listFC;      // free columns during SAP
    int* pred;
    int* tcol;        // per tree: accepted free column of the phase (or -1)
    int* grp_ticket;  // [n/64] arrival counters of a split relax round (one per column group)
    int* out_perm;    // [n] result, exported to the caller's buffers once the solve is done
    int* out_misc;    // [16]: certified, stats[8], pad, total_cost (double at [12])
    double* part_d;   // [MS_YMAX][n] partial minima of a split relax round
    int* part_i;      // [MS_YMAX][n] their rows
    int* part_r;      // [MS_YMAX][n] their trees
    SList S[2];
    // candidate lists (n <= SP_NMAX): SP_K columns / costs per row, bound of the dropped ones
    uint2* cl;        // {column, fp32 cost bits}
    double* cT;
};

// S[c] without dynamic indexing of the by-value kernel argument (which would push the whole
// struct into scratch memory)
__device__ __forceinline__ SList slist(const AsgWs& w, int c) {
    SList L;
    L.col = c ? w.S[1].col : w.S[0].col; L.row = c ? w.S[1].row : w.S[0].row;
    L.base = c ? w.S[1].base : w.S[0].base; L.rj = c ? w.S[1].rj : w.S[0].rj;
    L.root = c ? w.S[1].root : w.S[0].root;
    return L;
}

#ifndef MS_YMAX
#define MS_YMAX 4
#endif
#ifndef MS_SPLIT_MIN
#define MS_SPLIT_MIN 256
#endif
// MS_YMAX: a big relax round is split over this many workgroups per column group
// MS_SPLIT_MIN: ... when it has more than this many entries

static inline size_t asg_ws_bytes(int n) {
    size_t N = (size_t)n;
    size_t lists = (n <= 4096) ? N * 64 * 8 + 8 * N : 0;
    return 512 + 8 * N * (4 + 4) + 4 * N * (10 + 6) + 64 + 16 + 16 * N * MS_YMAX + lists + 256;
}

static inline AsgWs asg_carve(void* ws, int n) {
    AsgWs w; char* q = (char*)ws; size_t N = (size_t)n;
    w.st = (AsgState*)q; q += 512;
    w.p = (double*)q; q += 8 * N;
    w.bidval = (double*)q; q += 8 * N;
    w.dist = (double*)q; q += 8 * N;
    w.packed = (unsigned long long*)q; q += 8 * N;
    for (int c = 0; c < 2; ++c) { w.S[c].base = (double*)q; q += 8 * N; w.S[c].rj = (double*)q; q += 8 * N; }
    w.a = (int*)q; q += 4 * N;
    w.owner = (int*)q; q += 4 * N;
    w.bidcol = (int*)q; q += 4 * N;
    w.listA = (int*)q; q += 4 * N;
    w.listF = (int*)q; q += 4 * N;
    w.listFC = (int*)q; q += 4 * N;
    w.pred = (int*)q; q += 4 * N;
    w.tcol = (int*)q; q += 4 * N;
    for (int c = 0; c < 2; ++c) {
        w.S[c].col = (int*)q; q += 4 * N; w.S[c].row = (int*)q; q += 4 * N; w.S[c].root = (int*)q; q += 4 * N;
    }
    w.grp_ticket = (int*)q; q += 4 * N;
    w.out_perm = (int*)q; q += 4 * N;
    w.out_misc = (int*)q; q += 64;
    q = (char*)(((uintptr_t)q + 15) & ~(uintptr_t)15);
    w.part_d = (double*)q; q += 8 * N * MS_YMAX;
    w.part_i = (int*)q; q += 4 * N * MS_YMAX;
    w.part_r = (int*)q; q += 4 * N * MS_YMAX;
    w.cT = (double*)q; w.cl = nullptr;
    if (n <= 4096) { q += 8 * N; w.cl = (uint2*)q; q += 8 * N * 64; }
    return w;
}

extern "C" size_t cfm_asg_ws_bytes_internal(int n) { return asg_ws_bytes(n); }

// ordered bits for floats (total order)
__device__ __forceinline__ unsigned f2ord(float x) {
    unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// -------------------------------------------------------------- min / max ----
__global__ __launch_bounds__(256) void asg_minmax(const float* __restrict__ M, size_t n2,
                                                  AsgState* st) {
    float lo = INFINITY, hi = -INFINITY;
    const size_t n4 = n2 / 4;
    const float4* M4 = reinterpret_cast<const float4*>(M);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = M4[i];
        lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
        hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
        lo = fminf(lo, M[i]); hi = fmaxf(hi, M[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        hi = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
        atomicMin(&st->cmin_bits, f2ord(lo));
        atomicMax(&st->cmax_bits, f2ord(hi));
    }
}

// --------------------------------------------------------- wide: auction -----
#define WT 1024   // threads of the wide kernel (16 waves)
#define WIDE_PLDS_MAX 8192   // prices are staged into LDS for the bid rounds up to this n
// One wave per bidding row.  r_k = c_ik + p_k (fp64).  Top-2 over the row.
struct Top2 { double b; double s; int j; };

// branch-free (fp64 min / max are single instructions): same result as
//   if (r < b) { s = b; b = r; j = jr; } else if (r < s) s = r;
__device__ __forceinline__ void top2_push(Top2& t, double r, int j) {
    const double hi = fmax(t.b, r);
    t.j = (r < t.b) ? j : t.j;
    t.b = fmin(t.b, r);
    t.s = fmin(t.s, hi);
}

// wave64 DPP reductions (row_shr within 16-lane rows, then row_bcast 15 / 31); result uniform
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int asg_dpp_i(int oldv, int v) {
    return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double asg_dpp_d(double oldv, double v) {
    const int lo = asg_dpp_i<CTRL, ROWMASK>(__double2loint(oldv), __double2loint(v));
    const int hi = asg_dpp_i<CTRL, ROWMASK>(__double2hiint(oldv), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double asg_wave_min_d(double v) {
    v = fmin(v, asg_dpp_d<0x111, 0xf>(INFINITY, v));
    v = fmin(v, asg_dpp_d<0x112, 0xf>(INFINITY, v));
    v = fmin(v, asg_dpp_d<0x114, 0xf>(INFINITY, v));
    v = fmin(v, asg_dpp_d<0x118, 0xf>(INFINITY, v));
    v = fmin(v, asg_dpp_d<0x142, 0xa>(INFINITY, v));
    v = fmin(v, asg_dpp_d<0x143, 0xc>(INFINITY, v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int asg_wave_min_i(int v) {
    v = min(v, asg_dpp_i<0x111, 0xf>(0x7fffffff, v));
    v = min(v, asg_dpp_i<0x112, 0xf>(0x7fffffff, v));
    v = min(v, asg_dpp_i<0x114, 0xf>(0x7fffffff, v));
    v = min(v, asg_dpp_i<0x118, 0xf>(0x7fffffff, v));
    v = min(v, asg_dpp_i<0x142, 0xa>(0x7fffffff, v));
    v = min(v, asg_dpp_i<0x143, 0xc>(0x7fffffff, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ Top2 top2_merge(const Top2& x, double b2, double s2, int j2) {
    Top2 o;
    const bool take2 = (b2 < x.b) || (b2 == x.b && j2 < x.j);
    if (take2) { o.b = b2; o.j = j2; o.s = fmin(x.b, s2); }
    else       { o.b = x.b; o.j = x.j; o.s = fmin(x.s, b2); }
    return o;
}

// 64 columns of one row segment: top-2 of c + p with the prices in LDS
__device__ __forceinline__ void bid_segment(Top2& best, const float4 (&c)[16], const double* ps, int jbase) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int j = jbase + 256 * k;
        const double2 pa = *reinterpret_cast<const double2*>(ps + 256 * k);
        const double2 pb = *reinterpret_cast<const double2*>(ps + 256 * k + 2);
        top2_push(best, (double)c[k].x + pa.x, j + 0);
        top2_push(best, (double)c[k].y + pa.y, j + 1);
        top2_push(best, (double)c[k].z + pb.x, j + 2);
        top2_push(best, (double)c[k].w + pb.y, j + 3);
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads from piling up
    }
}

// wave top-2 -> bid (lane 0)
__device__ __forceinline__ void bid_commit(const AsgWs& w, Top2 best, int i, double eps, const double* p_lds,
                                           bool use_plds) {
    const double bmin = asg_wave_min_d(best.b);
    const int jwin = asg_wave_min_i(best.b == bmin ? best.j : 0x7fffffff);
    const double rest = (best.b == bmin && best.j == jwin) ? best.s : best.b;
    const double smin = asg_wave_min_d(rest);       // the best of everything but (bmin, jwin)
    if ((threadIdx.x & 63) == 0) {
        const double incr = (smin - bmin) + eps;          // >= eps >= 0
        const double bv = (use_plds ? p_lds[jwin] : w.p[jwin]) + incr;
        w.bidcol[i] = jwin;
        w.bidval[i] = bv;
        // prices may be negative (they start from the row + column reduction): ordered float bits
        const unsigned long long key =
            ((unsigned long long)f2ord((float)bv) << 32) | (unsigned)(i + 1);
        atomicMax(&w.packed[jwin], key);
    }
}

// One wave per bidding row.  `pre_a` is the (speculatively loaded) match of this wave's row,
// pst* the thread's share of the prices on their way into LDS: the bidder's first row segment
// is requested BEFORE the prices are written to LDS and the workgroup barrier, so the staging
// hides behind the row's latency.
__device__ __forceinline__ void wide_bid(const float* __restrict__ M, const AsgWs& w, const AsgState* st,
                         int wave_gid, int n_waves, int pre_a, bool stage_p, int n_host,
                         double2 pst0, double2 pst1, double2 pst2, double2 pst3) {
    extern __shared__ __attribute__((aligned(16))) char wide_lds_bid[];   // = the kernel's dynamic LDS
    double* p_lds = reinterpret_cast<double*>(wide_lds_bid);
    const int n = n_host;
    const double eps = st->eps;        // same 128-byte line as st->mode: an L1 hit by now
    const int lane = threadIdx.x & 63;
    const bool vec = ((n & 3) == 0);
    const bool fast = stage_p && (n & 4095) == 0;
    // No bidder list: wave <-> row, a row bids iff it is unmatched (pre_a = a[wave_gid] came with
    // the state block).  Rows beyond the first of a wave (n > number of waves) are checked as they come.
    int i = wave_gid;
    bool bids = (i < n) && (pre_a < 0);
    float4 c[16];
    if (fast && bids) {
        const float* rs = M + (size_t)i * n + lane * 4;
#pragma unroll
        for (int k = 0; k < 16; ++k) c[k] = *reinterpret_cast<const float4*>(rs + 256 * k);
    }
    if (stage_p) {
        const int j0 = threadIdx.x * 2, j1 = j0 + 2 * WT, j2 = j0 + 4 * WT, j3 = j0 + 6 * WT;
        if (j0 < n_host) *reinterpret_cast<double2*>(p_lds + j0) = pst0;
        if (j1 < n_host) *reinterpret_cast<double2*>(p_lds + j1) = pst1;
        if (j2 < n_host) *reinterpret_cast<double2*>(p_lds + j2) = pst2;
        if (j3 < n_host) *reinterpret_cast<double2*>(p_lds + j3) = pst3;
        __syncthreads();
    }
    if (fast) {
        while (i < n) {
            if (bids) {
                Top2 best; best.b = INFINITY; best.s = INFINITY; best.j = 0x7fffffff;
                bid_segment(best, c, p_lds + lane * 4, lane * 4);
                for (int seg = 4096; seg < n; seg += 4096) {
                    const float* rs = M + (size_t)i * n + seg + lane * 4;
#pragma unroll
                    for (int k = 0; k < 16; ++k) c[k] = *reinterpret_cast<const float4*>(rs + 256 * k);
                    bid_segment(best, c, p_lds + seg + lane * 4, seg + lane * 4);
                }
                bid_commit(w, best, i, eps, p_lds, true);
            }
            i += n_waves;
            bids = (i < n) && (w.a[i] < 0);
            if (bids) {
                const float* rs = M + (size_t)i * n + lane * 4;
#pragma unroll
                for (int k = 0; k < 16; ++k) c[k] = *reinterpret_cast<const float4*>(rs + 256 * k);
            }
        }
        return;
    }
    for (; i < n; i += n_waves) {
        if (!((i == wave_gid) ? (pre_a < 0) : (w.a[i] < 0))) continue;
        const float* row = M + (size_t)i * n;
        Top2 best; best.b = INFINITY; best.s = INFINITY; best.j = 0x7fffffff;
        if (vec) {
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                // 4 float4 in flight per lane per trip
                float4 c4[4]; double2 pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        c4[k] = *reinterpret_cast<const float4*>(row + j);
                        pa[k] = stage_p ? *reinterpret_cast<const double2*>(p_lds + j) : *reinterpret_cast<const double2*>(w.p + j);
                        pb[k] = stage_p ? *reinterpret_cast<const double2*>(p_lds + j + 2) : *reinterpret_cast<const double2*>(w.p + j + 2);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        top2_push(best, (double)c4[k].x + pa[k].x, j + 0);
                        top2_push(best, (double)c4[k].y + pa[k].y, j + 1);
                        top2_push(best, (double)c4[k].z + pb[k].x, j + 2);
                        top2_push(best, (double)c4[k].w + pb[k].y, j + 3);
                    }
                }
            }
        } else {
            for (int j = lane; j < n; j += 64) top2_push(best, (double)row[j] + w.p[j], j);
        }
        bid_commit(w, best, i, eps, p_lds, stage_p);
    }
}

// ------------------------------------------------------------ wide: SAP ------
// Row minima u_i = min_k (c_ik + p_k), one wave per row: for every row (before the column
// reduction) or for the free rows of the next multi-source phase.  Result in bidval[row].
__device__ __forceinline__ void wide_umin(const float* __restrict__ M, const AsgWs& w, const AsgState* st,
                          int wave_gid, int n_waves, bool roots_only) {
    const int n = st->n;
    const int cnt = roots_only ? st->nF : n;
    const int lane = threadIdx.x & 63;
    const bool vec = ((n & 3) == 0);
    for (int t = wave_gid; t < cnt; t += n_waves) {
        const int i = roots_only ? w.listF[t] : t;
        const float* row = M + (size_t)i * n;
        double m = INFINITY;
        if (vec) {
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                float4 c[4]; double2 pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        c[k] = *reinterpret_cast<const float4*>(row + j);
                        pa[k] = *reinterpret_cast<const double2*>(w.p + j);
                        pb[k] = *reinterpret_cast<const double2*>(w.p + j + 2);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        m = fmin(fmin(m, (double)c[k].x + pa[k].x), (double)c[k].y + pa[k].y);
                        m = fmin(fmin(m, (double)c[k].z + pb[k].x), (double)c[k].w + pb[k].y);
                    }
                }
            }
        } else {
            for (int j = lane; j < n; j += 64) m = fmin(m, (double)row[j] + w.p[j]);
        }
        m = wave_min_d(m);
        if (lane == 0) w.bidval[i] = m;
    }
}

// Column reduction of the free columns: p_k <- max_i (u_i - c_ik), the largest price at which
// column k is still not cheaper than any row's current minimum.  No u_i changes, every matched
// edge stays tight, the dual objective rises by the price drop.  One workgroup per column
// (strided reads: 64 B sector per row, only nFC columns).
__device__ __forceinline__ void wide_colred(const float* __restrict__ M, const AsgWs& w, const AsgState* st,
                            double* sh_d) {
    const int n = st->n, nFC = st->nFC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < nFC; t += gridDim.x) {
        const int k = w.listFC[t];
        double m = -INFINITY;
        for (int i0 = threadIdx.x; i0 < n; i0 += WT * 4) {
            float c[4]; double u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q * WT;
                c[q] = (i < n) ? M[(size_t)i * n + k] : 0.f;
                u[q] = (i < n) ? w.bidval[i] : -INFINITY;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmax(m, u[q] - (double)c[q]);
        }
        m = wave_max_d(m);
        if (lane == 0) sh_d[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = sh_d[0];
            for (int q = 1; q < WT / 64; ++q) r = fmax(r, sh_d[q]);
            if (r < w.p[k]) w.p[k] = r;      // prices of free columns only ever go down here
        }
        __syncthreads();
    }
}

// Initial prices by row + column reduction (the classical Jonker-Volgenant start): with
// u_i = min_j c_ij (bidval[], MODE_UMIN0) the price p_j = max_i (u_i - c_ij) <= 0 is the largest
// one that keeps every row's minimum where it is, and it makes every column tight for some row.
// The auction then starts two epsilon phases later (eps0 = 8e-3 instead of 0.2 of the cost range):
// 78 rounds instead of 113 on the C3 data.  Lane <-> column, the grid splits the rows, partial
// maxima through one ordered-double atomicMax per column and workgroup (packed[] is the scratch).
__device__ __forceinline__ void wide_initred(const float* __restrict__ M, const AsgWs& w, const AsgState* st,
                                             double* sh_d, int n) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_groups = (n + 63) / 64;
    int Y = gridDim.x / n_groups; Y = Y < 1 ? 1 : Y;
    constexpr int NW = WT / 64, Q = 8;
    for (int unit = blockIdx.x; unit < n_groups * Y; unit += gridDim.x) {
        const int g = unit % n_groups, y = unit / n_groups;
        const int k = g * 64 + lane;
        const bool ok = k < n;
        const int r_beg = (int)((long long)n * y / Y), r_end = (int)((long long)n * (y + 1) / Y);
        double m = -INFINITY;
        for (int r0 = r_beg + wv * Q; r0 < r_end; r0 += NW * Q) {
            float c[Q]; double u[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int r = r0 + q;
                const bool v = ok && r < r_end;
                c[q] = v ? M[(size_t)r * n + k] : 0.f;
                u[q] = (r < r_end) ? w.bidval[r] : -INFINITY;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) m = fmax(m, u[q] - (double)c[q]);
        }
        sh_d[wv * 64 + lane] = m;
        __syncthreads();
        if (wv == 0 && ok) {
#pragma unroll
            for (int q = 1; q < NW; ++q) m = fmax(m, sh_d[q * 64 + lane]);
            atomicMax(&w.packed[k], d2ord(m));
        }
        __syncthreads();
    }
}

// How many workgroups share one column group in a relax round of nS entries.
__device__ __forceinline__ int ms_split(int nS, int n_groups, int blocks) {
    if (nS <= MS_SPLIT_MIN) return 1;
    int y = blocks / n_groups;
    return y < 1 ? 1 : (y > MS_YMAX ? MS_YMAX : y);
}

// Relax every listed row.  A workgroup owns columns [64g, 64g+64): lane <-> column
// (single writer: dist/pred/tree stay consistent without atomics), its 16 waves split the
// list, 8 independent row loads in flight per lane, LDS merge.  The writer lane
// appends improved assigned columns to the NEXT list (one atomic per append).  A big round
// is split over Y workgroups per column group (every Y-th slice of the list each); they write
// per-column partial minima and the last of them to arrive merges them (see the end of the loop body).
__device__ __forceinline__ void wide_relax(const float* __restrict__ M, const AsgWs& w, AsgState* st,
                           double* sh_d, int* sh_i, int* sh_r) {
    const int n = st->n, nS = st->nS, cur = st->cur;
    const double dfree = st->dfree;
    const SList L = slist(w, cur), Nx = slist(w, cur ^ 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n_groups = (n + 63) / 64;
    constexpr int Q = 8, NW = WT / 64;
    const int Y = ms_split(nS, n_groups, gridDim.x);
    for (int unit = blockIdx.x; unit < n_groups * Y; unit += gridDim.x) {
        const int g = unit % n_groups, y = unit / n_groups;
        const int k = g * 64 + lane;
        const bool ok = k < n;
        const double pk = ok ? w.p[k] : 0.0;
        double best = INFINITY; int bi = 0x7fffffff, br = -1;
        for (int t0 = (y * NW + wv) * Q; t0 < nS; t0 += NW * Q * Y) {
            int ri[Q], cj[Q], rt[Q]; double bs[Q], rj[Q]; float c[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int t = t0 + q;
                const bool v = t < nS;
                ri[q] = v ? L.row[t] : 0; cj[q] = v ? L.col[t] : -1; rt[q] = v ? L.root[t] : -1;
                bs[q] = v ? L.base[t] : INFINITY; rj[q] = v ? L.rj[t] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                c[q] = (ok && bs[q] < dfree) ? M[(size_t)ri[q] * n + k] : 0.f;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (ok && bs[q] < dfree && k != cj[q]) {
                    double rc = ((double)c[q] + pk) - rj[q];
                    rc = fmax(rc, 0.0);                   // dual feasible up to rounding
                    const double cand = bs[q] + rc;
                    if (cand < best || (cand == best && ri[q] < bi)) { best = cand; bi = ri[q]; br = rt[q]; }
                }
            }
        }
        sh_d[wv * 64 + lane] = best; sh_i[wv * 64 + lane] = bi; sh_r[wv * 64 + lane] = br;
        __syncthreads();
        if (wv == 0 && ok) {
#pragma unroll
            for (int q = 1; q < NW; ++q) {
                const double c2 = sh_d[q * 64 + lane]; const int i2 = sh_i[q * 64 + lane];
                if (c2 < best || (c2 == best && i2 < bi)) { best = c2; bi = i2; br = sh_r[q * 64 + lane]; }
            }
            if (Y > 1) {
                w.part_d[(size_t)y * n + k] = best; w.part_i[(size_t)y * n + k] = bi; w.part_r[(size_t)y * n + k] = br;
            } else if (best < w.dist[k]) {
                // (requesting dist / owner / the owner's cost up front, next to the list entries,
                //  instead of here was measured: no change — the hops are not what bounds a round)
                w.dist[k] = best; w.pred[k] = bi;
                const int ow = w.owner[k];
                if (ow >= 0 && best < dfree) {
                    const int idx = atomicAdd(&st->nN, 1);
                    Nx.col[idx] = k; Nx.row[idx] = ow; Nx.base[idx] = best; Nx.root[idx] = br;
                    Nx.rj[idx] = (double)M[(size_t)ow * n + k] + pk;
                }
            }
        }
        __syncthreads();
        if (Y > 1) {
            // The Y workgroups of a column group hand their partial minima to the LAST one to arrive,
            // which merges them and finalises the group's 64 columns exactly as an unsplit round does
            // (one CU merging all 4096 columns in asg_ctrl was bound by that CU's bandwidth: ~20 us).
            // Hand-off: plain stores -> barrier -> lane 0: agent release, drained, device-scope ticket;
            // last arriver: agent acquire -> barrier -> plain loads.
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int t = atomicAdd(&w.grp_ticket[g], 1);
                const int last = (t == Y - 1) ? 1 : 0;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    w.grp_ticket[g] = 0;          // next use is in a later kernel
                }
                sh_i[0] = last;
            }
            __syncthreads();
            const int last = sh_i[0];
            if (last && wv == 0 && ok) {
                double mb = w.part_d[k]; int mi = w.part_i[k], mr = w.part_r[k];
                for (int yy = 1; yy < Y; ++yy) {
                    const double c2 = w.part_d[(size_t)yy * n + k]; const int i2 = w.part_i[(size_t)yy * n + k];
                    if (c2 < mb || (c2 == mb && i2 < mi)) { mb = c2; mi = i2; mr = w.part_r[(size_t)yy * n + k]; }
                }
                if (mb < w.dist[k]) {
                    w.dist[k] = mb; w.pred[k] = mi;
                    const int ow = w.owner[k];
                    if (ow >= 0 && mb < dfree) {
                        const int idx = atomicAdd(&st->nN, 1);
                        Nx.col[idx] = k; Nx.row[idx] = ow; Nx.base[idx] = mb; Nx.root[idx] = mr;
                        Nx.rj[idx] = (double)M[(size_t)ow * n + k] + pk;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------ wide: cert -----
// pre_a: the match of row wave_gid, loaded with the state block in the kernel prologue.
__device__ __forceinline__ void wide_cert(const float* __restrict__ M, const AsgWs& w, AsgState* st, int wave_gid,
                          int n_waves, int pre_a, double* sh_d) {
    const int n = st->n;
    const int lane = threadIdx.x & 63;
    double wmin = INFINITY, csum = 0.0; int bad = 0;
    for (int i = wave_gid; i < n; i += n_waves) {
        const int ai = (i == wave_gid) ? pre_a : w.a[i];
        if (ai < 0 || ai >= n || w.owner[ai] != i) { bad = 1; continue; }
        const float* row = M + (size_t)i * n;
        const double ui = (double)row[ai] + w.p[ai];
        double m = INFINITY;
        if ((n & 3) == 0) {
            // 4 float4 of the row and their prices in flight per lane and trip (the scalar loop ran
            // its 64 dependent trips at one L2 latency each: 114 us for the pass at n = 4096)
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                float4 c4[4]; double2 pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        c4[k] = *reinterpret_cast<const float4*>(row + j);
                        pa[k] = *reinterpret_cast<const double2*>(w.p + j);
                        pb[k] = *reinterpret_cast<const double2*>(w.p + j + 2);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 + 256 * k < n) {
                        m = fmin(m, fmin(fmin(((double)c4[k].x + pa[k].x) - ui, ((double)c4[k].y + pa[k].y) - ui),
                                         fmin(((double)c4[k].z + pb[k].x) - ui, ((double)c4[k].w + pb[k].y) - ui)));
                    }
                }
            }
        } else {
            for (int j = lane; j < n; j += 64) m = fmin(m, ((double)row[j] + w.p[j]) - ui);
        }
        wmin = fmin(wmin, m);
        if (lane == 0) csum += (double)row[ai];
    }
    wmin = wave_min_d(wmin);
    // one set of atomics per workgroup: 4096 waves adding into the same fp64 word serialise at the
    // L2 (the pass took 107 us at n = 4096 with the row loop already vectorised)
    const int wv = threadIdx.x >> 6;
    if (lane == 0) { sh_d[wv] = wmin; sh_d[16 + wv] = csum; sh_d[32 + wv] = bad ? 1.0 : 0.0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = sh_d[0], c = sh_d[16], b = sh_d[32];
        for (int q = 1; q < WT / 64; ++q) { m = fmin(m, sh_d[q]); c += sh_d[16 + q]; b += sh_d[32 + q]; }
        atomicMin(&st->minslack_ord, d2ord(m));
        if (c != 0.0) atomicAdd(&st->total_cost, c);
        if (b != 0.0) atomicOr(&st->cert_bad, 1);
    }
}

#include "assign_sparse.h"

__global__ __launch_bounds__(WT) void asg_wide(AsgWs w, int n_host) {
    extern __shared__ __attribute__((aligned(16))) char wide_lds[];   // modes are exclusive
    double* sh_d = reinterpret_cast<double*>(wide_lds);
    int* sh_i = reinterpret_cast<int*>(wide_lds + sizeof(double) * WT);
    int* sh_r = sh_i + WT;
    AsgState* st = w.st;
    // consecutive work items go to different workgroups (different CUs)
    const int wave_gid = (threadIdx.x >> 6) * gridDim.x + blockIdx.x;
    const int n_waves = gridDim.x * (WT / 64);
    // Every kernel of the state machine starts cold (the previous one ran elsewhere): each
    // dependent global access is a hop of its own.  Whether this wave's row is unmatched (= bids)
    // is loaded speculatively TOGETHER with the state block (n comes from the host), so a bid
    // round is {state, match, prices} -> row instead of state -> list -> {row, prices}.
    // The prices (all of them are needed by every bidder) are staged into LDS the same way, so
    // a bidder's row can be requested in one go.
    const bool stage_p = (n_host <= WIDE_PLDS_MAX) && ((n_host & 3) == 0);
    double2 pst0, pst1, pst2, pst3;     // WIDE_PLDS_MAX / (2 * WT) = 4 double2 per thread
    {
        const int j0 = threadIdx.x * 2, j1 = j0 + 2 * WT, j2 = j0 + 4 * WT, j3 = j0 + 6 * WT;
        const int nn = stage_p ? n_host : 0;
        pst0 = *reinterpret_cast<const double2*>(w.p + (j0 < nn ? j0 : 0));
        pst1 = *reinterpret_cast<const double2*>(w.p + (j1 < nn ? j1 : 0));
        pst2 = *reinterpret_cast<const double2*>(w.p + (j2 < nn ? j2 : 0));
        pst3 = *reinterpret_cast<const double2*>(w.p + (j3 < nn ? j3 : 0));
    }
    int pre_i = (wave_gid < n_host) ? w.a[wave_gid] : 0;
    int mode = st->mode;
    const float* __restrict__ M = st->Mptr;
    asm volatile("" : "+v"(pre_i), "+v"(pst0.x), "+v"(pst1.x), "+v"(pst2.x), "+v"(pst3.x) : "s"(mode) : "memory");   // all of it in flight
    if (mode == MODE_DONE || st->error) return;
    if (mode == MODE_AUCTION || mode == MODE_ARR) {
        wide_bid(M, w, st, wave_gid, n_waves, pre_i, stage_p, n_host, pst0, pst1, pst2, pst3);
    } else if (mode == MODE_SAP) wide_relax(M, w, st, sh_d, sh_i, sh_r);
    else if (mode == MODE_UMIN || mode == MODE_UMIN0) wide_umin(M, w, st, wave_gid, n_waves, false);
    else if (mode == MODE_INITRED) wide_initred(M, w, st, sh_d, n_host);
    else if (mode == MODE_ROOTMIN) wide_umin(M, w, st, wave_gid, n_waves, true);
    else if (mode == MODE_COLRED) wide_colred(M, w, st, sh_d);
    else if (mode == MODE_CERT) wide_cert(M, w, st, wave_gid, n_waves, pre_i, sh_d);
    else if (mode == MODE_BUILD) wide_build(M, w, st, wide_lds);
    else if (mode == MODE_SAP1) { if (blockIdx.x == 0) sp_solver(M, w, st, wide_lds); }
}

// ------------------------------------------------------------------ ctrl -----
#define CT 1024
// exclusive scan of one int per thread over a 1024-thread workgroup
__device__ int block_scan_excl(int v, int* total, int* sh /*>=17*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int q = 0; q < CT / 64; ++q) { int t = sh[q]; sh[q] = run; run += t; }
        sh[16] = run;
    }
    __syncthreads();
    const int res = inc - v + sh[wv];
    *total = sh[16];
    __syncthreads();
    return res;
}

// block argmin over doubles (ties -> lowest index)
__device__ void block_argmin(double v, int idx, double* out_v, int* out_i, double* shd, int* shi) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o, 64); const int i2 = __shfl_xor(idx, o, 64);
        if (v2 < v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
    if (lane == 0) { shd[wv] = v; shi[wv] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = shd[0]; int bi = shi[0];
        for (int q = 1; q < CT / 64; ++q)
            if (shd[q] < bv || (shd[q] == bv && shi[q] < bi)) { bv = shd[q]; bi = shi[q]; }
        shd[16] = bv; shi[16] = bi;
    }
    __syncthreads();
    *out_v = shd[16]; *out_i = shi[16];
    __syncthreads();
}

__device__ void ctrl_reset_assignment(const AsgWs& w, AsgState* st) {
    const int n = st->n;
    for (int i = threadIdx.x; i < n; i += CT) {
        w.a[i] = -1; w.owner[i] = -1; w.packed[i] = 0ull;
    }
    if (threadIdx.x == 0) { st->nU = n; st->round = 0; }
    __syncthreads();
}

// Block sum of one int per thread.
__device__ int block_sum_i(int v, int* sh /*>=17*/) {
    v = wave_sum_i(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int tot = 0;
#pragma unroll
    for (int q = 0; q < CT / 64; ++q) tot += sh[q];
    __syncthreads();
    return tot;
}

// Apply the winning bids.  There is no bidder list (a row bids iff it is unmatched), so all that
// is left to maintain is the number of unmatched rows: it drops by one for every object that was
// free before.  Two global hops: {bids, owners} of the thread's objects, then the winners' exact bids.
__device__ void ctrl_award(const AsgWs& w, AsgState* st, int* sh, int n, int nU_old) {
    int newly = 0;
    for (int j0 = threadIdx.x; j0 < n; j0 += 4 * CT) {
        unsigned long long key[4]; int ow[4]; double bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = j0 + q * CT;
            const bool ok = j < n;
            key[q] = ok ? w.packed[j] : 0ull; ow[q] = ok ? w.owner[j] : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = (key[q] != 0ull) ? w.bidval[(int)(key[q] & 0xffffffffull) - 1] : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (key[q] != 0ull) {
                const int j = j0 + q * CT;
                const int i = (int)(key[q] & 0xffffffffull) - 1;
                w.p[j] = bv[q]; w.owner[j] = i; w.a[i] = j; w.packed[j] = 0ull;
                if (ow[q] >= 0) w.a[ow[q]] = -1; else ++newly;
            }
        }
    }
    const int tot = block_sum_i(newly, sh);
    if (threadIdx.x == 0) st->nU = nU_old - tot;
    __syncthreads();
}

// Pruning radius of a multi-source phase.  One tree: the best free-column label (no label
// at or above it can matter).  Several trees: the q-quantile of the free-column labels (q = 1:
// the largest).  Any radius is valid — labels at or below it are final once no listed entry is
// below it, and only free columns at or below it are accepted — a smaller one trades fewer
// augmentations per phase for far fewer relaxations of rows whose labels are still poor.  It
// only ever decreases within a phase, so an entry skipped once is never needed later.
__device__ void ctrl_radius(const AsgWs& w, AsgState* st, double* shd, int* shi, double* scratch) {
    const int nFC = st->nFC;
    const bool single = (st->nF == 1);
    const double q = st->ms_q;
    if (!single && nFC <= CT && q < 1.0) {
        const int t = threadIdx.x;
        const double v = (t < nFC) ? w.dist[w.listFC[t]] : INFINITY;
        __syncthreads();
        if (t < nFC) scratch[t] = v;
        __syncthreads();
        int kq = (int)ceil(q * nFC) - 1;
        kq = kq < 0 ? 0 : (kq > nFC - 1 ? nFC - 1 : kq);
        if (t < nFC) {
            int rank = 0;
            for (int s2 = 0; s2 < nFC; ++s2) {
                const double vs = scratch[s2];
                rank += (vs < v || (vs == v && s2 < t)) ? 1 : 0;
            }
            if (rank == kq) { st->dfree = v; st->jfree = -1; }
        }
        __syncthreads();
        return;
    }
    double lm = INFINITY; int li = 0x7fffffff;
    for (int t = threadIdx.x; t < nFC; t += CT) {
        const int k = w.listFC[t];
        const double dk = single ? w.dist[k] : -w.dist[k];
        if (dk < lm || (dk == lm && k < li)) { lm = dk; li = k; }
    }
    double r; int jr;
    block_argmin(lm, li, &r, &jr, shd, shi);
    if (threadIdx.x == 0) { st->dfree = single ? r : -r; st->jfree = jr; }
    __syncthreads();
}

#define MS_NONE 0x7fffffff

// Start a multi-source phase: labels unset, one root entry per free row (u_r in bidval[]).
__device__ void ctrl_ms_begin(const AsgWs& w, AsgState* st) {
    const int n = st->n, nF = st->nF, cur = st->cur;
    const SList L = slist(w, cur);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += CT) { w.dist[k] = INFINITY; w.pred[k] = -1; }
    for (int g = threadIdx.x; g < (n + 63) / 64; g += CT) w.grp_ticket[g] = 0;
    for (int t = threadIdx.x; t < nF; t += CT) {
        const int r = w.listF[t];
        L.col[t] = -1; L.row[t] = r; L.base[t] = 0.0; L.rj[t] = w.bidval[r]; L.root[t] = t;
        w.listA[r] = t;                 // row -> tree index (listA is free after the auction)
        w.tcol[t] = MS_NONE; w.packed[t] = ~0ull;
    }
    if (threadIdx.x == 0) {
        st->nS = nF; st->nN = 0; st->dfree = INFINITY; st->jfree = -1; st->mode = MODE_SAP;
        st->st_ms_phases++;
    }
    __syncthreads();
}

// After a relax round: new radius, swap lists; returns true if another round is needed.
__device__ bool ctrl_sap_step(const AsgWs& w, AsgState* st, double* shd, int* shi, double* scratch) {
    ctrl_radius(w, st, shd, shi, scratch);
    const double dfree = st->dfree;
    const int nN = st->nN, cur = st->cur;
    const SList Nx = slist(w, cur ^ 1);
    int any = 0;
    for (int t = threadIdx.x; t < nN; t += CT) any |= (Nx.base[t] < dfree) ? 1 : 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) { st->cur = cur ^ 1; st->nS = any ? nN : 0; st->nN = 0; }
    __syncthreads();
    return any != 0;
}

// The forest has converged below the radius.  Every free column with a finite label belongs to
// exactly one tree (walk the predecessors to its free row); each tree accepts its nearest free
// column (ties: lowest column).  With D = the largest accepted label, the dual update
//     p_k += D - d_k  for every column with d_k < D   (u follows through the tight matched edges)
// keeps the duals feasible and makes every accepted path tight; the paths are vertex disjoint
// (different trees), so all of them are augmented.  Returns the number of augmented paths.
__device__ int ctrl_ms_finish(const AsgWs& w, AsgState* st, int* lds_a, int* lds_pred, bool use_lds,
                              double* shd, int* shi, int* sh) {
    const int n = st->n, nF = st->nF, nFC = st->nFC;
    const double radius = st->dfree;
    __syncthreads();
    if (use_lds) {
        for (int k = threadIdx.x; k < n; k += CT) { lds_pred[k] = w.pred[k]; lds_a[k] = w.a[k]; }
        __syncthreads();
    }
    const int* A = use_lds ? lds_a : w.a;
    const int* P = use_lds ? lds_pred : w.pred;
    int bad = 0;
    // pass 1: tree of every reached free column, per-tree best label
    for (int t = threadIdx.x; t < nFC; t += CT) {
        const int k = w.listFC[t];
        const double d = w.dist[k];
        int ti = -1;
        if (d < INFINITY && d <= radius) {      // labels above the radius are not final
            int j = k, guard = 0;
            for (;;) {
                const int i = P[j];
                if (i < 0 || ++guard > n + 1) { bad = 1; break; }
                const int aj = A[i];
                if (aj < 0) { ti = w.listA[i]; break; }
                j = aj;
            }
            if (ti >= 0) atomicMin(&w.packed[ti], d2ord(d));
        }
        w.bidcol[t] = ti;
    }
    bad = __syncthreads_or(bad);
    if (bad) { if (threadIdx.x == 0) st->error = 3; __syncthreads(); return 0; }
    // pass 2: accepted column of every tree
    for (int t = threadIdx.x; t < nFC; t += CT) {
        const int ti = w.bidcol[t];
        if (ti >= 0) {
            const int k = w.listFC[t];
            if (d2ord(w.dist[k]) == w.packed[ti]) atomicMin(&w.tcol[ti], k);
        }
    }
    __syncthreads();
    // radius D = largest accepted label
    double lm = INFINITY; int cnt = 0;
    for (int t = threadIdx.x; t < nF; t += CT) {
        if (w.tcol[t] != MS_NONE) { lm = fmin(lm, -ord2d(w.packed[t])); ++cnt; }
    }
    double negD; int dummy;
    block_argmin(lm, threadIdx.x, &negD, &dummy, shd, shi);
    int total;
    block_scan_excl(cnt, &total, sh);
    if (total == 0) { if (threadIdx.x == 0) st->error = 5; __syncthreads(); return 0; }
    const double D = -negD;
    for (int k = threadIdx.x; k < n; k += CT) {
        const double dk = w.dist[k];
        if (dk < D) w.p[k] += D - dk;
    }
    // augment (one thread per accepted tree; the paths are vertex disjoint)
    for (int t = threadIdx.x; t < nF; t += CT) {
        int j = w.tcol[t];
        if (j == MS_NONE) continue;
        const int r = w.listF[t];
        int guard = 0; bool closed = false;
        while (guard++ <= n) {
            const int i = P[j];
            const int jprev = A[i];          // the OLD match of row i (lds copy / not yet overwritten)
            w.owner[j] = i; w.a[i] = j;
            if (jprev < 0) { closed = (i == r); break; }
            j = jprev;
        }
        if (!closed) bad = 1;
    }
    bad = __syncthreads_or(bad);
    if (bad) { if (threadIdx.x == 0) st->error = 3; __syncthreads(); return 0; }
    // drop the matched rows / columns from the free lists (in-place, order preserving)
    int baseF = 0;
    for (int t0 = 0; t0 < nF; t0 += CT) {
        const int t = t0 + threadIdx.x;
        int r = -1, f = 0;
        if (t < nF) { r = w.listF[t]; f = (w.a[r] < 0) ? 1 : 0; }
        int tot;
        const int off = block_scan_excl(f, &tot, sh);
        if (f) w.listF[baseF + off] = r;
        baseF += tot;
    }
    int baseC = 0;
    for (int t0 = 0; t0 < nFC; t0 += CT) {
        const int t = t0 + threadIdx.x;
        int k = -1, f = 0;
        if (t < nFC) { k = w.listFC[t]; f = (w.owner[k] < 0) ? 1 : 0; }
        int tot;
        const int off = block_scan_excl(f, &tot, sh);
        if (f) w.listFC[baseC + off] = k;
        baseC += tot;
    }
    if (threadIdx.x == 0) {
        st->nF = baseF; st->nFC = baseC; st->fidx = 0;
        st->st_ms_augmented += total;
        st->st_total_row_scans += total;
        if (baseF != baseC || baseF != nF - total) st->error = 4;
    }
    __syncthreads();
    return total;
}

__device__ void ctrl_enter_cert(AsgState* st) {
    if (threadIdx.x == 0) {
        st->mode = MODE_CERT; st->minslack_ord = ~0ull; st->total_cost = 0.0; st->cert_bad = 0;
    }
}

__device__ __forceinline__ void asg_ctrl_body(const AsgWs& w, int* dyn) {
    __shared__ int sh[32];
    __shared__ double shd[32];
    __shared__ int shi[32];
    AsgState* st = w.st;
    const int n = st->n;
    int mode = st->mode;
    const float* __restrict__ M = st->Mptr;
    int* perm = w.out_perm;
    int* certified = w.out_misc;
    int* stats = w.out_misc + 1;
    double* total_cost = reinterpret_cast<double*>(w.out_misc + 12);
    const bool use_lds = (n <= 6144);
    if (mode == MODE_DONE || st->error) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        st->st_steps++;
        const long long now = wall_clock64();
        if (st->t_prev) st->t_acc[mode & 15] += now - st->t_prev;
        st->t_prev = now;
    }
    __syncthreads();

    if (mode == MODE_INIT) {
        if (threadIdx.x == 0) {
            st->cmin = (double)ord2f(st->cmin_bits);
            st->cmax = (double)ord2f(st->cmax_bits);
            double cr = st->cmax - st->cmin;
            if (!(cr > 0.0)) cr = 1.0;
            st->eps = cr * st->eps;          // eps/eps_last hold the fractions on entry
            st->eps_last = cr * st->eps_last;
            // every phase but the last is cut earlier: it only has to shape the prices
            const bool first_is_last = (st->eps / st->theta) < st->eps_last;
            st->stop = (int)((first_is_last ? st->stop_frac : fmax(st->stop_frac, st->stop_early)) * n);
            st->mode = MODE_UMIN0; st->phase = 0;
        }
        for (int k = threadIdx.x; k < n; k += CT) { w.p[k] = 0.0; w.packed[k] = 0ull; }
        return;
    }
    if (mode == MODE_UMIN0) {          // bidval[i] = min_j c_ij
        if (threadIdx.x == 0) { st->mode = MODE_INITRED; st->st_total_row_scans += n; }
        return;
    }
    if (mode == MODE_INITRED) {        // packed[k] = ordered max_i (u_i - c_ik)
        for (int k = threadIdx.x; k < n; k += CT) w.p[k] = ord2d(w.packed[k]);
        __syncthreads();
        if (threadIdx.x == 0) { st->mode = MODE_AUCTION; st->st_total_row_scans += n; }
        ctrl_reset_assignment(w, st);
        return;
    }

    if (mode == MODE_AUCTION || mode == MODE_ARR) {
        // snapshot everything the decision needs BEFORE thread 0 mutates the state
        const int bidders = st->nU, round = st->round, round_cap = st->round_cap, stop = st->stop;
        const int arr_round = st->arr_round, arr_cap = st->arr_cap;
        const double eps_cur = st->eps, eps_last = st->eps_last, theta = st->theta;
        __syncthreads();
        ctrl_award(w, st, sh, n, bidders);
        const int nU = st->nU;
        __syncthreads();
        if (threadIdx.x == 0) st->st_total_row_scans += bidders;
        if (mode == MODE_AUCTION) {
            const bool next_phase = (nU <= stop) || (round + 1 >= round_cap);
            if (threadIdx.x == 0) { st->round = round + 1; st->st_auction_rounds++; }
            __syncthreads();
            if (next_phase) {
                const double e2 = eps_cur / theta;
                if (e2 < eps_last) {
                    if (threadIdx.x == 0) { st->mode = MODE_ARR; st->eps = 0.0; st->arr_round = 0; }
                } else {
                    if (threadIdx.x == 0) {
                        st->eps = e2; st->phase++;
                        const bool is_last = (e2 / theta) < eps_last;
                        st->stop = (int)((is_last ? st->stop_frac : fmax(st->stop_frac, st->stop_early)) * n);
                    }
                }
                __syncthreads();
                ctrl_reset_assignment(w, st);
            }
            return;
        }
        // MODE_ARR
        if (threadIdx.x == 0) { st->arr_round = arr_round + 1; st->st_arr_rounds++; }
        __syncthreads();
        if (nU == 0) {
            if (threadIdx.x == 0) st->st_free_after_arr = 0;
            ctrl_enter_cert(st);
            return;
        }
        if (arr_round + 1 < arr_cap) return;
        // -> phase C: snapshot the free rows and the free columns, then column reduction
        {
            int baseR = 0;
            for (int i0 = 0; i0 < n; i0 += CT) {
                const int i = i0 + threadIdx.x;
                const int f = (i < n && w.a[i] < 0) ? 1 : 0;
                int tot;
                const int off = block_scan_excl(f, &tot, sh);
                if (f) w.listF[baseR + off] = i;
                baseR += tot;
            }
            if (baseR != nU && threadIdx.x == 0) st->error = 4;
        }
        {
            int base = 0;
            for (int k0 = 0; k0 < n; k0 += CT) {
                const int k = k0 + threadIdx.x;
                const int f = (k < n && w.owner[k] < 0) ? 1 : 0;
                int tot;
                const int off = block_scan_excl(f, &tot, sh);
                if (f) w.listFC[base + off] = k;
                base += tot;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                st->nFC = base; st->nF = nU; st->fidx = 0; st->st_free_after_arr = nU;
                st->mode = MODE_UMIN; st->cur = 0; st->nN = 0;
                if (base != nU) st->error = 4;
            }
        }
        return;
    }

    if (mode == MODE_UMIN) {           // bidval[i] = u_i for every row
        if (threadIdx.x == 0) { st->mode = MODE_COLRED; st->st_total_row_scans += n; }
        return;
    }
    if (mode == MODE_BUILD) {          // the wide pass has written the candidate lists
        if (threadIdx.x == 0) st->mode = MODE_SAP1;
        return;
    }
    if (mode == MODE_SAP1_DONE) {      // the one-workgroup solver has matched every row
        ctrl_enter_cert(st);
        return;
    }
    if (mode == MODE_COLRED || mode == MODE_ROOTMIN) {
        // free-column prices are reduced / the roots' u_r are known: hand the last few rows to
        // the candidate-list solver, otherwise grow a forest from all free rows
        const int nF = st->nF, sparse = st->sparse, handoff = st->handoff;
        __syncthreads();
        if (sparse && nF <= handoff) { if (threadIdx.x == 0) st->mode = MODE_BUILD; return; }
        ctrl_ms_begin(w, st);
        return;
    }
    if (mode == MODE_SAP) {
        const int scanned = st->nS;
        __syncthreads();
        if (threadIdx.x == 0) { st->st_sap_batches++; st->st_sap_row_scans += scanned; st->st_total_row_scans += scanned; }
        if (ctrl_sap_step(w, st, shd, shi, reinterpret_cast<double*>(dyn))) return;
        // converged below the radius: accept one path per tree
        ctrl_ms_finish(w, st, dyn, dyn + n, use_lds, shd, shi, sh);
        if (st->error) return;
        const int nF = st->nF;
        __syncthreads();
        if (nF == 0) { ctrl_enter_cert(st); return; }
        if (threadIdx.x == 0) st->mode = (st->sparse && nF <= st->handoff) ? MODE_BUILD : MODE_ROOTMIN;
        return;
    }

    if (mode == MODE_CERT) {
        // the wide pass has filled minslack / total_cost
        const double minslack = ord2d(st->minslack_ord);
        if (threadIdx.x == 0) st->st_total_row_scans += n;
        const double scale = fmax(fabs(st->cmax), fabs(st->cmin));
        const double tol = 1e-10 * fmax(scale, 1e-30);
        for (int i = threadIdx.x; i < n; i += CT) perm[i] = w.a[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            const int ok = (!st->cert_bad) && (minslack >= -tol);
            st->certified = ok;
            if (certified) *certified = ok;
            if (total_cost) *total_cost = st->total_cost;
            if (stats) {
                stats[0] = st->st_auction_rounds; stats[1] = st->st_arr_rounds;
                stats[2] = st->st_free_after_arr; stats[3] = st->st_sap_batches;
                stats[4] = st->st_sap_row_scans; stats[5] = st->st_total_row_scans;
                stats[6] = st->st_steps;
                stats[7] = (st->phase & 0xff) | ((st->st_ms_phases & 0xff) << 8) | (st->st_dense_fallbacks << 16);
            }
            __threadfence();
            st->mode = MODE_DONE;
        }
        return;
    }
}

__global__ __launch_bounds__(CT) void asg_ctrl(AsgWs w) {
    extern __shared__ __attribute__((aligned(16))) int dyn[];   // 2*n ints for the path walk
    const int mode0 = w.st->mode;
    asg_ctrl_body(w, dyn);
    __syncthreads();
    if (threadIdx.x == 0 && mode0 != MODE_DONE)        // t_prev = this launch's start
        w.st->t_ctrl[mode0 & 15] += (int)(wall_clock64() - w.st->t_prev);
}

// trivial sizes
__global__ void asg_trivial(const float* M, int n, int* perm, int* certified, double* total_cost,
                            int* stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (n == 1) { perm[0] = 0; if (total_cost) *total_cost = (double)M[0]; }
        if (certified) *certified = 1;
        if (stats) for (int k = 0; k < 8; ++k) stats[k] = 0;
    }
}

// copy the workspace-resident results to the caller's buffers
__global__ void asg_export(AsgWs w, int n, int* perm, int* certified, double* total_cost, int* stats) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) perm[i] = w.out_perm[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (certified) *certified = w.out_misc[0];
        if (total_cost) *total_cost = *reinterpret_cast<const double*>(w.out_misc + 12);
        if (stats) for (int k = 0; k < 8; ++k) stats[k] = w.out_misc[1 + k];
    }
}

// The (asg_wide, asg_ctrl) pairs take only workspace-derived arguments, so a chunk of them is
// captured once per host thread / workspace into a hipGraph and replayed: a solve is ~400 kernel
// launches, and with several couplings in flight on different streams the host launch rate
// (~2.7 us per launch across threads) was the limit.  Falls back to plain launches when the stream
// cannot be captured (the legacy default stream) or CFM_ASG_GRAPH=0.
struct AsgGraph {
    void* ws = nullptr; int n = 0, pairs = 0; size_t wide_dyn = 0;
    hipGraphExec_t exec = nullptr;         // `pairs` kernel pairs: the bulk of a solve
    hipGraphExec_t exec_small = nullptr;   // ASG_TAIL_PAIRS pairs: the polled tail
    hipStream_t stream = nullptr; int disabled = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
};
#define ASG_TAIL_PAIRS 8
static thread_local AsgGraph g_graph;

// 1 (default): bulk chunks, then 8-pair chunks with one chunk of look-ahead, each followed by a
// copy of the state and an event; 0: 64-pair chunks with a blocking copy.
static int asg_tail_poll_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CFM_ASG_TAILPOLL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

static int asg_graph_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CFM_ASG_GRAPH"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}

// poll buffer (pinned host memory): one per host thread, concurrent solves on different streams
// must not share it
static thread_local int* g_pinned = nullptr;

static int asg_run(const float* M, int B, int* perm, int* certified, double* total_cost, int* stats,
                   void* ws, void* stream, int use_sparse, int* cert_out) {
    if (!M || !perm || B < 0 || (B > 1 && !ws)) return CFM_EINVAL;
    if (B > (1 << 20)) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return 0;
    if (B == 1) {
        hipLaunchKernelGGL(asg_trivial, dim3(1), dim3(64), 0, s, M, B, perm, certified, total_cost, stats);
        return cfm_status();
    }
    if (((uintptr_t)ws & 15) != 0 || ((uintptr_t)M & 15) != 0) return CFM_EALIGN;
    const int n = B;
    AsgWs w = asg_carve(ws, n);
    if (!g_pinned) {
        int rc = cfm_hip(hipHostMalloc((void**)&g_pinned, 256, hipHostMallocDefault));
        if (rc) return rc;
    }
    int wide_blocks = (n + 15) / 16;        // one wave per row when everything bids
    if (wide_blocks > 512) wide_blocks = 512;
    if (g_wide_blocks_cap > 0 && wide_blocks > g_wide_blocks_cap) wide_blocks = g_wide_blocks_cap;
    if (wide_blocks < (n + 63) / 64) wide_blocks = (n + 63) / 64;
    if (wide_blocks < 1) wide_blocks = 1;
    AsgState h;
    memset(&h, 0, sizeof(h));
    h.wide_blocks = wide_blocks;
    h.mode = MODE_INIT; h.n = n; h.Mptr = M;
    h.eps = g_params.eps0_frac; h.eps_last = g_params.eps_last_frac; h.theta = g_params.theta;
    h.stop_frac = g_params.stop_frac; h.round_cap = g_params.round_cap; h.arr_cap = g_params.arr_cap;
    h.cmin_bits = 0xffffffffu; h.cmax_bits = 0u; h.minslack_ord = ~0ull;
    h.sparse = (use_sparse && n <= SP_NMAX) ? 1 : 0;
    h.handoff = g_params.handoff; h.ms_q = g_params.ms_q; h.stop_early = g_params.stop_early;
    size_t wide_dyn = sizeof(double) * WT + 2 * sizeof(int) * WT;
    if (n <= WIDE_PLDS_MAX && (size_t)n * sizeof(double) > wide_dyn) wide_dyn = (size_t)n * sizeof(double);
    if (h.sparse) {
        const size_t need = sp_lds_bytes(n);
        if (need > wide_dyn) wide_dyn = need;
        static int raised = 0;   // dynamic LDS above the 64 KiB default needs the attribute
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)asg_wide,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = (e == hipSuccess) ? 1 : -1;
            (void)hipGetLastError();
        }
        if (raised < 0 && wide_dyn > 64 * 1024) { h.sparse = 0; wide_dyn = 64 * 1024; }
    }
    int rc = cfm_hip(hipMemcpyAsync(w.st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    if (rc) return rc;
    const size_t n2 = (size_t)n * n;
    const int mm_blocks = (int)((n2 / 4 + 255) / 256 < 1024 ? (n2 / 4 + 255) / 256 + 1 : 1024);
    hipLaunchKernelGGL(asg_minmax, dim3(mm_blocks), dim3(256), 0, s, M, n2, w.st);
    // path walks in LDS (2 n ints, n <= 6144); never less than the 1024 doubles of ctrl_radius
    const size_t dyn = (n <= 6144) ? (size_t)2 * n * sizeof(int) : (size_t)CT * sizeof(double);
    hipLaunchKernelGGL(asg_ctrl, dim3(1), dim3(CT), dyn, s, w);
    rc = cfm_status();
    if (rc) return rc;

    // one hipGraph of `chunk` pairs per (thread, workspace, n), replayed
    const int gchunk = g_params.chunk;
    AsgGraph& G = g_graph;
    bool use_graph = asg_graph_enabled() && !G.disabled && n >= 256;
    if (use_graph && !(G.exec && G.ws == ws && G.n == n && G.pairs == gchunk && G.wide_dyn == wide_dyn && G.stream == s)) {
        if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
        if (G.exec_small) { (void)hipGraphExecDestroy(G.exec_small); G.exec_small = nullptr; }
        auto capture = [&](int npairs, hipGraphExec_t* out) -> hipError_t {
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            if (e != hipSuccess) return e;
            for (int c = 0; c < npairs; ++c) {
                hipLaunchKernelGGL(asg_wide, dim3(wide_blocks), dim3(WT), wide_dyn, s, w, n);
                hipLaunchKernelGGL(asg_ctrl, dim3(1), dim3(CT), dyn, s, w);
            }
            e = hipStreamEndCapture(s, &graph);
            if (e == hipSuccess && graph) e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            return e;
        };
        hipError_t e = capture(gchunk, &G.exec);
        if (e == hipSuccess && G.exec) e = capture(ASG_TAIL_PAIRS, &G.exec_small);
        for (int q = 0; q < 2 && e == hipSuccess; ++q)
            if (!G.ev[q]) e = hipEventCreateWithFlags(&G.ev[q], hipEventDisableTiming);
        if (e != hipSuccess || !G.exec || !G.exec_small) {
            (void)hipGetLastError();
            if (G.exec) { (void)hipGraphExecDestroy(G.exec); }
            if (G.exec_small) { (void)hipGraphExecDestroy(G.exec_small); }
            G.exec = nullptr; G.exec_small = nullptr; G.disabled = 1; use_graph = false;     // e.g. the legacy default stream
        } else {
            G.ws = ws; G.n = n; G.pairs = gchunk; G.wide_dyn = wide_dyn; G.stream = s;
        }
    }

    if (use_graph && n >= 1024 && asg_tail_poll_enabled()) {
        // The bulk (a solve at n = 4096 takes 135 - 230 pairs) goes out unpolled; the tail in
        // 8-pair chunks, each followed by a 64-byte copy of the state into its own pinned slot and
        // an event, with the NEXT chunk already queued when the host waits for a slot: no idle
        // gap, and a finished solve is followed by at most ~1.5 chunks of no-op pairs.
        const int bulk = (n >= 4096) ? 2 : 1;
        for (int r = 0; r < bulk; ++r) { rc = cfm_hip(hipGraphLaunch(G.exec, s)); if (rc) return rc; }
        int pairs = bulk * gchunk, cur = 0;
        auto chunk = [&](int slot) -> int {
            int r2 = cfm_hip(hipGraphLaunch(G.exec_small, s)); if (r2) return r2;
            r2 = cfm_hip(hipMemcpyAsync(g_pinned + 16 * slot, w.st, 64, hipMemcpyDeviceToHost, s)); if (r2) return r2;
            return cfm_hip(hipEventRecord(G.ev[slot], s));
        };
        rc = chunk(0); if (rc) return rc;
        pairs += ASG_TAIL_PAIRS;
        for (;;) {
            rc = chunk(cur ^ 1); if (rc) return rc;
            pairs += ASG_TAIL_PAIRS;
            rc = cfm_hip(hipEventSynchronize(G.ev[cur])); if (rc) return rc;
            const int* hs = g_pinned + 16 * cur;
            const int mode = hs[0], err = hs[9];
            if (err) return CFM_ENOCONV;
            if (mode == MODE_DONE) { if (cert_out) *cert_out = hs[15]; break; }
            if (pairs >= g_params.max_pairs) return CFM_ETIMEOUT;
            cur ^= 1;
        }
        hipLaunchKernelGGL(asg_export, dim3((n + 1023) / 1024 < 64 ? (n + 1023) / 1024 : 64), dim3(1024), 0, s, w, n,
                           perm, certified, total_cost, stats);
        return cfm_status();
    }

    int pairs = 0;
    for (;;) {
        // a typical solve needs 150 - 300 pairs: enqueue most of them before the first poll
        const int reps = (pairs == 0 && n >= 1024) ? 3 : 1;
        for (int r = 0; r < reps; ++r) {
            if (use_graph) {
                rc = cfm_hip(hipGraphLaunch(G.exec, s));
                if (rc) return rc;
            } else {
                for (int c = 0; c < gchunk; ++c) {
                    hipLaunchKernelGGL(asg_wide, dim3(wide_blocks), dim3(WT), wide_dyn, s, w, n);
                    hipLaunchKernelGGL(asg_ctrl, dim3(1), dim3(CT), dyn, s, w);
                }
            }
        }
        pairs += reps * gchunk;
        rc = cfm_status();
        if (rc) return rc;
        rc = cfm_hip(hipMemcpyAsync(g_pinned, w.st, 64, hipMemcpyDeviceToHost, s));
        if (rc) return rc;
        rc = cfm_hip(hipStreamSynchronize(s));
        if (rc) return rc;
        const int mode = g_pinned[0], err = g_pinned[9];
        if (err) return CFM_ENOCONV;
        if (mode == MODE_DONE) { if (cert_out) *cert_out = g_pinned[15]; break; }
        if (pairs >= g_params.max_pairs) return CFM_ETIMEOUT;
    }
    hipLaunchKernelGGL(asg_export, dim3((n + 1023) / 1024 < 64 ? (n + 1023) / 1024 : 64), dim3(1024), 0, s, w, n,
                       perm, certified, total_cost, stats);
    return cfm_status();
}

extern "C" int cfm_assign_exact_f32(const float* M, int B, int* perm, int* certified,
                                    double* total_cost, int* stats, void* ws, void* stream) {
    int cert = 1;
    int rc = asg_run(M, B, perm, certified, total_cost, stats, ws, stream, g_params.sparse, &cert);
    // The candidate-list path is exact by construction; should its certificate ever fail
    // (or its solver report an inconsistency) the dense state machine decides.
    if (g_params.sparse && B > 1 && B <= SP_NMAX && (rc == CFM_ENOCONV || (rc == 0 && !cert)))
        rc = asg_run(M, B, perm, certified, total_cost, stats, ws, stream, 0, &cert);
    return rc;
}
