// assign.hip — K4: exact optimal assignment (uniform, equal-size marginals).
//
// Replaces pot.emd(a, b, M) (torchcfm/optimal_transport.py:49,87) and
// scipy.optimize.linear_sum_assignment (:179) for the minibatch-OT coupling.
// With uniform marginals and B0 == B1 the optimal plan is a permutation / B, so
// the problem is the linear assignment problem on the fp32 cost matrix.
//
// MI355X design.  The 64 MiB (B=4096) cost matrix is Infinity-Cache resident, a
// full sweep costs ~15-25 us, a kernel boundary ~1.5 us: the algorithm is a chain of
// ~200 short, dependent, chip-wide steps.  It runs as a device-resident state machine:
// every step is ONE kernel launch that takes nothing but the workspace (so the chain is
// replayed as hipGraphs), does the step's wide work on all CUs, and whose LAST-ARRIVING
// workgroup (device-scope arrival ticket) performs the step's control decision and
// publishes the next mode.  Everything a control decision needs from the other
// workgroups of its launch travels through device-scope atomics (counters, ordered
// min / max words), so no release / acquire fence is paid; everything else is read by
// the NEXT launch, where the kernel boundary is the fence.  A kernel whose family does
// not own the current mode exits at once, so the host may replay a guessed program:
// correctness never depends on the guess.  Four kernels, one per register / LDS
// profile (no oversized LDS reservation):
//
//   asg_auction  (round 5; 512 <= n <= 8192) every bid of the solve — the epsilon > 0 phases and the epsilon = 0
//              stage — in ONE launch without global rounds: see "asynchronous phase A" below.  The synchronous
//              rounds of asg_step (phases A / B as described here) remain the path of the other sizes, the
//              fallback of a failed list certificate, and the A/B reference (cfm_assign_set_async(0, ...)).
//   asg_step   every chip-wide step   UMIN0, INITRED, AUCTION, ARR, CONVERT, UMIN, COLRED, ROOTMIN,
//                                     SAP, MS_FINISH, CERT       (125 VGPRs, <= 48 KiB LDS at n = 4096)
//   asg_build  candidate lists        BUILD            (n <= 4096; 8 waves, 128 KiB of row strips)
//   asg_solve  one-workgroup list solver   SOLVER      (n <= 4096; 152 KiB of solver state: the forest phases run here)
//
//   init     Jonker-Volgenant row + column reduction: u_i = min_j c_ij,
//            p_j = max_i (u_i - c_ij) (every column tight for some row) — the auction
//            then starts at eps = 8e-3 of the cost range instead of 0.2.
//   phase A  epsilon-scaling forward auction, Jacobi rounds, one launch per round and NO
//            award step: the whole state of an object is one 64-bit key, the order-
//            preserving image of its fp64 price with the bidder's row in the low bits
//            (the row bits are part of the price: a relative perturbation below 2^-38).
//            Prices only rise in a forward auction, so atomicMax(key) IS the award.  A row
//            is matched iff the key of the object it bid for last still carries its id.
//            wave <-> row, all keys staged into LDS as prices, the bidder's 16 float4 per
//            lane in flight at once, branch-free fp64 top-2, DPP wave reduction.
//            Each epsilon phase is cut when <= 2 % of the rows were unassigned at the
//            start of a round — the phases only have to produce good prices.
//   phase B  the same rounds with epsilon = 0 (Jonker-Volgenant "augmenting row
//            reduction").  A bid is rounded DOWN onto the key grid, so the bidder's new
//            object is its strict minimum: every kept pair is exactly tight and the duals
//            (u_i = min_k c_ik + p_k) are exactly feasible.
//   phase C  shortest augmenting paths for the remaining free rows.  First the free
//            columns are "column reduced" (their stale auction prices are lowered until
//            each is tight for some row: a pure dual ascent step).  Then MULTI-SOURCE
//            phases: one batched label-correcting search is grown from ALL free rows at
//            once (a shortest-path forest, one tree per free row), and ONE path per tree
//            that reached a free column is augmented (the trees are vertex disjoint; the
//            dual update with the radius D = the longest accepted path makes every
//            accepted path tight).  A phase costs the depth of one search but retires many
//            free rows.  With <= 64 free rows (n <= 4096: always, at the benchmark shapes)
//            the phases run inside the one-workgroup candidate-list solver (assign_sparse.h);
//            with more, chip-wide Bellman-Ford rounds on the dense matrix retire rows first
//            (every round relaxes all dirty rows, one lane per column: single writer, no
//            atomics).
//   phase D  fp64 certificate: dual feasibility + complementary slackness over
//            the whole matrix, total cost.
//
//   bid lists  (n <= 4096) a row scan of phase A / B leaves the per-lane best column + cost and a bound behind;
//            later bids of the row read those 512 bytes instead of the row while the bound decides the bid
//            (bid_from_list / bid_list_store): half of the ~80 k row evaluations of a C3 solve.
//   batches  cfm_assign_exact_batch_f32: nb problems of one size in ONE chain of launches — every kernel takes the
//            carving of problem 0 and a byte stride, blockIdx.y is the problem (asg_shift), every problem runs its own
//            state machine; 256 workgroups per launch in all, the bid rounds in their queue form (wide_bid_queue).
//
// Exactness comes from phases B-D (fp64 on exactly the fp32 costs the caller
// passed); phase A is a heuristic warm start.
#include "cfm_common.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <atomic>
#include <time.h>

enum { MODE_UMIN0 = 0,     // u_i = min_j c_ij, cost range, key / bid state reset          (f1)
       MODE_INITRED = 1,   // initial prices: p_j = max_i (u_i - c_ij), straight into the keys (f1)
       MODE_AUCTION = 2,   // epsilon > 0 rounds                                             (f1)
       MODE_ARR = 3,       // epsilon = 0 rounds                                             (f1)
       MODE_CONVERT = 4,   // keys -> prices, matches, free lists (workgroup 0)              (f1)
       MODE_UMIN = 5,      // u_i = min_k (c_ik + p_k) for every row                         (f2)
       MODE_COLRED = 6,    // lower the price of every free column until it is tight         (f2)
       MODE_ROOTMIN = 7,   // u_r for the free rows + start of a multi-source phase          (f2)
       MODE_SAP = 8,       // one relax round of the forest                                  (f2)
       MODE_MS_FINISH = 9, // accept one path per tree, dual update, augment (workgroup 0)   (f2)
       MODE_CERT = 10,     // certificate pass; its last workgroup exports the result        (f2)
       MODE_BUILD = 11,    // candidate lists                                                (build)
       MODE_SOLVER = 12,   // one-workgroup list solver                                      (solve)
       MODE_DONE = 13 };

struct AsgParams {
    double theta;          // epsilon reduction factor
    double eps0_frac;      // first epsilon  = eps0_frac  * (cmax - cmin)
    double eps_last_frac;  // last epsilon  >= eps_last_frac * (cmax - cmin)
    double stop_frac;      // cut a phase when unassigned <= stop_frac * n
    int round_cap;         // max rounds per epsilon phase
    int arr_cap;           // max epsilon = 0 rounds
    int chunk;             // launches per polled chunk
    int max_launches;      // safety cap on launches
    int sparse;            // 1: the last free rows go to the one-workgroup candidate-list solver (n <= 4096)
    int handoff;           // ... once at most this many free rows are left
    double stop_early;     // stop_frac of every epsilon phase but the last (0 = same as stop_frac)
    int wide_blocks_cap;   // upper bound on the grid of the wide kernels (0 = none)
    int bulk;              // asg_step launches enqueued before the first poll (n >= bulk_min_n)
    int bulk_min_n;
    int small;             // 1: problems of 2 <= n <= 256 take the one-workgroup solver (assign_small.h)
    int async_auction;     // 1: the epsilon > 0 phases run in ONE launch without global rounds (asg_auction; async_min_n <= n <= 8192)
    int async_blocks;      // ... on this many workgroups per problem in the batch entry (0: the grid of the other kernels)
    int async_last_div;    // ... the last phase is cut at stop_frac / this
    double async_theta;    // ... its epsilon reduction factor (a phase costs it microseconds, not ~15 launches: gentler scaling pays)
    int async_min_n;       // ... smallest n it is used for (round 6: 512 — n = 512, d = 2: 2.86 ms against 3.60 on the synchronous rounds; at
                           //     n <= 256 the one-workgroup solver stays ahead: C1 0.92 against 1.68 ms; gpurun_out -> profiles/r6_experiments.txt)
};

// Process-wide tuning defaults.  A solve works on a snapshot taken under the lock, so setters
// called from another thread never tear a running solve.
static std::mutex g_params_mu;
// async_blocks / async_last_div: measured in the C3 pipelined loop, three interleaved passes of nine regions each on one box
// (profiles/r5_async_sweep.txt): synchronous rounds 1.18 - 1.21 ms per step; asynchronous on 16 workgroups per problem
// 1.10 - 1.12, with the last phase cut at a quarter of the usual 2 % 1.07 - 1.11; 24 / 32 workgroups 1.09 - 1.13; cut / 8: 1.10 - 1.11.
// async_theta: 40 C3 instances (profiles/r5_async_sweep.txt): theta 5 / 4 / 3 / 2.5 / 2: lone solve 2.33 / 2.28 / 2.08 / 1.92 / 1.93 ms — gentler
// scaling leaves the list solver 15 free rows instead of 27 and shorter searches (1.04 vs 1.56 ms) for 0.1 ms more auction; the
// sequential step 2.96 -> 2.56 ms (2.47 at theta 2), the pipelined step 1.07 -> 1.01-1.04 on the same box (1.04-1.05 at theta 2).
static AsgParams g_params = {5.0, 8e-3, 1e-6, 0.02, 4000, 10, 10, 800000, 1, 64, 0.0, 0, 96, 512, 1, 2, 16, 4, 2.5, 512};   // (10 epsilon = 0 rounds: measured 2.35 ms per C3 solve against 2.56 with 15 and 2.46 with 8 once the forest phases ran in the list solver)
static AsgParams asg_params_snapshot() { std::lock_guard<std::mutex> lk(g_params_mu); return g_params; }

extern "C" void cfm_assign_set_params(double theta, double eps0_frac, double eps_last_frac,
                                      double stop_frac, int round_cap, int arr_cap, int chunk) {
    std::lock_guard<std::mutex> lk(g_params_mu);
    if (theta > 1.0) { g_params.theta = theta; g_params.async_theta = theta; }      // (an explicit factor applies to both forms of the auction)
    if (eps0_frac > 0) g_params.eps0_frac = eps0_frac;
    if (eps_last_frac > 0) g_params.eps_last_frac = eps_last_frac;
    if (stop_frac >= 0) g_params.stop_frac = stop_frac;
    if (round_cap > 0) g_params.round_cap = round_cap;
    if (arr_cap >= 0) g_params.arr_cap = arr_cap;
    if (chunk > 0) g_params.chunk = chunk;
}
extern "C" void cfm_assign_set_mode(int sparse) { std::lock_guard<std::mutex> lk(g_params_mu); g_params.sparse = sparse ? 1 : 0; }
// Upper bound on the workgroups of the wide kernels (0 = none): with several couplings in flight on
// different streams a smaller grid lets their kernels run side by side.
extern "C" void cfm_assign_set_wide_blocks(int cap) { std::lock_guard<std::mutex> lk(g_params_mu); g_params.wide_blocks_cap = cap > 0 ? cap : 0; }
extern "C" void cfm_assign_set_handoff(int handoff) {      // at most 64 free rows: one root slot each in the list solver
    std::lock_guard<std::mutex> lk(g_params_mu);
    if (handoff >= 0) g_params.handoff = handoff > 64 ? 64 : handoff;
}
extern "C" void cfm_assign_set_stop_early(double f) { std::lock_guard<std::mutex> lk(g_params_mu); if (f >= 0.0 && f < 1.0) g_params.stop_early = f; }
extern "C" void cfm_assign_set_async(int on, int blocks, int last_div) {
    std::lock_guard<std::mutex> lk(g_params_mu);
    g_params.async_auction = on < 0 ? 0 : (on > 2 ? 2 : on);        // 1: the epsilon > 0 phases; 2: the epsilon = 0 rounds too
    if (blocks >= 0) g_params.async_blocks = blocks;
    if (last_div > 0) g_params.async_last_div = last_div;      // (bits 8+: see asg_run)
}
extern "C" void cfm_assign_set_async_min_n(int n) { std::lock_guard<std::mutex> lk(g_params_mu); if (n >= 64) g_params.async_min_n = n; }
extern "C" void cfm_assign_get_async(int* out3) {
    std::lock_guard<std::mutex> lk(g_params_mu);
    out3[0] = g_params.async_auction; out3[1] = g_params.async_blocks; out3[2] = g_params.async_last_div;
}
extern "C" void cfm_assign_set_small(int on) { std::lock_guard<std::mutex> lk(g_params_mu); g_params.small = on > 0 ? on : 0; }
extern "C" void cfm_assign_set_bulk(int bulk, int min_n) {
    std::lock_guard<std::mutex> lk(g_params_mu);
    if (bulk >= 0) g_params.bulk = bulk;
    if (min_n > 0) g_params.bulk_min_n = min_n;
}

// 512 bytes at the head of the workspace.  Line 0 is read-mostly inside a launch (its first 64 bytes
// are what the host polls), line 1 holds every word that is updated with device-scope atomics inside
// a launch (read back by the last-arriving workgroup with atomic loads: never through the L1), the
// rest is bookkeeping that one launch writes and the next one reads.
struct AsgState {
    int mode, n, error, certified;
    const float* Mptr;     // kernels take the matrix from here: launch arguments depend on the workspace only
    int* out_perm; int* out_cert; double* out_cost; int* out_stats;   // caller's buffers
    int tag, rb;           // current bid tag (1..254), row bits of a key
    int round, phase, stop, arr_round, round_cap, arr_cap, sparse, handoff;   // (meta bits of a key = rb + ASG_RND_BITS)
    double eps, eps_last, theta, stop_frac;
    // ---- line 1
    alignas(128) unsigned long long arrive;   // arrivals of the current launch << 32 | their payload sum (bidders of the round)
    int nN;                    // entries appended to the next scan list
    int cert_bad;
    unsigned cmin_bits, cmax_bits;      // ordered-float min / max of the matrix
    int next_mode, pad0;                // decision of a workgroup-0 step
    unsigned long long fr_min, fr_max;  // ordered min / max of the free-column labels after a relax round
    unsigned long long minslack_ord;
    double total_cost;
    // ---- line 2
    alignas(128) int nF;
    int nFC, nS, cur;
    double dfree, cmin, cmax, stop_early;
    int st_auction_rounds, st_arr_rounds, st_free_after_arr, st_sap_batches;
    int st_sap_row_scans, st_total_row_scans, st_steps, st_dense_fallbacks;
    int st_ms_phases, st_ms_augmented, wide_blocks, st_list_bids;    // st_list_bids: bids served from a row's bid list
    long long t_prev;          // time accounting (100 MHz device clock): every control step books the
    long long t_acc[16];       // time since the previous one on the mode that launch ran in
};
static_assert(sizeof(AsgState) <= 512, "AsgState has 512 bytes at the head of the workspace");
static_assert(offsetof(AsgState, arrive) == 128 && offsetof(AsgState, nF) == 256, "AsgState layout");

// Control of the bid rounds WITHOUT an arrival: a bid round needs no result of its own launch, only the number of
// bidders of the PREVIOUS round to decide whether the epsilon phase goes on.  Every workgroup therefore takes that
// decision itself in its prologue (the count was accumulated with fire-and-forget atomics and is complete: the
// kernel boundary), from the control record the previous launch left in ctl[parity of this launch]; workgroup 0
// writes the next record into ctl[other parity] — nobody of this launch reads that one, however late it starts.
// The launch parity is a kernel argument (the graphs alternate it; every program has an even number of asg_step
// launches).  Saves the two dependent device-scope atomics + the control step at the end of each of ~100 rounds.
struct AucCtl {
    double eps;
    int mode, tag, round, phase, stop, arr_round;
    int r;                                   // index of this bid launch (bidder counts: slot r & 3)
    int auction_rounds, arr_rounds, row_scans;
    int pad[2];
};
struct AsgAuc {
    AucCtl ctl[2];
    alignas(128) int bidcnt[4];
    // asynchronous phase A (asg_auction): the phase word every workgroup polls (low 16 bits: index of the current epsilon
    // phase, bit 31: the phases are over), the hand-over flag, and the bid totals the workgroups add when they leave
    alignas(128) int async_word;
    int async_done, async_scans, async_list;
    alignas(128) int async_claim;      // row groups handed out so far (asg_auction: a workgroup works on the groups it has claimed)
    int async_adopted;                 // ... of them adopted by a workgroup that already held one (statistics)
};
static_assert(sizeof(AsgAuc) <= 512, "AsgAuc has 512 bytes of the workspace");

// Tuning aid, not part of the ABI: microseconds the last solve on `ws` spent in each mode (slots
// 0-15, index = MODE_*; launch + gap to the next launch).  Slot 16: the bids of the solve that were served from the
// row's bid list (512 bytes) instead of a row scan — they are part of stats[5].  Slot 17: row groups of the one-launch auction
// adopted by a workgroup that already held one; slot 18: claims made.  Slots 19-31 are zero.  Blocking.
extern "C" int cfm_assign_debug_times(const void* ws, double* us32) {
    if (!ws || !us32) return CFM_EINVAL;
    AsgState h;
    int rc = cfm_hip(hipMemcpy(&h, ws, sizeof(h), hipMemcpyDeviceToHost));
    if (rc) return rc;
    for (int q = 0; q < 16; ++q) { us32[q] = (double)h.t_acc[q] * 0.01; us32[16 + q] = 0.0; }
    us32[16] = (double)h.st_list_bids;
    AsgAuc a;                                             // (512 + 2048 bytes into the carving: asg_carve)
    rc = cfm_hip(hipMemcpy(&a, (const char*)ws + 512 + 2048, sizeof(a), hipMemcpyDeviceToHost));
    if (rc) return rc;
    us32[17] = (double)a.async_adopted;                   // row groups of the auction adopted by a workgroup that already held one
    us32[18] = (double)a.async_claim;                     // claims made (>= the groups: late workgroups claim past the end and leave)
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
    unsigned long long* arrive_sub;   // 16 first-level arrival words, one per 128-byte line
    AsgAuc* auc;                      // control records of the bid rounds
    double* p;        // prices (= -v), phase C on
    double* bidval;   // u_i (row minima)
    double* dist;     // SAP labels
    unsigned long long* key;   // per object: ordered fp64 price, bidder's row in the low bits (phases A / B);
                               // per-tree scratch of a multi-source phase afterwards
    int* a;           // row -> col (or -1)
    int* owner;       // col -> row (or -1)
    int* bidcol;      // row -> (tag << 24) | the object it bid for last
    int* listA;       // row -> tree index during a multi-source phase
    int* listF;       // free rows
    int* listFC;      // free columns
    int* pred;
    int* tcol;        // per tree: accepted free column of the phase (or MS_NONE)
    int* grp_ticket;  // [n] asg_auction: per workgroup, (phase + 1) << 16 | unmatched rows at its last look
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

// Batched solves: problem b of a batch lives in the same carving, `stride` bytes further on.  The kernels receive the
// carving of problem 0 and the stride; blockIdx.y is the problem (uniform: the adds are scalar).
__device__ __forceinline__ AsgWs asg_shift(const AsgWs& w0, size_t off) {
    if (off == 0) return w0;
    AsgWs w;
#define ASG_SH(f) w.f = reinterpret_cast<decltype(w.f)>(reinterpret_cast<char*>(w0.f) + off)
    ASG_SH(st); ASG_SH(arrive_sub); ASG_SH(auc); ASG_SH(p); ASG_SH(bidval); ASG_SH(dist); ASG_SH(key);
    ASG_SH(a); ASG_SH(owner); ASG_SH(bidcol); ASG_SH(listA); ASG_SH(listF); ASG_SH(listFC); ASG_SH(pred); ASG_SH(tcol);
    ASG_SH(grp_ticket); ASG_SH(part_d); ASG_SH(part_i); ASG_SH(part_r); ASG_SH(cT);
    ASG_SH(S[0].col); ASG_SH(S[0].row); ASG_SH(S[0].base); ASG_SH(S[0].rj); ASG_SH(S[0].root);
    ASG_SH(S[1].col); ASG_SH(S[1].row); ASG_SH(S[1].base); ASG_SH(S[1].rj); ASG_SH(S[1].root);
    w.cl = w0.cl ? reinterpret_cast<uint2*>(reinterpret_cast<char*>(w0.cl) + off) : nullptr;
#undef ASG_SH
    return w;
}

#define MS_YMAX 4
#define MS_SPLIT_MIN 256
// MS_YMAX: a big relax round is split over this many workgroups per column group
// MS_SPLIT_MIN: ... when it has more than this many entries

static inline size_t asg_ws_bytes(int n) {
    size_t N = ((size_t)n + 3) & ~(size_t)3;     // every array starts 16-byte aligned
    size_t lists = (n <= 4096) ? N * 64 * 8 + 8 * N : 0;
    return 512 + 2048 + 512 + 8 * N * (4 + 4) + 4 * N * (10 + 6) + 64 + 16 + 16 * N * MS_YMAX + lists + 256;
}

static inline AsgWs asg_carve(void* ws, int n) {
    AsgWs w; char* q = (char*)ws; size_t N = ((size_t)n + 3) & ~(size_t)3;
    w.st = (AsgState*)q; q += 512;
    w.arrive_sub = (unsigned long long*)q; q += 2048;
    w.auc = (AsgAuc*)q; q += 512;
    w.p = (double*)q; q += 8 * N;
    w.bidval = (double*)q; q += 8 * N;
    w.dist = (double*)q; q += 8 * N;
    w.key = (unsigned long long*)q; q += 8 * N;
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
    q += 4 * N + 64;   // (formerly the staged result)
    q = (char*)(((uintptr_t)q + 15) & ~(uintptr_t)15);
    w.part_d = (double*)q; q += 8 * N * MS_YMAX;
    w.part_i = (int*)q; q += 4 * N * MS_YMAX;
    w.part_r = (int*)q; q += 4 * N * MS_YMAX;
    w.cT = (double*)q; w.cl = nullptr;
    if (n <= 4096) { q += 8 * N; w.cl = (uint2*)q; q += 8 * N * 64; }
    return w;
}

extern "C" size_t cfm_asg_ws_bytes_internal(int n) { return asg_ws_bytes(n); }

// Tuning aid (library built with -DSP_PROFILE only): the solver's cycle counters of the last solve on
// `ws` — [0..4] init / fast batches / collect / a-posteriori test / augmentation, [5] dense batches,
// [6] fast batches run, [7] batches, [9..14] stages of a fast batch, [15] pending-list entries seen.
extern "C" int cfm_assign_debug_solver(const void* ws, int n, long long* out16) {
    if (!ws || !out16 || n < 2) return CFM_EINVAL;
    AsgWs w = asg_carve(const_cast<void*>(ws), n);
    return cfm_hip(hipMemcpy(out16, w.part_d, 16 * sizeof(long long), hipMemcpyDeviceToHost));
}

// The cost matrix pointer comes out of the state block, so the compiler cannot prove it global and
// would emit FLAT loads (which also count on the LDS counter and stall the LDS price reads of a bid):
// the kernels cast it to the global address space once.
typedef const __attribute__((address_space(1))) float* gfp;
typedef float asg_v4f __attribute__((ext_vector_type(4)));
#define ASG_GLOBAL(ptr) ((gfp)(ptr))
__device__ __forceinline__ float4 asg_ld4(gfp p) {     // 16-byte global load
    const asg_v4f v = *reinterpret_cast<const __attribute__((address_space(1))) asg_v4f*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// ordered bits for floats (total order)
__device__ __forceinline__ unsigned f2ord(float x) {
    unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned k) {
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// ------------------------------------------------- device-scope words (line 1) -----
#define ASG_AGENT __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ int asg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, ASG_AGENT); }
__device__ __forceinline__ unsigned asg_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, ASG_AGENT); }
__device__ __forceinline__ unsigned long long asg_ld(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, ASG_AGENT); }
__device__ __forceinline__ void asg_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, ASG_AGENT); }
__device__ __forceinline__ void asg_st(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, ASG_AGENT); }

// Every workgroup of a launch calls this once, after its share of the step: returns true (in all
// its threads) in the workgroup that arrives last.  `wait`: the step published results through
// device-scope atomics that the control step reads — every wave then first waits until its own
// atomics have been performed, so they are complete when the last ticket is drawn.  (Bid rounds
// need no wait: their atomics are read by the next launch, and the bidder count travels in the
// arrival word itself.)  Arrivals on ONE device-scope word serialise at ~11 ns each (3 us for 256
// workgroups, measured): the arrival is two-level, 16 first-level words in separate cache lines
// (blockIdx % 16), whose last arrivers forward the group's payload sum to the top word.
#define ASG_ARRIVE_GROUPS 16
__device__ __forceinline__ bool asg_arrive_last(AsgState* st, unsigned long long* sub, int* sh_flag,
                                                unsigned payload, bool wait) {
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned G = gridDim.x < ASG_ARRIVE_GROUPS ? gridDim.x : ASG_ARRIVE_GROUPS;
        const unsigned g = blockIdx.x % G;
        const unsigned gsize = (gridDim.x - g + G - 1) / G;        // workgroups with blockIdx % G == g
        unsigned long long* c = sub + 16 * g;
        const unsigned long long t = __hip_atomic_fetch_add(c, (1ull << 32) | (unsigned long long)payload,
                                                            __ATOMIC_RELAXED, ASG_AGENT);
        int last = 0; unsigned total = 0;
        if ((unsigned)(t >> 32) == gsize - 1u) {
            asg_st(c, 0ull);
            const unsigned gp = (unsigned)t + payload;
            const unsigned long long t2 = __hip_atomic_fetch_add(&st->arrive, (1ull << 32) | (unsigned long long)gp,
                                                                 __ATOMIC_RELAXED, ASG_AGENT);
            if ((unsigned)(t2 >> 32) == G - 1u) { last = 1; asg_st(&st->arrive, 0ull); total = (unsigned)t2 + gp; }
        }
        sh_flag[0] = last; sh_flag[1] = (int)total;
    }
    __syncthreads();
    return sh_flag[0] != 0;
}

// thread 0 of the deciding workgroup: book the time since the previous control step on `mode`
__device__ __forceinline__ void asg_book(AsgState* st, int mode) {
    st->st_steps++;
    const long long now = wall_clock64();
    if (st->t_prev) st->t_acc[mode & 15] += now - st->t_prev;
    st->t_prev = now;
}

__device__ __forceinline__ void asg_enter_cert(AsgState* st) {   // one thread
    asg_st(&st->minslack_ord, ~0ull);
    asg_st(reinterpret_cast<unsigned long long*>(&st->total_cost), 0ull);
    asg_st(&st->cert_bad, 0);
}

// ------------------------------------------------------------------ keys -----
// key = (d2ord(price) & ~meta_mask) | round << rb | row.  The PRICE of an object is the key with the
// meta bits cleared: a grid of 2^-(52 - mb) relative resolution (1.2e-10 at n = 4096).  A bid is
// rounded DOWN onto the grid, so it never exceeds what the bidder computed (its new object stays
// its minimum: exact complementary slackness with the grid prices).  Among bids at one price the
// later round wins, then the higher row: a zero-increment bid of an epsilon = 0 round (a row that
// is indifferent between its best two objects) still takes the object, as in the classical
// augmenting row reduction, and simultaneous bids have one winner.
#define ASG_RND_BITS 6
__device__ __forceinline__ unsigned long long asg_enc(double v, unsigned meta, int mb) {
    return (d2ord(v) & ~((1ull << mb) - 1ull)) | (unsigned long long)meta;
}
__device__ __forceinline__ double asg_price(unsigned long long key, int mb) {
    return ord2d(key & ~((1ull << mb) - 1ull));
}
__device__ __forceinline__ int asg_key_row(unsigned long long key, int rb) {
    return (int)(key & ((1ull << rb) - 1ull));
}

// --------------------------------------------------------- wide: auction -----
#define WT 1024   // threads of the wide kernels (16 waves)
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

// A row is matched iff it bid in the current tag and the object it bid for still carries its id.
__device__ __forceinline__ bool bid_matched(const AsgWs& w, const int* r_lds, bool use_lds, int bc, int tag,
                                            int i, int rb) {
    if (((unsigned)bc >> 24) != (unsigned)tag) return false;
    const int j = bc & 0xffffff;
    return (use_lds ? r_lds[j] : asg_key_row(w.key[j], rb)) == i;
}

// wave top-2 -> bid (lane 0): one atomicMax, no award step.  The bid p_j + (second - best) + eps is
// formed from the price the top-2 was computed with (the LDS snapshot of the round); without a
// snapshot (n > WIDE_PLDS_MAX) the price may have moved since, and the bid is formed from the
// cost instead: second - c_ij + eps.
__device__ __forceinline__ void bid_commit(gfp M, const AsgWs& w, Top2 best, int i, int n, double eps,
                                           const double* p_lds, bool use_lds, int tag, int rb, int rnd) {
    const double bmin = asg_wave_min_d(best.b);
    const int jwin = asg_wave_min_i(best.b == bmin ? best.j : 0x7fffffff);
    const double rest = (best.b == bmin && best.j == jwin) ? best.s : best.b;
    const double smin = asg_wave_min_d(rest);       // the best of everything but (bmin, jwin)
    if ((threadIdx.x & 63) == 0 && jwin != 0x7fffffff) {
        double bv;
        if (use_lds) bv = p_lds[jwin] + ((smin - bmin) + eps);          // increment >= eps >= 0
        else bv = (smin - (double)M[(size_t)i * n + jwin]) + eps;
        if (bv < INFINITY && bv > -INFINITY)
            atomicMax(&w.key[jwin], asg_enc(bv, ((unsigned)rnd << rb) | (unsigned)i, rb + ASG_RND_BITS));
        w.bidcol[i] = (tag << 24) | jwin;
    }
}

#define ASG_BL 64     // entries of a bid list = SP_K: the arrays are the list solver's
// Bid lists (n <= SP_NMAX; they live in the candidate-list arrays the list solver fills AFTER the rounds): a full row
// scan leaves, per lane, the best column of the lane's share of the row and its cost, and T = the smallest second-best
// of any lane — a lower bound of c + p for every column outside the list, and it stays one: prices only rise within a
// solve.  A later bid of the row reads the 64 entries (512 bytes instead of the 4 n of the row): with b <= T the list's
// best is the row's best, and min(second in the list, T) is a lower bound of its second best — exact when the second is
// <= T, otherwise a smaller (still valid: any increment in [eps, second - best + eps] keeps eps-complementary
// slackness) bid.  b > T: the row is scanned, which refreshes its list.  MODE_UMIN0 sets T = -inf (stale lists of the
// previous solve).  Measured at C3 (a round-3 host prototype): 47.5 k bids, 11.7 k of them full scans.
__device__ __forceinline__ bool bid_from_list(gfp M, const AsgWs& w, const double* p_lds, bool use_lds, uint2 e, double T,
                                              int i, int n, double eps, int tag, int rb, int rnd) {
    if (!(T > -INFINITY)) return false;                  // (uniform: one T per row)
    const bool okc = e.x < (unsigned)n;
    const int mb = rb + ASG_RND_BITS;
    const double pj = okc ? (use_lds ? p_lds[e.x] : asg_price(w.key[e.x], mb)) : 0.0;
    Top2 lb;
    lb.b = okc ? (double)__uint_as_float(e.y) + pj : INFINITY;
    lb.s = T; lb.j = okc ? (int)e.x : 0x7fffffff;
#define ASG_BL_PARTIAL 0
    const double bmin = asg_wave_min_d(lb.b);
    if (!(bmin <= T)) return false;
    if (!ASG_BL_PARTIAL) {
        // exact bids only: the second best of the list must be inside the bound too (min over the lanes but one winner)
        const int jw = asg_wave_min_i(lb.b == bmin ? lb.j : 0x7fffffff);
        const double s2 = asg_wave_min_d((lb.b == bmin && lb.j == jw) ? INFINITY : lb.b);
        if (!(s2 <= T)) return false;
    }
    bid_commit(M, w, lb, i, n, eps, p_lds, use_lds, tag, rb, rnd);
    return true;
}

// the list a full scan leaves behind (after the bid went out: off the round's critical path)
__device__ __forceinline__ void bid_list_store(gfp M, const AsgWs& w, const Top2& best, int i, int n) {
    const double T = asg_wave_min_d(best.s);
    const bool okc = best.j != 0x7fffffff;
    const float c = okc ? M[(size_t)i * n + best.j] : 0.f;
    // (counters: the 2 MiB of lists of a problem cost ~1.7 MiB of HBM write-back per bid round however few rows are
    //  re-stored, with plain, non-temporal and scoped stores alike — the dirty state is tracked in granules of several
    //  KiB; the same holds for the 32 KiB of keys.  0.3 us per round at the write rate of the chip.)
    w.cl[(size_t)i * ASG_BL + (threadIdx.x & 63)] = make_uint2(okc ? (unsigned)best.j : 0xffffffffu, __float_as_uint(c));
    if ((threadIdx.x & 63) == 0) w.cT[i] = T;
}

#define ASG_BL_ON 1
// The bid of ONE unmatched row by one wave: from its bid list if that decides it (e / T: the list entry of this lane
// and the bound), else a scan of the row, which refreshes the list.  Returns 1 (a bid) + 0x10000 if the list served it.
__device__ __forceinline__ int bid_row(gfp M, const AsgWs& w, const double* p_lds, int i, uint2 e, double T, bool stage_p,
                                       int n, double eps, int tag, int rb, int rnd) {
    // (an OPAQUE copy of the lane: every per-lane address of a row's bid — keys, prices, the row itself — is then
    //  formed inside the row loop.  Hoisted to the kernel's top as loop invariants they pushed asg_step one value past
    //  its 128 registers: 12 bytes of scratch and a reload on the round's critical path in round 3.)
    int lane = threadIdx.x & 63; asm volatile("" : "+v"(lane));
    const bool vec = ((n & 3) == 0);
    const bool fast = stage_p && (n & 4095) == 0;
    const bool lists = ASG_BL_ON && (w.cl != nullptr);
    const int mb = rb + ASG_RND_BITS;
    if (lists && bid_from_list(M, w, p_lds, stage_p, e, T, i, n, eps, tag, rb, rnd)) return 0x10001;
    gfp row = M + (size_t)i * n;
    Top2 best; best.b = INFINITY; best.s = INFINITY; best.j = 0x7fffffff;
    if (fast) {
        for (int seg = 0; seg < n; seg += 4096) {
            // the segment's 16 KB in flight at once: 16 float4 per lane
            float4 c[16];
            gfp rs = row + seg + lane * 4;
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = asg_ld4(rs + 256 * k);
            bid_segment(best, c, p_lds + seg + lane * 4, seg + lane * 4);
            __builtin_amdgcn_sched_barrier(0);     // the next segment's loads stay behind this one's arithmetic
        }
    } else if (vec && stage_p) {
        for (int j0 = lane * 4; j0 < n; j0 += 1024) {
            // 4 float4 in flight per lane per trip
            float4 c4[4]; double2 pa[4], pb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 256 * k;
                if (j < n) {
                    c4[k] = asg_ld4(row + j);
                    pa[k] = *reinterpret_cast<const double2*>(p_lds + j);
                    pb[k] = *reinterpret_cast<const double2*>(p_lds + j + 2);
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
    } else if (stage_p) {
        for (int j = lane; j < n; j += 64) top2_push(best, (double)row[j] + p_lds[j], j);
    } else if (vec) {
        // no price snapshot in LDS (n > WIDE_PLDS_MAX): the keys come with the costs, 4 + 4 per request
        for (int j0 = lane * 4; j0 < n; j0 += 1024) {
            float4 c4[4]; ulonglong2 ka[4], kb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 256 * k;
                if (j < n) {
                    c4[k] = asg_ld4(row + j);
                    ka[k] = *reinterpret_cast<const ulonglong2*>(w.key + j);
                    kb[k] = *reinterpret_cast<const ulonglong2*>(w.key + j + 2);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 256 * k;
                if (j < n) {
                    top2_push(best, (double)c4[k].x + asg_price(ka[k].x, mb), j + 0);
                    top2_push(best, (double)c4[k].y + asg_price(ka[k].y, mb), j + 1);
                    top2_push(best, (double)c4[k].z + asg_price(kb[k].x, mb), j + 2);
                    top2_push(best, (double)c4[k].w + asg_price(kb[k].y, mb), j + 3);
                }
            }
        }
    } else {
        for (int j = lane; j < n; j += 64) top2_push(best, (double)row[j] + asg_price(w.key[j], mb), j);
    }
    bid_commit(M, w, best, i, n, eps, p_lds, stage_p, tag, rb, rnd);
    if (lists) bid_list_store(M, w, best, i, n);
    return 1;
}

// One wave per row; a row bids iff it is unmatched.  pre_bc4 = bidcol of the wave's first four rows, pre_e / pre_T the
// bid list of its first row, requested with the keys in the kernel prologue.  Returns the number of bids of this wave
// (low 16 bits) and how many of them the lists served (high bits).
__device__ __forceinline__ int wide_bid(gfp M, const AsgWs& w, const double* p_lds, const int* r_lds,
                                        int wave_gid, int n_waves, int4 pre_bc4, uint2 pre_e, double pre_T, bool stage_p, int n,
                                        double eps, int tag, int rb, int rnd) {
    const int lane = threadIdx.x & 63;
    const bool lists = ASG_BL_ON && (w.cl != nullptr);
    int nbids = 0;
    int i = wave_gid;
    for (int k = 0; i < n; i += n_waves, ++k) {
        const int bc = k == 0 ? pre_bc4.x : k == 1 ? pre_bc4.y : k == 2 ? pre_bc4.z : k == 3 ? pre_bc4.w : w.bidcol[i];
        if (bid_matched(w, r_lds, stage_p, bc, tag, i, rb)) continue;
        uint2 e = pre_e; double T = pre_T;
        if (lists && k > 0) { e = w.cl[(size_t)i * ASG_BL + lane]; T = w.cT[i]; }
        nbids += bid_row(M, w, p_lds, i, e, T, stage_p, n, eps, tag, rb, rnd);
    }
    return nbids;
}

// The same round when the grid gives a workgroup MORE rows than it has waves (batches of problems share the chip;
// n > 8192): a wave that walked its rows one after the other would pay a dependent round trip per row, though only a
// few per cent of the rows bid in a typical round.  Thread t checks the workgroup's row t (its last bid arrived with the
// prologue, the owner rows are in LDS), the unmatched rows go into a queue in LDS, and the waves take them from there.
#define ASG_BQ 256                    // rows per workgroup this path takes (the grid is chosen accordingly)
// grp / G: the row group this call bids for — rows grp, grp + G, ... (asg_step: its own blockIdx.x of gridDim.x; asg_auction: a
// group it has claimed, see there)
__device__ __forceinline__ int wide_bid_queue(gfp M, const AsgWs& w, const double* p_lds, const int* r_lds, int* bq, int* bq_cnt,
                                              int my_bc, bool stage_p, int n, double eps, int tag, int rb, int rnd, int grp, int G) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool lists = ASG_BL_ON && (w.cl != nullptr);
    int tq = threadIdx.x; asm volatile("" : "+v"(tq));                  // (opaque: formed per call, never carried — and spilled — across the persistent loop of asg_auction)
    const int i_chk = grp + G * tq;                                     // row t of the group
    if (i_chk < n && threadIdx.x < ASG_BQ && !bid_matched(w, r_lds, stage_p, my_bc, tag, i_chk, rb))
        bq[atomicAdd(bq_cnt, 1)] = i_chk;
    __syncthreads();
    const int m = *bq_cnt;
    int nbids = 0;
    for (int k = wv; k < m; k += WT / 64) {
        const int i = bq[k];
        uint2 e = make_uint2(0xffffffffu, 0u); double T = -INFINITY;
        // (the lane's list address is formed per row from an opaque copy of the lane: hoisted out of the loop it was
        //  the one value the register allocator spilled — 8 bytes of scratch and a reload on the round's critical path)
        int ln = lane; asm volatile("" : "+v"(ln));
        if (lists) { e = w.cl[(size_t)i * ASG_BL + ln]; T = w.cT[i]; }
        nbids += bid_row(M, w, p_lds, i, e, T, stage_p, n, eps, tag, rb, rnd);
    }
    return nbids;
}

// MODE_UMIN0: row minima (bidval), the cost range (one pair of atomics per workgroup) and the reset
// of the key / bid state.  sh: >= 64 floats of LDS.
__device__ __forceinline__ void wide_umin0(gfp M, const AsgWs& w, AsgState* st,
                                           int wave_gid, int n_waves, int n, float* sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = blockIdx.x * WT + threadIdx.x; j < n; j += gridDim.x * WT) { w.key[j] = 0ull; w.bidcol[j] = -1; w.grp_ticket[j] = 0; }
    float lo = INFINITY, hi = -INFINITY;
    const bool vec = ((n & 3) == 0);
    for (int i = wave_gid; i < n; i += n_waves) {
        gfp row = M + (size_t)i * n;
        float m = INFINITY;
        if (vec) {
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                float4 c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    c[k] = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
                    if (j < n) c[k] = asg_ld4(row + j);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (j0 + 256 * k < n) {
                        m = fminf(fminf(m, c[k].x), fminf(c[k].y, fminf(c[k].z, c[k].w)));
                        hi = fmaxf(fmaxf(hi, c[k].x), fmaxf(c[k].y, fmaxf(c[k].z, c[k].w)));
                    }
                }
            }
        } else {
            for (int j = lane; j < n; j += 64) { const float c = row[j]; m = fminf(m, c); hi = fmaxf(hi, c); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
        if (lane == 0) { w.bidval[i] = (double)m; if (w.cl != nullptr) w.cT[i] = -INFINITY; }     // (no bid list yet)
        lo = fminf(lo, m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if (lane == 0) { sh[wv] = lo; sh[16 + wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < WT / 64; ++q) { lo = fminf(lo, sh[q]); hi = fmaxf(hi, sh[16 + q]); }
        if (lo <= hi) { atomicMin(&st->cmin_bits, f2ord(lo)); atomicMax(&st->cmax_bits, f2ord(hi)); }
    }
}

// MODE_INITRED: initial prices by row + column reduction (the classical Jonker-Volgenant start): with
// u_i = min_j c_ij (bidval[], MODE_UMIN0) the price p_j = max_i (u_i - c_ij) <= 0 is the largest
// one that keeps every row's minimum where it is, and it makes every column tight for some row.
// The auction then starts two epsilon phases later (eps0 = 8e-3 instead of 0.2 of the cost range).
// Lane <-> column, the grid splits the rows; the partial maxima go straight into the keys (the
// encoding is monotone, so the max of the encoded values is the encoded max; row bits = none).
__device__ __forceinline__ void wide_initred(gfp M, const AsgWs& w, double* sh_d, int n, int rb) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NW = WT / 64, Q = 8;
    const unsigned none = (1u << rb) - 1u;
    if ((n & 255) == 0 && n >= 3072 && n <= WIDE_PLDS_MAX) {
        // 16-byte form (round 6): a lane owns FOUR adjacent columns, a wave reads 1 KiB of a row per request instead of 256 B
        // (C3: 31.8 -> 19.8 us, batch of four 131 -> 52 us; profiles/r6_experiments.txt 16).  The step's LDS holds the prices + owners of n >= 3072 columns
        // (>= 36 KiB): room for the 16 x 256 partial maxima.
        const int n_groups = n >> 8;
        int Y = gridDim.x / n_groups; Y = Y < 1 ? 1 : Y;
        for (int unit = blockIdx.x; unit < n_groups * Y; unit += gridDim.x) {
            const int g = unit % n_groups, y = unit / n_groups;
            const int k = g * 256 + 4 * lane;
            const int r_beg = (int)((long long)n * y / Y), r_end = (int)((long long)n * (y + 1) / Y);
            double m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
            for (int r0 = r_beg + wv * Q; r0 < r_end; r0 += NW * Q) {
                float4 c[Q]; double u[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int r = r0 + q;
                    const bool v = r < r_end;
                    c[q] = asg_ld4(M + (size_t)(v ? r : r_beg) * n + k);
                    u[q] = v ? w.bidval[r] : -INFINITY;
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    m0 = fmax(m0, u[q] - (double)c[q].x); m1 = fmax(m1, u[q] - (double)c[q].y);
                    m2 = fmax(m2, u[q] - (double)c[q].z); m3 = fmax(m3, u[q] - (double)c[q].w);
                }
            }
            double* mine = sh_d + (size_t)(wv * 64 + lane) * 4;
            mine[0] = m0; mine[1] = m1; mine[2] = m2; mine[3] = m3;
            __syncthreads();
            if (wv < 4) {                                   // wave e reduces column 4 lane + e of the group over the 16 waves
                double m = -INFINITY;
#pragma unroll
                for (int q = 0; q < NW; ++q) m = fmax(m, sh_d[(size_t)(q * 64 + lane) * 4 + wv]);
                if (m > -INFINITY && m < INFINITY) atomicMax(&w.key[k + wv], asg_enc(m, none, rb + ASG_RND_BITS));
            }
            __syncthreads();
        }
        return;
    }
    const int n_groups = (n + 63) / 64;
    int Y = gridDim.x / n_groups; Y = Y < 1 ? 1 : Y;
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
            if (m > -INFINITY && m < INFINITY) atomicMax(&w.key[k], asg_enc(m, none, rb + ASG_RND_BITS));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ wide: SAP ------
// Row minima u_i = min_k (c_ik + p_k), one wave per row: for every row (before the column
// reduction) or for the free rows of the next multi-source phase.  Result in bidval[row]; for the
// roots the wave also files the root's entry of the first scan list (the start of the phase).
#define MS_NONE 0x7fffffff
__device__ __forceinline__ void wide_umin(gfp M, const AsgWs& w, const AsgState* st,
                          int wave_gid, int n_waves, bool roots_only, const int n) {
    const int cnt = roots_only ? st->nF : n;
    const int lane = threadIdx.x & 63;
    const bool vec = ((n & 3) == 0);
    const SList L = slist(w, roots_only ? st->cur : 0);
    for (int t = wave_gid; t < cnt; t += n_waves) {
        const int i = roots_only ? w.listF[t] : t;
        gfp row = M + (size_t)i * n;
        double m = INFINITY;
        if (vec) {
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                float4 c[4]; double2 pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        c[k] = asg_ld4(row + j);
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
        if (lane == 0) {
            w.bidval[i] = m;
            if (roots_only) {
                L.col[t] = -1; L.row[t] = i; L.base[t] = 0.0; L.rj[t] = m; L.root[t] = t;
                w.listA[i] = t;                 // row -> tree index
                w.tcol[t] = MS_NONE; w.key[t] = ~0ull;
            }
        }
    }
}

// the rest of a multi-source phase start: labels unset (all threads of the grid)
__device__ __forceinline__ void wide_ms_reset(const AsgWs& w, int n) {
    for (int k = blockIdx.x * WT + threadIdx.x; k < n; k += gridDim.x * WT) { w.dist[k] = INFINITY; w.pred[k] = -1; }
}

// Column reduction of the free columns: p_k <- max_i (u_i - c_ik), the largest price at which
// column k is still not cheaper than any row's current minimum.  No u_i changes, every matched
// edge stays tight, the dual objective rises by the price drop.  One workgroup per column
// (strided reads: 64 B sector per row, only nFC columns).
__device__ __forceinline__ void wide_colred(gfp M, const AsgWs& w, const AsgState* st,
                            double* sh_d, const int n) {
    const int nFC = st->nFC;
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

// Close a column group's round (wave 0, lane <-> column k): take the merged minimum, append an
// improved assigned column to the NEXT list, and contribute the labels of the group's free columns
// to the round's radius words (ordered min / max, one pair of atomics per group).
__device__ __forceinline__ void relax_close(const AsgWs& w, AsgState* st, const SList& L, const SList& Nx,
                                            int k, bool ok, double dfree, double best, int bi, int bt,
                                            double dk, int ow, double rjk) {
    // dk / ow / rjk (label, owner, owner's matched-edge value of column k) were requested at the start of
    // the group, so the close is: compare, store, one append atomic, the entry
    if (ok && best < dk) {
        w.dist[k] = best; w.pred[k] = bi; dk = best;
        if (ow >= 0 && best < dfree) {
            const int idx = atomicAdd(&st->nN, 1);
            Nx.col[idx] = k; Nx.row[idx] = ow; Nx.base[idx] = best; Nx.root[idx] = L.root[bt];   // the winner's tree
            Nx.rj[idx] = rjk;
        }
    }
    const bool fr = ok && ow < 0;
    const double mx = wave_max_d(fr ? dk : -INFINITY), mn = wave_min_d(fr ? dk : INFINITY);
    if ((threadIdx.x & 63) == 0 && mn < INFINITY) atomicMin(&st->fr_min, d2ord(mn));      // a free column with a finite label
    if ((threadIdx.x & 63) == 0 && mx > -INFINITY) atomicMax(&st->fr_max, d2ord(mx));     // infinite labels count for the max
}

// Relax every listed row.  A workgroup owns 16 columns; a wave covers 4 list entries x 16 columns per
// load (64 B of each row), 6 such loads in flight per lane, so one trip of the workgroup's 16 waves
// relaxes 384 entries and a round is one or two trips deep whatever its size.  The four entry
// sub-groups of a wave are merged with two lane exchanges, the waves through LDS, and lane <-> column
// of wave 0 is the single writer of the column (dist / pred / tree stay consistent without atomics
// or fences); it appends an improved assigned column to the NEXT list (one atomic per append).
// Ties go to the lowest row id, so the result does not depend on how the list is split.
__device__ __forceinline__ void wide_relax(gfp M, const AsgWs& w, AsgState* st,
                           double* sh_d, int* sh_i, int* sh_r, const int n) {
    const int nS = st->nS, cur = st->cur;
    const double dfree = st->dfree;
    const SList L = slist(w, cur), Nx = slist(w, cur ^ 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 4, cl = lane & 15;
    const int n_groups = (n + 15) / 16;
    constexpr int Q = 6, NW = WT / 64;     // (8 in flight would need 4 VGPRs more than the 128 of a 16-wave workgroup)
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int k = g * 16 + cl;
        const bool ok = k < n;
        const double pk = ok ? w.p[k] : 0.0;
        // what the close of the column needs, requested now (wave 0 only uses it)
        double dk0 = INFINITY; int ow0 = 0;
        if (wv == 0 && sub == 0 && ok) { dk0 = w.dist[k]; ow0 = w.owner[k]; }
        double best = INFINITY; int bi = 0x7fffffff, br = 0;     // br: list index of the best entry (its tree is looked up at the end)
        for (int t0 = wv * (4 * Q); t0 < nS; t0 += NW * 4 * Q) {
            int ri[Q]; double bs[Q], rj[Q]; float c[Q];
            // one base address per array and immediate offsets: entries past nS are read (the lists are
            // followed by other workspace arrays) and masked
            const int* prow = L.row + t0 + sub;
            const double* pbase = L.base + t0 + sub; const double* prj = L.rj + t0 + sub;
#pragma unroll
            for (int q = 0; q < Q; ++q) { ri[q] = prow[q * 4]; bs[q] = pbase[q * 4]; rj[q] = prj[q * 4]; }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool v = (t0 + q * 4 + sub) < nS;
                bs[q] = v ? bs[q] : INFINITY;
                ri[q] = v ? ri[q] : 0;
            }
#pragma unroll
            for (int q = 0; q < Q; ++q)
                c[q] = (ok && bs[q] < dfree) ? M[(size_t)ri[q] * n + k] : 0.f;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                // (the entry's own column needs no exclusion: its candidate is exactly its label, never below it)
                if (ok && bs[q] < dfree) {
                    double rc = ((double)c[q] + pk) - rj[q];
                    rc = fmax(rc, 0.0);                   // dual feasible up to rounding
                    const double cand = bs[q] + rc;
                    if (cand < best || (cand == best && ri[q] < bi)) { best = cand; bi = ri[q]; br = t0 + q * 4 + sub; }
                }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const double c2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64), r2 = __shfl_xor(br, o, 64);
            if (c2 < best || (c2 == best && i2 < bi)) { best = c2; bi = i2; br = r2; }
        }
        if (sub == 0) { sh_d[wv * 16 + cl] = best; sh_i[wv * 16 + cl] = bi; sh_r[wv * 16 + cl] = br; }
        double rjk = 0.0;
        if (wv == 0 && sub == 0 && ok && ow0 >= 0) rjk = (double)M[(size_t)ow0 * n + k] + pk;    // in flight across the barrier
        __syncthreads();
        if (wv == 0) {
            const bool okc = ok && sub == 0;
            // lane (sub, cl) merges the results of waves 4 sub .. 4 sub + 3, then two lane exchanges
            best = INFINITY; bi = 0x7fffffff; br = 0;
#pragma unroll
            for (int q = 0; q < NW / 4; ++q) {
                const int s2 = (sub * (NW / 4) + q) * 16 + cl;
                const double c2 = sh_d[s2]; const int i2 = sh_i[s2];
                if (c2 < best || (c2 == best && i2 < bi)) { best = c2; bi = i2; br = sh_r[s2]; }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const double c2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64), r2 = __shfl_xor(br, o, 64);
                if (c2 < best || (c2 == best && i2 < bi)) { best = c2; bi = i2; br = r2; }
            }
            relax_close(w, st, L, Nx, k, okc, dfree, best, bi, br, dk0, ow0, rjk);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------ wide: cert -----
// pre_a: the match of row wave_gid, loaded with the state block in the kernel prologue.
__device__ __forceinline__ void wide_cert(gfp M, const AsgWs& w, AsgState* st, int wave_gid,
                          int n_waves, int pre_a, double* sh_d, const int n) {
    const int lane = threadIdx.x & 63;
    double wmin = INFINITY, csum = 0.0; int bad = 0;
    for (int i = wave_gid; i < n; i += n_waves) {
        const int ai = (i == wave_gid) ? pre_a : w.a[i];
        if (ai < 0 || ai >= n || w.owner[ai] != i) { bad = 1; continue; }
        gfp row = M + (size_t)i * n;
        const double ui = (double)row[ai] + w.p[ai];
        double m = INFINITY;
        if ((n & 3) == 0) {
            // 4 float4 of the row and their prices in flight per lane and trip
            for (int j0 = lane * 4; j0 < n; j0 += 1024) {
                float4 c4[4]; double2 pa[4], pb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + 256 * k;
                    if (j < n) {
                        c4[k] = asg_ld4(row + j);
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
    // one set of atomics per workgroup: 4096 waves adding into the same fp64 word serialise at the L2
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
    __syncthreads();
}

#include "assign_sparse.h"
static_assert(SP_K == ASG_BL, "the bid lists use the list solver's arrays");
#include "assign_small.h"
static_assert(SP_ROOTS == 64, "the hand-off threshold (cfm_assign_set_handoff) is capped at the solver's root slots");

// ------------------------------------------------- one-workgroup helpers -----
#define CT WT
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

// MODE_CONVERT (workgroup 0): the auction state (keys, last bids) becomes prices, matches and the
// ordered lists of free rows / free columns.  Returns the number of free rows (all threads).
__device__ int ctrl_convert(const AsgWs& w, AsgState* st, int* sh) {
    const int n = st->n, tag = st->tag, rb = st->rb;
    for (int k = threadIdx.x; k < n; k += CT) { w.owner[k] = -1; w.p[k] = asg_price(w.key[k], rb + ASG_RND_BITS); }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += CT) {
        const int bc = w.bidcol[i];
        int ai = -1;
        if (((unsigned)bc >> 24) == (unsigned)tag) {
            const int j = bc & 0xffffff;
            if (j < n && asg_key_row(w.key[j], rb) == i) ai = j;
        }
        w.a[i] = ai;
        if (ai >= 0) w.owner[ai] = i;
    }
    __syncthreads();
    int baseR = 0;
    for (int i0 = 0; i0 < n; i0 += CT) {
        const int i = i0 + threadIdx.x;
        const int f = (i < n && w.a[i] < 0) ? 1 : 0;
        int tot;
        const int off = block_scan_excl(f, &tot, sh);
        if (f) w.listF[baseR + off] = i;
        baseR += tot;
    }
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
        st->nFC = base; st->nF = baseR; st->st_free_after_arr = baseR;
        st->cur = 0; asg_st(&st->nN, 0);
        if (base != baseR) st->error = 4;
    }
    __syncthreads();
    return baseR;
}

// MODE_MS_FINISH (workgroup 0).  The forest has converged below the radius.  Every free column with a
// finite label belongs to exactly one tree (walk the predecessors to its free row); each tree
// accepts its nearest free column (ties: lowest column).  With D = the largest accepted label, the
// dual update
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
            if (ti >= 0) atomicMin(&w.key[ti], d2ord(d));
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
            if (d2ord(w.dist[k]) == w.key[ti]) atomicMin(&w.tcol[ti], k);
        }
    }
    __syncthreads();
    // radius D = largest accepted label
    double lm = INFINITY; int cnt = 0;
    for (int t = threadIdx.x; t < nF; t += CT) {
        if (w.tcol[t] != MS_NONE) { lm = fmin(lm, -ord2d(w.key[t])); ++cnt; }
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
        st->nF = baseF; st->nFC = baseC;
        st->st_ms_augmented += total;
        st->st_total_row_scans += total;
        if (baseF != baseC || baseF != nF - total) st->error = 4;
    }
    __syncthreads();
    return total;
}

// ----------------------------------------------------------------- step ------
// What the bidder count of the previous round means for the next one (epsilon phases, epsilon = 0 rounds, hand-over).
// (the parameters come from the snapshot of the state block asg_step takes with its first batch of loads)
struct AucParams { int round_cap, arr_cap; double theta, eps_last, stop_frac, stop_early; };
__device__ __forceinline__ void auc_decide(AucCtl& C, const AucParams* st, int cnt, int n) {
    C.row_scans += cnt;
    const int tag_next = (C.tag % 254) + 1;
    if (C.mode == MODE_AUCTION) {
        C.auction_rounds++;
        const int round = C.round + 1;
        if (cnt <= C.stop || round >= st->round_cap) {
            const double e2 = C.eps / st->theta;
            if (e2 < st->eps_last) { C.mode = MODE_ARR; C.eps = 0.0; C.arr_round = 0; }
            else {
                C.eps = e2; C.phase++;
                const bool is_last = (e2 / st->theta) < st->eps_last;
                C.stop = (int)((is_last ? st->stop_frac : fmax(st->stop_frac, st->stop_early)) * n);
            }
            C.tag = tag_next; C.round = 0;    // every row is unassigned again, the prices stay
        } else C.round = round;
    } else {
        C.arr_rounds++;
        const int ar = C.arr_round + 1;
        C.arr_round = ar;
        if (cnt == 0 || ar >= st->arr_cap) C.mode = MODE_CONVERT;
    }
}

// control step of a launch: thread 0 of the last-arriving workgroup (MODE_CERT: the whole workgroup)
__device__ __forceinline__ void step_ctrl(const AsgWs& w, AsgState* st, int mode, int n, int payload, int par) {
    if (mode == MODE_CERT) {
        // the pass has filled minslack / total_cost: export the result to the caller's buffers
        int* perm = st->out_perm;
        for (int i = threadIdx.x; i < n; i += WT) perm[i] = w.a[i];
        if (threadIdx.x == 0) {
            asg_book(st, mode);
            const double minslack = ord2d(asg_ld(&st->minslack_ord));
            const double scale = fmax(fabs(st->cmax), fabs(st->cmin));
            const double tol = 1e-10 * fmax(scale, 1e-30);
            st->st_total_row_scans += n;
            const int ok = (!asg_ld(&st->cert_bad)) && (minslack >= -tol);
            st->certified = ok;
            const double total = __longlong_as_double((long long)asg_ld(reinterpret_cast<unsigned long long*>(&st->total_cost)));
            if (st->out_cert) *st->out_cert = ok;
            if (st->out_cost) *st->out_cost = total;
            int* stats = st->out_stats;
            if (stats) {
                stats[0] = st->st_auction_rounds; stats[1] = st->st_arr_rounds;
                stats[2] = st->st_free_after_arr; stats[3] = st->st_sap_batches;
                stats[4] = st->st_sap_row_scans; stats[5] = st->st_total_row_scans;
                stats[6] = st->st_steps;
                stats[7] = (st->phase & 0xff) | ((st->st_ms_phases & 0xff) << 8) | (st->st_dense_fallbacks << 16);
            }
            st->mode = MODE_DONE;
        }
        return;
    }
    if (threadIdx.x != 0) return;
    asg_book(st, mode);
    if (mode == MODE_SAP) {
        // New radius: one tree — the best free-column label (no label at or above it can matter);
        // several trees — the largest free-column label (infinite until every free column is
        // reached).  It only ever decreases within a phase, so an entry skipped once is never
        // needed later.  Entries were appended against the OLD radius: if none is left below the
        // new one the next round relaxes nothing, appends nothing, and the phase closes.
        const int nN = asg_ld(&st->nN);
        const unsigned long long f = (st->nF == 1) ? asg_ld(&st->fr_min) : asg_ld(&st->fr_max);
        st->st_sap_batches++; st->st_sap_row_scans += st->nS; st->st_total_row_scans += st->nS;
        st->dfree = (f == ~0ull || f == 0ull) ? INFINITY : ord2d(f);
        asg_st(&st->fr_min, ~0ull); asg_st(&st->fr_max, 0ull);
        st->cur ^= 1; st->nS = nN; asg_st(&st->nN, 0);
        if (nN == 0) st->mode = MODE_MS_FINISH;
    } else if (mode == MODE_UMIN0) {
        st->cmin = (double)ord2f(asg_ld(&st->cmin_bits));
        st->cmax = (double)ord2f(asg_ld(&st->cmax_bits));
        double cr = st->cmax - st->cmin;
        if (!(cr > 0.0) || !(cr < INFINITY)) cr = 1.0;
        st->eps = cr * st->eps;          // eps / eps_last hold the fractions on entry
        st->eps_last = cr * st->eps_last;
        // every phase but the last is cut earlier: it only has to shape the prices
        const bool first_is_last = (st->eps / st->theta) < st->eps_last;
        st->stop = (int)((first_is_last ? st->stop_frac : fmax(st->stop_frac, st->stop_early)) * n);
        st->st_total_row_scans += n;
        st->mode = MODE_INITRED;
    } else if (mode == MODE_INITRED) {
        st->st_total_row_scans += n;
        st->round = 0; st->phase = 0; st->tag = 1;
        // first control record of the bid rounds, for the NEXT launch (the other parity)
        AucCtl C;
        C.eps = st->eps; C.mode = MODE_AUCTION; C.tag = 1; C.round = 0; C.phase = 0; C.stop = st->stop; C.arr_round = 0;
        C.r = 0; C.auction_rounds = 0; C.arr_rounds = 0; C.row_scans = 0; C.pad[0] = C.pad[1] = 0;
        w.auc->ctl[par ^ 1] = C;
        for (int q = 0; q < 4; ++q) asg_st(&w.auc->bidcnt[q], 0);
        asg_st(&w.auc->async_word, 0); w.auc->async_done = 0; w.auc->async_scans = 0; w.auc->async_list = 0;
        asg_st(&w.auc->async_claim, 0); w.auc->async_adopted = 0;
        st->mode = MODE_AUCTION;
    } else if (mode == MODE_CONVERT || mode == MODE_MS_FINISH) {
        st->mode = asg_ld(&st->next_mode);
    } else if (mode == MODE_UMIN) {
        st->st_total_row_scans += n;
        st->mode = MODE_COLRED;
    } else if (mode == MODE_COLRED) {
        st->mode = (st->sparse && st->nF <= st->handoff) ? MODE_BUILD : MODE_ROOTMIN;
    } else if (mode == MODE_ROOTMIN) {
        st->nS = st->nF; asg_st(&st->nN, 0); st->dfree = INFINITY;
        asg_st(&st->fr_min, ~0ull); asg_st(&st->fr_max, 0ull);
        st->st_ms_phases++;
        st->st_total_row_scans += st->nF;
        st->mode = MODE_SAP;
    }
}

// ONE wide kernel for every chip-wide step (modes UMIN0 .. CERT): the host replays it without knowing
// which step comes next.  LDS (modes are exclusive): bid rounds — prices [n] fp64 + owner rows [n]
// int; relax — 16 KiB of merge buffers; MS_FINISH — 2 n ints for the path walks; the rest < 8 KiB.
// the first 128 bytes of AsgState (its line 0), read as one block
struct AsgHead {
    int mode, n, error, certified;
    const float* Mptr;
    int* out_perm; int* out_cert; double* out_cost; int* out_stats;
    int tag, rb;
    int round, phase, stop, arr_round, round_cap, arr_cap, sparse, handoff;
    double eps, eps_last, theta, stop_frac;
};
static_assert(sizeof(AsgHead) == 128 && offsetof(AsgHead, rb) == offsetof(AsgState, rb) && offsetof(AsgHead, stop_frac) == offsetof(AsgState, stop_frac)
              && offsetof(AsgHead, round_cap) == offsetof(AsgState, round_cap) && offsetof(AsgHead, Mptr) == offsetof(AsgState, Mptr), "AsgHead mirrors line 0 of AsgState");
__global__ __launch_bounds__(WT) void asg_step(AsgWs w0, int n_host, int par, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char step_lds[];
    const AsgWs w = asg_shift(w0, stride * blockIdx.y);
    __shared__ int sh[32];
    __shared__ double shd[32];
    __shared__ int shi[32];
    AsgState* st = w.st;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // consecutive work items go to different workgroups (different CUs)
    const int wave_gid = wv * gridDim.x + blockIdx.x;
    const int n_waves = gridDim.x * (WT / 64);
    // Every launch starts cold: what a bid round needs first — the state block, all keys (they go to
    // LDS as prices + owner rows) and the last bid of the wave's row — is requested together.
    const bool stage_p = (n_host <= WIDE_PLDS_MAX);
    constexpr int KP = WIDE_PLDS_MAX / (2 * WT);      // key pairs per thread (4 at 1024 threads)
    ulonglong2 kst[KP];
    {
        const int nn = stage_p ? n_host : 0;
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const int j = threadIdx.x * 2 + 2 * WT * q;
            kst[q] = *reinterpret_cast<const ulonglong2*>(w.key + (j < nn ? j : 0));
        }
    }
    // (a grid smaller than one wave per row — batches, n > 8192 — gives a wave several rows: their last bids are requested
    //  together, or every further row would add a dependent round trip to the round)
    int pre_bc = (wave_gid < n_host) ? w.bidcol[wave_gid] : -1;
    int pre_bc1 = (wave_gid + n_waves < n_host) ? w.bidcol[wave_gid + n_waves] : -1;
    int pre_bc2 = (wave_gid + 2 * n_waves < n_host) ? w.bidcol[wave_gid + 2 * n_waves] : -1;
    int pre_bc3 = (wave_gid + 3 * n_waves < n_host) ? w.bidcol[wave_gid + 3 * n_waves] : -1;
    // more rows per workgroup than waves: the queue form of the round (wide_bid_queue), thread t looks at row t
    const bool queue = (n_host > (int)gridDim.x * (WT / 64)) && (n_host <= (int)gridDim.x * ASG_BQ);
    int my_bc = -1;
    if (queue && threadIdx.x < ASG_BQ && (int)(blockIdx.x + gridDim.x * threadIdx.x) < n_host)
        my_bc = w.bidcol[blockIdx.x + gridDim.x * threadIdx.x];
    uint2 pre_e = make_uint2(0xffffffffu, 0u); double pre_T = -INFINITY;       // the row's bid list (see bid_from_list)
    if (w.cl != nullptr && wave_gid < n_host) { pre_e = w.cl[(size_t)wave_gid * ASG_BL + lane]; pre_T = w.cT[wave_gid]; }
    // The round's control data — the first 128 bytes of the state block (mode, error, rb, the caps and factors auc_decide
    // looks at), the control record ctl[par & 1] and all four bidder counters — is requested with the first batch too.
    // Read where they are used, each of them is its own dependent round trip to the memory-side cache (error after mode,
    // the record after mode, the counter after the record, round_cap after the counter, ...): six in a typical round
    // where one does (round 5: lone C3 solve 2.57 -> 2.43 ms, pipelined step 1.24 -> 1.20 ms; profiles/r5_experiments.txt).
    // ctl[par & 1] and bidcnt[(r - 1) & 3] were written by the previous launch and are not touched by this one (it writes
    // ctl[(par & 1) ^ 1], zeroes slot (r + 1) & 3, adds to slot r & 3).
    const AsgHead H = *reinterpret_cast<const AsgHead*>(st);
    const double pre_stop_early = st->stop_early;
    AucCtl preC = w.auc->ctl[par & 1];
    int pre_cnt0 = asg_ld(&w.auc->bidcnt[0]), pre_cnt1 = asg_ld(&w.auc->bidcnt[1]);
    int pre_cnt2 = asg_ld(&w.auc->bidcnt[2]), pre_cnt3 = asg_ld(&w.auc->bidcnt[3]);
    asm volatile("" : "+v"(pre_cnt0), "+v"(pre_cnt1), "+v"(pre_cnt2), "+v"(pre_cnt3) :: "memory");
    int mode = H.mode;
    gfp M = ASG_GLOBAL(H.Mptr);
    const AucParams AP{H.round_cap, H.arr_cap, H.theta, H.eps_last, H.stop_frac, pre_stop_early};
    asm volatile("" : "+v"(pre_bc), "+v"(pre_bc1), "+v"(pre_bc2), "+v"(pre_bc3), "+v"(my_bc), "+v"(pre_e.x), "+v"(pre_T), "+v"(kst[0].x), "+v"(kst[KP - 1].x) : "s"(mode) : "memory");   // all of it in flight
    if (mode > MODE_CERT || H.error) return;
    const int n = n_host;
    unsigned payload = 0;
    // bid rounds: every workgroup decides for itself what this launch is (see AucCtl)
    AucCtl C;
    const bool bidding = (mode == MODE_AUCTION || mode == MODE_ARR);
    if (bidding) {
        C = preC;
        if (C.r > 0) {
            // bidders of the previous round in the low 16 bits, those served from their lists above (lists: n <= SP_NMAX)
            const int sl = (C.r - 1) & 3;
            const int raw = sl == 0 ? pre_cnt0 : sl == 1 ? pre_cnt1 : sl == 2 ? pre_cnt2 : pre_cnt3;
            const bool lists = (w.cl != nullptr);
            const int nl = lists ? (raw >> 16) : 0;
            auc_decide(C, &AP, lists ? (raw & 0xffff) : raw, n);
            C.pad[0] += nl;          // row_scans: the bids (row evaluations); pad[0]: those served from the list (512 bytes each)
        }
        mode = C.mode;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            asg_book(st, bidding && C.r > 0 ? (C.mode == MODE_CONVERT ? MODE_ARR : C.mode) : MODE_AUCTION);
            if (mode == MODE_CONVERT) {
                // the rounds are over: their results go back into the state block, this launch is the CONVERT step
                st->tag = C.tag; st->eps = C.eps; st->phase = C.phase; st->round = C.round; st->arr_round = C.arr_round;
                st->st_auction_rounds = C.auction_rounds; st->st_arr_rounds = C.arr_rounds;
                st->st_total_row_scans += C.row_scans + w.auc->async_scans; st->st_list_bids = C.pad[0] + w.auc->async_list;
            } else {
                AucCtl N2 = C; N2.r = C.r + 1;
                w.auc->ctl[(par & 1) ^ 1] = N2;
                asg_st(&w.auc->bidcnt[(C.r + 1) & 3], 0);
            }
        }
        if (mode == MODE_CONVERT) __syncthreads();      // (workgroup 0: the state block before ctrl_convert reads it)
    }
    if (mode == MODE_AUCTION || mode == MODE_ARR) {
        double* p_lds = reinterpret_cast<double*>(step_lds);
        int* r_lds = reinterpret_cast<int*>(step_lds + (size_t)((n + 1) & ~1) * sizeof(double));
        const double eps = C.eps; const int tag = C.tag, rb = H.rb;
        const int rnd = (mode == MODE_ARR) ? min(C.arr_round + 1, (1 << ASG_RND_BITS) - 1) : 0;
        if (threadIdx.x == 0) { sh[0] = 0; sh[1] = 0; }
        if (stage_p) {
            const int mb = rb + ASG_RND_BITS;
            // (a pair may straddle n when n is odd: the arrays are padded by one element)
#pragma unroll
            for (int q = 0; q < KP; ++q) {
                const int j = threadIdx.x * 2 + 2 * WT * q;
                if (j < n) { *reinterpret_cast<double2*>(p_lds + j) = make_double2(asg_price(kst[q].x, mb), asg_price(kst[q].y, mb));
                             *reinterpret_cast<int2*>(r_lds + j) = make_int2(asg_key_row(kst[q].x, rb), asg_key_row(kst[q].y, rb)); }
            }
        }
        __syncthreads();
        __shared__ int bq[ASG_BQ];
        const int nb = queue ? wide_bid_queue(M, w, p_lds, r_lds, bq, &sh[1], my_bc, stage_p, n, eps, tag, rb, rnd, (int)blockIdx.x, (int)gridDim.x)
                             : wide_bid(M, w, p_lds, r_lds, wave_gid, n_waves, make_int4(pre_bc, pre_bc1, pre_bc2, pre_bc3), pre_e, pre_T, stage_p, n, eps, tag, rb, rnd);
        if (lane == 0 && nb) atomicAdd(&sh[0], nb);
        __syncthreads();
        // the round's bidders, for the decision the next launch takes (no arrival: nothing of this launch needs it)
        if (threadIdx.x == 0 && sh[0]) atomicAdd(&w.auc->bidcnt[C.r & 3], sh[0]);
        return;
    } else if (mode == MODE_SAP) {
        double* sh_d = reinterpret_cast<double*>(step_lds);
        int* sh_i = reinterpret_cast<int*>(step_lds + sizeof(double) * WT);
        wide_relax(M, w, st, sh_d, sh_i, sh_i + WT, n);
    } else if (mode == MODE_UMIN0) {
        wide_umin0(M, w, st, wave_gid, n_waves, n, reinterpret_cast<float*>(step_lds));
    } else if (mode == MODE_INITRED) {
        wide_initred(M, w, reinterpret_cast<double*>(step_lds), n, st->rb);
    } else if (mode == MODE_UMIN) {
        wide_umin(M, w, st, wave_gid, n_waves, false, n);
    } else if (mode == MODE_COLRED) {
        wide_colred(M, w, st, reinterpret_cast<double*>(step_lds), n);
    } else if (mode == MODE_ROOTMIN) {
        wide_umin(M, w, st, wave_gid, n_waves, true, n); wide_ms_reset(w, n);
    } else if (mode == MODE_CERT) {
        wide_cert(M, w, st, wave_gid, n_waves, (wave_gid < n) ? w.a[wave_gid] : 0, reinterpret_cast<double*>(step_lds), n);
    } else if (blockIdx.x == 0) {
        if (mode == MODE_CONVERT) {
            const int nF = ctrl_convert(w, st, sh);
            if (threadIdx.x == 0) {
                if (nF == 0) asg_enter_cert(st);
                asg_st(&st->next_mode, nF == 0 ? MODE_CERT : MODE_UMIN);
            }
        } else {      // MODE_MS_FINISH
            const bool use_lds = (n <= 6144);
            ctrl_ms_finish(w, st, reinterpret_cast<int*>(step_lds), reinterpret_cast<int*>(step_lds) + n, use_lds, shd, shi, sh);
            if (threadIdx.x == 0) {
                const int nF = st->nF;
                int nm = MODE_ROOTMIN;
                if (st->error) nm = MODE_DONE;
                else if (nF == 0) { asg_enter_cert(st); nm = MODE_CERT; }
                else if (st->sparse && nF <= st->handoff) nm = MODE_BUILD;
                asg_st(&st->next_mode, nm);
            }
        }
    }
    const bool wait = (mode == MODE_SAP || mode == MODE_UMIN0 || mode == MODE_CERT || mode == MODE_CONVERT || mode == MODE_MS_FINISH);
    if (asg_arrive_last(st, w.arrive_sub, &sh[30], payload, wait)) step_ctrl(w, st, mode, n, sh[31], par & 1);
}

// ------------------------------------------------------- asynchronous phase A ------
// The epsilon > 0 phases WITHOUT a global round.  One launch: every workgroup loops on its own rows at its own pace —
//   refresh: all keys, read past the L1 (device-scope loads), become the workgroup's price / owner snapshot in LDS;
//   bid:     its unmatched rows bid on that snapshot exactly as in a synchronous round (wide_bid_queue / wide_bid: bid
//            lists, row scans, one atomicMax per bid);
//   report:  the number of rows it found unmatched goes to its slot of cnt[] (tagged with the phase);
// and workgroup 0 adds up the slots after each of its own iterations and moves the phase word on (cut at <= 2 % unmatched
// like the synchronous rounds).  No kernel boundary and no grid barrier between "rounds": a workgroup's iteration is two
// dependent round trips (refresh, row or list) + an atomic, 2.5 - 3.5 us against 8.6 us for a round as a launch — and
// nobody waits for the slowest workgroup of the chip.
// Why this is allowed: (i) an auction with OUTDATED prices is still an auction — prices only rise (atomicMax), a bid
// computed from older (lower) prices p' <= p leaves c_ij + b = w' + eps <= min_k (c_ik + p_k) + eps if it is accepted
// (Bertsekas' asynchronous auction), and a bid below the object's current price is simply not accepted: the row finds
// itself unmatched at its next look; (ii) the bid lists stay valid for the same reason they do across rounds (their
// bound only needs prices that never fall); (iii) NOTHING downstream trusts phase A: the epsilon = 0 rounds start
// from the prices alone ("every row is unassigned again, the prices stay"), and phases B - D are exact on any prices.
// The only products of this kernel are the keys (prices), the bid lists and a few counters.
// Residency: the phases wait for workgroups that have not started yet (a slot never written counts as "all rows
// unmatched") — but only for ASG_ASYNC_GRACE looks of the controller, so two auction grids from different streams that
// each hold a part of the chip cannot stall each other for long; the grid is capped at the CU count (asg_run).  Every
// loop is capped.
#define ASG_ASYNC_ITER_CAP 60000
// Looks of the controller during which a workgroup that has not started counts as "all rows unmatched".  Its only purpose
// is to break the mutual wait of two chip-sized auction grids that each hold a part of the chip (a 40 ms stall instead of
// the loop caps' 0.3 s) — NOT to run ahead of late workgroups: beside the dense products of a training loop an auction
// workgroup (16 waves, the whole register file of a CU) often waits hundreds of microseconds for an EMPTY CU, and
// phases that went on without those rows left the list solver more than 64 free rows now and then — the dense
// fallback, 10-20 ms.  Measured (profiles/r5_async_sweep.txt): 64 looks: one region in three at 1.3-2.05 ms per step;
// 8192 looks: none above 1.1 in 33 regions but one (1.32).
#define ASG_ASYNC_GRACE 8192
#define ASG_GROUPS_MAX 512     // row groups of an auction grid (= its gridDim.x: <= 512, asg_run)
#define ASG_ADOPT_AFTER 8      // iterations (~5 us each) after which a running workgroup starts adopting unclaimed groups, one per iteration
// wave-uniform values that came out of vector loads: into scalar registers (the loop carries a dozen of them)
__device__ __forceinline__ double asg_uni_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int asg_uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__global__ __launch_bounds__(WT) void asg_auction(AsgWs w0, int n_host, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char auc_lds[];
    const AsgWs w = asg_shift(w0, stride * blockIdx.y);
    __shared__ int sh[12];
    __shared__ int bq[ASG_BQ];
    __shared__ short grp_list[ASG_GROUPS_MAX];       // the row groups this workgroup works on (claim order)
    __shared__ int own_val[ASG_GROUPS_MAX];          // its own reports by group (-1: not one of its groups)
    __shared__ int cst[6];                           // controller state (its lane 0 only): rounds of the phase, rounds in all, epsilon = 0 rounds, the phase's cut
                                                     // (in LDS, not registers: values only one lane updates inside the persistent loop cost every lane a VGPR)
    AsgState* st = w.st;
    const int n = n_host;
    const AsgHead H = *reinterpret_cast<const AsgHead*>(st);
    if (H.mode != MODE_AUCTION || H.error || w.auc->async_done) return;
    gfp M = ASG_GLOBAL(H.Mptr);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int G = (int)gridDim.x;                    // row groups = the grid the host chose; group g = rows g, g + G, ...
    // ---- claim a row group.  Rows are NOT tied to blockIdx: the workgroups take the groups in the order they START, and a
    // running workgroup ADOPTS groups nobody has claimed yet (below).  A phase ends when every group has reported, and
    // with the static mapping of round 5 a workgroup that could not start — beside the dense products of a training loop
    // a 1024-thread workgroup waits hundreds of microseconds for an empty CU; two or three auction grids from different
    // streams can each hold a part of the chip and wait for the rest of it — stalled the whole grid: 27 - 30 ms
    // (the grace) whenever a chip-sized grid of a lone solve met the grids of two prefetch jobs (profiles/
    // r6_tail_public.txt), and a fat tail of the pipelined step otherwise.  Now nobody waits for a workgroup that
    // is not running: its rows are somebody else's after ASG_ADOPT_AFTER iterations, and a workgroup that starts
    // with nothing left to claim exits at once.
    if (threadIdx.x == 0) sh[8] = __hip_atomic_fetch_add(&w.auc->async_claim, 1, __ATOMIC_RELAXED, ASG_AGENT);
    for (int g = threadIdx.x; g < ASG_GROUPS_MAX; g += WT) own_val[g] = -1;
    __syncthreads();
    const int first = asg_uni_i(sh[8]);
    if (first >= G) return;                          // every group is taken: nothing to do for this workgroup
    if (threadIdx.x == 0) grp_list[0] = (short)first;
    int ng = 1;                                      // (uniform) groups held
    const bool controller = (first == 0);            // the first workgroup to start moves the phase word
    const int rb = asg_uni_i(H.rb), mb = rb + ASG_RND_BITS;
    const double theta = asg_uni_d(H.theta), eps_last = asg_uni_d(H.eps_last);
    const double stop_frac = H.stop_frac, stop_early = st->stop_early;
    const int round_cap = asg_uni_i(H.round_cap);
    int* cnt = w.grp_ticket;
    double* p_lds = reinterpret_cast<double*>(auc_lds);
    int* r_lds = reinterpret_cast<int*>(auc_lds + (size_t)((n + 1) & ~1) * sizeof(double));
    constexpr int KP = WIDE_PLDS_MAX / (2 * WT);
    // local view of the schedule: phase index, its epsilon (the controller divides the same way) and tag
    int phase = 0; double eps = asg_uni_d(H.eps);
    int scans_tot = 0, lists_tot = 0;
    const int pad0 = asg_uni_i(st->pad0);
    const int last_div = (pad0 & 0xff) > 0 ? (pad0 & 0xff) : 1;
    const bool do_arr = ((pad0 >> 8) & 1) != 0;                      // the epsilon = 0 rounds run here too (below)
    const int grace = (pad0 >> 16) > 0 ? (pad0 >> 16) * 64 : ASG_ASYNC_GRACE;
    // (an epsilon = 0 iteration costs a workgroup ~5 us here, not a launch of the whole chip: 2 x the synchronous cap + 4.
    //  Measured over 40 C3 instances, profiles/r5_async_sweep.txt: 10 / 16 / 24 iterations leave 36.5 / 32.3 / 28.4 free
    //  rows to the list solver, lone solve 2.42 / 2.43 / 2.33 ms)
    const int arr_cap = 2 * asg_uni_i(H.arr_cap) + 4;
    int arr_it = 0;
    if (threadIdx.x == 0) {      // (cst[4] / cst[5]: the cut of a middle / the last phase, in rows)
        const int stop_mid = (int)(fmax(stop_frac, stop_early) * n), stop_last = (int)(stop_frac / last_div * n);
        cst[0] = 0; cst[1] = 0; cst[2] = 0; cst[3] = ((eps / theta) < eps_last) ? stop_last : stop_mid; cst[4] = stop_mid; cst[5] = stop_last;
    }
    bool idle = false;
    int step = 0;                                                    // group-steps done (parity of the LDS counters)
    for (int it = 0; it < ASG_ASYNC_ITER_CAP; ++it) {
        if (idle) __builtin_amdgcn_s_sleep(20);                      // nothing to bid for at the last look: poll a little slower
        // ---- adopt: groups nobody has claimed after this workgroup's first ASG_ADOPT_AFTER iterations belong to
        // workgroups that are not running; take one per iteration (thread 0 decides, the barrier below publishes)
        if (threadIdx.x == 0) {
            int got = -1;
            if (it >= ASG_ADOPT_AFTER && ng < ASG_GROUPS_MAX && asg_ld(&w.auc->async_claim) < G) {
                got = __hip_atomic_fetch_add(&w.auc->async_claim, 1, __ATOMIC_RELAXED, ASG_AGENT);
                if (got >= G) got = -1;
            }
            if (got >= 0) { grp_list[ng] = (short)got; atomicAdd(&w.auc->async_adopted, 1); }
            sh[9] = got;
        }
        // ---- refresh (past the L1: the keys are raised by other CUs' atomics, the phase word by the controller) ----
        const int word0 = asg_ld(&w.auc->async_word);
        unsigned long long kk[2 * KP];
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const int j = threadIdx.x * 2 + 2 * WT * q;
            const int jj = j < n ? j : 0;
            kk[2 * q] = asg_ld(&w.key[jj]); kk[2 * q + 1] = asg_ld(&w.key[jj + 1 < n ? jj + 1 : jj]);
        }
        // (the first group's last bids travel with the keys — a dependent round trip per iteration otherwise — and its LDS
        //  counters are reset in front of the refresh barrier; further, adopted groups pay both inside the loop below)
        int bc0 = -1;
        {   int tq = threadIdx.x; asm volatile("" : "+v"(tq));       // (opaque: the address is formed per iteration, not carried — and spilled — across the loop)
            if (tq < ASG_BQ && first + G * tq < n) bc0 = asg_ld(&w.bidcol[first + G * tq]); }
        if (threadIdx.x == 0) { sh[2] = word0; int* s0 = sh + 4 * (step & 1); s0[0] = 0; s0[1] = 0; }
#pragma unroll
        for (int q = 0; q < KP; ++q) {
            const int j = threadIdx.x * 2 + 2 * WT * q;
            if (j < n) { *reinterpret_cast<double2*>(p_lds + j) = make_double2(asg_price(kk[2 * q], mb), asg_price(kk[2 * q + 1], mb));
                         *reinterpret_cast<int2*>(r_lds + j) = make_int2(asg_key_row(kk[2 * q], rb), asg_key_row(kk[2 * q + 1], rb)); }
        }
        __syncthreads();
        // (the waves of a workgroup may have read different words: thread 0's copy decides for all of them)
        const int word = asg_uni_i(sh[2]);
        if (asg_uni_i(sh[9]) >= 0) ++ng;
        if (word < 0) break;                                         // (uniform) the phases are over
        const int wphase = word & 0xffff;
        while (phase < wphase) { eps = asg_uni_d(eps / theta); ++phase; }       // a new phase: every row is unassigned again (new tag)
        // The epsilon = 0 rounds (bit 30 of the word; phase = the number of epsilon phases run): the same loop with
        // epsilon = 0 and a fresh tag.  A kept pair is exactly tight whatever the timing: the bid b = p'_j + (second' -
        // best'), from a snapshot p' <= p, leaves c_ij + b = second' <= c_ik + p'_k <= c_ik + p_k for every other k if it
        // is accepted, and prices only rise afterwards.  Equal-price bids are decided by the round field as in the
        // synchronous rounds — here the bidder's own count of epsilon = 0 iterations: a later try beats an earlier one.
        const bool arr = ((word >> 30) & 1) != 0;
        const int tag = (phase % 254) + 1;                           // (the phases used tags 1 .. phase; this one is new)
        const double eps_use = arr ? 0.0 : eps;
        const int rnd = arr ? min(arr_it + 1, (1 << ASG_RND_BITS) - 1) : 0;
        arr_it += arr ? 1 : 0;
        const int slot_tag = (phase + 1) | (arr ? 0x4000 : 0);
        // ---- bid, group by group: the synchronous round's body (queue form) on this snapshot — thread t looks at row t of
        // the group, the unmatched ones go to the LDS queue and the waves take them from there; then the group's report
        int mine_all = 0;
        for (int q = 0; q < ng; ++q, ++step) {
            const int grp = asg_uni_i((int)grp_list[q]);
            int* shc = sh + 4 * (step & 1);        // (this step's counters: a slow wave may still be reading the previous step's)
            int my_bc = bc0;
            if (q > 0) {                           // (uniform)
                if (threadIdx.x == 0) { shc[0] = 0; shc[1] = 0; }
                my_bc = -1;
                if (threadIdx.x < ASG_BQ && grp + G * (int)threadIdx.x < n) my_bc = asg_ld(&w.bidcol[grp + G * (int)threadIdx.x]);
                __syncthreads();
            }
            const int nb = wide_bid_queue(M, w, p_lds, r_lds, bq, &shc[1], my_bc, true, n, eps_use, tag, rb, rnd, grp, G);
            if (lane == 0 && nb) atomicAdd(&shc[0], nb);
            __syncthreads();
            const int mine = asg_uni_i(shc[0]);                      // bids of this step = rows found unmatched (+ list-served ones above bit 16)
            mine_all += mine & 0xffff;
            scans_tot += mine & 0xffff; lists_tot += mine >> 16;
            if (threadIdx.x == 0) {
                const int rep = (slot_tag << 16) | (mine & 0xffff);
                asg_st(&cnt[grp], rep);
                own_val[grp] = rep;
            }
        }
        idle = (mine_all == 0);
        // ---- the controller decides ----
        if (controller && wv == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (own_val: written by lane 0 of this wave)
            int tot = 0;
            int ln = lane; asm volatile("" : "+v"(ln));              // (opaque: the slot address is formed here, not carried — and spilled — across the loop)
            for (int g = ln; g < G; g += 64) {
                // (the controller's own slots may not have landed yet: it reads its own reports from LDS)
                const int ov = own_val[g];
                const int c = ov >= 0 ? ov : asg_ld(&cnt[g]);
                const int rg = (n - g + G - 1) / G;
                // a slot of another phase: all of its rows (its workgroup is about to report).  A slot NEVER written (0
                // since the init step): nobody has reported for the group yet — all of its rows.  Since the groups are
                // claimed and adopted nobody waits for a workgroup that is not running; the grace of round 5
                // (ASG_ASYNC_GRACE looks) remains as a safety net only
                tot += ((c >> 16) == slot_tag) ? (c & 0xffff) : ((c != 0 || it < grace) ? rg : 0);
            }
            tot = wave_sum_i(tot);
            if (lane == 0) {
                int nw = 0;                                          // 0: go on; != 0: new word
                if (arr) {
                    const int ca = ++cst[2];
                    if (tot == 0 || ca >= arr_cap || it + 8 >= ASG_ASYNC_ITER_CAP) nw = (int)0x80000000u | phase;
                } else {
                const int cr = ++cst[0]; ++cst[1];
                if (tot <= cst[3] || cr >= round_cap || it + 8 >= ASG_ASYNC_ITER_CAP) {
                    const double e2 = eps / theta;
                    if (e2 < eps_last || it + 8 >= ASG_ASYNC_ITER_CAP)
                        nw = (do_arr && it + 8 < ASG_ASYNC_ITER_CAP) ? (0x40000000 | (phase + 1)) : ((int)0x80000000u | (phase + 1));
                    else {
                        nw = phase + 1; cst[0] = 0;
                        cst[3] = ((e2 / theta) < eps_last) ? cst[5] : cst[4];
                    }
                }
                }
                if (nw) asg_st(&w.auc->async_word, nw);
            }
        }
    }
    // ---- leave: the bid totals; the controller hands the state machine over to the epsilon = 0 rounds ----
    if (threadIdx.x == 0) {
        if (scans_tot) atomicAdd(&w.auc->async_scans, scans_tot);
        if (lists_tot) atomicAdd(&w.auc->async_list, lists_tot);
        if (controller) {
            // (groups never claimed — the whole solve ran on fewer workgroups than groups, all adopted — stay closed for
            //  workgroups that start after the hand-over: they leave on async_done / the mode)
            const int word = asg_ld(&w.auc->async_word);
            if (word >= 0) asg_st(&w.auc->async_word, (int)0x80000000u | (word & 0xffff));   // (left by the cap)
            const int nph = (word & 0xffff);
            // the epsilon = 0 rounds ran here (c_arr_rounds > 0): the next asg_step launch is the CONVERT step; else it
            // is their first round
            const int c_arr_rounds = cst[2], c_total_rounds = cst[1];
            AucCtl C;
            C.eps = 0.0; C.mode = c_arr_rounds > 0 ? MODE_CONVERT : MODE_ARR; C.tag = (nph % 254) + 1; C.round = 0; C.phase = nph; C.stop = 0;
            C.arr_round = c_arr_rounds;
            C.r = 0; C.auction_rounds = c_total_rounds; C.arr_rounds = c_arr_rounds; C.row_scans = 0; C.pad[0] = C.pad[1] = 0;
            w.auc->ctl[0] = C; w.auc->ctl[1] = C;                    // whichever parity the next asg_step launch has
            for (int q = 0; q < 4; ++q) asg_st(&w.auc->bidcnt[q], 0);
            w.auc->async_done = 1;
            asg_book(st, MODE_AUCTION);
        }
    }
}

// ---------------------------------------------------------------- build ------
__global__ __launch_bounds__(SP_BUILD_WAVES * 64) void asg_build(AsgWs w0, int n_host, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char build_lds[];
    const AsgWs w = asg_shift(w0, stride * blockIdx.y);
    __shared__ int sh_flag[4];
    AsgState* st = w.st;
    const int mode = st->mode;
    gfp M = ASG_GLOBAL(st->Mptr);
    if (mode != MODE_BUILD || st->error) return;
    wide_build(M, w, st, build_lds);
    if (asg_arrive_last(st, w.arrive_sub, sh_flag, 0u, false) && threadIdx.x == 0) { asg_book(st, mode); st->mode = MODE_SOLVER; }
}

// --------------------------------------------------------------- solver ------
__global__ __launch_bounds__(SP_T) void asg_solve(AsgWs w0, int n_host, size_t stride) {
    extern __shared__ __attribute__((aligned(16))) char solve_lds[];
    const AsgWs w = asg_shift(w0, stride * blockIdx.y);
    AsgState* st = w.st;
    const int mode = st->mode;
    gfp M = ASG_GLOBAL(st->Mptr);
    if (mode != MODE_SOLVER || st->error) return;
    sp_solver(M, w, st, solve_lds);      // one workgroup: publishes MODE_CERT (or an error) itself
}

// --------------------------------------------------------- init / trivial ----
__global__ void asg_init(AsgWs w, AsgState h) {
    // the state block arrives as a kernel argument: no host staging buffer, stream ordered
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&h);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(w.st);
    for (int q = threadIdx.x; q < (int)(sizeof(AsgState) / 8); q += blockDim.x) dst[q] = src[q];
    for (int q = threadIdx.x; q < 2048 / 8; q += blockDim.x) w.arrive_sub[q] = 0ull;
}

__global__ void asg_trivial(const float* M, int n, int* perm, int* certified, double* total_cost,
                            int* stats) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (n == 1) { perm[0] = 0; if (total_cost) *total_cost = (double)M[0]; }
        if (certified) *certified = 1;
        if (stats) for (int k = 0; k < 8; ++k) stats[k] = 0;
    }
}

// ------------------------------------------------------------------ host -----
// The kernels take only workspace-derived arguments, so the launch programs are captured once per
// host thread / workspace into hipGraphs and replayed (a solve is ~200 launches; with several
// couplings in flight on different streams the host launch rate would be the limit).  Falls back
// to plain launches when the stream cannot be captured (the legacy default stream) or CFM_ASG_GRAPH=0.
//   PRG_BULK   `bulk` x asg_step                                  (unpolled head of a solve)
//   PRG_CHUNK  `chunk` x asg_step, asg_build, asg_solve, 2 x asg_step   (polled; progresses from any state)
enum { PRG_CHUNK = 0, PRG_BULK = 1, PRG_COUNT = 2 };
struct AsgGraph {
    void* ws = nullptr; int n = 0, nb = 0, chunk = 0, bulk = 0, blocks = 0, sparse = 0;
    hipGraphExec_t exec[PRG_COUNT] = {nullptr, nullptr};
    hipStream_t stream = nullptr; int disabled = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int ev_blocking = -1;          // how ev[] were created (this thread's g_blocking_sync at the time)
};
// A host thread keeps the programs of its last few (workspace, size, batch, stream) combinations: a training loop
// alternates between a few of them (groups of couplings and a shorter last group, single solves), and capturing +
// instantiating the two programs costs milliseconds.
#define ASG_GRAPH_SLOTS 4
static thread_local AsgGraph g_graphs[ASG_GRAPH_SLOTS];
static thread_local unsigned g_graph_use[ASG_GRAPH_SLOTS];
static thread_local unsigned g_graph_clock = 0;
static thread_local int g_graph_off = 0;      // this thread's streams cannot be captured: plain launches from now on
// The host wait of a solve (ONE per solve since round 5): with HIP's default an event wait SPINS on a host core; a
// coupling worker of a training loop (cfm_amd.prefetch: 3 per rank, 8 ranks per node) burns a core each for the whole
// solve.  cfm_set_blocking_sync(1) makes this THREAD's solver waits yield: its events carry hipEventBlockingSync AND the
// wait itself is a poll (hipEventQuery) with a 20 us sleep in between — on this stack (ROCm 7, torch 2.10) a
// "blocking" event wait was measured to spin exactly like the default (tools/probe/blocking_sync_probe.py: thread CPU
// time == wall time for torch.cuda.Event(blocking=True) and for hipEventBlockingSync alike), so the flag alone buys
// nothing.  The poll notices completion up to one sleep (~60 us with the kernel's timer slack) late — once per job of
// several couplings, not per step.  Thread-local: a latency-critical lone solve on the caller's own thread keeps the spin.
static thread_local int g_blocking_sync = 0;
extern "C" void cfm_set_blocking_sync(int on) { g_blocking_sync = on ? 1 : 0; }
static hipError_t asg_wait(hipEvent_t ev) {
    if (!g_blocking_sync) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();                  // (hipErrorNotReady is sticky in hipGetLastError otherwise)
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
}
static hipError_t asg_events(AsgGraph& G) {
    if (G.ev_blocking != g_blocking_sync) {
        for (int q = 0; q < 2; ++q) if (G.ev[q]) { (void)hipEventDestroy(G.ev[q]); G.ev[q] = nullptr; }
        G.ev_blocking = g_blocking_sync;
    }
    const unsigned flags = hipEventDisableTiming | (g_blocking_sync ? hipEventBlockingSync : 0u);
    for (int q = 0; q < 2; ++q)
        if (!G.ev[q]) { hipError_t e = hipEventCreateWithFlags(&G.ev[q], flags); if (e != hipSuccess) return e; }
    return hipSuccess;
}
static AsgGraph& asg_graph_slot(void* ws, int n, int nb, hipStream_t s) {
    int hit = -1, lru = 0;
    for (int q = 0; q < ASG_GRAPH_SLOTS; ++q) {
        const AsgGraph& G = g_graphs[q];
        if (G.exec[0] && G.ws == ws && G.n == n && G.nb == nb && G.stream == s) hit = q;
        if (g_graph_use[q] < g_graph_use[lru]) lru = q;
    }
    const int q = hit >= 0 ? hit : lru;
    g_graph_use[q] = ++g_graph_clock;
    return g_graphs[q];
}

static int asg_graph_enabled() {
    static const int v = [] { const char* e = getenv("CFM_ASG_GRAPH"); return (e && e[0] == '0') ? 0 : 1; }();
    return v;
}

// dynamic LDS above the 64 KiB default needs the attribute: once per device
static int asg_raise_lds() {
    static std::once_flag once[16];
    static int ok[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    std::call_once(once[dev], [dev] {
        hipError_t e1 = hipFuncSetAttribute((const void*)asg_step, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        if (e1 == hipSuccess) e1 = hipFuncSetAttribute((const void*)asg_auction, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        hipError_t e2 = hipFuncSetAttribute((const void*)asg_build, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
        hipError_t e3 = hipFuncSetAttribute((const void*)asg_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ok[dev] = (e1 == hipSuccess ? 1 : 0) | (e2 == hipSuccess && e3 == hipSuccess ? 2 : 0);
        (void)hipGetLastError();
    });
    return ok[dev];
}

// poll buffer (pinned host memory): one per host thread, concurrent solves on different streams
// must not share it
static thread_local int* g_pinned = nullptr;
static thread_local int g_small_last[16];      // status block of this thread's last one-workgroup solve (phase times)
// tuning aid: {solves of this PROCESS that the dense state machine had to redo, last device error code} — process-wide
// since round 6: the couplings of a training loop run on prefetch worker threads, and the bench line reports the count
struct AsgFallback {
    std::atomic<int> v[2];
    struct Ref { std::atomic<int>& a; Ref& operator=(int x) { a.store(x, std::memory_order_relaxed); return *this; }
                 Ref& operator++() { a.fetch_add(1, std::memory_order_relaxed); return *this; } };
    Ref operator[](int q) { return Ref{v[q]}; }
};
static AsgFallback g_fallback;
extern "C" void cfm_assign_debug_fallback(int* out2) { out2[0] = g_fallback.v[0].load(); out2[1] = g_fallback.v[1].load(); }
extern "C" void cfm_assign_debug_small(int* out16) { for (int q = 0; q < 16; ++q) out16[q] = g_small_last[q]; }

// CUs the stream may use: the device's count, or the population of its CU mask (hipExtStreamCreateWithCUMask streams).
// Cached per host thread for its last few streams (the query is a host-side lookup, but it sits on every solve's path).
static int asg_stream_cus(hipStream_t s) {
    static int cus[CFM_MAX_DEVICES];
    const int di = cfm_device_index();
    int& c = cus[di];
    if (c <= 0 && (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, di) != hipSuccess || c <= 0)) c = 256;
    if (!s) return c;
    static thread_local struct { hipStream_t s; int dev, n; } cache[4];
    static thread_local unsigned clock_ = 0;
    for (auto& e : cache) if (e.s == s && e.dev == di && e.n > 0) return e.n;
    uint32_t mask[16] = {0};
    int n = c;
    if (hipExtStreamGetCUMask(s, 16, mask) == hipSuccess) {
        int pop = 0;
        for (int q = 0; q < 16; ++q) pop += __builtin_popcount(mask[q]);
        if (pop > 0 && pop < n) n = pop;
    } else (void)hipGetLastError();
    auto& e = cache[clock_++ & 3];
    e.s = s; e.dev = di; e.n = n;
    return n;
}

struct AsgLaunch {
    AsgWs w; int n, blocks, blocks_build, nb; size_t stride; size_t lds_step, lds_build, lds_solve; int sparse; hipStream_t s;
    int async_auction = 0, blocks_auction = 0;
    // every program holds an EVEN number of asg_step launches and starts on an even launch count, so the parity
    // argument (which control record a bid round reads, see AucCtl) is the position inside the program.
    // grid.y = the problems of a batch (one carving each, `stride` bytes apart)
    void step(int par) const { hipLaunchKernelGGL(asg_step, dim3(blocks, nb), dim3(WT), lds_step, s, w, n, par & 1, stride); }
    void program(int prg, int chunk, int bulk) const {
        int k = 0;
        // asynchronous phase A: ONE launch behind the two init steps (a no-op in any other state, like every kernel here)
        auto auction = [&]() { if (async_auction) hipLaunchKernelGGL(asg_auction, dim3(blocks_auction, nb), dim3(WT), lds_step, s, w, n, stride); };
        if (prg == PRG_BULK) {
            if (async_auction >= 2 && sparse) {
                // The WHOLE solve as the unpolled head when the bid rounds (epsilon > 0 and epsilon = 0) are the one auction
                // launch and the list solver closes the search: 2 init steps, the auction, convert / row minima / column
                // reduction (+ one spare step: an even count), list build, list solver, certificate (+ one spare).  A solve
                // that takes this road — every C3 instance seen so far — is finished when the head is; the others are
                // picked up by the polled chunks.  (Round 4's head was 96 steps, then chunks of 10 steps + the pair + 2: a
                // lone solve paid ~35 no-op launches, 0.15 ms, around its list build and behind its last step.)
                step(k++); step(k++);
                auction();
                step(k++); step(k++); step(k++); step(k++);
                hipLaunchKernelGGL(asg_build, dim3(blocks_build, nb), dim3(SP_BUILD_WAVES * 64), lds_build, s, w, n, stride);
                hipLaunchKernelGGL(asg_solve, dim3(1, nb), dim3(SP_T), lds_solve, s, w, n, stride);
                step(k++); step(k++);
                return;
            }
            for (int c = 0; c < bulk; ++c) { if (c == 2) auction(); step(k++); }
            return;
        }
        auction();
        for (int c = 0; c < chunk; ++c) step(k++);
        if (sparse) {
            hipLaunchKernelGGL(asg_build, dim3(blocks_build, nb), dim3(SP_BUILD_WAVES * 64), lds_build, s, w, n, stride);
            hipLaunchKernelGGL(asg_solve, dim3(1, nb), dim3(SP_T), lds_solve, s, w, n, stride);
            step(k++); step(k++);      // certificate + whatever the guess missed
        }
    }
    int count(int prg, int chunk, int bulk) const {
        if (prg == PRG_BULK && async_auction >= 2 && sparse) return bulk > 0 ? 11 : 0;
        return (prg == PRG_BULK ? bulk : chunk + (sparse ? 4 : 0)) + ((async_auction && (prg != PRG_BULK || bulk > 2)) ? 1 : 0);
    }
};

// the first 64 bytes of every problem's state block, gathered for ONE copy to the host
__global__ void asg_collect(const AsgState* st0, size_t stride, int nb, int* out) {
    const int b = threadIdx.x >> 4, q = threadIdx.x & 15;
    if (b < nb) out[16 * b + q] = reinterpret_cast<const int*>(reinterpret_cast<const char*>(st0) + stride * b)[q];
}

struct AsgProblem { const float* M; int* perm; int* certified; double* total_cost; int* stats; };
#define ASG_BATCH_MAX 16
#define ASG_PINNED_INTS (2 * ASG_BATCH_MAX * 16)

static int asg_pinned() {
    if (g_pinned) return 0;
    return cfm_hip(hipHostMalloc((void**)&g_pinned, ASG_PINNED_INTS * sizeof(int), hipHostMallocDefault));
}

// Solves nb problems of the same size on one chain of launches (grid.y = problem).  cert_out[b] / err_out[b]: the
// certificate and the device error code of problem b (0 = none).  nb == 1: the plain solve.
static int asg_run(const AsgProblem* pr, int nb, int B, void* ws, size_t stride, void* stream, const AsgParams& P,
                   int use_sparse, int* cert_out, int* err_out) {
    if (!pr || nb < 1 || nb > ASG_BATCH_MAX || B < 0 || (B > 1 && !ws)) return CFM_EINVAL;
    for (int b = 0; b < nb; ++b) if (!pr[b].M || !pr[b].perm) return CFM_EINVAL;
    if (B > (1 << 20)) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    for (int b = 0; b < nb; ++b) { cert_out[b] = 1; err_out[b] = 0; }
    if (B == 0) return 0;
    if (B == 1) {
        for (int b = 0; b < nb; ++b)
            hipLaunchKernelGGL(asg_trivial, dim3(1), dim3(64), 0, s, pr[b].M, B, pr[b].perm, pr[b].certified, pr[b].total_cost, pr[b].stats);
        return cfm_status();
    }
    if (((uintptr_t)ws & 15) != 0 || (stride & 15) != 0) return CFM_EALIGN;
    for (int b = 0; b < nb; ++b) if (((uintptr_t)pr[b].M & 15) != 0) return CFM_EALIGN;
    const int n = B;
    AsgLaunch L;
    L.w = asg_carve(ws, n); L.n = n; L.s = s; L.nb = nb; L.stride = nb > 1 ? stride : 0;
    int rc = asg_pinned();
    if (rc) return rc;
    int wide_blocks = (n + 15) / 16;        // one wave per row when everything bids
    if (wide_blocks > 512) wide_blocks = 512;
    if (P.wide_blocks_cap > 0 && wide_blocks > P.wide_blocks_cap) wide_blocks = P.wide_blocks_cap;
    L.blocks_build = wide_blocks < (n + 63) / 64 ? (n + 63) / 64 : wide_blocks;     // the list build streams the matrix once: its own grid
    // a batch shares the chip: ASG_BATCH_WGS workgroups in all (every workgroup of a bid round stages the prices whether
    // its rows bid or not; measured at n = 4096: 8 problems 9.1 ms with 256 workgroups each, 6.5 ms with 64); the rounds
    // then take the queue form (wide_bid_queue)
    const int floor_blocks = nb > 1 ? (n + ASG_BQ - 1) / ASG_BQ : (n + 63) / 64;
#define ASG_BATCH_WGS 256
    if (nb > 1 && wide_blocks > ASG_BATCH_WGS / nb) wide_blocks = ASG_BATCH_WGS / nb;
    if (wide_blocks < floor_blocks) wide_blocks = floor_blocks;      // (the queue form takes ASG_BQ rows per workgroup)
    if (wide_blocks < 1) wide_blocks = 1;
    L.blocks = wide_blocks;
    const int raised = asg_raise_lds();
    L.lds_step = sizeof(double) * WT + 2 * sizeof(int) * WT;                      // relax merge buffers
    if (n <= WIDE_PLDS_MAX) {                                            // bid rounds: prices + owner rows
        const size_t need = (size_t)((n + 1) & ~1) * sizeof(double) + (size_t)(n + 2) * sizeof(int);
        if (need > L.lds_step) L.lds_step = need;
    }
    if (n <= 6144 && (size_t)2 * n * sizeof(int) > L.lds_step) L.lds_step = (size_t)2 * n * sizeof(int);   // path walks of MS_FINISH
    L.lds_step = (L.lds_step + 15) & ~(size_t)15;
    if (L.lds_step > 64 * 1024 && !(raised & 1)) return CFM_EINVAL;
    L.sparse = (use_sparse && n <= SP_NMAX && (raised & 2)) ? 1 : 0;
    L.lds_build = sp_build_lds_bytes(n); L.lds_solve = sp_solver_lds_bytes(n);

    // asynchronous phase A: the keys must fit the LDS snapshot and the grid must give every workgroup at most ASG_BQ rows
    L.blocks_auction = wide_blocks;
    if (nb > 1 && P.async_blocks > 0 && P.async_blocks < wide_blocks) L.blocks_auction = P.async_blocks;
    if ((long)L.blocks_auction * ASG_BQ < n) L.blocks_auction = (n + ASG_BQ - 1) / ASG_BQ;      // (a workgroup takes at most ASG_BQ rows)
    {   // its workgroups (16 waves, the whole register file of a CU each) must be able to be resident TOGETHER: a phase ends
        // when the whole grid has reported, and a workgroup that has not started counts as "all rows unmatched"
        // — on the CUs THIS STREAM may use: a CU-masked stream (cfm_stream_create_cu_mask, ChipPartition) gives the grid
        // fewer than the device has, and workgroups that cannot start before others exit would be counted as "all rows
        // unmatched" for the whole grace (~40 ms) and then left out: > 64 free rows, the dense fallback
        int c = asg_stream_cus(s);
        if ((long)L.blocks_auction * nb > c) L.blocks_auction = c / nb > 0 ? c / nb : 1;
    }
    L.async_auction = (P.async_auction && n >= P.async_min_n && n <= WIDE_PLDS_MAX && (raised & 1) && (long)L.blocks_auction * ASG_BQ >= n) ? P.async_auction : 0;
    for (int b = 0; b < nb; ++b) {
        AsgState h;
        memset(&h, 0, sizeof(h));
        h.wide_blocks = wide_blocks;
        h.mode = MODE_UMIN0; h.n = n; h.Mptr = pr[b].M;
        h.out_perm = pr[b].perm; h.out_cert = pr[b].certified; h.out_cost = pr[b].total_cost; h.out_stats = pr[b].stats;
        h.eps = P.eps0_frac; h.eps_last = P.eps_last_frac; h.theta = L.async_auction ? P.async_theta : P.theta;
        h.stop_frac = P.stop_frac; h.round_cap = P.round_cap; h.arr_cap = P.arr_cap;
        h.cmin_bits = 0xffffffffu; h.cmax_bits = 0u; h.minslack_ord = ~0ull;
        h.fr_min = ~0ull; h.fr_max = 0ull;
        h.sparse = L.sparse; h.handoff = P.handoff; h.stop_early = P.stop_early;
        h.tag = 1; h.pad0 = (P.async_last_div & 0xff) | ((P.async_auction >= 2 ? 1 : 0) << 8) | ((P.async_last_div >> 8) << 16);   // (bits 16+: experiment — grace of unstarted workgroups / 64)
        { int rb = 1; while ((1 << rb) <= n) ++rb; h.rb = rb; }     // row ids 0 .. n-1 and the all-ones "none"
        hipLaunchKernelGGL(asg_init, dim3(1), dim3(64), 0, s, asg_carve((char*)ws + (size_t)b * L.stride, n), h);
    }
    rc = cfm_status();
    if (rc) return rc;

    int chunk = ((P.chunk > 0 ? P.chunk : 10) + 1) & ~1;        // even: see AsgLaunch::step
    // A batch pays for every problem that is not yet at its list build when the first build + solver pair comes by: it
    // is built and solved by the NEXT chunk, behind the others' solver (~1.9 ms at n = 4096).  The steps before the
    // build vary by ~+-6 between problems: 24 instead of 10 steps in front of the pair (a no-op step costs 3-5 us).
#define ASG_BATCH_CHUNK 24
    if (nb > 1 && chunk < ASG_BATCH_CHUNK) chunk = ASG_BATCH_CHUNK;
    // ... the unpolled head then is: 2 init steps, the auction launch, ~10 epsilon = 0 rounds + convert / row minima / column
    // reduction (the synchronous rounds needed ~96 launches here)
    int bulk = ((n >= P.bulk_min_n) ? P.bulk : 0) & ~1;
    if (L.async_auction && bulk > 16) bulk = 16;
    AsgGraph& G = asg_graph_slot(ws, n, nb, s);
    bool use_graph = asg_graph_enabled() && !g_graph_off && n >= 256;
    if (use_graph && !(G.exec[0] && G.ws == ws && G.n == n && G.nb == nb && G.chunk == chunk && G.bulk == bulk &&
                       G.blocks == wide_blocks + 1024 * L.blocks_auction && G.sparse == L.sparse + 2 * L.async_auction && G.stream == s)) {
        for (int q = 0; q < PRG_COUNT; ++q) if (G.exec[q]) { (void)hipGraphExecDestroy(G.exec[q]); G.exec[q] = nullptr; }
        hipError_t e = hipSuccess;
        // The programs are captured on a PRIVATE stream of this host thread, not on the caller's: while a stream captures,
        // HIP refuses any other stream's wait on an event that was recorded on it EARLIER (hipErrorStreamCaptureIsolation) —
        // a prefetch worker that starts a job of a new size while the training thread waits for the worker's previous job
        // (cfm_amd.prefetch: _Handle.result) raised exactly that, once in five runs of tests/test_gpu_prefetch.py.  The
        // kernels take workspace-derived arguments only, so where they are captured does not matter.
        static thread_local hipStream_t cap_stream = nullptr;
        if (!cap_stream) e = hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking);
        AsgLaunch Lc = L; Lc.s = cap_stream;
        for (int prg = 0; prg < PRG_COUNT && e == hipSuccess; ++prg) {
            if (L.count(prg, chunk, bulk) == 0) continue;
            hipGraph_t graph = nullptr;
            e = hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal);
            if (e != hipSuccess) break;
            Lc.program(prg, chunk, bulk);
            e = hipStreamEndCapture(cap_stream, &graph);
            if (e == hipSuccess && graph) e = hipGraphInstantiate(&G.exec[prg], graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (e == hipSuccess) e = asg_events(G);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (int q = 0; q < PRG_COUNT; ++q) if (G.exec[q]) { (void)hipGraphExecDestroy(G.exec[q]); G.exec[q] = nullptr; }
            g_graph_off = 1; use_graph = false;     // e.g. the legacy default stream
        } else {
            G.ws = ws; G.n = n; G.nb = nb; G.chunk = chunk; G.bulk = bulk; G.blocks = wide_blocks + 1024 * L.blocks_auction;
            G.sparse = L.sparse + 2 * L.async_auction; G.stream = s;
        }
    }
    rc = cfm_hip(asg_events(G)); if (rc) return rc;      // (also: the thread's blocking-sync choice changed since they were made)

    long launched = 0;
    auto run = [&](int prg) -> int {
        const int cnt = L.count(prg, chunk, bulk);
        if (cnt == 0) return 0;
        launched += cnt;
        if (use_graph) return cfm_hip(hipGraphLaunch(G.exec[prg], s));
        L.program(prg, chunk, bulk);
        return cfm_status();
    };
    // The head goes out unpolled (a solve at n = 4096 takes ~100 steps before the list solver); the rest
    // in chunks that make progress from any state, each followed by a copy of the first 64 bytes of every
    // problem's state into its own pinned slot and an event, with the NEXT chunk already queued when the host
    // waits for a slot: no idle gap.  A kernel that does not own the current mode is a ~2 us no-op.
    rc = run(PRG_BULK); if (rc) return rc;
    int cur = 0;
    int* stage = reinterpret_cast<int*>((char*)ws + (size_t)nb * stride);      // batches: 2 x nb x 64 bytes behind the carvings
    auto issue = [&](int slot, bool with_chunk = true) -> int {
        int r2 = with_chunk ? run(PRG_CHUNK) : 0; if (r2) return r2;
        int* host = g_pinned + 16 * ASG_BATCH_MAX * slot;
        if (nb == 1) r2 = cfm_hip(hipMemcpyAsync(host, L.w.st, 64, hipMemcpyDeviceToHost, s));
        else {
            hipLaunchKernelGGL(asg_collect, dim3(1), dim3(16 * ASG_BATCH_MAX), 0, s, L.w.st, L.stride, nb, stage + 16 * nb * slot);
            r2 = cfm_status();
            if (!r2) r2 = cfm_hip(hipMemcpyAsync(host, stage + 16 * nb * slot, 64 * (size_t)nb, hipMemcpyDeviceToHost, s));
        }
        if (r2) return r2;
        return cfm_hip(hipEventRecord(G.ev[slot], s));
    };
    // the head holds the whole solve (see AsgLaunch::program): look at the state behind it BEFORE queueing anything else —
    // a finished solve returns here, with no look-ahead chunk of no-ops to wait for
    const bool head_is_whole = L.async_auction >= 2 && L.sparse && bulk > 0;
    if (head_is_whole) {
        rc = issue(0, false); if (rc) return rc;
        rc = cfm_hip(asg_wait(G.ev[0])); if (rc) return rc;
        const int* hs = g_pinned;
        int open_ = 0;
        for (int b = 0; b < nb; ++b) {
            const int mode = hs[16 * b + 0], err = hs[16 * b + 2];
            if (err) err_out[b] = err;
            else if (mode == MODE_DONE) cert_out[b] = hs[16 * b + 3];
            else ++open_;
        }
        if (!open_) return 0;
        for (int b = 0; b < nb; ++b) { err_out[b] = 0; cert_out[b] = 1; }      // (re-read by the chunk loop)
    }
    rc = issue(0); if (rc) return rc;
    int result = 0;
    for (;;) {
        rc = issue(cur ^ 1); if (rc) return rc;
        rc = cfm_hip(asg_wait(G.ev[cur])); if (rc) return rc;
        const int* hs = g_pinned + 16 * ASG_BATCH_MAX * cur;
        int open_ = 0;
        for (int b = 0; b < nb; ++b) {
            const int mode = hs[16 * b + 0], err = hs[16 * b + 2];
            if (err) err_out[b] = err;                        // this problem stopped (its launches are no-ops now)
            else if (mode == MODE_DONE) cert_out[b] = hs[16 * b + 3];
            else ++open_;
        }
        if (!open_) break;
        if (launched >= P.max_launches) { result = CFM_ETIMEOUT; break; }
        cur ^= 1;
    }
    // the look-ahead chunk is still in flight: it is a string of no-ops on a finished state, but the
    // workspace (and the pinned slot it copies into) must not be reused under it
    rc = cfm_hip(asg_wait(G.ev[cur ^ 1]));
    return rc ? rc : result;
}

// one problem through the candidate-list machine, the dense state machine deciding should its certificate ever fail
static int asg_solve_one(const AsgProblem& pr, int B, void* ws, void* stream, const AsgParams& P, int first_sparse) {
    int cert = 1, err = 0;
    int rc = asg_run(&pr, 1, B, ws, 0, stream, P, first_sparse, &cert, &err);
    if (rc) return rc;
    if (err) { g_fallback[1] = err; rc = CFM_ENOCONV; }
    if (first_sparse && B > 1 && B <= SP_NMAX && (rc == CFM_ENOCONV || !cert)) {
        ++g_fallback[0];
        if (rc == 0) g_fallback[1] = -1;          // uncertified
        rc = asg_run(&pr, 1, B, ws, 0, stream, P, 0, &cert, &err);
        if (rc == 0 && err) { g_fallback[1] = err; rc = CFM_ENOCONV; }
    }
    if (rc == 0 && B > 1 && !cert) rc = CFM_ENOCONV;     // never hand back an uncertified permutation silently
    return rc;
}

extern "C" int cfm_assign_exact_f32(const float* M, int B, int* perm, int* certified,
                                    double* total_cost, int* stats, void* ws, void* stream) {
    const AsgParams P = asg_params_snapshot();
    if (P.small && B >= 2 && B <= SMA_N) {
        // one workgroup, one launch; a solve that hits its round caps or fails its certificate reports it and
        // the chip-wide state machine below takes over
        if (!M || !perm || !ws) return CFM_EINVAL;
        if (((uintptr_t)ws & 15) != 0) return CFM_EALIGN;
        hipStream_t s = (hipStream_t)stream;
        int rc0 = asg_pinned();
        if (rc0) return rc0;
        SmaParams Q;
        Q.theta = P.async_theta;            // (its epsilon phases are asynchronous since round 6: theta 2-3 measured 10 % ahead of 5, tools/asg_small_sweep.py)
        Q.eps0_frac = P.eps0_frac; Q.eps_last_frac = P.eps_last_frac;
        Q.stop_frac = P.stop_frac;
        Q.round_cap = P.round_cap; Q.arr_cap = P.arr_cap > 15 ? P.arr_cap : 15; Q.total_cap = 20000;   // (the one-workgroup solver was tuned with 15)
        Q.reserved = 0;
        int* status = (int*)ws;            // (the kernel clears `certified` and writes every word of the status block itself)
        hipLaunchKernelGGL(asg_small, dim3(1), dim3(SMA_T), 0, s, M, B, Q, perm, certified, total_cost, stats, status);
        rc0 = cfm_status();
        if (rc0) return rc0;
        rc0 = cfm_hip(hipMemcpyAsync(g_pinned, status, 64, hipMemcpyDeviceToHost, s));
        if (rc0) return rc0;
        rc0 = cfm_hip(hipStreamSynchronize(s));
        if (rc0) return rc0;
        for (int q = 0; q < 16; ++q) g_small_last[q] = g_pinned[q];
        if (g_pinned[0] == 1) return 0;
    }
    if (!M || !perm) return CFM_EINVAL;
    const AsgProblem pr = {M, perm, certified, total_cost, stats};
    return asg_solve_one(pr, B, ws, stream, P, P.sparse);
}

// nb problems of the same size in ONE chain of launches: every launch carries all problems (grid.y), so the
// latency-bound chain — ~110 launch boundaries, a one-workgroup list solver — is paid once per batch.  A problem
// whose candidate-list path stops or ends uncertified is redone alone on the dense state machine, like a single solve.
extern "C" size_t cfm_assign_batch_ws_bytes_internal(int n, int nb) {
    if (nb > ASG_BATCH_MAX) nb = ASG_BATCH_MAX;       // longer lists go through in groups of ASG_BATCH_MAX
    return (size_t)nb * cfm_align_up(asg_ws_bytes(n), 256) + 2 * 64 * (size_t)ASG_BATCH_MAX + 256;
}

extern "C" int cfm_assign_exact_batch_f32(const float* const* M, int nb, int B, int* const* perm, int* certified,
                                          double* total_cost, int* stats, void* ws, void* stream) {
    if (!M || !perm || nb < 0) return CFM_EINVAL;
    if (nb == 0) return 0;
    AsgParams P = asg_params_snapshot();
    // The batch entry is the THROUGHPUT form of the solve (couplings prefetched beside a model step): its launches are
    // capped at ASG_TP_WGS workgroups per problem whatever the batch size — also for a batch of one, two or three
    // (the first, small job of a prefetch run; the remainder of a run).  A bid round occupies the chip for as long as
    // its slowest workgroup whatever it does, and every workgroup stages the prices: fewer, fuller workgroups take
    // less of the chip from the other jobs and the dense products.  Measured in the C3 pipelined loop (round 4,
    // CFM_ASG_BLOCKS sweep, same box): 256 / 64 / 32 / 16 per problem for the odd-sized jobs: 1.201 / 1.163 / 1.162 /
    // 1.276 ms per step; a lone solve prefers the wide grid (3.30 vs 3.68 ms sequential): cfm_assign_exact_f32 keeps it.
#define ASG_TP_WGS 64
    if (P.wide_blocks_cap == 0) P.wide_blocks_cap = ASG_TP_WGS;      // (an explicit cfm_assign_set_wide_blocks cap is the caller's: never overridden)
    const size_t stride = cfm_align_up(asg_ws_bytes(B), 256);
    int rc = 0;
    for (int b0 = 0; b0 < nb && rc == 0; b0 += ASG_BATCH_MAX) {
        const int k = nb - b0 < ASG_BATCH_MAX ? nb - b0 : ASG_BATCH_MAX;
        AsgProblem pr[ASG_BATCH_MAX];
        int cert[ASG_BATCH_MAX], err[ASG_BATCH_MAX];
        for (int b = 0; b < k; ++b) {
            if (!M[b0 + b] || !perm[b0 + b]) return CFM_EINVAL;
            pr[b] = {M[b0 + b], perm[b0 + b], certified ? certified + b0 + b : nullptr,
                     total_cost ? total_cost + b0 + b : nullptr, stats ? stats + 8 * (size_t)(b0 + b) : nullptr};
        }
        const bool machine = !(P.small && B >= 2 && B <= SMA_N) && B > 1 && k > 1;
        if (!machine) {               // single problems and the one-workgroup sizes: one after the other
            for (int b = 0; b < k && rc == 0; ++b) {
                if (P.small && B >= 2 && B <= SMA_N || B <= 1)
                    rc = cfm_assign_exact_f32(pr[b].M, B, pr[b].perm, pr[b].certified, pr[b].total_cost, pr[b].stats, ws, stream);
                else
                    rc = asg_solve_one(pr[b], B, ws, stream, P, P.sparse);      // the chip-wide machine on the throughput grid
            }
            continue;
        }
        rc = asg_run(pr, k, B, ws, stride, stream, P, P.sparse, cert, err);
        for (int b = 0; b < k && rc == 0; ++b) {
            if (!err[b] && cert[b]) continue;
            if (!(P.sparse && B <= SP_NMAX)) { rc = CFM_ENOCONV; break; }
            ++g_fallback[0]; g_fallback[1] = err[b] ? err[b] : -1;
            rc = asg_solve_one(pr[b], B, (char*)ws + (size_t)b * stride, stream, P, 0);
        }
    }
    return rc;
}
