// assign_sparse.h — phase C of the exact assignment on candidate lists (n <= 4096).
//
// After the epsilon = 0 rounds the duals are feasible and every kept pair is tight; what
// is left is a handful of free rows, each needing one shortest-augmenting-path search.  The
// searches are sequential and each is ~40 label-correcting batches deep, so the dense form
// (one grid launch per batch) is bound by kernel boundaries, not by bandwidth.  Here:
//
//   wide_build   (whole grid, one wave per row)  keeps, for every row i, the SP_K = 64
//                columns with the smallest c_ik + p_k, their fp32 costs, and a bound T_i with
//                c_ik + p_k >= T_i for every column that was NOT kept.  Prices only rise
//                afterwards, so the bound stays valid for the rest of the solve.
//   sp_solver    (ONE workgroup = the asg_solve kernel, 16 waves, state in LDS: prices, labels, predecessors,
//                owners, the scan list)  runs all searches with workgroup barriers instead
//                of kernel boundaries.  A row is relaxed over its 64 candidates only; when a
//                search has converged every tree row is checked a posteriori:
//                    label(row) + (T_row - u_row) >= dfree
//                proves that none of its dropped edges could have produced a label below
//                dfree, i.e. that the sparse search equals the dense one.  Rows that fail
//                the test are relaxed over their full matrix row and the search resumes.
//
// Exactness therefore never depends on the lists (they only decide how much is read), and
// the fp64 certificate pass over the whole matrix still closes the solve.
#pragma once

#define SP_K 64
#define SP_NMAX 4096
#define SP_NOCOL 0xffffu
#define SP_DENSE 0x8000u      // list entry: relax over the full matrix row
#define SP_ROOT 0x4000u       // list entry: a free root row of the phase (low bits: its root slot, not a column)
#define SP_COLMASK 0x0fffu
#define SP_ROOTS 64           // a phase grows one tree per free row, at most this many (= the hand-off threshold)
#define SP_ROWMASK 0x0fffull  // pkey[k] = (label bit pattern with its low 12 bits cleared) | predecessor row
#define SP_NOKEY (~0ull)      // pkey[k]: column not reached in this phase
#define SP_TNONE 0x7fffffff   // tcol[slot]: the tree accepted no free column
#define SP_FREEROW 0xf000u    // a[i] >= SP_FREEROW: row i is free; during a phase a[i] = SP_FREEROW | its root slot
#define SP_INL_NEAR 1u        // inl[k] bit 0: column is in the near pending list
#define SP_INL_FAR 2u         // inl[k] bit 1: improved to a label above far_thr, to be re-bucketed

#define SP_BUILD_WAVES 8   // waves per workgroup that build lists
#define SP_CAP 64          // list entries per batch (16 waves x 4 entries kept in registers)
#define SP_SEED_HOPS 96    // longest walk to the root when the surviving trees of a phase are seeded into the next one

// dynamic LDS of the two candidate-list kernels: the list builder's strips (one fp32 row per building
// wave) and the solver's state (34 B per column)
static inline size_t sp_build_lds_bytes(int n) {
    if ((n & 1023) == 0 && n > 64) return 64;              // (the fast path of wide_build keeps the strip in registers)
    return (size_t)SP_BUILD_WAVES * (size_t)((n + 63) / 64) * 64 * sizeof(float) + 64;
}
static inline size_t sp_solver_lds_bytes(int n) { return (size_t)((n + 15) & ~15) * 37 + 4096; }

struct SpL {
    double* p;               // prices
    double* dist;            // labels of the current search (>= 0)
    unsigned long long* pkey;   // (label & ~0xfff) | predecessor row: ONE atomic min per improvement keeps the
                                // predecessor of the smallest label (ties within 2^-40 relative: the lowest row)
    unsigned char* slot;     // root slot (tree) of the column's label — a hint for the radius, written without
                             // atomics; the finish walks the predecessors for the true tree
    unsigned short* owner;   // col -> row (SP_NOCOL: free)
    unsigned short* a;       // row -> col (>= SP_FREEROW: free)
    unsigned short* fcol;    // free columns
    unsigned short* pl[2];   // pending columns: improved, assigned, not yet relaxed (ping-pong)
    unsigned char* ddone;    // row of this column already relaxed densely at its current label
    unsigned char* inl;      // SP_INL_NEAR | SP_INL_FAR (updated with 32-bit LDS atomics while a batch runs)
    double* lbase;           // scan list (<= SP_CAP): label of the entry when it was listed
    unsigned short* lcol;    // scan list: column | flags
    double* rd;              // 64 doubles of scratch
    int* ri;                 // 128 ints of scratch
    // multi-source phase: one tree per free row ("root slot" s)
    double* ru;                  // [SP_ROOTS] u of the root row = min_k (c + p)
    unsigned long long* tmin;    // [SP_ROOTS] smallest free-column label of the tree (bit pattern; ~0: none)
    int* tcol;                   // [SP_ROOTS] accepted free column
    unsigned short* rrow;        // [SP_ROOTS] root row of the slot (compacted free-row list)
    unsigned char* rdn;          // [SP_ROOTS] root row already relaxed over its full matrix row
};
// scratch slots
#define SP_RI_NPL 64     // [2] pending-list lengths (atomic append counters), one per list of the ping-pong pair
#define SP_RI_CSELF 72   // [2] entries wanted for the scan list by the barrier-free bookkeeping step (by source list)
#define SP_RD_NMIN 54    // [2] bit patterns: running smallest / largest label filed into each pending list (sp_file and the
#define SP_RD_NMAX 56    //     bookkeeping step's re-filing) — what lets a step start without a reduction over the list
#define SP_RI_NS 71      // entries in the scan list
#define SP_RI_FLAG 66
#define SP_RI_FREECHG 68 // a free column's label was lowered since the radius was last computed
#define SP_RD_DFREE 48   // radius of the phase
#define SP_RD_FAR 49     // labels above this are only flagged (SP_INL_FAR), not listed, until the near list is empty
#define SP_RI_RBAD 120   // 2 ints: mask of the root slots that failed the a-posteriori test
#define SP_RI_NFC 122    // free columns / free rows left after a phase
#define SP_RI_NFR 123
#define SP_RI_ANYD 124   // some root of the phase starts dense
#define SP_RI_SEEDN 126  // seeds of the next phase (columns of the trees that did not augment; list: fcol + SP_ROOTS)

__device__ __forceinline__ SpL sp_carve(char* lds, int n) {
    SpL L; char* q = lds; const size_t N = (size_t)n;   // 37 bytes per column (rounded up to 16 columns) + 4096
    L.rd = (double*)q; q += 64 * 8;
    L.lbase = (double*)q; q += SP_CAP * 8;
    L.ri = (int*)q; q += 128 * 4;
    L.lcol = (unsigned short*)q; q += SP_CAP * 2;
    q += 128;   // 512 + 512 + 512 + 128 + 128 = 1792: keeps the arrays below 16-byte aligned
    L.ru = (double*)q; q += SP_ROOTS * 8;
    L.tmin = (unsigned long long*)q; q += SP_ROOTS * 8;
    L.tcol = (int*)q; q += SP_ROOTS * 4;
    L.rrow = (unsigned short*)q; q += SP_ROOTS * 2;
    L.rdn = (unsigned char*)q; q += SP_ROOTS;          // + 1472 = 3264 (a multiple of 16) of the 4096 reserved
    const size_t N16 = (N + 15) & ~(size_t)15;
    L.p = (double*)q; q += 8 * N16;
    L.dist = (double*)q; q += 8 * N16;
    L.pkey = (unsigned long long*)q; q += 8 * N16;
    L.owner = (unsigned short*)q; q += 2 * N16;
    L.a = (unsigned short*)q; q += 2 * N16;
    L.fcol = (unsigned short*)q; q += 2 * N16;
    L.pl[0] = (unsigned short*)q; q += 2 * N16;
    L.pl[1] = (unsigned short*)q; q += 2 * N16;
    L.ddone = (unsigned char*)q; q += N16;
    L.inl = (unsigned char*)q; q += N16;
    L.slot = (unsigned char*)q;
    return L;
}

// ------------------------------------------------------------ list build -----
__device__ __forceinline__ int sp_count(const float* __restrict__ r, int nt, float tau) {
    int c = 0;
#pragma unroll 8
    for (int t = 0; t < nt; ++t) c += (r[t * 64] < tau) ? 1 : 0;
    return wave_sum_i(c);
}

// One wave per row.  Lane l owns columns l, 64 + l, ...; r[t] = fl32((c + p) - rowmin) >= 0
// lives in the wave's LDS strip (lane-major: conflict free).  A threshold tau with
// count(r < tau) in [32, 64] is found by bisection; members are r < tau, and every
// non-member satisfies (c + p) >= rowmin + tau * (1 - 2^-22) =: T.
__device__ __forceinline__ void wide_build(gfp M, const AsgWs& w, const AsgState* st,
                           char* lds) {
    const int n = st->n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= SP_BUILD_WAVES) return;
    const int nt = (n + 63) / 64;
    float* r = reinterpret_cast<float*>(lds) + (size_t)wv * nt * 64 + lane;   // r[t * 64]
    const int wave_gid = wv * gridDim.x + blockIdx.x, n_waves = gridDim.x * SP_BUILD_WAVES;
    const bool fastb = ((n & 1023) == 0) && n > SP_K;     // n <= SP_NMAX = 4096: at most 16 float4 per lane
    for (int i = wave_gid; i < n; i += n_waves) {
        gfp row = M + (size_t)i * n;
        float lmin = INFINITY;
        double m = INFINITY;
        // fast path (round 6): the strip stays in REGISTERS (64 floats per lane at n = 4096) — the bisection's counts and the
        // compaction were 64 ds_read_b32 per lane and pass.  asg_build 61.5 -> 47.0 us per C3 solve, 229 -> 178 us per batch of
        // four (gpurun_out/r6_build_ab.txt).  202 VGPRs: still 8 waves per CU; forced to 128 (amdgpu_waves_per_eu) it spills
        // 94 registers — measured no further
        float rr[64];
        if (fastb) {
            // whole row in one burst of float4 requests (lane owns columns 256 j + 4 lane + e <-> strip
            // slot t = 4 j + e); the prices are read twice (32 KiB, cache resident), the row once
            const int nj = n >> 8;
            float4 c4[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                c4[j] = asg_ld4(row + 256 * (j < nj ? j : 0) + 4 * lane);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nj) {
                    const double2 pa = *reinterpret_cast<const double2*>(w.p + 256 * j + 4 * lane);
                    const double2 pb = *reinterpret_cast<const double2*>(w.p + 256 * j + 4 * lane + 2);
                    m = fmin(m, fmin(fmin((double)c4[j].x + pa.x, (double)c4[j].y + pa.y),
                                     fmin((double)c4[j].z + pb.x, (double)c4[j].w + pb.y)));
                }
            }
            m = wave_min_d(m);
            asm volatile("" ::: "memory");       // (the prices are READ AGAIN below, not kept: 128 registers otherwise, and the strip spills)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < nj) {
                    const double2 pa = *reinterpret_cast<const double2*>(w.p + 256 * j + 4 * lane);
                    const double2 pb = *reinterpret_cast<const double2*>(w.p + 256 * j + 4 * lane + 2);
                    const float x0 = (float)(((double)c4[j].x + pa.x) - m), x1 = (float)(((double)c4[j].y + pa.y) - m);
                    const float x2 = (float)(((double)c4[j].z + pb.x) - m), x3 = (float)(((double)c4[j].w + pb.y) - m);
                    rr[4 * j + 0] = x0; rr[4 * j + 1] = x1; rr[4 * j + 2] = x2; rr[4 * j + 3] = x3;
                    lmin = fminf(lmin, fminf(fminf(x0, x1), fminf(x2, x3)));
                } else {
                    rr[4 * j + 0] = INFINITY; rr[4 * j + 1] = INFINITY; rr[4 * j + 2] = INFINITY; rr[4 * j + 3] = INFINITY;
                }
            }
        } else {
            for (int t0 = 0; t0 < nt; t0 += 8) {
                float c[8]; double pk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = (t0 + q) * 64 + lane;
                    c[q] = (k < n) ? row[k] : INFINITY; pk[q] = (k < n) ? w.p[k] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) m = fmin(m, (double)c[q] + pk[q]);
            }
            m = wave_min_d(m);
            for (int t0 = 0; t0 < nt; t0 += 8) {
                float c[8]; double pk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = (t0 + q) * 64 + lane;
                    c[q] = (k < n) ? row[k] : INFINITY; pk[q] = (k < n) ? w.p[k] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = (t0 + q) * 64 + lane;
                    if (t0 + q < nt) {
                        const float x = (k < n) ? (float)(((double)c[q] + pk[q]) - m) : INFINITY;
                        r[(t0 + q) * 64] = x; lmin = fminf(lmin, x);
                    }
                }
            }
        }
        float tau;
        auto count = [&](float tq) -> int {
            if (!fastb) return sp_count(r, nt, tq);
            int c = 0;
#pragma unroll
            for (int t = 0; t < 64; ++t) c += (rr[t] < tq) ? 1 : 0;          // (slots beyond the row hold +inf)
            return wave_sum_i(c);
        };
        if (n <= SP_K) {
            tau = INFINITY;
        } else {
            const float hi = wave_max_f(lmin);                 // count(r <= hi) >= 64
            const float t1 = __uint_as_float(__float_as_uint(hi) + 1u);
            if (count(t1) <= SP_K) {
                tau = t1;
            } else {
                float lo = 0.f, hh = t1; int clo = 0;
                for (int it = 0; it < 48 && clo < SP_K / 2; ++it) {
                    const float mid = 0.5f * (lo + hh);
                    if (!(mid > lo && mid < hh)) break;
                    const int cm = count(mid);
                    if (cm <= SP_K) { lo = mid; clo = cm; } else hh = mid;
                }
                tau = lo;
            }
        }
        const bool all = (tau == INFINITY);
        int off = 0;
        if (fastb) {
            int ln = lane; asm volatile("" : "+v"(ln));     // (opaque: the 64 column indices are formed per row — hoisted out of the row loop they spill)
            const unsigned long long below = (1ull << ln) - 1ull;
            gfp rowl = row + 4 * ln;
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                const int k = (t >> 2) * 256 + 4 * ln + (t & 3);
                const bool mem = rr[t] < tau;                                    // (fastb: n > SP_K, tau finite; slots beyond the row: +inf)
                const unsigned long long mask = __ballot(mem);
                if (mask) {
                    if (mem) {
                        const int pos = off + __popcll(mask & below);
                        w.cl[(size_t)i * SP_K + pos] = make_uint2((unsigned)k, __float_as_uint(rowl[(t >> 2) * 256 + (t & 3)]));
                    }
                    off += __popcll(mask);
                }
            }
        } else
        for (int t = 0; t < nt; ++t) {
            const int k = fastb ? ((t >> 2) * 256 + 4 * lane + (t & 3)) : (t * 64 + lane);
            const bool mem = (k < n) && (all || r[t * 64] < tau);
            const unsigned long long mask = __ballot(mem);
            if (mask) {
                if (mem) {
                    const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
                    w.cl[(size_t)i * SP_K + pos] = make_uint2((unsigned)k, __float_as_uint(row[k]));
                }
                off += __popcll(mask);
            }
        }
        if (lane >= off) w.cl[(size_t)i * SP_K + lane] = make_uint2(SP_NOCOL, 0x7f800000u);
        if (lane == 0) w.cT[i] = all ? INFINITY : (m + (double)tau * (1.0 - 2.4e-7) - 1e-290);
    }
}

// --------------------------------------------------------------- solver ------
// One workgroup is instruction-issue bound (16 waves share 4 SIMDs), so the batch loop keeps
// whole-workgroup work to the relaxations themselves; list bookkeeping runs in ONE wave on an
// explicit pending list (no O(n) scans, no block-wide reductions) while the others wait.

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which
// would make every barrier wait for the list prefetches issued for the NEXT batch; inside the
// solver loop all cross-wave traffic is LDS (M, the candidate lists and cT are read-only).
__device__ __forceinline__ void sp_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// wave64 DPP primitives (row_shr within 16-lane rows, then row_bcast 15 / 31)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int sp_dpp_i(int oldv, int v) {
    return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double sp_dpp_d(double oldv, double v) {
    const int lo = sp_dpp_i<CTRL, ROWMASK>(__double2loint(oldv), __double2loint(v));
    const int hi = sp_dpp_i<CTRL, ROWMASK>(__double2hiint(oldv), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the wave
__device__ __forceinline__ int sp_wave_scan(int v) {
    v += sp_dpp_i<0x111, 0xf>(0, v);
    v += sp_dpp_i<0x112, 0xf>(0, v);
    v += sp_dpp_i<0x114, 0xf>(0, v);
    v += sp_dpp_i<0x118, 0xf>(0, v);
    v += sp_dpp_i<0x142, 0xa>(0, v);
    v += sp_dpp_i<0x143, 0xc>(0, v);
    return v;
}
// min over the wave, result uniform
__device__ __forceinline__ double sp_wave_min(double v) {
    v = fmin(v, sp_dpp_d<0x111, 0xf>(INFINITY, v));
    v = fmin(v, sp_dpp_d<0x112, 0xf>(INFINITY, v));
    v = fmin(v, sp_dpp_d<0x114, 0xf>(INFINITY, v));
    v = fmin(v, sp_dpp_d<0x118, 0xf>(INFINITY, v));
    v = fmin(v, sp_dpp_d<0x142, 0xa>(INFINITY, v));
    v = fmin(v, sp_dpp_d<0x143, 0xc>(INFINITY, v));
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sp_wave_max(double v) { return -sp_wave_min(-v); }
__device__ __forceinline__ int sp_wave_total(int v) {
    return __builtin_amdgcn_readlane(sp_wave_scan(v), 63);
}

#define SP_T 1024
#define SP_NW (SP_T / 64)
#define SP_IPT 4   // columns per thread in the O(n) passes (n <= 4096)
#define SP_E (SP_CAP / SP_NW)   // list entries per wave, kept in registers by the fast batch

__device__ __forceinline__ double sp_block_min(double v, const SpL& L) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = sp_wave_min(v);
    if (lane == 0) L.rd[wv] = v;
    sp_sync();
    const double r = sp_wave_min(L.rd[lane & (SP_NW - 1)]);
    sp_sync();
    return r;
}

// Candidate update, ONE phase.  A candidate below the column's label lowers it with an LDS atomic min on the bit
// pattern (labels are >= +0), and its predecessor key — the same bits with the low 12 replaced by the row — goes
// into pkey[k] with a second atomic min: the key that survives belongs to the smallest label, so label and
// predecessor stay consistent without a second pass over the entries (two candidates closer than 2^-40 relative
// tie on the key: the lower row is kept, its path is longer than the label by less than that).  Only strict
// improvements write, so a chain of predecessors never closes a cycle.  The lane that lowered the label files an
// assigned column in the pending queue.
__device__ __forceinline__ void sp_file(const SpL& L, int k, double label, int plcur, double far_thr) {
    // (test-and-set on the column's inl byte: several lanes may lower one column in a batch, one files it)
    unsigned* w = reinterpret_cast<unsigned*>(L.inl) + (k >> 2);
    const int sh = 8 * (k & 3);
    if (label <= far_thr) {
        const unsigned old = atomicOr(w, SP_INL_NEAR << sh);
        if (!((old >> sh) & SP_INL_NEAR)) {
            const int pos = atomicAdd(&L.ri[SP_RI_NPL + plcur], 1);
            (plcur ? L.pl[1] : L.pl[0])[pos] = (unsigned short)k;
            // the list's running minimum (the next bookkeeping step starts from it without a pass over the list): ONE
            // fire-and-forget LDS atomic.  (Measured, profiles/r6_experiments.txt 9: a read-compare-update pair per bound
            // + 700 cycles per batch; a second same-address atomic for the maximum + 1 700; a per-lane range folded by
            // the wave + 600.  The running maximum only feeds the window heuristic: the re-filing keeps it.)
            atomicMin(reinterpret_cast<unsigned long long*>(L.rd) + SP_RD_NMIN + plcur, (unsigned long long)__double_as_longlong(label));
        }
    } else {
        atomicOr(w, SP_INL_FAR << sh);
    }
}
__device__ __forceinline__ void sp_improve(const SpL& L, int k, double cand, double cur, double dfree, unsigned row,
                                           unsigned slot, int plcur, double far_thr) {
    if (cand < cur && cand < dfree) {                 // labels >= the radius can never matter
        const unsigned long long nb = (unsigned long long)__double_as_longlong(cand);
        const unsigned long long old = atomicMin((unsigned long long*)&L.dist[k], nb);
        atomicMin(&L.pkey[k], (nb & ~SP_ROWMASK) | (unsigned long long)row);
        if (old > nb) {
            L.ddone[k] = 0; L.slot[k] = (unsigned char)slot;
            if (L.owner[k] != SP_NOCOL) sp_file(L, k, cand, plcur, far_thr);
            else L.ri[SP_RI_FREECHG] = 1;
        }
    }
}
__device__ __forceinline__ double sp_cand(double pk, float c, double rj, double base) {
    const double rc = fmax(((double)c + pk) - rj, 0.0);   // dual feasible up to rounding
    return base + rc;
}

// generic entry (dense entries allowed).  A root entry carries its root slot, a column entry inherits the slot of
// the tree that gave the column its label.
__device__ __forceinline__ void sp_entry(gfp M, const AsgWs& w, const SpL& L, int n,
                                         unsigned e, double base, int lane, int plcur,
                                         double dfree, double far_thr) {
    const bool dense = (e & SP_DENSE) != 0, root = (e & SP_ROOT) != 0;
    const int j = root ? -1 : (int)(e & SP_COLMASK);
    const unsigned slot = root ? (e & 63u) : (unsigned)L.slot[j];
    const int i = root ? (int)L.rrow[slot] : (int)L.owner[j];
    double rj = root ? L.ru[slot] : 0.0;
    if (!dense) {
        const uint2 cl = w.cl[(size_t)i * SP_K + lane];
        const unsigned col = cl.x; const float c = __uint_as_float(cl.y);
        const bool valid = col != SP_NOCOL;
        if (!root) {
            const unsigned long long hit = __ballot(valid && (int)col == j);
            float cij;
            if (hit) cij = __shfl(c, __ffsll((long long)hit) - 1, 64);
            else cij = M[(size_t)i * n + j];
            rj = (double)cij + L.p[j];                        // = u_i: the matched edge is tight
        }
        if (valid && (int)col != j)
            sp_improve(L, (int)col, sp_cand(L.p[col], c, rj, base), L.dist[col], dfree, (unsigned)i, slot, plcur, far_thr);
    } else {
        gfp row = M + (size_t)i * n;
        if (!root) rj = (double)row[j] + L.p[j];
        for (int k0 = 0; k0 < n; k0 += 64 * 8) {
            float c[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int k = k0 + q * 64 + lane; c[q] = (k < n) ? row[k] : 0.f; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q * 64 + lane;
                if (k < n && k != j)
                    sp_improve(L, k, sp_cand(L.p[k], c[q], rj, base), L.dist[k], dfree, (unsigned)i, slot, plcur, far_thr);
            }
        }
    }
}

// Radius of the phase (one wave; at most SP_ROOTS free columns, lane <-> free column).  Every tree that has reached
// a free column will accept its nearest one, so nothing at or above
//     R = max over those trees of (smallest free-column label of the tree)
// can matter any more.  The radius only ever shrinks (min with the current one): an entry skipped once is never
// needed later, whatever happens to the trees afterwards — the finish accepts labels <= the final radius only, and
// every label below it is final.  The tree of a column is read from its slot hint; a stale hint can only make the
// radius smaller than intended (it stays >= the smallest free-column label), which costs augmentations of this
// phase, never correctness.
// (Round 6 measured the alternative — establish the radius only once 30 / 50 / 75 / 100 % of the phase's trees have reached
//  a free column: fewer phases (4.3 -> 3.6) but deeper ones; 30 - 50 %: the same solver time within noise, 75 %+: 7 - 15 ms.
//  profiles/r6_sched_sweep.txt.  The first tree to arrive sets it.  Also measured and removed: pulling the 2 MiB of
//  candidate lists into the solver's own L2 before the searches start (the build kernel wrote them from every XCD): the
//  time per batch did not move (3.9 us with and without); batches of up to 128 entries (two passes of the fast batch in
//  front of one bookkeeping step): the SAME number of batches (269) — their count is the label depth of the searches,
//  not the batch capacity — 102 k instead of 97.5 k row evaluations, solver 1.23 instead of 1.05 ms.)
__device__ __forceinline__ double sp_radius(const SpL& L, int nFC, int lane, double dfree) {
    L.tmin[lane] = ~0ull;
    double d = INFINITY; unsigned sl = 0;
    if (lane < nFC) { const int k = L.fcol[lane]; d = L.dist[k]; sl = L.slot[k] & 63u; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reset before the minima (same wave: LDS is in order)
    if (d < INFINITY) atomicMin(&L.tmin[sl], (unsigned long long)__double_as_longlong(d));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tm = L.tmin[lane];
    const double mine = (tm == ~0ull) ? -INFINITY : __longlong_as_double((long long)tm);
    double R = sp_wave_max(mine);
    if (!(R > -INFINITY)) R = INFINITY;
    return fmin(dfree, R);
}

#ifdef SP_PROFILE
#define FB_TICK(slot) do { if (threadIdx.x == 0) { const long long t_ = clock64(); fb[slot] += t_ - tl; tl = t_; } } while (0)
#else
#define FB_TICK(slot) do { } while (0)
#endif

__device__ __forceinline__ double sp_rfl_d(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// One batch of <= SP_CAP sparse entries: ONE phase, ONE barrier.  Written in stages over the (at most SP_E)
// entries of this wave so that the LDS / global round trips of different entries overlap; everything that is
// uniform over the wave (entry, row, tree, base label, the lane holding the matched edge) is moved to scalar
// registers so the control flow is scalar.  The solver is instruction-issue bound (16 waves share 4 SIMDs): what
// counts is the number of instructions per entry.  The last wave refreshes the radius after the barrier, while the
// others start the bookkeeping (it is published for the NEXT batch: a stale, larger radius is always valid).
__device__ __forceinline__ void sp_fast_batch(gfp M, const AsgWs& w, const SpL& L,
                                              int n, int nS, int nFC, double dfree,
                                              int lane, int wv, int plcur, double far_thr, long long* fb) {
#ifdef SP_PROFILE
    long long tl = clock64();
#endif
    const int swv = __builtin_amdgcn_readfirstlane(wv);
    int nq = 0;
    if (nS > swv) { nq = (nS - swv + SP_NW - 1) / SP_NW; if (nq > SP_E) nq = SP_E; }
    int jj[SP_E], ii[SP_E], kk[SP_E]; bool root[SP_E], on[SP_E], use[SP_E];
    unsigned slot[SP_E];
    double bs[SP_E], cd[SP_E], pj[SP_E], pk[SP_E], dc[SP_E];
    uint2 cl[SP_E];
    // stage 1: entries (uniform LDS reads)
    unsigned ev[SP_E];
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        ev[q] = SP_ROOT; bs[q] = INFINITY;
        if (q < nq) { ev[q] = L.lcol[swv + SP_NW * q]; bs[q] = L.lbase[swv + SP_NW * q]; }
    }
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)ev[q]);
        ev[q] = e;
        bs[q] = sp_rfl_d(bs[q]);
        root[q] = (e & SP_ROOT) != 0;
        jj[q] = root[q] ? -1 : (int)(e & SP_COLMASK);
        on[q] = (q < nq) && bs[q] < dfree;
    }
    // stage 2: rows and trees (a root entry: its slot's row; a column entry: the owner and the slot of its label)
    unsigned ov[SP_E], sv[SP_E];
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        ov[q] = 0u; sv[q] = ev[q] & 63u;
        if (on[q]) {
            if (root[q]) ov[q] = L.rrow[ev[q] & 63u];
            else { ov[q] = L.owner[jj[q]]; sv[q] = L.slot[jj[q]]; }
        }
    }
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        ii[q] = __builtin_amdgcn_readfirstlane((int)ov[q]);
        slot[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)sv[q]) & 63u;
    }
    // stage 3: candidate lists (one 8-byte load per lane and entry, all in flight together)
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        cl[q] = make_uint2(SP_NOCOL, 0u);
        if (on[q]) cl[q] = w.cl[(size_t)ii[q] * SP_K + lane];
    }
    FB_TICK(0);
    // stages 4 - 6, one self-contained block per entry slot (no defaults carried across slots: the staged form below
    // spends a third of its instructions on moves / selects for slots that are off — and this kernel is issue bound;
    // the LDS round trips of a slot are covered by the other three waves of its SIMD)
#pragma unroll
    for (int q = 0; q < SP_E; ++q) {
        if (on[q]) {                                             // (wave uniform)
            double rj;
            if (root[q]) rj = sp_rfl_d(L.ru[slot[q]]);
            else {
                const unsigned long long hit = __ballot((int)cl[q].x == jj[q]);
                float cij;
                if (hit) cij = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)cl[q].y, __ffsll((long long)hit) - 1));
                else cij = M[(size_t)ii[q] * n + jj[q]];
                rj = (double)cij + sp_rfl_d(L.p[jj[q]]);         // = u_i: the matched edge is tight
            }
            const unsigned col = cl[q].x;
            if (col != SP_NOCOL && (int)col != jj[q]) {
                const int k = (int)col;
                const double cand = sp_cand(L.p[k], __uint_as_float(cl[q].y), rj, bs[q]);
                if (cand < L.dist[k] && cand < dfree) {          // labels >= the radius can never matter
                    const unsigned long long nb = (unsigned long long)__double_as_longlong(cand);
                    const unsigned long long old = atomicMin((unsigned long long*)&L.dist[k], nb);
                    atomicMin(&L.pkey[k], (nb & ~SP_ROWMASK) | (unsigned long long)(unsigned)ii[q]);
                    if (old > nb) {
                        L.ddone[k] = 0; L.slot[k] = (unsigned char)slot[q];
                        if (L.owner[k] != SP_NOCOL) sp_file(L, k, cand, plcur, far_thr);
                        else L.ri[SP_RI_FREECHG] = 1;           // only then can the radius move
                    }
                }
            }
        }
    }
    FB_TICK(1);
    sp_sync();
    FB_TICK(2);
    if (swv == SP_NW - 1 && L.ri[SP_RI_FREECHG]) {              // (free-column labels change a few times per phase)
        const double dnew = sp_radius(L, nFC, lane, dfree);
        if (lane == 0) { L.rd[SP_RD_DFREE] = dnew; L.ri[SP_RI_FREECHG] = 0; }
    }
    FB_TICK(3);
}

// Bookkeeping after a batch on the NEAR pending list (labels <= far_thr; columns improved to a label above far_thr
// are only flagged SP_INL_FAR and found again by the re-bucketing when the near list runs dry: a two-level bucket
// queue, so a batch never pays for the whole frontier).  Columns at or above dfree are dropped; of the rest the ones
// within `delta` of the smallest label (at most SP_CAP) become the next scan list — the window keeps the scan order
// close to the label order, which is what keeps the number of relaxations (and of batches) down: a FIFO bucket queue
// (64 buckets per epoch, one append per filing, no pass over the list) was built and measured in round 3 — its
// batches were 25 % cheaper but it needed 1.6 x as many (18 - 22 k scans against 11.5 k), 2.3 - 2.7 ms against 2.0.
// Publishes nS, the new list length and far_thr; every thread returns the same adapted delta.
// The solver is instruction-issue bound, so only as many waves as the list needs take part (thread <-> entry, up to
// SP_IPT entries per thread for lists beyond 1024): the others go straight to the barriers.
#define SP_FARMULT 3.0   // the near list holds the labels within this many windows of the smallest one (measured: 3 -> 1.82 ms, 5 -> 1.96, 8 -> 1.92 of solver time at C3 in round 3; round 6, on the barrier-free step: 1 / 1.5 / 2 / 3 / 4 -> 0.94-1.01 ms, all within run-to-run noise)
#define SP_RI_WANT 16    // 16 per-wave selection counts
#define SP_RI_KEEP 96    // 16 per-wave keep counts
// (round 3, after the cycle counters: the 16 waves no longer reduce the partial minima / counts redundantly and there
//  are no prefix scans over the waves — a wave adds its partials to three LDS words with atomics, and takes its block of
//  the scan list / the kept list with ONE returning atomic add each; the order of the lists is arbitrary anyway.  Two
//  barriers instead of three, and the kept-list counter IS the append counter of the next pending list.)
#define SP_TLO (SP_CAP / 2)   // the window doubles while a batch holds fewer entries than this
#define SP_RD_CMIN 52    // bit patterns (labels >= +0 order like integers): smallest / largest live pending label
#define SP_RD_CMAX 53
#define SP_RI_CNP 69     // live pending entries
#define SP_RI_CSEL 70    // entries wanted for the scan list (may exceed SP_CAP)
struct SpRange { double dmin, dmax; int npend; };
// FAST = false: the exact step — the label range of the live entries by a reduction over the list (two barriers inside).
// FAST = true (round 6): NO barrier inside.  The range comes from the list's running minimum / maximum (every filing and
// every re-filing keeps them: sp_file, below), the counters of the list being built were reset by the solver loop in
// front of the batch.  The running minimum can be STALE-LOW (an entry that set it was lowered out of the list's range or
// died under a smaller radius): the window then may select nothing although entries are pending — the loop falls back
// to the exact step for that one batch.  Stale-high cannot happen for a pending entry: a label only enters the list
// through sp_file / the re-filing, both of which record it.
template <int NSL, bool FAST>
__device__ __forceinline__ double sp_collect_n(const SpL& L, int plcur, double dfree, double delta,
                                               double far_thr, SpRange* rng) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int npl = L.ri[SP_RI_NPL + plcur];
    const int nwav = npl >= SP_T ? SP_NW : (npl + 63) / 64;        // waves that hold entries (uniform)
    const bool active = wv < nwav;
    const unsigned short* src = plcur ? L.pl[1] : L.pl[0];
    unsigned short* dst = plcur ? L.pl[0] : L.pl[1];
    unsigned long long* mm = reinterpret_cast<unsigned long long*>(L.rd);
    int kk[NSL]; double dd[NSL]; bool have[NSL], live[NSL];
    double lmin = INFINITY, lmax = 0.0; int np = 0;
    double dmin, dmax; int npend;
    if (FAST) {
        dmin = __longlong_as_double((long long)mm[SP_RD_NMIN + plcur]); dmax = fmax(dmin, __longlong_as_double((long long)mm[SP_RD_NMAX + plcur]));
        npend = npl;
    } else if (tid == 0) {
        L.rd[SP_RD_CMIN] = INFINITY; L.rd[SP_RD_CMAX] = 0.0; L.ri[SP_RI_CNP] = 0; L.ri[SP_RI_CSEL] = 0;
    }
#pragma unroll
    for (int e = 0; e < NSL; ++e) {
        have[e] = false; kk[e] = 0; dd[e] = INFINITY; live[e] = false;
        if (active) {
            const int t = e * SP_T + tid;
            have[e] = t < npl;
            if (have[e]) {
                kk[e] = src[t];
                dd[e] = L.dist[kk[e]];
                live[e] = dd[e] < dfree;
                if (!FAST && live[e]) { lmin = fmin(lmin, dd[e]); lmax = fmax(lmax, dd[e]); ++np; }
            }
        }
    }
    if (!FAST) {
        double wmin = INFINITY, wmax = 0.0; int wnp = 0;
        if (active) { wmin = sp_wave_min(lmin); wmax = sp_wave_max(lmax); wnp = sp_wave_total(np); }
        sp_sync();                                      // everybody has read the list length; the words above are reset
        if (active && lane == 0 && wnp > 0) {
            atomicMin(reinterpret_cast<unsigned long long*>(&L.rd[SP_RD_CMIN]), (unsigned long long)__double_as_longlong(wmin));
            atomicMax(reinterpret_cast<unsigned long long*>(&L.rd[SP_RD_CMAX]), (unsigned long long)__double_as_longlong(wmax));
            atomicAdd(&L.ri[SP_RI_CNP], wnp);
        }
        if (tid == 0) {                                 // the list being built: its counter and its running range
            L.ri[SP_RI_NPL + (plcur ^ 1)] = 0; mm[SP_RD_NMIN + (plcur ^ 1)] = 0x7ff0000000000000ull; mm[SP_RD_NMAX + (plcur ^ 1)] = 0ull;
        }
        sp_sync();
        dmin = L.rd[SP_RD_CMIN]; dmax = L.rd[SP_RD_CMAX]; npend = L.ri[SP_RI_CNP];
    }
    if (!(far_thr < INFINITY) && delta < INFINITY && npend > 2 * SP_CAP) far_thr = dmin + SP_FARMULT * delta;
    const double tau = dmin + delta;
    int* csel = FAST ? &L.ri[SP_RI_CSELF + plcur] : &L.ri[SP_RI_CSEL];
    if (active) {
        bool want[NSL]; int wpos[NSL]; int wtot = 0;
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            want[e] = live[e] && dd[e] <= tau;
            const unsigned long long mw = __ballot(want[e]);
            wpos[e] = wtot + __popcll(mw & ((1ull << lane) - 1ull));
            wtot += __popcll(mw);
        }
        int woff = 0;
        if (wtot > 0) {
            if (lane == 0) woff = atomicAdd(csel, wtot);
            woff = __builtin_amdgcn_readfirstlane(woff);
        }
        bool keep[NSL]; int kpos[NSL]; int ktot = 0;
        double kmin = INFINITY, kmax = 0.0;
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            const bool sel = want[e] && (woff + wpos[e]) < SP_CAP;      // the wants past the cap stay pending
            keep[e] = live[e] && !sel && dd[e] <= far_thr;
            want[e] = sel;
            const unsigned long long mk = __ballot(keep[e]);
            kpos[e] = ktot + __popcll(mk & ((1ull << lane) - 1ull));
            ktot += __popcll(mk);
            if (keep[e]) { kmin = fmin(kmin, dd[e]); kmax = fmax(kmax, dd[e]); }
        }
        int koff = 0;
        if (ktot > 0) {
            // the kept entries are re-filed: their labels go into the running range of the list being built
            const double wkmin = sp_wave_min(kmin), wkmax = sp_wave_max(kmax);
            if (lane == 0) {
                koff = atomicAdd(&L.ri[SP_RI_NPL + (plcur ^ 1)], ktot);
                atomicMin(&mm[SP_RD_NMIN + (plcur ^ 1)], (unsigned long long)__double_as_longlong(wkmin));
                atomicMax(&mm[SP_RD_NMAX + (plcur ^ 1)], (unsigned long long)__double_as_longlong(wkmax));
            }
            koff = __builtin_amdgcn_readfirstlane(koff);
        }
#pragma unroll
        for (int e = 0; e < NSL; ++e) {
            if (!have[e]) continue;
            const int k = kk[e];
            if (want[e]) {
                const int spos = woff + wpos[e];
                L.lcol[spos] = (unsigned short)k; L.lbase[spos] = dd[e]; L.inl[k] = 0;
            } else if (keep[e]) {
                dst[koff + kpos[e]] = (unsigned short)k;
            } else {
                L.inl[k] = live[e] ? (unsigned char)SP_INL_FAR : (unsigned char)0;
            }
        }
    }
    if (tid == 0) {
        L.rd[SP_RD_FAR] = far_thr;
#ifdef SP_PROFILE
        L.ri[125] += npl;
#endif
    }
    rng->dmin = dmin; rng->dmax = dmax; rng->npend = npend;
    // (the caller's barrier follows; nS = min(selected, SP_CAP) and the window adaptation are read from LDS after it)
    return delta;
}
// after the caller's barrier: the scan-list length and the adapted window (every thread computes the same)
__device__ __forceinline__ double sp_collect_result(const SpL& L, double delta, int* nS, int nsel, const SpRange& r) {
    *nS = nsel < SP_CAP ? nsel : SP_CAP;
    // adapt the window: aim at 32 .. 64 entries per batch
    // (a list of equal labels — the seeds of a phase, all 0 — says nothing about the window: it keeps its width)
    if (nsel > SP_CAP) delta = (r.dmax > r.dmin) ? 0.5 * fmin(delta, r.dmax - r.dmin) : delta;
    else if (nsel < SP_TLO && r.npend > nsel) delta = fmax(2.0 * delta, (r.dmax - r.dmin) * (1.0 / 64.0));
    return delta;
}
// (the near list rarely exceeds 1024 entries: one entry per thread then, without the code of the other three slots)
template <bool FAST>
__device__ __forceinline__ double sp_collect_all(const SpL& L, int plcur, double dfree, double delta, double far_thr, SpRange* rng) {
    if (L.ri[SP_RI_NPL + plcur] <= SP_T) return sp_collect_n<1, FAST>(L, plcur, dfree, delta, far_thr, rng);
    return sp_collect_n<SP_IPT, FAST>(L, plcur, dfree, delta, far_thr, rng);
}

#ifdef SP_PROFILE
#define SP_TICK(slot) do { if (tid == 0) { const long long t_ = clock64(); dbg[slot] += t_ - tlast; tlast = t_; } } while (0)
#else
#define SP_TICK(slot) do { } while (0)
#endif

// The searches, as MULTI-SOURCE phases (a shortest-path forest, one tree per free row, grown by the batches below;
// a column belongs to the tree that gave it its label).  When the forest has converged below the radius every tree
// that reached a free column accepts its nearest one (the trees are vertex disjoint, so all of these paths are
// augmented), the duals move by D = the largest accepted label:
//     p_k += D - d_k  for every column with d_k < D   (u follows through the tight matched edges)
// which keeps them feasible and makes every accepted path tight.  One phase costs the depth of one search and
// retires several rows: at n = 4096 the 25 - 30 rows left by the auction go in about five phases / 280 batches,
// against 65 chip-wide relax rounds + 370 batches when the forest ran on the dense matrix and only the last six
// rows came here one by one (a round-3 host prototype).
__device__ __forceinline__ void sp_solver(gfp M, const AsgWs& w, AsgState* st, char* lds) {
#ifdef SP_PROFILE
    long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long fbv[6] = {0, 0, 0, 0, 0, 0};
    long long* fb = fbv;
    long long tlast = clock64();
    int nfast = 0;
#else
    long long* fb = nullptr;
#endif
    const int n = st->n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SpL L = sp_carve(lds, n);
    int nFC = st->nFC, nFree = st->nF;
    int err = (nFree > SP_ROOTS || nFC != nFree) ? 9 : 0;
    for (int k = tid; k < n; k += SP_T) {
        L.p[k] = w.p[k];
        const int o = w.owner[k]; L.owner[k] = (o < 0) ? (unsigned short)SP_NOCOL : (unsigned short)o;
        const int ak = w.a[k]; L.a[k] = (ak < 0) ? (unsigned short)SP_NOCOL : (unsigned short)ak;
        L.pl[1][k] = (unsigned short)SP_NOCOL;          // (no seeds in front of the first phase, see SP_SEED_HOPS)
    }
    if (!err) {
        for (int t = tid; t < nFC; t += SP_T) L.fcol[t] = (unsigned short)w.listFC[t];
        for (int t = tid; t < nFree; t += SP_T) L.rrow[t] = (unsigned short)w.listF[t];
    }
    if (tid == 0) { L.ri[126] = 0; L.ri[125] = 0; L.ri[SP_RI_FLAG] = 0; }
    sp_sync();

    int batches = 0, scans = 0, dense_scans = 0, phases = 0;
    double delta_prev = INFINITY;          // the label window the previous phase ended with
    while (nFree > 0 && !err) {
        const int nR = nFree;                 // every free row is a root of this phase (slot s <-> rrow[s])
        ++phases;
        for (int k = tid; k < n; k += SP_T) {
            if (L.pl[1][k] != (unsigned short)SP_NOCOL) {            // a seed (below): label 0, its predecessor stays
                L.dist[k] = 0.0; L.pkey[k] &= SP_ROWMASK; L.inl[k] = 0;
            } else {
                L.dist[k] = INFINITY; L.pkey[k] = SP_NOKEY; L.inl[k] = 0;
            }
            L.ddone[k] = 0; L.slot[k] = 0;
        }
        // roots: u_r = min_k (c_rk + p_k), one wave per root.  The candidate minimum is the row minimum iff it does
        // not exceed the bound T_r of the dropped columns; otherwise take it over the full row and start the root
        // with a dense entry.
        if (tid == 0) L.ri[SP_RI_ANYD] = 0;
        sp_sync();
        for (int s0 = wv; s0 < nR; s0 += SP_NW) {
            const int i0 = L.rrow[s0];
            const uint2 cl = w.cl[(size_t)i0 * SP_K + lane];
            const double val = (cl.x != SP_NOCOL) ? (double)__uint_as_float(cl.y) + L.p[cl.x] : INFINITY;
            double u0 = sp_wave_min(val);
            const bool rdense = !(u0 <= w.cT[i0]);
            if (rdense) {
                double mm = INFINITY;
                for (int k = lane; k < n; k += 64) mm = fmin(mm, (double)M[(size_t)i0 * n + k] + L.p[k]);
                u0 = sp_wave_min(mm);
            }
            if (lane == 0) {
                L.a[i0] = (unsigned short)(SP_FREEROW | (unsigned)s0);      // free, root slot s0 of this phase
                L.ru[s0] = u0; L.rdn[s0] = rdense ? 1 : 0;
                L.lcol[s0] = (unsigned short)(SP_ROOT | (rdense ? SP_DENSE : 0u) | (unsigned)s0); L.lbase[s0] = 0.0;
                if (rdense) L.ri[SP_RI_ANYD] = 1;
            }
        }
        if (tid == 0) {
            L.ri[SP_RI_NPL] = 0; L.ri[SP_RI_NPL + 1] = 0; L.rd[SP_RD_DFREE] = INFINITY; L.rd[SP_RD_FAR] = INFINITY; L.ri[SP_RI_FREECHG] = 0;
            unsigned long long* mm = reinterpret_cast<unsigned long long*>(L.rd);
            mm[SP_RD_NMIN] = mm[SP_RD_NMIN + 1] = 0x7ff0000000000000ull; mm[SP_RD_NMAX] = mm[SP_RD_NMAX + 1] = 0ull;
            L.ri[SP_RI_CSELF] = 0; L.ri[SP_RI_CSELF + 1] = 0;
        }
        sp_sync();
        // the seeds take the slot of their root in THIS phase (the roots were renumbered above)
        const int nseed = (L.ri[SP_RI_SEEDN] < n - SP_ROOTS) ? L.ri[SP_RI_SEEDN] : (n > SP_ROOTS ? n - SP_ROOTS : 0);
        const unsigned short* seedlist = L.fcol + SP_ROOTS;
        for (int t = tid; t < nseed; t += SP_T) {
            const int k = seedlist[t];
            L.slot[k] = (unsigned char)(L.a[L.pl[1][k]] & 63u);
        }
        sp_sync();
        int nS = nR, plcur = 0;
        bool any_dense = L.ri[SP_RI_ANYD] != 0;
        double dfree = INFINITY;      // radius of the phase
        // label window of a batch above the smallest pending label.  A seeded phase starts from a quarter of the window the
        // previous phase ended with: the sweep files hundreds of columns at once, and with an open window the first
        // batches would take 64 of them in list order, whatever their labels (measured over 40 C3 instances, interleaved:
        // open window 937 / 983 us of solver, x 1/2: 883 / 868, x 1/4: 845 / 847, x 1/8: 833 / 838, x 1/16: 842 / 857)
        double delta = (nseed > 0 && !any_dense) ? 0.25 * delta_prev : INFINITY;
        double far_thr = INFINITY;    // near / far split of the pending columns
        SP_TICK(0);
        // ---- the seed sweep: the roots and the seeds all sit at label 0, final — no window, no bookkeeping between their
        // batches: 64 entries at a time straight from the seed list (the first batch: the roots + the first seeds); what
        // they improve is filed into pending list 0 as in any batch.  (A phase with a dense root — rare — takes the seeds
        // through the pending list instead.)
        if (nseed > 0 && !any_dense) {
            const int total = nR + nseed;
            for (int c = 0; c < total; c += SP_CAP) {
                const int m = (total - c) < SP_CAP ? (total - c) : SP_CAP;
                if (tid < m && c + tid >= nR) { L.lcol[tid] = seedlist[c + tid - nR]; L.lbase[tid] = 0.0; }
                sp_sync();
                sp_fast_batch(M, w, L, n, m, nFC, dfree, lane, wv, plcur, far_thr, fb);
                sp_sync();                                   // (the radius the batch's last wave may have published)
                dfree = L.rd[SP_RD_DFREE];
                ++batches; scans += m;
#ifdef SP_PROFILE
                SP_TICK(1); ++nfast;
#endif
            }
            nS = 0;
        } else if (nseed > 0) {
            for (int t = tid; t < nseed; t += SP_T) {
                const int k = seedlist[t];
                L.inl[k] = (unsigned char)SP_INL_NEAR;
                L.pl[0][atomicAdd(&L.ri[SP_RI_NPL], 1)] = (unsigned short)k;
                reinterpret_cast<unsigned long long*>(L.rd)[SP_RD_NMIN] = 0ull;      // the list's running minimum: label +0.0
            }
            sp_sync();
        }

        for (int guard = 0;; ++guard) {
            if (guard > 8 * n + 64) { err = 6; break; }
            // (in front of the batch, i.e. of its barrier: the counters the barrier-free bookkeeping step behind it appends
            //  to — the OTHER pending list, last read a whole step ago — and this step's selection counter)
            if (tid == 0) {
                unsigned long long* mm = reinterpret_cast<unsigned long long*>(L.rd);
                L.ri[SP_RI_NPL + (plcur ^ 1)] = 0; mm[SP_RD_NMIN + (plcur ^ 1)] = 0x7ff0000000000000ull; mm[SP_RD_NMAX + (plcur ^ 1)] = 0ull;
                L.ri[SP_RI_CSELF + plcur] = 0;
            }
            // (behind a seed sweep there is no batch — and so no barrier — between these resets and the bookkeeping step that
            //  adds to the same counters: without this one a fast wave's selection count or re-filed entries could be wiped by
            //  thread 0's late store, i.e. a pending column lost — one uncertified solve in ~50 000, found by
            //  tools/probe/fallback_hunt.py)
            if (nS == 0) sp_sync();
            if (nS > 0) {
                if (!any_dense) {
                    sp_fast_batch(M, w, L, n, nS, nFC, dfree, lane, wv, plcur, far_thr, fb);
                } else {
                    for (int t = wv; t < nS; t += SP_NW) {
                        const double b = L.lbase[t];
                        if (b < dfree) sp_entry(M, w, L, n, L.lcol[t], b, lane, plcur, dfree, far_thr);
                    }
                    sp_sync();
                    if (wv == SP_NW - 1 && L.ri[SP_RI_FREECHG]) {
                        const double dnew = sp_radius(L, nFC, lane, dfree);
                        if (lane == 0) { L.rd[SP_RD_DFREE] = dnew; L.ri[SP_RI_FREECHG] = 0; }
                    }
                }
                ++batches; scans += nS;
#ifdef SP_PROFILE
                if (!any_dense) { SP_TICK(1); ++nfast; } else { SP_TICK(5); }
#endif
            }
            // (the batch's radius is published by the last wave after the batch's barrier: it is read below, behind
            //  the barriers of the bookkeeping, which itself still works with the previous, larger one)
            SpRange rng;
            delta = sp_collect_all<true>(L, plcur, dfree, delta, far_thr, &rng);
            sp_sync();
            delta = sp_collect_result(L, delta, &nS, L.ri[SP_RI_CSELF + plcur], rng);
            dfree = L.rd[SP_RD_DFREE];
            far_thr = L.rd[SP_RD_FAR]; plcur ^= 1; any_dense = false;
            SP_TICK(2);
            if (nS > 0) continue;
            if (L.ri[SP_RI_NPL + plcur] > 0) {
                // entries are pending but the window selected none: the running minimum was stale — the exact step, once
                sp_sync();
                delta = sp_collect_all<false>(L, plcur, dfree, delta, far_thr, &rng);
                sp_sync();
                delta = sp_collect_result(L, delta, &nS, L.ri[SP_RI_CSEL], rng); far_thr = L.rd[SP_RD_FAR]; plcur ^= 1;
                SP_TICK(2);
                if (nS > 0) continue;
                if (L.ri[SP_RI_NPL + plcur] > 0) { err = 7; break; }   // tau >= dmin always selects something
            }

            // ---- near list empty: re-bucket the flagged far columns (all threads, O(n))
            {
                double dk4[SP_IPT]; bool far[SP_IPT]; double lm = INFINITY;
#pragma unroll
                for (int e = 0; e < SP_IPT; ++e) {
                    const int k = e * SP_T + tid;
                    far[e] = false; dk4[e] = INFINITY;
                    if (k < n && L.inl[k] == (unsigned char)SP_INL_FAR) {
                        const double d = L.dist[k];
                        if (d < dfree) { far[e] = true; dk4[e] = d; lm = fmin(lm, d); }
                        else L.inl[k] = 0;
                    }
                }
                const double fmin_ = sp_block_min(lm, L);
                if (fmin_ < INFINITY) {
                    far_thr = fmin_ + SP_FARMULT * delta;
                    int cnt2 = 0;
#pragma unroll
                    for (int e = 0; e < SP_IPT; ++e) { far[e] = far[e] && dk4[e] <= far_thr; cnt2 += far[e] ? 1 : 0; }
                    const int winc = sp_wave_scan(cnt2);
                    if (lane == 63) L.ri[32 + wv] = winc;
                    sp_sync();
                    const int wsum = (lane < SP_NW) ? L.ri[32 + lane] : 0;
                    const int wscan = sp_wave_scan(wsum);
                    const int nmove = __builtin_amdgcn_readlane(wscan, 63);
                    int off2 = __shfl(wscan - wsum, wv, 64) + (winc - cnt2);
                    unsigned short* near = plcur ? L.pl[1] : L.pl[0];
#pragma unroll
                    for (int e = 0; e < SP_IPT; ++e) {
                        const int k = e * SP_T + tid;
                        if (far[e]) { near[off2++] = (unsigned short)k; L.inl[k] = (unsigned char)SP_INL_NEAR; }
                    }
                    if (tid == 0) { L.ri[SP_RI_NPL + plcur] = nmove; L.rd[SP_RD_FAR] = far_thr; }
                    sp_sync();
                    delta = sp_collect_all<false>(L, plcur, dfree, delta, far_thr, &rng);
                    sp_sync();
                    delta = sp_collect_result(L, delta, &nS, L.ri[SP_RI_CSEL], rng); far_thr = L.rd[SP_RD_FAR]; plcur ^= 1;
                    SP_TICK(2);
                    if (nS > 0) continue;
                    err = 8; break;                          // fmin_ <= far_thr: something must be selected
                }
            }

            // ---- converged on the lists: a-posteriori pruning test for every root and every tree row
            if (wv == 0) {
                bool c = false;
                if (lane < nR && !L.rdn[lane]) c = !(w.cT[L.rrow[lane]] - L.ru[lane] >= dfree);
                const unsigned long long m = __ballot(c);
                if (lane == 0) { L.ri[SP_RI_RBAD] = (int)(unsigned)m; L.ri[SP_RI_RBAD + 1] = (int)(unsigned)(m >> 32); }
            }
            int cnt = 0; bool sel[SP_IPT]; double dk[SP_IPT];
#pragma unroll
            for (int e = 0; e < SP_IPT; ++e) {
                const int k = e * SP_T + tid;
                sel[e] = false; dk[e] = 0.0;
                if (k < n && L.owner[k] != SP_NOCOL && !L.ddone[k]) {
                    const double d = L.dist[k];
                    if (d < dfree) {
                        const int i = L.owner[k];
                        int kq = k; asm volatile("" : "+v"(kq));     // (M + k hoisted to the kernel's top was spilled: 0 scratch)
                        const double rj = (double)M[(size_t)i * n + kq] + L.p[k];
                        if (!(d + (w.cT[i] - rj) >= dfree)) { sel[e] = true; dk[e] = d; }
                    }
                }
                cnt += sel[e] ? 1 : 0;
            }
            int nsel, off;
            {
                const int winc = sp_wave_scan(cnt);
                if (lane == 63) L.ri[32 + wv] = winc;
                sp_sync();
                const int wsum = (lane < SP_NW) ? L.ri[32 + lane] : 0;
                const int wscan = sp_wave_scan(wsum);
                nsel = __builtin_amdgcn_readlane(wscan, 63);
                off = __shfl(wscan - wsum, wv, 64) + (winc - cnt);
            }
            const unsigned long long rbad = (unsigned long long)(unsigned)L.ri[SP_RI_RBAD] |
                                            ((unsigned long long)(unsigned)L.ri[SP_RI_RBAD + 1] << 32);
            const int nrb = __popcll(rbad);
            const int ccap = SP_CAP - nrb;               // list slots left for tree rows (the rest wait for the next test)
#pragma unroll
            for (int e = 0; e < SP_IPT; ++e) {
                const int k = e * SP_T + tid;
                if (sel[e]) {
                    if (off < ccap) {
                        L.lcol[off] = (unsigned short)(k | SP_DENSE); L.lbase[off] = dk[e]; L.ddone[k] = 1;
                    }
                    ++off;
                }
            }
            nS = nsel < ccap ? nsel : ccap;
            if (tid < SP_ROOTS && ((rbad >> tid) & 1ull)) {
                const int pos = nS + __popcll(rbad & ((1ull << tid) - 1ull));
                L.lcol[pos] = (unsigned short)(SP_ROOT | SP_DENSE | (unsigned)tid); L.lbase[pos] = 0.0; L.rdn[tid] = 1;
            }
            nS += nrb;
            any_dense = nS > 0;
            dense_scans += nS;
            sp_sync();
            SP_TICK(3);
            if (nS == 0) break;
        }
        if (err) break;
        delta_prev = delta;

        // ---- phase done: every tree accepts its nearest free column at or below the radius
        if (tid < SP_ROOTS) { L.tmin[tid] = ~0ull; L.tcol[tid] = SP_TNONE; }
        sp_sync();
        int myslot = -1, myk = 0; unsigned long long myd = ~0ull; int bad = 0;
        if (wv == 0 && lane < nFC) {
            const int k = L.fcol[lane]; const double d = L.dist[k];
            if (d < INFINITY && d <= dfree) {                 // labels above the radius are not final
                int j = k, g2 = 0;
                for (;;) {
                    const unsigned long long pr = L.pkey[j];
                    if (pr == SP_NOKEY || ++g2 > n + 1) { bad = 1; break; }
                    const int i = (int)(pr & SP_ROWMASK);
                    const int aj = L.a[i];
                    if (aj >= (int)SP_FREEROW) {              // a free row: the root of the tree, marked with its slot
                        myslot = aj & 63;
                        if (aj == (int)SP_NOCOL || myslot >= nR || (int)L.rrow[myslot] != i) { bad = 1; myslot = -1; }
                        break;
                    }
                    j = aj;
                }
                if (myslot >= 0) { myk = k; myd = (unsigned long long)__double_as_longlong(d); atomicMin(&L.tmin[myslot], myd); }
            }
        }
        sp_sync();
        if (myslot >= 0 && L.tmin[myslot] == myd) atomicMin(&L.tcol[myslot], myk);       // ties: the lowest column
        if (bad) L.ri[SP_RI_FLAG] = 3;
        sp_sync();
        if (L.ri[SP_RI_FLAG] == 3) { err = 3; break; }
        double D; int nacc;
        {
            const bool acc = lane < nR && L.tcol[lane] != SP_TNONE;
            const unsigned long long tm = acc ? L.tmin[lane] : 0ull;
            D = sp_wave_max(acc ? __longlong_as_double((long long)tm) : -INFINITY);
            nacc = __popcll(__ballot(acc));
        }
        if (nacc == 0 || !(D < INFINITY)) { err = 5; break; }
        for (int k = tid; k < n; k += SP_T) {
            const double dk = L.dist[k];
            if (dk < D) L.p[k] += D - dk;                     // v_k -= (D - d_k)
        }
        // augment: one lane per accepted tree (vertex-disjoint paths)
        if (wv == 0 && lane < nR && L.tcol[lane] != SP_TNONE) {
            int j = L.tcol[lane], g2 = 0; const int r = L.rrow[lane]; bool closed = false;
            while (g2++ <= n) {
                const int i = (int)(L.pkey[j] & SP_ROWMASK);
                const int jprev = L.a[i];
                L.owner[j] = (unsigned short)i; L.a[i] = (unsigned short)j;
                if (i == r) { closed = true; break; }
                j = jprev;
                if (j >= (int)SP_FREEROW) break;
            }
            if (!closed) L.ri[SP_RI_FLAG] = 3;
        }
        sp_sync();
        // drop the matched rows / columns from the free lists (order preserving; both have <= 64 entries)
        if (wv == 0) {
            const int k = (lane < nFC) ? (int)L.fcol[lane] : 0;
            const bool keepc = lane < nFC && L.owner[k] == SP_NOCOL;
            const unsigned long long mc = __ballot(keepc);
            const int r = (lane < nFree) ? (int)L.rrow[lane] : 0;
            const bool keepr = lane < nFree && L.a[r] >= SP_FREEROW;
            const unsigned long long mr = __ballot(keepr);
            const unsigned long long below = (1ull << lane) - 1ull;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (keepc) L.fcol[__popcll(mc & below)] = (unsigned short)k;
            if (keepr) L.rrow[__popcll(mr & below)] = (unsigned short)r;
            if (lane == 0) { L.ri[SP_RI_NFC] = __popcll(mc); L.ri[SP_RI_NFR] = __popcll(mr); }
        }
        sp_sync();
        if (L.ri[SP_RI_FLAG] == 3) { err = 3; break; }
        {
            const int nfc2 = L.ri[SP_RI_NFC], nfr2 = L.ri[SP_RI_NFR];
            if (nfc2 != nfr2 || nfr2 != nFree - nacc) { err = 4; break; }
            nFC = nfc2; nFree = nfr2;
        }
        sp_sync();
        // ---- (round 6) SEEDS of the next phase: the trees whose root did not augment.  Under the updated duals every column
        // of such a tree with d < D sits at distance 0 from its (still free) root over the SAME predecessors (tree edges
        // become tight: rc' = rc - (d_child - d_parent) = 0; the trees are vertex disjoint, so an augmentation elsewhere
        // touches none of their vertices; and distances from a shrinking root set never fall).  The next phase therefore
        // starts with these columns labelled 0 and re-scans them 64 at a time straight from a list (the seed sweep at the
        // phase start) instead of walking the tree again from its root hop by hop, a handful of zero-label entries per
        // batch: 113 of the 319 batches of a C3 solve held such entries, 6 each (profiles/r6_experiments.txt 17).  Membership is decided exactly, by the
        // walk to the root: a free row = a surviving root; a row whose match is the column the walk came from = an
        // augmented path (its matches were flipped) = an orphan, which is simply not seeded — as is a walk that runs out of
        // hops.  Seeding is an optimisation of the search order only: labels, predecessors and the finish are unchanged.
        if (nFree > 0) {
            // (the candidates — matched columns with d < D, a few hundred of the n — are compacted first, so that every walk
            //  has a thread of its own: one walk after the other per thread cost 54 k cycles per phase end, this form 10 k)
            if (tid == 0) L.ri[SP_RI_NPL] = 0;
            sp_sync();
            for (int k = tid; k < n; k += SP_T) {
                L.pl[1][k] = (unsigned short)SP_NOCOL;
                if (L.owner[k] != SP_NOCOL && L.dist[k] < D) L.pl[0][atomicAdd(&L.ri[SP_RI_NPL], 1)] = (unsigned short)k;
            }
            sp_sync();
            const int ncand = L.ri[SP_RI_NPL];
            if (tid == 0) L.ri[SP_RI_SEEDN] = 0;
            sp_sync();
            unsigned short* seedlist = L.fcol + SP_ROOTS;       // (the free columns use the first nFC <= SP_ROOTS entries of fcol)
            for (int t = tid; t < ncand; t += SP_T) {
                const int k = L.pl[0][t];
                int j = k;
                for (int g2 = 0; g2 < SP_SEED_HOPS; ++g2) {
                    const unsigned long long pr = L.pkey[j];
                    if (pr == SP_NOKEY) break;
                    const int i = (int)(pr & SP_ROWMASK);
                    const int aj = L.a[i];
                    if (aj >= (int)SP_FREEROW) {
                        if (aj != (int)SP_NOCOL) {
                            const int pos = atomicAdd(&L.ri[SP_RI_SEEDN], 1);
                            if (pos < n - SP_ROOTS) { seedlist[pos] = (unsigned short)k; L.pl[1][k] = (unsigned short)i; }   // (room of the list)
                        }
                        break;
                    }
                    if (aj == j) break;
                    j = aj;
                }
            }
            sp_sync();
        } else if (tid == 0) L.ri[SP_RI_SEEDN] = 0;
        SP_TICK(4);
    }

    for (int k = tid; k < n; k += SP_T) {
        w.p[k] = L.p[k];
        w.owner[k] = (L.owner[k] == SP_NOCOL) ? -1 : (int)L.owner[k];
        w.a[k] = (L.a[k] >= SP_FREEROW) ? -1 : (int)L.a[k];
    }
#ifdef SP_PROFILE
    if (tid == 0) {
        long long* out = reinterpret_cast<long long*>(w.part_d);     // (scratch of the former split relax rounds)
        for (int q = 0; q < 5; ++q) out[q] = dbg[q];
        out[5] = dbg[5]; out[6] = nfast; out[7] = batches; out[8] = phases; for (int q = 0; q < 6; ++q) out[9 + q] = fbv[q]; out[15] = L.ri[125];
    }
#endif
    if (tid == 0) {
        st->nFC = nFC; st->nF = nFree;
        st->st_sap_batches += batches;
        st->st_sap_row_scans += scans;
        st->st_total_row_scans += scans;
        st->st_dense_fallbacks += dense_scans;
        st->st_ms_phases += phases;
        if (err) st->error = err;
        asg_book(st, MODE_SOLVER);
        asg_enter_cert(st);
        st->mode = MODE_CERT;
    }
}
