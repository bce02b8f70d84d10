// assign_small.h — exact optimal assignment for SMALL problems (2 <= n <= 256) in ONE launch of ONE workgroup.
//
// Replaces pot.emd(a, b, M) (torchcfm/optimal_transport.py:49,87) at the batch sizes of the reference's
// tutorials (B = 128 / 256: examples/2D_tutorials/Flow_matching_tutorial.ipynb cell 16, conditional MNIST).
// The chip-wide state machine of assign.hip pays ~6.5 us per step whatever the grid (a step is a chain of
// dependent L2 round trips) and ~200 steps per solve: 2-3 ms at n = 256.  Here the whole problem lives in
// one workgroup: the cost matrix in REGISTERS, prices / matches / search labels in 14 KB of LDS, LDS atomics
// and workgroup barriers where the state machine has L2 atomics and kernel boundaries.
//
// Layout (round 6): a HALF-wave owns a row.  Half h = 2 * wave + (lane >> 5) of the 32 halves owns rows h, h + 32,
// ... (8 row slots); a lane holds 8 adjacent columns of each of them: 64 VGPRs.  One instruction stream serves two
// bidders, a reduction over a row is 5 DPP steps (4 inside the 16-lane DPP rows + one row broadcast) for both of
// them at once.  Until round 5 a whole wave owned a row (4 columns per lane): a bid was ~140 instructions, the
// workgroup is issue bound (16 waves on 4 SIMDs), and C1 (n = 256, d = 2: 23 bids per row) spent 0.66 of its
// 0.855 ms in 416 bid rounds.
//
// Same algorithm as assign.hip, phase by phase.  Costs are used as C = (c - cmin) * S with S a power of two that
// puts the cost range just below 2^44 (an exact rescaling); auction prices live on the integer grid of that scale
// (a bid is rounded DOWN onto it, so the bidder's new object is its strict minimum and epsilon = 0 rounds leave
// every kept pair exactly tight).
//   init     Jonker-Volgenant column reduction: p_j = max_i (min_k C_ik - C_ij).
//   phase A  epsilon-scaling forward auction, ASYNCHRONOUS inside the workgroup (round 6; the chip-wide auction has
//            been asynchronous since round 5): the state of an object is one 64-bit LDS word  price << 9 | owner + 1;
//            prices only rise, a returning ds_max_u64 is the award; a row is matched iff the object it bid for last
//            still carries its id.  Every half-wave loops on its own: find an unmatched row of mine, top-2 of
//            C_ij + p_j over the row, raise the object word — no rounds, no barriers inside a phase; the count of
//            objects without an owner (= unmatched rows) is kept exactly by the winners and ends the phase at
//            <= 2 % unmatched.  The top-2 runs in fp32 while epsilon is >= 64 fp32 ulps of the price scale (the
//            phases only have to deliver PRICES for the exact phases below; the bid itself is formed in fp64 from
//            the object's current word, corrected for a stale price copy), in fp64 for the last phase(s).
//   phase B  epsilon = 0 rounds in fp64 on a price snapshot (JV augmenting row reduction; words price << 14 |
//            round << 8 | row): a kept pair is exactly tight.  Deterministic given the prices.
//   phase C  rows that are not tight are released; one Dijkstra search per free row (the half-wave that owns the
//            row being scanned relaxes all columns and picks the next one; hand-over through LDS + barrier),
//            JV dual update, augmentation.
//   phase D  the same fp64 certificate as assign.hip (dual feasibility + complementary slackness on the
//            ORIGINAL fp32 costs, tolerance 1e-10 of the cost scale) + total cost.
// The optimum of generic costs is unique, so the permutation does not depend on the interleaving of phase A; on
// tied costs WHICH optimal permutation comes back may vary from run to run (as for the chip-wide solver;
// INTEGRATION.md).  A solve that exceeds its caps (adversarial ties) reports it in `status` and the host falls
// back to the chip-wide solver; nothing is ever returned uncertified.
#pragma once

#define SMA_T 1024
#define SMA_NW (SMA_T / 64)
#define SMA_NH (SMA_T / 32)     // half-waves = row owners
#define SMA_N 256
#define SMA_OB 9            // owner bits of a phase A object word (owner + 1; 0 = none)
#define SMA_RB 8            // row id bits of a phase B object word
#define SMA_QB 6            // round bits (epsilon = 0 rounds)
#define SMA_SHIFT (SMA_RB + SMA_QB)

struct SmaParams {
    double theta, eps0_frac, eps_last_frac, stop_frac;
    int round_cap, arr_cap, total_cap;
    int reserved;
};

struct alignas(16) SmaShared {
    unsigned long long key[SMA_N];      // phase A: price << 9 | owner + 1;  phase B: price << 14 | round << 8 | row
    double pd[SMA_N];                   // prices (integer valued): live copy in phase A, the round's snapshot in phase B
    double u[SMA_N];                    // row duals (phase C)
    double dist[SMA_N];                 // search labels
    unsigned pi[SMA_N];                 // phase A: copies of the prices on the grid of the integer codes
    short arow[SMA_N];                  // row -> column or -1
    short bidcol[SMA_N];                // row -> column it bid for in this round or -1 (phase B)
    short colrow[SMA_N];                // column -> row or -1 (phase C)
    short pred[SMA_N];                  // column -> row that labelled it
    short freelist[SMA_N];
    unsigned char done[SMA_N];          // column closed by the search
    unsigned fm[2][SMA_NH];             // phase B, per half-wave: bit r = row (half + 32 r) is unmatched
    int nfree[2];
    unsigned pmin[2];                   // phase A: the lowest price at the end of a phase (even / odd phases)
    int actl[4];                        // phase A: [0] objects without an owner, [1] cut flag, [2] bids, [3] a wave passed its total cap
    double ctld[4];                     // search hand-over: [0] label of the next row, [1] shortest path length
    int ctl[8];                         // search hand-over: [0] next row or -1, [1] found column, [2] error; [3] free rows, [4] certificate
    unsigned long long red[SMA_NW + 2];
    double redd[2 * SMA_NW + 2];
};

// bits of an integer-valued double 0 <= x < 2^51 as an unsigned integer (x + 2^52 has x in its mantissa)
__device__ __forceinline__ unsigned long long sma_u64(double x) {
    return (unsigned long long)__double_as_longlong(x + 4503599627370496.0) & 0x000fffffffffffffull;
}
__device__ __forceinline__ double sma_f64(unsigned long long k) {      // the inverse, k < 2^52
    return __longlong_as_double((long long)(k | 0x4330000000000000ull)) - 4503599627370496.0;
}
// wave-uniform values read from LDS, moved to scalar registers: a value the compiler cannot prove uniform lives in a
// VGPR — and so does every counter and flag derived from it (the kernel has 64 VGPRs beside the matrix)
__device__ __forceinline__ int sma_ui(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double sma_ud(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ void sma_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// scaled cost (c - cmin) * S as ONE fma: mcs = -cmin * S is exact (S is a power of two); INFINITY for a padded
// column.  (The empty asm keeps the compiler from hoisting the 64 conversions of a lane out of the loops over the
// row slots: as loop invariants they would need 128 more registers than the 128 a 1024-thread workgroup has.)
__device__ __forceinline__ double sma_c(float c, double mcs, double S) {
    asm volatile("" : "+v"(c));
    return fma((double)c, S, mcs);
}

// Minimum over each 32-lane half of a wave; every lane gets its own half's result.  A 32-bit min takes its DPP
// operand INSIDE the instruction (v_min_*_dpp x, x, x: a lane whose source lies outside its DPP row is left alone,
// which is the identity here): 4 steps inside the 16-lane rows, then lane 15 of rows 0 / 2 broadcast into rows 1 / 3:
// lanes 31 and 63 hold the two minima.  (s_nop 1: the two wait states between a VALU write and a DPP read of the
// same register, which the compiler cannot insert inside an asm block.)
#define SMA_HMIN(OP)                                                                                            \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                     \
                 "s_nop 1" : "+v"(x))
__device__ __forceinline__ unsigned sma_hmin_u32(unsigned x, int hi) {
    SMA_HMIN("v_min_u32_dpp");
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 31), b = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
    return hi ? b : a;
}
// (signed: fp32 values that are >= 0 up to rounding — a value a few ulps below zero must not sort above +inf)
__device__ __forceinline__ int sma_hmin_i32(int x, int hi) {
    SMA_HMIN("v_min_i32_dpp");
    const int a = __builtin_amdgcn_readlane(x, 31), b = __builtin_amdgcn_readlane(x, 63);
    return hi ? b : a;
}
// NON-NEGATIVE doubles (scaled costs + prices, search labels, +inf for padding) on their bit patterns: for x >= 0 the
// IEEE-754 bits order like the values: the minimum is the lexicographic minimum of (high word, low word).
__device__ __forceinline__ double sma_hmin_pos(double v, int hi) {
    const unsigned vh = (unsigned)__double2hiint(v), vl = (unsigned)__double2loint(v);
    const unsigned mh = sma_hmin_u32(vh, hi);
    const unsigned ml = sma_hmin_u32(vh == mh ? vl : 0xffffffffu, hi);
    return __hiloint2double((int)mh, (int)ml);
}
__device__ __forceinline__ int sma_med3_i32(int a, int b, int c) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The 8 entries of row slot r (0..7) of one half-wave: r is wave uniform.  The 16 registers of a lane's 8 slots are
// passed BY VALUE: a switch over loads m[2 r] of the register array is turned into ONE load with a computed index by
// the optimiser, which sends the whole matrix to scratch (measured: every pick became two scratch loads, a scan
// went from 0.25 to 0.7 us); a switch over values can only become moves.
#define SMA_M16(m) m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]
__device__ __forceinline__ void sma_pick1(float4 a0, float4 b0, float4 a1, float4 b1, float4 a2, float4 b2, float4 a3, float4 b3,
                                          float4 a4, float4 b4, float4 a5, float4 b5, float4 a6, float4 b6, float4 a7, float4 b7,
                                          int r, float4& a, float4& b) {
    switch (r) {
        case 0: a = a0; b = b0; break;
        case 1: a = a1; b = b1; break;
        case 2: a = a2; b = b2; break;
        case 3: a = a3; b = b3; break;
        case 4: a = a4; b = b4; break;
        case 5: a = a5; b = b5; break;
        case 6: a = a6; b = b6; break;
        default: a = a7; b = b7; break;
    }
}
// first set bit of each half of a ballot as a lane id (0 when the half's mask is empty: callers mask those halves)
__device__ __forceinline__ void sma_first_lanes(unsigned long long ball, int& l0, int& l1) {
    const unsigned lo = (unsigned)ball, hi = (unsigned)(ball >> 32);
    l0 = __builtin_amdgcn_readfirstlane(lo ? __ffs((int)lo) - 1 : 0);
    l1 = __builtin_amdgcn_readfirstlane(hi ? 31 + __ffs((int)hi) : 32);
}

// top-2 of C_ij + p_j over a row held by a half-wave (8 columns per lane), fp64 (phase B).  Results are per lane = its
// half's: best and second best value and the best column.
struct SmaTopD { double w1, w2; int j1; };

__device__ __forceinline__ SmaTopD sma_top2_f64(const float (&c)[8], const double (&p)[8], double mcs, double S, int lane) {
    const int hi = lane >> 5;
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = sma_c(c[k], mcs, S) + p[k];
    double b1 = fmin(v[0], v[1]), b2 = fmax(v[0], v[1]);
#pragma unroll
    for (int k = 2; k < 8; ++k) { b2 = fmin(b2, fmax(b1, v[k])); b1 = fmin(b1, v[k]); }
    int k1 = 7;
#pragma unroll
    for (int k = 6; k >= 0; --k) k1 = v[k] == b1 ? k : k1;
    SmaTopD t;
    t.w1 = sma_hmin_pos(b1, hi);
    int win0, win1;
    sma_first_lanes(__ballot(b1 == t.w1), win0, win1);
    t.w2 = sma_hmin_pos(lane == (hi ? win1 : win0) ? b2 : b1, hi);
    const int jj = 8 * (lane & 31) + k1;
    const int ja = __builtin_amdgcn_readlane(jj, win0), jb = __builtin_amdgcn_readlane(jj, win1);
    t.j1 = hi ? jb : ja;
    return t;
}

// the 8 prices of a lane's columns
__device__ __forceinline__ void sma_prices_d(const SmaShared& sh, int hl, double (&p)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double2 t = *reinterpret_cast<const double2*>(&sh.pd[8 * hl + 2 * q]);
        p[2 * q] = t.x; p[2 * q + 1] = t.y;
    }
}

// C_ij of column j (uniform per half-wave) of the row a half holds in c[]: every lane of the half gets it
__device__ __forceinline__ float sma_entry(const float (&c)[8], int j, int lane) {
    const int k = j & 7;
    float v = c[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        float t = c[q];
        asm volatile("" : "+v"(t));      // opaque: a select chain over array elements is folded into c[k], and the array sent to LDS / scratch
        v = k == q ? t : v;
    }
    return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * ((j >> 3) + (lane & 32)), __float_as_int(v)));
}

// LDS atomics without a result, as single instructions: the compiler's atomic optimiser wraps an atomicOr / atomicAnd /
// atomicSub whose result is unused in a wave-wide reduction (mbcnt, DPP scan, readlane: ~40 instructions) — two lanes
// of a wave issue these, with different addresses.
__device__ __forceinline__ unsigned sma_lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void sma_lds_or(unsigned* p, unsigned v) { asm volatile("ds_or_b32 %0, %1" :: "v"(sma_lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void sma_lds_and(unsigned* p, unsigned v) { asm volatile("ds_and_b32 %0, %1" :: "v"(sma_lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void sma_lds_dec(int* p) { asm volatile("ds_sub_u32 %0, %1" :: "v"(sma_lds_addr(p)), "v"(1u) : "memory"); }
__device__ __forceinline__ void sma_lds_minu(unsigned* p, unsigned v) { asm volatile("ds_min_u32 %0, %1" :: "v"(sma_lds_addr(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void sma_lds_max(int* p, int v) { asm volatile("ds_max_i32 %0, %1" :: "v"(sma_lds_addr(p)), "v"(v) : "memory"); }

// whole-wave minimum of a u32 (6 DPP steps), uniform result
__device__ __forceinline__ unsigned sma_wave_min_u32(unsigned x) {
    asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(x));
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// One Dijkstra scan by the WAVE whose half (row & 1) holds `row`: the other half takes columns 4..7 of every lane of
// the owner (v_permlane32_swap_b32: lanes 32..63 of the first operand <-> lanes 0..31 of the second), so a lane relaxes
// 4 columns as in the whole-wave layout; the nearest open column is closed and the search handed to the row matched
// to it (or the free column reached is reported).
__device__ __forceinline__ void sma_scan(SmaShared& sh, const float4& ca, const float4& cb, int row, double mcs, double S, int lane) {
    const int hl = lane & 31, oh = row & 1;
    const bool mine = (lane >> 5) == oh;
    float c[4];
    {
        const float own[4] = {ca.x, ca.y, ca.z, ca.w}, far[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = __float_as_int(far[q]);
            const auto sw = __builtin_amdgcn_permlane32_swap(f, f, false, false);   // [0]: the low half's value in both halves, [1]: the high half's
            c[q] = mine ? own[q] : __int_as_float(oh ? (int)sw[1] : (int)sw[0]);
        }
    }
    const int cb0 = 8 * hl + (mine ? 0 : 4);
    const double base = sma_ud(sh.ctld[0]), ur = sma_ud(sh.u[row]);
    const double2 pa = *reinterpret_cast<const double2*>(&sh.pd[cb0]);
    const double2 pb = *reinterpret_cast<const double2*>(&sh.pd[cb0 + 2]);
    const double2 da = *reinterpret_cast<const double2*>(&sh.dist[cb0]);
    const double2 db = *reinterpret_cast<const double2*>(&sh.dist[cb0 + 2]);
    const unsigned dn = *reinterpret_cast<const unsigned*>(&sh.done[cb0]);
    const short4 cr = *reinterpret_cast<const short4*>(&sh.colrow[cb0]);      // (not behind the reduction: one LDS round trip less)
    double d[4] = {da.x, da.y, db.x, db.y};
    // (reduced costs are >= 0 up to rounding: after a dual update a label can come out a few ulps below zero when
    //  base == 0 on tied instances — the reduction orders BIT PATTERNS, where a negative double sorts above +inf:
    //  clamped here, exactly as the chip-wide relax rounds clamp theirs)
    const double nd[4] = {base + fmax((sma_c(c[0], mcs, S) + pa.x) - ur, 0.0), base + fmax((sma_c(c[1], mcs, S) + pa.y) - ur, 0.0),
                          base + fmax((sma_c(c[2], mcs, S) + pb.x) - ur, 0.0), base + fmax((sma_c(c[3], mcs, S) + pb.y) - ur, 0.0)};
    // (selects, not branches: the four columns' arithmetic interleaves; only the LDS writes sit under a lane mask)
    double b1 = INFINITY; int k1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool open = ((dn >> (8 * k)) & 0xffu) == 0u;
        const bool imp = open && nd[k] < d[k];
        d[k] = imp ? nd[k] : d[k];
        if (imp) { sh.dist[cb0 + k] = nd[k]; sh.pred[cb0 + k] = (short)row; }
        const bool bet = open && d[k] < b1;
        b1 = bet ? d[k] : b1; k1 = bet ? k : k1;
    }
    // (labels are >= 0 or +inf: the bits order like the values.)  The high words decide unless two lanes agree in sign,
    // exponent and 20 mantissa bits: one reduction instead of two for nearly every scan.
    const unsigned vh = (unsigned)__double2hiint(b1), vl = (unsigned)__double2loint(b1);
    const unsigned mh = sma_wave_min_u32(vh);
    unsigned long long ball = __ballot(vh == mh);
    if (__popcll(ball) > 1) {
        const unsigned ml = sma_wave_min_u32(vh == mh ? vl : 0xffffffffu);
        ball = __ballot(vh == mh && vl == ml);
    }
    const int win = __builtin_amdgcn_readfirstlane(__ffsll((long long)ball) - 1);
    const double w1 = __hiloint2double((int)mh, __builtin_amdgcn_readlane((int)vl, win));
    const int js = __builtin_amdgcn_readlane(cb0 + k1, win);
    const int crk = k1 == 0 ? cr.x : k1 == 1 ? cr.y : k1 == 2 ? cr.z : cr.w;
    const int nr = __builtin_amdgcn_readlane(crk, win);
    if (lane == 0) {
        if (!(w1 < INFINITY)) { sh.ctl[0] = -1; sh.ctl[2] = 1; }
        else {
            sh.done[js] = 1;
            if (nr < 0) { sh.ctl[0] = -1; sh.ctl[1] = js; sh.ctld[1] = w1; }
            else { sh.ctl[0] = nr; sh.ctld[0] = w1; }
        }
    }
}

// ---- phase A works on 32-bit INTEGER codes (round 6).  A cost is coded as  floor((c - cmin) * 2^29 / 2^e) with its low 8
// bits replaced by the COLUMN index (range * 2^29 / 2^e in [2^28, 2^29): the perturbation is < 4.8e-7 of the cost range,
// below the last epsilon; padded columns are 2^31 | j); prices are multiples of 256 below SMA_PMAX.  value = code + price
// is ONE integer add, exact; its low byte IS the column, so the best column needs no argmin chain, values are unique
// within a row, and "the winner lane" of the second reduction is simply the lane whose best equals the reduced best.
// The codes live in the matrix registers during phase A (the fp32 costs are loaded again from L2 for the exact phases).
// A bid that would leave the 32-bit window cuts the phase with actl[1] = 2: the host falls back to the chip-wide solver.
#define SMA_PMAX 0x60000000u
#define SMA_CBITS 29

__device__ __forceinline__ unsigned sma_med3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// minima of the two halves of a wave (uniform)
__device__ __forceinline__ void sma_hmin2_u32(unsigned x, unsigned& a, unsigned& b) {
    SMA_HMIN("v_min_u32_dpp");
    a = (unsigned)__builtin_amdgcn_readlane((int)x, 31); b = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// One bid of row slot r of the two halves of a wave (act: this lane's half has an unmatched row in the slot; ca / cb: the
// slot's codes — STATIC registers, the caller unrolls over the slots).  The lane's 8 price copies are read, value = code +
// price, top-2 per lane (min / med3), the halves' best (5 DPP steps), the object's word is read while the second
// reduction runs, then the first lane of the half bids  price + (w2 - w1) + epsilon  with a returning ds_max_u64 and on
// an award refreshes the price copy.  The row's bit in its half's mask is cleared BEFORE the word is raised and set
// again if the bid fails; the winner sets the bit of the row it displaced: LDS operations of a lane are performed in
// order, so a displaced row's bit is always set after its own clear.  Objects without an owner are counted down by the
// first award: actl[0] = unmatched rows.
__device__ __forceinline__ void sma_async_bid(SmaShared& sh, const float4& ca, const float4& cb, int r, bool act, unsigned eps,
                                              int lane, int hb, int& nbids, uint4 qa, uint4 qb, bool fresh) {
    const int hi = lane >> 5, hl = lane & 31;
    if (!fresh) {                        // (the first bid of a pass uses the copies read with the pass's masks: one round trip less)
        qa = *reinterpret_cast<const uint4*>(&sh.pi[8 * hl]);
        qb = *reinterpret_cast<const uint4*>(&sh.pi[8 * hl + 4]);
    }
    const unsigned v[8] = {__float_as_uint(ca.x) + qa.x, __float_as_uint(ca.y) + qa.y, __float_as_uint(ca.z) + qa.z, __float_as_uint(ca.w) + qa.w,
                           __float_as_uint(cb.x) + qb.x, __float_as_uint(cb.y) + qb.y, __float_as_uint(cb.z) + qb.z, __float_as_uint(cb.w) + qb.w};
    unsigned b1 = min(v[0], v[1]), b2 = max(v[0], v[1]);
#pragma unroll
    for (int k = 2; k < 8; ++k) { b2 = sma_med3_u32(b1, b2, v[k]); b1 = min(b1, v[k]); }
    unsigned w1a, w1b, w2a, w2b;
    sma_hmin2_u32(b1, w1a, w1b);
    const unsigned w1 = hi ? w1b : w1a;
    const int j1 = (int)(w1 & 0xffu);
    const unsigned long long cur = sh.key[j1];                          // (in flight during the second reduction)
    sma_hmin2_u32(b1 == w1 ? b2 : b1, w2a, w2b);                        // (values are unique within a row: the low byte is the column)
    const unsigned geA = ((w2a & ~0xffu) - (w1a & ~0xffu)) + eps, geB = ((w2b & ~0xffu) - (w1b & ~0xffu)) + eps;
    if (hl == 0 && act) {
        const int i = hb + 32 * r;
        const unsigned pnew = (unsigned)(cur >> SMA_OB) + (hi ? geB : geA);       // (no wrap: price < SMA_PMAX, gap < 2^31)
        if (pnew < SMA_PMAX) {
            sma_lds_and(&sh.fm[0][hb], ~(1u << r));
            const unsigned long long nk = ((unsigned long long)pnew << SMA_OB) | (unsigned long long)(i + 1);
            const unsigned long long old = atomicMax(&sh.key[j1], nk);
            asm volatile("" ::: "memory");
            ++nbids;
            if (old < nk) {
                sh.arow[i] = (short)j1;
                sh.pi[j1] = pnew;
                const int ow = (int)(old & ((1ull << SMA_OB) - 1ull));
                if (ow == 0) sma_lds_dec(&sh.actl[0]);
                else sma_lds_or(&sh.fm[0][(ow - 1) & 31], 1u << ((ow - 1) >> 5));
            } else {
                sh.pi[j1] = (unsigned)(old >> SMA_OB);
                sma_lds_or(&sh.fm[0][hb], 1u << r);
            }
        } else sma_lds_max(&sh.actl[1], 2);
    }
}

// One asynchronous epsilon phase of one wave (two bidders): loops until <= stop rows are unmatched or the phase is cut.
// A pass reads the unmatched count, the cut flag and the wave's two row masks, then walks the 8 row slots in order
// (unrolled: the slot's registers are static — a dynamic pick was a tree of scalar branches, 650 of the 2050 cycles of an
// iteration in the profiling build) and bids for every slot in which either half has an unmatched row.
__device__ __forceinline__ int sma_async_phase(SmaShared& sh, const float4 (&m)[16], const SmaParams& P, unsigned eps, int stop,
                                               int lane, int wv, int& nbids) {
    const int hi = lane >> 5, hb = 2 * wv + hi;
    int it = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        const int2 ac = *reinterpret_cast<const int2*>(&sh.actl[0]);
        const uint2 fw = *reinterpret_cast<const uint2*>(&sh.fm[0][2 * wv]);
        const uint4 qa = *reinterpret_cast<const uint4*>(&sh.pi[8 * (lane & 31)]);
        const uint4 qb = *reinterpret_cast<const uint4*>(&sh.pi[8 * (lane & 31) + 4]);
        // (all loads in flight before the first branch: the compiler sinks a load behind the branch that needs it)
        asm volatile("" :: "v"(ac.x), "v"(fw.x), "v"(qa.x), "v"(qb.x));
        const int nf = __builtin_amdgcn_readfirstlane(ac.x), cut = __builtin_amdgcn_readfirstlane(ac.y);
        if (nf <= stop || cut) break;
        if (it > 64 * P.round_cap) { if (lane == 0) sma_lds_max(&sh.actl[1], 1); break; }      // (64 per bid, 1 per look without one)
        const unsigned m0 = (unsigned)__builtin_amdgcn_readfirstlane((int)fw.x), m1 = (unsigned)__builtin_amdgcn_readfirstlane((int)fw.y);
        const unsigned mm = m0 | m1;
        if (!mm) { ++it; __builtin_amdgcn_s_sleep(1); continue; }
        const unsigned mine = hi ? m1 : m0;
        bool fresh = true;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (!((mm >> r) & 1u)) continue;
            it += 64;
            sma_async_bid(sh, m[2 * r], m[2 * r + 1], r, ((mine >> r) & 1u) != 0u, eps, lane, hb, nbids, qa, qb, fresh);
            fresh = false;
        }
    }
    return it;
}

// the fp32 costs of a half-wave's 8 rows (8 columns per lane; +inf outside the problem)
__device__ __forceinline__ void sma_load(gfp M, int n, int hb, int hl, float4 (&m)[16]) {
    if ((n & 3) == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = hb + 32 * r;
            float4 a = make_float4(INFINITY, INFINITY, INFINITY, INFINITY), b = a;
            if (i < n && 8 * hl < n) a = asg_ld4(M + (size_t)i * n + 8 * hl);
            if (i < n && 8 * hl + 4 < n) b = asg_ld4(M + (size_t)i * n + 8 * hl + 4);
            m[2 * r] = a; m[2 * r + 1] = b;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = hb + 32 * r;
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = 8 * hl + k;
                e[k] = (i < n && j < n) ? M[(size_t)i * n + j] : INFINITY;
            }
            m[2 * r] = make_float4(e[0], e[1], e[2], e[3]); m[2 * r + 1] = make_float4(e[4], e[5], e[6], e[7]);
        }
    }
}

__global__ __launch_bounds__(SMA_T) void asg_small(const float* __restrict__ Mraw, int n, SmaParams P, int* __restrict__ perm,
                                                  int* __restrict__ certified, double* __restrict__ total_cost,
                                                  int* __restrict__ stats, int* __restrict__ status) {
    __shared__ SmaShared sh;
    gfp M = ASG_GLOBAL(Mraw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hi = lane >> 5, hl = lane & 31, hb = 2 * wv + hi;       // my half-wave; rows hb + 32 r
    const unsigned long long tk0 = wall_clock64();
    if (tid == 0 && certified) *certified = 0;          // (set by this thread again at the end, if the certificate holds)
    // ---- the matrix -> registers; cost range
    float4 m[16];
    float lmin = INFINITY, lmax = -INFINITY;
    sma_load(M, n, hb, hl, m);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const float e[4] = {m[q].x, m[q].y, m[q].z, m[q].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lmin = fminf(lmin, e[k]);                       // (a padded entry is +inf: it cannot lower the minimum)
            if (e[k] < INFINITY || (hb + 32 * (q >> 1) < n && 8 * hl + 4 * (q & 1) + k < n)) lmax = fmaxf(lmax, e[k]);
        }
    }
    if (tid < SMA_N) {
        sh.key[tid] = 0ull; sh.arow[tid] = -1; sh.bidcol[tid] = -1; sh.colrow[tid] = -1;
    }
    if (tid < 2 * SMA_NH) (&sh.fm[0][0])[tid] = 0u;
    if (tid < 2) sh.nfree[tid] = 0;
    if (tid < 4) sh.actl[tid] = 0;
    if (tid < 2) sh.pmin[tid] = 0xffffffffu;
    {
        const double a = asg_wave_min_d((double)lmin), b = -asg_wave_min_d(-(double)lmax);
        if (lane == 0) { sh.redd[wv] = a; sh.redd[SMA_NW + wv] = b; }
    }
    sma_sync();
    double cmin = sh.redd[0], cmax = sh.redd[SMA_NW];
#pragma unroll
    for (int q = 1; q < SMA_NW; ++q) { cmin = fmin(cmin, sh.redd[q]); cmax = fmax(cmax, sh.redd[SMA_NW + q]); }
    cmin = sma_ud(cmin); cmax = sma_ud(cmax);
    const double range = cmax - cmin;
    const double CM = 17592186044416.0;                      // 2^44: upper bound of the scaled cost range
    double S = 1.0;
    if (range > 0.0 && range < INFINITY) { int e; (void)frexp(range, &e); S = ldexp(1.0, 44 - e); }   // range * S in [2^43, 2^44)
    const double mcs = -cmin * S;
    const bool finite = cmin > -INFINITY && cmax < INFINITY && cmin == cmin && cmax == cmax && S < 1.0e38 && S > 1.0e-38;
    // ---- init: column reduction  p_j = max_i (min_k C_ik - C_ij) + 2^44  (>= 0)
    {
        double best[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) best[k] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float c[8] = {m[2 * r].x, m[2 * r].y, m[2 * r].z, m[2 * r].w, m[2 * r + 1].x, m[2 * r + 1].y, m[2 * r + 1].z, m[2 * r + 1].w};
            double cs[8], lm = INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) { cs[k] = sma_c(c[k], mcs, S); lm = fmin(lm, cs[k]); }
            const double rmin = sma_hmin_pos(fmax(lm, 0.0), hi);
            if (hb + 32 * r < n) {
#pragma unroll
                for (int k = 0; k < 8; ++k) best[k] = fmax(best[k], rmin - cs[k]);
            }
        }
        // the two halves of a wave hold the same columns: both raise the word
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j = 8 * hl + k;
            if (j < n && best[k] > -INFINITY)
                atomicMax(&sh.key[j], sma_u64(floor(best[k] + CM)) << SMA_OB);
        }
    }
    sma_sync();
    unsigned* pi = sh.pi;
    if (tid < SMA_N) {
        // prices on the grid of the integer codes: exact price >> 15, a multiple of 256
        const unsigned long long pr = sh.key[tid] >> SMA_OB;
        const unsigned pq = (unsigned)(pr >> (44 - SMA_CBITS)) & ~0xffu;
        pi[tid] = pq;
        sh.key[tid] = (unsigned long long)pq << SMA_OB;
    }
    unsigned fullmask = 0u;                 // rows of half-wave `tid` (tid < 32)
    for (int r = 0; r < 8; ++r) if ((tid & 31) + 32 * r < n) fullmask |= 1u << r;
    if (tid == 0) sh.actl[0] = n;
    if (tid < SMA_NH) sh.fm[0][tid] = fullmask;
    // the matrix registers -> integer codes (see sma_async_phase)
    {
        const double Si = ldexp(S, SMA_CBITS - 44), mci = -cmin * Si;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float e[4] = {m[q].x, m[q].y, m[q].z, m[q].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned j = (unsigned)(8 * hl + 4 * (q & 1) + k);
                const bool in = hb + 32 * (q >> 1) < n && (int)j < n;
                const double cs = fmin(fmax(fma((double)e[k], Si, mci), 0.0), 536870911.0);
                const unsigned code = in ? (unsigned)cs : 0x80000000u;
                e[k] = __uint_as_float((code & ~0xffu) | j);
            }
            m[q] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }
    sma_sync();

    const unsigned long long tk1 = wall_clock64();
    // ---- phase A: asynchronous epsilon > 0 phases (control variables are replicated: every thread takes the same decisions)
    const double CI = 536870912.0;          // 2^29: the integer codes' cost range bound
    double eps = fmax(256.0, P.eps0_frac * CI);
    const double eps_last = P.eps_last_frac * CI;
    const int stop = (int)(P.stop_frac * n);
    int st_auction = 0, st_arr = 0, st_scans = 0, st_cut = 0, err = 0, nbids = 0, its_total = 0;
    if (!finite) err = 3;
    while (!err) {
        const unsigned epsi = ((unsigned)eps + 255u) & ~0xffu;            // a multiple of 256, >= 256
        const int it = sma_async_phase(sh, m, P, epsi, stop, lane, wv, nbids);
        its_total += it >> 6;
        ++st_auction;
        if (its_total > P.total_cap && lane == 0) sh.actl[3] = 1;
        sma_sync();
        const int over = sma_ui(sh.actl[3]), cutv = sma_ui(sh.actl[1]);
        st_cut += cutv;
        // phase end: the lowest price is taken off every price (the auction only sees differences; the spread of the prices
        // stays below range + epsilon, but every phase lifts them all — with 2 or 3 columns, by the cost range per phase:
        // unshifted they left the 32-bit window after 7 phases)
        const int par = (st_auction - 1) & 1;
        if (tid < n) sma_lds_minu(&sh.pmin[par], (unsigned)(sh.key[tid] >> SMA_OB));
        if (tid == 0) sh.pmin[par ^ 1] = 0xffffffffu;
        sma_sync();
        // price copies from the words (a copy may be behind), every row unmatched again
        const double e2 = eps / P.theta;
        const bool last = e2 < eps_last;
        if (tid < SMA_N) {
            const unsigned pm = sh.pmin[par];
            const unsigned long long pr = (sh.key[tid] >> SMA_OB) - (tid < n ? pm : 0u);
            pi[tid] = (unsigned)pr;
            sh.key[tid] = pr << SMA_OB;
            sh.arow[tid] = -1;
        }
        if (tid == 0) { sh.actl[0] = n; sh.actl[1] = 0; }
        if (tid < SMA_NH) sh.fm[0][tid] = fullmask;
        if (over) err = 1;
        if (cutv >= 2) err = 6;                 // prices left the 32-bit window
        sma_sync();
        if (last) break;
        eps = fmax(256.0, e2);
    }
    const unsigned long long tk1b = wall_clock64();
    // the exact phases: fp32 costs back into the registers, prices onto the 2^44 grid (exact: code price << 15)
    sma_load(M, n, hb, hl, m);
    if (tid < SMA_N) {
        const unsigned long long pr = (unsigned long long)pi[tid] << (44 - SMA_CBITS);
        sh.pd[tid] = sma_f64(pr);
        sh.key[tid] = pr << SMA_SHIFT;
    }
    // ---- phase B: epsilon = 0 rounds on a price snapshot, every row unmatched at entry
    int cur = 0, arr_round = 0;
    if (tid < SMA_NH) { sh.fm[0][tid] = fullmask; sh.fm[1][tid] = 0u; }
    if (tid == 0) sh.nfree[0] = n;
    sma_sync();
    while (!err) {
        const int nxt = cur ^ 1;
        const int cnt = sma_ui(sh.nfree[cur]);
        {
            unsigned mkA = (unsigned)__builtin_amdgcn_readfirstlane((int)sh.fm[cur][2 * wv]);
            unsigned mkB = (unsigned)__builtin_amdgcn_readfirstlane((int)sh.fm[cur][2 * wv + 1]);
            if (mkA | mkB) {
                double p[8];
                sma_prices_d(sh, hl, p);
                const unsigned rnd = (unsigned)(arr_round + 1);
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (!(((mkA | mkB) >> r) & 1u)) continue;
                    const bool act = (((hi ? mkB : mkA) >> r) & 1u) != 0u;
                    const float c[8] = {m[2 * r].x, m[2 * r].y, m[2 * r].z, m[2 * r].w, m[2 * r + 1].x, m[2 * r + 1].y, m[2 * r + 1].z, m[2 * r + 1].w};
                    const SmaTopD t = sma_top2_f64(c, p, mcs, S, lane);
                    if (hl == 0 && act) {
                        const int i = hb + 32 * r;
                        const double pnew = floor(sh.pd[t.j1] + (t.w2 - t.w1));     // DOWN onto the price grid
                        if (pnew < 1.0e15) {
                            atomicMax(&sh.key[t.j1], (sma_u64(pnew) << SMA_SHIFT) | (unsigned long long)((rnd << SMA_RB) | (unsigned)i));
                            sh.bidcol[i] = (short)t.j1;
                        }
                    }
                }
            }
        }
        sma_sync();
        if (tid < SMA_N) {
            const int cand = sh.bidcol[tid];
            const int c = cand >= 0 ? cand : sh.arow[tid];
            const bool won = c >= 0 && (int)(sh.key[c] & 0xffull) == tid;
            sh.arow[tid] = won ? (short)c : (short)-1;
            sh.bidcol[tid] = -1;
            if (tid < n && !won) { atomicOr(&sh.fm[nxt][tid & 31], 1u << (tid >> 5)); atomicAdd(&sh.nfree[nxt], 1); }
            sh.pd[tid] = sma_f64(sh.key[tid] >> SMA_SHIFT);
        } else if (tid < SMA_N + SMA_NH) sh.fm[cur][tid - SMA_N] = 0u;
        else if (tid == SMA_N + SMA_NH) sh.nfree[cur] = 0;
        sma_sync();
        cur = nxt;
        st_scans += cnt;
        ++st_arr; ++arr_round;
        if (cnt == 0 || arr_round >= P.arr_cap) break;
    }

    const unsigned long long tk2 = wall_clock64();
    // ---- phase C entry: duals u_i = min_k (C_ik + p_k); a matched row keeps its column iff that pair is tight
    int st_free = 0, st_searches = 0;
    if (!err) {
        double p[8];
        sma_prices_d(sh, hl, p);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = hb + 32 * r;
            const float c[8] = {m[2 * r].x, m[2 * r].y, m[2 * r].z, m[2 * r].w, m[2 * r + 1].x, m[2 * r + 1].y, m[2 * r + 1].z, m[2 * r + 1].w};
            double v[8], lm = INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = sma_c(c[k], mcs, S) + p[k]; lm = fmin(lm, v[k]); }
            const double ui = sma_hmin_pos(lm, hi);
            const int a = i < n ? sh.arow[i] : -1;
            const int aa = a >= 0 ? a : 0;
            double mine = v[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                double t = v[q];
                asm volatile("" : "+v"(t));      // (opaque, as in sma_entry)
                mine = (aa & 7) == q ? t : mine;
            }
            const int src = 4 * ((aa >> 3) + (lane & 32));
            const double own = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(mine)),
                                                __builtin_amdgcn_ds_bpermute(src, __double2loint(mine)));
            if (hl == 0 && i < n) {
                sh.u[i] = ui;
                if (a >= 0) { if (own == ui) sh.colrow[a] = (short)i; else sh.arow[i] = -1; }
            }
        }
        sma_sync();
        if (wv == 0) {                 // compact list of the free rows
            int base = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool fr = i < n && sh.arow[i] < 0;
                const unsigned long long b = __ballot(fr);
                if (fr) sh.freelist[base + __popcll(b & ((1ull << lane) - 1ull))] = (short)i;
                base += __popcll(b);
            }
            if (lane == 0) sh.ctl[3] = base;
        }
        sma_sync();
        st_free = sma_ui(sh.ctl[3]);
        // heavily tied costs (small integers, identical points) leave MOST rows not exactly tight; the
        // searches below run one after the other, the chip-wide machine grows them as one forest:
        // hand the instance over (measured at n = 256, costs in {0..4}: 28 ms here, 5.5 ms there)
        if (st_free > 32 && 2 * st_free > n) err = 5;
    }

    const unsigned long long tk3 = wall_clock64();
    // ---- phase C: one shortest augmenting path per free row.
    // (Round 6 measured the alternatives at C1, 8 instances: up to 8 searches in lockstep from the same duals, applied when
    //  their closed columns are disjoint: 63 % are not, and a batch is as long as its longest search — no gain; all roots
    //  at once with label-correcting sweeps (no reduction, no hand-over, every half-wave relaxing side by side): 74 sweeps
    //  of 3 barriers and ~3 600 row relaxations against ~450 scans — 0.55 ms against 0.41.)
    int st_sscans = 0;
    for (int f = 0; f < st_free && !err; ++f) {
        const int root = sma_ui(sh.freelist[f]);
        if (tid < SMA_N) { sh.dist[tid] = INFINITY; sh.done[tid] = 0; }
        if (tid == 0) { sh.ctl[0] = root; sh.ctl[1] = -1; sh.ctl[2] = 0; sh.ctld[0] = 0.0; }
        if (wv == ((root & 31) >> 1)) {       // the root's dual from the current prices
            float4 ca, cb;
            sma_pick1(SMA_M16(m), root >> 5, ca, cb);
            const float c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
            double p[8], lm = INFINITY;
            sma_prices_d(sh, hl, p);
#pragma unroll
            for (int k = 0; k < 8; ++k) lm = fmin(lm, sma_c(c[k], mcs, S) + p[k]);
            const double ui = sma_hmin_pos(lm, hi);
            if (lane == ((root & 1) << 5)) sh.u[root] = ui;
        }
        sma_sync();
        ++st_searches;
        int guard = 0;
        for (;;) {
            const int row = sma_ui(sh.ctl[0]);
            if (row < 0) break;
            if (++guard > n + 2) { err = 2; break; }
            if (wv == ((row & 31) >> 1)) {
                // one code copy per row slot (static registers)
                switch (row >> 5) {
#define SMA_CASE(R) case R: sma_scan(sh, m[2 * R], m[2 * R + 1], row, mcs, S, lane); break;
                    SMA_CASE(0) SMA_CASE(1) SMA_CASE(2) SMA_CASE(3) SMA_CASE(4) SMA_CASE(5) SMA_CASE(6)
                    default: sma_scan(sh, m[14], m[15], row, mcs, S, lane); break;
#undef SMA_CASE
                }
            }
            ++st_scans; ++st_sscans;
            sma_sync();
        }
        if (err) break;
        const int jfree = sma_ui(sh.ctl[1]);
        if (sma_ui(sh.ctl[2]) || jfree < 0) { err = 2; break; }
        const double dfin = sma_ud(sh.ctld[1]);
        // dual update of the scanned columns, then the augmentation (one thread: the path is a chain)
        if (tid < SMA_N && sh.done[tid] && tid != jfree) sh.pd[tid] = sh.pd[tid] + (dfin - sh.dist[tid]);
        if (tid == 0) {
            int j = jfree, hops = 0;
            for (;;) {
                const int i = sh.pred[j];
                const int jn = sh.arow[i];
                sh.arow[i] = (short)j; sh.colrow[j] = (short)i;
                if (i == root || ++hops > n) break;
                j = jn;
            }
        }
        sma_sync();
        // duals of the matched rows from the new prices: u_i = C_{i a_i} + p_{a_i}
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = hb + 32 * r;
            const int a = i < n ? sh.arow[i] : -1;
            const float c[8] = {m[2 * r].x, m[2 * r].y, m[2 * r].z, m[2 * r].w, m[2 * r + 1].x, m[2 * r + 1].y, m[2 * r + 1].z, m[2 * r + 1].w};
            const double ci = sma_c(sma_entry(c, a >= 0 ? a : 0, lane), mcs, S);
            if (hl == 0 && a >= 0) sh.u[i] = ci + sh.pd[a];
        }
        sma_sync();
    }

    const unsigned long long tk4 = wall_clock64();
    // ---- phase D: certificate on the original costs, total cost, export
    if (!err) {
        const double q = 1.0 / S;
        double p[8];
        sma_prices_d(sh, hl, p);
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] *= q;
        double wmin = INFINITY, csum = 0.0; int bad = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = hb + 32 * r;
            const int a = i < n ? sh.arow[i] : -1;
            const bool okrow = a >= 0 && a < n && sh.colrow[a] == i;
            const float c[8] = {m[2 * r].x, m[2 * r].y, m[2 * r].z, m[2 * r].w, m[2 * r + 1].x, m[2 * r + 1].y, m[2 * r + 1].z, m[2 * r + 1].w};
            const double ca = (double)sma_entry(c, okrow ? a : 0, lane);
            if (i < n) {
                if (!okrow) { bad = 1; continue; }
                const double ui = ca + sh.pd[a] * q;
                double s = INFINITY;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (8 * hl + k < n) s = fmin(s, ((double)c[k] + p[k]) - ui);
                wmin = fmin(wmin, s);
                if (hl == 0) csum += ca;
            }
        }
        wmin = asg_wave_min_d(wmin);
        // (row sums of the two halves of a wave, in a fixed order)
        const double c0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(csum), 0), __builtin_amdgcn_readlane(__double2loint(csum), 0));
        const double c1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(csum), 32), __builtin_amdgcn_readlane(__double2loint(csum), 32));
        const unsigned long long anybad = __ballot(bad);
        if (lane == 0) { sh.redd[wv] = wmin; sh.redd[SMA_NW + wv] = c0 + c1; sh.red[wv] = anybad ? 1ull : 0ull; }
        sma_sync();
        if (tid == 0) {
            double ms = sh.redd[0], tot = sh.redd[SMA_NW]; unsigned long long b = sh.red[0];
            for (int qq = 1; qq < SMA_NW; ++qq) { ms = fmin(ms, sh.redd[qq]); tot += sh.redd[SMA_NW + qq]; b |= sh.red[qq]; }
            const double scale = fmax(fabs(cmax), fabs(cmin));
            const double tol = 1e-10 * fmax(scale, 1e-30);
            const int ok = (!b) && (ms >= -tol);
            if (!ok) err = 4;
            sh.ctl[4] = ok;
            if (ok) {
                if (certified) *certified = 1;
                if (total_cost) *total_cost = tot;
            }
        }
        sma_sync();
        if (sma_ui(sh.ctl[4])) { if (tid < n) perm[tid] = sh.arow[tid]; }
        else err = 4;
    }
    // bids of the asynchronous phases (the bidder lanes counted their own)
    if (nbids) atomicAdd(&sh.actl[2], nbids);
    sma_sync();
    if (tid == 0) {
        const int bids = sh.actl[2];
        if (stats) {
            stats[0] = its_total; stats[1] = st_arr; stats[2] = st_free; stats[3] = st_searches;
            stats[4] = st_scans; stats[5] = bids; stats[6] = 1; stats[7] = 0x40000000;     // bit 30: the one-workgroup path
        }
        status[1] = its_total; status[2] = st_free; status[3] = st_scans;
        // phase times in 10 ns ticks: load + init, bid phases, convert, searches, certificate
        status[4] = (int)(tk1 - tk0); status[5] = (int)(tk2 - tk1); status[6] = (int)(tk3 - tk2); status[7] = (int)(tk4 - tk3);
        status[8] = (int)(wall_clock64() - tk4);
        status[9] = bids; status[10] = st_auction; status[11] = st_arr; status[12] = st_cut; status[13] = st_searches; status[14] = st_sscans; status[15] = (int)(tk1b - tk1);
        __threadfence_system();
        status[0] = err ? -err : 1;
    }
}
