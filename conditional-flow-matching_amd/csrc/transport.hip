// transport.hip — K4r: exact optimal transport between uniform marginals of DIFFERENT sizes.
//
// Replaces pot.emd(a, b, M) (torchcfm/optimal_transport.py:49,79,87) for x0.shape[0] != x1.shape[0]: masses 1 / B0 on
// the rows, 1 / B1 on the columns.  In integer units (g = gcd(B0, B1)): every row supplies p = B1 / g units, every
// column takes q = B0 / g, one unit = 1 / lcm(B0, B1) of mass — the TRANSPORTATION problem on the B0 x B1 matrix itself,
// not the lcm x lcm assignment problem (whose rows / columns are p- and q-fold repeated: the massively tied regime of
// every assignment solver; optimal_transport.py's exact_plan_rect takes that route up to lcm = 8192 only).
//
// Method: successive shortest augmenting paths with node potentials (the Hungarian method for the transportation
// problem), dense: row duals u, column duals v, reduced costs c - u - v >= 0, flow only on tight entries.
//   start    u_i = min_j c_ij, v = 0; every row pushes what its cheapest column still takes.
//   search   from a row r with supply left: labels per column (dist), every scanned column adds ALL its support rows
//            to the tree at its own label (flow entries are tight), a row in the tree relaxes all columns; the search
//            stops at the nearest column with demand left.  Duals: u_i += D - d_i on the tree rows, v_j += dist_j - D
//            on the scanned columns; the path takes min(supply left, demand left, smallest flow on its backward
//            entries) units.  Repeated until the row is empty, row after row.
//   finish   fp64 certificate (c - u - v >= -tol everywhere, = 0 on the support), plan = units / lcm, cost.
// The support is kept as an edge list (a basic solution has < B0 + B1 entries; capacity 2 (B0 + B1), compacted when
// full, error if that does not help) whose entries are chained per column: "the support rows of column j" and "the
// entry (i, j)" are a walk down that column's chain (a pass of the wave over the whole list per path hop was most of
// the first version's time).
//
// MI355X shape: ONE wavefront.  The method is a chain of dependent steps (a search step = pick the nearest
// unscanned column, walk its support, relax one row); all state lives in LDS (<= 112 KiB for B0 + B1 <= 2048), the
// matrix too when it fits, and a single wave needs no barrier between steps.  Measured (tools/transport_bench.py):
// 127 x 128 (d = 2) 89 ms — 914 searches, 55 k row relaxations at 1.6 us: a lone wave exposes every LDS round trip
// (~100 cycles, four per pass over the columns) — 255 x 256 0.66 s, 200 x 333 (d = 16) 0.30 s; the first version
// (whole-list passes instead of the column chains, shuffle reductions: 155 ms / 1.3 s / 0.59 s) took 2.7 s (d = 64) to
// 10.9 s (d = 2) at 512 x 500 and 25 s at 1000 x 1024.  Nearly equal sizes cascade partial flows down long chains
// (round-3 host prototypes counted them; the same finding held for an auction), so this is the EXACT path for the
// sizes of the reference's tutorials and tests — the Python side sends B0 + B1 <= 512 here — not a fast one, and not
// for 4096 vs 4000.  (POT's network simplex on the host: about a millisecond at 127 x 128.)
#include "cfm_common.h"
#include <mutex>

#define TP_NMAX 2048            // B0 + B1
#define TP_BIG 1.0e300

struct TpArgs {
    const float* M; int B0, B1, p, q;
    double* plan; double* total_cost; int* info;
    int stage_m; int ecap; long long scan_cap;
};

// wave64 DPP reductions (row_shr within 16-lane rows, then row_bcast 15 / 31; the result is read from lane 63): the
// shuffle form costs 18 dependent LDS-crossbar round trips per arg-min, and a lone wave hides none of them
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
__device__ __forceinline__ void tp_argmin(double& d, int& j) {          // wave arg-min, ties to the smaller index; uniform
    const double dm = tp_wave_min_d(d);
    j = tp_wave_min_i(d == dm ? j : 0x7fffffff);
    d = dm;
}

// index of the entry (i, j) — a walk down column j's list (every lane reads the same words) — or -1
__device__ __forceinline__ int tp_find(const int* er, const int* enext, const int* chead, int cap, int i, int j) {
    int e = chead[j];
    for (int w = 0; e >= 0 && w <= cap; ++w) {
        if (er[e] == i) return e;
        e = enext[e];
    }
    return -1;
}

__global__ __launch_bounds__(64) void tp_solve(TpArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x;
    const int B0 = A.B0, B1 = A.B1, p = A.p, q = A.q, ecap = A.ecap;
    // ---- carve LDS: columns, rows, edges, (matrix)
    char* z = lds;
    double* dist = (double*)z; z += 8 * (size_t)B1;
    double* v = (double*)z; z += 8 * (size_t)B1;
    double* u = (double*)z; z += 8 * (size_t)B0;
    double* dr = (double*)z; z += 8 * (size_t)B0;
    int* pred = (int*)z; z += 4 * (size_t)B1;
    int* rd = (int*)z; z += 4 * (size_t)B1;
    int* scn = (int*)z; z += 4 * (size_t)B1;
    int* par = (int*)z; z += 4 * (size_t)B0;
    int* rs = (int*)z; z += 4 * (size_t)B0;
    int* intree = (int*)z; z += 4 * (size_t)B0;
    int* tlist = (int*)z; z += 4 * (size_t)B0;
    int* er = (int*)z; z += 4 * (size_t)ecap;
    int* ec = (int*)z; z += 4 * (size_t)ecap;
    int* eu = (int*)z; z += 4 * (size_t)ecap;
    int* enext = (int*)z; z += 4 * (size_t)ecap;     // entries of one column are chained: chead[j] -> ... -> -1
    int* chead = (int*)z; z += 4 * (size_t)B1;
    z = (char*)(((uintptr_t)z + 15) & ~(uintptr_t)15);
    const float* Mx = A.M;
    if (A.stage_m) {
        float* ms = (float*)z;
        for (size_t k = lane; k < (size_t)B0 * B1; k += 64) ms[k] = A.M[k];
        Mx = ms;
    }
    int ne = 0, status = 1;
    long long scans = 0; int searches = 0;
    float cmax_abs = 0.f;
    for (int j = lane; j < B1; j += 64) { v[j] = 0.0; rd[j] = q; scn[j] = 0; chead[j] = -1; }
    for (int i = lane; i < B0; i += 64) { intree[i] = 0; rs[i] = p; }
    __syncthreads();
    // ---- start: row minima, greedy push into the cheapest column
    for (int i = 0; i < B0; ++i) {
        double bd = TP_BIG; int bj = 0x7fffffff;
        for (int j = lane; j < B1; j += 64) {
            const float c = Mx[(size_t)i * B1 + j];
            cmax_abs = fmaxf(cmax_abs, fabsf(c));
            if ((double)c < bd) { bd = (double)c; bj = j; }
        }
        tp_argmin(bd, bj);
        const int dlt = min(p, rd[bj]);
        __syncthreads();
        if (lane == 0) {
            u[i] = bd;
            if (dlt > 0) { er[ne] = i; ec[ne] = bj; eu[ne] = dlt; enext[ne] = chead[bj]; chead[bj] = ne; rs[i] = p - dlt; rd[bj] -= dlt; }
        }
        if (dlt > 0) ++ne;
        __syncthreads();
    }
    cmax_abs = wave_max_f(cmax_abs);
    // ---- rows with supply left: shortest augmenting paths
    for (int r = 0; r < B0 && status == 1; ++r) {
        int guard = 0;
        while (status == 1) {
            __syncthreads();
            if (rs[r] <= 0) break;
            if (++guard > p + 1) { status = -5; break; }          // every augmentation moves >= 1 unit of the row
            ++searches;
            // root
            const double ur = u[r];
            for (int j = lane; j < B1; j += 64) {
                dist[j] = ((double)Mx[(size_t)r * B1 + j] - ur) - v[j]; pred[j] = r; scn[j] = 0;
            }
            ++scans;
            int nt = 1;
            if (lane == 0) { intree[r] = 1; dr[r] = 0.0; par[r] = -1; tlist[0] = r; }
            __syncthreads();
            int jsink = -1; double D = 0.0;
            for (int it = 0; it <= B1; ++it) {
                double bd = TP_BIG; int bj = 0x7fffffff;
                for (int j = lane; j < B1; j += 64)
                    if (!scn[j] && (dist[j] < bd || (dist[j] == bd && j < bj))) { bd = dist[j]; bj = j; }
                tp_argmin(bd, bj);
                if (bj == 0x7fffffff) { status = -6; break; }      // nothing left to scan and no demand reached
                D = bd;
                if (rd[bj] > 0) { jsink = bj; break; }
                __syncthreads();
                if (lane == 0) scn[bj] = 1;
                // the support rows of column bj join the tree at label D and relax every column
                __syncthreads();
                int e = chead[bj];
                for (int w = 0; e >= 0 && w <= ecap; ++w) {
                    const int i = er[e], units = eu[e], nx = enext[e];
                    if (units > 0 && !intree[i]) {
                        __syncthreads();
                        if (lane == 0) { intree[i] = 1; dr[i] = D; par[i] = bj; tlist[nt] = i; }
                        ++nt; ++scans;
                        const double ui = u[i];
                        for (int j = lane; j < B1; j += 64) {
                            if (scn[j] || j == bj) continue;
                            const double nd = D + (((double)Mx[(size_t)i * B1 + j] - ui) - v[j]);
                            if (nd < dist[j]) { dist[j] = nd; pred[j] = i; }
                        }
                        __syncthreads();
                    }
                    e = nx;
                }
                __syncthreads();
                if (scans > A.scan_cap) { status = -7; break; }
            }
            if (status != 1) break;
            if (jsink < 0) { status = -6; break; }
            __syncthreads();
            // duals: tree rows up by D - d_i, scanned columns down by D - dist_j
            for (int k = lane; k < nt; k += 64) { const int i = tlist[k]; u[i] += D - dr[i]; }
            for (int j = lane; j < B1; j += 64) if (scn[j]) v[j] += dist[j] - D;
            __syncthreads();
            // bottleneck along the path (backward entries: a tree row and the column it came from)
            int dlt = min(rs[r], rd[jsink]);
            {
                int j = jsink;
                for (int hop = 0; hop <= B0; ++hop) {
                    const int i = pred[j], pj = par[i];
                    if (pj < 0) break;
                    const int e = tp_find(er, enext, chead, ecap, i, pj);
                    if (e < 0) { status = -8; break; }
                    dlt = min(dlt, eu[e]);
                    j = pj;
                }
            }
            if (status != 1) break;
            if (dlt <= 0) { status = -9; break; }
            // push dlt units
            {
                int j = jsink;
                for (int hop = 0; hop <= B0 && status == 1; ++hop) {
                    const int i = pred[j], pj = par[i];
                    int e = tp_find(er, enext, chead, ecap, i, j);
                    __syncthreads();
                    if (e >= 0) { if (lane == 0) eu[e] += dlt; }
                    else {
                        if (ne >= ecap) {
                            // compaction: drop the empty entries (one lane; rare)
                            if (lane == 0) {
                                int w = 0;
                                for (int k = 0; k < ne; ++k) if (eu[k] > 0) { er[w] = er[k]; ec[w] = ec[k]; eu[w] = eu[k]; ++w; }
                                for (int k = 0; k < B1; ++k) chead[k] = -1;
                                for (int k = 0; k < w; ++k) { enext[k] = chead[ec[k]]; chead[ec[k]] = k; }
                                dist[0] = (double)w;      // (dist is dead here: hand the count to the wave)
                            }
                            __syncthreads();
                            ne = (int)dist[0];
                            __syncthreads();
                            if (ne >= ecap) { status = -10; break; }
                        }
                        if (lane == 0) { er[ne] = i; ec[ne] = j; eu[ne] = dlt; enext[ne] = chead[j]; chead[j] = ne; }
                        ++ne;
                    }
                    __syncthreads();
                    if (pj < 0) break;
                    e = tp_find(er, enext, chead, ecap, i, pj);
                    if (e < 0) { status = -8; break; }
                    __syncthreads();
                    if (lane == 0) eu[e] -= dlt;
                    __syncthreads();
                    j = pj;
                }
            }
            if (status != 1) break;
            __syncthreads();
            if (lane == 0) { rs[r] -= dlt; rd[jsink] -= dlt; }
            for (int k = lane; k < nt; k += 64) intree[tlist[k]] = 0;
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- certificate, plan, cost
    int bad = 0;
    if (status == 1) {
        const double tol = 1e-10 * fmax((double)cmax_abs, 1e-30);
        for (int j = lane; j < B1; j += 64) if (rd[j] != 0) ++bad;
        for (int i = 0; i < B0; ++i) {
            const double ui = u[i];
            for (int j = lane; j < B1; j += 64)
                if ((((double)Mx[(size_t)i * B1 + j] - ui) - v[j]) < -tol) ++bad;
        }
        double tot = 0.0;
        const double L = (double)B0 * (double)p;
        for (int e = lane; e < ne; e += 64) {
            if (eu[e] <= 0) continue;
            const int i = er[e], j = ec[e];
            const double c = (double)Mx[(size_t)i * B1 + j];
            if (fabs((c - u[i]) - v[j]) > tol) ++bad;
            tot += c * (double)eu[e];
            A.plan[(size_t)i * B1 + j] = (double)eu[e] / L;
        }
        tot = wave_sum_d(tot) / L;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if (bad) status = -11;
        if (lane == 0 && A.total_cost) *A.total_cost = tot;
    }
    if (lane == 0 && A.info) {
        int nsup = 0;
        for (int e = 0; e < ne; ++e) nsup += eu[e] > 0 ? 1 : 0;
        A.info[0] = status; A.info[1] = searches; A.info[2] = (int)(scans > 0x7fffffffLL ? 0x7fffffff : scans);
        A.info[3] = nsup; A.info[4] = bad; A.info[5] = p; A.info[6] = q; A.info[7] = A.stage_m;
    }
}

static size_t tp_lds_state(int B0, int B1, int ecap) {
    return (size_t)B1 * (8 + 8 + 4 + 4 + 4 + 4) + (size_t)B0 * (8 + 8 + 4 + 4 + 4 + 4) + (size_t)ecap * 16 + 32;
}

static int tp_gcd(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

extern "C" int cfm_transport_exact_f32(const float* M, int B0, int B1, double* plan, double* total_cost, int* info,
                                       void* stream) {
    if (!M || !plan || !info || B0 < 1 || B1 < 1) return CFM_EINVAL;
    if (B0 + B1 > TP_NMAX) return CFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    static int raised_d[CFM_MAX_DEVICES];
    static std::once_flag once_d[CFM_MAX_DEVICES];
    const int dvi = cfm_device_index();
    int& raised = raised_d[dvi];
    std::call_once(once_d[dvi], [&raised] {
        hipError_t e = hipFuncSetAttribute((const void*)tp_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        raised = (e == hipSuccess) ? 1 : -1;
    });
    if (raised < 0) return CFM_EINVAL;
    const int g = tp_gcd(B0, B1);
    TpArgs A;
    A.M = M; A.B0 = B0; A.B1 = B1; A.p = B1 / g; A.q = B0 / g;
    A.plan = plan; A.total_cost = total_cost; A.info = info;
    A.ecap = 2 * (B0 + B1);
    const size_t state = tp_lds_state(B0, B1, A.ecap);
    const size_t mbytes = (size_t)B0 * B1 * sizeof(float);
    const size_t budget = 158 * 1024;
    if (state > budget) return CFM_EINVAL;
    A.stage_m = (state + mbytes + 16 <= budget) ? 1 : 0;
    A.scan_cap = 4000000LL;       // status -7 beyond (1000 x 1024, d = 8: 2.0 M row relaxations, 25 s): bounds the run time of a single-wave kernel
    const size_t lds = state + (A.stage_m ? mbytes + 16 : 0);
    int rc = cfm_hip(hipMemsetAsync(plan, 0, sizeof(double) * (size_t)B0 * B1, s));
    if (rc) return rc;
    rc = cfm_hip(hipMemsetAsync(info, 0, 8 * sizeof(int), s));
    if (rc) return rc;
    hipLaunchKernelGGL(tp_solve, dim3(1), dim3(64), lds, s, A);
    return cfm_status();
}
