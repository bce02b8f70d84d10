// ode.hip — K11: ODE solve of dx/dt = MLP([x, t]) (torchdyn-style drivers).
//
// Replaces NeuralODE(torch_wrapper(model), solver="euler"|"dopri5").trajectory
// (torchdyn is third-party and absent from the reference tree; the algorithm
// restated here is the one written down in SURVEY.md Appendix A.4 and mirrored,
// line for line, by oracle/cfm_oracle.py::dopri5_trajectory — "torchdyn-style
// dopri5", parity unpinned by the reference itself).
//
//  euler  : fixed steps on t_span, fully asynchronous (no host sync).
//  dopri5 : Dormand-Prince 5(4), FSAL, one global RMS error norm over the batch
//           (hairer_norm over all B*d elements), every t_span point is a step
//           end.  Stage combinations and the scaled error norm are fused
//           elementwise kernels; the scalar step controller runs on the host in
//           fp32 (one 8-byte read-back per step attempt).
#include "cfm_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

int cfm_mlp_forward_impl(const float* x, const float* t, float tval, int has_t, int t_per_row,
                         const float* const* W, const float* const* b, const int* dims,
                         int n_layers, int B, float* out, void* ws, hipStream_t s);
extern "C" size_t cfm_mlp_ws_bytes_internal(int B, int width);

struct Stages { const float* k[7]; float c[7]; int n; };

// out = x + dt * sum_s c[s] * k[s]
__global__ __launch_bounds__(256) void ode_combine(size_t n, const float* __restrict__ x, float dt,
                                                   Stages st, float* __restrict__ out,
                                                   float* __restrict__ out2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float acc = st.c[0] * st.k[0][i];
#pragma unroll
        for (int s = 1; s < 7; ++s)
            if (s < st.n) acc = fmaf(st.c[s], st.k[s][i], acc);
        const float v = fmaf(dt, acc, x[i]);
        out[i] = v;
        if (out2) out2[i] = v;
    }
}

// sum over elements of ( dt*sum_s e[s]k[s] / (atol + rtol*max(|x|,|xn|)) )^2
__global__ __launch_bounds__(256) void ode_error(size_t n, const float* __restrict__ x,
                                                 const float* __restrict__ xn, float dt, Stages st,
                                                 float atol, float rtol, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float e = st.c[0] * st.k[0][i];
#pragma unroll
        for (int s = 1; s < 7; ++s)
            if (s < st.n) e = fmaf(st.c[s], st.k[s][i], e);
        e *= dt;
        const float sc = atol + rtol * fmaxf(fabsf(x[i]), fabsf(xn[i]));
        const float r = e / sc;
        acc += (double)r * (double)r;
    }
    acc = wave_sum_d(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// sum of ((a - b) / (atol + rtol*|x0|))^2   (b may be NULL)
__global__ __launch_bounds__(256) void ode_sqnorm(size_t n, const float* __restrict__ a,
                                                  const float* __restrict__ b,
                                                  const float* __restrict__ x0, float atol,
                                                  float rtol, double* __restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float num = b ? a[i] - b[i] : a[i];
        const float r = num / (atol + rtol * fabsf(x0[i]));
        acc += (double)r * (double)r;
    }
    acc = wave_sum_d(acc);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

static inline int ode_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

// rendezvous area of the persistent small-field solver: 3 rows of SM_MAXGRID fp64 partial-sum slots
#define SM_MAXGRID 1024
#define ODE_SYNC_BYTES (3 * SM_MAXGRID * 8 + 128 + 1792)   // tail: profiling stamps (SM_PROF builds)
extern "C" size_t cfm_ode_ws_bytes_internal(int B, int width, int d) {
    return cfm_mlp_ws_bytes_internal(B, width) + sizeof(float) * (size_t)B * d * 10 + 512 + ODE_SYNC_BYTES;
}

struct OdeWs {
    float* act;       // MLP activations
    float* k[7];
    float* x; float* xn; float* xt;
    double* red;      // 32 doubles
    char* sync;       // ODE_SYNC_BYTES
};

static OdeWs ode_carve(void* ws, int B, int width, int d) {
    OdeWs w; char* q = (char*)ws;
    w.red = (double*)q; q += 256;
    w.sync = q; q += ODE_SYNC_BYTES;
    w.act = (float*)q; q += cfm_align_up(cfm_mlp_ws_bytes_internal(B, width), 256) - 256 + 256;
    const size_t n = (size_t)B * d;
    for (int s = 0; s < 7; ++s) { w.k[s] = (float*)q; q += sizeof(float) * n; }
    w.x = (float*)q; q += sizeof(float) * n;
    w.xn = (float*)q; q += sizeof(float) * n;
    w.xt = (float*)q;
    return w;
}

static int maxwidth(const int* dims, int n_layers) {
    int m = 1;
    for (int l = 1; l < n_layers; ++l) m = dims[l] > m ? dims[l] : m;
    return m;
}

static int check_mlp(const int* dims, int n_layers, int* d_out) {
    if (!dims || n_layers < 1) return CFM_EINVAL;
    const int d = dims[n_layers];
    if (dims[0] != d + 1) return CFM_EINVAL;   // time-varying vector field: [x, t] -> dx
    *d_out = d;
    return 0;
}

#define SM_WMAX 64   // largest layer width of the fused small-field paths (= SM_W below)
static int g_ode_fused = -1;     // -1: from the environment (CFM_ODE_FUSED=0 disables), else 0 / 1
extern "C" void cfm_ode_set_fused(int on) { g_ode_fused = on ? 1 : 0; }
static int ode_small_enabled() {
    if (g_ode_fused < 0) { const char* e = getenv("CFM_ODE_FUSED"); g_ode_fused = (e && e[0] == '0') ? 0 : 1; }
    return g_ode_fused;
}
static int ode_euler_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                           const float* t_span, int n_t, float* traj, float* tspan_dev, hipStream_t s);

extern "C" int cfm_ode_euler_mlp_f32(const float* const* W, const float* const* b, const int* dims,
                                     int n_layers, const float* x0, int B, const float* t_span,
                                     int n_t, float* traj, int* nfe, void* ws, void* stream) {
    int d;
    if (!W || !b || !x0 || !t_span || !traj || !ws || B <= 0 || n_t < 1) return CFM_EINVAL;
    int rc = check_mlp(dims, n_layers, &d);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int width = maxwidth(dims, n_layers);
    OdeWs w = ode_carve(ws, B, width, d);
    const size_t n = (size_t)B * d;
    rc = cfm_hip(hipMemcpyAsync(traj, x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (rc) return rc;
    // small vector fields: rows are independent and the steps are fixed, so ONE launch integrates the
    // whole t_span (a workgroup walks its 64-row tile through every step, weights resident in LDS)
    if (ode_small_enabled() && n_layers == 4 && dims[1] <= SM_WMAX && dims[2] <= SM_WMAX && dims[3] <= SM_WMAX &&
        d + 1 <= SM_WMAX && n >= (size_t)n_t && n_t >= 2) {
        rc = ode_euler_small(W, b, dims, B, d, t_span, n_t, traj, w.xt, s);
        if (nfe) *nfe = n_t - 1;
        return rc;
    }
    int evals = 0;
    for (int k = 0; k + 1 < n_t; ++k) {
        const float t = t_span[k], dt = t_span[k + 1] - t_span[k];
        const float* xk = traj + (size_t)k * n;
        rc = cfm_mlp_forward_impl(xk, nullptr, t, 1, 0, W, b, dims, n_layers, B, w.k[0], w.act, s);
        if (rc) return rc;
        ++evals;
        Stages st{}; st.k[0] = w.k[0]; st.c[0] = 1.f; st.n = 1;
        for (int q = 1; q < 7; ++q) { st.k[q] = w.k[0]; st.c[q] = 0.f; }
        hipLaunchKernelGGL(ode_combine, dim3(ode_blocks(n)), dim3(256), 0, s, n, xk, dt, st,
                           traj + (size_t)(k + 1) * n, (float*)nullptr);
    }
    if (nfe) *nfe = evals;
    return cfm_status();
}

// Dormand-Prince 5(4) tableau
static const float DP_C[6] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f, 1.f};
static const double DP_A[6][6] = {
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
static const double DP_BSOL[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
static const double DP_BALT[7] = {1951.0 / 21600, 0, 22642.0 / 50085, 451.0 / 720, -12231.0 / 42400,
                                  649.0 / 6300, 1.0 / 60};

static double* g_ode_pinned = nullptr;

static int read_red(hipStream_t s, const double* dev, int count, double* host) {
    if (!g_ode_pinned) {
        int rc = cfm_hip(hipHostMalloc((void**)&g_ode_pinned, 64, hipHostMallocDefault));
        if (rc) return rc;
    }
    int rc = cfm_hip(hipMemcpyAsync(g_ode_pinned, dev, sizeof(double) * count, hipMemcpyDeviceToHost, s));
    if (rc) return rc;
    rc = cfm_hip(hipStreamSynchronize(s));
    if (rc) return rc;
    for (int i = 0; i < count; ++i) host[i] = g_ode_pinned[i];
    return 0;
}

static int ode_dopri5_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                            const float* t_span, int n_t, float atol, float rtol, float* traj,
                            int* n_steps, int* nfe, float* xbuf, float* kbuf, float* tspan_dev,
                            void* state_dev, char* sync_dev, float t0, float dt0, int evals0, hipStream_t s);

extern "C" int cfm_ode_dopri5_mlp_f32(const float* const* W, const float* const* b, const int* dims,
                                      int n_layers, const float* x0, int B, const float* t_span,
                                      int n_t, float atol, float rtol, float* traj, int* n_steps,
                                      int* nfe, void* ws, void* stream) {
    int d;
    if (!W || !b || !x0 || !t_span || !traj || !ws || B <= 0 || n_t < 2) return CFM_EINVAL;
    int rc = check_mlp(dims, n_layers, &d);
    if (rc) return rc;
    for (int k = 0; k + 1 < n_t; ++k)
        if (!(t_span[k + 1] > t_span[k])) return CFM_EINVAL;   // forward integration only
    hipStream_t s = (hipStream_t)stream;
    const int width = maxwidth(dims, n_layers);
    OdeWs w = ode_carve(ws, B, width, d);
    const size_t n = (size_t)B * d;
    const int nb = ode_blocks(n);
    const float order = 5.f;
    int evals = 0, steps = 0;

    auto f = [&](float t, const float* xin, float* kout) -> int {
        ++evals;
        return cfm_mlp_forward_impl(xin, nullptr, t, 1, 0, W, b, dims, n_layers, B, kout, w.act, s);
    };
    auto rms = [&](double sumsq) -> float { return (float)sqrt(sumsq / (double)n); };

    rc = cfm_hip(hipMemcpyAsync(w.x, x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (rc) return rc;
    rc = cfm_hip(hipMemcpyAsync(traj, x0, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (rc) return rc;

    float t = t_span[0];
    const float T = t_span[n_t - 1];
    rc = f(t, w.x, w.k[0]);
    if (rc) return rc;

    // ---- init_step (Hairer II.4) ----
    float dt;
    {
        rc = cfm_hip(hipMemsetAsync(w.red, 0, 64, s));
        if (rc) return rc;
        hipLaunchKernelGGL(ode_sqnorm, dim3(nb), dim3(256), 0, s, n, w.x, (const float*)nullptr, w.x, atol, rtol, w.red + 0);
        hipLaunchKernelGGL(ode_sqnorm, dim3(nb), dim3(256), 0, s, n, w.k[0], (const float*)nullptr, w.x, atol, rtol, w.red + 1);
        double r2[3];
        rc = read_red(s, w.red, 2, r2);
        if (rc) return rc;
        const float d0 = rms(r2[0]), d1 = rms(r2[1]);
        const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
        Stages st{}; st.n = 1; st.c[0] = 1.f;
        for (int q = 0; q < 7; ++q) st.k[q] = w.k[0];
        hipLaunchKernelGGL(ode_combine, dim3(nb), dim3(256), 0, s, n, w.x, h0, st, w.xt, (float*)nullptr);
        rc = f(t + h0, w.xt, w.k[1]);
        if (rc) return rc;
        hipLaunchKernelGGL(ode_sqnorm, dim3(nb), dim3(256), 0, s, n, w.k[1], w.k[0], w.x, atol, rtol, w.red + 2);
        rc = read_red(s, w.red + 2, 1, r2);
        if (rc) return rc;
        const float d2 = rms(r2[0]) / h0;
        float h1;
        if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
        else h1 = powf(0.01f / fmaxf(d1, d2), 1.0f / (order + 1.f));
        dt = fminf(100.f * h0, h1);
    }

    // small vector fields: the whole step attempt in one kernel, controller on the device
    if (ode_small_enabled() && n_layers == 4 && dims[1] <= SM_WMAX && dims[2] <= SM_WMAX && dims[3] <= SM_WMAX &&
        d + 1 <= SM_WMAX && n >= (size_t)n_t)
        return ode_dopri5_small(W, b, dims, B, d, t_span, n_t, atol, rtol, traj, n_steps, nfe, w.x, w.k[0], w.xt,
                                (void*)(w.red + 16), w.sync, t, dt, evals, s);

    int ckpt = 1;  // next t_span index to land on
    const int max_attempts = 1000000;
    while (t < T && steps < max_attempts) {
        if (t + dt > T) dt = T - t;
        float dt_old = dt; bool flag = false;
        if (ckpt < n_t && t + dt > t_span[ckpt]) { dt_old = dt; flag = true; dt = t_span[ckpt] - t; }
        const bool lands = (ckpt < n_t) && (flag || t + dt == t_span[ckpt]);
        // stages k2..k6, then x_new and k7 (FSAL)
        for (int sIdx = 0; sIdx < 6; ++sIdx) {
            Stages st{}; st.n = sIdx + 1;
            for (int q = 0; q < 7; ++q) { st.k[q] = w.k[q < 7 ? q : 0]; st.c[q] = q <= sIdx ? (float)DP_A[sIdx][q] : 0.f; }
            float* dst = (sIdx == 5) ? w.xn : w.xt;
            hipLaunchKernelGGL(ode_combine, dim3(nb), dim3(256), 0, s, n, w.x, dt, st, dst, (float*)nullptr);
            rc = f(t + DP_C[sIdx] * dt, dst, w.k[sIdx + 1]);
            if (rc) return rc;
        }
        {
            Stages st{}; st.n = 7;
            for (int q = 0; q < 7; ++q) { st.k[q] = w.k[q]; st.c[q] = (float)(DP_BSOL[q] - DP_BALT[q]); }
            rc = cfm_hip(hipMemsetAsync(w.red + 4, 0, 8, s));
            if (rc) return rc;
            hipLaunchKernelGGL(ode_error, dim3(nb), dim3(256), 0, s, n, w.x, w.xn, dt, st, atol, rtol, w.red + 4);
        }
        double e2;
        rc = read_red(s, w.red + 4, 1, &e2);
        if (rc) return rc;
        ++steps;
        const float ratio = rms(e2);
        const bool accept = ratio <= 1.f;
        if (accept) {
            if (lands) {
                t = t_span[ckpt];
                rc = cfm_hip(hipMemcpyAsync(traj + (size_t)ckpt * n, w.xn, n * sizeof(float), hipMemcpyDeviceToDevice, s));
                if (rc) return rc;
                ++ckpt;
            } else {
                t = t + dt;
            }
            float* tmp = w.x; w.x = w.xn; w.xn = tmp;          // x <- x_new
            tmp = w.k[0]; w.k[0] = w.k[6]; w.k[6] = tmp;        // k1 <- k7 (FSAL)
        }
        if (flag) dt = dt_old - dt;
        // adapt_step(dt, ratio, safety=0.9, min_factor=0.2, max_factor=10, order=5)
        float factor;
        if (ratio == 0.f) factor = 10.f;
        else {
            const float minf = ratio < 1.f ? 1.f : 0.2f;
            factor = fminf(10.f, fmaxf(0.9f / powf(ratio, 1.f / order), minf));
        }
        dt = dt * factor;
        if (!(dt > 1e-12f)) dt = 1e-12f;   // guard (documented deviation)
    }
    if (n_steps) *n_steps = steps;
    if (nfe) *nfe = evals;
    rc = cfm_hip(hipStreamSynchronize(s));
    if (rc) return rc;
    return (t < T) ? CFM_ENOCONV : 0;
}

// =====================================================================================
// Fused Dormand-Prince step for SMALL vector fields (4 linear layers, every width <= 64:
// the reference's 2-D tutorials and single-cell models, MLP(dim, w=64)).
//
// The layer-per-kernel driver above spends ~31 launches and one host read-back per step
// attempt: 266 us per step at B = 8192, d = 50, w = 64 where the arithmetic is ~20 us.
// Rows are independent inside a step (only the error norm couples them), so here ONE persistent
// kernel does the whole adaptive solve: a workgroup keeps the four weight matrices in LDS (68 KB)
// and owns a 16-row tile (SM_MB = 1: B = 8192 -> 512 workgroups, two per CU, so one workgroup's
// MFMA phase overlaps the other's SELU epilogue; measured 3.46 ms against 3.87 ms for 32-row tiles
// with two accumulator chains per wave and one workgroup per CU), holds x and k1..k7 of its tile
// in MFMA accumulator layout in registers, and runs the six stage evaluations back to back.  Wave w
// owns output columns 16w..16w+15: a v_mfma_f32_16x16x4_f32 accumulator chain with the same
// ascending-k fp32 fma chain and epilogue as mlp_layer: bitwise the same field values.  The
// accept / reject decision and the next step size are taken ON THE DEVICE by every workgroup from
// the same all-reduced error norm (same fp32 controller as the host loop above); the host launches
// once and reads the step counters back.
// =====================================================================================
#define SM_W 64
#define SM_LD 68     // row stride = 4 (mod 64): fragment reads (row = lane & 15, k = lane >> 4) hit 64 distinct banks
#define SM_MB 1      // 16-row blocks per tile = independent MFMA accumulator chains per wave
#define SM_ROWS (16 * SM_MB)
#define SM_V (4 * SM_MB)   // tile floats per lane: element i -> row sm_row(i, lane), column 16 * wave + (lane & 15)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct SmTile { float v[SM_V]; };
__device__ __forceinline__ int sm_row(int i, int lane) { return 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3); }


// same SELU as mlp.hip (bitwise)
__device__ __forceinline__ float selu_f(float x) {
    return x > 0.f ? 1.0507009873554805f * x : (1.0507009873554805f * 1.6732632423543772f) * expm1f(x);
}
// tableau entries as compile-time constants (indices are constants after unrolling); the casts
// mirror the host driver: (float)DP_A[s][q], (float)(DP_BSOL[q] - DP_BALT[q])
__device__ __forceinline__ double DP_A_dev(int s, int q) {
    constexpr double A[6][6] = {
        {1.0 / 5, 0, 0, 0, 0, 0},
        {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
        {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
        {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
        {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
    return A[s][q];
}
__device__ __forceinline__ float DP_C_dev(int s) {
    constexpr float C[6] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f, 1.f};
    return C[s];
}
__device__ __forceinline__ double DP_E_dev(int q) {
    constexpr double BS[7] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0};
    constexpr double BA[7] = {1951.0 / 21600, 0, 22642.0 / 50085, 451.0 / 720, -12231.0 / 42400, 649.0 / 6300, 1.0 / 60};
    return BS[q] - BA[q];
}

struct SmState { float t, dt; int ckpt, steps, evals, par, done, pad; };
struct SmArgs { const float* W[4]; const float* b[4]; int dims[5]; };

// the step-size clipping the host loop does before every attempt
__device__ __forceinline__ void sm_prestep(const SmState& st, const float* __restrict__ tspan, int n_t,
                                           float& dt, float& dt_old, bool& flag, bool& lands) {
    const float T = tspan[n_t - 1];
    dt = st.dt;
    if (st.t + dt > T) dt = T - st.t;
    dt_old = dt; flag = false;
    if (st.ckpt < n_t && st.t + dt > tspan[st.ckpt]) { dt_old = dt; flag = true; dt = tspan[st.ckpt] - st.t; }
    lands = (st.ckpt < n_t) && (flag || st.t + dt == tspan[st.ckpt]);
}

// Workgroup barrier that orders LDS traffic only: global stores of the tile (trajectory rows) stay in
// flight across it instead of being drained (s_waitcnt vmcnt(0)) the way __syncthreads() would
__device__ __forceinline__ void sm_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// one layer on the tile: out(C layout) = A[SM_ROWS x K] * W_l[64 x K]^T, this wave's 16 columns
__device__ __forceinline__ void sm_gemm(const float* __restrict__ Abuf, const float* __restrict__ Wl, int K,
                                        int wv, int lane, f32x4 (&c)[SM_MB]) {
#pragma unroll
    for (int m = 0; m < SM_MB; ++m) c[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = lane >> 4;
    const float* ap = Abuf + fr * SM_LD + fk;
    const float* bp = Wl + (wv * 16 + fr) * SM_LD + fk;
    // every layer runs the full 16 k-steps (rows / columns beyond K are zero in both operands, and
    // fma(0, 0, acc) leaves acc alone), fully unrolled: all operand reads are in flight before the
    // first MFMA issues, then the accumulator chain(s) run back to back in ascending k
    (void)K;
    float a[SM_MB][SM_W / 4], b[SM_W / 4];
#pragma unroll
    for (int j = 0; j < SM_W / 4; ++j) {
#pragma unroll
        for (int m = 0; m < SM_MB; ++m) a[m][j] = ap[16 * m * SM_LD + 4 * j];
        b[j] = bp[4 * j];
    }
#pragma unroll
    for (int j = 0; j < SM_W / 4; ++j) {
#pragma unroll
        for (int m = 0; m < SM_MB; ++m) c[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][j], b[j], c[m], 0, 0, 0);
    }
}

// f(t, y) for the tile; y arrives in C layout, the result leaves in C layout (columns >= d are 0)
__device__ __forceinline__ SmTile sm_field(const SmTile& y, float t, const SmArgs& A, int d, float* Abuf0,
                                           float* Abuf1, const float* Wl, const float* bl, const float* wt,
                                           int wv, int lane) {
    const int col = wv * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < SM_V; ++i) Abuf0[sm_row(i, lane) * SM_LD + col] = (col < d) ? y.v[i] : 0.f;
    sm_lds_barrier();
    SmTile acc;
    float* src = Abuf0; float* dst = Abuf1;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int K = (l == 0) ? d : A.dims[l];
        const int N = A.dims[l + 1];
        f32x4 c[SM_MB];
        sm_gemm(src, Wl + l * SM_W * SM_LD, K, wv, lane, c);
        const float bv = (col < N) ? bl[l * SM_W + col] : 0.f;
        const float wtc = (l == 0 && col < N) ? wt[col] : 0.f;
#pragma unroll
        for (int i = 0; i < SM_V; ++i) {
            float v = c[i >> 2][i & 3] + bv;
            if (l == 0) v = fmaf(t, wtc, v);
            if (l < 3) v = selu_f(v);
            acc.v[i] = (col < N) ? v : 0.f;
        }
        if (l < 3) {
#pragma unroll
            for (int i = 0; i < SM_V; ++i) dst[sm_row(i, lane) * SM_LD + col] = acc.v[i];
            sm_lds_barrier();
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    return acc;
}

// weights -> LDS, zero padded to [4][64][SM_LD]; biases; the time column of layer 0
__device__ __forceinline__ void sm_stage_weights(const SmArgs& A, int d, float* Wl, float* bl, float* wt, int tid) {
    for (int l = 0; l < 4; ++l) {
        const int in_l = A.dims[l], out_l = A.dims[l + 1];
        const int K = (l == 0) ? d : in_l;
        for (int e = tid; e < SM_W * SM_LD; e += 256) {
            const int r = e / SM_LD, k = e % SM_LD;
            Wl[l * SM_W * SM_LD + e] = (r < out_l && k < K) ? A.W[l][(size_t)r * in_l + k] : 0.f;
        }
        if (tid < SM_W) bl[l * SM_W + tid] = (tid < out_l) ? A.b[l][tid] : 0.f;
    }
    if (tid < SM_W) wt[tid] = (tid < A.dims[1]) ? A.W[0][(size_t)tid * A.dims[0] + d] : 0.f;
}

// Grid-wide rendezvous + all-reduce of a persistent launch (grid <= workgroups that are resident at
// once).  There is no arrival counter: the fp64 partial sum a workgroup publishes IS its arrival.  Slots
// hold a negative NaN until their owner stores its (non-negative) partial; wave 0 of every workgroup polls
// all slots (L2 reads of distinct words, no same-address atomics to serialise) and, once none is NaN, has
// the values in hand: it adds them in a fixed order, so the total is the same bit pattern in every
// workgroup and in every run.  Three slot rows rotate (attempt % 3): at the start of attempt a the owner
// re-arms its slot of row (a + 1) % 3, which held attempt a - 2 (everybody finished reading that before
// publishing a - 1, which the owner saw at rendezvous a - 1), and its release store of attempt a orders
// the re-arm before anything a reader of attempt a + 1 can see.  Measured on MI355X, 256 workgroups:
// an arrival counter + polling cost 7-9 us per rendezvous, a two-level counter tree 15 us.
// The wait is bounded (wall clock, ~4 s): a mis-sized launch turns into an error code, never a hung GPU.
__device__ __forceinline__ bool sm_grid_allsum(double* __restrict__ row, double mine, int tid, int lane, int wv,
                                               double* sh_total, int* sh_ok) {
    if (wv == 0) {
        if (lane == 0) __hip_atomic_store(&row[blockIdx.x], mine, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        double tsum;
        for (;;) {
            tsum = 0.0;
            bool all = true;
            for (int i = lane; i < (int)gridDim.x; i += 64) {
                const double v = __hip_atomic_load(&row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && (v >= 0.0);
                tsum += v;
            }
            if (__all(all)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {                  // the clock read is slow: look at it rarely
                const unsigned long long now = wall_clock64();
                if (!t0) t0 = now;
                else if (now - t0 > 400000000ull) { ok = 0; break; }
            }
        }
        tsum = wave_sum_d(tsum);
        if (lane == 0) { *sh_total = tsum; *sh_ok = ok; }
    }
    (void)tid;
    sm_lds_barrier();
    return *sh_ok != 0;
}

// The whole adaptive integration in ONE persistent launch.  A workgroup owns the row tiles
// blockIdx.x, blockIdx.x + gridDim.x, ... for the entire solve.  RESIDENT (one tile per workgroup, B <=
// 32 x resident workgroups = 8192 on MI355X): x and k1 live in registers from the first attempt to the
// last, an accepted step is a register move, and the only global traffic of an attempt is the trajectory
// row it lands on.  Otherwise x / k1 travel between a workgroup and its own rows of the (L2-resident)
// parity buffers.  The only cross-workgroup value of a step attempt is the squared error norm: every
// workgroup stores its fp64 partial, one grid rendezvous, and then every workgroup adds the partials up in
// the same fixed order (so the solve is reproducible bit for bit), derives the same accept / reject
// decision and next step size from that sum (the fp32 controller of the host loop above).  `lines` is only
// touched by SM_PROF builds (phase stamps).
template <bool RESIDENT>
__global__ __launch_bounds__(256) void ode_small_dopri(SmArgs A, int B, int d, SmState* __restrict__ st_io,
                                                    float* __restrict__ xbuf, float* __restrict__ kbuf,
                                                    const float* __restrict__ tspan, int n_t, float atol, float rtol,
                                                    float* __restrict__ traj, double* __restrict__ partial,
                                                    unsigned long long* __restrict__ lines, int max_attempts) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    float* Wl = small_lds;                           // [4][64][SM_LD]
    float* bl = Wl + 4 * SM_W * SM_LD;               // [4][64]
    float* wt = bl + 4 * SM_W;                       // [64] time column of layer 0
    float* Ab0 = wt + SM_W;                          // [SM_ROWS][SM_LD]
    float* Ab1 = Ab0 + SM_ROWS * SM_LD;
    __shared__ double redw[4];
    __shared__ double sh_total;
    __shared__ int sh_ok;
    SmState st = st_io[0];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    sm_stage_weights(A, d, Wl, bl, wt, tid);
    const size_t n = (size_t)B * d;
    const int col = wv * 16 + (lane & 15);
    const float T = tspan[n_t - 1];
    SmTile x, k0, k1, k2, k3, k4, k5, k6, y;
    if (RESIDENT) {
#pragma unroll
        for (int i = 0; i < SM_V; ++i) {
            const int gr = blockIdx.x * SM_ROWS + sm_row(i, lane);
            const bool ok = gr < B && col < d;
            x.v[i] = ok ? xbuf[(size_t)gr * d + col] : 0.f;
            k0.v[i] = ok ? kbuf[(size_t)gr * d + col] : 0.f;
        }
    }
    __syncthreads();                                  // weights staged
    int attempt = 0, err = 0;
    for (; !st.done; ++attempt) {
        if (attempt >= max_attempts) { err = 1; break; }
        if (tid == 0)                                 // re-arm this workgroup's slot of the next attempt
            __hip_atomic_store(&partial[(size_t)((attempt + 1) % 3) * SM_MAXGRID + blockIdx.x], __longlong_as_double(-1ll),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float dt, dt_old; bool flag, lands;
        sm_prestep(st, tspan, n_t, dt, dt_old, flag, lands);
        const float* x_in = xbuf + (size_t)st.par * n;
        const float* k1_in = kbuf + (size_t)st.par * n;
        float* x_out = xbuf + (size_t)(st.par ^ 1) * n;
        float* k7_out = kbuf + (size_t)(st.par ^ 1) * n;
        double esum = 0.0;
        for (int row0 = blockIdx.x * SM_ROWS; row0 < B; row0 += gridDim.x * SM_ROWS) {
            if (!RESIDENT) {
#pragma unroll
                for (int i = 0; i < SM_V; ++i) {
                    const int gr = row0 + sm_row(i, lane);
                    const bool ok = gr < B && col < d;
                    x.v[i] = ok ? x_in[(size_t)gr * d + col] : 0.f;
                    k0.v[i] = ok ? k1_in[(size_t)gr * d + col] : 0.f;
                }
                sm_lds_barrier();                     // previous tile done with the activation buffers
            }
            // stage S (a literal): y = x + dt * sum_{q<=S} a[S][q] k_q ; KOUT = f(t + c[S] dt, y)
#define SM_STAGE(S, KOUT)                                                                            \
            {                                                                                        \
                _Pragma("unroll") for (int i = 0; i < SM_V; ++i) {                                   \
                    float acc = (float)DP_A_dev(S, 0) * k0.v[i];                                     \
                    if (S >= 1) acc = fmaf((float)DP_A_dev(S, 1), k1.v[i], acc);                     \
                    if (S >= 2) acc = fmaf((float)DP_A_dev(S, 2), k2.v[i], acc);                     \
                    if (S >= 3) acc = fmaf((float)DP_A_dev(S, 3), k3.v[i], acc);                     \
                    if (S >= 4) acc = fmaf((float)DP_A_dev(S, 4), k4.v[i], acc);                     \
                    if (S >= 5) acc = fmaf((float)DP_A_dev(S, 5), k5.v[i], acc);                     \
                    y.v[i] = fmaf(dt, acc, x.v[i]);                                                  \
                }                                                                                    \
                KOUT = sm_field(y, st.t + DP_C_dev(S) * dt, A, d, Ab0, Ab1, Wl, bl, wt, wv, lane);   \
            }
            SM_STAGE(0, k1) SM_STAGE(1, k2) SM_STAGE(2, k3) SM_STAGE(3, k4) SM_STAGE(4, k5) SM_STAGE(5, k6)
#undef SM_STAGE
            // y is x_new (the 5th-order solution), k6 = f(t + dt, x_new): error + outputs
#pragma unroll
            for (int i = 0; i < SM_V; ++i) {
                const int gr = row0 + sm_row(i, lane);
                if (gr < B && col < d) {
                    float e = (float)DP_E_dev(0) * k0.v[i];
                    e = fmaf((float)DP_E_dev(1), k1.v[i], e);
                    e = fmaf((float)DP_E_dev(2), k2.v[i], e);
                    e = fmaf((float)DP_E_dev(3), k3.v[i], e);
                    e = fmaf((float)DP_E_dev(4), k4.v[i], e);
                    e = fmaf((float)DP_E_dev(5), k5.v[i], e);
                    e = fmaf((float)DP_E_dev(6), k6.v[i], e);
                    e *= dt;
                    const float sc = atol + rtol * fmaxf(fabsf(x.v[i]), fabsf(y.v[i]));
                    const float rr = e / sc;
                    esum += (double)rr * (double)rr;
                    if (!RESIDENT) {
                        x_out[(size_t)gr * d + col] = y.v[i];
                        k7_out[(size_t)gr * d + col] = k6.v[i];
                    }
                }
            }
        }
        esum = wave_sum_d(esum);
        if (lane == 0) redw[wv] = esum;
        if (RESIDENT) sm_lds_barrier(); else __syncthreads();   // (the streamed path re-reads its rows below)
        if (!sm_grid_allsum(partial + (size_t)(attempt % 3) * SM_MAXGRID, redw[0] + redw[1] + redw[2] + redw[3], tid, lane,
                            wv, &sh_total, &sh_ok)) { err = 2; break; }
        // accept / reject, next step size (identical in every workgroup and lane)
        const float ratio = (float)sqrt(sh_total / (double)n);
        const bool accept = ratio <= 1.f;
        if (RESIDENT) {
            if (accept) {
                if (lands) {
                    float* dst = traj + (size_t)st.ckpt * n;
#pragma unroll
                    for (int i = 0; i < SM_V; ++i) {
                        const int gr = blockIdx.x * SM_ROWS + sm_row(i, lane);
                        if (gr < B && col < d) dst[(size_t)gr * d + col] = y.v[i];
                    }
                }
                x = y; k0 = k6;                       // FSAL
            }
        } else if (accept && lands) {
            const float* xn = xbuf + (size_t)(st.par ^ 1) * n;
            float* dst = traj + (size_t)st.ckpt * n;
            for (int row0 = blockIdx.x * SM_ROWS; row0 < B; row0 += gridDim.x * SM_ROWS) {
                const int rows = (B - row0 < SM_ROWS) ? B - row0 : SM_ROWS;
                const size_t base = (size_t)row0 * d;
                for (int e = tid; e < rows * d; e += 256) dst[base + e] = xn[base + e];   // own rows, written above
            }
        }
        SmState nx = st;
        nx.steps = st.steps + 1; nx.evals = st.evals + 6;
        if (accept) {
            nx.t = lands ? tspan[st.ckpt] : st.t + dt;
            if (lands) nx.ckpt = st.ckpt + 1;
            nx.par = st.par ^ 1;
        }
        float ndt = dt;
        if (flag) ndt = dt_old - dt;
        float factor;
        if (ratio == 0.f) factor = 10.f;
        else {
            const float minf = ratio < 1.f ? 1.f : 0.2f;
            factor = fminf(10.f, fmaxf(0.9f / powf(ratio, 1.f / 5.f), minf));
        }
        ndt = ndt * factor;
        if (!(ndt > 1e-12f)) ndt = 1e-12f;
        nx.dt = ndt;
        nx.done = (nx.t < T) ? 0 : 1;
        st = nx;
    }
    if (blockIdx.x == 0 && tid == 0) { st.pad = err; st_io[1] = st; }
}

static int ode_dopri5_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                            const float* t_span, int n_t, float atol, float rtol, float* traj,
                            int* n_steps, int* nfe, float* xbuf, float* kbuf, float* tspan_dev,
                            void* state_dev, char* sync_dev, float t0, float dt0, int evals0, hipStream_t s) {
    SmArgs A;
    for (int l = 0; l < 4; ++l) { A.W[l] = W[l]; A.b[l] = b[l]; }
    for (int l = 0; l < 5; ++l) A.dims[l] = dims[l];
    const size_t lds = sizeof(float) * (4 * SM_W * SM_LD + 4 * SM_W + SM_W + 2 * SM_ROWS * SM_LD);
    static int raised_d[CFM_MAX_DEVICES], resident_d[CFM_MAX_DEVICES];
    static std::once_flag once_d[CFM_MAX_DEVICES];
    const int dvi = cfm_device_index();
    int& raised = raised_d[dvi]; int& resident = resident_d[dvi];
    std::call_once(once_d[dvi], [lds, &raised, &resident] {
        hipError_t e = hipFuncSetAttribute((const void*)ode_small_dopri<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipError_t e2 = hipFuncSetAttribute((const void*)ode_small_dopri<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        int ok = (e == hipSuccess && e2 == hipSuccess) ? 1 : -1;
        // workgroups that can be resident at once: the grid rendezvous needs grid <= this
        int dev = 0, cus = 0, pa = 0, pb = 0;
        if (ok > 0 && hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&pa, (const void*)ode_small_dopri<true>, 256, lds) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&pb, (const void*)ode_small_dopri<false>, 256, lds) == hipSuccess &&
            cus > 0 && pa > 0 && pb > 0)
            resident = cus * (pa < pb ? pa : pb);
        else
            ok = -1;
        (void)hipGetLastError();
        raised = ok;
    });
    if (raised < 0) return CFM_EINVAL;
    int rc = cfm_hip(hipMemcpyAsync(tspan_dev, t_span, sizeof(float) * n_t, hipMemcpyHostToDevice, s));
    if (rc) return rc;
    SmState h[2];
    memset(h, 0, sizeof(h));
    h[0].t = t0; h[0].dt = dt0; h[0].ckpt = 1; h[0].steps = 0; h[0].evals = evals0; h[0].par = 0; h[0].done = 0;
    rc = cfm_hip(hipMemcpyAsync(state_dev, h, sizeof(h), hipMemcpyHostToDevice, s));
    if (rc) return rc;
    double* partial = (double*)sync_dev;
    unsigned long long* lines = (unsigned long long*)(sync_dev + 3 * SM_MAXGRID * 8);
    rc = cfm_hip(hipMemsetAsync(partial, 0xFF, 3 * SM_MAXGRID * 8, s));   // every slot: negative NaN = not arrived
    if (rc) return rc;
    SmState* st = (SmState*)state_dev;
    const int tiles = (B + SM_ROWS - 1) / SM_ROWS;
    int grid = tiles < resident ? tiles : resident;
    if (grid > SM_MAXGRID) grid = SM_MAXGRID;
    // One persistent solve at a time per process: two of them launched from two streams could each get only part
    // of their workgroups resident and wait for the rest forever (the bounded wait would turn that into
    // CFM_ETIMEOUT after 4 s).  The call is synchronous anyway: the lock is held until the solve has finished.
    static std::mutex persistent_mu;
    std::lock_guard<std::mutex> persistent_lock(persistent_mu);
    if (tiles <= grid)
        hipLaunchKernelGGL(ode_small_dopri<true>, dim3(grid), dim3(256), lds, s, A, B, d, st, xbuf, kbuf, tspan_dev, n_t, atol,
                           rtol, traj, partial, lines, 1000000);
    else
        hipLaunchKernelGGL(ode_small_dopri<false>, dim3(grid), dim3(256), lds, s, A, B, d, st, xbuf, kbuf, tspan_dev, n_t, atol,
                           rtol, traj, partial, lines, 1000000);
    rc = cfm_status();
    if (rc) return rc;
    SmState cur;
    rc = cfm_hip(hipMemcpyAsync(&cur, st + 1, sizeof(SmState), hipMemcpyDeviceToHost, s));
    if (rc) return rc;
    rc = cfm_hip(hipStreamSynchronize(s));
    if (rc) return rc;
    if (n_steps) *n_steps = cur.steps;
    if (nfe) *nfe = cur.evals;
    if (cur.pad == 2) return CFM_ETIMEOUT;
    return cur.done && cur.pad == 0 ? 0 : CFM_ENOCONV;
}

// ------------------------------------------------------------ SF2M: Euler-Maruyama ----
// y <- y + h (v(te, y) + s(te, y)) + g sqrt|h| xi  for every step of the grid, the whole trajectory of a tile in ONE
// launch: both small fields (flow v and score s: 4 layers, widths <= 64) live in LDS (2 x 69 KB), the state in
// registers.  Same arithmetic, in the same order, as the launch-per-step scheme (sde.py: two forward passes on
// mlp_layer + cfm_sde_em_step_f32: f = fma(1, s, +-v); r = fma(h, f, y); r = fma(g sqrt|h|, xi, r)) — with the
// caller's noise (xi != NULL) the trajectory is bit-equal to it.  xi == NULL: N(0, 1) from Philox4x32-10 in the
// kernel (counter = step, element index; key = seed), Box-Muller.
// Replaces torchsde.sdeint(SDE(model, score_model), x0, ts, method="euler", dt=...) of SF2M_tutorial.ipynb cell 5 and
// runner/src/models/components/solver.py:157-182 for the small fields those examples train.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned step, unsigned long long elem, float (&z)[4]) {
    unsigned c[4] = {(unsigned)elem, (unsigned)(elem >> 32), step, 0x5f2du};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    // Box-Muller on (0, 1] uniforms
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(c[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u1, &s0, &c0); sincosf(6.283185307179586f * u3, &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}

struct EmStep { float te, h, gs; int is_out; };      // per step: field time, step, g sqrt|h|, trajectory point after it

__global__ __launch_bounds__(256) void ode_small_em(SmArgs F, SmArgs S, int has_s, int B, int d,
                                                    const EmStep* __restrict__ steps, int n_steps, int reverse,
                                                    const float* __restrict__ xi, unsigned long long seed,
                                                    const float* __restrict__ y0, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    float* WlF = small_lds;
    float* blF = WlF + 4 * SM_W * SM_LD;
    float* wtF = blF + 4 * SM_W;
    float* WlS = wtF + SM_W;
    float* blS = WlS + 4 * SM_W * SM_LD;
    float* wtS = blS + 4 * SM_W;
    float* Ab0 = wtS + SM_W;
    float* Ab1 = Ab0 + SM_ROWS * SM_LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    sm_stage_weights(F, d, WlF, blF, wtF, tid);
    if (has_s) sm_stage_weights(S, d, WlS, blS, wtS, tid);
    const size_t n = (size_t)B * d;
    const int col = wv * 16 + (lane & 15);
    for (int row0 = blockIdx.x * SM_ROWS; row0 < B; row0 += gridDim.x * SM_ROWS) {
        SmTile x;
#pragma unroll
        for (int i = 0; i < SM_V; ++i) {
            const int gr = row0 + sm_row(i, lane);
            x.v[i] = (gr < B && col < d) ? y0[(size_t)gr * d + col] : 0.f;
        }
        __syncthreads();
        int oidx = 0;
        for (int k = 0; k < n_steps; ++k) {
            const EmStep st = steps[k];
            const SmTile v = sm_field(x, st.te, F, d, Ab0, Ab1, WlF, blF, wtF, wv, lane);
            SmTile sc;
            if (has_s) sc = sm_field(x, st.te, S, d, Ab0, Ab1, WlS, blS, wtS, wv, lane);
            static_assert(SM_V == 4, "ode_small_em draws ONE Philox block of 4 normals per lane and step: with SM_MB > 1 "
                                     "elements i and i + 4 would share a normal (draw SM_V / 4 blocks, counter word + (i >> 2))");
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (!xi && st.gs != 0.f) philox_normal4(seed, (unsigned)k, (unsigned long long)(row0 / SM_ROWS) * 256 + tid, z);
#pragma unroll
            for (int i = 0; i < SM_V; ++i) {
                const int gr = row0 + sm_row(i, lane);
                const bool ok = gr < B && col < d;
                float f = reverse ? -v.v[i] : v.v[i];
                if (has_s) f = fmaf(1.0f, sc.v[i], f);
                float r = fmaf(st.h, f, x.v[i]);
                const float noise = xi ? (ok ? xi[(size_t)k * n + (size_t)gr * d + col] : 0.f) : z[i & 3];
                r = fmaf(st.gs, noise, r);
                x.v[i] = ok ? r : 0.f;
                if (st.is_out && ok) out[(size_t)oidx * n + (size_t)gr * d + col] = r;
            }
            oidx += st.is_out ? 1 : 0;
        }
        __syncthreads();
    }
}

// steps_host: n_steps records {te, h, g sqrt|h|, is_out} (host); ws: >= 16 n_steps bytes of device scratch.
// Ws == NULL: no score field.  Returns CFM_EINVAL for anything but two 4-layer fields of widths <= 64 with a time
// column (the caller then steps launch by launch).
extern "C" int cfm_sde_em_mlp_f32(const float* const* Wf, const float* const* bf, const float* const* Ws,
                                  const float* const* bs, const int* dims, int n_layers, const float* y0, int B,
                                  const void* steps_host, int n_steps, int reverse, const float* xi,
                                  unsigned long long seed, float* out, void* ws, void* stream) {
    if (!Wf || !bf || !dims || !y0 || !out || !steps_host || !ws || B < 0 || n_steps < 0) return CFM_EINVAL;
    if (n_layers != 4) return CFM_EINVAL;
    const int d = dims[4];
    if (dims[0] != d + 1 || d > SM_W) return CFM_EINVAL;
    for (int l = 1; l <= 3; ++l) if (dims[l] > SM_W || dims[l] < 1) return CFM_EINVAL;
    if (B == 0 || n_steps == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    SmArgs F, S;
    for (int l = 0; l < 4; ++l) { F.W[l] = Wf[l]; F.b[l] = bf[l]; S.W[l] = Ws ? Ws[l] : Wf[l]; S.b[l] = bs ? bs[l] : bf[l]; }
    for (int l = 0; l < 5; ++l) { F.dims[l] = dims[l]; S.dims[l] = dims[l]; }
    const size_t lds = sizeof(float) * (2 * (4 * SM_W * SM_LD + 4 * SM_W + SM_W) + 2 * SM_ROWS * SM_LD);
    static int raised_d[CFM_MAX_DEVICES];
    static std::once_flag once_d[CFM_MAX_DEVICES];
    const int dvi = cfm_device_index();
    int& raised = raised_d[dvi];
    std::call_once(once_d[dvi], [&raised] {
        hipError_t e = hipFuncSetAttribute((const void*)ode_small_em, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        raised = (e == hipSuccess) ? 1 : -1;
    });
    if (raised < 0) return CFM_EINVAL;
    int rc = cfm_hip(hipMemcpyAsync(ws, steps_host, sizeof(EmStep) * (size_t)n_steps, hipMemcpyHostToDevice, s));
    if (rc) return rc;
    const int tiles = (B + SM_ROWS - 1) / SM_ROWS;
    hipLaunchKernelGGL(ode_small_em, dim3(tiles < 4096 ? tiles : 4096), dim3(256), lds, s, F, S, (Ws && bs) ? 1 : 0, B, d,
                       (const EmStep*)ws, n_steps, reverse, xi, seed, y0, out);
    return cfm_status();
}

// Fixed-step Euler for the same small fields: x_{k+1} = x_k + dt_k f(t_k, x_k), every step of the
// tile inside one launch (same arithmetic as ode_combine: fmaf(dt, 1.f * k, x)).
__global__ __launch_bounds__(256) void ode_small_euler(SmArgs A, int B, int d, const float* __restrict__ tspan, int n_t,
                                                    float* __restrict__ traj) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    float* Wl = small_lds;
    float* bl = Wl + 4 * SM_W * SM_LD;
    float* wt = bl + 4 * SM_W;
    float* Ab0 = wt + SM_W;
    float* Ab1 = Ab0 + SM_ROWS * SM_LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    sm_stage_weights(A, d, Wl, bl, wt, tid);
    const size_t n = (size_t)B * d;
    const int col = wv * 16 + (lane & 15);
    for (int row0 = blockIdx.x * SM_ROWS; row0 < B; row0 += gridDim.x * SM_ROWS) {
        SmTile x;
#pragma unroll
        for (int i = 0; i < SM_V; ++i) {
            const int gr = row0 + sm_row(i, lane);
            x.v[i] = (gr < B && col < d) ? traj[(size_t)gr * d + col] : 0.f;
        }
        __syncthreads();
        for (int k = 0; k + 1 < n_t; ++k) {
            const float t = tspan[k], dt = tspan[k + 1] - tspan[k];
            const SmTile f = sm_field(x, t, A, d, Ab0, Ab1, Wl, bl, wt, wv, lane);
#pragma unroll
            for (int i = 0; i < SM_V; ++i) {
                x.v[i] = fmaf(dt, 1.f * f.v[i], x.v[i]);
                const int gr = row0 + sm_row(i, lane);
                if (gr < B && col < d) traj[(size_t)(k + 1) * n + (size_t)gr * d + col] = x.v[i];
            }
        }
    }
}

static int ode_euler_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                           const float* t_span, int n_t, float* traj, float* tspan_dev, hipStream_t s) {
    SmArgs A;
    for (int l = 0; l < 4; ++l) { A.W[l] = W[l]; A.b[l] = b[l]; }
    for (int l = 0; l < 5; ++l) A.dims[l] = dims[l];
    const size_t lds = sizeof(float) * (4 * SM_W * SM_LD + 4 * SM_W + SM_W + 2 * SM_ROWS * SM_LD);
    static int raised_d[CFM_MAX_DEVICES];
    static std::once_flag once_d[CFM_MAX_DEVICES];
    const int dvi = cfm_device_index();
    int& raised = raised_d[dvi];
    std::call_once(once_d[dvi], [&raised] {
        hipError_t e = hipFuncSetAttribute((const void*)ode_small_euler, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipGetLastError();
        raised = (e == hipSuccess) ? 1 : -1;
    });
    if (raised < 0) return CFM_EINVAL;
    int rc = cfm_hip(hipMemcpyAsync(tspan_dev, t_span, sizeof(float) * n_t, hipMemcpyHostToDevice, s));
    if (rc) return rc;
    const int tiles = (B + SM_ROWS - 1) / SM_ROWS;
    hipLaunchKernelGGL(ode_small_euler, dim3(tiles < 4096 ? tiles : 4096), dim3(256), lds, s, A, B, d, tspan_dev, n_t, traj);
    return cfm_status();
}
