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

extern "C" size_t cfm_ode_ws_bytes_internal(int B, int width, int d) {
    return cfm_mlp_ws_bytes_internal(B, width) + sizeof(float) * (size_t)B * d * 10 + 512;
}

struct OdeWs {
    float* act;       // MLP activations
    float* k[7];
    float* x; float* xn; float* xt;
    double* red;      // 8 doubles
};

static OdeWs ode_carve(void* ws, int B, int width, int d) {
    OdeWs w; char* q = (char*)ws;
    w.red = (double*)q; q += 256;
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
                            void* state_dev, double* red_dev, float t0, float dt0, int evals0, hipStream_t s);

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
                                (void*)(w.red + 16), w.red + 6, t, dt, evals, s);

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
// Rows are independent inside a step (only the error norm couples them), so here ONE
// kernel does the whole attempt: a workgroup keeps the four weight matrices in LDS
// (66 KB), owns 64 rows, holds x and k1..k7 of its tile in MFMA accumulator layout in
// registers, and runs the six stage evaluations back to back (v_mfma_f32_32x32x2_f32, the
// same instruction, k order and epilogue as mlp_layer: bitwise the same field values).
// A second small kernel turns the summed error into the accept / reject decision and the
// next step size ON THE DEVICE (same fp32 controller as the host loop above) and copies
// accepted t_span landings into the trajectory; the host only pumps (step, control) pairs
// and polls a 32-byte state block.
// =====================================================================================
#define SM_W 64
#define SM_LD 65

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// one layer on the 64-row tile: out(C layout) = A[64 x K] * W_l[64 x K]^T
__device__ __forceinline__ f32x16 sm_gemm(const float* __restrict__ Abuf, const float* __restrict__ Wl, int K,
                                          int wm, int wn, int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int fr = lane & 31, fk = lane >> 5;
    const float* ap = Abuf + (wm * 32 + fr) * SM_LD + fk;
    const float* bp = Wl + (wn * 32 + fr) * SM_LD + fk;
    // software pipeline: the operands of k-step kk+2 are read from LDS while the MFMA of kk runs (the read
    // past the last step stays inside the padded 65-float row)
    const int Kp = (K + 1) & ~1;
    float a = ap[0], b = bp[0];
    for (int kk = 0; kk < Kp; kk += 2) {
        const float an = ap[kk + 2], bn = bp[kk + 2];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        a = an; b = bn;
    }
    return acc;
}

// f(t, y) for the tile; y arrives in C layout, the result leaves in C layout (columns >= d are 0)
__device__ __forceinline__ f32x16 sm_field(const f32x16& y, float t, const SmArgs& A, int d, float* Abuf0,
                                           float* Abuf1, const float* Wl, const float* bl, const float* wt,
                                           int wm, int wn, int lane) {
    const int col = wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        Abuf0[row * SM_LD + col] = (col < d) ? y[r] : 0.f;
    }
    __syncthreads();
    f32x16 acc;
    float* src = Abuf0; float* dst = Abuf1;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const int K = (l == 0) ? d : A.dims[l];
        const int N = A.dims[l + 1];
        acc = sm_gemm(src, Wl + l * SM_W * SM_LD, K, wm, wn, lane);
        const float bv = (col < N) ? bl[l * SM_W + col] : 0.f;
        const float wtc = (l == 0 && col < N) ? wt[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r] + bv;
            if (l == 0) v = fmaf(t, wtc, v);
            if (l < 3) v = selu_f(v);
            acc[r] = (col < N) ? v : 0.f;
        }
        if (l < 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[row * SM_LD + col] = acc[r];
            }
            __syncthreads();
            float* tmp = src; src = dst; dst = tmp;
        }
    }
    return acc;
}

__global__ __launch_bounds__(256) void ode_small_step(SmArgs A, int B, int d, const SmState* __restrict__ st_all,
                                                   int attempt, float* __restrict__ xbuf, float* __restrict__ kbuf,
                                                   const float* __restrict__ tspan, int n_t, float atol, float rtol,
                                                   double* __restrict__ red) {
    extern __shared__ __attribute__((aligned(16))) float small_lds[];
    float* Wl = small_lds;                              // [4][64][65]
    float* bl = Wl + 4 * SM_W * SM_LD;               // [4][64]
    float* wt = bl + 4 * SM_W;                       // [64] time column of layer 0
    float* Ab0 = wt + SM_W;                          // [64][65]
    float* Ab1 = Ab0 + SM_W * SM_LD;
    const SmState st = st_all[attempt & 1];
    if (st.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    // weights -> LDS (zero padded)
    for (int l = 0; l < 4; ++l) {
        const int in_l = A.dims[l], out_l = A.dims[l + 1];
        const int K = (l == 0) ? d : in_l;
        for (int e = tid; e < SM_W * SM_W; e += 256) {
            const int r = e / SM_W, k = e % SM_W;
            Wl[(l * SM_W + r) * SM_LD + k] = (r < out_l && k < K) ? A.W[l][(size_t)r * in_l + k] : 0.f;
        }
        if (tid < SM_W) bl[l * SM_W + tid] = (tid < out_l) ? A.b[l][tid] : 0.f;
    }
    if (tid < SM_W) wt[tid] = (tid < A.dims[1]) ? A.W[0][(size_t)tid * A.dims[0] + d] : 0.f;

    float dt, dt_old; bool flag, lands;
    sm_prestep(st, tspan, n_t, dt, dt_old, flag, lands);
    const size_t n = (size_t)B * d;
    const float* x_in = xbuf + (size_t)st.par * n;
    const float* k1_in = kbuf + (size_t)st.par * n;
    float* x_out = xbuf + (size_t)(st.par ^ 1) * n;
    float* k7_out = kbuf + (size_t)(st.par ^ 1) * n;
    const int col = wn * 32 + (lane & 31);
    double esum = 0.0;
    for (int row0 = blockIdx.x * SM_W; row0 < B; row0 += gridDim.x * SM_W) {
        f32x16 x, k0, k1, k2, k3, k4, k5, k6, y;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bool ok = gr < B && col < d;
            x[r] = ok ? x_in[(size_t)gr * d + col] : 0.f;
            k0[r] = ok ? k1_in[(size_t)gr * d + col] : 0.f;
        }
        __syncthreads();                              // weights staged / previous tile done with the buffers
        // stage S (a literal): y = x + dt * sum_{q<=S} a[S][q] k_q ; KOUT = f(t + c[S] dt, y)
#define SM_STAGE(S, KOUT)                                                                            \
        {                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
                float acc = (float)DP_A_dev(S, 0) * k0[r];                                           \
                if (S >= 1) acc = fmaf((float)DP_A_dev(S, 1), k1[r], acc);                           \
                if (S >= 2) acc = fmaf((float)DP_A_dev(S, 2), k2[r], acc);                           \
                if (S >= 3) acc = fmaf((float)DP_A_dev(S, 3), k3[r], acc);                           \
                if (S >= 4) acc = fmaf((float)DP_A_dev(S, 4), k4[r], acc);                           \
                if (S >= 5) acc = fmaf((float)DP_A_dev(S, 5), k5[r], acc);                           \
                y[r] = fmaf(dt, acc, x[r]);                                                          \
            }                                                                                        \
            KOUT = sm_field(y, st.t + DP_C_dev(S) * dt, A, d, Ab0, Ab1, Wl, bl, wt, wm, wn, lane);  \
        }
        SM_STAGE(0, k1) SM_STAGE(1, k2) SM_STAGE(2, k3) SM_STAGE(3, k4) SM_STAGE(4, k5) SM_STAGE(5, k6)
#undef SM_STAGE
        // y is x_new (the 5th-order solution), k6 = f(t + dt, x_new): error + outputs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr < B && col < d) {
                float e = (float)DP_E_dev(0) * k0[r];
                e = fmaf((float)DP_E_dev(1), k1[r], e);
                e = fmaf((float)DP_E_dev(2), k2[r], e);
                e = fmaf((float)DP_E_dev(3), k3[r], e);
                e = fmaf((float)DP_E_dev(4), k4[r], e);
                e = fmaf((float)DP_E_dev(5), k5[r], e);
                e = fmaf((float)DP_E_dev(6), k6[r], e);
                e *= dt;
                const float sc = atol + rtol * fmaxf(fabsf(x[r]), fabsf(y[r]));
                const float rr = e / sc;
                esum += (double)rr * (double)rr;
                x_out[(size_t)gr * d + col] = y[r];
                k7_out[(size_t)gr * d + col] = k6[r];
            }
        }
    }
    esum = wave_sum_d(esum);
    __shared__ double redw[4];
    if (lane == 0) redw[wv] = esum;
    __syncthreads();
    if (tid == 0) atomicAdd(&red[attempt & 1], redw[0] + redw[1] + redw[2] + redw[3]);
}

// accept / reject, next step size, trajectory landing (every workgroup derives the same decision
// from the same inputs; workgroup 0 publishes the next state)
__global__ __launch_bounds__(256) void ode_small_ctrl(int B, int d, SmState* __restrict__ st_all, int attempt,
                                                   const float* __restrict__ xbuf, const float* __restrict__ tspan,
                                                   int n_t, float* __restrict__ traj, double* __restrict__ red) {
    const SmState st = st_all[attempt & 1];
    if (st.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) st_all[(attempt + 1) & 1] = st;
        return;
    }
    float dt, dt_old; bool flag, lands;
    sm_prestep(st, tspan, n_t, dt, dt_old, flag, lands);
    const size_t n = (size_t)B * d;
    const float ratio = (float)sqrt(red[attempt & 1] / (double)n);
    const bool accept = ratio <= 1.f;
    if (accept && lands) {
        const float* xn = xbuf + (size_t)(st.par ^ 1) * n;
        float* dst = traj + (size_t)st.ckpt * n;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = xn[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
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
        nx.done = (nx.t < tspan[n_t - 1]) ? 0 : 1;
        st_all[(attempt + 1) & 1] = nx;
        red[(attempt + 1) & 1] = 0.0;
    }
}

static int ode_dopri5_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                            const float* t_span, int n_t, float atol, float rtol, float* traj,
                            int* n_steps, int* nfe, float* xbuf, float* kbuf, float* tspan_dev,
                            void* state_dev, double* red_dev, float t0, float dt0, int evals0, hipStream_t s) {
    SmArgs A;
    for (int l = 0; l < 4; ++l) { A.W[l] = W[l]; A.b[l] = b[l]; }
    for (int l = 0; l < 5; ++l) A.dims[l] = dims[l];
    const size_t lds = sizeof(float) * (4 * SM_W * SM_LD + 4 * SM_W + SM_W + 2 * SM_W * SM_LD);
    static int raised = 0;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)ode_small_step, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        raised = (e == hipSuccess) ? 1 : -1;
        (void)hipGetLastError();
    }
    if (raised < 0) return CFM_EINVAL;
    int rc = cfm_hip(hipMemcpyAsync(tspan_dev, t_span, sizeof(float) * n_t, hipMemcpyHostToDevice, s));
    if (rc) return rc;
    SmState h[2];
    memset(h, 0, sizeof(h));
    h[0].t = t0; h[0].dt = dt0; h[0].ckpt = 1; h[0].steps = 0; h[0].evals = evals0; h[0].par = 0; h[0].done = 0;
    rc = cfm_hip(hipMemcpyAsync(state_dev, h, sizeof(h), hipMemcpyHostToDevice, s));
    if (rc) return rc;
    rc = cfm_hip(hipMemsetAsync(red_dev, 0, 2 * sizeof(double), s));
    if (rc) return rc;
    SmState* st = (SmState*)state_dev;
    int tiles = (B + SM_W - 1) / SM_W;
    const int grid = tiles < 1024 ? tiles : 1024;
    int copy_grid = (int)(((size_t)B * d + 255) / 256); if (copy_grid > 128) copy_grid = 128; if (copy_grid < 1) copy_grid = 1;
    int attempt = 0;
    const int max_attempts = 1000000;
    SmState cur;
    for (;;) {
        const int chunk = (attempt == 0) ? (n_t - 1) + 4 : 8;
        for (int c = 0; c < chunk; ++c, ++attempt) {
            hipLaunchKernelGGL(ode_small_step, dim3(grid), dim3(256), lds, s, A, B, d, st, attempt, xbuf, kbuf, tspan_dev,
                               n_t, atol, rtol, red_dev);
            hipLaunchKernelGGL(ode_small_ctrl, dim3(copy_grid), dim3(256), 0, s, B, d, st, attempt, xbuf, tspan_dev, n_t,
                               traj, red_dev);
        }
        rc = cfm_status();
        if (rc) return rc;
        rc = cfm_hip(hipMemcpyAsync(&cur, st + (attempt & 1), sizeof(SmState), hipMemcpyDeviceToHost, s));
        if (rc) return rc;
        rc = cfm_hip(hipStreamSynchronize(s));
        if (rc) return rc;
        if (cur.done) break;
        if (attempt >= max_attempts) return CFM_ENOCONV;
    }
    if (n_steps) *n_steps = cur.steps;
    if (nfe) *nfe = cur.evals;
    return 0;
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
    float* Ab1 = Ab0 + SM_W * SM_LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    for (int l = 0; l < 4; ++l) {
        const int in_l = A.dims[l], out_l = A.dims[l + 1];
        const int K = (l == 0) ? d : in_l;
        for (int e = tid; e < SM_W * SM_W; e += 256) {
            const int r = e / SM_W, k = e % SM_W;
            Wl[(l * SM_W + r) * SM_LD + k] = (r < out_l && k < K) ? A.W[l][(size_t)r * in_l + k] : 0.f;
        }
        if (tid < SM_W) bl[l * SM_W + tid] = (tid < out_l) ? A.b[l][tid] : 0.f;
    }
    if (tid < SM_W) wt[tid] = (tid < A.dims[1]) ? A.W[0][(size_t)tid * A.dims[0] + d] : 0.f;
    const size_t n = (size_t)B * d;
    const int col = wn * 32 + (lane & 31);
    for (int row0 = blockIdx.x * SM_W; row0 < B; row0 += gridDim.x * SM_W) {
        f32x16 x;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            x[r] = (gr < B && col < d) ? traj[(size_t)gr * d + col] : 0.f;
        }
        __syncthreads();
        for (int k = 0; k + 1 < n_t; ++k) {
            const float t = tspan[k], dt = tspan[k + 1] - tspan[k];
            const f32x16 f = sm_field(x, t, A, d, Ab0, Ab1, Wl, bl, wt, wm, wn, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x[r] = fmaf(dt, 1.f * f[r], x[r]);
                const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gr < B && col < d) traj[(size_t)(k + 1) * n + (size_t)gr * d + col] = x[r];
            }
        }
    }
}

static int ode_euler_small(const float* const* W, const float* const* b, const int* dims, int B, int d,
                           const float* t_span, int n_t, float* traj, float* tspan_dev, hipStream_t s) {
    SmArgs A;
    for (int l = 0; l < 4; ++l) { A.W[l] = W[l]; A.b[l] = b[l]; }
    for (int l = 0; l < 5; ++l) A.dims[l] = dims[l];
    const size_t lds = sizeof(float) * (4 * SM_W * SM_LD + 4 * SM_W + SM_W + 2 * SM_W * SM_LD);
    static int raised = 0;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute((const void*)ode_small_euler, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        raised = (e == hipSuccess) ? 1 : -1;
        (void)hipGetLastError();
    }
    if (raised < 0) return CFM_EINVAL;
    int rc = cfm_hip(hipMemcpyAsync(tspan_dev, t_span, sizeof(float) * n_t, hipMemcpyHostToDevice, s));
    if (rc) return rc;
    const int tiles = (B + SM_W - 1) / SM_W;
    hipLaunchKernelGGL(ode_small_euler, dim3(tiles < 2048 ? tiles : 2048), dim3(256), lds, s, A, B, d, tspan_dev, n_t, traj);
    return cfm_status();
}
