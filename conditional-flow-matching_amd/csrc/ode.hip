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
