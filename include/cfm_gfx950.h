/*
 * cfm_gfx950.h — C ABI of libcfm_gfx950.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the minibatch-OT coupling + CFM sampling hot path of
 * TorchCFM (atong01/conditional-flow-matching).  The reference has no FFI of
 * its own (it is pure Python); each entry point below replaces the third-party
 * native routine or eager-op chain that the cited reference line executes.
 * Citations are relative to the reference repository root.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (contiguous,
 *     row-major) unless the parameter is documented "host";
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the
 *     call returns without synchronising, EXCEPT cfm_assign_exact_f32,
 *     cfm_assign_exact_batch_f32 and cfm_ode_*_mlp_f32 whose control flow is data dependent: they pump their
 *     step kernels on `stream` and poll a few bytes of device state, so they
 *     return only when the result is resident in the output buffers;
 *   - no entry point allocates or frees device memory: scratch comes from the
 *     caller through `ws` (size from cfm_workspace_bytes);
 *   - return value: 0 = ok, <0 = invalid argument (CFM_E*), >0 = hipError_t;
 *   - no exceptions cross the boundary; no global state except a lazily
 *     initialised per-device properties cache, per HOST THREAD the captured launch
 *     programs (hipGraphs) of the exact solver's last four (workspace, size, batch,
 *     stream) combinations + 2 KiB of pinned memory, and the solver-parameter table
 *     of include/cfm_gfx950_tuning.h (measurement / tuning exports that no
 *     binding needs; a solve snapshots the table under a mutex when it starts).
 */
#ifndef CFM_GFX950_H
#define CFM_GFX950_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header (and cfm_gfx950_tuning.h) declares leaves it. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define CFM_ABI_VERSION 3

/* error codes (negative) */
#define CFM_EINVAL   (-1)  /* bad shape / null pointer / unsupported size      */
#define CFM_EALIGN   (-2)  /* pointer not aligned as documented                */
#define CFM_ENOCONV  (-3)  /* solver hit its step cap without a certificate    */
#define CFM_ETIMEOUT (-4)  /* device state machine made no progress            */

/* ops for cfm_workspace_bytes */
#define CFM_OP_SINKHORN      1
#define CFM_OP_ASSIGN        2
#define CFM_OP_SAMPLE_DENSE  3
#define CFM_OP_MLP           4
#define CFM_OP_ODE           5
#define CFM_OP_UNBALANCED    6   /* also the partial (Dykstra) solver */
#define CFM_OP_COST          7   /* cfm_sqeuclid_cost_ws_f32 (B0, B1, d)  */
#define CFM_OP_MLP_TRAIN     8   /* cfm_mlp_backward_f32 (B, widest layer, largest weight's element count) */
#define CFM_OP_TRANSPORT     9   /* cfm_transport_exact_f32 (B0, B1, 0) */

/* variants for cfm_sample_xt_ut_f32 (reference class in parentheses) */
#define CFM_VARIANT_ICFM   0  /* ConditionalFlowMatcher / ExactOT...          */
#define CFM_VARIANT_SB     1  /* SchrodingerBridgeConditionalFlowMatcher      */
#define CFM_VARIANT_TARGET 2  /* TargetConditionalFlowMatcher                 */
#define CFM_VARIANT_VP     3  /* VariancePreservingConditionalFlowMatcher     */

int cfm_abi_version(void);

/* Bytes of device scratch an op needs for the given problem (0 on bad op). */
size_t cfm_workspace_bytes(int op, int B0, int B1, int d);

/* Runtime — a HIP stream restricted to a subset of the chip's compute units (hipExtStreamCreateWithCUMask).
 * The reference overlaps nothing: its coupling runs on the host between two model steps
 * (torchcfm/conditional_flow_matching.py:271-272, examples/images/cifar10/train_cifar10.py:141-151).  Here the
 * couplings of the next minibatches run beside the model step (cfm_amd.prefetch); the latency-bound rounds of the
 * exact solver and the dense fp32-MFMA products then fight for workgroup slots on every CU unless the chip is
 * PARTITIONED: solver streams on one CU subset, dense streams on the complement.
 * cu_mask: HOST array of n_words 32-bit words; bit i = logical CU i of the current device (gfx950 in SPX mode:
 * XCD i % 8, CU i / 8 of that XCD); every XCD must keep at least one CU (CFM_EINVAL otherwise: a queue without
 * CUs on one XCD never drains).  *stream receives a hipStream_t the caller owns (cfm_stream_destroy). */
int cfm_stream_create_cu_mask(const uint32_t* cu_mask, int n_words, void** stream);
int cfm_stream_destroy(void* stream);

/* K1 — squared-Euclidean cost matrix  M[i,j] = sum_k (x0[i,k]-x1[j,k])^2.
 * Replaces  torch.cdist(x0, x1) ** 2       torchcfm/optimal_transport.py:84
 * (also :176, :297 with power 1 -> see cfm_euclid_cost_f32).
 * If opt_max != NULL it receives max(M) (one float; for normalize_cost, :85-86).
 * x0 [B0,d], x1 [B1,d], M [B0,B1], fp32. */
int cfm_sqeuclid_cost_f32(const float* x0, const float* x1, int B0, int B1, int d,
                          float* M, float* opt_max, void* stream);

/* K1, matrix-core form — same contract and result class as cfm_sqeuclid_cost_f32, with scratch:
 * for d >= 64 and B0, B1 >= 256 the Gram form |a|^2 + |b|^2 - 2<a,b> of the points centred on a
 * sample mean runs on v_mfma_f32_32x32x2_f32, and every entry where that form cancels (result
 * below 1/8 of |a|^2 + |b|^2: near-duplicates, an x-vs-x diagonal) is recomputed in the
 * direct-difference form; other shapes take the direct kernels of cfm_sqeuclid_cost_f32.
 * ws: cfm_workspace_bytes(CFM_OP_COST,B0,B1,d) bytes, 16-byte aligned. */
int cfm_sqeuclid_cost_ws_f32(const float* x0, const float* x1, int B0, int B1, int d,
                             float* M, float* opt_max, void* ws, void* stream);

/* M[i] *= 1/(*maxval)   — `M / M.max()`     torchcfm/optimal_transport.py:86 */
int cfm_scale_inv_f32(float* M, size_t n, const float* maxval, void* stream);

/* In-place sqrt — Euclidean (power=1) cost for wasserstein(), :297-299. */
int cfm_sqrt_inplace_f32(float* M, size_t n, void* stream);

/* K5 — log-domain Sinkhorn with POT loop semantics (uniform marginals).
 * Replaces  pot.sinkhorn(a, b, M, reg)      torchcfm/optimal_transport.py:51,87
 * in its numerically stable form (POT method="sinkhorn_log"): u0 = 0,
 *   v = log b - LSE_i(-M/reg + u),  u = log a - LSE_j(-M/reg + v),
 * marginal check every `check_every` iterations (ii % check_every == 0):
 * err = || sum_i exp(-M/reg + u + v) - b ||_2 ; stop when err < stop_thr or
 * after max_iter iterations (reg, stop_thr are doubles: the reference passes
 * Python floats to POT).  The iteration runs with fp32 exp() (HBM-bound fast
 * path) and switches itself to fp64 exp once the measured violation approaches
 * the fp32 noise floor, so stop_thr = 1e-9 is honoured.  Outputs f = reg*u [B0],
 * g = reg*v [B1] (fp32), *iters_done, *last_err (device scalars).
 * ws: cfm_workspace_bytes(CFM_OP_SINKHORN,B0,B1,0) bytes, 16-byte aligned;
 * it keeps the fp64 potentials (see cfm_sinkhorn_potentials_f64). */
int cfm_sinkhorn_log_f32(const float* M, int B0, int B1, double reg, int max_iter,
                         double stop_thr, int check_every, float* f, float* g,
                         int* iters_done, float* last_err, void* ws, void* stream);

/* K5, variant B — the same solve for low-dimensional clouds (1 <= d <= 8) with the cost entry
 * recomputed from the coordinates inside the LSE passes instead of streamed from the B0 x B1 matrix
 * (cdist at torchcfm/optimal_transport.py:84 + pot.sinkhorn at :51,87 in one: no B^2 traffic).  The
 * entry is formed exactly as cfm_sqeuclid_cost_f32 forms it for d <= 8, so f, g and the potentials in
 * `ws` belong to that matrix and feed cfm_plan_sample_dense / cfm_sinkhorn_plan_f64 / _cost_f64 unchanged.
 * x0 [B0,d], x1 [B1,d]; everything else as cfm_sinkhorn_log_f32 (same ws size and layout). */
int cfm_sinkhorn_log_points_f32(const float* x0, const float* x1, int B0, int B1, int d, double reg,
                                int max_iter, double stop_thr, int check_every, float* f, float* g,
                                int* iters_done, float* last_err, void* ws, void* stream);

/* Copy the fp64 log-scalings u [B0], v [B1] left in `ws` by the last
 * cfm_sinkhorn_log_f32 call on it. */
int cfm_sinkhorn_potentials_f64(const void* ws, int B0, int B1, double* u, double* v,
                                void* stream);

/* Dense plan  pi[i,j] = exp(u_i + v_j - M[i,j]/reg)  in fp64 from the fp64
 * log-scalings in `ws`  — what get_map() hands back to Python,
 * torchcfm/optimal_transport.py:87 (return value of pot.sinkhorn). */
int cfm_sinkhorn_plan_f64(const float* M, int B0, int B1, double reg, const void* ws,
                          double* pi, void* stream);

/* sum_ij pi_ij * M_ij — pot.sinkhorn2 value, torchcfm/optimal_transport.py:288.
 * out: one double. */
int cfm_sinkhorn_cost_f64(const float* M, int B0, int B1, double reg, const void* ws,
                          double* out, void* stream);

/* K5u — unbalanced entropic OT, uniform marginals, kernel space, fp64.
 * Replaces  pot.unbalanced.sinkhorn_knopp_unbalanced(a, b, M, reg, reg_m)
 *                                   torchcfm/optimal_transport.py:52-53,87
 * following the in-repo statement of POT's loop
 * (runner/src/models/components/sinkhorn_knopp_unbalanced.py:113-201, reg_m_1 = reg_m_2):
 *   K = exp(M / -reg);  u = (a / K v)^fi,  v = (b / K^T u)^fi,  fi = reg_m / (reg_m + reg);
 *   numerical error (K^T u == 0, nan, inf) -> the previous (u, v) and stop;
 *   every 10th iteration err = mean of the relative sup-norm changes of u and v; stop at
 *   err <= stop_thr or after max_iter iterations (POT defaults 1000, 1e-6).
 * plan [B0,B1] fp64 receives u_i K_ij v_j (what get_map() returns, :87).
 * info (device int32[4], may be NULL): {iterations, status (1 = numerical error, previous
 * iterate returned), zeros in K, non-finite entries in K}.
 * ws: cfm_workspace_bytes(CFM_OP_UNBALANCED,B0,B1,0) bytes, 16-byte aligned. */
int cfm_unbalanced_sinkhorn_f64(const float* M, int B0, int B1, double reg, double reg_m,
                                int max_iter, double stop_thr, double* plan, int* info,
                                void* ws, void* stream);

/* K5p — entropic partial OT (Dykstra), uniform marginals, transported mass m, fp64.
 * Replaces  pot.partial.entropic_partial_wasserstein(a, b, M, reg)
 *                                   torchcfm/optimal_transport.py:54-55,87
 * (m = min(sum a, sum b) = 1 when the reference calls it; POT defaults numItermax = 1000,
 * stopThr = 1e-100, error ||K_prev - K||_F every 10th iteration).  The three B0 x B1
 * Dykstra corrections are constant along rows / columns / everywhere, so the iteration is
 * carried on two scaling vectors and a scalar over K0 = exp(M / -reg) * m / sum.
 * If K0 has zeros POT's 0/0 poisons the plan with NaN; so does this (info[2] counts them).
 * plan, info, ws as for cfm_unbalanced_sinkhorn_f64. */
int cfm_partial_entropic_f64(const float* M, int B0, int B1, double reg, double m,
                             int max_iter, double stop_thr, double* plan, int* info,
                             void* ws, void* stream);

/* K4 — exact optimal assignment for uniform, equal-size marginals.
 * Replaces  pot.emd(a, b, M)                torchcfm/optimal_transport.py:49,87
 * and       scipy.optimize.linear_sum_assignment(M)                    :179.
 * M [B,B] fp32.  perm[i] = column matched to row i (int32 [B]).
 * *certified (device int) = 1 iff the fp64 dual certificate holds
 * (complementary slackness + dual feasibility to 1e-10*max|M|);
 * *total_cost (device double) = sum_i M[i,perm[i]];
 * stats (device int32[8], may be NULL): {auction_rounds, arr_rounds,
 *  free_rows_after_arr, sap_batches, sap_row_scans, total_row_scans, steps,
 *  eps_phases | multi_source_phases << 8 | dense_fallback_row_scans << 16}.
 * 2 <= B <= 256 (the reference's tutorial batches) is solved by ONE launch of ONE workgroup
 * (cost matrix in registers); stats[7] then has bit 30 set and stats[6] = 1.  A solve that hits
 * that path's round caps is redone by the chip-wide state machine; the call never returns 0
 * with an uncertified permutation (CFM_ENOCONV instead).
 * ws: cfm_workspace_bytes(CFM_OP_ASSIGN,B,B,0) bytes. */
int cfm_assign_exact_f32(const float* M, int B, int* perm, int* certified,
                         double* total_cost, int* stats, void* ws, void* stream);

/* K4, batched — nb independent assignment problems of the same size B in ONE chain of launches.
 * The reference couples one minibatch per call (torchcfm/optimal_transport.py:87, called from
 * conditional_flow_matching.py:271); the solve is a latency-bound chain (~110 dependent launches and a
 * one-workgroup list solver), so the couplings of the next nb minibatches of a training loop — they depend on
 * the data only — cost little more than one when every launch carries all of them (grid.y = problem).
 * Results are those of nb calls of cfm_assign_exact_f32 (same kernels, same state machine per problem).
 * M, perm: HOST arrays of nb device pointers ([B,B] fp32 / int32 [B]); certified [nb], total_cost [nb],
 * stats [nb][8] device arrays (each may be NULL).  nb > 16 is processed in groups of 16.
 * ws: cfm_workspace_bytes(CFM_OP_ASSIGN,B,B,nb) bytes. */
int cfm_assign_exact_batch_f32(const float* const* M, int nb, int B, int* const* perm, int* certified,
                               double* total_cost, int* stats, void* ws, void* stream);

/* K4r — exact OT between uniform marginals of DIFFERENT sizes (masses 1/B0 on the rows, 1/B1 on the columns).
 * Replaces  pot.emd(a, b, M)  for x0.shape[0] != x1.shape[0]   torchcfm/optimal_transport.py:49,79,87
 * The transportation problem on the B0 x B1 matrix itself (every row supplies B1/g units, every column takes B0/g,
 * g = gcd; no lcm x lcm expansion): the primal-dual method with one shortest-path forest into the open columns and a
 * push of ALL open supply down that forest per phase — one workgroup, node state in LDS (matrix and flows too when they
 * fit), then a chip-wide plan / certificate pass.  B0 + B1 <= 2048 (CFM_EINVAL beyond).
 * sigma (device int32[min(B0,B1)], or NULL): optional warm start — an optimal assignment of the rows of the SMALLER side
 * to distinct indices of the larger side (the square solver on the matrix padded with zero rows gives one); it is
 * validated (distinct, in range, its duals must exist) and ignored otherwise: the result never depends on it.  With
 * it, sizes that differ by one (127 vs 128: the case the lcm route cannot take) need ONE phase.
 * plan [B0,B1] fp64 (written whole: zero off the support, units / lcm on it); *total_cost = <plan, M>;
 * info (device int32[8]): {status (1 = optimal and certified in fp64: reduced costs >= -1e-10 max|M|, zero on the
 * support, marginals exact; < 0 = failed, nothing certified), phases, label-correcting sweeps, support size, violations,
 * units per row, units per column of the oriented problem, bit 0: matrix staged in LDS, bit 1: warm start used}.
 * ws: cfm_workspace_bytes(CFM_OP_TRANSPORT,B0,B1,0) bytes, 16-byte aligned.  Asynchronous on `stream`; the caller
 * reads info[0]. */
int cfm_transport_exact_f32(const float* M, int B0, int B1, const int* sigma, double* plan, double* total_cost,
                            int* info, void* ws, void* stream);

/* K6 (exact path) — draw n index pairs from the permutation plan.
 * Replaces sample_map()                      torchcfm/optimal_transport.py:116-121
 * for pi = P_perm / B:  np.random.choice over the flattened plan consumes n
 * uniforms u01 (drawn by the caller from np.random) and returns, per draw,
 * the first flat index whose cdf exceeds u:  i = floor(u*B) (cdf steps k/B),
 * j = perm[i].  u01: device double[n];  i,j: device int64[n]. */
int cfm_plan_sample_perm(const int* perm, const double* u01, int B, int n,
                         int64_t* i, int64_t* j, void* stream);

/* K6 (entropic path) — draw n index pairs from the dense Sinkhorn plan without
 * materialising it: fp64 row sums of exp(u_i+v_j-M_ij/reg), row scan, inverse
 * cdf in flattened (row-major) order, same searchsorted(side="right")
 * semantics as np.random.choice              torchcfm/optimal_transport.py:116-121.
 * ws: cfm_workspace_bytes(CFM_OP_SAMPLE_DENSE,B0,B1,0); `sk_ws` is the Sinkhorn
 * workspace holding u,v. */
int cfm_plan_sample_dense(const float* M, int B0, int B1, double reg, const void* sk_ws,
                          const double* u01, int n, int64_t* i, int64_t* j, void* ws,
                          void* stream);

/* Same draw from an explicit fp64 plan pi [B0,B1] (API path sample_map(pi,...)
 * with a caller-supplied plan; entries of `pi` may be zeroed between calls to
 * implement replace=False exactly as numpy does). */
int cfm_plan_sample_pi_f64(const double* pi, int B0, int B1, const double* u01, int n,
                           int64_t* i, int64_t* j, void* ws, void* stream);

/* K6 (trajectories) — one draw per entry of `rows` from the CONDITIONAL distribution of that plan row.
 * Replaces the inner loop of sample_trajectory()   torchcfm/optimal_transport.py:237-246
 *     for i in indices[-1]:  j.append(np.random.choice(pi.shape[1], p=pi[i] / pi[i].sum()))
 * np.random.choice consumes one uniform per call (u01[q], drawn by the caller in order) and returns the first column
 * whose running sum of the row exceeds u x (row total).  rows: device int64[n] (clamped to [0, B0)); j: device
 * int64[n].  _dense reads the entropic plan from the potentials in `sk_ws` and the cost row (never materialised;
 * ws: cfm_workspace_bytes(CFM_OP_SAMPLE_DENSE,B0,B1,0)); _pi_f64 reads an explicit fp64 plan.  The indices of a chain
 * of time slices stay on the device (the reference moves every B x B plan to the host and loops in Python). */
int cfm_plan_sample_rows_dense(const float* M, int B0, int B1, double reg, const void* sk_ws,
                               const int64_t* rows, const double* u01, int n, int64_t* j, void* ws,
                               void* stream);
int cfm_plan_sample_rows_pi_f64(const double* pi, int B0, int B1, const int64_t* rows, const double* u01,
                                int n, int64_t* j, void* stream);

/* K7+K8 — fused gather + probability-path sample + conditional flow.
 * Replaces x0[i], x1[j]                      torchcfm/optimal_transport.py:145
 * and the eager chain of                     torchcfm/conditional_flow_matching.py
 *   :82-83,126-129,153-154 (ICFM/OT), :446,474-478 (SB),
 *   :349-350,368,393-394 (Target), :588-589,617-618 (VP)
 * with the reference's operation order and no FMA contraction (bit-equal to
 * eager fp32).  i, j may be NULL (identity).  t [B], eps [B,d] (eps may be
 * NULL only when sigma terms vanish is NOT assumed: pass eps always).
 * VP: c0 = cos(pi/2 t), c1 = sin(pi/2 t) are passed in ([B] each, computed by
 * the caller's tensor library so they match its libm bit for bit); NULL for
 * the other variants, except SB where c0 (optional) = sigma_t = sigma*sqrt(t(1-t))
 * as the caller's library computes it (eager sqrt is not IEEE on every backend).
 * `sigma` is the Python-side value (double): the kernel
 * uses (float)sigma and, for TARGET, (float)(1.0 - sigma) exactly as eager
 * PyTorch casts Python scalars.  Outputs xt, ut [B,d]; x0g, x1g (may be NULL)
 * receive the gathered pairs.  xt_in (may be NULL): use this xt instead of
 * sampling one (compute_conditional_flow(x0, x1, t, xt) with a caller's xt);
 * xt may then be NULL. */
int cfm_sample_xt_ut_f32(int variant, const float* x0, const float* x1,
                         const int64_t* i, const int64_t* j, const float* t,
                         const float* eps, double sigma, const float* c0,
                         const float* c1, const float* xt_in, int B, int d, float* xt,
                         float* ut, float* x0g, float* x1g, void* stream);

/* Row gather  out[b,:] = src[idx[b],:]  (labels y0[i], y1[j]; :215-218).
 * elem_bytes = bytes per row. */
int cfm_gather_rows(const void* src, const int64_t* idx, int n, size_t row_bytes,
                    void* out, void* stream);

/* K10 — MLP vector field  Linear-SELU x3 + Linear  with the time column folded
 * in.  Replaces MLP.forward + torch_wrapper.forward
 *   torchcfm/models/models.py:10-21, torchcfm/utils.py:51-52.
 * x [B,d]; t: device float, either one scalar (t_per_row=0) or [B]
 * (t_per_row=1), or NULL when the net is not time varying;
 * W[l] [out_l, in_l] row-major (torch.nn.Linear layout), b[l] [out_l];
 * dims[n_layers+1] (host) = {d(+1 if time varying), w, ..., out};
 * W, b: host arrays of device pointers.  out [B, dims[n_layers]].
 * ws: cfm_workspace_bytes(CFM_OP_MLP, B, max width, 0). */
int cfm_mlp_forward_f32(const float* x, const float* t, int t_per_row,
                        const float* const* W, const float* const* b,
                        const int* dims, int n_layers, int B, float* out, void* ws,
                        void* stream);

/* K10 (training) — the same network with its hidden activations kept, its backward pass and the
 * optimizer step.  Replaces what autograd and torch.optim.Adam execute for
 *   vt = net(torch.cat([xt, t[:, None]], -1)); loss.backward(); optim.step()
 *   examples/images/cifar10/train_cifar10.py:141-151, torchcfm/models/models.py:10-21.
 * Forward: x [B, dims[0]] already holds every input column (the caller concatenated the time);
 * hidden / preact: host arrays of n_layers-1 device buffers [B, dims[l+1]] <- selu(z_l) / z_l.
 * Backward: acts / preact: host arrays of n_layers device pointers (acts[0] = x, acts[l] = hidden[l-1],
 * preact[l] = the forward's preact[l-1], preact[0] unused);
 * dout [B, dims[n_layers]]; writes dW[l] [dims[l+1], dims[l]], db[l] [dims[l+1]] and, when dx is not
 * NULL, dx [B, dims[0]].  Split-K partial sums are reduced in a fixed order (deterministic).
 * ws: cfm_workspace_bytes(CFM_OP_MLP_TRAIN, B, widest layer incl. input/output, largest dims[l]*dims[l+1]). */
int cfm_mlp_forward_train_f32(const float* x, const float* const* W, const float* const* b,
                              const int* dims, int n_layers, int B, float* const* hidden,
                              float* const* preact, float* out, void* stream);
int cfm_mlp_backward_f32(const float* const* acts, const float* const* preact, const float* const* W,
                         const int* dims, int n_layers, int B, const float* dout, float* const* dW,
                         float* const* db, float* dx, void* ws, void* stream);
/* One regression step of the vector field on a coupled batch, everything but the optimizer update, in 13
 * launches of this library's kernels (4-layer field) and nothing in between:
 *     v = net([xt, t]);  loss = mean((v - ut)^2);  dW, db = d loss / d parameters
 * Replaces `vt = model(torch.cat([xt, t[:, None]], -1)); loss = torch.mean((vt - ut) ** 2); loss.backward()` of the
 * reference's training loops: examples/images/cifar10/train_cifar10.py:147-149,
 * examples/2D_tutorials/Flow_matching_tutorial.ipynb cell 9 (torchcfm/models/models.py:10-21 is the net).
 * xt [B, dims[0] - 1] and t [B] when the net is time varying (the time column is never materialised), or
 * xt [B, dims[0]] and t = NULL.  hidden / preact as in cfm_mlp_forward_train_f32; g [B, dims[n_layers]] receives
 * d loss / d v; dW[l] [dims[l+1], dims[l]], db[l]; loss: device float.  Deterministic (fixed-order reductions).
 * layer_done (HOST array of n_layers hipEvent_t, or NULL): the data-parallel form of the reference's DDP wrapper
 * (examples/images/cifar10/train_cifar10_ddp.py:92 — gradient buckets all-reduced while the backward still runs):
 * layer l's split partials are reduced right after its weight-gradient product (one small launch per layer instead of
 * one at the end; same sums in the same order, bit-equal gradients) and layer_done[l] is recorded on `stream`, so the
 * caller's communication stream can all-reduce dW[l], db[l] under the remaining layers' products.
 * ws: cfm_workspace_bytes(CFM_OP_MLP_TRAIN, B, widest layer incl. input/output, largest dims[l]*dims[l+1]). */
int cfm_mlp_regression_step_f32(const float* xt, const float* t, const float* ut,
                                const float* const* W, const float* const* b, const int* dims, int n_layers,
                                int B, float* const* hidden, float* const* preact, float* g,
                                float* const* dW, float* const* db, float* loss, void* const* layer_done,
                                void* ws, void* stream);
/* One torch.optim.Adam step (amsgrad=False, maximize=False) on n_tensors fp32 tensors in ONE launch.
 * table: DEVICE array of n_tensors records {float* param; const float* grad; float* exp_avg;
 * float* exp_avg_sq; uint64 numel} (40 bytes each).  step >= 1 is the step count AFTER this update
 * (bias corrections 1 - beta^step are formed in double on the host, as torch does).
 * grad_scale (> 0): 1.0, or 1 / world size for a data-parallel run whose gradient buffers hold the all-reduced SUM:
 * the gradient is multiplied first (one fp32 rounding, what `grad.mul_(1 / world)` does) and written back, so .grad
 * holds the mean afterwards as under DDP (train_cifar10_ddp.py:92) — no separate elementwise launch. */
int cfm_adam_step_f32(const void* table, int n_tensors, double lr, double beta1, double beta2, double eps,
                      double weight_decay, int step, double grad_scale, void* stream);

/* SF2M sampling — one Euler-Maruyama step  y <- y + dt (v + score_sign * s) + g sqrt(|dt|) xi,  in place.
 * Replaces the step torchsde.sdeint(sde, x0, ts, method="euler") takes for the reference's SDE
 * (f = drift + score, g = sigma): examples/2D_tutorials/SF2M_tutorial.ipynb cell 5,
 * runner/src/models/components/solver.py:129-139,157-182.  s and xi may be NULL. */
int cfm_sde_em_step_f32(float* y, const float* v, const float* s, const float* xi, double dt, double g,
                        double score_sign, size_t n, void* stream);
/* SF2M sampling — the WHOLE Euler-Maruyama trajectory of a batch in one launch, for two small MLP fields (flow v and
 * score s: 4 layers, widths <= 64, time column last; Ws = bs = NULL: no score):
 *     y <- y + h (+-v(te, y) + s(te, y)) + g sqrt|h| xi        (reverse: -v, fields evaluated at te = 1 - t)
 * Replaces torchsde.sdeint(SDE(model, score_model, ...), x0, ts, method="euler", dt=...): SF2M_tutorial.ipynb cell 5,
 * runner/src/models/components/solver.py:129-139,157-182.  steps_host: n_steps records {float te, h, g_sqrt_h;
 * int is_out} on the HOST (copied into ws: >= 16 n_steps bytes of device scratch); out [n_out, B, d] receives the
 * state after every step with is_out != 0.  xi: caller's N(0,1) noise [n_steps, B, d] (then the trajectory is
 * bit-equal to stepping with cfm_mlp_forward_f32 + cfm_sde_em_step_f32), or NULL: Philox4x32-10 noise from `seed`
 * in the kernel.  CFM_EINVAL for other field shapes (step launch by launch instead). */
int cfm_sde_em_mlp_f32(const float* const* Wf, const float* const* bf, const float* const* Ws,
                       const float* const* bs, const int* dims, int n_layers, const float* y0, int B,
                       const void* steps_host, int n_steps, int reverse, const float* xi,
                       unsigned long long seed, float* out, void* ws, void* stream);
/* Mixture-RBF kernel sum  out[0] += sum_e sum_q exp(-gammas[q] * D[e])  over a squared-distance matrix D
 * (n elements, device fp32; gammas device fp32[n_gamma]; out device double, zeroed by the caller).
 * Replaces the K_XX / K_XY / K_YY matrices of mix_rbf_mmd2: runner/src/models/components/mmd.py:43-63,80-110. */
int cfm_rbf_mix_sum_f32(const float* D, size_t n, const float* gammas, int n_gamma, double* out, void* stream);

/* K11 — ODE solve of dx/dt = MLP([x, t]) on a time grid (torchdyn-style).
 * Replaces NeuralODE(torch_wrapper(model), solver=...).trajectory(x, t_span)
 *   call sites: examples/2D_tutorials/Flow_matching_tutorial.ipynb cells 11/16,
 *   examples/images/cifar10/utils_cifar.py:63-68.
 * t_span: host float[n_t].  traj: device [n_t,B,d].
 * euler: fixed steps on t_span.  dopri5: adaptive Dormand-Prince 5(4), one
 * global RMS error norm over the batch, every t_span point is a step end.
 * n_steps/nfe: host ints (accepted+rejected steps, function evaluations). */
int cfm_ode_euler_mlp_f32(const float* const* W, const float* const* b, const int* dims,
                          int n_layers, const float* x0, int B, const float* t_span,
                          int n_t, float* traj, int* nfe, void* ws, void* stream);
int cfm_ode_dopri5_mlp_f32(const float* const* W, const float* const* b, const int* dims,
                           int n_layers, const float* x0, int B, const float* t_span,
                           int n_t, float atol, float rtol, float* traj, int* n_steps,
                           int* nfe, void* ws, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CFM_GFX950_H */

