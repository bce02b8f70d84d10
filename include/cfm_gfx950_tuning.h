/*
 * cfm_gfx950_tuning.h — measurement and tuning exports of libcfm_gfx950.so.
 *
 * NOT part of the operator ABI (include/cfm_gfx950.h): no binding of the reference needs any of these, the Python
 * mirror calls them only from tools/ (sweeps, profiles) and tests.  They are the one piece of process-wide state of
 * the library: a table of solver parameters.  A solve works on a snapshot of the table taken under a mutex when it
 * starts, so a setter called from another thread never tears a running solve; the getters read back what the LAST
 * solve of the calling thread (or on the given workspace) left behind and block until it is resident.
 */
#ifndef CFM_GFX950_TUNING_H
#define CFM_GFX950_TUNING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* exact assignment (cfm_assign_exact_f32): epsilon schedule (theta, first / last epsilon as fractions of the cost
 * range, phase cut), round caps, launches per polled chunk.  Arguments <= 0 (stop_frac, arr_cap: < 0) keep the value. */
void cfm_assign_set_params(double theta, double eps0_frac, double eps_last_frac, double stop_frac, int round_cap,
                           int arr_cap, int chunk);
void cfm_assign_set_mode(int sparse);            /* 0: no candidate-list solver (dense state machine only) */
void cfm_assign_set_handoff(int handoff);        /* free rows at which phase C moves into the list solver (<= 64) */
void cfm_assign_set_stop_early(double f);        /* phase cut of every epsilon phase but the last */
void cfm_assign_set_wide_blocks(int cap);        /* upper bound on the grid of the chip-wide step kernel (0: none) */
void cfm_assign_set_bulk(int bulk, int min_n);   /* launches enqueued before the first poll, for n >= min_n */
/* on = 2 (the DEFAULT): every bid of a solve — the epsilon > 0 phases AND the epsilon = 0 rounds — in the one-launch
 * asynchronous auction (asg_auction), the whole solve as one unpolled program of 12 launches; on = 1: only the epsilon > 0
 * phases there, the epsilon = 0 rounds as synchronous launches; on = 0: every round a launch (the A/B reference; always
 * the path of n < 512 and n > 8192).  blocks >= 0: workgroups per problem of the auction in the batch entry (0: as the
 * other kernels); last_div > 0: its last phase is cut at stop_frac / last_div.  The grid is capped at the CUs the
 * stream may use (CU-masked streams); a grid that cannot hold 1/256 of the rows per workgroup falls back to on = 0. */
void cfm_assign_set_async(int on, int blocks, int last_div);
void cfm_assign_set_async_min_n(int n);          /* smallest n that takes the one-launch auction (default 512; >= 64) */
void cfm_assign_get_async(int* out3);            /* {on, blocks, last_div} as set (tests restore what they changed) */
void cfm_assign_set_small(int on);               /* 0: problems of n <= 256 take the chip-wide machine too */
void cfm_set_blocking_sync(int on);              /* THIS host thread's solver waits: 1 = sleep in the driver (hipEventBlockingSync) instead of spinning on a core; cfm_amd.prefetch sets it for its worker threads */
void cfm_ode_set_fused(int on);                  /* 0: layer-per-kernel ODE stages instead of the fused small-field drivers */
void cfm_mlp_set_glds(int mode);               /* MLP layers on 64 x 64 tiles: 0 = register-staged operand loads (gemm_core.h: the plain ascending-k chain, bit-equal to the fused small-field ODE drivers), 1 = direct-to-LDS DMA (gemm_glds64.h) when the rows are 16-byte aligned, 2 (default) = for 4-byte aligned rows too */
int cfm_mlp_get_glds(void);                    /* the mode in force (tests restore what they changed) */

/* read-backs (blocking) */
int cfm_assign_debug_times(const void* ws, double* us32);          /* microseconds per mode of the last solve on ws */
int cfm_assign_debug_solver(const void* ws, int n, long long* out16);   /* -DSP_PROFILE builds: list-solver cycle counters */
void cfm_assign_debug_small(int* out16);                           /* status block of this thread's last one-workgroup solve */
void cfm_assign_debug_fallback(int* out2);                         /* {solves of this PROCESS (all threads) redone by the dense machine, last device error} */
int cfm_plan_zero_entries_f64(double* pi, const int64_t* flat, int n, void* stream);   /* pi.flat[flat[q]] = 0 (sample_map(replace=False) bookkeeping of the mirror) */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CFM_GFX950_TUNING_H */
