/* sinkhorn_oracle.c — TEST INFRASTRUCTURE (CPU oracle), not part of the product path.
 *
 * Float64 log-domain Sinkhorn with POT's loop semantics (ot.bregman.sinkhorn_log, called through
 * ot.sinkhorn at torchcfm/optimal_transport.py:51,87): uniform marginals,
 * u = v = 0 start, per iteration the column update v = log b - LSE_i(-M/reg + u) then the row
 * update u = log a - LSE_j(-M/reg + v), every `check_every`-th iteration the column-marginal
 * violation err = || sum_i exp(-M/reg + u_i + v_j) - b ||_2, stop when err < stopThr.
 * Same arithmetic as oracle/cfm_oracle.py::sinkhorn_log (which CPU tests pin it to), threaded with
 * OpenMP so that the full BASELINE sizes (B = 4096 / 8192, up to 1000 iterations) finish in
 * seconds on the GPU box's host cores.  The cost matrix arrives as the fp32 values the device
 * kernels see.
 *
 *   gcc -O2 -fopenmp -shared -fPIC oracle/sinkhorn_oracle.c -o oracle/_build/libsk_oracle.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

int cfm_oracle_sinkhorn_log(const float* M, int n, int m, double reg, int numItermax, double stopThr,
                            int check_every, double* u, double* v, double* err_out) {
    const double loga = log(1.0 / n), logb = log(1.0 / m);
    const double ir = -1.0 / reg;
    double* cmax = (double*)malloc(sizeof(double) * m);
    double* csum = (double*)malloc(sizeof(double) * m);
    double err = 1.0;
    int it = numItermax;
    for (int i = 0; i < n; ++i) u[i] = 0.0;
    for (int j = 0; j < m; ++j) v[j] = 0.0;
    const int JB = 512;                              /* column block of a thread */
    for (int ii = 0; ii < numItermax; ++ii) {
        /* v_j = logb - LSE_i(Mr_ij + u_i): two passes per column block (max, then sum of exp) */
#pragma omp parallel for schedule(static)
        for (int j0 = 0; j0 < m; j0 += JB) {
            const int j1 = j0 + JB < m ? j0 + JB : m;
            for (int j = j0; j < j1; ++j) cmax[j] = -INFINITY;
            for (int i = 0; i < n; ++i) {
                const float* r = M + (size_t)i * m; const double ui = u[i];
                for (int j = j0; j < j1; ++j) { const double x = (double)r[j] * ir + ui; if (x > cmax[j]) cmax[j] = x; }
            }
            for (int j = j0; j < j1; ++j) csum[j] = 0.0;
            for (int i = 0; i < n; ++i) {
                const float* r = M + (size_t)i * m; const double ui = u[i];
                for (int j = j0; j < j1; ++j) csum[j] += exp(((double)r[j] * ir + ui) - cmax[j]);
            }
            for (int j = j0; j < j1; ++j) v[j] = logb - (cmax[j] + log(csum[j]));
        }
        /* u_i = loga - LSE_j(Mr_ij + v_j) */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            const float* r = M + (size_t)i * m;
            double mx = -INFINITY;
            for (int j = 0; j < m; ++j) { const double x = (double)r[j] * ir + v[j]; if (x > mx) mx = x; }
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += exp(((double)r[j] * ir + v[j]) - mx);
            u[i] = loga - (mx + log(s));
        }
        if (check_every > 0 && ii % check_every == 0) {
#pragma omp parallel for schedule(static)
            for (int j0 = 0; j0 < m; j0 += JB) {
                const int j1 = j0 + JB < m ? j0 + JB : m;
                for (int j = j0; j < j1; ++j) csum[j] = 0.0;
                for (int i = 0; i < n; ++i) {
                    const float* r = M + (size_t)i * m; const double ui = u[i];
                    for (int j = j0; j < j1; ++j) csum[j] += exp((double)r[j] * ir + ui + v[j]);
                }
            }
            double e2 = 0.0;
            for (int j = 0; j < m; ++j) { const double dlt = csum[j] - 1.0 / m; e2 += dlt * dlt; }
            err = sqrt(e2);
            if (err < stopThr) { it = ii + 1; break; }
        }
    }
    free(cmax); free(csum);
    if (err_out) *err_out = err;
    return it;
}
