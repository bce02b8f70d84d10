// cfm_common.h — shared device helpers for libcfm_gfx950.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/cfm_gfx950.h"
#include "../../include/cfm_gfx950_tuning.h"   // every export is declared in one of the two headers (-fvisibility=hidden)

#define CFM_WAVE 64

static inline int cfm_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
static inline int cfm_hip(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

static inline size_t cfm_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute and occupancy answers are PER DEVICE: every once-only setup is keyed by the current device
#define CFM_MAX_DEVICES 16
static inline int cfm_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CFM_MAX_DEVICES) dev = 0;
    return dev;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8; give each
// XCD a contiguous range of logical ids so neighbouring tiles share its L2.
// Bijective for any grid size (speed only, never correctness).
__device__ __forceinline__ unsigned cfm_xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    if (nwg < nx) return bid;
    unsigned q = nwg / nx, r = nwg % nx, xcd = bid % nx, k = bid / nx;
    unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// ---- wave64 reductions (all 64 lanes active) ----
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// order-preserving map double -> uint64 (total order incl. negatives)
__device__ __forceinline__ unsigned long long d2ord(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
