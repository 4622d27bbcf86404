// Shared device/host helpers for libpof_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pof_abi.h"

#define POF_WAVE 64

// hipGetLastError() reports (and clears) the last error of ANY earlier HIP call of this thread -- e.g. one
// raised by the caller's own previous launch.  Every entry point takes that state first, so that
// POF_CHECK_LAUNCH() only sees its own launches -- but it does not swallow it: the code is parked in a
// thread-local slot that the caller reads (and clears) with pof_take_stale_error().
extern thread_local int pof_stale_error_slot;
#define POF_CLEAR_STALE_ERROR()                                  \
    do {                                                         \
        const hipError_t stale_ = hipGetLastError();             \
        if (stale_ != hipSuccess) pof_stale_error_slot = (int)stale_; \
    } while (0)

#define POF_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return POF_E_LAUNCH; \
    } while (0)

static inline hipStream_t pof_stream(pof_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// The translation units are built with -ffp-contract=off: a*b+c is two
// roundings unless fma() is written out.  That is what makes the float64 index
// math of the cutout and the association distances bit-identical to NumPy.

// Correctly rounded x / c for a divisor known per launch; `rc` = RN(1/c).
// q0 = RN(x*rc) is within 1 ulp of x/c; the exact FMA residual and one FMA
// correction give RN(x/c) (Markstein).  Checked against '/' on 1.2e9 random
// operands for the divisors this library uses (tools/divconst_check.c).
__device__ __forceinline__ double pof_div_const(double x, double c, double rc)
{
    double q = x * rc;
    double r = fma(-q, c, x);
    return fma(r, rc, q);
}

__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over the 64 lanes on the vector pipe only (no LDS-pipe shuffles): four row_shr adds inside each row of 16
// lanes, then row_bcast15 / row_bcast31 carry the row totals upwards.  The total is valid in LANE 63 only.
__device__ __forceinline__ float wave_sum_to_lane63_f32(float v)
{
#define POF_DPP_ADD(ctrl, rmask, bctl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, bctl))
    POF_DPP_ADD(0x111, 0xF, true);    // row_shr:1
    POF_DPP_ADD(0x112, 0xF, true);    // row_shr:2
    POF_DPP_ADD(0x114, 0xF, true);    // row_shr:4
    POF_DPP_ADD(0x118, 0xF, true);    // row_shr:8   -> lane 15 of every row holds the row's sum
    POF_DPP_ADD(0x142, 0xA, false);   // row_bcast15 into rows 1 and 3
    POF_DPP_ADD(0x143, 0xC, false);   // row_bcast31 into rows 2 and 3 -> lane 63 holds the wave's sum
#undef POF_DPP_ADD
    return v;
}
