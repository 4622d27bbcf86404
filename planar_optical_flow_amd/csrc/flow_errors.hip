// A12: end-point error / average angular error reductions
//   prototype.py:27-32 (per-sample EPE), dr_spaam.py:22-27 (masked EPE),
//   eval_utils.py:129-134 (EPE + AAE; note atan2(x, y) argument order).
// One workgroup per sample; per-point norms in float32 like torch, sums in
// float64.  HBM bound: 16 B (+4 B mask) read per point.
#include "pof_common.h"

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void flow_errors_kernel(const float *pred, const float *target,
                                                               const float *mask, int N, double *epe_sum,
                                                               double *aae_sum, double *cnt)
{
    __shared__ double s_red[3][kThreads / 64];
    const int b = blockIdx.x;
    const float2 *p = reinterpret_cast<const float2 *>(pred) + (long long)b * N;
    const float2 *t = reinterpret_cast<const float2 *>(target) + (long long)b * N;
    const float *m = mask ? mask + (long long)b * N : nullptr;
    double e = 0.0, a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < N; i += kThreads) {
        if (m && m[i] != 1.0f) continue;
        const float2 pv = p[i], tv = t[i];
        const float dx = pv.x - tv.x, dy = pv.y - tv.y;
        e += (double)sqrtf(dx * dx + dy * dy);
        a += (double)fabsf(atan2f(pv.x, pv.y) - atan2f(tv.x, tv.y));
        c += 1.0;
    }
    e = wave_sum_f64(e);
    a = wave_sum_f64(a);
    c = wave_sum_f64(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[0][w] = e;
        s_red[1][w] = a;
        s_red[2][w] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kThreads / 64; ++k) {
            e += s_red[0][k];
            a += s_red[1][k];
            c += s_red[2][k];
        }
        if (epe_sum) epe_sum[b] = e;
        if (aae_sum) aae_sum[b] = a;
        if (cnt) cnt[b] = c;
    }
}

}  // namespace

extern "C" int pof_flow_errors(const float *pred, const float *target, const float *mask, int B, int N,
                               double *epe_sum, double *aae_sum, double *cnt, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!pred || !target || B < 0 || N < 1) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    flow_errors_kernel<<<B, kThreads, 0, pof_stream(stream)>>>(pred, target, mask, N, epe_sum, aae_sum, cnt);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
