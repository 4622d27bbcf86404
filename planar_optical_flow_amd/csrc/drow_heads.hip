// N2: the detector's two heads (src/depracted/model/dr_spaam.py:104-121, `_forward_fused_cutout`): average over
// the remaining positions of the last trunk block, then the 1x1 convolutions conv_cls / conv_reg -- on a length-1
// sequence two small dense layers.  Through the framework that is a reduction, a rescale and two GEMM launches of
// [S, 128] x [128, 1 | 2]; in the streaming step (one scan per call, hipGraph replay) every launch costs ~5 us
// whatever it does, so the four are one kernel here.
//
// feat [S][C][L] float32 -> pred_cls [S][n_cls], pred_reg [S][2].  One wave per sequence: a lane owns channels
// lane, lane + 64, ... (their L values are contiguous), the per-output dot products are wave sums.
#include "pof_common.h"

namespace {

constexpr int kHeadWaves = 4;
constexpr int kHeadMaxOut = 8;

__global__ __launch_bounds__(64 * kHeadWaves) void drow_heads_kernel(const float *__restrict__ feat, int S, int C, int L,
                                                                     const float *__restrict__ w_cls,
                                                                     const float *__restrict__ b_cls, int n_cls,
                                                                     const float *__restrict__ w_reg,
                                                                     const float *__restrict__ b_reg,
                                                                     float *__restrict__ pred_cls,
                                                                     float *__restrict__ pred_reg)
{
    const int s = blockIdx.x * kHeadWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= S) return;
    const int n_out = n_cls + 2;
    float acc[kHeadMaxOut];
#pragma unroll
    for (int o = 0; o < kHeadMaxOut; ++o) acc[o] = 0.0f;
    const float inv = 1.0f / (float)L;
    for (int c = lane; c < C; c += 64) {
        const float *p = feat + ((long long)s * C + c) * L;
        float sum = 0.0f;
        for (int l = 0; l < L; ++l) sum += p[l];
        const float m = sum * inv;
#pragma unroll
        for (int o = 0; o < kHeadMaxOut; ++o)
            if (o < n_out) acc[o] = fmaf(m, o < n_cls ? w_cls[o * C + c] : w_reg[(o - n_cls) * C + c], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < kHeadMaxOut; ++o) {
        if (o >= n_out) break;
        const float t = wave_sum_f32(acc[o]);
        if (lane == 0) {
            if (o < n_cls) pred_cls[(long long)s * n_cls + o] = t + b_cls[o];
            else pred_reg[(long long)s * 2 + (o - n_cls)] = t + b_reg[o - n_cls];
        }
    }
}

}  // namespace

extern "C" int pof_drow_heads(const float *feat, int S, int C, int L, const float *w_cls, const float *b_cls, int n_cls,
                              const float *w_reg, const float *b_reg, float *pred_cls, float *pred_reg,
                              pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!feat || !w_cls || !b_cls || !w_reg || !b_reg || !pred_cls || !pred_reg) return POF_E_BADARG;
    if (S < 0 || C < 1 || L < 1 || n_cls < 1) return POF_E_BADARG;
    if (n_cls + 2 > kHeadMaxOut) return POF_E_SHAPE;
    if (S == 0) return POF_OK;
    drow_heads_kernel<<<(S + kHeadWaves - 1) / kHeadWaves, 64 * kHeadWaves, 0, pof_stream(stream)>>>(
        feat, S, C, L, w_cls, b_cls, n_cls, w_reg, b_reg, pred_cls, pred_reg);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
