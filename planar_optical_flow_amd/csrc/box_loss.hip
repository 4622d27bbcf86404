// configs[3]: the box-regression loss (src/model/box_regression.py:52-67 `regression_loss2`) and its
// gradient with respect to the prediction, in one launch:
//
//   3 targets: mean_b (|d0| + |d1|) + alpha * mean_b |d2|                 d = pred - target
//   5 targets: mean_b |d0| + mean_b (|d1| + |d2| + |d3|) + alpha * mean_b |d4|
//   => loss = (1 / B) sum_b sum_j c_j |d_bj|, c_j = 1 except c_last = alpha;  d loss / d pred_bj = c_j sign(d_bj) / B
//
// As the framework composes it the loss is 12 element-wise / reduction launches forward and 15 backward on a [256 x 3]
// tensor -- a seventh of the kernel nodes of a whole training step (profiles/r3_boxhead_step_order.txt).  One workgroup;
// float64 accumulation (the result is within half an ulp of the exactly rounded sum; the framework's float32 means
// differ from it in the last bits).
#include "pof_common.h"

namespace {

constexpr int kLossThreads = 256;

__global__ __launch_bounds__(kLossThreads) void regression_loss2_kernel(const float *__restrict__ pred,
                                                                        const float *__restrict__ target, long long B,
                                                                        int T, float alpha, float *__restrict__ loss,
                                                                        float *__restrict__ dpred)
{
    __shared__ double s_w[kLossThreads / 64];
    const long long n = B * T;
    const float inv_b = 1.0f / (float)B;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < n; i += kLossThreads) {
        const int j = (int)(i % T);
        const float c = j == T - 1 ? alpha : 1.0f;
        const float d = pred[i] - target[i];
        acc += (double)c * (double)fabsf(d);
        if (dpred) dpred[i] = d > 0.0f ? c * inv_b : d < 0.0f ? -c * inv_b : 0.0f * d;    // NaN stays NaN
    }
    acc = wave_sum_f64(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kLossThreads / 64; ++w) t += s_w[w];
        *loss = (float)(t / (double)B);
    }
}

}  // namespace

extern "C" int pof_regression_loss2(const float *pred, const float *target, long long B, int T, double alpha, float *loss,
                                    float *dpred, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!pred || !target || !loss) return POF_E_BADARG;
    if (B < 1 || (T != 3 && T != 5)) return POF_E_SHAPE;      // the reference defines the loss for 3 and 5 targets only
    regression_loss2_kernel<<<1, kLossThreads, 0, pof_stream(stream)>>>(pred, target, B, T, (float)alpha, loss, dpred);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
