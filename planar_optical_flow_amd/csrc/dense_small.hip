// configs[3]: the dense layers of the box-regression head (src/model/box_regression.py:26-45 `_fc`,
// :139-141 fc1 1024 -> 512, fc2 512 -> 256, fc3 256 -> target_dim), forward:
//
//   out[b][n] = sum_k x[b][k] * w[n][k] + bias[n]            (torch.nn.Linear; x [B][K], w [N][K] as torch keeps them)
//
// at the head's batch (256 rows).  These are 0.07-0.27 GFLOP problems: what they cost is a launch and a latency chain,
// and the BLAS library's choice for an output of at most 256 x 256 is ONE 256 x 256 tile on one CU -- fc2 takes
// 118 us there (tools/bench_dense_small.py: 257 rows or 264 columns take 19 us), 12 % of the whole training step.
//
// Here: float32 MFMA (v_mfma_f32_32x32x2_f32), one workgroup per 32 x 32 output tile, its four waves split K
// (8-element chunks dealt round-robin) and meet in LDS; wave 0 adds them in wave order -- deterministic, no atomics.
// Both operands are K-contiguous, so a lane reads 16 bytes of ITS row per chunk: lane (r, h) of the A operand holds
// x[b0 + r][8c + 4h .. 8c + 4h + 3], of the B operand w[n0 + r][same k]; the four components feed four MFMAs (the
// k-pairs {8c + j, 8c + 4 + j} -- any pairing is valid as long as both operands use the same one).  Four chunks of
// loads are in flight ahead of the MFMAs.  256 x 512 -> 256: 64 workgroups, 16 chunks per wave.
#include "pof_common.h"

namespace {

using f32x16 = float __attribute__((ext_vector_type(16)));
using F4V = float __attribute__((ext_vector_type(4)));
constexpr int kDnWaves = 4;
constexpr int kDnAhead = 4;     // chunks of loads in flight per wave

struct DenseArgs {
    const float *x, *w, *bias;
    float *out;
    int B, K, N;
};

__global__ __launch_bounds__(64 * kDnWaves) void dense_small_kernel(DenseArgs a)
{
    __shared__ float s_acc[kDnWaves - 1][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    // rows past the end read the last row (their products land in output rows / columns that are not stored)
    const float *px = a.x + (long long)min(b0 + r, a.B - 1) * a.K + 4 * h;
    const float *pw = a.w + (long long)min(n0 + r, a.N - 1) * a.K + 4 * h;
    const int nchunk = (a.K + 7) >> 3;
    f32x16 acc = f32x16{0};
    for (int c0 = wave; c0 < nchunk; c0 += kDnWaves * kDnAhead) {
        F4V va[kDnAhead], vb[kDnAhead];
#pragma unroll
        for (int u = 0; u < kDnAhead; ++u) {
            const int c = c0 + u * kDnWaves;
            const bool ok = c < nchunk && 8 * c + 4 * h < a.K;       // K is a multiple of 4: a lane's four k exist together
            va[u] = ok ? *reinterpret_cast<const F4V *>(px + 8 * c) : F4V{0.0f, 0.0f, 0.0f, 0.0f};
            vb[u] = ok ? *reinterpret_cast<const F4V *>(pw + 8 * c) : F4V{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int u = 0; u < kDnAhead; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[u][j], vb[u][j], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) s_acc[wave - 1][v][lane] = acc[v];
    }
    __syncthreads();
    if (wave > 0) return;
    const int n = n0 + r;
    const float bias = (a.bias && n < a.N) ? a.bias[n] : 0.0f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float s = acc[v];
#pragma unroll
        for (int w = 0; w < kDnWaves - 1; ++w) s += s_acc[w][v][lane];
        // C/D layout: column (B-operand row) = lane & 31, row (A-operand row) = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
        const int b = b0 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (b < a.B && n < a.N) a.out[(long long)b * a.N + n] = s + bias;
    }
}

}  // namespace

extern "C" int pof_linear_bias(const float *x, const float *w, const float *bias, int B, int K, int N, float *out,
                               pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!x || !w || !out) return POF_E_BADARG;
    if (B < 0 || K < 1 || N < 1) return POF_E_BADARG;
    if (K & 3) return POF_E_SHAPE;                               // 16-byte operand loads
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    const long long gx = (B + 31) / 32, gy = (N + 31) / 32;
    if (gx > 0x7fffffffLL || gy > 65535) return POF_E_SHAPE;
    DenseArgs a{x, w, bias, out, B, K, N};
    dense_small_kernel<<<dim3((unsigned)gx, (unsigned)gy), 64 * kDnWaves, 0, pof_stream(stream)>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
