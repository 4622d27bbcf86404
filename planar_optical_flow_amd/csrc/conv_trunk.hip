// N2: the DROW / DR-SPAAM conv trunk layer for inference:
//   y = max_pool1d?( LeakyReLU_0.1( BatchNorm_eval( Conv1d(k = 3, pad = 1)(x) ) ), 2 )
// (src/depracted/model/dr_spaam.py:8-19 `_conv3x3`, :86-92 `_forward_conv`) on S short
// sequences at once: x [S][Ci][L] -> out [S][Co][L or L/2], float32.
//
// 450 x T x B sequences of 56 / 28 / 14 / 7 points: MIOpen falls to its naive / im2col paths
// on these shapes (6 TFLOP/s end to end).  Here the layer is an implicit GEMM on the float32
// MFMA (v_mfma_f32_32x32x2_f32): rows = output channels, columns = (sequence, position),
// K = (tap, input channel).
//   * B operand (activations): lane (n, h) needs x[seq(n)][ci + h][l(n) + tap - 1]: for a fixed
//     (tap, ci) the 32 columns are consecutive floats, so the operand is a coalesced global load
//     (zero at the sequence borders); the three taps re-read the same lines from L1.
//   * A operand (weights): pre-transposed on the host to [tap][ci][co]; a K-chunk (16 input
//     channels x 3 taps) x (128 output channels) is staged in LDS per workgroup and every lane
//     reads consecutive co.
//   * one wave = 32 columns x up to 128 output channels (4 accumulator tiles): one activation
//     load feeds 4 MFMAs; loads are issued a block of 8 k-steps ahead.
//   * epilogue: y = acc * scale[co] + shift[co] (BatchNorm folded with the conv bias), LeakyReLU,
//     optional max over position pairs (adjacent lanes), store.
#include "pof_common.h"

namespace {

constexpr int kCvWaves = 4;
constexpr int kCvCC = 16;            // input channels per LDS weight chunk
constexpr int kCvRows = 3 * kCvCC;   // K rows per chunk (tap-major)
using f32x16 = float __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *x;       // [S][Ci][L]
    const float *wt;      // [3][Ci][Co]
    const float *scale;   // [Co]
    const float *shift;   // [Co]
    float *out;           // [S][Co][Lout]
    int S, Ci, Co, L, pool;
    float slope;
};

template <int CT>
__global__ __launch_bounds__(64 * kCvWaves) void conv3_kernel(ConvArgs a)
{
    constexpr int COG = 32 * CT;                       // output channels per workgroup
    __shared__ float s_w[kCvRows][COG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int co0 = blockIdx.y * COG;
    const long long ncol = (long long)a.S * a.L;
    const long long n_g = ((long long)blockIdx.x * kCvWaves + wave) * 32 + r;   // this lane's column
    const bool col_ok = n_g < ncol;
    const long long nc = col_ok ? n_g : ncol - 1;
    const int seq = (int)(nc / a.L), l = (int)(nc - (long long)seq * a.L);
    const float *xs = a.x + (long long)seq * a.Ci * a.L + l;                    // x[seq][0][l]
    const bool tap_ok[3] = {col_ok && l > 0, col_ok, col_ok && l < a.L - 1};

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0};

    for (int ci0 = 0; ci0 < a.Ci; ci0 += kCvCC) {
        const int cc = min(kCvCC, a.Ci - ci0);
        __syncthreads();
        // weight chunk: rows (tap, ci_local), columns co0 .. co0 + COG
        for (int e = threadIdx.x; e < kCvRows * COG; e += 64 * kCvWaves) {
            const int row = e / COG, c = e - row * COG;
            const int tap = row / kCvCC, cl = row - tap * kCvCC;
            const bool ok = cl < cc && co0 + c < a.Co;
            s_w[row][c] = ok ? a.wt[((long long)tap * a.Ci + ci0 + cl) * a.Co + co0 + c] : 0.0f;
        }
        __syncthreads();
        // activations of the whole chunk first (3 taps x 8 channel pairs = 24 loads in flight)
        float xb[3][kCvCC / 2];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int p = 0; p < kCvCC / 2; ++p) {
                const int cl = min(2 * p + h, cc - 1);                          // clamped: masked below
                // border / tail lanes read a valid neighbour address (clamped tap offset), zeroed afterwards
                const int off = tap_ok[tap] ? tap - 1 : 0;
                xb[tap][p] = xs[(long long)(ci0 + cl) * a.L + off];
            }
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int p = 0; p < kCvCC / 2; ++p) {
                const bool ok = tap_ok[tap] && (2 * p + h < cc);
                const float bv = ok ? xb[tap][p] : 0.0f;
                const int row = tap * kCvCC + 2 * p + h;
#pragma unroll
                for (int t = 0; t < CT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w[row][t * 32 + r], bv, acc[t], 0, 0, 0);
            }
    }
    // epilogue.  C/D layout: col = lane & 31 (column n), row = (reg & 3) + 8 * (reg >> 2) + 4 * h (co)
    const int Lout = a.pool ? a.L / 2 : a.L;
    float *os = a.out + (long long)seq * a.Co * Lout + (a.pool ? l / 2 : l);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = co0 + t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const bool co_ok = co < a.Co;
            const int cs = co_ok ? co : a.Co - 1;
            float y = acc[t][reg] * a.scale[cs] + a.shift[cs];
            y = y > 0.0f ? y : y * a.slope;
            if (a.pool) {
                const float other = __shfl_xor(y, 1, 64);       // the pair (2j, 2j+1) sits on adjacent lanes
                y = fmaxf(y, other);
                if (col_ok && co_ok && !(l & 1) && l + 1 < a.L) os[(long long)co * Lout] = y;
            } else {
                if (col_ok && co_ok) os[(long long)co * Lout] = y;
            }
        }
}

}  // namespace

extern "C" int pof_conv3_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift,
                                  int S, int Ci, int Co, int L, int pool, double negative_slope, float *out,
                                  pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!x || !wt || !scale || !shift || !out) return POF_E_BADARG;
    if (S < 0 || Ci < 1 || Co < 1 || L < 1) return POF_E_BADARG;
    if (pool && (L < 2 || (L & 1))) return POF_E_SHAPE;   // pooled pairs sit on adjacent lanes: even L
    if (S == 0) return POF_OK;
    ConvArgs a;
    a.x = x; a.wt = wt; a.scale = scale; a.shift = shift; a.out = out;
    a.S = S; a.Ci = Ci; a.Co = Co; a.L = L; a.pool = pool ? 1 : 0; a.slope = (float)negative_slope;
    const long long ncol = (long long)S * L;
    const long long tiles = (ncol + 31) / 32;
    const long long gx = (tiles + kCvWaves - 1) / kCvWaves;
    if (gx > 0x7fffffffLL) return POF_E_SHAPE;
    hipStream_t s = pof_stream(stream);
    if (Co <= 64) {
        conv3_kernel<2><<<dim3((unsigned)gx, (Co + 63) / 64), 64 * kCvWaves, 0, s>>>(a);
    } else {
        conv3_kernel<4><<<dim3((unsigned)gx, (Co + 127) / 128), 64 * kCvWaves, 0, s>>>(a);
    }
    POF_CHECK_LAUNCH();
    return POF_OK;
}
