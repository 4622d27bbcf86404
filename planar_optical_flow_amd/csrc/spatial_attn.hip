// A10: _SpatialAttention.forward after the embedding conv
//      (src/depracted/model/dr_spaam.py:163-217), banded.
//
// The reference builds the full N x N similarity, masks it to the +-hw band and
// multiplies the N x N softmax into the template.  Only the band is non-zero,
// so both steps are done on the band (O(N*w) instead of O(N^2)):
//
//   attn_band_kernel   one wave64 per point: w dot products of length E,
//                      band[b,i,k] = <emb_x[i], emb_t[clamp(i-hw+k)]> (clamped
//                      duplicates kept, like the reference's gather), masked
//                      softmax over the DISTINCT in-window columns ->
//                      prob[b,i,k] (0 for clamped duplicates).
//   attn_merge_kernel  out[i] = alpha*x[i] + (1-alpha) * sum_k prob[i,k]*tmpl[i-hw+k]
//                      Each lane owns one float4 column and walks down the
//                      points of its segment with a W-deep register ring of
//                      template rows: every template row is loaded once per
//                      segment (+2hw halo rows), all loads/stores are 16 B per
//                      lane and fully coalesced, the weights are wave-uniform.
//
// Roofline: HBM.  Algorithmic bytes per (sample, step): x + tmpl + out rows,
// 3 * N * F * 4 (= 19.4 MB at N=450, F=3584) + emb/band/prob (0.5 MB).
#include "pof_common.h"

namespace {

constexpr int kMaxW = 15;

__global__ __launch_bounds__(256) void attn_band_kernel(const float *emb_x, const float *emb_t, int N,
                                                        int E, int W, float *band, float *prob)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int hw = W / 2;
    const float *ex = emb_x + ((long long)b * N + i) * E;
    const float *et = emb_t + (long long)b * N * E;
    float mine = -INFINITY;
    for (int k = 0; k < W; ++k) {
        const int j = min(max(i - hw + k, 0), N - 1);
        const float *tj = et + (long long)j * E;
        float part = 0.0f;
        for (int e = lane; e < E; e += 64) part = fmaf(ex[e], tj[e], part);
        part = wave_sum_f32(part);
        if (lane == k) mine = part;
    }
    const int ju = i - hw + lane;  // unclamped column of this lane's slot
    const bool slot = lane < W;
    const bool distinct = slot && ju >= 0 && ju <= N - 1;
    const float mx = wave_max_f32(slot ? mine : -INFINITY);
    const float ex_ = distinct ? expf(mine - mx) : 0.0f;
    const float sum = wave_sum_f32(ex_);
    if (slot) {
        const long long o = ((long long)b * N + i) * W + lane;
        if (band) band[o] = mine;
        prob[o] = ex_ / sum;
    }
}

template <int W>
__global__ __launch_bounds__(128) void attn_merge_kernel(const float4 *x, const float4 *tmpl,
                                                         const float *prob, float4 *out, int N, int F4,
                                                         int L, float alpha, float one_minus_alpha)
{
    constexpr int HW = W / 2;
    const int col = blockIdx.x * 128 + threadIdx.x;
    if (col >= F4) return;
    const int b = blockIdx.z;
    const int s0 = blockIdx.y * L;
    const int s1 = min(N, s0 + L);
    const long long sample = (long long)b * N;
    const float4 *T = tmpl + sample * F4 + col;
    const float4 *X = x + sample * F4 + col;
    float4 *O = out + sample * F4 + col;
    const float *P = prob + sample * W;
    const int base = s0 - HW;
    const int rmax = s1 - 1 + HW;
    float4 win[W];
#pragma unroll
    for (int u = 0; u < W; ++u) win[u] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int m = 0; base + m * W <= rmax; ++m) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int r = base + m * W + u;
            if (r <= rmax) {
                win[u] = (r >= 0 && r <= N - 1) ? T[(long long)r * F4] : make_float4(0.f, 0.f, 0.f, 0.f);
                const int i = r - HW;
                if (i >= s0) {
                    const float *p = P + (long long)i * W;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const float pk = p[k];
                        const float4 t = win[(u + k + 1) % W];
                        acc.x = fmaf(pk, t.x, acc.x);
                        acc.y = fmaf(pk, t.y, acc.y);
                        acc.z = fmaf(pk, t.z, acc.z);
                        acc.w = fmaf(pk, t.w, acc.w);
                    }
                    const float4 xv = X[(long long)i * F4];
                    float4 o;
                    o.x = alpha * xv.x + one_minus_alpha * acc.x;
                    o.y = alpha * xv.y + one_minus_alpha * acc.y;
                    o.z = alpha * xv.z + one_minus_alpha * acc.z;
                    o.w = alpha * xv.w + one_minus_alpha * acc.w;
                    O[(long long)i * F4] = o;
                }
            }
        }
    }
}

template <int W>
void launch_merge(const float *x, const float *tmpl, const float *prob, float *out, int B, int N, int F,
                  int L, double alpha, hipStream_t s)
{
    const int F4 = F / 4;
    dim3 grid((F4 + 127) / 128, (N + L - 1) / L, B);
    attn_merge_kernel<W><<<grid, 128, 0, s>>>(reinterpret_cast<const float4 *>(x),
                                             reinterpret_cast<const float4 *>(tmpl), prob,
                                             reinterpret_cast<float4 *>(out), N, F4, L, (float)alpha,
                                             (float)(1.0 - alpha));
}

}  // namespace

extern "C" int pof_spatial_attention(const float *emb_x, const float *emb_t, const float *x,
                                     const float *tmpl, int B, int N, int E, int F, int window,
                                     double alpha, float *band, float *prob, float *out,
                                     pof_stream_t stream)
{
    if (!emb_x || !emb_t || !x || !tmpl || !prob || !out) return POF_E_BADARG;
    if (B < 0 || N < 1 || E < 1 || F < 1) return POF_E_BADARG;
    // the reference uses hw = int(window/2) neighbours each side: an even window
    // behaves like window+1
    const int W = 2 * (window / 2) + 1;
    if (W < 1 || W > kMaxW) return POF_E_SHAPE;
    if (F % 4 != 0) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    hipStream_t s = pof_stream(stream);
    attn_band_kernel<<<dim3((N + 3) / 4, B), 256, 0, s>>>(emb_x, emb_t, N, E, W, band, prob);
    POF_CHECK_LAUNCH();
    // segment length: whole scan per lane when the batch alone fills the chip,
    // shorter segments (more workgroups, a little halo re-read) for small batches
    const long long colblocks = (F / 4 + 127) / 128;
    int L = N;
    while (L > 32 && colblocks * ((N + L - 1) / L) * B < 4096) L = (L + 1) / 2;
    switch (W) {
        case 1: launch_merge<1>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 3: launch_merge<3>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 5: launch_merge<5>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 7: launch_merge<7>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 9: launch_merge<9>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 11: launch_merge<11>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        case 13: launch_merge<13>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
        default: launch_merge<15>(x, tmpl, prob, out, B, N, F, L, alpha, s); break;
    }
    POF_CHECK_LAUNCH();
    return POF_OK;
}
