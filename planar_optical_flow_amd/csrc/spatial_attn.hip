// A10: _SpatialAttention.forward after the embedding conv
//      (src/depracted/model/dr_spaam.py:163-217), banded.
//
// The reference builds the full N x N similarity, masks it to the +-hw band and
// multiplies the N x N softmax into the template.  Only the band is non-zero,
// so both steps are done on the band (O(N*w) instead of O(N^2)):
//
//   attn_band_kernel   32 points per workgroup, emb rows staged in LDS: w dot products of length E,
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

// One workgroup = 32 consecutive points of one sample.  The 32 emb_x rows and the
// 32 + W - 1 emb_t rows they meet are staged in LDS (row stride E+1: lanes that walk
// consecutive rows hit distinct banks); lane = point, the 4 lane groups split E, every lane
// keeps its W partial dot products in registers (1 + W LDS reads per W FMAs).  Partial sums
// meet in LDS, then the first 32 lanes run the masked softmax of their point.
constexpr int kBandTile = 32;
constexpr int kBandThreads = 128;

template <int W>
__global__ __launch_bounds__(kBandThreads) void attn_band_kernel(const float *emb_x, const float *emb_t,
                                                                 int N, int E, float *band, float *prob)
{
    constexpr int HW = W / 2;
    constexpr int RT = kBandTile + W - 1;   // template rows staged
    extern __shared__ float s_band[];
    const int ld = E + 1;
    float *sx = s_band;                     // [kBandTile][ld]
    float *st = sx + kBandTile * ld;        // [RT][ld]
    float *sred = st + RT * ld;             // [4][kBandTile][W]

    const int b = blockIdx.y;
    const int i0 = blockIdx.x * kBandTile;
    const float *ex = emb_x + (long long)b * N * E;
    const float *et = emb_t + (long long)b * N * E;
    if ((E & 3) == 0) {  // rows are 16-B aligned: wide, independent loads, scalar LDS writes (odd row stride)
        const int E4 = E >> 2;
#pragma unroll 4
        for (int e = threadIdx.x; e < kBandTile * E4; e += kBandThreads) {
            const int r = e / E4, c = e - r * E4;
            const int row = min(i0 + r, N - 1);
            const float4 v = reinterpret_cast<const float4 *>(ex + (long long)row * E)[c];
            float *d = sx + r * ld + 4 * c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
#pragma unroll 4
        for (int e = threadIdx.x; e < RT * E4; e += kBandThreads) {
            const int r = e / E4, c = e - r * E4;
            const int row = min(max(i0 - HW + r, 0), N - 1);   // clamped like the reference's gather
            const float4 v = reinterpret_cast<const float4 *>(et + (long long)row * E)[c];
            float *d = st + r * ld + 4 * c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
        for (int e = threadIdx.x; e < kBandTile * E; e += kBandThreads) {
            const int r = e / E, c = e - r * E;
            const int row = min(i0 + r, N - 1);
            sx[r * ld + c] = ex[(long long)row * E + c];
        }
        for (int e = threadIdx.x; e < RT * E; e += kBandThreads) {
            const int r = e / E, c = e - r * E;
            const int row = min(max(i0 - HW + r, 0), N - 1);
            st[r * ld + c] = et[(long long)row * E + c];
        }
    }
    __syncthreads();
    const int pt = threadIdx.x & (kBandTile - 1), grp = threadIdx.x / kBandTile;   // 4 groups
    const int e0 = grp * ((E + 3) / 4), e1 = min(E, e0 + (E + 3) / 4);
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.0f;
    for (int e = e0; e < e1; ++e) {
        const float xv = sx[pt * ld + e];
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = fmaf(xv, st[(pt + k) * ld + e], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < W; ++k) sred[(grp * kBandTile + pt) * W + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < kBandTile && i0 + pt < N) {
        const int i = i0 + pt;
        float sim[W];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            sim[k] = (sred[pt * W + k] + sred[(kBandTile + pt) * W + k]) +
                     (sred[(2 * kBandTile + pt) * W + k] + sred[(3 * kBandTile + pt) * W + k]);
            mx = fmaxf(mx, sim[k]);
        }
        float ev[W], sum = 0.0f;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int ju = i - HW + k;  // unclamped column: clamped duplicates get no weight
            ev[k] = (ju >= 0 && ju <= N - 1) ? expf(sim[k] - mx) : 0.0f;
            sum += ev[k];
        }
        const long long o = ((long long)b * N + i) * W;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            if (band) band[o + k] = sim[k];
            prob[o + k] = ev[k] / sum;
        }
    }
}

template <int W>
void launch_band(const float *emb_x, const float *emb_t, int B, int N, int E, float *band, float *prob,
                 hipStream_t s)
{
    const size_t lds = ((size_t)(2 * kBandTile + W - 1) * (E + 1) + 4 * kBandTile * W) * sizeof(float);
    attn_band_kernel<W><<<dim3((N + kBandTile - 1) / kBandTile, B), kBandThreads, lds, s>>>(emb_x, emb_t, N, E, band,
                                                                                            prob);
}

// TRANS = false: forward merge   out[i] = alpha*x[i] + (1-alpha) * sum_k prob[i][k] * tmpl[i-HW+k]
// TRANS = true : backward wrt the template, driven by the output gradient g (= `tmpl` argument):
//                out[j]  = (1-alpha) * sum_k prob[j-HW+k][W-1-k] * g[j-HW+k]     (d tmpl)
//                out2[j] = alpha * g[j]                                          (d x)
// x is read once and the outputs are written once: streaming loads / stores keep them from
// evicting the template rows that neighbouring segments re-read from L2.
using F4V = float __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(float4 *p, const float4 &v)
{
    F4V t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<F4V *>(p));
}
__device__ __forceinline__ float4 stream_load(const float4 *p)
{
    const F4V t = __builtin_nontemporal_load(reinterpret_cast<const F4V *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

template <int W, bool TRANS>
__global__ __launch_bounds__(128) void attn_merge_kernel(const float4 *x, const float4 *tmpl,
                                                         const float *prob, float4 *out, float4 *out2,
                                                         int N, int F4, int L, float alpha,
                                                         float one_minus_alpha)
{
    constexpr int HW = W / 2;
    const int col = blockIdx.x * 128 + threadIdx.x;
    if (col >= F4) return;
    const int b = blockIdx.z;
    const int s0 = blockIdx.y * L;
    const int s1 = min(N, s0 + L);
    const long long sample = (long long)b * N;
    const float4 *T = tmpl + sample * F4 + col;
    const float4 *X = TRANS ? T : x + sample * F4 + col;
    float4 *O = out + sample * F4 + col;
    float4 *O2 = TRANS ? out2 + sample * F4 + col : nullptr;
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
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        float pk;
                        if (TRANS) {
                            const int src = i - HW + k;  // row whose window contains column i at slot W-1-k
                            pk = (src >= 0 && src <= N - 1) ? P[(long long)src * W + (W - 1 - k)] : 0.0f;
                        } else {
                            pk = P[(long long)i * W + k];
                        }
                        const float4 t = win[(u + k + 1) % W];
                        acc.x = fmaf(pk, t.x, acc.x);
                        acc.y = fmaf(pk, t.y, acc.y);
                        acc.z = fmaf(pk, t.z, acc.z);
                        acc.w = fmaf(pk, t.w, acc.w);
                    }
                    float4 o;
                    if (TRANS) {
                        const float4 gv = win[(u + HW + 1) % W];  // row i itself
                        o = make_float4(one_minus_alpha * acc.x, one_minus_alpha * acc.y, one_minus_alpha * acc.z,
                                        one_minus_alpha * acc.w);
                        stream_store(O2 + (long long)i * F4, make_float4(alpha * gv.x, alpha * gv.y, alpha * gv.z, alpha * gv.w));
                    } else {
                        const float4 xv = stream_load(X + (long long)i * F4);
                        o.x = alpha * xv.x + one_minus_alpha * acc.x;
                        o.y = alpha * xv.y + one_minus_alpha * acc.y;
                        o.z = alpha * xv.z + one_minus_alpha * acc.z;
                        o.w = alpha * xv.w + one_minus_alpha * acc.w;
                    }
                    stream_store(O + (long long)i * F4, o);
                }
            }
        }
    }
}

template <int W, bool TRANS>
void launch_merge(const float *x, const float *tmpl, const float *prob, float *out, float *out2, int B, int N,
                  int F, int L, double alpha, hipStream_t s)
{
    const int F4 = F / 4;
    dim3 grid((F4 + 127) / 128, (N + L - 1) / L, B);
    attn_merge_kernel<W, TRANS><<<grid, 128, 0, s>>>(
        reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(tmpl), prob,
        reinterpret_cast<float4 *>(out), reinterpret_cast<float4 *>(out2), N, F4, L, (float)alpha,
        (float)(1.0 - alpha));
}

template <bool TRANS>
int dispatch_merge(int W, const float *x, const float *tmpl, const float *prob, float *out, float *out2, int B,
                   int N, int F, double alpha, hipStream_t s)
{
    // segment length: whole scan per lane when the batch alone fills the chip,
    // shorter segments (more workgroups, a little halo re-read) for small batches
    const long long colblocks = (F / 4 + 127) / 128;
    int L = N;
    while (L > 32 && colblocks * ((N + L - 1) / L) * B < 3072) L = (L + 1) / 2;
    switch (W) {
        case 1: launch_merge<1, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 3: launch_merge<3, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 5: launch_merge<5, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 7: launch_merge<7, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 9: launch_merge<9, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 11: launch_merge<11, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 13: launch_merge<13, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 15: launch_merge<15, TRANS>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        default: return POF_E_SHAPE;
    }
    return POF_OK;
}

// ---- backward --------------------------------------------------------------------
// One wave per point i: dp[k] = (1-alpha) <g[i], tmpl[i-HW+k]> over F (distinct columns
// only), softmax backward ds = p (dp - sum p dp), dsim = ds + g_band -> dsim[b,i,k].
__global__ __launch_bounds__(256) void attn_dsim_kernel(const float4 *g_out, const float4 *tmpl,
                                                        const float *prob, const float *g_band, int N,
                                                        int F4, int W, float one_minus_alpha, float *dsim)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= N) return;
    const int hw = W / 2;
    const float4 *g = g_out + ((long long)b * N + i) * F4;
    const float4 *T = tmpl + (long long)b * N * F4;
    float mine = 0.0f;
    for (int k = 0; k < W; ++k) {
        const int j = i - hw + k;
        float part = 0.0f;
        if (j >= 0 && j <= N - 1) {
            const float4 *tj = T + (long long)j * F4;
            for (int c = lane; c < F4; c += 64) {
                const float4 a = g[c], t = tj[c];
                part = fmaf(a.x, t.x, part);
                part = fmaf(a.y, t.y, part);
                part = fmaf(a.z, t.z, part);
                part = fmaf(a.w, t.w, part);
            }
        }
        part = wave_sum_f32(part);
        if (lane == k) mine = part * one_minus_alpha;
    }
    const bool slot = lane < W;
    const long long o = ((long long)b * N + i) * W + lane;
    const float p = slot ? prob[o] : 0.0f;
    const float s = wave_sum_f32(p * mine);
    if (slot) dsim[o] = p * (mine - s) + (g_band ? g_band[o] : 0.0f);
}

// d emb_x[i] = sum_k dsim[i,k] emb_t[clamp(i-hw+k)]
// d emb_t[j] = sum over (i,k) with clamp(i-hw+k) == j of dsim[i,k] emb_x[i]
__global__ __launch_bounds__(256) void attn_demb_kernel(const float *emb_x, const float *emb_t,
                                                        const float *dsim, int N, int E, int W,
                                                        float *d_emb_x, float *d_emb_t)
{
    const int b = blockIdx.y;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= N) return;
    const int hw = W / 2;
    const float *ex = emb_x + (long long)b * N * E;
    const float *et = emb_t + (long long)b * N * E;
    const float *ds = dsim + (long long)b * N * W;
    for (int e = lane; e < E; e += 64) {
        float ax = 0.0f, at = 0.0f;
        for (int k = 0; k < W; ++k) {
            const int j = min(max(r - hw + k, 0), N - 1);
            ax = fmaf(ds[(long long)r * W + k], et[(long long)j * E + e], ax);
        }
        // rows whose window reaches column r (r plays the role of j)
        for (int i = max(r - hw, 0); i <= min(r + hw, N - 1); ++i) {
            for (int k = 0; k < W; ++k) {
                const int j = min(max(i - hw + k, 0), N - 1);
                if (j == r) at = fmaf(ds[(long long)i * W + k], ex[(long long)i * E + e], at);
            }
        }
        d_emb_x[((long long)b * N + r) * E + e] = ax;
        d_emb_t[((long long)b * N + r) * E + e] = at;
    }
}

}  // namespace

extern "C" int pof_spatial_attention(const float *emb_x, const float *emb_t, const float *x,
                                     const float *tmpl, int B, int N, int E, int F, int window,
                                     double alpha, float *band, float *prob, float *out,
                                     pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
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
    if ((size_t)(2 * kBandTile + W - 1) * (E + 1) * sizeof(float) > 60 * 1024) return POF_E_SHAPE;  // E <= ~190
    switch (W) {
        case 1: launch_band<1>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 3: launch_band<3>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 5: launch_band<5>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 7: launch_band<7>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 9: launch_band<9>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 11: launch_band<11>(emb_x, emb_t, B, N, E, band, prob, s); break;
        case 13: launch_band<13>(emb_x, emb_t, B, N, E, band, prob, s); break;
        default: launch_band<15>(emb_x, emb_t, B, N, E, band, prob, s); break;
    }
    POF_CHECK_LAUNCH();
    const int rc = dispatch_merge<false>(W, x, tmpl, prob, out, nullptr, B, N, F, alpha, s);
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_spatial_attention_backward(const float *emb_x, const float *emb_t, const float *tmpl,
                                              const float *prob, const float *g_out, const float *g_band,
                                              int B, int N, int E, int F, int window, double alpha,
                                              float *dsim, float *d_emb_x, float *d_emb_t, float *d_x,
                                              float *d_tmpl, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!emb_x || !emb_t || !tmpl || !prob || !g_out || !dsim || !d_emb_x || !d_emb_t || !d_x || !d_tmpl)
        return POF_E_BADARG;
    if (B < 0 || N < 1 || E < 1 || F < 1) return POF_E_BADARG;
    const int W = 2 * (window / 2) + 1;
    if (W < 1 || W > kMaxW) return POF_E_SHAPE;
    if (F % 4 != 0) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    hipStream_t s = pof_stream(stream);
    attn_dsim_kernel<<<dim3((N + 3) / 4, B), 256, 0, s>>>(reinterpret_cast<const float4 *>(g_out),
                                                          reinterpret_cast<const float4 *>(tmpl), prob, g_band,
                                                          N, F / 4, W, (float)(1.0 - alpha), dsim);
    POF_CHECK_LAUNCH();
    attn_demb_kernel<<<dim3((N + 3) / 4, B), 256, 0, s>>>(emb_x, emb_t, dsim, N, E, W, d_emb_x, d_emb_t);
    POF_CHECK_LAUNCH();
    const int rc = dispatch_merge<true>(W, nullptr, g_out, prob, d_tmpl, d_x, B, N, F, alpha, s);
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
}
