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
// Storage type T = float or _Float16 (BASELINE config 5: float16 storage, float32 arithmetic): a
// lane owns 4 consecutive elements of a row (16 or 8 bytes) and computes on them as float4.
template <typename T>
using Col4 = T __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ float4 col_load(const Col4<T> *p)
{
    const Col4<T> t = *p;
    return make_float4((float)t.x, (float)t.y, (float)t.z, (float)t.w);
}
template <typename T>
__device__ __forceinline__ void stream_store(Col4<T> *p, const float4 &v)
{
    Col4<T> t = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
    __builtin_nontemporal_store(t, p);
}
template <typename T>
__device__ __forceinline__ float4 stream_load(const Col4<T> *p)
{
    const Col4<T> t = __builtin_nontemporal_load(p);
    return make_float4((float)t.x, (float)t.y, (float)t.z, (float)t.w);
}

template <int W, bool TRANS, typename T>
__global__ __launch_bounds__(128) void attn_merge_kernel(const Col4<T> *x, const Col4<T> *tmpl,
                                                         const float *prob, Col4<T> *out, Col4<T> *out2,
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
    const Col4<T> *Tm = tmpl + sample * F4 + col;
    const Col4<T> *X = TRANS ? Tm : x + sample * F4 + col;
    Col4<T> *O = out + sample * F4 + col;
    Col4<T> *O2 = TRANS ? out2 + sample * F4 + col : nullptr;
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
                win[u] = (r >= 0 && r <= N - 1) ? col_load<T>(Tm + (long long)r * F4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                        stream_store<T>(O2 + (long long)i * F4, make_float4(alpha * gv.x, alpha * gv.y, alpha * gv.z, alpha * gv.w));
                    } else {
                        const float4 xv = stream_load<T>(X + (long long)i * F4);
                        o.x = alpha * xv.x + one_minus_alpha * acc.x;
                        o.y = alpha * xv.y + one_minus_alpha * acc.y;
                        o.z = alpha * xv.z + one_minus_alpha * acc.z;
                        o.w = alpha * xv.w + one_minus_alpha * acc.w;
                    }
                    stream_store<T>(O + (long long)i * F4, o);
                }
            }
        }
    }
}

template <int W, bool TRANS, typename T>
void launch_merge(const T *x, const T *tmpl, const float *prob, T *out, T *out2, int B, int N,
                  int F, int L, double alpha, hipStream_t s)
{
    const int F4 = F / 4;
    dim3 grid((F4 + 127) / 128, (N + L - 1) / L, B);
    attn_merge_kernel<W, TRANS, T><<<grid, 128, 0, s>>>(
        reinterpret_cast<const Col4<T> *>(x), reinterpret_cast<const Col4<T> *>(tmpl), prob,
        reinterpret_cast<Col4<T> *>(out), reinterpret_cast<Col4<T> *>(out2), N, F4, L, (float)alpha,
        (float)(1.0 - alpha));
}

template <bool TRANS, typename T>
int dispatch_merge(int W, const T *x, const T *tmpl, const float *prob, T *out, T *out2, int B,
                   int N, int F, double alpha, hipStream_t s)
{
    // segment length: whole scan per lane when the batch alone fills the chip,
    // shorter segments (more workgroups, a little halo re-read) for small batches
    const long long colblocks = (F / 4 + 127) / 128;
    int L = N;
    // small batches: segments down to 8 points (a 10-row halo per segment re-read from L2) -- at one scan per call the
    // walk is bound by its 105 workgroups, not by bytes: B = 1 forward 37 -> 27 us, B = 8 69 -> 46 us (tools/exp_attn_small.py)
    while (L > 8 && colblocks * ((N + L - 1) / L) * B < 3072) L = (L + 1) / 2;
    switch (W) {
        case 1: launch_merge<1, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 3: launch_merge<3, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 5: launch_merge<5, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 7: launch_merge<7, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 9: launch_merge<9, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 11: launch_merge<11, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 13: launch_merge<13, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        case 15: launch_merge<15, TRANS, T>(x, tmpl, prob, out, out2, B, N, F, L, alpha, s); break;
        default: return POF_E_SHAPE;
    }
    return POF_OK;
}

// ---- backward --------------------------------------------------------------------
// One wave per 16 consecutive points: dp[i][k] = (1-alpha) <g[i], tmpl[i-hw+k]> is the band of the
// 16 x 32 block  P = G_rows x Tmpl_rows^T  (rows i0..i0+15 against rows i0-hw..i0-hw+31), contracted
// over F on the float32 MFMA (v_mfma_f32_16x16x4_f32; lane (r, q) holds row r, k = q).  F is the
// contiguous axis, so every lane loads 16 bytes of its row per 16-float step and the four MFMAs of
// the step take elements 0..3 (both operands use the same permuted k order); 4 lanes cover 64
// contiguous bytes of a row.  Steps are register double-buffered in blocks of kDsU.  The epilogue
// drops the block into LDS and 16 lanes run the softmax backward of their row:
//   ds = p * (dp - sum_k p dp),  dsim = ds + g_band.
constexpr int kDsWaves = 4;
constexpr int kDsU = 4;
using f32x4 = float __attribute__((ext_vector_type(4)));

//
// FWD = true is the forward similarity with the same structure (rows of emb_x against rows of emb_t,
// contracted over E): band[i][k] = <emb_x[i], emb_t[clamp(i-hw+k)]> (clamped duplicates kept) and
// the masked softmax over the distinct columns -> prob (written through `dsim`), band through `g_band_out`.
template <bool FWD>
__global__ __launch_bounds__(64 * kDsWaves) void attn_dsim_kernel(const float *g_out, const float *tmpl,
                                                                 const float *prob, const float *g_band, int B, int N,
                                                                 int F, int W, float one_minus_alpha, float *dsim,
                                                                 float *band_out)
{
    __shared__ float s_p[kDsWaves][16 * 33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblk = (N + 15) >> 4;
    const int unit_raw = blockIdx.x * kDsWaves + wave;
    const bool live = unit_raw < B * nblk;
    const int unit = live ? unit_raw : B * nblk - 1;   // tail waves recompute the last unit, store nothing
    const int b = unit / nblk, i0 = (unit - b * nblk) * 16;
    const int hw = W / 2, cb = i0 - hw;
    const int r = lane & 15, q = lane >> 4;
    const int ri = i0 + r, j0 = cb + r, j1 = cb + 16 + r;
    const bool vi = ri < N, v0 = j0 >= 0 && j0 < N, v1 = j1 >= 0 && j1 < N;
    const float *base_g = g_out + (long long)b * N * F;
    const float *base_t = tmpl + (long long)b * N * F;
    // per-lane element offsets (rows clamped so that masked lanes still load inside the sample)
    const long long og = (long long)min(ri, N - 1) * F + 4 * q;
    const long long o0 = (long long)min(max(j0, 0), N - 1) * F + 4 * q;
    const long long o1 = (long long)min(max(j1, 0), N - 1) * F + 4 * q;
    const int steps = (F + 15) >> 4;

    f32x4 acc0 = {0}, acc1 = {0};
    f32x4 xg[2][kDsU], x0[2][kDsU], x1[2][kDsU];
    auto load_block = [&](int set, int s0) {
#pragma unroll
        for (int u = 0; u < kDsU; ++u) {
            // a step past the end, or the last lanes of a partial step, re-read the row's last 16 bytes
            const int f = min(16 * min(s0 + u, steps - 1) + 4 * q, F - 4) - 4 * q;
            xg[set][u] = *reinterpret_cast<const f32x4 *>(base_g + og + f);
            x0[set][u] = *reinterpret_cast<const f32x4 *>(base_t + o0 + f);
            x1[set][u] = *reinterpret_cast<const f32x4 *>(base_t + o1 + f);
        }
    };
    auto mac_block = [&](int set, int s0) {
#pragma unroll
        for (int u = 0; u < kDsU; ++u) {
            const bool ok = (s0 + u < steps) && (16 * (s0 + u) + 4 * q < F);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a = (vi && ok) ? xg[set][u][t] : 0.0f;
                const float b0 = (v0 && ok) ? x0[set][u][t] : 0.0f;
                const float b1 = (v1 && ok) ? x1[set][u][t] : 0.0f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc1, 0, 0, 0);
            }
        }
    };
    load_block(0, 0);
    for (int s0 = 0; s0 < steps; s0 += 2 * kDsU) {
        load_block(1, s0 + kDsU);
        mac_block(0, s0);
        if (s0 + kDsU < steps) {
            load_block(0, s0 + 2 * kDsU);
            mac_block(1, s0 + kDsU);
        }
    }
    // C/D layout (16x16): col = lane & 15, row = 4 * (lane >> 4) + reg
    float *P = s_p[wave];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        P[(4 * q + reg) * 33 + r] = acc0[reg];
        P[(4 * q + reg) * 33 + 16 + r] = acc1[reg];
    }
    __syncthreads();
    if (lane < 16 && live && i0 + lane < N) {
        const int i = i0 + lane;
        const long long o = ((long long)b * N + i) * W;
        if (FWD) {
            float mx = -INFINITY;
            for (int k = 0; k < W; ++k) {
                const int jc = min(max(i - hw + k, 0), N - 1);
                mx = fmaxf(mx, P[lane * 33 + jc - cb]);
            }
            float sum = 0.0f;
            for (int k = 0; k < W; ++k) {
                const int ju = i - hw + k;  // unclamped column: clamped duplicates get no weight
                if (ju >= 0 && ju <= N - 1) sum += expf(P[lane * 33 + ju - cb] - mx);
            }
            for (int k = 0; k < W; ++k) {
                const int ju = i - hw + k;
                const float sim = P[lane * 33 + min(max(ju, 0), N - 1) - cb];
                if (band_out) band_out[o + k] = sim;
                dsim[o + k] = (ju >= 0 && ju <= N - 1) ? expf(sim - mx) / sum : 0.0f;
            }
        } else {
            float s = 0.0f;
            for (int k = 0; k < W; ++k) {
                const int j = i - hw + k;   // unclamped column: out-of-range slots carry no weight
                const float dp = (j >= 0 && j <= N - 1) ? P[lane * 33 + lane + k] * one_minus_alpha : 0.0f;
                s = fmaf(prob[o + k], dp, s);
            }
            for (int k = 0; k < W; ++k) {
                const int j = i - hw + k;
                const float dp = (j >= 0 && j <= N - 1) ? P[lane * 33 + lane + k] * one_minus_alpha : 0.0f;
                dsim[o + k] = prob[o + k] * (dp - s) + (g_band ? g_band[o + k] : 0.0f);
            }
        }
    }
}

// d emb_x[i] = sum_k dsim[i,k] emb_t[clamp(i-hw+k)]
// d emb_t[j] = sum over (i,k) with clamp(i-hw+k) == j of dsim[i,k] emb_x[i]
// One workgroup = kDeTile consecutive points of one sample.  The emb rows its windows meet (tile + 2 hw
// rows, clamped at the ends of the scan exactly like the forward's gather) are staged in LDS once with
// coalesced loads, the dsim rows next to them; the clamp(i-hw+k) == j bookkeeping of d emb_t collapses
// into one weight per (j, i): dsim[i][j-i+hw] in the interior, the sum of the run of k that the clamp
// folds onto j at the two ends of the scan.  Every output is then W multiply-adds out of LDS
// (the previous form walked global memory per lane: 77 us at B = 64, 14 % of the attention backward).
constexpr int kDeTile = 16;
constexpr int kDeChunk = 128;          // embedding columns per workgroup (blockIdx.z walks the chunks)

__global__ __launch_bounds__(256) void attn_demb_kernel(const float *emb_x, const float *emb_t,
                                                        const float *dsim, int N, int E, int W,
                                                        float *d_emb_x, float *d_emb_t)
{
    extern __shared__ __align__(16) float sde[];
    const int hw = W / 2, R = kDeTile + 2 * hw;
    const int Efull = E, e_first = blockIdx.z * kDeChunk;
    E = min(kDeChunk, Efull - e_first);      // from here on: the chunk
    float *s_t = sde;                 // [R][E]  emb_t rows clamp(i0 - hw + q)
    float *s_x = s_t + R * E;         // [R][E]  emb_x rows i0 - hw + q (zero outside the scan)
    float *s_d = s_x + R * E;         // [R][W]  dsim rows i0 - hw + q (zero outside the scan)
    float *s_w = s_d + R * W;         // [kDeTile][W] weight of row (j - hw + i') in d emb_t[j]
    const int b = blockIdx.y, i0 = blockIdx.x * kDeTile, tid = threadIdx.x;
    const float *ex = emb_x + (long long)b * N * Efull + e_first;
    const float *et = emb_t + (long long)b * N * Efull + e_first;
    const float *ds = dsim + (long long)b * N * W;
    for (int idx = tid; idx < R * E; idx += 256) {
        const int q = idx / E, e = idx - q * E;
        const int row = i0 - hw + q;
        s_t[idx] = et[(long long)min(max(row, 0), N - 1) * Efull + e];
        s_x[idx] = (row >= 0 && row < N) ? ex[(long long)row * Efull + e] : 0.0f;
    }
    for (int idx = tid; idx < R * W; idx += 256) {
        const int q = idx / W, k = idx - q * W;
        const int row = i0 - hw + q;
        s_d[idx] = (row >= 0 && row < N) ? ds[(long long)row * W + k] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < kDeTile * W; idx += 256) {
        const int p = idx / W, ip = idx - p * W;      // target row j = i0 + p, source row i = j - hw + ip
        const int j = i0 + p, i = j - hw + ip;
        float wsum = 0.0f;
        if (j < N && i >= 0 && i < N) {
            // rows whose window reaches column j: clamp(i-hw+k) == j has the single solution k = j-i+hw for an
            // interior j and a run of k at the two ends of the scan
            int k_lo = j - i + hw, k_hi = k_lo;
            if (j == 0) k_lo = 0;
            if (j == N - 1) k_hi = W - 1;
            k_lo = max(k_lo, 0);
            k_hi = min(k_hi, W - 1);
            const float *drow = s_d + (p + ip) * W;   // staged row index of i: i - (i0 - hw) = p + ip
            for (int k = k_lo; k <= k_hi; ++k) wsum += drow[k];
        }
        s_w[idx] = wsum;
    }
    __syncthreads();
    for (int idx = tid; idx < kDeTile * E; idx += 256) {
        const int p = idx / E, e = idx - p * E;
        const int i = i0 + p;
        if (i >= N) break;
        const float *drow = s_d + (p + hw) * W;       // dsim row of point i
        float ax = 0.0f, at = 0.0f;
        for (int k = 0; k < W; ++k) {
            ax = fmaf(drow[k], s_t[(p + k) * E + e], ax);
            at = fmaf(s_w[p * W + k], s_x[(p + k) * E + e], at);
        }
        d_emb_x[((long long)b * N + i) * Efull + e_first + e] = ax;
        d_emb_t[((long long)b * N + i) * Efull + e_first + e] = at;
    }
}

}  // namespace

namespace {
template <typename T>
int attention_entry(const float *emb_x, const float *emb_t, const T *x, const T *tmpl, int B, int N, int E, int F,
                    int window, double alpha, float *band, float *prob, T *out, pof_stream_t stream)
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
    if (E % 4 == 0) {
        // banded emb_x . emb_t^T on the float32 MFMA (same kernel as the backward's g . tmpl^T)
        const long long units = (long long)B * ((N + 15) / 16);
        attn_dsim_kernel<true><<<(unsigned)((units + kDsWaves - 1) / kDsWaves), 64 * kDsWaves, 0, s>>>(
            emb_x, emb_t, nullptr, nullptr, B, N, E, W, 0.0f, prob, band);
    } else {
        // embedding sizes that are not a multiple of 4 floats: LDS-tiled VALU form
        if ((size_t)(2 * kBandTile + W - 1) * (E + 1) * sizeof(float) > 60 * 1024) return POF_E_SHAPE;
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
    }
    POF_CHECK_LAUNCH();
    const int rc = dispatch_merge<false, T>(W, x, tmpl, prob, out, static_cast<T *>(nullptr), B, N, F, alpha, s);
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
}
}  // namespace

extern "C" int pof_spatial_attention(const float *emb_x, const float *emb_t, const float *x,
                                     const float *tmpl, int B, int N, int E, int F, int window,
                                     double alpha, float *band, float *prob, float *out,
                                     pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return attention_entry<float>(emb_x, emb_t, x, tmpl, B, N, E, F, window, alpha, band, prob, out, stream);
}

extern "C" int pof_spatial_attention_f16(const float *emb_x, const float *emb_t, const void *x_f16,
                                         const void *tmpl_f16, int B, int N, int E, int F, int window,
                                         double alpha, float *band, float *prob, void *out_f16,
                                         pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return attention_entry<_Float16>(emb_x, emb_t, static_cast<const _Float16 *>(x_f16),
                                     static_cast<const _Float16 *>(tmpl_f16), B, N, E, F, window, alpha, band, prob,
                                     static_cast<_Float16 *>(out_f16), stream);
}

namespace {

// ---- backward, fused form -------------------------------------------------------------------------
// The two large passes of the backward both read the output gradient g: the transposed merge (d tmpl, d x) and the
// band product dp[i][k] = <g[i], tmpl[i-hw+k]>.  Here they are ONE walk down the points: a lane owns one 16-byte
// column of the rows and keeps two W-deep register rings, of g rows (for d tmpl, as the transposed merge does) and
// of tmpl rows; at point i the W dot products of its 4 columns are summed over the wave's 64 lanes (DPP adds) and lane 63
// writes them to partial[column block][b][i][k].  g, tmpl are read once, d tmpl, d x written once: the
// algorithmic 4 * N * F * 4 bytes (the two-pass form reads g twice and tmpl twice).  A small finishing pass sums
// the column blocks in a fixed order (deterministic, no atomics) and runs the softmax backward of each row.
template <int W>
__global__ __launch_bounds__(128) void attn_bwd_fused_kernel(const float4 *__restrict__ g, const float4 *__restrict__ tmpl,
                                                             const float *__restrict__ prob, float4 *__restrict__ d_tmpl,
                                                             float4 *__restrict__ d_x, float *__restrict__ partial,
                                                             int B, int N, int F4, int L, float alpha,
                                                             float one_minus_alpha)
{
    constexpr int HW = W / 2;
    const int col = blockIdx.x * 128 + threadIdx.x;
    const bool active = col < F4;                     // inactive lanes stay for the wave sums, contribute zeros
    const int colc = active ? col : F4 - 1;
    const int b = blockIdx.z;
    const int s0 = blockIdx.y * L;
    const int s1 = min(N, s0 + L);
    const long long sample = (long long)b * N;
    const float4 *G = g + sample * F4 + colc;
    const float4 *Tm = tmpl + sample * F4 + colc;
    float4 *DT = d_tmpl + sample * F4 + colc;
    float4 *DX = d_x + sample * F4 + colc;
    const float *P = prob + sample * W;
    const int cb = blockIdx.x * 2 + (threadIdx.x >> 6);            // column block = one wave = 256 floats of a row
    float *part = partial + ((long long)cb * B + b) * N * W;
    const int lane = threadIdx.x & 63;
    const int base = s0 - HW;
    const int rmax = s1 - 1 + HW;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 win[W], twin[W];
#pragma unroll
    for (int u = 0; u < W; ++u) { win[u] = zero; twin[u] = zero; }

    for (int m = 0; base + m * W <= rmax; ++m) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const int r = base + m * W + u;
            if (r <= rmax) {
                const bool in = r >= 0 && r <= N - 1 && active;
                win[u] = in ? G[(long long)r * F4] : zero;
                twin[u] = in ? Tm[(long long)r * F4] : zero;
                const int i = r - HW;
                if (i >= s0) {
                    float4 acc = zero;
                    const float4 gv = win[(u + HW + 1) % W];          // row i of g
                    float dp[W];
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        const int src = i - HW + k;   // row whose window contains column i at slot W-1-k
                        const float pk = (src >= 0 && src <= N - 1) ? P[(long long)src * W + (W - 1 - k)] : 0.0f;
                        const float4 t = win[(u + k + 1) % W];
                        acc.x = fmaf(pk, t.x, acc.x);
                        acc.y = fmaf(pk, t.y, acc.y);
                        acc.z = fmaf(pk, t.z, acc.z);
                        acc.w = fmaf(pk, t.w, acc.w);
                        const float4 tt = twin[(u + k + 1) % W];       // tmpl row i - HW + k
                        dp[k] = fmaf(gv.w, tt.w, fmaf(gv.z, tt.z, fmaf(gv.y, tt.y, gv.x * tt.x)));
                    }
                    if (active) {
                        stream_store<float>(reinterpret_cast<Col4<float> *>(DX + (long long)i * F4),
                                            make_float4(alpha * gv.x, alpha * gv.y, alpha * gv.z, alpha * gv.w));
                        stream_store<float>(reinterpret_cast<Col4<float> *>(DT + (long long)i * F4),
                                            make_float4(one_minus_alpha * acc.x, one_minus_alpha * acc.y,
                                                        one_minus_alpha * acc.z, one_minus_alpha * acc.w));
                    }
#pragma unroll
                    for (int k = 0; k < W; ++k) dp[k] = wave_sum_to_lane63_f32(dp[k]);   // vector pipe, not 66 ds_bpermute
                    if (lane == 63) {
#pragma unroll
                        for (int k = 0; k < W; ++k) part[(long long)i * W + k] = dp[k];
                    }
                }
            }
        }
    }
}

// dsim[i][k] = p * (dp - sum_k p dp) + g_band, dp = (1 - alpha) * sum over the column blocks (in order)
__global__ __launch_bounds__(256) void attn_dsim_finish_kernel(const float *__restrict__ partial, int ncb,
                                                               const float *__restrict__ prob,
                                                               const float *__restrict__ g_band, long long rows, int N,
                                                               int W, float one_minus_alpha, float *__restrict__ dsim)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;      // (b, i)
    if (row >= rows) return;
    const int i = (int)(row % N), hw = W / 2;
    float dp[kMaxW];
    float s = 0.0f;
    for (int k = 0; k < W; ++k) {
        float v = 0.0f;
        for (int c = 0; c < ncb; ++c) v += partial[((long long)c * rows + row) * W + k];
        const int j = i - hw + k;            // unclamped column: out-of-range slots carry no weight
        dp[k] = (j >= 0 && j <= N - 1) ? v * one_minus_alpha : 0.0f;
        s = fmaf(prob[row * W + k], dp[k], s);
    }
    for (int k = 0; k < W; ++k)
        dsim[row * W + k] = prob[row * W + k] * (dp[k] - s) + (g_band ? g_band[row * W + k] : 0.0f);
}

template <int W>
void launch_bwd_fused(const float *g, const float *tmpl, const float *prob, float *d_tmpl, float *d_x, float *partial,
                      int B, int N, int F, int L, double alpha, hipStream_t s)
{
    const int F4 = F / 4;
    dim3 grid((F4 + 127) / 128, (N + L - 1) / L, B);
    attn_bwd_fused_kernel<W><<<grid, 128, 0, s>>>(reinterpret_cast<const float4 *>(g),
                                                  reinterpret_cast<const float4 *>(tmpl), prob,
                                                  reinterpret_cast<float4 *>(d_tmpl), reinterpret_cast<float4 *>(d_x),
                                                  partial, B, N, F4, L, (float)alpha, (float)(1.0 - alpha));
}

}  // namespace

extern "C" size_t pof_spatial_attention_backward_workspace_bytes(int B, int N, int F, int window)
{
    const int W = 2 * (window / 2) + 1;
    if (B < 1 || N < 1 || F < 4 || W < 1) return 0;
    const size_t ncb = 2 * (size_t)((F / 4 + 127) / 128);
    return ncb * (size_t)B * N * W * sizeof(float);
}

extern "C" int pof_spatial_attention_backward_fused(const float *emb_x, const float *emb_t, const float *tmpl,
                                                    const float *prob, const float *g_out, const float *g_band,
                                                    int B, int N, int E, int F, int window, double alpha,
                                                    float *dsim, float *d_emb_x, float *d_emb_t, float *d_x,
                                                    float *d_tmpl, void *workspace, size_t workspace_bytes,
                                                    pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!emb_x || !emb_t || !tmpl || !prob || !g_out || !dsim || !d_emb_x || !d_emb_t || !d_x || !d_tmpl || !workspace)
        return POF_E_BADARG;
    if (B < 0 || N < 1 || E < 1 || F < 1) return POF_E_BADARG;
    const int W = 2 * (window / 2) + 1;
    if (W < 1 || W > kMaxW) return POF_E_SHAPE;
    if (F % 4 != 0) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    if (workspace_bytes < pof_spatial_attention_backward_workspace_bytes(B, N, F, window)) return POF_E_WORKSPACE;
    hipStream_t s = pof_stream(stream);
    float *partial = static_cast<float *>(workspace);
    const int ncb = 2 * ((F / 4 + 127) / 128);
    {
        // segment length as for the merge kernel: whole scan per lane when the batch alone fills the chip
        const long long colblocks = (F / 4 + 127) / 128;
        int L = N;
            // the fused walk carries two rings: 16-point segments are its optimum at small batches (B = 1 70 -> 57 us;
        // 8-point segments lose again at B = 8)
        while (L > 16 && colblocks * ((N + L - 1) / L) * B < 3072) L = (L + 1) / 2;
        switch (W) {
            case 1: launch_bwd_fused<1>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 3: launch_bwd_fused<3>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 5: launch_bwd_fused<5>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 7: launch_bwd_fused<7>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 9: launch_bwd_fused<9>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 11: launch_bwd_fused<11>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 13: launch_bwd_fused<13>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            case 15: launch_bwd_fused<15>(g_out, tmpl, prob, d_tmpl, d_x, partial, B, N, F, L, alpha, s); break;
            default: return POF_E_SHAPE;
        }
    }
    POF_CHECK_LAUNCH();
    {
        const long long rows = (long long)B * N;
        attn_dsim_finish_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, s>>>(partial, ncb, prob, g_band, rows, N, W,
                                                                              (float)(1.0 - alpha), dsim);
    }
    POF_CHECK_LAUNCH();
    {
        const int ec = E < kDeChunk ? E : kDeChunk;
        const size_t lds = ((size_t)2 * (kDeTile + W - 1) * ec + (size_t)(kDeTile + W - 1) * W + (size_t)kDeTile * W) * sizeof(float);
        attn_demb_kernel<<<dim3((N + kDeTile - 1) / kDeTile, B, (E + kDeChunk - 1) / kDeChunk), 256, lds, s>>>(
            emb_x, emb_t, dsim, N, E, W, d_emb_x, d_emb_t);
    }
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
    {
        const long long units = (long long)B * ((N + 15) / 16);
        attn_dsim_kernel<false><<<(unsigned)((units + kDsWaves - 1) / kDsWaves), 64 * kDsWaves, 0, s>>>(
            g_out, tmpl, prob, g_band, B, N, F, W, (float)(1.0 - alpha), dsim, nullptr);
    }
    POF_CHECK_LAUNCH();
    {
        const int ec = E < kDeChunk ? E : kDeChunk;
        const size_t lds = ((size_t)2 * (kDeTile + W - 1) * ec + (size_t)(kDeTile + W - 1) * W + (size_t)kDeTile * W) * sizeof(float);
        attn_demb_kernel<<<dim3((N + kDeTile - 1) / kDeTile, B, (E + kDeChunk - 1) / kDeChunk), 256, lds, s>>>(
            emb_x, emb_t, dsim, N, E, W, d_emb_x, d_emb_t);
    }
    POF_CHECK_LAUNCH();
    const int rc = dispatch_merge<true, float>(W, static_cast<const float *>(nullptr), g_out, prob, d_tmpl, d_x, B, N, F, alpha, s);
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
}
