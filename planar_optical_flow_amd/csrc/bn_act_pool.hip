// N2, training side: the BatchNorm1d(train) + LeakyReLU [+ max_pool1d(2)] tail of every trunk unit
// (src/depracted/model/dr_spaam.py:8-19 `_conv`, :86-92 `_conv_and_pool`), forward and backward.
//
// The convolution passes of a training step run on conv3_kernel / conv3_wgrad_kernel (conv_trunk.hip,
// conv_wgrad.hip); what costs as much around them in the framework's own form is the element-wise tail: at the
// reference's batch (8 scans x 450 cutouts x 5 scans = 18 000 sequences of 48 points) the framework's BatchNorm /
// LeakyReLU / max-pool kernels and their backward passes take 17 ms of a 41 ms step (profiles/r2_train_step_kernel_stats.txt) -- sequences of 6..48
// points are a poor fit for kernels written for image planes.  Here the tail is four streaming passes:
//
//   forward   bn_stats      read y                 -> per-(chunk, channel) sum / sum of squares (float64)
//             bn_finalize   C workgroups           -> mean, 1/std, scale, shift, running statistics
//             bn_apply      read y, write z        -> z = [pool](lrelu(y * scale + shift))
//   backward  bn_bwd_reduce read y, dz             -> per-(chunk, channel) sum dU, sum dU * xhat
//             bn_bwd_final  C workgroups           -> dgamma, dbeta, the two per-channel means
//             bn_bwd_dgrad  read y, dz, write dy   -> dy = gamma / std * (dU - mean(dU) - xhat * mean(dU * xhat))
//
// with dU = (dz routed to the pool's winner) * lrelu'(u), u = y * scale + shift recomputed from y (nothing but y
// and 2C floats is kept for the backward pass: no activation copy, no pool indices).
//
// Layout: y [S][C][L] float32 as the convolution wrote it.  One sample's [C][L] plane is P = C * L contiguous
// floats; a lane owns 4 consecutive plane positions (one 16-byte load) and walks K samples, so its channel and
// per-channel constants are fixed for the whole loop, a wave reads 1 KB runs, and the loads of successive
// samples are independent.  A workgroup's plane slice is a whole number of channels (W = multiple of
// lcm(L, 4) <= 1024 positions), so the statistics of a channel never straddle workgroups and the partial sums
// are deterministic (no atomics).  HBM-bound: algorithmic bytes per element 4 (stats), 8 or 6 (apply),
// 8 or 6 (reduce), 12 or 10 (dgrad) -- the second figure with pooling.
#include "pof_common.h"

namespace {

constexpr int kBnThreads = 256;
constexpr int kBnSlice = 4 * kBnThreads;      // plane positions per workgroup, at most

struct BnGeo {
    long long S;        // sequences
    long long nchunk;   // sample chunks (grid.x) = G * ncg
    long long Sg, ncg;  // sequences and chunks per statistics group
    int G;              // statistics groups: `groups` equal contiguous ranges of the sequences, each normalised with
                        // its own batch statistics (the five scans of a DR-SPAAM window in one launch)
    int C, L, P;        // channels, points per sequence, P = C * L
    int W;              // positions per plane slice (multiple of lcm(L, 4))
    int K;              // samples per chunk
    int nslice;         // grid.y
};

int gcd_int(int a, int b) { return b ? gcd_int(b, a % b) : a; }

// K samples per workgroup: enough workgroups to fill the part several times over, few enough partial sums
bool make_geo(long long S, int C, int L, int G, int wg_target, int kmax, BnGeo *g)
{
    if (S <= 0 || C <= 0 || L <= 0 || L > 256 || G < 1 || S % G != 0) return false;
    const long long P = (long long)C * L;
    if (P % 4 != 0 || P > (1ll << 30)) return false;
    const int unit = L / gcd_int(L, 4) * 4;
    g->S = S; g->C = C; g->L = L; g->P = (int)P;
    g->W = kBnSlice / unit * unit;
    g->nslice = (int)((P + g->W - 1) / g->W);
    if (g->nslice > 65535) return false;
    g->G = G;
    g->Sg = S / G;
    long long want = wg_target / g->nslice / G;
    if (want < 1) want = 1;
    long long K = (g->Sg + want - 1) / want;
    if (K < 1) K = 1;
    if (K > kmax) K = kmax;
    g->K = (int)K;
    g->ncg = (g->Sg + K - 1) / K;
    g->nchunk = g->ncg * G;
    return g->nchunk <= 0x7fffffffll;
}

// Row loads keep the default cache policy -- the apply pass re-reads what the statistics pass just read, part of it still
// in the Infinity Cache (non-temporal loads: 0.125 -> 0.150 ms on the forward of [18000 x 64 x 56]); the outputs are
// written once and read by another kernel much later: non-temporal stores (4.7 -> 6.1 TB/s on the same pass).
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
__device__ __forceinline__ void st4(float *p, float a, float b, float c, float d)
{
    using F4 = float __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(F4{a, b, c, d}, reinterpret_cast<F4 *>(p));
}
__device__ __forceinline__ void st2(float *p, float a, float b)
{
    using F2 = float __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(F2{a, b}, reinterpret_cast<F2 *>(p));
}

template <int POOL> constexpr int kSampleUnroll = POOL == 2 ? 1 : 4;

struct Lane {
    int p0;          // first plane position of the lane
    int grp;         // statistics group of the workgroup's samples
    bool active;
    long long s0, s1;
};

__device__ __forceinline__ Lane lane_of(const BnGeo &g)
{
    Lane l;
    const int off = 4 * threadIdx.x;
    l.p0 = blockIdx.y * g.W + off;
    l.active = off < g.W && l.p0 < g.P;
    l.grp = (int)(blockIdx.x / g.ncg);
    const long long chunk = blockIdx.x - l.grp * g.ncg, end = (l.grp + 1) * g.Sg;
    l.s0 = l.grp * g.Sg + chunk * g.K;
    l.s1 = l.s0 + g.K < end ? l.s0 + g.K : end;
    return l;
}

// sum the per-position partials of every channel of this slice and write them to partial[c][chunk]
__device__ __forceinline__ void channel_reduce(const BnGeo &g, const double (&a)[4], const double (&b)[4],
                                               double *s_a, double *s_b, double2 *partial)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s_a[4 * threadIdx.x + i] = a[i];
        s_b[4 * threadIdx.x + i] = b[i];
    }
    __syncthreads();
    const int base = blockIdx.y * g.W;
    const int span = g.P - base < g.W ? g.P - base : g.W;
    const int nch = span / g.L, c0 = base / g.L;
    // `tpc` lanes per channel, each sums a strided share of the channel's L positions; a shuffle tree joins them
    // (tpc is a power of two <= 8, so the lanes of a channel sit in one wave)
    const int tpc = nch * 8 <= kBnThreads ? 8 : nch * 4 <= kBnThreads ? 4 : nch * 2 <= kBnThreads ? 2 : 1;
    for (int j0 = 0; j0 < nch; j0 += kBnThreads / tpc) {
        const int j = j0 + threadIdx.x / tpc, r = threadIdx.x % tpc;
        double sa = 0.0, sb = 0.0;
        if (j < nch)
            for (int i = r; i < g.L; i += tpc) { sa += s_a[j * g.L + i]; sb += s_b[j * g.L + i]; }
        for (int o = tpc >> 1; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
        if (j < nch && r == 0) partial[(long long)(c0 + j) * g.nchunk + blockIdx.x] = make_double2(sa, sb);
    }
}

__global__ __launch_bounds__(kBnThreads) void bn_stats_kernel(const float *__restrict__ y, BnGeo g,
                                                              double2 *__restrict__ partial)
{
    __shared__ double s_a[kBnSlice], s_b[kBnSlice];
    const Lane l = lane_of(g);
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (l.active) {
        const float *src = y + l.s0 * g.P + l.p0;
#pragma unroll 4
        for (long long s = l.s0; s < l.s1; ++s, src += g.P) {
            const float4 v = ld4(src);
            const double d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += d[i]; b[i] = fma(d[i], d[i], b[i]); }
        }
    }
    channel_reduce(g, a, b, s_a, s_b, partial);
}

__device__ __forceinline__ double2 block_sum2(double2 v, double2 *s_w)
{
    v.x = wave_sum_f64(v.x);
    v.y = wave_sum_f64(v.y);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    double2 t = make_double2(0.0, 0.0);
    for (int w = 0; w < kBnThreads / 64; ++w) { t.x += s_w[w].x; t.y += s_w[w].y; }
    return t;
}

__global__ __launch_bounds__(kBnThreads) void bn_finalize_kernel(const double2 *__restrict__ partial, BnGeo g,
                                                                 const float *gamma, const float *beta,
                                                                 float *running_mean, float *running_var,
                                                                 double momentum, double eps, float *save_mean,
                                                                 float *save_invstd, float *scale, float *shift)
{
    __shared__ double2 s_w[kBnThreads / 64];
    const int c = blockIdx.x;
    const double n = (double)g.Sg * g.L;
    // the groups in order: the running statistics see G successive updates, as G separate calls would give
    float rm = running_mean ? running_mean[c] : 0.0f, rv = running_var ? running_var[c] : 0.0f;
    for (int grp = 0; grp < g.G; ++grp) {
        double2 acc = make_double2(0.0, 0.0);
        const double2 *p = partial + (long long)c * g.nchunk + grp * g.ncg;
        for (long long i = threadIdx.x; i < g.ncg; i += kBnThreads) { acc.x += p[i].x; acc.y += p[i].y; }
        acc = block_sum2(acc, s_w);
        if (threadIdx.x == 0) {
            const double mean = acc.x / n;
            double var = acc.y / n - mean * mean;
            var = var > 0.0 ? var : 0.0;
            const float invstd = (float)(1.0 / sqrt(var + eps));
            const float m = (float)mean;
            const float sc = gamma[c] * invstd;
            const int o = grp * g.C + c;
            save_mean[o] = m;
            save_invstd[o] = invstd;
            scale[o] = sc;
            shift[o] = fmaf(-m, sc, beta[c]);
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            rm = (float)((1.0 - momentum) * rm + momentum * mean);
            rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
        }
        __syncthreads();      // s_w is reused by the next group
    }
    if (threadIdx.x == 0) {
        if (running_mean) running_mean[c] = rm;
        if (running_var) running_var[c] = rv;
    }
}

__device__ __forceinline__ float lrelu(float u, float slope) { return u > 0.0f ? u : u * slope; }

// POOL: 0 none, 1 max over pairs (max_pool1d(2)), 2 max over the whole row (the PointNet's max over points,
// src/model/box_regression.py:37-38; L a power of two >= 4, so a row is L / 4 neighbouring lanes of one wave)
template <int POOL>
__global__ __launch_bounds__(kBnThreads) void bn_apply_kernel(const float *__restrict__ y, BnGeo g,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, float slope,
                                                              float *__restrict__ out)
{
    const Lane l = lane_of(g);
    if (!l.active) return;
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = l.grp * g.C + (l.p0 + i) / g.L;
        sc[i] = scale[c]; sh[i] = shift[c];
    }
    const float *src = y + l.s0 * g.P + l.p0;
    const int po = POOL == 2 ? g.C : POOL == 1 ? g.P / 2 : g.P;
    float *dst = out + l.s0 * po + (POOL == 2 ? l.p0 / g.L : POOL == 1 ? l.p0 / 2 : l.p0);
    const int lpr = g.L >> 2;       // lanes of a row (POOL == 2)
    const bool row_head = (threadIdx.x & (lpr - 1)) == 0;
#pragma unroll kSampleUnroll<POOL>     // (a loop with cross-lane operations is not unrolled with a remainder)
    for (long long s = l.s0; s < l.s1; ++s, src += g.P, dst += po) {
        const float4 v = ld4(src);
        const float z0 = lrelu(fmaf(v.x, sc[0], sh[0]), slope), z1 = lrelu(fmaf(v.y, sc[1], sh[1]), slope);
        const float z2 = lrelu(fmaf(v.z, sc[2], sh[2]), slope), z3 = lrelu(fmaf(v.w, sc[3], sh[3]), slope);
        if (POOL == 2) {
            float m = fmaxf(fmaxf(z0, z1), fmaxf(z2, z3));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                if (o < lpr) m = fmaxf(m, __shfl_xor(m, o, 64));
            if (row_head) *dst = m;
        } else if (POOL == 1) st2(dst, z0 >= z1 ? z0 : z1, z2 >= z3 ? z2 : z3);
        else st4(dst, z0, z1, z2, z3);
    }
}

struct BwdConst { float sc[4], sh[4], mu[4], is[4]; };

// scale / shift are rebuilt from (gamma, beta, mean, 1/std) with the forward's own operations (bn_finalize_kernel),
// so that the recomputed u = y * scale + shift has the forward's bits: the pool winner and the activation sign
// depend on it
__device__ __forceinline__ BwdConst bwd_const(const BnGeo &g, int grp, int p0, const float *gamma, const float *beta,
                                              const float *mean, const float *invstd)
{
    BwdConst k;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (p0 + i) / g.L;
        k.mu[i] = mean[grp * g.C + c]; k.is[i] = invstd[grp * g.C + c];
        k.sc[i] = gamma[c] * k.is[i];
        k.sh[i] = fmaf(-k.mu[i], k.sc[i], beta[c]);
    }
    return k;
}

// gradient with respect to the BatchNorm output u, and xhat, for the lane's four positions of one sample
template <int POOL>
__device__ __forceinline__ void grad_u(const float *src, const float *gsrc, const BwdConst &k, float slope, int L,
                                       float (&du)[4], float (&xh)[4])
{
    const float4 v4 = ld4(src);
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float u[4], gz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[i] = fmaf(v[i], k.sc[i], k.sh[i]);
        xh[i] = (v[i] - k.mu[i]) * k.is[i];
    }
    if (POOL == 2) {
        // the row's FIRST maximum takes the gradient: (value, position) reduced over the row's lanes
        float z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = lrelu(u[i], slope);
        const int lpr = L >> 2, pos0 = 4 * (threadIdx.x & (lpr - 1));
        float bv = z[0];
        int bi = pos0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (z[i] > bv) { bv = z[i]; bi = pos0 + i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            if (o >= lpr) continue;
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        const float d = *gsrc;
#pragma unroll
        for (int i = 0; i < 4; ++i) gz[i] = pos0 + i == bi ? d : 0.0f;
    } else if (POOL == 1) {
        // the pair's winner takes the gradient; max_pool1d keeps the FIRST maximum on a tie
        const float2 d = ld2(gsrc);
        const bool f0 = lrelu(u[0], slope) >= lrelu(u[1], slope), f1 = lrelu(u[2], slope) >= lrelu(u[3], slope);
        gz[0] = f0 ? d.x : 0.0f; gz[1] = f0 ? 0.0f : d.x;
        gz[2] = f1 ? d.y : 0.0f; gz[3] = f1 ? 0.0f : d.y;
    } else {
        const float4 d = ld4(gsrc);
        gz[0] = d.x; gz[1] = d.y; gz[2] = d.z; gz[3] = d.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) du[i] = u[i] > 0.0f ? gz[i] : gz[i] * slope;
}

template <int POOL>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_reduce_kernel(const float *__restrict__ y,
                                                                   const float *__restrict__ dz, BnGeo g,
                                                                   const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ invstd, float slope,
                                                                   double2 *__restrict__ partial)
{
    __shared__ double s_a[kBnSlice], s_b[kBnSlice];
    const Lane l = lane_of(g);
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (l.active) {
        const BwdConst k = bwd_const(g, l.grp, l.p0, gamma, beta, mean, invstd);
        const int po = POOL == 2 ? g.C : POOL == 1 ? g.P / 2 : g.P;
        const float *src = y + l.s0 * g.P + l.p0;
        const float *gsrc = dz + l.s0 * po + (POOL == 2 ? l.p0 / g.L : POOL == 1 ? l.p0 / 2 : l.p0);
#pragma unroll kSampleUnroll<POOL>     // (a loop with cross-lane operations is not unrolled with a remainder)
        for (long long s = l.s0; s < l.s1; ++s, src += g.P, gsrc += po) {
            float du[4], xh[4];
            grad_u<POOL>(src, gsrc, k, slope, g.L, du, xh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += (double)du[i]; b[i] = fma((double)du[i], (double)xh[i], b[i]); }
        }
    }
    channel_reduce(g, a, b, s_a, s_b, partial);
}

__global__ __launch_bounds__(kBnThreads) void bn_bwd_final_kernel(const double2 *__restrict__ partial, BnGeo g,
                                                                  float *dgamma, float *dbeta, float *k1, float *k2)
{
    __shared__ double2 s_w[kBnThreads / 64];
    const int c = blockIdx.x;
    const double n = (double)g.Sg * g.L;
    double tot_a = 0.0, tot_b = 0.0;
    for (int grp = 0; grp < g.G; ++grp) {
        double2 acc = make_double2(0.0, 0.0);
        const double2 *p = partial + (long long)c * g.nchunk + grp * g.ncg;
        for (long long i = threadIdx.x; i < g.ncg; i += kBnThreads) { acc.x += p[i].x; acc.y += p[i].y; }
        acc = block_sum2(acc, s_w);
        if (threadIdx.x == 0) {
            k1[grp * g.C + c] = (float)(acc.x / n);
            k2[grp * g.C + c] = (float)(acc.y / n);
            tot_a += acc.x;
            tot_b += acc.y;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dbeta[c] = (float)tot_a;
        dgamma[c] = (float)tot_b;
    }
}

// DSUM: also the per-channel sum of dy (= the gradient of a bias added in front of the BatchNorm, i.e. the
// convolution's bias; zero in exact arithmetic, the framework reduces dy for it in a pass of its own)
template <int POOL, bool DSUM>
__global__ __launch_bounds__(kBnThreads) void bn_bwd_dgrad_kernel(const float *__restrict__ y,
                                                                  const float *__restrict__ dz, BnGeo g,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd,
                                                                  const float *__restrict__ k1,
                                                                  const float *__restrict__ k2, float slope,
                                                                  float *__restrict__ dy,
                                                                  double2 *__restrict__ partial)
{
    __shared__ double s_a[DSUM ? kBnSlice : 1], s_b[DSUM ? kBnSlice : 1];
    const Lane l = lane_of(g);
    double acc[4] = {0, 0, 0, 0};
    if (l.active) {
        const BwdConst k = bwd_const(g, l.grp, l.p0, gamma, beta, mean, invstd);
        float m1[4], m2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = l.grp * g.C + (l.p0 + i) / g.L;
            m1[i] = k1[c]; m2[i] = k2[c];
        }
        const int po = POOL == 2 ? g.C : POOL == 1 ? g.P / 2 : g.P;
        const float *src = y + l.s0 * g.P + l.p0;
        const float *gsrc = dz + l.s0 * po + (POOL == 2 ? l.p0 / g.L : POOL == 1 ? l.p0 / 2 : l.p0);
        float *dst = dy + l.s0 * g.P + l.p0;
        float part[4] = {0, 0, 0, 0};            // at most kStreamK terms each
#pragma unroll kSampleUnroll<POOL>     // (a loop with cross-lane operations is not unrolled with a remainder)
        for (long long s = l.s0; s < l.s1; ++s, src += g.P, gsrc += po, dst += g.P) {
            float du[4], xh[4], r[4];
            grad_u<POOL>(src, gsrc, k, slope, g.L, du, xh);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = k.sc[i] * ((du[i] - m1[i]) - xh[i] * m2[i]);
                if (DSUM) part[i] += r[i];
            }
            st4(dst, r[0], r[1], r[2], r[3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (double)part[i];
    }
    if (DSUM) {
        const double zero[4] = {0, 0, 0, 0};
        channel_reduce(g, acc, zero, s_a, s_b, partial);
    }
}

__global__ __launch_bounds__(kBnThreads) void bn_sum_final_kernel(const double2 *__restrict__ partial,
                                                                  long long nchunk, float *out)
{
    __shared__ double2 s_w[kBnThreads / 64];
    const int c = blockIdx.x;
    double2 acc = make_double2(0.0, 0.0);
    for (long long i = threadIdx.x; i < nchunk; i += kBnThreads) acc.x += partial[c * nchunk + i].x;
    acc = block_sum2(acc, s_w);
    if (threadIdx.x == 0) out[c] = (float)acc.x;
}

// workspace: double2 partial[C * nchunk_max] | float coef[4 * G * C]
constexpr int kStatsWgs = 4096, kStatsK = 64;      // reductions: few partial sums
constexpr int kStreamWgs = 8192, kStreamK = 16;  // element-wise passes: short loops, many workgroups

size_t partial_bytes(const BnGeo &g) { return (size_t)g.C * (size_t)g.nchunk * sizeof(double2); }

// the reductions (gs) and the bias sum of the dgrad pass (ga) use the partial-sum area one after the other
size_t workspace_need(const BnGeo &gs, const BnGeo &ga)
{
    const size_t p = partial_bytes(gs) > partial_bytes(ga) ? partial_bytes(gs) : partial_bytes(ga);
    return p + 4 * (size_t)gs.G * gs.C * sizeof(float);
}

// pool: 0 none | 1 pairs (even L) | 2 whole row (L a power of two in 4 .. 256)
bool pool_ok(int pool, int L)
{
    if (pool == 0) return true;
    if (pool == 1) return (L & 1) == 0;
    return pool == 2 && L >= 4 && (L & (L - 1)) == 0;
}

}  // namespace

extern "C" size_t pof_bn_lrelu_pool_workspace_bytes(long long S, int C, int L, int groups)
{
    BnGeo gs, ga;
    if (!make_geo(S, C, L, groups, kStatsWgs, kStatsK, &gs) || !make_geo(S, C, L, groups, kStreamWgs, kStreamK, &ga))
        return 0;
    return workspace_need(gs, ga);
}

extern "C" int pof_bn_lrelu_pool_forward(const float *y, long long S, int C, int L, int groups, const float *gamma,
                                         const float *beta, float *running_mean, float *running_var, double momentum,
                                         double eps, double negative_slope, int pool, float *out, float *save_mean,
                                         float *save_invstd, void *workspace, size_t workspace_bytes,
                                         pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!y || !gamma || !beta || !out || !save_mean || !save_invstd || !workspace) return POF_E_BADARG;
    BnGeo gs, ga;
    if (!make_geo(S, C, L, groups, kStatsWgs, kStatsK, &gs) || !make_geo(S, C, L, groups, kStreamWgs, kStreamK, &ga))
        return POF_E_SHAPE;
    if (!pool_ok(pool, L)) return POF_E_SHAPE;
    if (!(eps >= 0.0)) return POF_E_BADARG;
    if (workspace_bytes < workspace_need(gs, ga)) return POF_E_WORKSPACE;
    double2 *partial = static_cast<double2 *>(workspace);
    float *coef = reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_need(gs, ga)) - 4 * groups * C;
    float *scale = coef, *shift = coef + groups * C;
    hipStream_t st = pof_stream(stream);
    bn_stats_kernel<<<dim3((unsigned)gs.nchunk, gs.nslice), kBnThreads, 0, st>>>(y, gs, partial);
    POF_CHECK_LAUNCH();
    bn_finalize_kernel<<<C, kBnThreads, 0, st>>>(partial, gs, gamma, beta, running_mean, running_var, momentum, eps,
                                                 save_mean, save_invstd, scale, shift);
    POF_CHECK_LAUNCH();
    const dim3 grid((unsigned)ga.nchunk, ga.nslice);
    if (pool == 2) bn_apply_kernel<2><<<grid, kBnThreads, 0, st>>>(y, ga, scale, shift, (float)negative_slope, out);
    else if (pool) bn_apply_kernel<1><<<grid, kBnThreads, 0, st>>>(y, ga, scale, shift, (float)negative_slope, out);
    else bn_apply_kernel<0><<<grid, kBnThreads, 0, st>>>(y, ga, scale, shift, (float)negative_slope, out);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_bn_lrelu_pool_backward(const float *y, const float *dz, long long S, int C, int L, int groups,
                                          const float *gamma, const float *beta, const float *save_mean,
                                          const float *save_invstd, double negative_slope, int pool, float *dy,
                                          float *dgamma, float *dbeta, float *dbias_in, void *workspace,
                                          size_t workspace_bytes, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!y || !dz || !gamma || !beta || !save_mean || !save_invstd || !dy || !dgamma || !dbeta || !workspace)
        return POF_E_BADARG;
    BnGeo gs, ga;
    if (!make_geo(S, C, L, groups, kStatsWgs, kStatsK, &gs) || !make_geo(S, C, L, groups, kStreamWgs, kStreamK, &ga))
        return POF_E_SHAPE;
    if (!pool_ok(pool, L)) return POF_E_SHAPE;
    if (workspace_bytes < workspace_need(gs, ga)) return POF_E_WORKSPACE;
    double2 *partial = static_cast<double2 *>(workspace);
    float *coef = reinterpret_cast<float *>(static_cast<char *>(workspace) + workspace_need(gs, ga)) - 4 * groups * C;
    float *k1 = coef + 2 * groups * C, *k2 = coef + 3 * groups * C;
    const float slope = (float)negative_slope;
    hipStream_t st = pof_stream(stream);
    const dim3 rgrid((unsigned)gs.nchunk, gs.nslice), dgrid((unsigned)ga.nchunk, ga.nslice);
#define POF_REDUCE(P_) bn_bwd_reduce_kernel<P_><<<rgrid, kBnThreads, 0, st>>>(y, dz, gs, gamma, beta, save_mean, save_invstd, slope, partial)
    if (pool == 2) POF_REDUCE(2); else if (pool) POF_REDUCE(1); else POF_REDUCE(0);
#undef POF_REDUCE
    POF_CHECK_LAUNCH();
    bn_bwd_final_kernel<<<C, kBnThreads, 0, st>>>(partial, gs, dgamma, dbeta, k1, k2);
    POF_CHECK_LAUNCH();
#define POF_DGRAD(P_, D_) bn_bwd_dgrad_kernel<P_, D_><<<dgrid, kBnThreads, 0, st>>>( \
        y, dz, ga, gamma, beta, save_mean, save_invstd, k1, k2, slope, dy, partial)
    if (dbias_in) { if (pool == 2) POF_DGRAD(2, true); else if (pool) POF_DGRAD(1, true); else POF_DGRAD(0, true); }
    else { if (pool == 2) POF_DGRAD(2, false); else if (pool) POF_DGRAD(1, false); else POF_DGRAD(0, false); }
#undef POF_DGRAD
    POF_CHECK_LAUNCH();
    if (dbias_in) {
        bn_sum_final_kernel<<<C, kBnThreads, 0, st>>>(partial, ga.nchunk, dbias_in);
        POF_CHECK_LAUNCH();
    }
    return POF_OK;
}
