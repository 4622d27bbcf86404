// A9: Prototype._fusion (src/depracted/model/prototype.py:118-156), banded.
//
// feat1, feat2 [B][C][n] float32 -> out [B][D][n], D = 2*max_disp+1:
//   out[b,d,i] = sum_c sum_k f1[b,c,clamp(i+k-hk)] * f2[b,c,clamp(clamp(i+d-max_disp)+k-hk)]
// The reference forms the full n x n correlation with a GEMM and gathers the
// band; here only the band is computed (O(n*D) instead of O(n^2)), which is
// what keeps BASELINE config 5 (n = 450) HBM-bound: algorithmic bytes per
// sample 2*C*n*4 read + D*n*4 written (119 244 B at C=256, n=57), 8 flop/B.
//
// One workgroup = one sample x 64 consecutive points; 4 wave64s split the
// channels, each lane owns one point and keeps its D accumulators in registers.
// Channel chunks are staged through LDS ([c][position], lanes read consecutive
// addresses -> conflict free).  float32 multiply-add, like torch.matmul.
#include "pof_common.h"

namespace {

constexpr int kTile = 64;    // points per workgroup (= lanes per wave)
constexpr int kGroups = 4;   // channel groups (= waves)
constexpr int kChunk = 32;   // channels staged per pass
constexpr int kMaxD = 15;    // max 2*max_disp+1
constexpr int kMaxK = 5;     // max kernel_size

template <int K, int D>
__global__ __launch_bounds__(kTile *kGroups) void band_corr_kernel(const float *f1, const float *f2,
                                                                   float *out, int C, int n)
{
    constexpr int HK = K / 2, MD = D / 2;
    constexpr int W1 = kTile + 2 * HK;            // staged f1 positions
    constexpr int W2 = kTile + 2 * (MD + HK);     // staged f2 positions
    __shared__ float s1[kChunk][W1];
    __shared__ float s2[kChunk][W2];
    __shared__ float s_red[kGroups][D][kTile];

    const int b = blockIdx.y;
    const int i0 = blockIdx.x * kTile;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = i0 + lane;
    const float *g1 = f1 + (long long)b * C * n;
    const float *g2 = f2 + (long long)b * C * n;
    const bool interior = (i - MD >= 0) && (i + MD <= n - 1);

    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.0f;

    for (int c0 = 0; c0 < C; c0 += kChunk) {
        const int cc = min(kChunk, C - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < cc * W1; e += kTile * kGroups) {
            const int c = e / W1, x = e - c * W1;
            const int src = min(max(i0 - HK + x, 0), n - 1);
            s1[c][x] = g1[(long long)(c0 + c) * n + src];
        }
        for (int e = threadIdx.x; e < cc * W2; e += kTile * kGroups) {
            const int c = e / W2, y = e - c * W2;
            const int src = min(max(i0 - MD - HK + y, 0), n - 1);
            s2[c][y] = g2[(long long)(c0 + c) * n + src];
        }
        __syncthreads();
        if (i < n) {
            for (int c = grp; c < cc; c += kGroups) {
                float p1[K];
#pragma unroll
                for (int k = 0; k < K; ++k) p1[k] = s1[c][lane + k];
                if (interior) {
                    float w[D + K - 1];
#pragma unroll
                    for (int y = 0; y < D + K - 1; ++y) w[y] = s2[c][lane + y];
#pragma unroll
                    for (int d = 0; d < D; ++d)
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[d] = fmaf(p1[k], w[d + k], acc[d]);
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const int j = min(max(i + d - MD, 0), n - 1);
                        const int base = j - (i0 - MD - HK) - HK;  // staged position of j-HK
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[d] = fmaf(p1[k], s2[c][base + k], acc[d]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) s_red[grp][d][lane] = acc[d];
    __syncthreads();
    for (int e = threadIdx.x; e < D * kTile; e += kTile * kGroups) {
        const int d = e / kTile, l = e - d * kTile;
        if (i0 + l < n) {
            float v = s_red[0][d][l];
#pragma unroll
            for (int g = 1; g < kGroups; ++g) v += s_red[g][d][l];
            out[((long long)b * D + d) * n + i0 + l] = v;
        }
    }
}

// Backward: scatter form.  Each lane owns (channel, point i) and adds its
// contributions to d_f1[c, clamp(i+k-hk)] and d_f2[c, clamp(j_d+k-hk)] in LDS
// (ds_add_f32), the tile is then written out.  One workgroup = one sample x
// kBwdCh channels x all n points (n <= kBwdMaxN).
constexpr int kBwdCh = 8;
constexpr int kBwdMaxN = 512;

__global__ __launch_bounds__(256) void band_corr_bwd_kernel(const float *f1, const float *f2,
                                                            const float *g_out, float *d_f1, float *d_f2,
                                                            int C, int n, int K, int D)
{
    extern __shared__ float smem_f[];
    float *s_g = smem_f;                 // [D][n]
    float *s_d1 = s_g + D * n;           // [kBwdCh][n]
    float *s_d2 = s_d1 + kBwdCh * n;     // [kBwdCh][n]
    const int b = blockIdx.y, c0 = blockIdx.x * kBwdCh;
    const int cc = min(kBwdCh, C - c0);
    const int hk = K / 2, md = D / 2;
    for (int e = threadIdx.x; e < D * n; e += blockDim.x) s_g[e] = g_out[(long long)b * D * n + e];
    for (int e = threadIdx.x; e < 2 * kBwdCh * n; e += blockDim.x) s_d1[e] = 0.0f;
    __syncthreads();
    const float *g1 = f1 + ((long long)b * C + c0) * n;
    const float *g2 = f2 + ((long long)b * C + c0) * n;
    for (int e = threadIdx.x; e < cc * n; e += blockDim.x) {
        const int c = e / n, i = e - c * n;
        const float *r1 = g1 + (long long)c * n, *r2 = g2 + (long long)c * n;
        for (int k = 0; k < K; ++k) {
            const int a1 = min(max(i + k - hk, 0), n - 1);
            const float v1 = r1[a1];
            float acc1 = 0.0f;
            for (int d = 0; d < D; ++d) {
                const int j = min(max(i + d - md, 0), n - 1);
                const int a2 = min(max(j + k - hk, 0), n - 1);
                const float gv = s_g[d * n + i];
                acc1 = fmaf(gv, r2[a2], acc1);
                atomicAdd(&s_d2[c * n + a2], gv * v1);
            }
            atomicAdd(&s_d1[c * n + a1], acc1);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cc * n; e += blockDim.x) {
        d_f1[((long long)b * C + c0) * n + e] = s_d1[e];
        d_f2[((long long)b * C + c0) * n + e] = s_d2[e];
    }
}

template <int K>
int launch_k(const float *f1, const float *f2, float *out, int B, int C, int n, int D, hipStream_t s)
{
    dim3 grid((n + kTile - 1) / kTile, B), block(kTile * kGroups);
    switch (D) {
        case 1: band_corr_kernel<K, 1><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 3: band_corr_kernel<K, 3><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 5: band_corr_kernel<K, 5><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 7: band_corr_kernel<K, 7><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 9: band_corr_kernel<K, 9><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 11: band_corr_kernel<K, 11><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 13: band_corr_kernel<K, 13><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        case 15: band_corr_kernel<K, 15><<<grid, block, 0, s>>>(f1, f2, out, C, n); break;
        default: return POF_E_SHAPE;
    }
    return POF_OK;
}

}  // namespace

extern "C" int pof_band_correlation(const float *feat1, const float *feat2, float *out, int B, int C,
                                    int n, int kernel_size, int max_disp, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!feat1 || !feat2 || !out || B < 0 || C < 1 || n < 1) return POF_E_BADARG;
    if (kernel_size < 1 || kernel_size > kMaxK || (kernel_size & 1) == 0) return POF_E_SHAPE;
    if (max_disp < 0 || 2 * max_disp + 1 > kMaxD) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    const int D = 2 * max_disp + 1;
    hipStream_t s = pof_stream(stream);
    int rc;
    switch (kernel_size) {
        case 1: rc = launch_k<1>(feat1, feat2, out, B, C, n, D, s); break;
        case 3: rc = launch_k<3>(feat1, feat2, out, B, C, n, D, s); break;
        default: rc = launch_k<5>(feat1, feat2, out, B, C, n, D, s); break;
    }
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_band_correlation_backward(const float *feat1, const float *feat2, const float *g_out,
                                             float *d_feat1, float *d_feat2, int B, int C, int n,
                                             int kernel_size, int max_disp, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!feat1 || !feat2 || !g_out || !d_feat1 || !d_feat2 || B < 0 || C < 1 || n < 1) return POF_E_BADARG;
    if (kernel_size < 1 || kernel_size > kMaxK || (kernel_size & 1) == 0) return POF_E_SHAPE;
    if (max_disp < 0 || 2 * max_disp + 1 > kMaxD) return POF_E_SHAPE;
    if (n > kBwdMaxN) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    const int D = 2 * max_disp + 1;
    const size_t lds = (size_t)(D + 2 * kBwdCh) * n * sizeof(float);
    band_corr_bwd_kernel<<<dim3((C + kBwdCh - 1) / kBwdCh, B), 256, lds, pof_stream(stream)>>>(
        feat1, feat2, g_out, d_feat1, d_feat2, C, n, kernel_size, D);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
