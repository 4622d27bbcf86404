// N3: BoxRegressor input preparation for MANY detections per launch.
//
// The reference prepares one detection per call on the host (box_regressor.py:43-75):
//   seg = points[norm(points - centre) <= radius]                 (:94-105, float64)
//   len(seg) < min_segment_size -> skipped
//   len(seg) > input_size: shuffle, keep the first input_size    (uniform random subset)
//   else: shuffle, np.repeat(seg, input_size // n), append the first input_size % n rows
//         of the repeated array, shuffle again
//   x = hstack(seg - centre, det_ori) -> float32 [input_size][D + 1]
// Here one workgroup prepares one detection: radius query over all points -> candidate list
// in LDS -> bitonic sort by a counter-based hash of (seed, detection, point index) (a uniform
// random order that does not depend on scheduling) -> the same multiset of rows.  The rows come
// out in hash order without the second shuffle: the consumer (PointNet: shared MLP + max over
// points) is invariant to row order, and the reference's own order depends on the global NumPy
// RNG state, so parity is on the multiset.  The same path serves anns_to_segments
// (src/data_handle/jrdb_handle.py:178-256: radius query around perturbed box centres).
#include "pof_common.h"

namespace {

constexpr int kSegThreads = 256;
constexpr int kSegCap = 4096;   // candidates sorted in LDS; larger segments are pre-thinned by hash

__device__ __forceinline__ uint32_t mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t point_hash(uint32_t seed, uint32_t det, uint32_t idx)
{
    return mix32(idx * 0x9e3779b9u + mix32(det * 0x7f4a7c15u + mix32(seed)));
}

struct SegArgsIn {
    const double *points;   // [Np][D]
    const double *centers;  // [S][D]
    const double *oris;     // [S]
    int Np, D, M, min_size;
    double radius;
    uint32_t seed;
    float *x;               // [S][M][D+1]
    int32_t *count;         // [S]
    uint8_t *mask;          // optional [S][Np]: 1 where the point is inside the detection's disc
};

__global__ __launch_bounds__(kSegThreads) void segment_inputs_kernel(SegArgsIn a)
{
    __shared__ unsigned long long s_key[kSegCap];   // (hash << 32) | point index
    __shared__ int s_n, s_m;
    const int det = blockIdx.x, tid = threadIdx.x;
    const int D = a.D;
    double c[3] = {0.0, 0.0, 0.0};
    for (int d = 0; d < D; ++d) c[d] = a.centers[(long long)det * D + d];
    if (tid == 0) {
        s_n = 0;
        s_m = 0;
    }
    __syncthreads();
    auto inside = [&](int i) {
        double s2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double e = a.points[(long long)i * D + d] - c[d];
            s2 += e * e;
        }
        return sqrt(s2) <= a.radius;      // np.linalg.norm: sqrt of the float64 sum of squares
    };
    // pass 1: count; segments that fit are listed right away
    for (int i0 = 0; i0 < a.Np; i0 += kSegThreads) {
        const int i = i0 + tid;
        const bool in = i < a.Np && inside(i);
        if (a.mask && i < a.Np) a.mask[(long long)det * a.Np + i] = in ? 1 : 0;
        if (in) {
            const int pos = atomicAdd(&s_n, 1);
            if (pos < kSegCap) s_key[pos] = ((unsigned long long)point_hash(a.seed, det, i) << 32) | (uint32_t)i;
        }
    }
    __syncthreads();
    const int n = s_n;
    int m = min(n, kSegCap);
    if (n > kSegCap) {
        // pass 2: only a random subset of size M <= 1024 is needed -> keep the ~3072 smallest hashes
        // (binomial spread ~55: never fewer than M, never more than the list holds)
        const uint32_t thr = (uint32_t)(4294967296.0 * (3072.0 / (double)n));
        for (int i0 = 0; i0 < a.Np; i0 += kSegThreads) {
            const int i = i0 + tid;
            if (i < a.Np && inside(i)) {
                const uint32_t h = point_hash(a.seed, det, i);
                if (h < thr) {
                    const int pos = atomicAdd(&s_m, 1);
                    if (pos < kSegCap) s_key[pos] = ((unsigned long long)h << 32) | (uint32_t)i;
                }
            }
        }
        __syncthreads();
        m = min(s_m, kSegCap);
    }
    if (tid == 0) a.count[det] = n;
    float *x = a.x + (long long)det * a.M * (D + 1);
    if (n < a.min_size || n == 0) {
        for (int e = tid; e < a.M * (D + 1); e += kSegThreads) x[e] = 0.0f;
        return;
    }
    // bitonic sort of the m keys (ascending), padded with the maximum key
    int mp = 1;
    while (mp < m) mp <<= 1;
    for (int e = m + tid; e < mp; e += kSegThreads) s_key[e] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= mp; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < mp; i += kSegThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ki = s_key[i], kl = s_key[l];
                    const bool up = (i & k) == 0;
                    if (up ? (ki > kl) : (ki < kl)) {
                        s_key[i] = kl;
                        s_key[l] = ki;
                    }
                }
            }
            __syncthreads();
        }
    }
    // rows: random subset when the segment is larger than M, else repeat + pad
    const int rep = n > a.M ? 1 : a.M / n;
    const double ori = a.oris[det];
    for (int j = tid; j < a.M; j += kSegThreads) {
        int src;
        if (n > a.M) src = j;
        else src = (j < n * rep) ? j / rep : (j - n * rep) / rep;   // rows of np.repeat(seg, rep), then its first rows again
        src = min(src, m - 1);                                      // never past the sorted candidates
        const int pi = (int)(uint32_t)s_key[src];
        for (int d = 0; d < D; ++d) x[j * (D + 1) + d] = (float)(a.points[(long long)pi * D + d] - c[d]);
        x[j * (D + 1) + D] = (float)ori;
    }
}

// Training-side twin (src/data_handle/jrdb_dataset.py:99-156): the segments are already cut
// (CSR over a point pool), every sample subtracts its detection centre, optionally appends one
// per-sample value as an extra column (the reference's random input angle), optionally drops
// a random fraction of its points first (`shuffle; input[int(n * drop):]`), then the same
// fixed-size resampling.  One workgroup per sample, same hash order as above.
struct ResampleArgs {
    const double *points;    // [P][D]
    const int32_t *seg_off;  // [S+1]
    const double *centers;   // [S][D]
    const double *extra;     // [S] or null
    int D, M;
    double drop;
    uint32_t seed;
    float *x;                // [S][M][D + (extra ? 1 : 0)]
    int32_t *count;          // [S] points kept after the random drop
};

__global__ __launch_bounds__(kSegThreads) void segment_resample_kernel(ResampleArgs a)
{
    __shared__ unsigned long long s_key[kSegCap];
    const int smp = blockIdx.x, tid = threadIdx.x;
    const int D = a.D, W = D + (a.extra ? 1 : 0);
    const int p0 = a.seg_off[smp], n_all = a.seg_off[smp + 1] - p0;
    float *x = a.x + (long long)smp * a.M * W;
    const int dropped = (int)((double)n_all * a.drop);          // int(len(input) * random_drop)
    const int n = n_all - dropped;
    if (tid == 0) a.count[smp] = n;
    if (n <= 0 || n_all > kSegCap) {                            // (the launcher rejects oversized segments)
        for (int e = tid; e < a.M * W; e += kSegThreads) x[e] = 0.0f;
        return;
    }
    int mp = 1;
    while (mp < n_all) mp <<= 1;
    for (int e = tid; e < mp; e += kSegThreads)
        s_key[e] = e < n_all ? (((unsigned long long)point_hash(a.seed, smp, e) << 32) | (uint32_t)e) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= mp; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < mp; i += kSegThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ki = s_key[i], kl = s_key[l];
                    const bool up = (i & k) == 0;
                    if (up ? (ki > kl) : (ki < kl)) {
                        s_key[i] = kl;
                        s_key[l] = ki;
                    }
                }
            }
            __syncthreads();
        }
    }
    // hash order = the shuffle; its first `dropped` entries are the dropped points
    const int rep = n > a.M ? 1 : a.M / n;
    const double ex = a.extra ? a.extra[smp] : 0.0;
    for (int j = tid; j < a.M; j += kSegThreads) {
        const int src = n > a.M ? j : ((j < n * rep) ? j / rep : (j - n * rep) / rep);
        const int pi = p0 + (int)(uint32_t)s_key[dropped + src];
        for (int d = 0; d < D; ++d)
            x[j * W + d] = (float)(a.points[(long long)pi * D + d] - a.centers[(long long)smp * D + d]);
        if (a.extra) x[j * W + D] = (float)ex;
    }
}

}  // namespace

extern "C" int pof_segment_inputs(const double *points, int Np, int D, const double *centers, const double *oris,
                                  int S, double radius, int input_size, int min_segment_size, uint32_t seed,
                                  float *x, int32_t *count, uint8_t *mask, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!points || !centers || !oris || !x || !count) return POF_E_BADARG;
    if (Np < 0 || S < 0 || input_size < 1 || !(radius >= 0.0)) return POF_E_BADARG;
    if (D < 2 || D > 3) return POF_E_SHAPE;
    if (input_size > 1024) return POF_E_SHAPE;      // the thinning pass keeps ~3072 candidates
    if (S == 0) return POF_OK;
    SegArgsIn a;
    a.points = points; a.centers = centers; a.oris = oris; a.Np = Np; a.D = D; a.M = input_size;
    a.min_size = min_segment_size; a.radius = radius; a.seed = seed; a.x = x; a.count = count; a.mask = mask;
    segment_inputs_kernel<<<S, kSegThreads, 0, pof_stream(stream)>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_segment_resample(const double *points, int D, const int32_t *seg_offsets, int S, int max_segment,
                                    const double *centers, const double *extra, double random_drop, int input_size,
                                    uint32_t seed, float *x, int32_t *count, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!points || !seg_offsets || !centers || !x || !count) return POF_E_BADARG;
    if (S < 0 || input_size < 1 || !(random_drop >= 0.0) || !(random_drop < 1.0)) return POF_E_BADARG;
    if (D < 2 || D > 3) return POF_E_SHAPE;
    if (max_segment > kSegCap) return POF_E_SHAPE;      // the caller states the longest segment
    if (S == 0) return POF_OK;
    ResampleArgs a;
    a.points = points; a.seg_off = seg_offsets; a.centers = centers; a.extra = extra; a.D = D; a.M = input_size;
    a.drop = random_drop; a.seed = seed; a.x = x; a.count = count;
    segment_resample_kernel<<<S, kSegThreads, 0, pof_stream(stream)>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
