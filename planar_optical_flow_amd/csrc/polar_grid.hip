// N4: scans_to_polar_grid (src/utils/utils.py:492-531), batched.
//
// scans [B][T][N] float32 -> grid [B][T][R][N] float32, R = int((max-min)/bin) + 1.
// Per (t, i): g = int((clip(r) - min) / bin) (float32 arithmetic, truncation);
//   grid[.., q, i] = clip((q - g) * bin, -c, c)   (0 when c <= 0)      q != g
//                  = clip(r)                                           q == g
// and with `normalize` the TSDF is scaled by / mag * 2 and the hit cell holds
// (clip(r) - mid) / mag * 2 (float32 throughout, as NumPy >= 2 evaluates the reference).
//
// Pure write stream: T*N*4 bytes in, T*R*N*4 bytes out per sample (R = 31 by default).
// One lane owns 4 consecutive points of one scan row and writes them for every range
// bin with 16-byte streaming stores (each wave store = 1 KB contiguous).
#include "pof_common.h"

namespace {

constexpr int kPolarThreads = 128;   // 450 points = 113 lanes of 4

struct PolarArgs {
    const float *scans;
    float *out;
    long long rows;     // B * T
    int N, R;
    float minr, maxr, bin, clipv, mag, mid;
    int normalize, use_tsdf;
};

__device__ __forceinline__ float cell(const PolarArgs &a, int q, int g, float val)
{
    float t = 0.0f;
    if (a.use_tsdf) {
        t = (float)(q - g) * a.bin;
        t = fminf(fmaxf(t, -a.clipv), a.clipv);
    }
    if (a.normalize) t = t / a.mag * 2.0f;
    return q == g ? val : t;
}

template <int V>
__global__ __launch_bounds__(kPolarThreads) void polar_grid_kernel(PolarArgs a)
{
    const long long row = blockIdx.y;
    const int i0 = (blockIdx.x * kPolarThreads + threadIdx.x) * V;
    if (i0 >= a.N) return;
    const float *src = a.scans + row * a.N + i0;
    float val[V];
    int g[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        // np.clip(NaN) stays NaN; fmin/fmax would drop it -> explicit compares
        float r = src[v];
        r = r < a.minr ? a.minr : r;
        r = r > a.maxr ? a.maxr : r;
        g[v] = (int)((r - a.minr) / a.bin);
        val[v] = a.normalize ? (r - a.mid) / a.mag * 2.0f : r;
    }
    float *dst = a.out + (row * a.R) * a.N + i0;
    for (int q = 0; q < a.R; ++q) {
        if (V == 4) {
            using F4V = float __attribute__((ext_vector_type(4)));
            const F4V o = {cell(a, q, g[0], val[0]), cell(a, q, g[1], val[1]), cell(a, q, g[2], val[2]),
                           cell(a, q, g[3], val[3])};
            __builtin_nontemporal_store(o, reinterpret_cast<F4V *>(dst + (long long)q * a.N));
        } else {
            dst[(long long)q * a.N] = cell(a, q, g[0], val[0]);
        }
    }
}

}  // namespace

extern "C" int pof_polar_grid(const float *scans, int B, int T, int N, double min_range, double max_range,
                              double range_bin_size, double tsdf_clip, int normalize, float *out,
                              pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!scans || !out || B < 0 || T < 1 || N < 1) return POF_E_BADARG;
    if (!(range_bin_size > 0.0) || !(max_range >= min_range)) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    PolarArgs a;
    a.scans = scans; a.out = out; a.rows = (long long)B * T; a.N = N;
    a.R = (int)((max_range - min_range) / range_bin_size) + 1;
    a.minr = (float)min_range; a.maxr = (float)max_range; a.bin = (float)range_bin_size;
    a.clipv = (float)tsdf_clip; a.use_tsdf = tsdf_clip > 0.0;
    a.mag = (float)(max_range - min_range); a.mid = (float)(0.5 * (max_range - min_range));
    a.normalize = normalize;
    if (a.rows > 65535LL * 32768LL) return POF_E_SHAPE;
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(scans) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    hipStream_t s = pof_stream(stream);
    // grid.y <= 65535: rows beyond that go in further launches
    for (long long r0 = 0; r0 < a.rows; r0 += 65535) {
        PolarArgs c = a;
        const long long nr = a.rows - r0 < 65535 ? a.rows - r0 : 65535;
        c.scans = scans + r0 * N;
        c.out = out + r0 * a.R * N;
        if (vec) polar_grid_kernel<4><<<dim3((N / 4 + kPolarThreads - 1) / kPolarThreads, (unsigned)nr), kPolarThreads, 0, s>>>(c);
        else polar_grid_kernel<1><<<dim3((N + kPolarThreads - 1) / kPolarThreads, (unsigned)nr), kPolarThreads, 0, s>>>(c);
        POF_CHECK_LAUNCH();
    }
    return POF_OK;
}
