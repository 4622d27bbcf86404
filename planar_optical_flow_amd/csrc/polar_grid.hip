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
// polar_grid_flat_kernel (round 2): one workgroup per scan row.  The row's output block [R][N] is one contiguous
// run of R*N floats; it is written as a FLAT stream of 16-byte, 16-byte-aligned non-temporal stores (a wave store =
// 1 KB of whole cache lines) whatever N is -- the first version needed N % 4 == 0 for its 16-byte stores and
// fell back to 4-byte stores at N = 450 (3.4 TB/s; aligned one-shot stores reach 6-6.9 TB/s on this part,
// profiles/r2_store_ceiling.txt).  Per-point values (bin index g, hit value) and the 2R+1 possible TSDF values
// clip((q - g) * bin) [/ mag * 2] are staged in LDS once per row, so an output element costs two LDS reads and
// a select; (q, i) advance incrementally (no division per element).
// polar_grid_kernel (rows that do not fit LDS): one lane owns 4 consecutive points of one scan row and writes
// them for every range bin.
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

__global__ __launch_bounds__(256) void polar_grid_flat_kernel(PolarArgs a)
{
    extern __shared__ __align__(16) unsigned char psm[];
    int *s_g = reinterpret_cast<int *>(psm);
    float *s_val = reinterpret_cast<float *>(s_g + a.N);
    float *s_t = s_val + a.N;                      // [2R + 1]: TSDF value of q - g = d at index d + R
    const long long row = blockIdx.x;
    const int N = a.N, R = a.R, tid = threadIdx.x;
    const float *src = a.scans + row * N;
    for (int i = tid; i < N; i += 256) {
        // np.clip(NaN) stays NaN; fmin/fmax would drop it -> explicit compares
        float r = src[i];
        r = r < a.minr ? a.minr : r;
        r = r > a.maxr ? a.maxr : r;
        s_g[i] = (int)((r - a.minr) / a.bin);
        s_val[i] = a.normalize ? (r - a.mid) / a.mag * 2.0f : r;
    }
    for (int d = tid; d < 2 * R + 1; d += 256) s_t[d] = cell(a, d - R, 0, 0.0f - 1.0f);   // q - g = d - R != 0 except d = R (unused)
    __syncthreads();
    const long long G0 = row * R * N;              // global element index of the block
    float *blk = a.out + G0;
    const int total = R * N;
    // elements in front of the first 16-byte boundary (the block starts on a 4-byte boundary)
    const int head = (int)(((16 - (reinterpret_cast<uintptr_t>(blk) & 15)) & 15) >> 2);
    auto value = [&](int q, int i) -> float {
        const int g = s_g[i];
        return q == g ? s_val[i] : s_t[q - g + R];
    };
    if (tid < head && tid < total) blk[tid] = value(tid / N, tid % N);
    const int nbody = total > head ? (total - head) >> 2 : 0;
    const int tail0 = head + 4 * nbody;
    if (tid < total - tail0) {
        const int e = tail0 + tid;
        blk[e] = value(e / N, e % N);
    }
    // 16-byte groups: group k = elements head + 4k ... ; lane strides by 256 groups = 1024 elements
    int e = head + 4 * tid;
    int q = e / N, i = e - q * N;
    const int dq = 1024 / N, di = 1024 - dq * N;
    using F4V = float __attribute__((ext_vector_type(4)));
    for (int k = tid; k < nbody; k += 256) {
        float v[4];
        int qq = q, ii = i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = value(qq, ii);
            if (++ii == N) {
                ii = 0;
                ++qq;
            }
        }
        const F4V o = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<F4V *>(blk + e));
        e += 1024;
        q += dq;
        i += di;
        if (i >= N) {
            i -= N;
            ++q;
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
    const size_t lds = (size_t)N * 8 + (size_t)(2 * a.R + 1) * 4;
    if (lds <= 60 * 1024 && (long long)a.R * N < (1LL << 30) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
        // flat form: one workgroup per scan row, aligned 16-byte stores for any N
        for (long long r0 = 0; r0 < a.rows; r0 += 0x7fffffffLL) {
            PolarArgs c = a;
            const long long nr = a.rows - r0 < 0x7fffffffLL ? a.rows - r0 : 0x7fffffffLL;
            c.scans = scans + r0 * N;
            c.out = out + r0 * a.R * N;
            polar_grid_flat_kernel<<<(unsigned)nr, 256, lds, s>>>(c);
            POF_CHECK_LAUNCH();
        }
        return POF_OK;
    }
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
