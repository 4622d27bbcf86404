// A8: scans_to_cutout (src/utils/utils.py:259-334) for a whole batch.
//
// scans [B][T][N] float32  ->  out [B][Ns][T][P] float32   (Ns = ceil(N/stride))
//
// Roofline: HBM, dominated by the output: (T*N*4 read + Ns*T*P*4 written) per
// sample = 513 000 B at T=5, N=450, P=56.  The float64 index math (angle ->
// fractional index -> floor/lerp) must round exactly like NumPy, so it stays in
// float64 with contraction off; the gathers are served from LDS.
//
// Two launches per call:
//   1. cutout_area_kernel  (area_mode only) per-sample max window width ->
//      s_area[b] = ceil(max/P) if any window covers more than P raw points, else 0
//      (the reference takes this max over the whole (T,N) call, utils.py:304-308);
//   2. cutout_kernel       one workgroup per (sample, tile of output points):
//      phase A computes the per-(t,point) window parameters into LDS,
//      phase B produces 4 consecutive cutout samples per thread (float4 stores,
//      the tile's output region is one contiguous span).
#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileI = 32;          // output points per workgroup
constexpr int kMaxT = 16;           // scans per window
constexpr int kMaxPairs = kTileI * kMaxT;

struct CutArgs {
    const float *scans;
    int B, T, N, Ns, stride;
    const double *tab;
    int centered, fixed, P, area_mode;
    float half_width;   // float32(0.5 * window_width)
    float depth_f32;    // float32(window_depth)
    double depth;       // window_depth
    double rdepth;      // RN(1/window_depth)
    double padding;
    float *out;
    int32_t *s_area;
    int32_t *dbg_lo;
    int rows_in_lds;    // 1: whole [T][N] sample staged in LDS
};

// Window of one (t, point): everything the reference derives from `dists`.
struct Window {
    double a0;   // phi_i - half_alpha                      (float64)
    float ha;    // half_alpha                              (float32)
    float da;    // 2*half_alpha/(P-1)                      (float32)
    float d;     // the range that sizes the window         (float32)
};

__device__ __forceinline__ Window make_window(float d, double phi_i, float half_width, int P)
{
    Window w;
    w.d = d;
    float x = __fdiv_rn(half_width, fmaxf(d, 1e-2f));
    // correctly rounded float32 arctangent (float64 evaluation, rounded once)
    w.ha = (float)atan((double)x);
    w.da = __fdiv_rn(2.0f * w.ha, (float)(P - 1));
    w.a0 = phi_i - (double)w.ha;
    return w;
}

__device__ __forceinline__ double frac_index(const Window &w, double step, int k, double phi0,
                                             double dphi, double rdphi)
{
    double ang = w.a0 + (double)k * step;
    return pof_div_const(ang - phi0, dphi, rdphi);
}

__global__ __launch_bounds__(kThreads) void cutout_area_kernel(CutArgs a)
{
    __shared__ double s_max[kThreads / 64];
    __shared__ int s_any[kThreads / 64];
    const int b = blockIdx.x;
    const float *smp = a.scans + (long long)b * a.T * a.N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int rowsT = a.fixed ? a.T : 1;  // !fixed: every t has the same window
    double mx = -1.0e300;
    int any = 0;
    for (int p = threadIdx.x; p < rowsT * a.Ns; p += kThreads) {
        const int t = a.fixed ? p / a.Ns : a.T - 1;
        const int i = (p % a.Ns) * a.stride;
        Window w = make_window(smp[t * a.N + i], a.tab[i], a.half_width, a.P);
        const double step = (double)w.da;
        double width = frac_index(w, step, a.P - 1, phi0, dphi, rdphi) - frac_index(w, step, 0, phi0, dphi, rdphi);
        mx = fmax(mx, width);
        any |= width > (double)a.P;
    }
    mx = wave_max_f64(mx);
    any = __any(any);
    if ((threadIdx.x & 63) == 0) {
        s_max[threadIdx.x >> 6] = mx;
        s_any[threadIdx.x >> 6] = any;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kThreads / 64; ++wv) {
            mx = fmax(mx, s_max[wv]);
            any |= s_any[wv];
        }
        a.s_area[b] = any ? (int)ceil(mx / (double)a.P) : 0;
    }
}

template <bool VEC4, bool LDSROWS>
__global__ __launch_bounds__(kThreads) void cutout_kernel(CutArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS: window table for the tile, then (optionally) the sample's rows
    double *s_a0 = reinterpret_cast<double *>(smem);
    float *s_da = reinterpret_cast<float *>(s_a0 + kMaxPairs);
    float *s_d = s_da + kMaxPairs;
    float *s_daa = s_d + kMaxPairs;
    int *s_isarea = reinterpret_cast<int *>(s_daa + kMaxPairs);
    float *s_rows = reinterpret_cast<float *>(s_isarea + kMaxPairs);

    const int b = blockIdx.y;
    const int j0 = blockIdx.x * kTileI;
    const int nj = min(kTileI, a.Ns - j0);
    const int T = a.T, N = a.N, P = a.P;
    const float *smp = a.scans + (long long)b * T * N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int s_area = (a.area_mode && a.s_area) ? a.s_area[b] : 0;
    const int PA = s_area * P;

    if (LDSROWS) {
        for (int e = threadIdx.x; e < T * N; e += kThreads) s_rows[e] = smp[e];
    }
    // ---- phase A: window parameters of the tile's (point, t) pairs ----------
    const int npairs = nj * T;
    for (int p = threadIdx.x; p < npairs; p += kThreads) {
        const int jj = p / T, t = p - jj * T;
        const int i = (j0 + jj) * a.stride;
        const float d = smp[(a.fixed ? t : T - 1) * N + i];
        Window w = make_window(d, a.tab[i], a.half_width, P);
        s_a0[p] = w.a0;
        s_da[p] = w.da;
        s_d[p] = d;
        int isarea = 0;
        float daa = 0.0f;
        if (s_area > 0) {
            const double step = (double)w.da;
            double width = frac_index(w, step, P - 1, phi0, dphi, rdphi) - frac_index(w, step, 0, phi0, dphi, rdphi);
            isarea = width > (double)P;
            daa = __fdiv_rn(2.0f * w.ha, (float)(PA - 1));
        }
        s_isarea[p] = isarea;
        s_daa[p] = daa;
    }
    __syncthreads();

    const double nm1 = (double)(N - 1);
    constexpr int KV = VEC4 ? 4 : 1;
    const int per_pair = P / KV;               // float4 groups per (point, t)
    const int total = npairs * per_pair;
    float *out_tile = a.out + ((long long)b * a.Ns + j0) * T * P;

    for (int g = threadIdx.x; g < total; g += kThreads) {
        const int p = g / per_pair;
        const int k0 = (g - p * per_pair) * KV;
        const int jj = p / T, t = p - jj * T;
        Window w;
        w.a0 = s_a0[p];
        w.da = s_da[p];
        w.d = s_d[p];
        const double step = (double)w.da;
        const bool isarea = s_isarea[p] != 0;
        const double step_a = (double)s_daa[p];
        const double lo_clip = (double)(w.d - a.depth_f32), hi_clip = (double)(w.d + a.depth_f32);
        float res[KV];
#pragma unroll
        for (int u = 0; u < KV; ++u) {
            const int k = k0 + u;
            const double idx = frac_index(w, step, k, phi0, dphi, rdphi);
            const bool outb = (idx < 0.0) || (idx > nm1);
            const double fl = fmin(fmax(floor(idx), 0.0), nm1);
            const int lo = (int)fl;
            const int hi = min(lo + 1, N - 1);
            const double ratio = fmin(fmax(idx - fl, 0.0), 1.0);
            const float vlo = LDSROWS ? s_rows[t * N + lo] : smp[t * N + lo];
            const float vhi = LDSROWS ? s_rows[t * N + hi] : smp[t * N + hi];
            double ct = (double)vlo + ratio * (double)(vhi - vlo);
            if (isarea) {
                // area sampling: mean of s_area nearest-neighbour samples (float32 sum, in order)
                float acc = 0.0f;
                for (int s = 0; s < s_area; ++s) {
                    double ia = frac_index(w, step_a, k * s_area + s, phi0, dphi, rdphi);
                    ia = rint(fmin(fmax(ia, 0.0), nm1));
                    const float v = LDSROWS ? s_rows[t * N + (int)ia] : smp[t * N + (int)ia];
                    acc = (s == 0) ? v : acc + v;
                }
                ct = (double)__fdiv_rn(acc, (float)s_area);
            }
            if (outb) ct = a.padding;
            ct = fmin(fmax(ct, lo_clip), hi_clip);
            if (a.centered) {
                ct = ct - (double)w.d;
                ct = pof_div_const(ct, a.depth, a.rdepth);
            }
            res[u] = (float)ct;
            if (a.dbg_lo) a.dbg_lo[(((long long)b * P + k) * T + t) * a.Ns + (j0 + jj)] = lo;
        }
        if (VEC4) {
            reinterpret_cast<float4 *>(out_tile)[g] = make_float4(res[0], res[1], res[2], res[KV - 1]);
        } else {
            out_tile[g] = res[0];
        }
    }
}

}  // namespace

extern "C" int pof_cutout(const float *scans, int B, int T, int N, const double *tab, int stride,
                          int centered, int fixed, double window_width, double window_depth,
                          int num_cutout_pts, double padding_val, int area_mode, float *out,
                          int32_t *workspace, int32_t *dbg_lo, pof_stream_t stream)
{
    if (!scans || !tab || !out || B < 0 || T < 1 || N < 2 || stride < 1 || num_cutout_pts < 2)
        return POF_E_BADARG;
    if (area_mode && !workspace) return POF_E_WORKSPACE;
    if (!(window_depth > 0.0) || !(window_width > 0.0)) return POF_E_BADARG;
    if (T > kMaxT) return POF_E_SHAPE;  // windows of up to 16 scans
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    CutArgs a;
    a.scans = scans; a.B = B; a.T = T; a.N = N; a.stride = stride;
    a.Ns = (N + stride - 1) / stride;
    a.tab = tab; a.centered = centered; a.fixed = fixed; a.P = num_cutout_pts; a.area_mode = area_mode;
    a.half_width = (float)(0.5 * window_width);
    a.depth_f32 = (float)window_depth;
    a.depth = window_depth;
    a.rdepth = 1.0 / window_depth;
    a.padding = padding_val;
    a.out = out; a.s_area = area_mode ? workspace : nullptr; a.dbg_lo = dbg_lo;
    hipStream_t s = pof_stream(stream);
    if (area_mode) {
        cutout_area_kernel<<<B, kThreads, 0, s>>>(a);
        POF_CHECK_LAUNCH();
    }
    const size_t table_bytes = kMaxPairs * (sizeof(double) + 3 * sizeof(float) + sizeof(int));
    const size_t row_bytes = (size_t)T * N * sizeof(float);
    a.rows_in_lds = table_bytes + row_bytes <= 64 * 1024;
    const size_t lds = table_bytes + (a.rows_in_lds ? row_bytes : 0);
    dim3 grid((a.Ns + kTileI - 1) / kTileI, B);
    const bool vec4 = (num_cutout_pts % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec4 && a.rows_in_lds) cutout_kernel<true, true><<<grid, kThreads, lds, s>>>(a);
    else if (vec4) cutout_kernel<true, false><<<grid, kThreads, lds, s>>>(a);
    else if (a.rows_in_lds) cutout_kernel<false, true><<<grid, kThreads, lds, s>>>(a);
    else cutout_kernel<false, false><<<grid, kThreads, lds, s>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
