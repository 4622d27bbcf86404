// A8: scans_to_cutout (src/utils/utils.py:259-334) for a whole batch.
//
// scans [B][T][N] float32  ->  out [B][Ns][T][P] float32   (Ns = ceil(N/stride))
//
// Roofline: HBM, dominated by the output: (T*N*4 read + Ns*T*P*4 written) per
// sample = 513 000 B at T=5, N=450, P=56.  The float64 index math (angle ->
// fractional index -> floor/lerp) must round exactly like NumPy, so it stays in
// float64 with contraction off; at ~19 float64 instructions per output the VALU
// time is of the same order as the HBM time, which is why the kernel is laid
// out to make every per-window quantity wave-uniform:
//
//   1. cutout_area_kernel  (area_mode only) per-sample max window width ->
//      s_area[b] = ceil(max/P) if any window covers more than P raw points, else 0
//      (the reference takes this max over the whole (T,N) call, utils.py:304-308);
//   2. cutout_kernel       one workgroup per (sample, tile of 32 output points):
//      phase A: one lane per window ((point,t) when `fixed`, point otherwise)
//               computes the window table into LDS (correctly rounded float32
//               half-angle, float64 start angle / step / clip bounds);
//      phase B: one wave64 per window, lane = cutout sample k (P <= 64 lanes
//               active; 64/P windows per wave when P <= 32).  Window parameters
//               are LDS broadcasts, the lane's float64(k) is loop invariant, the
//               range rows sit in LDS as (value, next-value - value) float2 so
//               one ds_read_b64 feeds the lerp, and each wave stores P
//               consecutive floats (the tile's output region is contiguous).
#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kTileI = 32;          // output points per workgroup
constexpr int kMaxT = 16;           // scans per window

struct CutArgs {
    const float *scans;
    int B, T, N, Ns, stride;
    const double *tab;
    int centered, fixed, P, area_mode;
    float half_width;   // float32(0.5 * window_width)
    float depth_f32;    // float32(window_depth)
    double depth;       // window_depth
    double rdepth;      // RN(1/window_depth)
    int depth_pow2;     // 1/window_depth is exact: multiply instead of divide
    double padding;
    float *out;
    int32_t *s_area;
    int32_t *dbg_lo;
};

// Window of one (t, point): everything the reference derives from `dists`.
struct Window {
    double a0;   // phi_i - half_alpha                      (float64)
    float ha;    // half_alpha                              (float32)
    float da;    // 2*half_alpha/(P-1)                      (float32)
};

__device__ __forceinline__ Window make_window(float d, double phi_i, float half_width, int P)
{
    Window w;
    float x = __fdiv_rn(half_width, fmaxf(d, 1e-2f));
    // correctly rounded float32 arctangent (float64 evaluation, rounded once)
    w.ha = (float)atan((double)x);
    w.da = __fdiv_rn(2.0f * w.ha, (float)(P - 1));
    w.a0 = phi_i - (double)w.ha;
    return w;
}

// fractional scan index of cutout sample k: ((a0 + k*step) - phi0) / dphi
__device__ __forceinline__ double frac_index(double a0, double step, double kd, double phi0,
                                             double dphi, double rdphi)
{
    double ang = a0 + kd * step;
    return pof_div_const(ang - phi0, dphi, rdphi);
}

__global__ __launch_bounds__(kThreads) void cutout_area_kernel(CutArgs a)
{
    __shared__ double s_max[kWaves];
    __shared__ int s_any[kWaves];
    const int b = blockIdx.x;
    const float *smp = a.scans + (long long)b * a.T * a.N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int rowsT = a.fixed ? a.T : 1;  // !fixed: every t has the same window
    const double pm1 = (double)(a.P - 1);
    double mx = -1.0e300;
    int any = 0;
    for (int p = threadIdx.x; p < rowsT * a.Ns; p += kThreads) {
        const int t = a.fixed ? p / a.Ns : a.T - 1;
        const int i = (p % a.Ns) * a.stride;
        Window w = make_window(smp[t * a.N + i], a.tab[i], a.half_width, a.P);
        const double step = (double)w.da;
        double width = frac_index(w.a0, step, pm1, phi0, dphi, rdphi) - frac_index(w.a0, step, 0.0, phi0, dphi, rdphi);
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
        for (int wv = 1; wv < kWaves; ++wv) {
            mx = fmax(mx, s_max[wv]);
            any |= s_any[wv];
        }
        a.s_area[b] = any ? (int)ceil(mx / (double)a.P) : 0;
    }
}

// LDS window table: structure of arrays with `cap` entries each, carved from
// dynamic LDS (cap = kTileI*T when `fixed`, kTileI otherwise).
struct WinTable {
    double *a0, *step, *lo_clip, *hi_clip, *dd, *step_a;
    int *isarea;
    __device__ WinTable(unsigned char *base, int cap)
    {
        a0 = reinterpret_cast<double *>(base);
        step = a0 + cap;
        lo_clip = step + cap;
        hi_clip = lo_clip + cap;
        dd = hi_clip + cap;
        step_a = dd + cap;
        isarea = reinterpret_cast<int *>(step_a + cap);
    }
};
// bytes per entry, rounded so the rows that follow stay 16-byte aligned
__host__ __device__ inline size_t win_table_bytes(int cap) { return ((size_t)cap * 52 + 15) & ~(size_t)15; }

template <bool LDSROWS>
__global__ __launch_bounds__(kThreads) void cutout_kernel(CutArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int cap = a.fixed ? kTileI * a.T : kTileI;
    WinTable wt(smem, cap);
    float2 *s_rows = reinterpret_cast<float2 *>(smem + win_table_bytes(cap));  // [T][N] (value, delta)

    const int b = blockIdx.y;
    const int j0 = blockIdx.x * kTileI;
    const int nj = min(kTileI, a.Ns - j0);
    const int T = a.T, N = a.N, P = a.P;
    const float *smp = a.scans + (long long)b * T * N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int s_area = (a.area_mode && a.s_area) ? a.s_area[b] : 0;
    const int PA = s_area * P;

    if (LDSROWS) {
        for (int e = threadIdx.x; e < T * N; e += kThreads) {
            const int i = e % N;
            const float v = smp[e];
            const float nx = (i + 1 < N) ? smp[e + 1] : v;   // hi index clamps to N-1
            s_rows[e] = make_float2(v, nx - v);
        }
    }
    // ---- phase A: window table ------------------------------------------------
    // fixed: one window per (point, t), entry jj*T + t;  else one per point, entry jj
    const int nwin = a.fixed ? nj * T : nj;
    for (int p = threadIdx.x; p < nwin; p += kThreads) {
        const int jj = a.fixed ? p / T : p;
        const int t = a.fixed ? p - jj * T : T - 1;
        const int i = (j0 + jj) * a.stride;
        const float d = smp[t * N + i];
        Window w = make_window(d, a.tab[i], a.half_width, P);
        const double step = (double)w.da;
        wt.a0[p] = w.a0;
        wt.step[p] = step;
        wt.lo_clip[p] = (double)(d - a.depth_f32);
        wt.hi_clip[p] = (double)(d + a.depth_f32);
        wt.dd[p] = (double)d;
        int isarea = 0;
        double step_a = 0.0;
        if (s_area > 0) {
            double width = frac_index(w.a0, step, (double)(P - 1), phi0, dphi, rdphi) -
                           frac_index(w.a0, step, 0.0, phi0, dphi, rdphi);
            isarea = width > (double)P;
            step_a = (double)__fdiv_rn(2.0f * w.ha, (float)(PA - 1));
        }
        wt.isarea[p] = isarea;
        wt.step_a[p] = step_a;
    }
    __syncthreads();

    // ---- phase B: one wave per window (or 64/P windows per wave) -----------------
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub_per_wave = (P <= 32) ? 64 / P : 1;      // windows handled side by side
    const int sub = (P <= 32) ? lane / P : 0;
    const bool sub_ok = sub < sub_per_wave;
    const double nm1 = (double)(N - 1);
    float *out_tile = a.out + ((long long)b * a.Ns + j0) * T * P;
    const int tcount = a.fixed ? 1 : T;

    for (int kbase = 0; kbase < P; kbase += 64) {
        const int k = (P <= 32) ? lane - sub * P : kbase + lane;
        const bool k_ok = sub_ok && k < P;
        const double kd = (double)k;
        for (int p0 = wave * sub_per_wave; p0 < nwin; p0 += kWaves * sub_per_wave) {
            const int p = p0 + sub;
            if (!(k_ok && p < nwin)) continue;
            const double a0 = wt.a0[p], step = wt.step[p];
            const double lo_clip = wt.lo_clip[p], hi_clip = wt.hi_clip[p], dd = wt.dd[p];
            const bool isarea = wt.isarea[p] != 0;
            const double idx = frac_index(a0, step, kd, phi0, dphi, rdphi);
            // idx < 0 via the sign bit (idx is never -0: RN(x - x) = +0), idx > N-1 compared
            const bool outb = (__double2hiint(idx) < 0) || (idx > nm1);
            // in range: trunc == floor and idx - floor(idx) is exact; out of range the
            // value is overwritten by the padding, only the LDS address must stay legal
            const int lo = min(max((int)idx, 0), N - 1);
            const double ratio = idx - floor(idx);
            const int jj = a.fixed ? p / T : p;
            const int tfirst = a.fixed ? p - jj * T : 0;
            if (a.dbg_lo) {
                for (int tt = 0; tt < tcount; ++tt)
                    a.dbg_lo[(((long long)b * P + k) * T + tfirst + tt) * a.Ns + (j0 + jj)] = lo;
            }
            for (int tt = 0; tt < tcount; ++tt) {
                const int t = tfirst + tt;
                float2 vd;
                if (LDSROWS) {
                    vd = s_rows[t * N + lo];
                } else {
                    const float v = smp[t * N + lo];
                    vd = make_float2(v, smp[t * N + min(lo + 1, N - 1)] - v);
                }
                double ct = (double)vd.x + ratio * (double)vd.y;
                if (isarea) {
                    // area sampling: mean of s_area nearest-neighbour samples (float32 sum, in order)
                    const double step_a = wt.step_a[p];
                    float acc = 0.0f;
                    for (int s = 0; s < s_area; ++s) {
                        double ia = frac_index(a0, step_a, (double)(k * s_area + s), phi0, dphi, rdphi);
                        ia = rint(fmin(fmax(ia, 0.0), nm1));
                        const float v = LDSROWS ? s_rows[t * N + (int)ia].x : smp[t * N + (int)ia];
                        acc = (s == 0) ? v : acc + v;
                    }
                    ct = (double)__fdiv_rn(acc, (float)s_area);
                }
                if (outb) ct = a.padding;
                ct = fmin(fmax(ct, lo_clip), hi_clip);
                if (a.centered) {
                    ct = ct - dd;
                    ct = a.depth_pow2 ? ct * a.rdepth : pof_div_const(ct, a.depth, a.rdepth);
                }
                out_tile[((long long)jj * T + t) * P + k] = (float)ct;
            }
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
    {
        int e;
        a.depth_pow2 = frexp(window_depth, &e) == 0.5;  // power of two: x/depth == x*(1/depth) exactly
    }
    a.padding = padding_val;
    a.out = out; a.s_area = area_mode ? workspace : nullptr; a.dbg_lo = dbg_lo;
    hipStream_t s = pof_stream(stream);
    if (area_mode) {
        cutout_area_kernel<<<B, kThreads, 0, s>>>(a);
        POF_CHECK_LAUNCH();
    }
    const size_t row_bytes = (size_t)T * N * sizeof(float2);
    const size_t tbl = win_table_bytes(fixed ? kTileI * T : kTileI);
    const bool rows_in_lds = tbl + row_bytes <= 64 * 1024;
    const size_t lds = tbl + (rows_in_lds ? row_bytes : 0);
    dim3 grid((a.Ns + kTileI - 1) / kTileI, B);
    if (rows_in_lds) cutout_kernel<true><<<grid, kThreads, lds, s>>>(a);
    else cutout_kernel<false><<<grid, kThreads, lds, s>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
