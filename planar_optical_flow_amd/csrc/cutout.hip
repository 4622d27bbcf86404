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
//   2. cutout_kernel       one workgroup per (sample, tile of ~256/T output points):
//      phase A: one lane per window ((point,t) when `fixed`, point otherwise)
//               computes the window table into LDS (correctly rounded float32
//               half-angle, float64 start angle / step / clip bounds);
//      phase B: every lane produces 4 consecutive cutout samples of one window
//               (one float4 store; the tile's output region is one contiguous
//               span).  The window parameters are read from LDS once per 4
//               outputs, the range rows sit in LDS (loaded into registers at
//               kernel entry, their latency hidden behind phase A), floor/ratio
//               use trunc + v_fract (exact in range), np.clip is compare + select.
#include <cstdlib>
#include <string>
#include <type_traits>

#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxWin = 256;        // window-table entries per workgroup (phase A: one lane each)
constexpr int kMaxT = 16;           // scans per window

struct CutArgs {
    const float *scans;
    int B, T, N, Ns, stride;
    int tile;           // output points per workgroup
    int span_cap;       // LDSMODE 2: row elements per scan that fit the LDS row buffer
    const double *tab;
    int centered, fixed, P, area_mode;
    float half_width;   // float32(0.5 * window_width)
    float depth_f32;    // float32(window_depth)
    double depth;       // window_depth
    double rdepth;      // RN(1/window_depth)
    int depth_pow2;     // 1/window_depth is exact: multiply instead of divide
    int value_mode;     // 0: float64 value path (bit-exact); 1: float32 value path (exact indices)
    float padding_f32, rdepth_f32;
    double padding;
    float *out;
    _Float16 *out16;    // config 5 storage: float16 output (same values, rounded once more); out is null then
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
    // k*step is exact in float64 (k < 2^12, step has a 24-bit significand), so the reference's
    // RN(a0 + RN(k*step)) is one fused multiply-add
    const double ang = fma(kd, step, a0);
    return pof_div_const(ang - phi0, dphi, rdphi);
}

// s_area[b] = max over the sample's windows of (width > P ? ceil(width / P) : 0).
// ceil is monotone, so this equals ceil(max width / P) whenever any window exceeds P
// (what the reference computes) and 0 otherwise.  Grid = (window chunks, B) with an
// integer atomicMax per workgroup; s_area is zeroed by the launcher.
__global__ __launch_bounds__(kThreads) void cutout_area_kernel(CutArgs a)
{
    __shared__ int s_red[kWaves];
    const int b = blockIdx.y;
    const float *smp = a.scans + (long long)b * a.T * a.N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int rowsT = a.fixed ? a.T : 1;  // !fixed: every t has the same window
    const double pm1 = (double)(a.P - 1);
    int best = 0;
    const int total = rowsT * a.Ns;
    for (int p = blockIdx.x * kThreads + threadIdx.x; p < total; p += gridDim.x * kThreads) {
        const int t = a.fixed ? p / a.Ns : a.T - 1;
        const int i = (p % a.Ns) * a.stride;
        Window w = make_window(smp[t * a.N + i], a.tab[i], a.half_width, a.P);
        const double step = (double)w.da;
        const double width = frac_index(w.a0, step, pm1, phi0, dphi, rdphi) - frac_index(w.a0, step, 0.0, phi0, dphi, rdphi);
        if (width > (double)a.P) best = max(best, (int)ceil(width / (double)a.P));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < kWaves; ++wv) best = max(best, s_red[wv]);
        if (best > 0) atomicMax(&a.s_area[b], best);
    }
}

// LDS window table: structure of arrays with `cap` entries each, carved from
// dynamic LDS (cap = tile*T when `fixed`, tile otherwise; <= kMaxWin).
struct WinTable {
    double *a0, *step, *dd, *step_a;
    float *ylo, *yhi, *ypad;      // output-space clip bounds and the padded output value
    int *isarea, *out_off, *row_off, *alist, *krange;
    __device__ WinTable(unsigned char *base, int cap)
    {
        a0 = reinterpret_cast<double *>(base);
        step = a0 + cap;
        dd = step + cap;
        step_a = dd + cap;
        ylo = reinterpret_cast<float *>(step_a + cap);
        yhi = ylo + cap;
        ypad = yhi + cap;
        isarea = reinterpret_cast<int *>(ypad + cap);
        out_off = isarea + cap;
        row_off = out_off + cap;
        alist = row_off + cap;
        krange = alist + cap;
    }
};
// bytes per entry (4 doubles + 3 floats + 5 ints), rounded so the rows that follow stay 16-byte aligned
__host__ __device__ inline size_t win_table_bytes(int cap) { return ((size_t)cap * 64 + 15) & ~(size_t)15; }

// The tail of the reference's arithmetic, ct -> output: optional centring and scaling in
// float64, then the float32 cast.  It is monotone in ct, so np.clip(ct, lo, hi) before it
// equals a float32 clamp after it with the bounds pushed through the same function --
// which turns two float64 compare/selects per sample into one v_med3_f32.
template <int VMODE>
__device__ __forceinline__ float finish_value(const CutArgs &a, double ct, double dd)
{
    // VMODE 1: 1/depth is a power of two: scaling by it commutes with the rounding to float32 (no overflow /
    // underflow at these magnitudes), so the float64 multiply becomes a float32 one
    if (VMODE == 1) return (float)(ct - dd) * a.rdepth_f32;
    if (a.centered) {
        ct = ct - dd;
        ct = a.depth_pow2 ? ct * a.rdepth : pof_div_const(ct, a.depth, a.rdepth);
    }
    return (float)ct;
}

// One workgroup = one sample x `tile` output points.
//   LDSMODE  1: the sample's whole [T][N] rows are staged in LDS (parked in registers at
//               kernel entry, latency hidden behind phase A);
//            2: rows too large for LDS (N = 3600): after phase A the workgroup reduces
//               the index range its windows touch and stages only that span of every
//               row (falls back to L2 gathers when even the span does not fit);
//            0: gathers straight from global memory
//            3 (round 3; VMODE 1 only): as 1, but the rows are staged as float32 PAIRS {v[i] / depth, (v[i+1] - v[i]) /
//               depth}, so that a lerp tap is one 8-byte LDS read and the float32 subtraction and the scaling by
//               1/depth (a power of two: it commutes with every rounding of the sequence) leave the per-sample path:
//               11 -> 8 vector instructions per tap.  Worth 7 % where one index serves T scans (fixed = False);
//               nothing at fixed = True, which is bound by the float64-rate instructions (a float64-pair form that
//               also drops the two conversions needs 2x the LDS and measured slower: profiles/r3_cutout_experiments.txt)
//   P4       > 0: P/8 as a compile-time constant (7, 6, 4 for P = 56, 48, 32): the lane ->
//            (window, k-group) split is a multiply and every lane produces 8 consecutive
//            cutout samples (two float4 stores), which amortises the per-window LDS reads;
//            0: 4 samples per lane with a runtime P/4;  -1: P % 4 != 0, one sample per lane.
//   VMODE    0: float64 value path, any depth / centring (bit-exact);
//            1: same, specialised for centred output with a power-of-two depth (the
//               reference's configs): (ct - d) * (1/depth);
//            2: float32 value path -- the index math (angle -> fractional index -> floor,
//               out-of-range, area indices) stays float64 and exact, only the lerp / clip /
//               centre arithmetic runs in float32 (|error| <= 1e-5 in the normalised output;
//               saturated samples are exactly +-1).  Opt-in (value_mode = 1).
//   DBG      also write the inds_ct_low debug tensor (tests only)
#ifndef POF_CUTOUT_MINWG
#define POF_CUTOUT_MINWG 4      // workgroups per CU the register allocation is held to (tools: experimental builds)
#endif
template <int LDSMODE, int P4, int VMODE, bool DBG>
__global__ __launch_bounds__(kThreads, POF_CUTOUT_MINWG) void cutout_kernel(CutArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int cap = a.fixed ? a.tile * a.T : a.tile;
    WinTable wt(smem, cap);
    float *s_rows = reinterpret_cast<float *>(smem + win_table_bytes(cap));  // [T][N] range rows

    const int b = blockIdx.y;
    const int j0 = blockIdx.x * a.tile;
    const int nj = min(a.tile, a.Ns - j0);
    const int T = a.T, N = a.N, P = a.P;
    const float *smp = a.scans + (long long)b * T * N;
    const double phi0 = a.tab[0], dphi = a.tab[1] - a.tab[0], rdphi = 1.0 / dphi;
    const int s_area = (a.area_mode && a.s_area) ? a.s_area[b] : 0;
    const int PA = s_area * P;
    constexpr bool LDSROWS = LDSMODE == 1 || LDSMODE == 3;
    constexpr int PAIR = LDSMODE == 3 ? 1 : 0;
    static_assert(PAIR == 0 || (VMODE == 1 && !DBG), "pair rows: centred output with a power-of-two depth only");
    using PairT = float2;
    PairT *s_pair = reinterpret_cast<PairT *>(s_rows);
    __shared__ int s_span_lo, s_span_hi, s_acount;
    // k as float64, k < 64: phase B reads its eight k values per lane from here (two per 16-byte LDS read) instead
    // of forming them with eight float64 additions
    __shared__ __align__(16) double s_kd[64];
    if (threadIdx.x < 64) s_kd[threadIdx.x] = (double)threadIdx.x;
    if (threadIdx.x == 0) {
        s_span_lo = N;
        s_span_hi = -1;
        s_acount = 0;
    }
    __syncthreads();

    // Row loads are issued first and parked in registers; their latency is covered
    // by the arctangents of phase A, and they are written to LDS just before the
    // barrier.  (float2 loads: a sample starts on an 8-byte boundary when T*N is even.)
    constexpr int kMaxStage = 10;
    const int nvec = (T * N) >> 1;
    const bool vec_stage = LDSROWS && ((T * N) & 1) == 0 && nvec <= kMaxStage * kThreads &&
                           ((reinterpret_cast<uintptr_t>(a.scans) & 7) == 0);
    float2 stage[kMaxStage];
    if (vec_stage) {
#pragma unroll
        for (int v = 0; v < kMaxStage; ++v) {
            const int e = threadIdx.x + v * kThreads;
            stage[v] = (e < nvec) ? reinterpret_cast<const float2 *>(smp)[e] : make_float2(0.f, 0.f);
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
        {
            // clip bounds are float32 in the reference (dists -/+ window_depth on a float32 array)
            const double lo_c = (double)(d - a.depth_f32), hi_c = (double)(d + a.depth_f32), dd_ = (double)d;
            constexpr int VT = (VMODE == 1) ? 1 : 0;   // the exact tail also defines the bounds of VMODE 2
            const float ylo = finish_value<VT>(a, lo_c, dd_), yhi = finish_value<VT>(a, hi_c, dd_);
            const double padc = fmin(fmax(a.padding, lo_c), hi_c);
            wt.ylo[p] = ylo;
            wt.yhi[p] = yhi;
            wt.ypad[p] = finish_value<VT>(a, padc, dd_);
            wt.dd[p] = dd_;
        }
        wt.out_off[p] = a.fixed ? p * P : jj * T * P;   // (jj*T + t)*P, t = 0 when !fixed
        wt.row_off[p] = a.fixed ? t : 0;                // first scan row this window reads
        // The fractional index is monotone in k (every rounding of its sequence is), so the samples that fall
        // inside the field of view, 0 <= idx <= N-1, are one interval [klo, khi] of k: found here once per
        // window (exact evaluations at the ends; a short exact search only for windows that stick out of the
        // field of view), so that phase B needs no per-sample range test for the windows that lie inside.
        const double i_first = frac_index(w.a0, step, 0.0, phi0, dphi, rdphi);
        const double i_last = frac_index(w.a0, step, (double)(P - 1), phi0, dphi, rdphi);
        int klo = 0, khi = P - 1;
        {
            const double nm1_ = (double)(N - 1);
            auto neg = [](double x) { return __double2hiint(x) < 0; };
            if (neg(i_first) || i_last > nm1_) {
                const double c1 = step * rdphi, c0 = (w.a0 - phi0) * rdphi;
                if (neg(i_first)) {
                    const double e = ceil(-c0 / c1);
                    klo = (e >= 0.0 && e <= (double)P) ? (int)e : (e > (double)P ? P : 0);
                    while (klo > 0 && !neg(frac_index(w.a0, step, (double)(klo - 1), phi0, dphi, rdphi))) --klo;
                    while (klo < P && neg(frac_index(w.a0, step, (double)klo, phi0, dphi, rdphi))) ++klo;
                }
                if (i_last > nm1_) {
                    const double e = floor((nm1_ - c0) / c1);
                    khi = (e >= -1.0 && e <= (double)(P - 1)) ? (int)e : (e > (double)(P - 1) ? P - 1 : -1);
                    while (khi < P - 1 && !(frac_index(w.a0, step, (double)(khi + 1), phi0, dphi, rdphi) > nm1_)) ++khi;
                    while (khi >= 0 && frac_index(w.a0, step, (double)khi, phi0, dphi, rdphi) > nm1_) --khi;
                }
            }
        }
        wt.krange[p] = klo | (khi << 16);
        if (LDSMODE == 2) {
            // index range touched by this window (lerp: floor(idx), +1; area samples stay in
            // the same angular span; one extra element each side for the float32 step rounding)
            const int lo_w = min(max((int)floor(i_first) - 1, 0), N - 1);
            const int hi_w = min(max((int)floor(i_last) + 2, 0), N - 1);
            atomicMin(&s_span_lo, lo_w);
            atomicMax(&s_span_hi, hi_w);
        }
        int isarea = 0;
        double step_a = 0.0;
        if (s_area > 0) {
            const double width = i_last - i_first;
            isarea = width > (double)P;
            step_a = (double)__fdiv_rn(2.0f * w.ha, (float)(PA - 1));
        }
        wt.isarea[p] = isarea;
        wt.step_a[p] = step_a;
        if (isarea) wt.alist[atomicAdd(&s_acount, 1)] = p;
    }
    if (PAIR) {
        // {v, next - v} scaled by 1/depth; "next" of a row's last beam is the next row's first (0 behind the last
        // row) -- the word the two-tap read of the plain layout finds there; it only ever meets ratio 0
        const float sc = a.rdepth_f32;
        auto mk = [&](float v, float nx) {
            PairT r;
            r.x = sc * v;
            r.y = sc * (nx - v);
            return r;
        };
        if (vec_stage) {
            // the element behind a lane's two is the next lane's first: a shuffle inside the wave, a word through LDS
            // at the wave's end (lane 0 of every wave publishes its first elements)
            __shared__ float s_first[kWaves][kMaxStage + 1];
            const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
            if (ln == 0) {
#pragma unroll
                for (int v = 0; v < kMaxStage; ++v) s_first[wv][v] = stage[v].x;
                s_first[wv][kMaxStage] = 0.0f;
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < kMaxStage; ++v) {
                const int e = threadIdx.x + v * kThreads;
                float nx = __shfl_down(stage[v].x, 1, 64);
                if (ln == 63) nx = wv + 1 < kWaves ? s_first[wv + 1][v] : s_first[0][v + 1];
                if (2 * e + 2 >= T * N) nx = 0.0f;
                if (e < nvec) {
                    s_pair[2 * e] = mk(stage[v].x, stage[v].y);
                    s_pair[2 * e + 1] = mk(stage[v].y, nx);
                }
            }
        } else {
            for (int e = threadIdx.x; e < T * N; e += kThreads) s_pair[e] = mk(smp[e], e + 1 < T * N ? smp[e + 1] : 0.0f);
        }
    } else if (LDSROWS) {
        if (threadIdx.x == 0) s_rows[T * N] = 0.0f;      // pad word behind the last row (second lerp tap)
        if (vec_stage) {
#pragma unroll
            for (int v = 0; v < kMaxStage; ++v) {
                const int e = threadIdx.x + v * kThreads;
                if (e < nvec) reinterpret_cast<float2 *>(s_rows)[e] = stage[v];
            }
        } else {
            for (int e = threadIdx.x; e < T * N; e += kThreads) s_rows[e] = smp[e];
        }
    }
    __syncthreads();
    // row addressing: element idx of scan row t lives at rows[(t * rstride) + rbase + idx]
    int rstride = N, rbase = 0;
    bool span_lds = false;
    if (LDSMODE == 2) {
        const int lo_s = s_span_lo, span = s_span_hi - s_span_lo + 1;
        span_lds = span > 0 && span <= a.span_cap;
        if (span_lds) {
            for (int e = threadIdx.x; e < T * span; e += kThreads) {
                const int t = e / span, x = e - t * span;
                s_rows[e] = smp[t * N + lo_s + x];
            }
            rstride = span;
            rbase = -lo_s;
        }
        __syncthreads();
    }
    // row source as a compile-time tag: a pointer selected at run time would turn every tap into a flat_load
    // (pair rows: the value comes back scaled by 1/depth -- sums, the mean and the centring commute with it)
    auto fetch_from = [&](auto lds_tag, int off) -> float {
        if (PAIR) return (float)s_pair[off].x;
        if (decltype(lds_tag)::value) return s_rows[off];
        return smp[off];
    };

    // ---- phase B ----------------------------------------------------------------
    constexpr int KV = (P4 > 0) ? 8 : (P4 == 0 ? 4 : 1);   // cutout samples per lane and iteration
    const int per_win = (P4 > 0) ? P4 : (P4 == 0 ? P / 4 : P);
    const int total = nwin * per_win;
    const double nm1 = (double)(N - 1);
    const long long tile_off = ((long long)b * a.Ns + j0) * T * P;
    float *out_tile = a.out ? a.out + tile_off : nullptr;
    _Float16 *out_tile16 = a.out16 ? a.out16 + tile_off : nullptr;
    const int tcount = a.fixed ? 1 : T;

    // One (window, KV-sample group) of the interpolation path; area-sampled windows (more than P raw points
    // under the window) go through area_group below, so that neither path carries the other's registers.
    auto group = [&](auto full_tag, auto lds_tag, const int p, const int k0) {
        auto fetch = [&](int off) -> float { return fetch_from(lds_tag, off); };
        // FULL: all KV samples of this lane lie inside the field of view (phase A's k interval): no range
        // test, no index clamp, no padding select
        constexpr bool FULL = decltype(full_tag)::value;
        const double a0 = wt.a0[p], step = wt.step[p];
        const double dd = PAIR ? wt.dd[p] * a.rdepth : wt.dd[p];       // pair rows: everything in units of the depth
        const float ylo = wt.ylo[p], yhi = wt.yhi[p], ypad = wt.ypad[p];
        const int out_off = wt.out_off[p], row_off = wt.row_off[p];
        const int kr = FULL ? 0 : wt.krange[p];
        const int klo = kr & 0xffff, khi = kr >> 16;
        const double kd0 = (double)k0;
        using RatioT = typename std::conditional<VMODE == 2, float, double>::type;
        RatioT ratio[KV];
        int lo[KV];
        bool outb[KV];
#pragma unroll
        for (int u = 0; u < KV; ++u) outb[u] = FULL ? false : (k0 + u < klo || k0 + u > khi);
        if (VMODE == 2) {
            // Approximate-then-verify index: idx' = c0 + k*c1 with c0 = (a0-phi0)/dphi, c1 = step/dphi
            // is within 3e-12 of the reference's rounding sequence (N <= 4096), so floor and fract
            // agree with it unless idx' is within 1e-6 of an integer; only those lanes (2e-6 of all
            // samples) run the exact sequence.  One float64 FMA instead of five.
            const double c1 = step * rdphi;
            const double base = fma(kd0, c1, (a0 - phi0) * rdphi);
#pragma unroll
            for (int u = 0; u < KV; ++u) {
                const double idx = fma((double)u, c1, base);
                float rf = (float)__builtin_amdgcn_fract(idx);
                int l = (int)idx;
                if (!(rf >= 1e-6f && rf <= 1.0f - 1e-6f)) {   // near an integer (or NaN): exact sequence
                    const double ie = frac_index(a0, step, kd0 + (double)u, phi0, dphi, rdphi);
                    l = (int)ie;
                    rf = (float)__builtin_amdgcn_fract(ie);
                }
                lo[u] = FULL ? l : min(max(l, 0), N - 1);
                ratio[u] = rf;
            }
        } else {
            double kdv[KV];
            if (P4 > 0) {
#pragma unroll
                for (int u = 0; u < KV; u += 2) {
                    const double2 t2 = *reinterpret_cast<const double2 *>(&s_kd[k0 + u]);   // k0 is a multiple of 8
                    kdv[u] = t2.x;
                    kdv[(u + 1) % KV] = t2.y;
                }
            } else {
#pragma unroll
                for (int u = 0; u < KV; ++u) kdv[u] = kd0 + (double)u;
            }
#pragma unroll
            for (int u = 0; u < KV; ++u) {
                const double idx = frac_index(a0, step, kdv[u], phi0, dphi, rdphi);
                // in range trunc == floor and fract(idx) == idx - floor(idx) exactly; out of
                // range the value is replaced by the padding, only the address must stay legal
                lo[u] = FULL ? (int)idx : min(max((int)idx, 0), N - 1);
                ratio[u] = __builtin_amdgcn_fract(idx);
            }
        }
        if (DBG) {
            const int jj = a.fixed ? p / T : p, tfirst = a.fixed ? p - jj * T : 0;
            for (int tt = 0; tt < tcount; ++tt)
#pragma unroll
                for (int u = 0; u < KV; ++u)
                    a.dbg_lo[(((long long)b * P + k0 + u) * T + tfirst + tt) * a.Ns + (j0 + jj)] = lo[u];
        }
        for (int tt = 0; tt < tcount; ++tt) {
            const int roff = (row_off + tt) * rstride + rbase;
            float res[KV];
#pragma unroll
            for (int u = 0; u < KV; ++u) {
                float y;
                if (PAIR) {
                    const PairT tv = s_pair[roff + lo[u]];
                    const double ct = (double)tv.x + ratio[u] * (double)tv.y;
                    y = (float)(ct - dd);
                } else {
                    float vlo, vhi;
                    if (LDSMODE == 1) {
                        // both taps with one LDS access (adjacent words).  At lo = N-1 the second tap is the
                        // next row's first element (or the zeroed pad word after the last row): only
                        // idx == N-1 exactly gets there in range, with ratio 0, and 0 * finite = 0
                        // byte address = (lo << 2) + row base: one v_lshl_add_u32, both taps from one ds_read2_b32
                        const float *tap = reinterpret_cast<const float *>(
                            reinterpret_cast<const unsigned char *>(s_rows) + ((lo[u] << 2) + (roff << 2)));
                        vlo = tap[0];
                        vhi = tap[1];
                    } else {
                        vlo = fetch(roff + lo[u]);
                        vhi = fetch(roff + min(lo[u] + 1, N - 1));
                    }
                    if (VMODE == 2) {
                        float v = fmaf((float)ratio[u], vhi - vlo, vlo);
                        const float df = (float)dd;  // exact: a float32 value
                        if (a.centered) v = a.depth_pow2 ? (v - df) * a.rdepth_f32 : __fdiv_rn(v - df, a.depth_f32);
                        y = v;
                    } else {
                        const double ct = (double)vlo + ratio[u] * (double)(vhi - vlo);
                        y = finish_value<VMODE>(a, ct, dd);
                    }
                }
                // np.clip pushed through the monotone tail (one v_med3_f32; ylo <= yhi), then the
                // out-of-FOV padding
                y = __builtin_amdgcn_fmed3f(y, ylo, yhi);
                res[u] = (!FULL && outb[u]) ? ypad : y;
            }
            const int o_el = out_off + tt * P + k0;
            if (out_tile16) {   // uniform: float16 storage, 2 bytes per sample (8 samples = one 16-byte store)
                using H4 = _Float16 __attribute__((ext_vector_type(4)));
                _Float16 *dst = out_tile16 + o_el;
                if (KV >= 4) {
#pragma unroll
                    for (int v = 0; v < KV / 4; ++v) {
                        H4 hv = {(_Float16)res[(4 * v) % KV], (_Float16)res[(4 * v + 1) % KV],
                                 (_Float16)res[(4 * v + 2) % KV], (_Float16)res[(4 * v + 3) % KV]};
                        reinterpret_cast<H4 *>(dst)[v] = hv;
                    }
                } else {
                    dst[0] = (_Float16)res[0];
                }
            } else {
                float *dst = out_tile + o_el;
                if (KV >= 4) {
                    // ordinary stores: the area list visits windows out of order, and partial-line
                    // streaming stores cost 2.3x on the fixed=False shape (L2 no longer merges them)
#pragma unroll
                    for (int v = 0; v < KV / 4; ++v)
                        reinterpret_cast<float4 *>(dst)[v] =
                            make_float4(res[(4 * v) % KV], res[(4 * v + 1) % KV], res[(4 * v + 2) % KV], res[(4 * v + 3) % KV]);
                } else {
                    dst[0] = res[0];
                }
            }
        }
    };

    // Area-sampled windows: output k = float32 mean, in sample order, of the s_area nearest-neighbour samples
    // j = k * s_area ... of the finer grid.  Index of sample j = rint(clip(ia(j))), ia = the rounding sequence
    // of idx with the finer step.  ia is a linear function of j up to ~1e-12 beams, so floor(ia + 0.5) is
    // tracked in 32.32 fixed point, q(j) = q(0) + j * dq: the beam is the high word.  Truncating dq costs at most
    // j * 2^-32 < 2^-21 beam over a window (P * s_area < 2^11, guarded below), so an output with a sample whose
    // low word comes within 2^-20 beam of a tie -- and every window that leaves the field of view -- is summed
    // again with the exact sequence: the indices are the reference's in every case.
    // Round 3: one call = `k_count` consecutive outputs of ONE window from k_begin on.  The window's parameters
    // are read once per call, the per-output float64 setup (fma, floor, two conversions) is one 64-bit add, the
    // mean is a three-instruction correctly rounded division by the launch-constant s_area
    // (tools/divconst32_check.c), and the outputs leave as 16-byte stores (round 2 stored every output on its
    // own: one line transaction per output, 0.26 ms of memory-pipe time on the dense shape).
    const float area_fs = (float)max(s_area, 1), area_rs = __fdiv_rn(1.0f, area_fs);
    const bool area_small_div = s_area <= 64;                   // the range divconst32_check.c covers
    auto area_outputs = [&](auto lds_tag, const int p, const int k_begin, const int k_count) {
        auto fetch = [&](int off) -> float { return fetch_from(lds_tag, off); };
        const double a0 = wt.a0[p], step_a = wt.step_a[p];
        const double dd = PAIR ? wt.dd[p] * a.rdepth : wt.dd[p];
        const float ylo = wt.ylo[p], yhi = wt.yhi[p], ypad = wt.ypad[p];
        const int out_off = wt.out_off[p], row_off = wt.row_off[p];
        const int kr = wt.krange[p];
        const int klo = kr & 0xffff, khi = kr >> 16;
        const double area_c1 = step_a * rdphi;
        const double area_c0 = (a0 - phi0) * rdphi + 0.5;
        // windows that stick out of the field of view, huge N and non-finite windows take the exact loop
        const bool area_fast = (kr == ((P - 1) << 16)) && N < (1 << 30) && area_c1 >= 0.0 && area_c1 < 1024.0 &&
                               area_c0 >= 0.0 && PA < (1 << 11);
        unsigned long long dq = 0, dk = 0, qk0 = 0;
        if (area_fast) {
            const double fl = floor(area_c0);
            const unsigned long long q0 =
                ((unsigned long long)(unsigned)(int)fl << 32) | (unsigned long long)(unsigned)((area_c0 - fl) * 4294967296.0);
            dq = (unsigned long long)(area_c1 * 4294967296.0);
            dk = dq * (unsigned long long)s_area;                // per output
            qk0 = q0 + dk * (unsigned long long)(unsigned)k_begin;
        }
        constexpr int VS = (P4 >= 0) ? 4 : 1;                    // outputs per store
        for (int tt = 0; tt < tcount; ++tt) {
            const int roff = (row_off + tt) * rstride + rbase;
            unsigned long long qk = qk0;
            for (int kb = k_begin; kb < k_begin + k_count; kb += VS) {
                float yv[VS];
#pragma unroll
                for (int u = 0; u < VS; ++u) {
                    const int k = kb + u;
                    float acc = 0.0f;
                    bool exact = !area_fast;
                    if (area_fast) {
                        unsigned long long q = qk;
                        unsigned tie = 0xffffffffu;      // min over the samples of (low word + 2^12) mod 2^32
                        int sdone = 0;
                        for (; sdone + 4 <= s_area; sdone += 4) {
                            const unsigned long long q1 = q + dq, q2 = q1 + dq, q3 = q2 + dq;
                            const float v0 = fetch(roff + (int)(q >> 32)), v1 = fetch(roff + (int)(q1 >> 32));
                            const float v2 = fetch(roff + (int)(q2 >> 32)), v3 = fetch(roff + (int)(q3 >> 32));
                            tie = min(min(tie, (unsigned)q + 4096u), min((unsigned)q1 + 4096u, (unsigned)q2 + 4096u));
                            tie = min(tie, (unsigned)q3 + 4096u);
                            acc = (sdone == 0) ? v0 : acc + v0;
                            acc = acc + v1;
                            acc = acc + v2;
                            acc = acc + v3;
                            q = q3 + dq;
                        }
                        for (; sdone < s_area; ++sdone) {
                            tie = min(tie, (unsigned)q + 4096u);
                            const float v = fetch(roff + (int)(q >> 32));
                            acc = (sdone == 0) ? v : acc + v;
                            q += dq;
                        }
                        exact = tie < 8192u;
                        qk += dk;
                    }
                    if (exact) {
                        for (int sidx = 0; sidx < s_area; ++sidx) {
                            double ia = frac_index(a0, step_a, (double)(k * s_area + sidx), phi0, dphi, rdphi);
                            ia = ia < 0.0 ? 0.0 : ia;
                            ia = ia > nm1 ? nm1 : ia;
                            const int ri = (int)rint(ia);
                            const float v = fetch(roff + min(max(ri, 0), N - 1));
                            acc = (sidx == 0) ? v : acc + v;
                        }
                    }
                    float mean_a;
                    if (area_small_div) {   // RN(acc / s_area): q0 = acc * RN(1/s), one exact residual, one correction
                        const float m0 = acc * area_rs;
                        mean_a = fmaf(fmaf(-m0, area_fs, acc), area_rs, m0);
                    } else {
                        mean_a = __fdiv_rn(acc, area_fs);
                    }
                    float y;
                    if (PAIR) {
                        y = (float)((double)mean_a - dd);
                    } else if (VMODE == 2) {
                        const float df = (float)dd;
                        y = a.centered ? (a.depth_pow2 ? (mean_a - df) * a.rdepth_f32 : __fdiv_rn(mean_a - df, a.depth_f32))
                                       : mean_a;
                    } else {
                        y = finish_value<VMODE>(a, (double)mean_a, dd);
                    }
                    y = __builtin_amdgcn_fmed3f(y, ylo, yhi);
                    if (k < klo || k > khi) y = ypad;
                    if (DBG) {
                        const double idx = frac_index(a0, wt.step[p], (double)k, phi0, dphi, rdphi);
                        const int jj = a.fixed ? p / T : p, tfirst = a.fixed ? p - jj * T : 0;
                        a.dbg_lo[(((long long)b * P + k) * T + tfirst + tt) * a.Ns + (j0 + jj)] = min(max((int)idx, 0), N - 1);
                    }
                    yv[u] = y;
                }
                const int o_el = out_off + tt * P + kb;
                if (VS == 4) {
                    if (out_tile16) {
                        using H4 = _Float16 __attribute__((ext_vector_type(4)));
                        H4 hv = {(_Float16)yv[0], (_Float16)yv[1 % VS], (_Float16)yv[2 % VS], (_Float16)yv[3 % VS]};
                        *reinterpret_cast<H4 *>(out_tile16 + o_el) = hv;
                    } else {
                        *reinterpret_cast<float4 *>(out_tile + o_el) = make_float4(yv[0], yv[1 % VS], yv[2 % VS], yv[3 % VS]);
                    }
                } else {
                    if (out_tile16) out_tile16[o_el] = (_Float16)yv[0];
                    else out_tile[o_el] = yv[0];
                }
            }
        }
    };

    // B1: every window that is not area-sampled.  The lanes of a wave whose samples all lie inside the field of
    // view (all but the windows at the two ends of the scan) take the variant without range tests.
    for (int g0 = 0; g0 < total; g0 += kThreads) {
        const int g = g0 + threadIdx.x;
        const bool act = g < total;
        const int p = act ? ((P4 > 0) ? g / P4 : g / per_win) : 0;
        const int k0 = (g - p * per_win) * KV;
        const bool work = act && !wt.isarea[p];
        const int kr = wt.krange[p];
        const bool full = (kr & 0xffff) <= k0 && k0 + KV - 1 <= (kr >> 16);
        if (LDSROWS || (LDSMODE == 2 && span_lds)) {
            if (__all(full || !work)) {
                if (work) group(std::true_type{}, std::true_type{}, p, k0);
            } else {
                if (work) group(std::false_type{}, std::true_type{}, p, k0);
            }
        } else {
            if (work) group(std::false_type{}, std::false_type{}, p, k0);
        }
    }
    // B2: the area-sampled windows.  Where most windows of the tile are area-sampled (the dense 0.1-degree scans:
    // 97 %) every lane takes the window it built in phase A and walks all P outputs: the window's parameters are
    // read once per 56 outputs, and the lanes of a wave read consecutive points' (or scans') windows.  Where they
    // are few (0.5-degree scans: the near field only) the list phase A compacted is spread over the lanes, KV
    // outputs each, so that a handful of windows does not serialise on a handful of lanes.
    const int n_area = s_area > 0 ? s_acount : 0;
    const bool lds_rows = LDSROWS || (LDSMODE == 2 && span_lds);
    if (2 * n_area >= nwin) {
        for (int p = threadIdx.x; p < nwin; p += kThreads) {
            if (!wt.isarea[p]) continue;
            if (lds_rows) area_outputs(std::true_type{}, p, 0, P);
            else area_outputs(std::false_type{}, p, 0, P);
        }
    } else {
        for (int g = threadIdx.x; g < n_area * per_win; g += kThreads) {
            const int q = (P4 > 0) ? g / P4 : g / per_win;
            if (lds_rows) area_outputs(std::true_type{}, wt.alist[q], (g - q * per_win) * KV, KV);
            else area_outputs(std::false_type{}, wt.alist[q], (g - q * per_win) * KV, KV);
        }
    }
}

template <int LDSROWS, int FAST, bool DBG>
void launch_cutout2(const CutArgs &a, dim3 grid, size_t lds, hipStream_t s, bool vec4)
{
    if (!vec4) cutout_kernel<LDSROWS, -1, FAST, DBG><<<grid, kThreads, lds, s>>>(a);
    else if (a.P == 56) cutout_kernel<LDSROWS, 7, FAST, DBG><<<grid, kThreads, lds, s>>>(a);
    else if (a.P == 48) cutout_kernel<LDSROWS, 6, FAST, DBG><<<grid, kThreads, lds, s>>>(a);
    else if (a.P == 32) cutout_kernel<LDSROWS, 4, FAST, DBG><<<grid, kThreads, lds, s>>>(a);
    else cutout_kernel<LDSROWS, 0, FAST, DBG><<<grid, kThreads, lds, s>>>(a);
}

template <int LDSROWS>
void launch_cutout(const CutArgs &a, dim3 grid, size_t lds, hipStream_t s, bool vec4)
{
    if constexpr (LDSROWS == 3) {      // pair rows: the caller has checked centred / power-of-two depth / exact values / no debug
        launch_cutout2<LDSROWS, 1, false>(a, grid, lds, s, vec4);
    } else {
        const bool fast = a.centered && a.depth_pow2;
        if (a.dbg_lo) {
            // test-only variants, no need to specialise further
            if (a.value_mode == 1) launch_cutout2<LDSROWS, 2, true>(a, grid, lds, s, vec4);
            else launch_cutout2<LDSROWS, 0, true>(a, grid, lds, s, vec4);
        } else if (a.value_mode == 1) {
            launch_cutout2<LDSROWS, 2, false>(a, grid, lds, s, vec4);
        } else if (fast) {
            launch_cutout2<LDSROWS, 1, false>(a, grid, lds, s, vec4);
        } else {
            launch_cutout2<LDSROWS, 0, false>(a, grid, lds, s, vec4);
        }
    }
}

}  // namespace

namespace {
// Clears the per-sample area maxima.  A kernel rather than hipMemsetAsync: as a graph node the 4-byte memset of
// a one-sample batch hung the second replay of a captured streaming step on ROCm 7.2 (tools/diag_stream.py).
__global__ void clear_area_kernel(int32_t *s_area, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) s_area[i] = 0;
}

int cutout_launch(const float *scans, int B, int T, int N, const double *tab, int stride, int centered, int fixed,
                  double window_width, double window_depth, int num_cutout_pts, double padding_val, int area_mode,
                  int value_mode, float *out32, _Float16 *out16, int32_t *workspace, int32_t *dbg_lo,
                  pof_stream_t stream)
{
    void *out = out32 ? static_cast<void *>(out32) : static_cast<void *>(out16);
    if (value_mode < 0 || value_mode > 1) return POF_E_BADARG;
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
    a.value_mode = value_mode;
    a.padding_f32 = (float)padding_val;
    a.rdepth_f32 = (float)(1.0 / window_depth);
    a.out = out32; a.out16 = out16; a.s_area = area_mode ? workspace : nullptr; a.dbg_lo = dbg_lo;
    hipStream_t s = pof_stream(stream);
    if (area_mode) {
        // POF_CUTOUT_CLEAR=memset restores round 1's hipMemsetAsync node -- for tools/diag_graph_dump.py only, which
        // captures and dumps the streaming step in that form WITHOUT replaying it
        static const bool memset_form = [] {
            const char *e = std::getenv("POF_CUTOUT_CLEAR");
            return e && std::string(e) == "memset";
        }();
        if (memset_form) {
            if (hipMemsetAsync(workspace, 0, sizeof(int32_t) * (size_t)B, s) != hipSuccess) return POF_E_LAUNCH;
        } else {
            clear_area_kernel<<<(B + 255) / 256, 256, 0, s>>>(workspace, B);
        }
        POF_CHECK_LAUNCH();
        const int windows = (fixed ? T : 1) * a.Ns;
        int chunks = (windows + kThreads - 1) / kThreads;
        // enough workgroups to fill the chip at small B, at most one window per lane
        const int want_chunks = (2048 + B - 1) / B;
        if (chunks > want_chunks) chunks = want_chunks;
        cutout_area_kernel<<<dim3(chunks, B), kThreads, 0, s>>>(a);
        POF_CHECK_LAUNCH();
    }
    const size_t row_bytes = (size_t)T * N * sizeof(float);
    // one window per lane in phase A: tile*T ~ 256 windows when `fixed`, 256 points otherwise
    a.tile = fixed ? (kMaxWin / T > 0 ? kMaxWin / T : 1) : kMaxWin;
    if (a.tile > a.Ns) a.tile = a.Ns;
    const size_t tbl = win_table_bytes(fixed ? a.tile * T : a.tile);
    const bool rows_in_lds = tbl + row_bytes + 16 <= 64 * 1024;
    // otherwise: per-tile span staging with a 48 KB row buffer
    a.span_cap = (int)((48 * 1024) / ((size_t)T * sizeof(float)));
    if (a.span_cap > N) a.span_cap = N;
    const bool span_mode = !rows_in_lds && a.span_cap >= 64;
    const size_t lds = tbl + (rows_in_lds ? row_bytes + 16 : (span_mode ? (size_t)a.span_cap * T * sizeof(float) : 0));
    dim3 grid((a.Ns + a.tile - 1) / a.tile, B);
    // 16-byte float4 stores / 8-byte half4 stores both need P % 4 == 0 and an aligned base
    const bool vec4 = (num_cutout_pts % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // pair rows (see cutout_kernel): the reference's configuration -- centred output, power-of-two depth, exact values
    static const bool pair_off = [] { const char *e = std::getenv("POF_CUTOUT_PAIR"); return e && atoi(e) == 0; }();
    const bool pair = rows_in_lds && centered && a.depth_pow2 && value_mode == 0 && !dbg_lo && !pair_off &&
                      tbl + 2 * row_bytes <= 64 * 1024;
    if (pair) { launch_cutout<3>(a, grid, tbl + 2 * row_bytes, s, vec4); POF_CHECK_LAUNCH(); return POF_OK; }
    if (rows_in_lds) launch_cutout<1>(a, grid, lds, s, vec4);
    else if (span_mode) launch_cutout<2>(a, grid, lds, s, vec4);
    else launch_cutout<0>(a, grid, lds, s, vec4);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
}  // namespace

extern "C" int pof_cutout_ex(const float *scans, int B, int T, int N, const double *tab, int stride,
                             int centered, int fixed, double window_width, double window_depth,
                             int num_cutout_pts, double padding_val, int area_mode, int value_mode,
                             float *out, int32_t *workspace, int32_t *dbg_lo, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return cutout_launch(scans, B, T, N, tab, stride, centered, fixed, window_width, window_depth, num_cutout_pts,
                         padding_val, area_mode, value_mode, out, nullptr, workspace, dbg_lo, stream);
}

extern "C" int pof_cutout_f16(const float *scans, int B, int T, int N, const double *tab, int stride,
                              int centered, int fixed, double window_width, double window_depth,
                              int num_cutout_pts, double padding_val, int area_mode, int value_mode,
                              void *out_f16, int32_t *workspace, int32_t *dbg_lo, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return cutout_launch(scans, B, T, N, tab, stride, centered, fixed, window_width, window_depth, num_cutout_pts,
                         padding_val, area_mode, value_mode, nullptr, static_cast<_Float16 *>(out_f16), workspace,
                         dbg_lo, stream);
}

extern "C" int pof_cutout(const float *scans, int B, int T, int N, const double *tab, int stride,
                          int centered, int fixed, double window_width, double window_depth,
                          int num_cutout_pts, double padding_val, int area_mode, float *out,
                          int32_t *workspace, int32_t *dbg_lo, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return pof_cutout_ex(scans, B, T, N, tab, stride, centered, fixed, window_width, window_depth, num_cutout_pts,
                         padding_val, area_mode, 0, out, workspace, dbg_lo, stream);
}
