// A1-A7: angle table, fused scan -> xy -> rigid-motion flow -> canonical frame,
// detection association and point masks.  One launch per batch.
//
// Layout in HBM
//   ranges   float32, one row of N per sample (row b at ranges + b*sample_stride)
//   tab      float64 [3N]: phi[N], then (cos, sin)[N] interleaved
//   outputs  batched, C-contiguous: xy/flow [B][N][2], cls/closest [B][N] int64,
//            reg [B][N][2] float32, masks [B][N] float32
// Roofline: HBM.  Algorithmic bytes per scan (N = 450, float32 flow only):
//   4N read + 8N written = 5 400 B; with association + masks 4N + (8+8+8+4)N.
// Arithmetic is float64 (about 12 flop/point for the flow, 10 per detection for
// the association): two orders of magnitude below the float64 vector peak at
// the HBM rate, so nothing here is worth MFMA.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "pof_common.h"

namespace {

constexpr int kThreads = 256;

__global__ void laser_phi_kernel(double start, double stop, double step, int n, double *tab)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // numpy.linspace: arange(n) * step + start, last element forced to `stop`
    double p = (double)i * step;
    p = p + start;
    if (i == n - 1 && n > 1) p = stop;
    double s, c;
    sincos(p, &s, &c);
    tab[i] = p;
    tab[n + 2 * i] = c;
    tab[n + 2 * i + 1] = s;
}

struct PreArgs {
    const float *ranges;
    long long sample_stride;
    int B, N, D;
    const double *tab;
    const double *odom0, *odom1;
    int flow_kind, canonical;
    void *xy, *flow;
    const int32_t *det_offsets;
    const double *det_rphi;
    const uint8_t *det_cls;
    double ra0, ra1, ra2;   // association radius per class
    double sd0, sd1, sd2;   // max squared distance with RN(sqrt(s)) <= dynamic-mask radius
    double sa0, sa1, sa2;   // max squared distance with RN(sqrt(s)) <  association radius
    int32_t lb0, lb1, lb2;  // label per class
    int64_t *closest, *target_cls;
    float *target_reg, *dyn_mask, *valid_mask, *exclude_mask;
    double *ws_rec;         // [B][kRecStride]: motion[7], detection count, kInline inline detections
    double *ws_det;         // [D][6]: cx, cy, assoc radius, dyn s-threshold, label, assoc s-threshold
};

constexpr int kDetStride = 6;
constexpr int kInline = 8;   // detections stored inline in the per-sample record
// per-sample record (doubles): [0,8) motion[7] + detection count; [8, 56) kInline detections x kDetStride;
// then the float32 companions the flat kernel stages without arithmetic: [56, 60) = 8 floats of motion,
// [60, 84) = kInline x 6 floats {cx, cy, Dlo, Dhi, A, pad} (the screen of screen_entry below)
constexpr int kRecF32Mot = 8 + kInline * kDetStride;
constexpr int kRecF32Det = kRecF32Mot + 4;
constexpr int kRecStride = kRecF32Det + kInline * 3;

// Per-sample rigid motion from the two (sin, cos) pairs the params kernel
// evaluates on two lanes.  mot[0..3] = 2x2 matrix (row major), mot[4..5] =
// translation, mot[6] = dphi.
//   kind 0: (sA,cA) = sincos(phi0), (sB,cB) = sincos(phi1)
//   kind 1: (sA,cA) = sincos(phi0), (sB,cB) = sincos(phi1 - phi0)
//   kind 2: (sB,cB) = sincos(phi1)
__device__ void motion_params(int kind, const double *o0, const double *o1, double sA, double cA,
                              double sB, double cB, double *mot)
{
    const double tx = o1[0] - o0[0], ty = o1[1] - o0[1];
    if (kind == 0) {
        // get_displacement_from_odometry: R0, R1 stored in float32; the float32
        // 2x2 product and the float64 products below use the FMA order of the
        // BLAS the reference calls (sgemm/dgemm/gemv: fma(a1, b1, a0*b0)).
        const float c0 = (float)cA, s0 = (float)sA, c1 = (float)cB, s1 = (float)sB;
        // A = R0^T = [[c0, s0], [-s0, c0]],  R1 = [[c1, -s1], [s1, c1]]
        const float p00 = fmaf(s0, s1, c0 * c1);
        const float p01 = fmaf(s0, c1, c0 * (-s1));
        const float p10 = fmaf(c0, s1, (-s0) * c1);
        const float p11 = fmaf(c0, c1, (-s0) * (-s1));
        mot[0] = 1.0 - (double)p00;
        mot[1] = 0.0 - (double)p01;
        mot[2] = 0.0 - (double)p10;
        mot[3] = 1.0 - (double)p11;
        mot[4] = fma((double)c0, tx, (double)s0 * ty);
        mot[5] = fma((double)(-s0), tx, (double)c0 * ty);
        mot[6] = 0.0;
    } else if (kind == 1) {
        // get_flow_target: float64 throughout
        mot[0] = cB; mot[1] = -sB; mot[2] = sB; mot[3] = cB;
        // trans_world @ rot_0.T
        mot[4] = fma(ty, -sA, tx * cA);
        mot[5] = fma(ty, cA, tx * sA);
        mot[6] = 0.0;
    } else if (kind == 2) {
        // get_velocity_from_odometry: float32 R1, cross matrix dphi*[[0,-1],[1,0]]
        const float c1 = (float)cB, s1 = (float)sB;
        mot[0] = mot[1] = mot[2] = mot[3] = 0.0;
        mot[4] = fma((double)c1, tx, (double)s1 * ty);
        mot[5] = fma((double)(-s1), tx, (double)c1 * ty);
        mot[6] = o1[2] - o0[2];
    } else if (kind == 3) {
        // bin/data_prepare.get_flow_target (:29-47): o0 = odometry difference (dx, dy, dphi),
        // o1[0] = dt.  v = dxy/(dt+reg), w = dphi/(dt+reg); flow = (w x r + v) * dt
        const double dt = o1[0], den = dt + 1e-6;
        mot[0] = o0[2] / den;   // w
        mot[1] = o0[0] / den;   // vx
        mot[2] = o0[1] / den;   // vy
        mot[3] = dt;
        mot[4] = mot[5] = mot[6] = 0.0;
    } else {
        // scan-pair alignment (src/utils/dataset.py:76-93): o0 = (dx, dy, dphi), o1[0] = scan_dir
        // (sA,cA) = sincos(dphi), (sB,cB) = sincos(scan_dir); float32 matrices
        const float c = (float)cA, sn = (float)sA, cd = (float)cB, sd = (float)sB;
        mot[0] = (double)c; mot[1] = (double)sn; mot[2] = (double)(-sn); mot[3] = (double)c;
        mot[4] = fma(o0[1], (double)(-sd), o0[0] * (double)cd);
        mot[5] = fma(o0[1], (double)cd, o0[0] * (double)sd);
        mot[6] = 0.0;
    }
}

// angle whose sincos lane `which` (0/1) of a sample evaluates
__device__ __forceinline__ double motion_angle(int kind, int which, const double *o0, const double *o1)
{
    if (kind == 1) return which ? o1[2] - o0[2] : o0[2];
    if (kind == 4) return which ? o1[0] : o0[2];
    return which ? o1[2] : o0[2];
}

// per-point evaluation shared by the streaming and the xy-input kernels
__device__ __forceinline__ void apply_motion(int kind, double px, double py, double m0, double m1, double m2,
                                             double m3, double t0, double t1, double dph, double &fx,
                                             double &fy)
{
    if (kind == 0) {
        fx = fma(py, m1, px * m0) - t0;
        fy = fma(py, m3, px * m2) - t1;
    } else if (kind == 1) {
        fx = (fma(py, m1, px * m0) - t0) - px;
        fy = (fma(py, m3, px * m2) - t1) - py;
    } else if (kind == 2) {
        // -lin - xy @ cross^T, cross^T = [[0, dphi], [-dphi, 0]]
        fx = -t0 - (py * -dph);
        fy = -t1 - (px * dph);
    } else if (kind == 3) {
        // np.cross([0,0,w],[x,y,0])[:2] = (0*0 - w*y, w*x - 0*0); + v; * dt
        fx = ((0.0 - m0 * py) + m1) * m3;
        fy = ((m0 * px - 0.0) + m2) * m3;
    } else {
        fx = fma(py, m1, px * m0) + t0;
        fy = fma(py, m3, px * m2) + t1;
    }
}

// Launch 1 (tiny): everything that needs a transcendental and is shared by all
// points of a sample goes to the workspace, so that the streaming kernel below
// is transcendental free and needs ONE memory round trip:
//   ws_rec[b] = { rigid motion (7), detection count, first kInline detections }
//   ws_det[g] = every detection (CSR order), read only by samples with more
//               than kInline detections.
// detection entry = { cx, cy, assoc radius, dyn s-threshold, label, assoc s-threshold }
__device__ __forceinline__ void det_entry(const PreArgs &a, int g, double *w)
{
    const double dr = a.det_rphi[2 * g], dp = a.det_rphi[2 * g + 1];
    double s, c;
    sincos(dp, &s, &c);
    const int cl = a.det_cls[g];
    w[0] = dr * c;
    w[1] = dr * s;
    w[2] = cl == 0 ? a.ra0 : (cl == 1 ? a.ra1 : a.ra2);
    w[3] = cl == 0 ? a.sd0 : (cl == 1 ? a.sd1 : a.sd2);   // dist <= dyn radius   <=>  s <= w[3]
    w[4] = (double)(cl == 0 ? a.lb0 : (cl == 1 ? a.lb1 : a.lb2));
    w[5] = cl == 0 ? a.sa0 : (cl == 1 ? a.sa1 : a.sa2);   // dist <  assoc radius <=>  s <= w[5]
}

// float32 screen of one detection against float32 squared distances s2f.  With all coordinates below
// 100 m, |s2f - s2| <= 1e-3 + 1e-4 * s2 (s2 = the float64 squared distance the reference's decision is
// made on; derivation in DESIGN.md section 3.1), hence
//   s2f <= Dlo = sd (1 - 1e-4) - 1e-3   =>  s2 <= sd   (surely inside the dynamic-mask radius)
//   s2f >  Dhi = sd (1 + 1e-4) + 1e-3   =>  s2 >  sd   (surely outside)
//   s2f >  A   = sa (1 + 1e-4) + 1e-3   =>  s2 >  sa   (surely not an association candidate)
// Dlo is rounded down, Dhi and A up.  A far detection gets (-inf, +inf, +inf): never decided here.
__device__ __forceinline__ float round_dn(double v)
{
    float f = (float)v;
    if ((double)f > v) f = __uint_as_float(__float_as_uint(f) + (f > 0.0f ? -1 : 1));
    return f;
}
__device__ __forceinline__ float round_up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1 : -1));
    return f;
}
__device__ __forceinline__ void screen_entry(double cx, double cy, double sd, double sa, float4 *e, float *ea)
{
    const bool far = !(fabs(cx) + fabs(cy) < 100.0);
    const float dlo = round_dn(sd * (1.0 - 1e-4) - 1e-3), dhi = round_up(sd * (1.0 + 1e-4) + 1e-3);
    const float aa = round_up(sa * (1.0 + 1e-4) + 1e-3);
    *e = make_float4((float)cx, (float)cy, far ? -INFINITY : dlo, far ? INFINITY : dhi);
    *ea = far ? INFINITY : aa;
}

// number of params jobs of a batch: 2 per sample (the two motion angles) + kInline
// detection slots per sample
__host__ __device__ inline int params_job_count(int B, bool have_dets)
{
    return 2 * B + (have_dets ? B * kInline : 0);
}

// job t of the params work: one sincos per lane.  Jobs [0, 2B) = (sample, angle)
// pairs; then B*kInline detection slots: slot s of a sample evaluates detections
// s, s+kInline, ... (the first goes into the inline record, and samples with more
// than kInline detections additionally get all their rows in the CSR table).
__device__ __forceinline__ void params_work(const PreArgs &a, int t)
{
    const int nm = 2 * a.B;
    if (t < nm) {
        const int b = t >> 1, which = t & 1;
        const double *o0 = a.odom0 + 3 * b, *o1 = a.odom1 + 3 * b;
        double s = 0.0, c = 1.0;
        if (a.flow) {
            const double ang = motion_angle(a.flow_kind, which, o0, o1);
            sincos(ang, &s, &c);
        }
        // partner lane (t ^ 1) holds the other angle of the same sample
        const double s_o = __shfl_xor(s, 1, 64), c_o = __shfl_xor(c, 1, 64);
        if (which == 0) {
            double *rec = a.ws_rec + (long long)b * kRecStride;
            if (a.flow) {
                motion_params(a.flow_kind, o0, o1, s, c, s_o, c_o, rec);
                float *mf = reinterpret_cast<float *>(rec + kRecF32Mot);
#pragma unroll
                for (int k = 0; k < 7; ++k) mf[k] = (float)rec[k];
            }
            rec[7] = a.det_offsets ? (double)(a.det_offsets[b + 1] - a.det_offsets[b]) : 0.0;
        }
    } else if (a.det_offsets && t < nm + a.B * kInline) {
        const int u = t - nm;
        const int b = u / kInline, slot = u - b * kInline;
        const int d0 = a.det_offsets[b], cnt = a.det_offsets[b + 1] - d0;
        for (int idx = slot; idx < cnt; idx += kInline) {
            double w[kDetStride];
            det_entry(a, d0 + idx, w);
            if (idx < kInline) {
                double *r = a.ws_rec + (long long)b * kRecStride + 8 + idx * kDetStride;
#pragma unroll
                for (int c = 0; c < kDetStride; ++c) r[c] = w[c];
                float4 e;
                float ea;
                screen_entry(w[0], w[1], w[3], w[5], &e, &ea);
                float *f = reinterpret_cast<float *>(a.ws_rec + (long long)b * kRecStride + kRecF32Det) + idx * 6;
                f[0] = e.x; f[1] = e.y; f[2] = e.z; f[3] = e.w; f[4] = ea; f[5] = 0.0f;
            }
            if (cnt > kInline) {
                double *r = a.ws_det + (long long)(d0 + idx) * kDetStride;
#pragma unroll
                for (int c = 0; c < kDetStride; ++c) r[c] = w[c];
            }
        }
    }
}

__global__ __launch_bounds__(256) void scan_params_kernel(PreArgs a)
{
    params_work(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// A value that is the same in every lane, moved to scalar registers.
__device__ __forceinline__ double to_sgpr(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Outputs are written once and never re-read by this kernel: streaming (non-temporal)
// stores keep them from displacing the tables and records in L2 (13.3 vs 16.3 us for the
// bare I/O shape of the headline launch, tools/ubench/stream_shape.hip).
using F4V = float __attribute__((ext_vector_type(4)));
using F2V = float __attribute__((ext_vector_type(2)));
using LL2V = long long __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_f4(float *base, long long idx4, float x, float y, float z, float w)
{
    F4V v = {x, y, z, w};
    __builtin_nontemporal_store(v, reinterpret_cast<F4V *>(base) + idx4);
}
__device__ __forceinline__ void stream_f2(float *base, long long idx2, float x, float y)
{
    F2V v = {x, y};
    __builtin_nontemporal_store(v, reinterpret_cast<F2V *>(base) + idx2);
}
__device__ __forceinline__ void stream_ll2(long long *base, long long idx2, long long x, long long y)
{
    LL2V v = {x, y};
    __builtin_nontemporal_store(v, reinterpret_cast<LL2V *>(base) + idx2);
}

template <typename OutT>
__device__ __forceinline__ void store2(OutT *base, long long idx, double a, double b)
{
    using V = typename std::conditional<sizeof(OutT) == 4, float2, double2>::type;
    V v;
    v.x = (OutT)a;
    v.y = (OutT)b;
    reinterpret_cast<V *>(base)[idx] = v;
}

// Launch 2: one workgroup per (SPB samples, 512-point chunk).  Each lane owns
// the same PTS point indices in SPB consecutive samples, so the cos/sin entries
// are loaded once and SPB range rows are in flight together.  Everything a
// workgroup needs is requested up front (range rows, table, motions,
// detections -> LDS): one memory round trip, then the association loop runs out
// of LDS.  The launcher uses SPB = 1 (see launch_preprocess).  PTS == 2: 16-byte
// streaming stores throughout.
constexpr int kDetTile = 32;

// float32 copy of a detection for the prefilter.  With |coordinates| < 100 m the
// float32 squared distance is within 4.8e-5*sqrt(s) < 1e-3 + 1e-4*s of the
// float64 one, so "s2f - mgn >= thr" / "s2f + mgn < thr" decide the float64
// comparison; anything closer to a threshold, any far detection (thresholds set
// to +inf / force exact) and every point inside the association disc take the
// exact float64 path.
__device__ __forceinline__ float4 prefilter_entry(double cx, double cy, double sd, double sa)
{
    const bool far = !(fabs(cx) + fabs(cy) < 100.0);
    return make_float4((float)cx, (float)cy, (float)sd, far ? INFINITY : (float)sa);
}

template <typename OutT, int PTS, int SPB>
__device__ __forceinline__ void scan_main(const PreArgs &a, const int block_y)
{
    __shared__ double s_det[SPB][kDetTile][5];  // cx, cy, assoc radius, dyn s-threshold, assoc s-threshold
    __shared__ int s_lab[SPB][kDetTile];
    __shared__ float4 s_detf[SPB][kDetTile];    // float32 prefilter copy: cx, cy, dyn thr, assoc thr

    const int b0 = block_y * SPB;
    const int tid = threadIdx.x;
    const int i0 = (blockIdx.x * kThreads + tid) * PTS;
    const int N = a.N;
    const bool live = i0 < N;  // N % PTS == 0 is guaranteed by the launcher
    const bool want_flow = a.flow != nullptr;
    const bool want_assoc = a.det_offsets != nullptr;

    // ---- issue every load first ---------------------------------------------
    int ndet[SPB];
    float r[SPB][PTS];
    double mot[SPB][7];
#pragma unroll
    for (int q = 0; q < SPB; ++q) {
        const int b = min(b0 + q, a.B - 1);  // tail workgroup: clamp, stores are guarded
        const float *row = a.ranges + (long long)b * a.sample_stride;
#pragma unroll
        for (int k = 0; k < PTS; ++k) r[q][k] = 0.0f;
        if (live) {
            if (PTS == 2) {
                float2 v = *reinterpret_cast<const float2 *>(row + i0);
                r[q][0] = v.x;
                r[q][PTS - 1] = v.y;
            } else {
                r[q][0] = row[i0];
            }
        }
#pragma unroll
        for (int c = 0; c < 7; ++c) mot[q][c] = 0.0;
        ndet[q] = 0;
        if (want_flow || want_assoc) {
            // wave-uniform record written by the previous launch: constant address
            // space -> scalar loads, the record lives in SGPRs
            typedef const __attribute__((address_space(4))) double *cptr;
            cptr m = (cptr)(a.ws_rec + (long long)b * kRecStride);
#pragma unroll
            for (int c = 0; c < 7; ++c) mot[q][c] = m[c];
            ndet[q] = (int)m[7];
        }
    }
    double cs[PTS], sn[PTS];
#pragma unroll
    for (int k = 0; k < PTS; ++k) cs[k] = sn[k] = 0.0;
    if (live) {
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            double2 t = *reinterpret_cast<const double2 *>(a.tab + N + 2 * (i0 + k));
            cs[k] = t.x;
            sn[k] = t.y;
        }
    }
    if (want_assoc) {
        // inline detections of every sample -> LDS; the address depends on the sample
        // index only, so these loads fly together with the range rows (lanes that map
        // to slots past the sample's count stage unused values)
        const int qsel = tid / kInline, j = tid - qsel * kInline;
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
            if (qsel == q) {
                const int b = min(b0 + q, a.B - 1);
                const double *w = a.ws_rec + (long long)b * kRecStride + 8 + j * kDetStride;
                s_det[q][j][0] = w[0];
                s_det[q][j][1] = w[1];
                s_det[q][j][2] = w[2];
                s_det[q][j][3] = w[3];
                s_det[q][j][4] = w[5];
                s_lab[q][j] = (int)w[4];
                s_detf[q][j] = prefilter_entry(w[0], w[1], w[3], w[5]);
            }
        }
    }
    bool synced = false;

#pragma unroll
    for (int q = 0; q < SPB; ++q) {
        const int b = b0 + q;
        const bool ok = live && b < a.B;
        double px[PTS], py[PTS];
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            px[k] = (double)r[q][k] * cs[k];
            py[k] = (double)r[q][k] * sn[k];
        }
        float pxf[PTS], pyf[PTS];
        bool far[PTS];
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            pxf[k] = (float)px[k];
            pyf[k] = (float)py[k];
            far[k] = !(fabsf(pxf[k]) + fabsf(pyf[k]) < 100.0f);  // margin analysis assumes < 100 m (NaN -> exact)
        }
        const long long o = (long long)b * N + i0;  // flat point index of the first owned point

        if (ok && a.xy) {
            OutT *xy = static_cast<OutT *>(a.xy);
#pragma unroll
            for (int k = 0; k < PTS; ++k) store2<OutT>(xy, o + k, px[k], py[k]);
        }

        // ---- rigid-motion flow ---------------------------------------------
        if (ok && want_flow) {
            OutT *fl = static_cast<OutT *>(a.flow);
            const double m0 = mot[q][0], m1 = mot[q][1], m2 = mot[q][2], m3 = mot[q][3];
            const double t0 = mot[q][4], t1 = mot[q][5], dph = mot[q][6];
            double fxs[PTS], fys[PTS];
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                double fx, fy;
                apply_motion(a.flow_kind, px[k], py[k], m0, m1, m2, m3, t0, t1, dph, fx, fy);
                if (a.canonical) {
                    // einsum('ijk,ik->ij'): c*fx + (-s)*fy ; s*fx + c*fy, no fusion
                    double gx = cs[k] * fx + (-sn[k]) * fy;
                    double gy = sn[k] * fx + cs[k] * fy;
                    fx = gx;
                    fy = gy;
                }
                fxs[k] = fx;
                fys[k] = fy;
            }
            if (PTS == 2 && sizeof(OutT) == 4) {
                stream_f4(reinterpret_cast<float *>(fl), o / 2, (float)fxs[0], (float)fys[0], (float)fxs[PTS - 1],
                          (float)fys[PTS - 1]);
            } else {
#pragma unroll
                for (int k = 0; k < PTS; ++k) store2<OutT>(fl, o + k, fxs[k], fys[k]);
            }
        }

        float vmask[PTS], dmask[PTS];
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            vmask[k] = (r[q][k] >= 20.0f) ? 0.0f : 1.0f;
            dmask[k] = 1.0f;
        }

        // ---- association + dynamic mask ------------------------------------
        if (want_assoc) {
            double best[PTS];
            int bidx[PTS];
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                best[k] = 0.0;  // the prepended zero column
                bidx[k] = 0;
            }
            if (!synced) {
                __syncthreads();  // first tiles of all SPB samples are in LDS
                synced = true;
            }
            // tile 0 = the inline detections; further tiles (crowded samples only) come
            // from the CSR table through the same LDS buffer
            int dfirst = 0;
            if (ndet[q] > kInline) dfirst = a.det_offsets[min(b, a.B - 1)];
            for (int base = 0; base < ndet[q]; base += (base == 0 ? kInline : kDetTile)) {
                const int cnt = min(base == 0 ? kInline : kDetTile, ndet[q] - base);
                if (base != 0) {
                    __syncthreads();
                    if (tid < cnt) {
                        const double *w = a.ws_det + (long long)(dfirst + base + tid) * kDetStride;
                        s_det[q][tid][0] = w[0];
                        s_det[q][tid][1] = w[1];
                        s_det[q][tid][2] = w[2];
                        s_det[q][tid][3] = w[3];
                        s_det[q][tid][4] = w[5];
                        s_lab[q][tid] = (int)w[4];
                        s_detf[q][tid] = prefilter_entry(w[0], w[1], w[3], w[5]);
                    }
                    __syncthreads();
                }
                float4 fnext = s_detf[q][0];
                for (int j = 0; j < cnt; ++j) {
                    // float32 prefilter with a conservative margin: the exact float64
                    // test below runs only for lanes it cannot classify (inside the
                    // association disc, or within the margin of a threshold)
                    const float4 f = fnext;
                    fnext = s_detf[q][min(j + 1, kDetTile - 1)];  // next entry in flight during this one
                    bool exact[PTS];
                    bool any_exact = false;
#pragma unroll
                    for (int k = 0; k < PTS; ++k) {
                        const float exf = pxf[k] - f.x, eyf = pyf[k] - f.y;
                        const float s2f = fmaf(exf, exf, eyf * eyf);
                        const float mgn = fmaf(s2f, 1e-4f, 1e-3f);
                        exact[k] = far[k] || (s2f - mgn < f.w) || (fabsf(s2f - f.z) <= mgn);
                        if (!exact[k] && s2f < f.z) dmask[k] = 0.0f;   // surely within the dyn radius
                        any_exact |= exact[k];
                    }
                    if (any_exact) {
                        const double cx = s_det[q][j][0], cy = s_det[q][j][1];
                        const double sd = s_det[q][j][3], sa = s_det[q][j][4];
#pragma unroll
                        for (int k = 0; k < PTS; ++k) {
                            if (!exact[k]) continue;
                            const double ex = px[k] - cx, ey = py[k] - cy;
                            const double s2 = ex * ex + ey * ey;   // cdist: (dx*dx) + (dy*dy), no fusion
                            if (s2 <= sd) dmask[k] = 0.0f;         // dist <= dyn radius
                            if (s2 <= sa) {                        // dist < assoc radius: the only case that can win
                                const double v = sqrt(s2) - s_det[q][j][2];
                                if (v < best[k]) {
                                    best[k] = v;
                                    bidx[k] = base + j + 1;
                                }
                            }
                        }
                    }
                }
            }
            if (ok) {
                long long cls[PTS];
                float gx[PTS], gy[PTS];
#pragma unroll
                for (int k = 0; k < PTS; ++k) {
                    cls[k] = 0;
                    gx[k] = gy[k] = 0.0f;
                    if (bidx[k] > 0) {
                        // winner's centre and label: still in LDS unless the sample overflowed
                        // the inline tile (then from the CSR table).  global_to_canonical via the
                        // angle-difference identity: sin(dp-phi)*dr = cy*cos(phi) - cx*sin(phi)
                        double wx, wy;
                        if (ndet[q] <= kInline) {
                            wx = s_det[q][bidx[k] - 1][0];
                            wy = s_det[q][bidx[k] - 1][1];
                            cls[k] = s_lab[q][bidx[k] - 1];
                        } else {
                            const double *w = a.ws_det + (long long)(dfirst + bidx[k] - 1) * kDetStride;
                            wx = w[0];
                            wy = w[1];
                            cls[k] = (long long)w[4];
                        }
                        gx[k] = (float)(wy * cs[k] - wx * sn[k]);
                        gy[k] = (float)((wx * cs[k] + wy * sn[k]) - (double)r[q][k]);
                    }
                }
                if (PTS == 2) {
                    if (a.closest) stream_ll2(reinterpret_cast<long long *>(a.closest), o / 2, bidx[0], bidx[PTS - 1]);
                    if (a.target_cls) stream_ll2(reinterpret_cast<long long *>(a.target_cls), o / 2, cls[0], cls[PTS - 1]);
                    if (a.target_reg) stream_f4(a.target_reg, o / 2, gx[0], gy[0], gx[PTS - 1], gy[PTS - 1]);
                } else {
                    if (a.closest) a.closest[o] = bidx[0];
                    if (a.target_cls) a.target_cls[o] = cls[0];
                    if (a.target_reg) reinterpret_cast<float2 *>(a.target_reg)[o] = make_float2(gx[0], gy[0]);
                }
            }
        }

        if (ok) {
            if (PTS == 2) {
                if (a.dyn_mask) stream_f2(a.dyn_mask, o / 2, dmask[0], dmask[PTS - 1]);
                if (a.valid_mask) stream_f2(a.valid_mask, o / 2, vmask[0], vmask[PTS - 1]);
                if (a.exclude_mask) stream_f2(a.exclude_mask, o / 2, dmask[0] * vmask[0], dmask[PTS - 1] * vmask[PTS - 1]);
            } else {
                if (a.dyn_mask) a.dyn_mask[o] = dmask[0];
                if (a.valid_mask) a.valid_mask[o] = vmask[0];
                if (a.exclude_mask) a.exclude_mask[o] = dmask[0] * vmask[0];
            }
        }
    }
}

// ---- flat form of the streaming rows (the headline launch) ---------------------------------------
// One workgroup chunk = 256 CONSECUTIVE POINTS OF THE FLAT [B*N] AXIS (128 lanes x 2 points), not one
// sample.  A sample's output rows are 8N = 3600 B (N = 450): with one workgroup per sample every row
// boundary splits a 128-byte line between two workgroups on different XCDs, and the bare I/O shape of the
// launch runs at 4.6-4.9 TB/s; with line-aligned 2048-byte (flow / reg / cls) and 1024-byte (mask) chunks
// the same bytes stream at 6.2-6.3 TB/s (tools/ubench/store_ceiling.hip, profiles/r2_store_*).
// A chunk touches at most two samples (N >= 256): their records (motion, inline detections) go through
// LDS, each lane picks its own sample's.  Crowded samples (more than kInline detections) read the rest
// of their detections from the CSR table in the workspace (L2 hits, exact test only).
//
// Grid = (P, rows): chunk = x + P * y with P = N / gcd(N, 256) chunks = the period after which the chunks'
// point-in-scan phases repeat (P * 256 points are a whole number of samples), so that sample and point
// index follow from small per-phase numbers without a 64-bit division (which alone cost more vector
// instructions than the flow arithmetic).  Several chunks per workgroup (records and range rows of K
// chunks requested together) were measured and lost: 15.4 (K = 1) / 16.3 (2) / 19.8 us (4) -- the launch
// is bound by vector-instruction issue, not by the first memory round trip (profiles/r2_headline_pmc.txt).
constexpr int kFlatThreads = 128;

struct FlatSmem {
    double mot[2][8];               // motion[7], detection count
    float motf[2][8];               // float32 copy of the motion (float32-output flow path)
    double det[2][kInline][5];      // cx, cy, assoc radius, dyn s-threshold, assoc s-threshold
    int lab[2][kInline];
    float4 detf[2][kInline + 1];    // float32 screen: cx, cy, Dlo, Dhi  (+1: the loop reads one ahead)
    float deta[2][kInline + 1];     // float32 screen: A
};

template <typename T>
__device__ __forceinline__ T fma_t(T a, T b, T c);
template <>
__device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return fma(a, b, c); }
template <>
__device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return fmaf(a, b, c); }

// apply_motion in either precision (the float64 form is the reference's operation order)
template <typename T>
__device__ __forceinline__ void apply_motion_t(int kind, T px, T py, T m0, T m1, T m2, T m3, T t0, T t1, T dph,
                                               T &fx, T &fy)
{
    if (kind == 0) {
        fx = fma_t<T>(py, m1, px * m0) - t0;
        fy = fma_t<T>(py, m3, px * m2) - t1;
    } else if (kind == 1) {
        fx = (fma_t<T>(py, m1, px * m0) - t0) - px;
        fy = (fma_t<T>(py, m3, px * m2) - t1) - py;
    } else if (kind == 2) {
        fx = -t0 - (py * -dph);
        fy = -t1 - (px * dph);
    } else if (kind == 3) {
        fx = ((T)0 - m0 * py + m1) * m3;
        fy = ((m0 * px - (T)0) + m2) * m3;
    } else {
        fx = fma_t<T>(py, m1, px * m0) + t0;
        fy = fma_t<T>(py, m3, px * m2) + t1;
    }
}

template <typename OutT>
__device__ __forceinline__ void flat_chunk(const PreArgs &a, const FlatSmem &sm, const int q, const int b,
                                           const bool ok, const long long o2, const float2 rv,
                                           const double2 t0v, const double2 t1v, const double *tab_cs)
{
    // float32 outputs: the flow is evaluated in float32 (error ~1e-7 m against the float64 reference at 25 m
    // range; stated bar 1e-5 m in the tests, 1e-4 m in BASELINE.json).  Association, masks and regression
    // targets are decided on the float64 values exactly as before.
    constexpr bool kF32 = sizeof(OutT) == 4;
    const bool want_flow = a.flow != nullptr;
    const bool want_assoc = a.det_offsets != nullptr;
    const float r[2] = {rv.x, rv.y};
    const float csf[2] = {(float)t0v.x, (float)t1v.x}, snf[2] = {(float)t0v.y, (float)t1v.y};
    F2V pxf = {r[0] * csf[0], r[1] * csf[1]}, pyf = {r[0] * snf[0], r[1] * snf[1]};
    // float64 cos / sin are needed on the rare exact paths only: re-read there (L1 / L2 hits) instead of
    // holding 8 registers across the whole body (the 64-register budget of 8 waves per SIMD)
    auto cs64 = [&](int k) { return tab_cs[2 * k]; };
    auto sn64 = [&](int k) { return tab_cs[2 * k + 1]; };
    auto p64x = [&](int k) { return (double)r[k] * cs64(k); };
    auto p64y = [&](int k) { return (double)r[k] * sn64(k); };

    if (ok && a.xy) {
        OutT *xy = static_cast<OutT *>(a.xy);
        store2<OutT>(xy, 2 * o2, p64x(0), p64y(0));
        store2<OutT>(xy, 2 * o2 + 1, p64x(1), p64y(1));
    }

    // ---- rigid-motion flow ---------------------------------------------------
    if (want_flow) {
        if (kF32 && a.flow_kind != 1) {   // kind 1 subtracts the point from its moved image: float64 only
            const float4 ma = *reinterpret_cast<const float4 *>(&sm.motf[q][0]);
            const float4 mb = *reinterpret_cast<const float4 *>(&sm.motf[q][4]);
            float fxs[2], fys[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float fx, fy;
                apply_motion_t<float>(a.flow_kind, pxf[k], pyf[k], ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, fx, fy);
                if (a.canonical) {
                    const float gx = fmaf(csf[k], fx, -snf[k] * fy);
                    const float gy = fmaf(snf[k], fx, csf[k] * fy);
                    fx = gx;
                    fy = gy;
                }
                fxs[k] = fx;
                fys[k] = fy;
            }
            if (ok) stream_f4(reinterpret_cast<float *>(a.flow), o2, fxs[0], fys[0], fxs[1], fys[1]);
        } else {
            // float64 outputs (and float32 kind 1): the reference's operation order in float64
            const double m0 = sm.mot[q][0], m1 = sm.mot[q][1], m2 = sm.mot[q][2], m3 = sm.mot[q][3];
            const double tr0 = sm.mot[q][4], tr1 = sm.mot[q][5], dph = sm.mot[q][6];
            OutT *fl = static_cast<OutT *>(a.flow);
            const double cs[2] = {t0v.x, t1v.x}, sn[2] = {t0v.y, t1v.y};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                double fx, fy;
                apply_motion(a.flow_kind, (double)r[k] * cs[k], (double)r[k] * sn[k], m0, m1, m2, m3, tr0, tr1, dph, fx, fy);
                if (a.canonical) {
                    // einsum('ijk,ik->ij'): c*fx + (-s)*fy ; s*fx + c*fy, no fusion
                    const double gx = cs[k] * fx + (-sn[k]) * fy;
                    const double gy = sn[k] * fx + cs[k] * fy;
                    fx = gx;
                    fy = gy;
                }
                if (ok) store2<OutT>(fl, 2 * o2 + k, fx, fy);
            }
        }
    }

    float vmask[2], dmask[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        vmask[k] = (r[k] >= 20.0f) ? 0.0f : 1.0f;
        dmask[k] = 1.0f;
    }

    // ---- association + dynamic mask ------------------------------------------
    if (want_assoc) {
        // winner so far: 1-based index; 0 = the prepended zero column.  dist - radius of a candidate is
        // negative, so the first candidate beats the zero column without a square root; only a second
        // candidate of the same point (two overlapping discs: rare) needs the two values, and then the
        // incumbent's are recomputed from its index with the same operations.
        int bidx[2] = {0, 0};
        int dfirst = 0;
        auto candidate = [&](int k, int j1, double s2, double ra) {
            if (bidx[k] != 0) {
                const double *w = (bidx[k] <= kInline) ? &sm.det[q][bidx[k] - 1][0]
                                                       : a.ws_det + (long long)(dfirst + bidx[k] - 1) * kDetStride;
                const double exd = p64x(k) - w[0], eyd = p64y(k) - w[1];
                if (!(sqrt(s2) - ra < sqrt(exd * exd + eyd * eyd) - w[2])) return;
            }
            bidx[k] = j1;
        };
        const int nd = ok ? (int)sm.mot[q][7] : 0;
        const int n_in = min(nd, kInline);
        bool far[2], inside[2] = {false, false}, all_out[2] = {true, true};
#pragma unroll
        for (int k = 0; k < 2; ++k) far[k] = !(fabsf(pxf[k]) + fabsf(pyf[k]) < 100.0f);   // NaN -> exact
        float4 fnext = sm.detf[q][0];
        float anext = sm.deta[q][0];
        for (int j = 0; j < n_in; ++j) {
            const float4 f = fnext;
            const float fa = anext;
            fnext = sm.detf[q][j + 1];
            anext = sm.deta[q][j + 1];
            const F2V cx2 = {f.x, f.x}, cy2 = {f.y, f.y};
            const F2V ex = pxf - cx2, ey = pyf - cy2;
            F2V s2f = ey * ey;
            s2f = __builtin_elementwise_fma(ex, ex, s2f);
            bool need = false;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                inside[k] |= s2f[k] <= f.z;
                all_out[k] &= s2f[k] > f.w;
                need |= (s2f[k] <= fa) || far[k];
            }
            if (need) {
                const double cx = sm.det[q][j][0], cy = sm.det[q][j][1], sa = sm.det[q][j][4];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (!((s2f[k] <= fa) || far[k])) continue;
                    const double exd = p64x(k) - cx, eyd = p64y(k) - cy;
                    const double s2 = exd * exd + eyd * eyd;   // cdist: (dx*dx) + (dy*dy), no fusion
                    if (s2 <= sa) candidate(k, j + 1, s2, sm.det[q][j][2]);   // dist < assoc radius
                }
            }
        }
        // dynamic mask: decided by the screen unless a point sits within the error band of a threshold
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (inside[k] && !far[k]) {
                dmask[k] = 0.0f;
            } else if (far[k] || !all_out[k]) {
                const double pxd = p64x(k), pyd = p64y(k);
                for (int j = 0; j < n_in; ++j) {
                    const double exd = pxd - sm.det[q][j][0], eyd = pyd - sm.det[q][j][1];
                    if (exd * exd + eyd * eyd <= sm.det[q][j][3]) dmask[k] = 0.0f;   // dist <= dyn radius
                }
            }
        }
        if (nd > kInline) {
            // crowded sample: detections kInline.. from the CSR table (every lane of the sample reads the
            // same rows: L1 / L2 hits), exact test only
            dfirst = a.det_offsets[b];
            for (int j = kInline; j < nd; ++j) {
                const double *w = a.ws_det + (long long)(dfirst + j) * kDetStride;
                const double cx = w[0], cy = w[1], ra = w[2], sd = w[3], sa = w[5];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double exd = p64x(k) - cx, eyd = p64y(k) - cy;
                    const double s2 = exd * exd + eyd * eyd;
                    if (s2 <= sd) dmask[k] = 0.0f;
                    if (s2 <= sa) candidate(k, j + 1, s2, ra);
                }
            }
        }
        if (ok) {
            long long cls[2];
            float gx[2], gy[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                cls[k] = 0;
                gx[k] = gy[k] = 0.0f;
                if (bidx[k] > 0) {
                    // winner's centre and label.  global_to_canonical via the angle-difference identity:
                    // sin(dp-phi)*dr = cy*cos(phi) - cx*sin(phi)
                    double wx, wy;
                    if (bidx[k] <= kInline) {
                        wx = sm.det[q][bidx[k] - 1][0];
                        wy = sm.det[q][bidx[k] - 1][1];
                        cls[k] = sm.lab[q][bidx[k] - 1];
                    } else {
                        const double *w = a.ws_det + (long long)(dfirst + bidx[k] - 1) * kDetStride;
                        wx = w[0];
                        wy = w[1];
                        cls[k] = (long long)w[4];
                    }
                    const double c = cs64(k), sn_ = sn64(k);
                    gx[k] = (float)(wy * c - wx * sn_);
                    gy[k] = (float)((wx * c + wy * sn_) - (double)r[k]);
                }
            }
            if (a.closest) stream_ll2(reinterpret_cast<long long *>(a.closest), o2, bidx[0], bidx[1]);
            if (a.target_cls) stream_ll2(reinterpret_cast<long long *>(a.target_cls), o2, cls[0], cls[1]);
            if (a.target_reg) stream_f4(a.target_reg, o2, gx[0], gy[0], gx[1], gy[1]);
        }
    }

    if (ok) {
        if (a.dyn_mask) stream_f2(a.dyn_mask, o2, dmask[0], dmask[1]);
        if (a.valid_mask) stream_f2(a.valid_mask, o2, vmask[0], vmask[1]);
        if (a.exclude_mask) stream_f2(a.exclude_mask, o2, dmask[0] * vmask[0], dmask[1] * vmask[1]);
    }
}

template <typename OutT>
__device__ __forceinline__ void scan_flat(const PreArgs &a, const int phase, const int row, const int period,
                                          const float inv_half_n)
{
    __shared__ FlatSmem sm;

    const int tid = threadIdx.x;
    const int N = a.N, halfN = N >> 1;
    // chunk = phase + period * row.  period * 256 points are a whole number of samples (sstep), so the sample
    // and the point-in-scan index come from the phase alone -- no wide division: x < 2^24 is exact in float32
    const unsigned x = (unsigned)phase * kFlatThreads;            // pairs since the row's first sample
    unsigned sp = (unsigned)((float)x * inv_half_n);
    if (sp * (unsigned)halfN > x) --sp;
    else if ((sp + 1) * (unsigned)halfN <= x) ++sp;
    const int sstep = period * kFlatThreads / halfN;              // samples per row of chunks
    const int sA = row * sstep + (int)sp;                         // workgroup-uniform
    int local = (int)(x - sp * (unsigned)halfN) + tid;
    const int q = local >= halfN ? 1 : 0;                         // kFlatThreads <= halfN: at most one wrap
    local -= q ? halfN : 0;
    const int i0 = 2 * local;
    const int b = sA + q;
    const bool ok = b < a.B;
    const long long o2 = (long long)b * halfN + local;            // flat pair index: point o = 2 * o2

    // ---- issue every load first ---------------------------------------------
    // unconditional (clamped) load: a load inside `if (ok)` makes the compiler wait for it at the end of the
    // branch, i.e. BEFORE the table and record loads are even issued -- two memory round trips instead of one
    const float2 rv = *reinterpret_cast<const float2 *>(a.ranges + (long long)min(b, a.B - 1) * a.sample_stride + i0);
    const double2 t0v = *reinterpret_cast<const double2 *>(a.tab + N + 2 * i0);
    const double2 t1v = *reinterpret_cast<const double2 *>(a.tab + N + 2 * i0 + 2);
    if (a.flow != nullptr || a.det_offsets != nullptr) {
        // records -> LDS: a pure copy (the float32 motion and the detection screens were written by the
        // params job that produced the record)
        if (tid < 16) {
            const int sq = tid >> 3, c = tid & 7;
            const int bs = min(sA + sq, a.B - 1);
            const double *rec = a.ws_rec + (long long)bs * kRecStride;
            sm.mot[sq][c] = rec[c];
            sm.motf[sq][c] = reinterpret_cast<const float *>(rec + kRecF32Mot)[c];
        } else if (a.det_offsets != nullptr && tid < 16 + 2 * kInline) {
            const int w_ = tid - 16, sq = w_ / kInline, d = w_ - sq * kInline;
            const int bs = min(sA + sq, a.B - 1);
            const double *rec = a.ws_rec + (long long)bs * kRecStride;
            const double *w = rec + 8 + d * kDetStride;
            const float *f = reinterpret_cast<const float *>(rec + kRecF32Det) + d * 6;
            sm.det[sq][d][0] = w[0];
            sm.det[sq][d][1] = w[1];
            sm.det[sq][d][2] = w[2];
            sm.det[sq][d][3] = w[3];
            sm.det[sq][d][4] = w[5];
            sm.lab[sq][d] = (int)w[4];
            sm.detf[sq][d] = make_float4(f[0], f[1], f[2], f[3]);
            sm.deta[sq][d] = f[4];
        }
        __syncthreads();
    }
    flat_chunk<OutT>(a, sm, q, b, ok, o2, rv, t0v, t1v, a.tab + N + 2 * i0);
}

// grid (period, rows + extra): rows [0, main_rows) stream the current batch, the rest run the params jobs of
// the NEXT batch (chained form; params rows last, see scan_preprocess_chain_kernel)
template <typename OutT>
__global__ __launch_bounds__(kFlatThreads, 8) void scan_flat_kernel(PreArgs a, PreArgs nx, int main_rows, float inv_half_n)
{
    if ((int)blockIdx.y >= main_rows) {
        params_work(nx, (((int)blockIdx.y - main_rows) * (int)gridDim.x + (int)blockIdx.x) * kFlatThreads + threadIdx.x);
        return;
    }
    scan_flat<OutT>(a, blockIdx.x, blockIdx.y, gridDim.x, inv_half_n);
}

// ---- A4 stand-alone rotation ------------------------------------------------
template <typename T>
__global__ void rotate_flow_kernel(const T *in, T *out, const double *tab, long long total, int N,
                                   int to_canonical)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int i = (int)(p % N);
    double c = tab[N + 2 * i], s = tab[N + 2 * i + 1];
    double fx = (double)in[2 * p], fy = (double)in[2 * p + 1];
    double gx, gy;
    if (to_canonical) {
        gx = c * fx + (-s) * fy;
        gy = s * fx + c * fy;
    } else {
        gx = c * fx + s * fy;
        gy = (-s) * fx + c * fy;
    }
    out[2 * p] = (T)gx;
    out[2 * p + 1] = (T)gy;
}

// ---- A5 ---------------------------------------------------------------------
__global__ void det_to_canonical_kernel(const float *ranges, const double *tab, const double *det_r,
                                        const double *det_phi, double *dx, double *dy, long long total,
                                        int N)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int i = (int)(p % N);
    double s, c;
    sincos(det_phi[p] - tab[i], &s, &c);
    dx[p] = s * det_r[p];
    dy[p] = c * det_r[p] - (double)ranges[p];
}

__global__ void canonical_to_det_kernel(const float *ranges, const double *tab, const double *dx,
                                        const double *dy, double *det_r, double *det_phi,
                                        long long total, int N)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int i = (int)(p % N);
    double ty = (double)ranges[p] + dy[p];
    double tphi = atan2(dx[p], ty);
    det_phi[p] = tphi + tab[i];
    det_r[p] = ty / cos(tphi);
}

template <typename OutT, int PTS, int SPB>
__global__ __launch_bounds__(kThreads, 8) void scan_preprocess_kernel(PreArgs a)
{
    scan_main<OutT, PTS, SPB>(a, blockIdx.y);
}

// Chained launch: rows [0, main_rows) of the grid stream the current batch, the
// remaining rows run the params jobs of the NEXT batch into its workspace (independent
// data, so no intra-launch dependency).  The streaming rows are HBM-bound and leave
// the vector ALU mostly idle; the next batch's sincos work hides under them and the
// separate params launch (and its kernel boundary) disappears from the steady state.
template <typename OutT, int PTS, int SPB>
__global__ __launch_bounds__(kThreads, 8) void scan_preprocess_chain_kernel(PreArgs a, PreArgs nx, int main_rows)
{
    // params rows LAST: measured 20.2 us per step against 22.7 us with the params rows first
    // (dispatched first, their long sincos chains hold CU slots the streaming rows need)
    if ((int)blockIdx.y >= main_rows) {
        if (blockIdx.x == 0) params_work(nx, ((int)blockIdx.y - main_rows) * kThreads + threadIdx.x);
        return;
    }
    scan_main<OutT, PTS, SPB>(a, blockIdx.y);
}

// ---- A3 on caller-supplied cartesian points ------------------------------------
// get_displacement_from_odometry / get_velocity_from_odometry take scanner-frame
// xy, not ranges (src/utils/utils.py:609-662): convenience path, one workgroup
// per sample, motion evaluated in the prologue.
__global__ __launch_bounds__(256) void flow_from_xy_kernel(const double *xy, const double *odom0,
                                                           const double *odom1, int kind, int canonical,
                                                           const double *tab, double *out, int N)
{
    __shared__ double s_sc[4];
    __shared__ double s_mot[8];
    const int b = blockIdx.x;
    const double *o0 = odom0 + 3 * b, *o1 = odom1 + 3 * b;
    if (threadIdx.x < 2) {
        const int which = threadIdx.x;
        const double ang = motion_angle(kind, which, o0, o1);
        sincos(ang, &s_sc[2 * which], &s_sc[2 * which + 1]);
    }
    __syncthreads();
    if (threadIdx.x == 0) motion_params(kind, o0, o1, s_sc[0], s_sc[1], s_sc[2], s_sc[3], s_mot);
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const long long p = (long long)b * N + i;
        const double px = xy[2 * p], py = xy[2 * p + 1];
        double fx, fy;
        apply_motion(kind, px, py, s_mot[0], s_mot[1], s_mot[2], s_mot[3], s_mot[4], s_mot[5], s_mot[6], fx, fy);
        if (canonical) {
            const double c = tab[N + 2 * i], sn = tab[N + 2 * i + 1];
            const double gx = c * fx + (-sn) * fy, gy = sn * fx + c * fy;
            fx = gx;
            fy = gy;
        }
        out[2 * p] = fx;
        out[2 * p + 1] = fy;
    }
}

// ---- A2 inverse: xy_to_rphi (src/utils/utils.py:39-43) -------------------------
__global__ void xy_to_rphi_kernel(const double *x, const double *y, double *r, double *phi, long long n)
{
    long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    r[p] = hypot(x[p], y[p]);
    phi[p] = atan2(y[p], x[p]);
}

}  // namespace

extern "C" int pof_flow_from_xy(const double *xy, const double *odom0, const double *odom1, int flow_kind,
                                int canonical, const double *tab, double *flow, int B, int N,
                                pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!xy || !odom0 || !odom1 || !flow || B < 0 || N < 1) return POF_E_BADARG;
    if (flow_kind < 0 || flow_kind > 4) return POF_E_BADARG;
    if (canonical && !tab) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    flow_from_xy_kernel<<<B, 256, 0, pof_stream(stream)>>>(xy, odom0, odom1, flow_kind, canonical, tab, flow, N);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_xy_to_rphi(const double *x, const double *y, double *r, double *phi, long long n,
                              pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!x || !y || !r || !phi || n < 0) return POF_E_BADARG;
    if (n == 0) return POF_OK;
    xy_to_rphi_kernel<<<(unsigned)((n + 255) / 256), 256, 0, pof_stream(stream)>>>(x, y, r, phi, n);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_laser_phi(double angle_inc, int num_pts, double *tab, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!tab || num_pts < 2) return POF_E_BADARG;
    // host scalar arithmetic, identical to the Python float arithmetic of the reference
    const double fov = (double)(num_pts - 1) * angle_inc;
    const double start = -fov * 0.5, stop = fov * 0.5;
    const double step = (stop - start) / (double)(num_pts - 1);
    laser_phi_kernel<<<(num_pts + 255) / 256, 256, 0, pof_stream(stream)>>>(start, stop, step, num_pts, tab);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

// Largest float64 s with RN(sqrt(s)) <= r (strict = false) or < r (strict = true).
// sqrt is monotone and correctly rounded (IEEE, host libm), so comparing the
// SQUARED distance with this threshold decides "dist <= r" / "dist < r" exactly
// as the reference does on the rounded distance -- without a square root per point.
static double sq_threshold(double r, bool strict)
{
    if (!(r > 0.0)) return (!strict && r == 0.0) ? 0.0 : -1.0;
    auto ok = [&](double s) { double q = std::sqrt(s); return strict ? (q < r) : (q <= r); };
    double c = r * r;
    for (int it = 0; it < 64 && !ok(c); ++it) c = std::nextafter(c, 0.0);
    for (int it = 0; it < 64; ++it) {
        double up = std::nextafter(c, HUGE_VAL);
        if (!ok(up)) break;
        c = up;
    }
    return c;
}

extern "C" size_t pof_scan_preprocess_workspace_bytes(int B, int D)
{
    if (B < 0 || D < 0) return 0;
    return ((size_t)B * kRecStride + (size_t)D * kDetStride) * sizeof(double) + 64;
}

namespace {

// class constants (radii, labels, exact squared-distance thresholds) of a params job set
void fill_class_constants(PreArgs &a, bool have_dets, const double *assoc_radius, const int32_t *labels,
                          const double *dyn_radius)
{
    double thr[6] = {0, 0, 0, 0, 0, 0};
    if (have_dets)
        for (int k = 0; k < 3; ++k) {
            thr[k] = sq_threshold(dyn_radius[k], false);
            thr[3 + k] = sq_threshold(assoc_radius[k], true);
        }
    a.ra0 = have_dets ? assoc_radius[0] : 0.0; a.ra1 = have_dets ? assoc_radius[1] : 0.0;
    a.ra2 = have_dets ? assoc_radius[2] : 0.0;
    a.sd0 = thr[0]; a.sd1 = thr[1]; a.sd2 = thr[2];
    a.sa0 = thr[3]; a.sa1 = thr[4]; a.sa2 = thr[5];
    a.lb0 = have_dets ? labels[0] : 0; a.lb1 = have_dets ? labels[1] : 0; a.lb2 = have_dets ? labels[2] : 0;
}

void bind_workspace(PreArgs &a, void *workspace, int B)
{
    // workspace: 64-byte aligned per-sample records, then the CSR detection table
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 63) & ~(uintptr_t)63;
    a.ws_rec = reinterpret_cast<double *>(base);
    a.ws_det = a.ws_rec + (size_t)B * kRecStride;
}

int launch_preprocess(const float *ranges, long long sample_stride, int B, int N, const double *tab,
                      const double *odom0, const double *odom1, int flow_kind, int canonical, int out_f64,
                      void *xy, void *flow, const int32_t *det_offsets, const double *det_rphi,
                      const uint8_t *det_cls, int D, const double *assoc_radius, const int32_t *labels,
                      const double *dyn_radius, int64_t *closest, int64_t *target_cls, float *target_reg,
                      float *dyn_mask, float *valid_mask, float *exclude_mask, void *workspace,
                      size_t workspace_bytes, int phases, const pof_scan_inputs *next, pof_stream_t stream)
{
    if ((phases & 3) == 0 || (phases & ~3)) return POF_E_BADARG;
    if (!ranges || !tab || B < 0 || N < 1 || D < 0) return POF_E_BADARG;
    if (flow && (!odom0 || !odom1)) return POF_E_BADARG;
    if (flow_kind < 0 || flow_kind > 4) return POF_E_BADARG;
    if (det_offsets && (!assoc_radius || !labels || !dyn_radius)) return POF_E_BADARG;
    if (det_offsets && D > 0 && (!det_rphi || !det_cls)) return POF_E_BADARG;
    if (sample_stride < N) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;  // grid.y limit; callers chunk larger batches
    const bool need_ws = flow || det_offsets;
    if (need_ws && (!workspace || workspace_bytes < pof_scan_preprocess_workspace_bytes(B, det_offsets ? D : 0)))
        return POF_E_WORKSPACE;
    PreArgs a;
    a.ranges = ranges; a.sample_stride = sample_stride; a.B = B; a.N = N; a.D = det_offsets ? D : 0;
    a.tab = tab;
    a.odom0 = odom0; a.odom1 = odom1; a.flow_kind = flow_kind; a.canonical = canonical;
    a.xy = xy; a.flow = flow; a.det_offsets = det_offsets; a.det_rphi = det_rphi; a.det_cls = det_cls;
    fill_class_constants(a, det_offsets != nullptr, assoc_radius, labels, dyn_radius);
    a.closest = closest; a.target_cls = target_cls; a.target_reg = target_reg;
    a.dyn_mask = dyn_mask; a.valid_mask = valid_mask; a.exclude_mask = exclude_mask;
    bind_workspace(a, workspace, B);
    hipStream_t s = pof_stream(stream);
    if (need_ws && (phases & 1)) {
        const int total = params_job_count(B, det_offsets != nullptr);
        scan_params_kernel<<<(total + 255) / 256, 256, 0, s>>>(a);
        POF_CHECK_LAUNCH();
    }
    if (!(phases & 2)) return POF_OK;

    // params job set of the next batch (chained form)
    PreArgs nx = a;
    int next_jobs = 0;
    if (next) {
        if (next->B < 0 || next->D < 0 || next->B > 65535) return POF_E_BADARG;
        if (next->want_flow && (!next->odom0 || !next->odom1)) return POF_E_BADARG;
        if (next->flow_kind < 0 || next->flow_kind > 4) return POF_E_BADARG;
        if (next->det_offsets && next->D > 0 && (!next->det_rphi || !next->det_cls)) return POF_E_BADARG;
        if (next->B > 0 && (next->want_flow || next->det_offsets)) {
            if (!next->workspace ||
                next->workspace_bytes < pof_scan_preprocess_workspace_bytes(next->B, next->det_offsets ? next->D : 0))
                return POF_E_WORKSPACE;
            nx.B = next->B; nx.D = next->det_offsets ? next->D : 0;
            nx.odom0 = next->odom0; nx.odom1 = next->odom1; nx.flow_kind = next->flow_kind;
            nx.flow = next->want_flow ? reinterpret_cast<void *>(1) : nullptr;  // only tested against null
            nx.det_offsets = next->det_offsets; nx.det_rphi = next->det_rphi; nx.det_cls = next->det_cls;
            fill_class_constants(nx, next->det_offsets != nullptr, next->assoc_radius, next->labels, next->dyn_radius);
            bind_workspace(nx, next->workspace, next->B);
            next_jobs = params_job_count(nx.B, nx.det_offsets != nullptr);
        }
    }

    // float2 row loads need 8-byte aligned rows, 16-byte stores aligned outputs
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec2 = (N % 2 == 0) && (sample_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(ranges) & 7) == 0) &&
                      al16(xy) && al16(flow) && al16(closest) && al16(target_cls) && al16(target_reg) &&
                      al16(dyn_mask) && al16(valid_mask) && al16(exclude_mask);
    // One sample per workgroup.  (Two samples per workgroup share the table loads and were
    // faster with ordinary stores; with streaming stores one is: 15.4 vs 17.9 us per step.)
    bool chained = false;
    const long long flat_chunks = ((long long)B * (N / 2) + kFlatThreads - 1) / kFlatThreads;
    int g = N, h = 2 * kFlatThreads;
    while (h) { const int t = g % h; g = h; h = t; }
    const int period = N / g;                                        // chunks after which the point phases repeat
    const long long rows = (flat_chunks + period - 1) / period;
    const int extra_rows = (int)(((long long)next_jobs + (long long)period * kFlatThreads - 1) / ((long long)period * kFlatThreads));
    if (vec2 && N >= 2 * kFlatThreads && (long long)period * kFlatThreads < (1 << 24) && rows + extra_rows <= 65535) {
        // flat form: line-aligned output chunks per workgroup (see scan_flat)
        dim3 grid(period, (unsigned)(rows + extra_rows));
        const float inv = 1.0f / (float)(N / 2);
        if (out_f64) scan_flat_kernel<double><<<grid, kFlatThreads, 0, s>>>(a, nx, (int)rows, inv);
        else scan_flat_kernel<float><<<grid, kFlatThreads, 0, s>>>(a, nx, (int)rows, inv);
        chained = true;
    } else if (vec2) {
        dim3 grid((N / 2 + kThreads - 1) / kThreads, B);
        const int extra = (next_jobs + kThreads - 1) / kThreads;
        if (extra > 0 && B + extra <= 65535) {
            dim3 gc(grid.x, B + extra);
            if (out_f64) scan_preprocess_chain_kernel<double, 2, 1><<<gc, kThreads, 0, s>>>(a, nx, B);
            else scan_preprocess_chain_kernel<float, 2, 1><<<gc, kThreads, 0, s>>>(a, nx, B);
            chained = true;
        } else {
            if (out_f64) scan_preprocess_kernel<double, 2, 1><<<grid, kThreads, 0, s>>>(a);
            else scan_preprocess_kernel<float, 2, 1><<<grid, kThreads, 0, s>>>(a);
        }
    } else {
        dim3 grid((N + kThreads - 1) / kThreads, B);
        if (out_f64) scan_preprocess_kernel<double, 1, 1><<<grid, kThreads, 0, s>>>(a);
        else scan_preprocess_kernel<float, 1, 1><<<grid, kThreads, 0, s>>>(a);
    }
    POF_CHECK_LAUNCH();
    if (next_jobs > 0 && !chained) {
        // shapes the chained kernel is not built for: the next batch's params as their own launch
        scan_params_kernel<<<(next_jobs + 255) / 256, 256, 0, s>>>(nx);
        POF_CHECK_LAUNCH();
    }
    return POF_OK;
}

}  // namespace

extern "C" int pof_scan_preprocess_phase(const float *ranges, long long sample_stride, int B, int N,
                                         const double *tab, const double *odom0, const double *odom1,
                                         int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                                         const int32_t *det_offsets, const double *det_rphi,
                                         const uint8_t *det_cls, int D, const double *assoc_radius,
                                         const int32_t *labels, const double *dyn_radius, int64_t *closest,
                                         int64_t *target_cls, float *target_reg, float *dyn_mask,
                                         float *valid_mask, float *exclude_mask, void *workspace,
                                         size_t workspace_bytes, int phases, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return launch_preprocess(ranges, sample_stride, B, N, tab, odom0, odom1, flow_kind, canonical, out_f64, xy, flow,
                             det_offsets, det_rphi, det_cls, D, assoc_radius, labels, dyn_radius, closest,
                             target_cls, target_reg, dyn_mask, valid_mask, exclude_mask, workspace,
                             workspace_bytes, phases, nullptr, stream);
}

extern "C" int pof_scan_preprocess_chained(const float *ranges, long long sample_stride, int B, int N,
                                           const double *tab, const double *odom0, const double *odom1,
                                           int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                                           const int32_t *det_offsets, const double *det_rphi,
                                           const uint8_t *det_cls, int D, const double *assoc_radius,
                                           const int32_t *labels, const double *dyn_radius, int64_t *closest,
                                           int64_t *target_cls, float *target_reg, float *dyn_mask,
                                           float *valid_mask, float *exclude_mask, void *workspace,
                                           size_t workspace_bytes, const pof_scan_inputs *next,
                                           pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return launch_preprocess(ranges, sample_stride, B, N, tab, odom0, odom1, flow_kind, canonical, out_f64, xy, flow,
                             det_offsets, det_rphi, det_cls, D, assoc_radius, labels, dyn_radius, closest,
                             target_cls, target_reg, dyn_mask, valid_mask, exclude_mask, workspace,
                             workspace_bytes, 2, next, stream);
}

extern "C" int pof_scan_preprocess(const float *ranges, long long sample_stride, int B, int N,
                                   const double *tab, const double *odom0, const double *odom1,
                                   int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                                   const int32_t *det_offsets, const double *det_rphi,
                                   const uint8_t *det_cls, int D, const double *assoc_radius,
                                   const int32_t *labels, const double *dyn_radius, int64_t *closest,
                                   int64_t *target_cls, float *target_reg, float *dyn_mask,
                                   float *valid_mask, float *exclude_mask, void *workspace,
                                   size_t workspace_bytes, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return pof_scan_preprocess_phase(ranges, sample_stride, B, N, tab, odom0, odom1, flow_kind, canonical, out_f64,
                                     xy, flow, det_offsets, det_rphi, det_cls, D, assoc_radius, labels, dyn_radius,
                                     closest, target_cls, target_reg, dyn_mask, valid_mask, exclude_mask, workspace,
                                     workspace_bytes, 3, stream);
}

extern "C" int pof_rotate_flow(const void *flow_in, void *flow_out, const double *tab, int B, int N,
                               int to_canonical, int is_f64, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!flow_in || !flow_out || !tab || B < 0 || N < 1) return POF_E_BADARG;
    long long total = (long long)B * N;
    if (total == 0) return POF_OK;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (is_f64)
        rotate_flow_kernel<double><<<blocks, 256, 0, pof_stream(stream)>>>(
            static_cast<const double *>(flow_in), static_cast<double *>(flow_out), tab, total, N, to_canonical);
    else
        rotate_flow_kernel<float><<<blocks, 256, 0, pof_stream(stream)>>>(
            static_cast<const float *>(flow_in), static_cast<float *>(flow_out), tab, total, N, to_canonical);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_det_to_canonical(const float *ranges, const double *tab, const double *det_r,
                                    const double *det_phi, double *dx, double *dy, int B, int N,
                                    pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!ranges || !tab || !det_r || !det_phi || !dx || !dy || B < 0 || N < 1) return POF_E_BADARG;
    long long total = (long long)B * N;
    if (total == 0) return POF_OK;
    det_to_canonical_kernel<<<(unsigned)((total + 255) / 256), 256, 0, pof_stream(stream)>>>(
        ranges, tab, det_r, det_phi, dx, dy, total, N);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_canonical_to_det(const float *ranges, const double *tab, const double *dx,
                                    const double *dy, double *det_r, double *det_phi, int B, int N,
                                    pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!ranges || !tab || !det_r || !det_phi || !dx || !dy || B < 0 || N < 1) return POF_E_BADARG;
    long long total = (long long)B * N;
    if (total == 0) return POF_OK;
    canonical_to_det_kernel<<<(unsigned)((total + 255) / 256), 256, 0, pof_stream(stream)>>>(
        ranges, tab, dx, dy, det_r, det_phi, total, N);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_abi_version(void) { return POF_ABI_VERSION; }

thread_local int pof_stale_error_slot = 0;

extern "C" int pof_take_stale_error(void)
{
    const int e = pof_stale_error_slot;
    pof_stale_error_slot = 0;
    return e;
}

extern "C" const char *pof_error_string(int code)
{
    switch (code) {
        case POF_OK: return "ok";
        case POF_E_BADARG: return "bad argument (null pointer, illegal enum or negative size)";
        case POF_E_SHAPE: return "inconsistent or unsupported shape";
        case POF_E_LAUNCH: return "HIP launch failed";
        case POF_E_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
