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
// Per-sample record, kRecStride doubles = 832 bytes:
//   float64 part  [0, 8)   motion[7] + detection count
//                 [8, 56)  kInline detections x kDetStride {cx, cy, assoc radius, dyn s-thr, label, assoc s-thr}
//   flat image    (384 contiguous bytes: what a wave of the flat kernel copies into LDS, 24 lanes x 16 bytes)
//                 +0    float32 header, 16 floats: [0,7) motion, [7] detection count (int bits), [8,10) class of
//                       the inline detections (8 x u8), [10,12) "far" flag of the inline detections (8 x u8)
//                 +64   kInline x float4 {cx, cy, -sd, sa - sd}: the float32 screen (flat_run); unused slots
//                       hold the null entry {0, 0, +inf, 0}
//                 +192  kInline x {cx, cy} float64: centres of the inline detections
//                 +320  motion[7] + detection count, float64 (float64-output kernels)
constexpr int kRecImg = 8 + kInline * kDetStride;           // in doubles: byte 448
constexpr int kRecHdrFloats = 16;
constexpr int kImgScreen = kRecHdrFloats * 4;               // byte offsets inside the image
constexpr int kImgCtr = kImgScreen + 16 * kInline;          // 192
constexpr int kImgMot = kImgCtr + 16 * kInline;             // 320
constexpr int kImgBytes = kImgMot + 64;                     // 384
constexpr int kRecStride = kRecImg + kImgBytes / 8;         // 104 doubles

// Per-sample rigid motion from the two (sin, cos) pairs the params kernel
// evaluates on two lanes.  mot[0..3] = 2x2 matrix (row major), mot[4..5] =
// translation, mot[6] = dphi.
//   kind 0: (sA,cA) = sincos(phi0), (sB,cB) = sincos(phi1)
//   kind 1: (sA,cA) = sincos(phi0), (sB,cB) = sincos(phi1 - phi0)
//   kind 2: (sB,cB) = sincos(phi1)
// A zero the compiler cannot hoist: as a loop-invariant constant vector the zero fields of the records were kept
// live round the params loop of the flat kernel and spilled (the only scratch use of that kernel).
__device__ __forceinline__ double opaque_zero()
{
    double z;
    asm volatile("v_mov_b64 %0, 0" : "=v"(z));
    return z;
}

__device__ __forceinline__ void motion_params(int kind, const double *o0, const double *o1, double sA, double cA,
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
        mot[6] = opaque_zero();
    } else if (kind == 1) {
        // get_flow_target: float64 throughout
        mot[0] = cB; mot[1] = -sB; mot[2] = sB; mot[3] = cB;
        // trans_world @ rot_0.T
        mot[4] = fma(ty, -sA, tx * cA);
        mot[5] = fma(ty, cA, tx * sA);
        mot[6] = opaque_zero();
    } else if (kind == 2) {
        // get_velocity_from_odometry: float32 R1, cross matrix dphi*[[0,-1],[1,0]]
        const float c1 = (float)cB, s1 = (float)sB;
        mot[0] = mot[1] = mot[2] = mot[3] = opaque_zero();
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
        mot[4] = mot[5] = mot[6] = opaque_zero();
    } else {
        // scan-pair alignment (src/utils/dataset.py:76-93): o0 = (dx, dy, dphi), o1[0] = scan_dir
        // (sA,cA) = sincos(dphi), (sB,cB) = sincos(scan_dir); float32 matrices
        const float c = (float)cA, sn = (float)sA, cd = (float)cB, sd = (float)sB;
        mot[0] = (double)c; mot[1] = (double)sn; mot[2] = (double)(-sn); mot[3] = (double)c;
        mot[4] = fma(o0[1], (double)(-sd), o0[0] * (double)cd);
        mot[5] = fma(o0[1], (double)cd, o0[0] * (double)sd);
        mot[6] = opaque_zero();
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
// sincos for the params jobs.  The library routine carries the Payne-Hanek reduction for arbitrary magnitudes: 96
// registers, which the 64-register budget of the flat kernel (8 streaming waves per SIMD) turns into scratch for
// the whole launch.  Odometry headings and detection bearings are angles of a few radians, so the params jobs use
// the classic medium-range scheme instead -- Cody-Waite reduction by pi/2 in up to three 33-bit steps (exact
// products for |n| < 2^20, i.e. |x| < 2^19 * pi/2 ~ 8.2e5 rad) followed by the fdlibm minimax polynomials on
// [-pi/4, pi/4] with the reduction's tail as correction: below 1 ulp, like the library's.  Beyond that range
// FALLBACK selects the library routine (the stand-alone params kernel) or NaN (the params blocks inside the flat
// kernel: documented input domain of the chained / multi entry points).
__device__ __forceinline__ double poly_sin(double x, double y)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
__device__ __forceinline__ double poly_cos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
    const double qx = ax > 0.78125 ? 0.28125 : 0.25 * ax;
    const double hz = 0.5 * z - qx, a = 1.0 - qx;
    return a - (hz - (z * r - x * y));
}
template <bool FALLBACK>
__device__ __forceinline__ void params_sincos(double x, double *sn, double *cs)
{
    const double ax = fabs(x);
    if (!(ax < 823549.0)) {                 // 2^19 * pi/2 = 823549.66...; NaN and infinities come here too
        if (FALLBACK) {
            sincos(x, sn, cs);
        } else {
            *sn = *cs = __builtin_nan("");
        }
        return;
    }
    const double invpio2 = 6.36619772367581382433e-01, p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11,
                 p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21, p3 = 2.02226624871116645580e-21,
                 p3t = 8.47842766036889956997e-32;
    const double fn = rint(ax * invpio2);
    double r = ax - fn * p1;                // fn * p1 is exact (20 + 33 bits)
    double w = fn * p1t;
    double y0 = r - w;
    const int ex = (__double2hiint(ax) >> 20) & 0x7ff;
    if (ex - ((__double2hiint(y0) >> 20) & 0x7ff) > 16) {       // cancellation: the next 33 bits of pi/2
        double t = r;
        w = fn * p2;
        r = t - w;
        w = fn * p2t - ((t - r) - w);
        y0 = r - w;
        if (ex - ((__double2hiint(y0) >> 20) & 0x7ff) > 49) {   // and the next
            t = r;
            w = fn * p3;
            r = t - w;
            w = fn * p3t - ((t - r) - w);
            y0 = r - w;
        }
    }
    double y1 = (r - y0) - w;
    int n = (int)fn;
    if (x < 0.0) {
        y0 = -y0;
        y1 = -y1;
        n = -n;
    }
    const double s = poly_sin(y0, y1), c = poly_cos(y0, y1);
    // quadrant: 0 (s, c), 1 (c, -s), 2 (-s, -c), 3 (-c, s) -- as selects (a switch becomes a table in scratch)
    const bool odd = (n & 1) != 0;
    double ss = odd ? c : s, cc = odd ? s : c;
    if (n & 2) ss = -ss;
    if ((n + 1) & 2) cc = -cc;
    *sn = ss;
    *cs = cc;
}

// AP: pointer to the arguments -- a plain pointer to a by-value kernel argument, or a constant-address-space
// pointer into the kernel-argument segment (slot chosen at run time, scan_flat_kernel)
template <typename AP>
__device__ __forceinline__ void det_entry(AP a, int g, double *w)
{
    const double dr = a->det_rphi[2 * g], dp = a->det_rphi[2 * g + 1];
    double s, c;
    sincos(dp, &s, &c);
    const int cl = a->det_cls[g];
    w[0] = dr * c;
    w[1] = dr * s;
    w[2] = cl == 0 ? a->ra0 : (cl == 1 ? a->ra1 : a->ra2);
    w[3] = cl == 0 ? a->sd0 : (cl == 1 ? a->sd1 : a->sd2);   // dist <= dyn radius   <=>  s <= w[3]
    w[4] = (double)(cl == 0 ? a->lb0 : (cl == 1 ? a->lb1 : a->lb2));
    w[5] = cl == 0 ? a->sa0 : (cl == 1 ? a->sa1 : a->sa2);   // dist <  assoc radius <=>  s <= w[5]
}

// float32 screen of the flat kernel (flat_run).  For a detection j the kernel tracks
//   dd_j = fl(ex*ex + fl(ey*ey - sd_j))     ex, ey = float32 point - float32 centre, sd_j rounded to float32
// and keeps the minimum over the sample's inline detections.  With all coordinates below 100 m and sd <= 400 m^2,
// |dd_j - (s2_j - sd_j)| <= E(s2_j) = 1e-3 + 1e-4 * s2_j, where s2_j is the float64 squared distance the
// reference's decision is made on (derivation in DESIGN.md section 3.1).  Hence, with M_j = 1e-3 + 1e-4 * sd_j:
//   dd_j <= -M_j  =>  s2_j <= sd_j   (surely inside the dynamic-mask radius)
//   dd_j >   M_j  =>  s2_j >  sd_j   (surely outside)
//   s2_j <= sa_j (association candidate)  =>  dd_j <= sa_j - sd_j + M_j
// so the minimum alone decides a point unless it lies in one of the error bands; those points, every point
// beyond 100 m, and all points of a sample with a "far" detection (centre beyond 100 m or sd > 400) take the
// exact float64 loop.
__device__ __forceinline__ bool det_is_far(double cx, double cy, double sd)
{
    return !(fabs(cx) + fabs(cy) < 100.0) || !(sd <= 400.0);
}

// number of params jobs of a batch: kInline (8) lanes per sample
__host__ __device__ inline int params_job_count(int B, bool have_dets)
{
    (void)have_dets;
    return kInline * B;
}

// Params job t: lane `slot` = t & 7 of sample t >> 3.  Slots 0 and 1 evaluate the two motion angles (slot 0 combines
// them); then slot s evaluates detections s, s + 8, ... -- one sincos per pass of ONE loop (a single call site, and
// nothing carried round the loop but indices: the job fits the 64 registers of the streaming waves without
// scratch) -- into the sample's record.
// What decides the layout is the STORE pattern, not the arithmetic.  The params blocks share their CUs' memory
// pipes with streaming waves that are bound by exactly those pipes; a store instruction whose 64 lanes hit 64
// different cache lines costs 64 line transactions.  Round 3's first form (two lanes per sample, every lane
// writing its sample's fields one by one) issued ~2 400 line transactions per wave, a third of what the streaming
// waves of the same CU need for a whole batch, and measured 1.2 us per 4096-scan step (2.2 us at 8 detections per
// sample).  Here the eight slot lanes of a sample write its eight 16-byte screen entries, its eight 16-byte
// centres and its class / far bytes as contiguous runs (one line each), slot 0 writes the header and the float64
// motion as 16-byte stores: ~11 store instructions per wave of 8 samples, <= 8 lines each.  WRITE_F64: also the
// float64 part that only the per-sample kernels read (the flat kernel needs the 384-byte image and, for crowded
// samples, the CSR table).
template <bool FALLBACK, bool WRITE_F64, typename AP>
__device__ __forceinline__ void params_work(AP a, int t)
{
    if (t >= kInline * a->B) return;               // whole groups of 8 lanes
    const int b = t >> 3, slot = t & 7;
    double *rec = a->ws_rec + (long long)b * kRecStride;
    unsigned char *img = reinterpret_cast<unsigned char *>(rec + kRecImg);
    int d0 = 0, cnt = 0;
    if (a->det_offsets) {
        d0 = a->det_offsets[b];
        cnt = a->det_offsets[b + 1] - d0;
    }
    // pass 0: the motion angle (slots 0, 1); pass p >= 1: detection slot + 8 (p - 1)
    const int npass = 1 + (cnt - slot + kInline - 1) / kInline;
#pragma unroll 1
    for (int pass = a->flow ? 0 : 1; pass < max(npass, 1); ++pass) {
        const int idx = slot + kInline * (pass - 1);
        const bool active = pass == 0 ? slot < 2 : idx < cnt;
        double ang = 0.0, dr = 0.0;
        if (pass == 0) {
            if (active) ang = motion_angle(a->flow_kind, slot, a->odom0 + 3 * b, a->odom1 + 3 * b);
        } else if (active) {
            dr = a->det_rphi[2 * (d0 + idx)];
            ang = a->det_rphi[2 * (d0 + idx) + 1];
        }
        double sn, cs;
        params_sincos<FALLBACK>(ang, &sn, &cs);
        if (pass == 0) {
            // every lane of the wave is here together (pass 0 is each lane's first): the partner's angle
            const double s_o = __shfl_xor(sn, 1, 64), c_o = __shfl_xor(cs, 1, 64);
            if (slot == 0) {
                // straight into the image's float64 motion (a private array here is indexed by the flow kind
                // after inlining and lands in scratch); the float32 header and the float64 part are copies of it
                double *m64 = reinterpret_cast<double *>(img + kImgMot);
                motion_params(a->flow_kind, a->odom0 + 3 * b, a->odom1 + 3 * b, sn, cs, s_o, c_o, m64);
                m64[7] = (double)cnt;
                float4 *hf = reinterpret_cast<float4 *>(img);
                hf[0] = make_float4((float)m64[0], (float)m64[1], (float)m64[2], (float)m64[3]);
                hf[1] = make_float4((float)m64[4], (float)m64[5], (float)m64[6], __int_as_float(cnt));
                if (WRITE_F64) {
                    double2 *r2 = reinterpret_cast<double2 *>(rec);
#pragma unroll
                    for (int k = 0; k < 4; ++k) r2[k] = make_double2(m64[2 * k], m64[2 * k + 1]);
                }
            }
        } else if (active) {
            const unsigned cl = a->det_cls[d0 + idx] > 1 ? 2u : (unsigned)a->det_cls[d0 + idx];
            double w[kDetStride];
            w[0] = dr * cs;
            w[1] = dr * sn;
            w[2] = cl == 0 ? a->ra0 : (cl == 1 ? a->ra1 : a->ra2);
            w[3] = cl == 0 ? a->sd0 : (cl == 1 ? a->sd1 : a->sd2);   // dist <= dyn radius   <=>  s <= w[3]
            w[4] = (double)(cl == 0 ? a->lb0 : (cl == 1 ? a->lb1 : a->lb2));
            w[5] = cl == 0 ? a->sa0 : (cl == 1 ? a->sa1 : a->sa2);   // dist <  assoc radius <=>  s <= w[5]
            if (idx < kInline) {
                const bool far = det_is_far(w[0], w[1], w[3]);
                reinterpret_cast<float4 *>(img + kImgScreen)[idx] =
                    make_float4((float)w[0], (float)w[1], far ? INFINITY : -(float)w[3], (float)(w[5] - w[3]));
                reinterpret_cast<double2 *>(img + kImgCtr)[idx] = make_double2(w[0], w[1]);
                img[32 + idx] = (unsigned char)cl;
                img[40 + idx] = far ? 1 : 0;
                if (WRITE_F64) {
                    double2 *r2 = reinterpret_cast<double2 *>(rec + 8 + idx * kDetStride);
                    r2[0] = make_double2(w[0], w[1]);
                    r2[1] = make_double2(w[2], w[3]);
                    r2[2] = make_double2(w[4], w[5]);
                }
            }
            if (cnt > kInline) {
                double2 *r2 = reinterpret_cast<double2 *>(a->ws_det + (long long)(d0 + idx) * kDetStride);
                r2[0] = make_double2(w[0], w[1]);
                r2[1] = make_double2(w[2], w[3]);
                r2[2] = make_double2(w[4], w[5]);
            }
        }
    }
    if (!a->flow && slot == 0) {
        // no motion pass: the count still belongs in the header (and in the float64 copies)
        reinterpret_cast<int *>(img)[7] = cnt;
        reinterpret_cast<double *>(img + kImgMot)[7] = (double)cnt;
        if (WRITE_F64) rec[7] = (double)cnt;
    }
    if (a->det_offsets && slot >= cnt) {
        // unused inline slot: the null screen entry (never the minimum), not far
        reinterpret_cast<float4 *>(img + kImgScreen)[slot] = make_float4(0.0f, 0.0f, INFINITY, 0.0f);
        img[40 + slot] = 0;
    }
}

__global__ __launch_bounds__(256) void scan_params_kernel(PreArgs a)
{
    params_work<true, true>(&a, blockIdx.x * blockDim.x + threadIdx.x);
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
// Pointers that were not direct kernel arguments (a batch slot read from the kernel-argument segment) carry no
// address space: without the cast every access through them is a flat_ instruction (LDS aperture check, counted
// in both vmcnt and lgkmcnt).
#define GLOBAL_AS __attribute__((address_space(1)))
using F4V = float __attribute__((ext_vector_type(4)));
using F2V = float __attribute__((ext_vector_type(2)));
using LL2V = long long __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_f4(float *base, long long idx4, float x, float y, float z, float w)
{
    F4V v = {x, y, z, w};
    __builtin_nontemporal_store(v, (GLOBAL_AS F4V *)base + idx4);
}
__device__ __forceinline__ void stream_f2(float *base, long long idx2, float x, float y)
{
    F2V v = {x, y};
    __builtin_nontemporal_store(v, (GLOBAL_AS F2V *)base + idx2);
}
__device__ __forceinline__ void stream_ll2(long long *base, long long idx2, long long x, long long y)
{
    LL2V v = {x, y};
    __builtin_nontemporal_store(v, (GLOBAL_AS LL2V *)base + idx2);
}

template <typename OutT>
__device__ __forceinline__ void store2(OutT *base, long long idx, double a, double b)
{
    using V = OutT __attribute__((ext_vector_type(2)));
    V v = {(OutT)a, (OutT)b};
    ((GLOBAL_AS V *)base)[idx] = v;
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
// One WAVE = 64 lanes x 2 points = 128 CONSECUTIVE POINTS OF THE FLAT [B*N] AXIS, not one sample.  A
// sample's output rows are 8N = 3600 B (N = 450): with one workgroup per sample every row boundary splits a
// 128-byte line between two workgroups on different XCDs, and the bare I/O shape of the launch runs at
// 4.6-4.9 TB/s; with line-aligned 1024-byte (flow / reg / cls) and 512-byte (mask) runs per wave the same
// bytes stream at 6.2-6.3 TB/s (tools/ubench/store_ceiling.hip, profiles/r2_store_*).
// A wave touches at most two samples (N >= 128 * CPW): it copies the float32 part of their records (motion,
// screen of the inline detections) and the float64 centres into its OWN slice of LDS -- no workgroup barrier,
// LDS operations of one wave are ordered -- and each lane picks its own sample's.  Crowded samples (more than
// kInline detections) read the rest of their detections from the CSR table in the workspace (exact test only).
//
// Round 3: the launch covers up to kMaxSlots (POF_SCAN_MAX_SLOTS) batches (a data loader hands over several ring slots at once):
// wave-chunk w of the grid belongs to batch k = #{i : wave0[i] <= w}; the batch's pointers are read from the
// kernel arguments with a scalar-indexed load.  Everything the wave needs per point is a compile-time
// configuration (CFG >= 0: which outputs exist, flow kind, canonical) or a host-computed constant (the
// division by N/2 is a multiply-high); CFG < 0 keeps the run-time flags for the remaining combinations.
constexpr int kMaxSlots = POF_SCAN_MAX_SLOTS;
constexpr int kWaveLds = 2 * kImgBytes;                    // the flat images of the wave's two samples: 768 B

enum : unsigned {
    kOutXy = 1u << 0, kOutFlow = 1u << 1, kHasDets = 1u << 2, kOutClosest = 1u << 3, kOutCls = 1u << 4,
    kOutReg = 1u << 5, kOutDyn = 1u << 6, kOutValid = 1u << 7, kOutExcl = 1u << 8, kCanonical = 1u << 9,
    kKindShift = 10,
};
// the headline configuration: displacement flow in the canonical frame, target_cls, target_reg, exclude mask
constexpr int kCfgHeadline = (int)(kOutFlow | kHasDets | kOutCls | kOutReg | kOutExcl | kCanonical);

struct FlatBatch {
    const float *ranges;
    long long sample_stride;
    void *xy, *flow;
    int64_t *closest, *target_cls;
    float *target_reg, *dyn_mask, *valid_mask, *exclude_mask;
    const int32_t *det_offsets;
    const double *ws_rec, *ws_det;
    int B;
    int wave0;              // first wave-chunk of this batch in the launch
};
static_assert(sizeof(FlatBatch) % 8 == 0, "FlatBatch is read from the kernel arguments in 8-byte words");

struct FlatArgs {
    const double *tab;      // [3N] float64: phi, then (cos, sin) interleaved
    const float *tabf;      // [N][2] = (float)cos, (float)sin, or nullptr (converted in the kernel)
    int N, halfN;
    unsigned magic_m;       // x / halfN = mulhi(x, magic_m) >> magic_s for x < 2^31
    int magic_s;
    unsigned flags;         // run-time configuration (CFG < 0)
    int nb;                 // batches in this launch
    int total_waves;        // wave-chunks of all batches
    int main_blocks;        // blocks that stream; blocks behind them run params jobs
    double ra[3], sd[3], sa[3];
    int lb[3];
    float m_dyn;            // max_c (1e-3 + 1e-4 sd_c), rounded up
    float thr_assoc;        // max_c (sa_c - sd_c) + m_assoc, rounded up
    float m_assoc;          // max_c (1e-3 + 1e-4 max(sd_c, sa_c)) + the rounding of (sa - sd) to float32, rounded up
    int uni_waves;          // wave-chunks per batch when every batch of the launch has the same count, else 0
    unsigned uni_m;         // w / uni_waves = mulhi(w, uni_m) >> uni_s (the slot of wave-chunk w without a search)
    int uni_s;
    int pad_[2];
    int wave0[kMaxSlots];   // first wave-chunk of each batch
    FlatBatch b[kMaxSlots];
};

struct ParamsMulti {
    int nb;
    int first, total;          // the params blocks are blocks [first, first + total) of the grid
    int uni_blocks;            // params blocks per batch when all batches have the same count, else 0
    unsigned uni_m;            // blk / uni_blocks = mulhi(blk, uni_m) >> uni_s
    int uni_s;
    int blk0[kMaxSlots + 1];   // first params block of each batch (unused slots: the total); blk0[kMaxSlots] = total
    PreArgs p[kMaxSlots];
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

template <int CFG>
struct FlatCfg {
    unsigned rt;
    __device__ __forceinline__ bool has(unsigned bit) const { return CFG >= 0 ? ((unsigned)CFG & bit) != 0 : (rt & bit) != 0; }
    __device__ __forceinline__ int kind() const { return CFG >= 0 ? (CFG >> kKindShift) & 7 : (int)(rt >> kKindShift) & 7; }
};

using D2V = double __attribute__((ext_vector_type(2)));

// what a lane loads for one 128-point run of its wave: issued for all runs BEFORE the record images are staged,
// so that the wave pays one memory round trip (loads placed behind the LDS hand-off would wait for it)
struct RunLoads {
    float2 rv;          // range pair
    float4 tf;          // float32 (cos, sin) of the two beams (tabf given)
    D2V t0v, t1v;       // float64 (cos, sin) (float64 kernels, or no tabf)
    int local, q;
    bool ok;
};

template <typename OutT>
__device__ __forceinline__ RunLoads flat_loads(const FlatArgs &L, const FlatBatch &bt, const int lane, const int sA,
                                               const int l0)
{
    constexpr bool kF32 = sizeof(OutT) == 4;
    RunLoads in;
    int local = l0 + lane;
    const int q = local >= L.halfN ? 1 : 0;               // 64 <= halfN: at most one wrap inside a run
    local -= q ? L.halfN : 0;
    in.local = local;
    in.q = q;
    in.ok = sA + q < bt.B;
    const unsigned i0 = 2u * (unsigned)local;
    // range pair: clamped row, unconditional (a load inside `if (ok)` is waited for before the next is issued)
    const GLOBAL_AS float *rowA = (const GLOBAL_AS float *)bt.ranges + (long long)sA * bt.sample_stride;
    const unsigned roff = (q && in.ok ? (unsigned)bt.sample_stride : 0u) + i0;
    const F2V rr = *(const GLOBAL_AS F2V *)(rowA + roff);
    in.rv = make_float2(rr.x, rr.y);
    in.tf = make_float4(0.f, 0.f, 0.f, 0.f);
    in.t0v = D2V{0.0, 0.0};
    in.t1v = D2V{0.0, 0.0};
    if (kF32 && L.tabf) {
        in.tf = *reinterpret_cast<const float4 *>(L.tabf + 2u * i0);
    } else {
        const double *tab_cs = L.tab + L.N;
        in.t0v = *reinterpret_cast<const D2V *>(tab_cs + 2u * i0);
        in.t1v = *reinterpret_cast<const D2V *>(tab_cs + 2u * i0 + 2);
    }
    return in;
}

// one 128-point run of a wave: pair index o2 (flat, within the batch) = pair0 + lane
template <typename OutT, int CFG>
__device__ __forceinline__ void flat_run(const FlatArgs &L, const FlatBatch &bt, const FlatCfg<CFG> cfg,
                                         const unsigned char *lds, const int lane, const int sA, const int l0,
                                         const long long pair0, const RunLoads &in)
{
    constexpr bool kF32 = sizeof(OutT) == 4;
    const int halfN = L.halfN, N = L.N;
    const int local = in.local, q = in.q;
    const int b = sA + q;
    const bool ok = in.ok;
    const int i0 = 2 * local;
    const float2 rv = in.rv;
    const double *tab_cs = L.tab + N;
    float csf[2], snf[2];
    const D2V t0v = in.t0v, t1v = in.t1v;
    if (kF32 && L.tabf) {
        csf[0] = in.tf.x; snf[0] = in.tf.y; csf[1] = in.tf.z; snf[1] = in.tf.w;
    } else {
        csf[0] = (float)t0v.x; snf[0] = (float)t0v.y; csf[1] = (float)t1v.x; snf[1] = (float)t1v.y;
    }
    const float r[2] = {rv.x, rv.y};
    F2V pxf = {r[0] * csf[0], r[1] * csf[1]}, pyf = {r[0] * snf[0], r[1] * snf[1]};
    // float64 cos / sin are needed on the rare exact paths only: re-read there (L1 / L2 hits) instead of
    // holding 8 registers across the whole body
    auto cs64 = [&](int k) { return tab_cs[2u * (unsigned)(i0 + k)]; };
    auto sn64 = [&](int k) { return tab_cs[2u * (unsigned)(i0 + k) + 1]; };
    auto p64x = [&](int k) { return (double)r[k] * cs64(k); };
    auto p64y = [&](int k) { return (double)r[k] * sn64(k); };
    const unsigned char *hdr = lds + q * kImgBytes;       // this lane's sample: its flat image

    if (cfg.has(kOutXy) && ok) {
        OutT *xy = static_cast<OutT *>(bt.xy) + 4 * pair0;
        store2<OutT>(xy, 2u * (unsigned)lane, p64x(0), p64y(0));
        store2<OutT>(xy, 2u * (unsigned)lane + 1, p64x(1), p64y(1));
    }

    // ---- rigid-motion flow ---------------------------------------------------
    if (cfg.has(kOutFlow)) {
        const int kind = cfg.kind();
        if (kF32 && kind != 1) {   // kind 1 subtracts the point from its moved image: float64 only
            const float4 ma = *reinterpret_cast<const float4 *>(hdr);
            const float4 mb = *reinterpret_cast<const float4 *>(hdr + 16);
            float fxs[2], fys[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float fx, fy;
                apply_motion_t<float>(kind, pxf[k], pyf[k], ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, fx, fy);
                if (cfg.has(kCanonical)) {
                    const float gx = fmaf(csf[k], fx, -snf[k] * fy);
                    const float gy = fmaf(snf[k], fx, csf[k] * fy);
                    fx = gx;
                    fy = gy;
                }
                fxs[k] = fx;
                fys[k] = fy;
            }
            if (ok) stream_f4(reinterpret_cast<float *>(bt.flow) + 4 * pair0, (unsigned)lane, fxs[0], fys[0], fxs[1], fys[1]);
        } else {
            // float64 outputs (and float32 kind 1): the reference's operation order in float64
            const double *mot = reinterpret_cast<const double *>(hdr + kImgMot);
            const double m0 = mot[0], m1 = mot[1], m2 = mot[2], m3 = mot[3];
            const double tr0 = mot[4], tr1 = mot[5], dph = mot[6];
            OutT *fl = static_cast<OutT *>(bt.flow) + 4 * pair0;
            double cs[2], sn[2];
            if (kF32 && L.tabf) {
                cs[0] = cs64(0); sn[0] = sn64(0); cs[1] = cs64(1); sn[1] = sn64(1);
            } else {
                cs[0] = t0v.x; sn[0] = t0v.y; cs[1] = t1v.x; sn[1] = t1v.y;
            }
            double f[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                double fx, fy;
                apply_motion(kind, (double)r[k] * cs[k], (double)r[k] * sn[k], m0, m1, m2, m3, tr0, tr1, dph, fx, fy);
                if (cfg.has(kCanonical)) {
                    // einsum('ijk,ik->ij'): c*fx + (-s)*fy ; s*fx + c*fy, no fusion
                    const double gx = cs[k] * fx + (-sn[k]) * fy;
                    const double gy = sn[k] * fx + cs[k] * fy;
                    fx = gx;
                    fy = gy;
                }
                f[2 * k] = fx;
                f[2 * k + 1] = fy;
            }
            if (ok) {
                if (kF32) {
                    stream_f4(reinterpret_cast<float *>(fl), (unsigned)lane, (float)f[0], (float)f[1], (float)f[2], (float)f[3]);
                } else {
                    D2V v0 = {f[0], f[1]}, v1 = {f[2], f[3]};
                    GLOBAL_AS D2V *dst = (GLOBAL_AS D2V *)fl + 2u * (unsigned)lane;
                    __builtin_nontemporal_store(v0, dst);
                    __builtin_nontemporal_store(v1, dst + 1);
                }
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
    if (cfg.has(kHasDets)) {
        const bool want_assoc = cfg.has(kOutClosest) || cfg.has(kOutCls) || cfg.has(kOutReg);
        const bool want_dyn = cfg.has(kOutDyn) || cfg.has(kOutExcl);
        const int4 h1 = *reinterpret_cast<const int4 *>(hdr + 16);      // {t0, t1, dphi, count}
        const int2 h2 = *reinterpret_cast<const int2 *>(hdr + 32);      // class bytes
        const int2 h3 = *reinterpret_cast<const int2 *>(hdr + 40);      // far flags
        const int nd = ok ? h1.w : 0;
        const int n_in = min(nd, kInline);
        // uniform trip count: the larger of the two samples' counts; the shorter sample's slots beyond its
        // count hold the null entry
        const int nA = l0 < halfN ? reinterpret_cast<const int *>(lds)[7] : 0;
        const int nB = l0 + 63 >= halfN ? reinterpret_cast<const int *>(lds + kImgBytes)[7] : 0;
        const int n_loop = __builtin_amdgcn_readfirstlane(min(max(nA, nB), kInline));
        F2V mind = {INFINITY, INFINITY};
        const unsigned char *ent = hdr + kImgScreen;
        for (int j = 0; j < n_loop; ++j) {
            const float4 e = *reinterpret_cast<const float4 *>(ent + 16 * j);
            const F2V cx2 = {e.x, e.x}, cy2 = {e.y, e.y}, ns2 = {e.z, e.z};
            const F2V ex = pxf - cx2, ey = pyf - cy2;
            F2V d = __builtin_elementwise_fma(ey, ey, ns2);
            d = __builtin_elementwise_fma(ex, ex, d);
            mind = __builtin_elementwise_min(mind, d);
        }
        const bool samp_far = (h3.x | h3.y) != 0;
        int bidx[2] = {0, 0};
        int dfirst = 0;
        auto cls_of = [&](int j) { return ((j < 4 ? h2.x : h2.y) >> (8 * (j & 3))) & 3; };
        auto sel3 = [&](const double *v, int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : v[2]); };
        // winner so far: 1-based index; 0 = the prepended zero column.  dist - radius of a candidate is
        // negative, so the first candidate beats the zero column without a square root; only a second
        // candidate of the same point (two overlapping discs: rare) needs the two values, and then the
        // incumbent's are recomputed from its index with the same operations.
        auto candidate = [&](int k, int j1, double s2, double ra) {
            if (bidx[k] != 0) {
                double wx, wy, wra;
                if (bidx[k] <= kInline) {
                    const D2V c = *reinterpret_cast<const D2V *>(hdr + kImgCtr + 16 * (bidx[k] - 1));
                    wx = c.x; wy = c.y; wra = sel3(L.ra, cls_of(bidx[k] - 1));
                } else {
                    const GLOBAL_AS double *w = (const GLOBAL_AS double *)bt.ws_det + (long long)(dfirst + bidx[k] - 1) * kDetStride;
                    wx = w[0]; wy = w[1]; wra = w[2];
                }
                const double exd = p64x(k) - wx, eyd = p64y(k) - wy;
                if (!(sqrt(s2) - ra < sqrt(exd * exd + eyd * eyd) - wra)) return;
            }
            bidx[k] = j1;
        };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool exk = samp_far || !(fabsf(pxf[k]) + fabsf(pyf[k]) < 100.0f);   // NaN -> exact
            bool need_dyn = false;
            if (!exk && mind[k] <= -L.m_dyn) dmask[k] = 0.0f;
            else need_dyn = want_dyn && (exk || mind[k] <= L.m_dyn);
            bool need_assoc = want_assoc && (exk || mind[k] <= L.thr_assoc);
            if (need_assoc && !exk) {
                // a point near an association disc (a few lanes of most waves): which detection?  The same float32
                // d_j as above against the detection's own sa_j - sd_j (entry.w): d_j <= w - m surely a candidate,
                // d_j > w + m surely not.  One sure candidate and nothing in the error band decides the point without
                // any float64 arithmetic; anything else (overlapping discs, a band) takes the exact loop below.
                int ncand = 0, jc = 0;
                bool unsure = false;
                for (int j = 0; j < n_in; ++j) {
                    const float4 e = *reinterpret_cast<const float4 *>(ent + 16 * j);
                    const float exf = pxf[k] - e.x, eyf = pyf[k] - e.y;
                    const float d = fmaf(exf, exf, fmaf(eyf, eyf, e.z));
                    if (d <= e.w + L.m_assoc) {
                        if (d <= e.w - L.m_assoc) { ++ncand; jc = j; }
                        else unsure = true;
                    }
                }
                if (!unsure && ncand <= 1) {
                    if (ncand == 1) bidx[k] = jc + 1;
                    need_assoc = false;
                }
            }
            if (need_dyn || need_assoc) {
                const double pxd = p64x(k), pyd = p64y(k);
                for (int j = 0; j < n_in; ++j) {
                    const D2V c = *reinterpret_cast<const D2V *>(hdr + kImgCtr + 16 * j);
                    const int cl = cls_of(j);
                    const double exd = pxd - c.x, eyd = pyd - c.y;
                    const double s2 = exd * exd + eyd * eyd;   // cdist: (dx*dx) + (dy*dy), no fusion
                    if (need_dyn && s2 <= sel3(L.sd, cl)) dmask[k] = 0.0f;                 // dist <= dyn radius
                    if (need_assoc && s2 <= sel3(L.sa, cl)) candidate(k, j + 1, s2, sel3(L.ra, cl));   // dist < assoc radius
                }
            }
        }
        if (nd > kInline) {
            // crowded sample: detections kInline.. from the CSR table (every lane of the sample reads the
            // same rows: L1 / L2 hits), exact test only
            dfirst = ((const GLOBAL_AS int32_t *)bt.det_offsets)[b];
            for (int j = kInline; j < nd; ++j) {
                const GLOBAL_AS double *w = (const GLOBAL_AS double *)bt.ws_det + (long long)(dfirst + j) * kDetStride;
                const double cx = w[0], cy = w[1], ra = w[2], sd = w[3], sa = w[5];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double exd = p64x(k) - cx, eyd = p64y(k) - cy;
                    const double s2 = exd * exd + eyd * eyd;
                    if (s2 <= sd) dmask[k] = 0.0f;
                    if (want_assoc && s2 <= sa) candidate(k, j + 1, s2, ra);
                }
            }
        }
        if (ok && want_assoc) {
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
                        const D2V c = *reinterpret_cast<const D2V *>(hdr + kImgCtr + 16 * (bidx[k] - 1));
                        wx = c.x; wy = c.y;
                        const int cl = cls_of(bidx[k] - 1);
                        cls[k] = cl == 0 ? L.lb[0] : (cl == 1 ? L.lb[1] : L.lb[2]);
                    } else {
                        const GLOBAL_AS double *w = (const GLOBAL_AS double *)bt.ws_det + (long long)(dfirst + bidx[k] - 1) * kDetStride;
                        wx = w[0];
                        wy = w[1];
                        cls[k] = (long long)w[4];
                    }
                    const double c = cs64(k), sn_ = sn64(k);
                    gx[k] = (float)(wy * c - wx * sn_);
                    gy[k] = (float)((wx * c + wy * sn_) - (double)r[k]);
                }
            }
            if (cfg.has(kOutClosest)) stream_ll2(reinterpret_cast<long long *>(bt.closest) + 2 * pair0, (unsigned)lane, bidx[0], bidx[1]);
            if (cfg.has(kOutCls)) stream_ll2(reinterpret_cast<long long *>(bt.target_cls) + 2 * pair0, (unsigned)lane, cls[0], cls[1]);
            if (cfg.has(kOutReg)) stream_f4(bt.target_reg + 4 * pair0, (unsigned)lane, gx[0], gy[0], gx[1], gy[1]);
        }
    }

    if (ok) {
        if (cfg.has(kOutDyn)) stream_f2(bt.dyn_mask + 2 * pair0, (unsigned)lane, dmask[0], dmask[1]);
        if (cfg.has(kOutValid)) stream_f2(bt.valid_mask + 2 * pair0, (unsigned)lane, vmask[0], vmask[1]);
        if (cfg.has(kOutExcl)) stream_f2(bt.exclude_mask + 2 * pair0, (unsigned)lane, dmask[0] * vmask[0], dmask[1] * vmask[1]);
    }
}

// grid = main_blocks (THREADS / 64 wave-chunks each, CPW runs of 128 points per wave-chunk) + the blocks that
// run the params jobs of the NEXT batches (chained form; params blocks last: dispatched first, their long
// sincos chains would hold the slots the streaming waves need)
// A slot of an array inside the kernel arguments, chosen at run time: indexing the by-value argument itself makes
// the compiler copy the whole array to scratch; reading the slot's words through the kernarg segment pointer
// (constant address space, uniform offset) is a handful of scalar loads.
typedef const __attribute__((address_space(4))) unsigned char *kernarg_ptr;
template <typename S>
__device__ __forceinline__ S kernarg_struct(kernarg_ptr p)
{
    static_assert(sizeof(S) % 8 == 0, "kernarg slots are read in 8-byte words");
    S out;
    unsigned long long *w = reinterpret_cast<unsigned long long *>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S) / 8); ++i)
        w[i] = *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long *>(p + 8 * i);
    return out;
}
constexpr size_t kFlatArgsKernargBytes = (sizeof(FlatArgs) + 7) & ~(size_t)7;   // offset of the second argument

template <typename OutT, int CFG, int THREADS, int CPW>
__global__ __launch_bounds__(THREADS, 8) void scan_flat_kernel(const FlatArgs L, const ParamsMulti P)
{
    constexpr int WAVES = THREADS / 64;
    __shared__ __align__(16) unsigned char smem[WAVES][kWaveLds];
    kernarg_ptr ka = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    // Role of this block.  The params blocks (P.total of them) sit at block indices [P.first, P.first + P.total):
    // behind the streaming blocks they would form the launch's tail (their chains -- offsets -> detection ->
    // sincos -> stores -- are two dependent memory round trips long), in front of them they would hold the slots the
    // first streaming waves need; in the middle their latency hides under the streaming waves around them.
    int sblock = (int)blockIdx.x;
    if (sblock >= P.first) {
        if (sblock < P.first + P.total) {
#ifndef POF_FLAT_NO_PARAMS
            // a params block belongs to ONE batch (uniform slot index -> scalar loads of its arguments)
            const int blk = sblock - P.first;
            int k = 0, b0 = P.blk0[0];
            if (P.uni_blocks) {      // equal batches (a loader's ring): the slot is a division, not a search
                k = (int)(__umulhi((unsigned)blk, P.uni_m) >> P.uni_s);
                b0 = k * P.uni_blocks;
            } else {
#pragma unroll
                for (int i = 1; i < kMaxSlots; ++i) k += (blk >= P.blk0[i]) ? 1 : 0;
                k = __builtin_amdgcn_readfirstlane(k);
#pragma unroll
                for (int i = 1; i < kMaxSlots; ++i) b0 = (k >= i) ? P.blk0[i] : b0;
            }
            const auto *pa = reinterpret_cast<const __attribute__((address_space(4))) PreArgs *>(
                ka + kFlatArgsKernargBytes + offsetof(ParamsMulti, p) + (size_t)k * sizeof(PreArgs));
            params_work<false, false>(pa, (blk - b0) * THREADS + (int)threadIdx.x);   // jobs past the batch's count do nothing
#endif
            return;
        }
        sblock -= P.total;
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const int w = sblock * WAVES + wv;
    if (w >= L.total_waves) return;
    // batch of this wave-chunk: a division when the batches are equal (a loader's ring), else the search; wave0[i] of
    // the unused slots is INT_MAX.  (More slots per launch were tried -- kernel arguments of 12 KB launch fine,
    // tools/ubench/kernarg_size.hip -- and gain nothing: with more than ~1 GB of resident ring slots (2 x 20 batches:
    // 2.4 GB) the step goes from 11 to 14-15 us whatever the launch shape; profiles/r3_headline_ring.txt)
    int k = 0;
    if (L.uni_waves) {
        k = (int)(__umulhi((unsigned)w, L.uni_m) >> L.uni_s);
    } else {
#pragma unroll
        for (int i = 1; i < kMaxSlots; ++i) k += (w >= L.wave0[i]) ? 1 : 0;
    }
    k = __builtin_amdgcn_readfirstlane(k);
    const FlatBatch bt = kernarg_struct<FlatBatch>(ka + offsetof(FlatArgs, b) + (size_t)k * sizeof(FlatBatch));
    const FlatCfg<CFG> cfg{L.flags};
    unsigned char *lds = smem[wv];

    // pairs [p0, p0 + 64 * CPW) of the batch's flat pair axis
    const unsigned p0 = (unsigned)(w - bt.wave0) * (64u * CPW);
    const int sA = (int)(__umulhi(p0, L.magic_m) >> L.magic_s);      // p0 / halfN
    const int l0 = (int)(p0 - (unsigned)sA * (unsigned)L.halfN);

    // run c starts 64 * c pairs further; addressed relative to sA (a run that begins in sA + 1 already has q = 1
    // in every lane; it cannot reach sA + 2: N >= 128 * CPW)
    RunLoads in[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) in[c] = flat_loads<OutT>(L, bt, lane, sA, l0 + 64 * c);

    if (cfg.has(kOutFlow) || cfg.has(kHasDets)) {
        // flat images of samples sA and sA + 1 -> this wave's LDS slice, a plain copy of 2 x 384 contiguous
        // bytes: lanes 0..23 sample sA, lanes 32..55 sample sA + 1, 16 bytes each
        {
            const int sq = lane >> 5, i = lane & 31;
            if (i < kImgBytes / 16) {
                const double *rec = bt.ws_rec + (long long)min(sA + sq, bt.B - 1) * kRecStride;
                *reinterpret_cast<F4V *>(lds + sq * kImgBytes + 16 * i) = ((const GLOBAL_AS F4V *)(rec + kRecImg))[i];
            }
        }
        // same-wave LDS hand-off: the hardware keeps a wave's LDS operations in order; the fences keep the
        // compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c)
        flat_run<OutT, CFG>(L, bt, cfg, lds, lane, sA, l0 + 64 * c, (long long)p0 + 64 * c, in[c]);
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
        if (blockIdx.x == 0) params_work<false, true>(&nx, ((int)blockIdx.y - main_rows) * kThreads + threadIdx.x);
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

float f32_round_up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, INFINITY);
    return f;
}

// tuning knobs of the flat kernel, POF_FLAT_TUNE="threads,cpw[,ppos]": threads per workgroup, 128-point runs per
// wave, position of the params blocks in the grid in percent of the streaming blocks (0 first ... 100 last)
void flat_tuning(int *threads, int *cpw, int *ppos)
{
    static int t = 0, c = 0, pp = 0;
    if (t == 0) {
        int tt = 128, cc = 1, qq = 50;
        if (const char *e = std::getenv("POF_FLAT_TUNE")) {
            int a = 0, b = 0, q = 50;
            const int n = std::sscanf(e, "%d,%d,%d", &a, &b, &q);
            if (n >= 2 && (a == 64 || a == 128 || a == 256) && (b == 1 || b == 2)) {
                tt = a;
                cc = b;
                if (n == 3 && q >= 0 && q <= 100) qq = q;
            }
        }
        c = cc;
        pp = qq;
        t = tt;
    }
    *threads = t;
    *cpw = c;
    *ppos = pp;
}

struct FlatCommon {
    const double *tab;
    const float *tabf;
    int N, flow_kind, canonical, out_f64;
    const double *assoc_radius, *dyn_radius;
    const int32_t *labels;
};

// can the flat kernel take this shape?  (pairs of beams per lane, a wave's points within two samples)
bool flat_shape_ok(int N, int cpw) { return N % 2 == 0 && N >= 128 * cpw; }

unsigned flat_flags(const FlatBatch &b, const FlatCommon &c)
{
    unsigned f = 0;
    if (b.xy) f |= kOutXy;
    if (b.flow) f |= kOutFlow;
    if (b.det_offsets) f |= kHasDets;
    if (b.closest) f |= kOutClosest;
    if (b.target_cls) f |= kOutCls;
    if (b.target_reg) f |= kOutReg;
    if (b.dyn_mask) f |= kOutDyn;
    if (b.valid_mask) f |= kOutValid;
    if (b.exclude_mask) f |= kOutExcl;
    if (c.canonical) f |= kCanonical;
    f |= (unsigned)c.flow_kind << kKindShift;
    return f;
}

template <typename OutT, int CFG>
void flat_dispatch(int threads, int cpw, dim3 grid, hipStream_t s, const FlatArgs &L, const ParamsMulti &P)
{
#define POF_FLAT_CASE(T, C) \
    if (threads == T && cpw == C) { scan_flat_kernel<OutT, CFG, T, C><<<grid, T, 0, s>>>(L, P); return; }
    POF_FLAT_CASE(64, 1)
    POF_FLAT_CASE(64, 2)
    POF_FLAT_CASE(128, 1)
    POF_FLAT_CASE(128, 2)
    POF_FLAT_CASE(256, 1)
    POF_FLAT_CASE(256, 2)
#undef POF_FLAT_CASE
}

// fill the params job set of one "next" batch; returns a status
int bind_next(const pof_scan_inputs *next, PreArgs &nx, int *jobs)
{
    *jobs = 0;
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
        *jobs = params_job_count(nx.B, nx.det_offsets != nullptr);
    }
    return POF_OK;
}

// x / d = mulhi(x, m) >> sh for 0 <= x < 2^31 and 2 <= d < 2^31 (m = ceil(2^(31 + L) / d), L = ceil(log2 d), sh = L - 1)
void magic_u31(unsigned d, unsigned *m, int *sh)
{
    int lg = 1;
    while ((1u << lg) < d) ++lg;
    *m = (unsigned)(((1ull << (31 + lg)) + d - 1) / d);
    *sh = lg - 1;
}

// One launch: stream n_cur batches (same N, same set of outputs) and evaluate the params of n_next batches.
int launch_flat(const FlatBatch *cur, int n_cur, const FlatCommon &c, const pof_scan_inputs *const *next, int n_next,
                hipStream_t s)
{
    int threads, cpw, ppos;
    flat_tuning(&threads, &cpw, &ppos);
    if (!flat_shape_ok(c.N, cpw)) cpw = 1;
    FlatArgs L = {};
    L.tab = c.tab; L.tabf = c.tabf; L.N = c.N; L.halfN = c.N / 2;
    int lg = 0;
    while ((1u << lg) < (unsigned)L.halfN) ++lg;
    L.magic_m = (unsigned)(((1ull << (31 + lg)) + (unsigned)L.halfN - 1) / (unsigned)L.halfN);
    L.magic_s = lg - 1;
    L.flags = flat_flags(cur[0], c);
    L.nb = n_cur;
    PreArgs cc = {};
    fill_class_constants(cc, cur[0].det_offsets != nullptr, c.assoc_radius, c.labels, c.dyn_radius);
    L.ra[0] = cc.ra0; L.ra[1] = cc.ra1; L.ra[2] = cc.ra2;
    L.sd[0] = cc.sd0; L.sd[1] = cc.sd1; L.sd[2] = cc.sd2;
    L.sa[0] = cc.sa0; L.sa[1] = cc.sa1; L.sa[2] = cc.sa2;
    L.lb[0] = cc.lb0; L.lb[1] = cc.lb1; L.lb[2] = cc.lb2;
    double md = 0.0, ma = 0.0, gap = -HUGE_VAL, span = 0.0;
    for (int k = 0; k < 3; ++k) {
        md = std::fmax(md, 1e-3 + 1e-4 * std::fmax(L.sd[k], 0.0));
        ma = std::fmax(ma, 1e-3 + 1e-4 * std::fmax(std::fmax(L.sd[k], L.sa[k]), 0.0));
        gap = std::fmax(gap, L.sa[k] - L.sd[k]);
        span = std::fmax(span, std::fabs(L.sa[k] - L.sd[k]));
    }
    L.m_dyn = f32_round_up(md * (1.0 + 1e-6));
    L.m_assoc = f32_round_up(ma * 1.001 + 2.4e-7 * span);
    L.thr_assoc = f32_round_up(gap + (double)L.m_assoc + 1e-6 * std::fabs(gap));
    long long waves = 0;
    const long long pairs_per_wave = 64LL * cpw;
    for (int i = 0; i < n_cur; ++i) {
        if (flat_flags(cur[i], c) != L.flags) return POF_E_BADARG;   // one configuration per launch
        L.b[i] = cur[i];
        L.b[i].wave0 = (int)waves;
        L.wave0[i] = (int)waves;
        waves += ((long long)cur[i].B * L.halfN + pairs_per_wave - 1) / pairs_per_wave;
        if ((long long)cur[i].B * L.halfN >= (1LL << 31) || waves >= (1LL << 30)) return POF_E_SHAPE;
    }
    for (int i = n_cur; i < kMaxSlots; ++i) L.wave0[i] = 0x7fffffff;
    L.total_waves = (int)waves;
    L.uni_waves = 0; L.uni_m = 0; L.uni_s = 0;
    if (n_cur > 0 && waves % n_cur == 0) {
        const long long per = waves / n_cur;
        bool same = per > 1;
        for (int i = 0; i < n_cur && same; ++i) same = L.wave0[i] == (int)(per * i);
        if (same) { L.uni_waves = (int)per; magic_u31((unsigned)per, &L.uni_m, &L.uni_s); }
    }
    const int wpb = threads / 64;
    L.main_blocks = (int)((waves + wpb - 1) / wpb);
    ParamsMulti P = {};
    P.nb = 0;
    int extra = 0;
    for (int i = 0; i < n_next; ++i) {
        PreArgs nx = {};
        int jobs = 0;
        const int rc = bind_next(next[i], nx, &jobs);
        if (rc != POF_OK) return rc;
        if (jobs == 0) continue;
        P.blk0[P.nb] = extra;
        P.p[P.nb] = nx;
        extra += (jobs + threads - 1) / threads;
        ++P.nb;
    }
    for (int i = P.nb; i <= kMaxSlots; ++i) P.blk0[i] = extra;
    P.total = extra;
    P.uni_blocks = 0; P.uni_m = 0; P.uni_s = 0;
    if (P.nb > 0 && extra % P.nb == 0) {
        const int per = extra / P.nb;
        bool same = per > 1;
        for (int i = 0; i < P.nb && same; ++i) same = P.blk0[i] == per * i;
        if (same) { P.uni_blocks = per; magic_u31((unsigned)per, &P.uni_m, &P.uni_s); }
    }
    P.first = (int)((long long)L.main_blocks * ppos / 100);
    if (L.main_blocks + extra == 0) return POF_OK;
    dim3 grid((unsigned)(L.main_blocks + extra));
    const bool headline = (L.flags == (unsigned)kCfgHeadline);
    if (c.out_f64) {
        if (headline) flat_dispatch<double, kCfgHeadline>(threads, cpw, grid, s, L, P);
        else flat_dispatch<double, -1>(threads, cpw, grid, s, L, P);
    } else {
        if (headline) flat_dispatch<float, kCfgHeadline>(threads, cpw, grid, s, L, P);
        else flat_dispatch<float, -1>(threads, cpw, grid, s, L, P);
    }
    return POF_OK;
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
        const int rc = bind_next(next, nx, &next_jobs);
        if (rc != POF_OK) return rc;
    }

    // float2 row loads need 8-byte aligned rows, 16-byte stores aligned outputs
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool vec2 = (N % 2 == 0) && (sample_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(ranges) & 7) == 0) &&
                      al16(xy) && al16(flow) && al16(closest) && al16(target_cls) && al16(target_reg) &&
                      al16(dyn_mask) && al16(valid_mask) && al16(exclude_mask);
    bool chained = false;
    if (vec2 && flat_shape_ok(N, 1) && (long long)B * (N / 2) < (1LL << 31)) {
        // flat form: line-aligned output runs per wave (see scan_flat_kernel)
        FlatBatch fb = {};
        fb.ranges = ranges; fb.sample_stride = sample_stride; fb.xy = xy; fb.flow = flow;
        fb.closest = closest; fb.target_cls = target_cls; fb.target_reg = target_reg;
        fb.dyn_mask = dyn_mask; fb.valid_mask = valid_mask; fb.exclude_mask = exclude_mask;
        fb.det_offsets = det_offsets; fb.ws_rec = a.ws_rec; fb.ws_det = a.ws_det; fb.B = B;
        FlatCommon fc = {tab, nullptr, N, flow_kind, canonical, out_f64, assoc_radius, dyn_radius, labels};
        const pof_scan_inputs *nl[1] = {next};
        const int rc = launch_flat(&fb, 1, fc, nl, (next && next_jobs > 0) ? 1 : 0, s);
        if (rc != POF_OK) return rc;
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

extern "C" int pof_scan_preprocess_multi(const pof_scan_batch *cur, int n_cur, const pof_scan_inputs *next,
                                         int n_next, int N, const double *tab, const float *tab_cs_f32,
                                         int flow_kind, int canonical, int out_f64, const double *assoc_radius,
                                         const int32_t *labels, const double *dyn_radius, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (n_cur < 0 || n_next < 0 || n_cur > POF_SCAN_MAX_SLOTS || n_next > POF_SCAN_MAX_SLOTS) return POF_E_BADARG;
    if (n_cur + n_next == 0) return POF_E_BADARG;
    if ((n_cur > 0 && !cur) || (n_next > 0 && !next)) return POF_E_BADARG;
    if (n_cur > 0 && (!tab || N < 1)) return POF_E_BADARG;
    if (flow_kind < 0 || flow_kind > 4) return POF_E_BADARG;
    if (n_cur > 0 && !flat_shape_ok(N, 1)) return POF_E_SHAPE;   // odd or short scans: pof_scan_preprocess_chained
    FlatBatch fb[kMaxSlots] = {};
    for (int i = 0; i < n_cur; ++i) {
        const pof_scan_batch &c = cur[i];
        if (!c.ranges || c.B < 0 || c.D < 0) return POF_E_BADARG;
        if (c.det_offsets && (!assoc_radius || !labels || !dyn_radius)) return POF_E_BADARG;
        if (c.sample_stride < N || (c.sample_stride & 1)) return POF_E_SHAPE;
        auto al = [](const void *q, uintptr_t m) { return (reinterpret_cast<uintptr_t>(q) & m) == 0; };
        if (!al(c.ranges, 7) || !al(c.xy, 15) || !al(c.flow, 15) || !al(c.closest, 15) || !al(c.target_cls, 15) ||
            !al(c.target_reg, 15) || !al(c.dyn_mask, 15) || !al(c.valid_mask, 15) || !al(c.exclude_mask, 15))
            return POF_E_SHAPE;
        const bool need_ws = c.flow || c.det_offsets;
        if (need_ws && (!c.workspace || c.workspace_bytes < pof_scan_preprocess_workspace_bytes(c.B, c.det_offsets ? c.D : 0)))
            return POF_E_WORKSPACE;
        PreArgs w = {};
        bind_workspace(w, c.workspace, c.B);
        fb[i].ranges = c.ranges; fb[i].sample_stride = c.sample_stride; fb[i].xy = c.xy; fb[i].flow = c.flow;
        fb[i].closest = c.closest; fb[i].target_cls = c.target_cls; fb[i].target_reg = c.target_reg;
        fb[i].dyn_mask = c.dyn_mask; fb[i].valid_mask = c.valid_mask; fb[i].exclude_mask = c.exclude_mask;
        fb[i].det_offsets = c.det_offsets; fb[i].ws_rec = w.ws_rec; fb[i].ws_det = w.ws_det; fb[i].B = c.B;
    }
    // empty batches stream nothing
    int n_live = 0;
    for (int i = 0; i < n_cur; ++i)
        if (fb[i].B > 0) fb[n_live++] = fb[i];
    const pof_scan_inputs *nl[kMaxSlots] = {};
    for (int i = 0; i < n_next; ++i) nl[i] = next + i;
    FlatCommon fc = {tab, tab_cs_f32, N, flow_kind, canonical, out_f64, assoc_radius, dyn_radius, labels};
    if (n_live == 0) {
        // params only: a plain params launch per batch
        for (int i = 0; i < n_next; ++i) {
            PreArgs nx = {};
            int jobs = 0;
            const int rc = bind_next(nl[i], nx, &jobs);
            if (rc != POF_OK) return rc;
            if (jobs > 0) {
                scan_params_kernel<<<(jobs + 255) / 256, 256, 0, pof_stream(stream)>>>(nx);
                POF_CHECK_LAUNCH();
            }
        }
        return POF_OK;
    }
    const int rc = launch_flat(fb, n_live, fc, nl, n_next, pof_stream(stream));
    if (rc != POF_OK) return rc;
    POF_CHECK_LAUNCH();
    return POF_OK;
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
