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
#include <type_traits>

#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kDetTile = 64;  // detections staged in LDS per pass

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
    int B, N;
    const double *tab;
    const double *odom0, *odom1;
    int flow_kind, canonical;
    void *xy, *flow;
    const int32_t *det_offsets;
    const double *det_rphi;
    const uint8_t *det_cls;
    double assoc_radius[3];
    int32_t labels[3];
    double dyn_radius[3];
    int64_t *closest, *target_cls;
    float *target_reg, *dyn_mask, *valid_mask, *exclude_mask;
};

// Per-sample rigid motion, evaluated once per workgroup (lane 0) into LDS.
// mot[0..3] = 2x2 matrix (row major), mot[4..5] = translation, mot[6] = dphi.
__device__ void motion_params(int kind, const double *o0, const double *o1, double *mot)
{
    if (kind == 0) {
        // get_displacement_from_odometry: R0, R1 stored in float32; the float32
        // 2x2 product and the float64 products below use the FMA order of the
        // BLAS the reference calls (sgemm/dgemm/gemv: fma(a1, b1, a0*b0)).
        float c0 = (float)cos(o0[2]), s0 = (float)sin(o0[2]);
        float c1 = (float)cos(o1[2]), s1 = (float)sin(o1[2]);
        // A = R0^T = [[c0, s0], [-s0, c0]],  R1 = [[c1, -s1], [s1, c1]]
        float p00 = fmaf(s0, s1, c0 * c1);
        float p01 = fmaf(s0, c1, c0 * (-s1));
        float p10 = fmaf(c0, s1, (-s0) * c1);
        float p11 = fmaf(c0, c1, (-s0) * (-s1));
        mot[0] = 1.0 - (double)p00;
        mot[1] = 0.0 - (double)p01;
        mot[2] = 0.0 - (double)p10;
        mot[3] = 1.0 - (double)p11;
        double tx = o1[0] - o0[0], ty = o1[1] - o0[1];
        mot[4] = fma((double)c0, tx, (double)s0 * ty);
        mot[5] = fma((double)(-s0), tx, (double)c0 * ty);
    } else if (kind == 1) {
        // get_flow_target: float64 throughout
        double s0, c0, s1, c1;
        sincos(o0[2], &s0, &c0);
        double dphi = o1[2] - o0[2];
        sincos(dphi, &s1, &c1);
        double tx = o1[0] - o0[0], ty = o1[1] - o0[1];
        mot[0] = c1; mot[1] = -s1; mot[2] = s1; mot[3] = c1;
        // trans_world @ rot_0.T
        mot[4] = fma(ty, -s0, tx * c0);
        mot[5] = fma(ty, c0, tx * s0);
    } else {
        // get_velocity_from_odometry: float32 R1, cross matrix dphi*[[0,-1],[1,0]]
        float c1 = (float)cos(o1[2]), s1 = (float)sin(o1[2]);
        double tx = o1[0] - o0[0], ty = o1[1] - o0[1];
        mot[4] = fma((double)c1, tx, (double)s1 * ty);
        mot[5] = fma((double)(-s1), tx, (double)c1 * ty);
        mot[6] = o1[2] - o0[2];
    }
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

template <typename OutT, int PTS>
__global__ __launch_bounds__(kThreads) void scan_preprocess_kernel(PreArgs a)
{
    __shared__ double s_mot[8];
    __shared__ double s_cx[kDetTile], s_cy[kDetTile], s_dr[kDetTile], s_dphi[kDetTile];
    __shared__ double s_ra[kDetTile], s_rd[kDetTile];
    __shared__ int s_lab[kDetTile];

    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int i0 = (blockIdx.x * kThreads + tid) * PTS;
    const int N = a.N;
    const bool want_flow = a.flow != nullptr;
    const bool want_assoc = a.det_offsets != nullptr;

    if (want_flow && tid == 0) motion_params(a.flow_kind, a.odom0 + 3 * b, a.odom1 + 3 * b, s_mot);

    // ---- load the points this thread owns ---------------------------------
    float r[PTS];
    double px[PTS], py[PTS], cs[PTS], sn[PTS];
    const float *row = a.ranges + (long long)b * a.sample_stride;
    const bool live = i0 < N;  // N % PTS == 0 is guaranteed by the launcher
    if (live) {
        if (PTS == 2) {
            float2 v = *reinterpret_cast<const float2 *>(row + i0);
            r[0] = v.x;
            r[PTS - 1] = v.y;
        } else {
            r[0] = row[i0];
        }
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            double2 t = *reinterpret_cast<const double2 *>(a.tab + N + 2 * (i0 + k));
            cs[k] = t.x;
            sn[k] = t.y;
            px[k] = (double)r[k] * cs[k];
            py[k] = (double)r[k] * sn[k];
        }
    }
    const long long o = (long long)b * N + i0;  // flat point index of the first owned point

    if (live && a.xy) {
        OutT *xy = static_cast<OutT *>(a.xy);
#pragma unroll
        for (int k = 0; k < PTS; ++k) store2<OutT>(xy, o + k, px[k], py[k]);
    }

    __syncthreads();

    // ---- rigid-motion flow -------------------------------------------------
    if (live && want_flow) {
        OutT *fl = static_cast<OutT *>(a.flow);
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            double fx, fy;
            if (a.flow_kind == 0) {
                fx = fma(py[k], s_mot[1], px[k] * s_mot[0]) - s_mot[4];
                fy = fma(py[k], s_mot[3], px[k] * s_mot[2]) - s_mot[5];
            } else if (a.flow_kind == 1) {
                double x1 = fma(py[k], s_mot[1], px[k] * s_mot[0]) - s_mot[4];
                double y1 = fma(py[k], s_mot[3], px[k] * s_mot[2]) - s_mot[5];
                fx = x1 - px[k];
                fy = y1 - py[k];
            } else {
                // -lin - xy @ cross^T, cross^T = [[0, dphi], [-dphi, 0]]
                fx = -s_mot[4] - (py[k] * -s_mot[6]);
                fy = -s_mot[5] - (px[k] * s_mot[6]);
            }
            if (a.canonical) {
                // einsum('ijk,ik->ij'): c*fx + (-s)*fy ; s*fx + c*fy, no fusion
                double gx = cs[k] * fx + (-sn[k]) * fy;
                double gy = sn[k] * fx + cs[k] * fy;
                fx = gx;
                fy = gy;
            }
            store2<OutT>(fl, o + k, fx, fy);
        }
    }

    // ---- valid mask (needs no detections) ----------------------------------
    float vmask[PTS], dmask[PTS];
#pragma unroll
    for (int k = 0; k < PTS; ++k) {
        vmask[k] = (live && r[k] >= 20.0f) ? 0.0f : 1.0f;
        dmask[k] = 1.0f;
    }

    // ---- association + dynamic mask ----------------------------------------
    if (want_assoc) {
        const int d0 = a.det_offsets[b], d1 = a.det_offsets[b + 1];
        double best[PTS];
        int bidx[PTS];
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            best[k] = 0.0;  // the prepended zero column
            bidx[k] = 0;
        }
        for (int base = d0; base < d1; base += kDetTile) {
            const int cnt = min(kDetTile, d1 - base);
            __syncthreads();
            if (tid < cnt) {
                double dr = a.det_rphi[2 * (base + tid)], dp = a.det_rphi[2 * (base + tid) + 1];
                double s, c;
                sincos(dp, &s, &c);
                int cl = a.det_cls[base + tid];
                cl = cl > 2 ? 2 : cl;
                s_cx[tid] = dr * c;
                s_cy[tid] = dr * s;
                s_dr[tid] = dr;
                s_dphi[tid] = dp;
                s_ra[tid] = a.assoc_radius[cl];
                s_rd[tid] = a.dyn_radius[cl];
                s_lab[tid] = a.labels[cl];
            }
            __syncthreads();
            if (live) {
                for (int j = 0; j < cnt; ++j) {
                    const double cx = s_cx[j], cy = s_cy[j], ra = s_ra[j], rd = s_rd[j];
#pragma unroll
                    for (int k = 0; k < PTS; ++k) {
                        double ex = px[k] - cx, ey = py[k] - cy;
                        double dist = sqrt(ex * ex + ey * ey);
                        double v = dist - ra;
                        if (v < best[k]) {
                            best[k] = v;
                            bidx[k] = base - d0 + j + 1;
                        }
                        if (dist <= rd) dmask[k] = 0.0f;
                    }
                }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < PTS; ++k) {
                if (a.closest) a.closest[o + k] = bidx[k];
                long long cls = 0;
                float gx = 0.0f, gy = 0.0f;
                if (bidx[k] > 0) {
                    // re-read the winning detection (rare path: a few % of points)
                    const int g = d0 + bidx[k] - 1;
                    const double dr = a.det_rphi[2 * g], dp = a.det_rphi[2 * g + 1];
                    int cl = a.det_cls[g];
                    cl = cl > 2 ? 2 : cl;
                    cls = a.labels[cl];
                    const double phi = a.tab[i0 + k];
                    double s, c;
                    sincos(dp - phi, &s, &c);
                    gx = (float)(s * dr);
                    gy = (float)(c * dr - (double)r[k]);
                }
                if (a.target_cls) a.target_cls[o + k] = cls;
                if (a.target_reg) reinterpret_cast<float2 *>(a.target_reg)[o + k] = make_float2(gx, gy);
            }
        }
    }

    if (live) {
#pragma unroll
        for (int k = 0; k < PTS; ++k) {
            if (a.dyn_mask) a.dyn_mask[o + k] = dmask[k];
            if (a.valid_mask) a.valid_mask[o + k] = vmask[k];
            if (a.exclude_mask) a.exclude_mask[o + k] = dmask[k] * vmask[k];
        }
    }
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

}  // namespace

extern "C" int pof_laser_phi(double angle_inc, int num_pts, double *tab, pof_stream_t stream)
{
    if (!tab || num_pts < 2) return POF_E_BADARG;
    // host scalar arithmetic, identical to the Python float arithmetic of the reference
    const double fov = (double)(num_pts - 1) * angle_inc;
    const double start = -fov * 0.5, stop = fov * 0.5;
    const double step = (stop - start) / (double)(num_pts - 1);
    laser_phi_kernel<<<(num_pts + 255) / 256, 256, 0, pof_stream(stream)>>>(start, stop, step, num_pts, tab);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_scan_preprocess(const float *ranges, long long sample_stride, int B, int N,
                                   const double *tab, const double *odom0, const double *odom1,
                                   int flow_kind, int canonical, int out_f64, void *xy, void *flow,
                                   const int32_t *det_offsets, const double *det_rphi,
                                   const uint8_t *det_cls, const double *assoc_radius,
                                   const int32_t *labels, const double *dyn_radius, int64_t *closest,
                                   int64_t *target_cls, float *target_reg, float *dyn_mask,
                                   float *valid_mask, float *exclude_mask, pof_stream_t stream)
{
    if (!ranges || !tab || B < 0 || N < 1) return POF_E_BADARG;
    if (flow && (!odom0 || !odom1)) return POF_E_BADARG;
    if (flow_kind < 0 || flow_kind > 2) return POF_E_BADARG;
    if (det_offsets && (!assoc_radius || !labels || !dyn_radius)) return POF_E_BADARG;
    if (sample_stride < N) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;  // grid.y limit; callers chunk larger batches
    PreArgs a;
    a.ranges = ranges; a.sample_stride = sample_stride; a.B = B; a.N = N; a.tab = tab;
    a.odom0 = odom0; a.odom1 = odom1; a.flow_kind = flow_kind; a.canonical = canonical;
    a.xy = xy; a.flow = flow; a.det_offsets = det_offsets; a.det_rphi = det_rphi; a.det_cls = det_cls;
    for (int k = 0; k < 3; ++k) {
        a.assoc_radius[k] = det_offsets ? assoc_radius[k] : 0.0;
        a.labels[k] = det_offsets ? labels[k] : 0;
        a.dyn_radius[k] = det_offsets ? dyn_radius[k] : 0.0;
    }
    a.closest = closest; a.target_cls = target_cls; a.target_reg = target_reg;
    a.dyn_mask = dyn_mask; a.valid_mask = valid_mask; a.exclude_mask = exclude_mask;
    // float2 row loads need 8-byte aligned rows
    const bool vec2 = (N % 2 == 0) && (sample_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(ranges) & 7) == 0);
    hipStream_t s = pof_stream(stream);
    if (vec2) {
        dim3 grid((N / 2 + kThreads - 1) / kThreads, B);
        if (out_f64) scan_preprocess_kernel<double, 2><<<grid, kThreads, 0, s>>>(a);
        else scan_preprocess_kernel<float, 2><<<grid, kThreads, 0, s>>>(a);
    } else {
        dim3 grid((N + kThreads - 1) / kThreads, B);
        if (out_f64) scan_preprocess_kernel<double, 1><<<grid, kThreads, 0, s>>>(a);
        else scan_preprocess_kernel<float, 1><<<grid, kThreads, 0, s>>>(a);
    }
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_rotate_flow(const void *flow_in, void *flow_out, const double *tab, int B, int N,
                               int to_canonical, int is_f64, pof_stream_t stream)
{
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
    if (!ranges || !tab || !det_r || !det_phi || !dx || !dy || B < 0 || N < 1) return POF_E_BADARG;
    long long total = (long long)B * N;
    if (total == 0) return POF_OK;
    canonical_to_det_kernel<<<(unsigned)((total + 255) / 256), 256, 0, pof_stream(stream)>>>(
        ranges, tab, dx, dy, det_r, det_phi, total, N);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_abi_version(void) { return POF_ABI_VERSION; }

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
