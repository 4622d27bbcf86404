// A13: jump-distance segmentation of a scan and per-segment least squares
//   src/depracted/model/adaboost_person_det.py:71-90   (cuts)
//   src/depracted/model/adaboost_person_det.py:102-210 (features; line fit = 2x2
//   normal equations, circle fit = 3x3 normal equations of A=[-2x,-2y,1])
//
// One workgroup per scan: points -> LDS (float64 xy), cut flags -> block scan ->
// segment ids and start offsets; then one lane per segment accumulates the
// centred moments of its (contiguous) point range and solves both fits in
// registers.  Centring the points first makes A^T A block diagonal
// ([[4Suu,4Suv,0],[4Suv,4Svv,0],[0,0,n]]), so the 3x3 solve reduces to one
// well-conditioned 2x2 solve plus a division; the result equals pinv(A) b for
// any segment with full column rank.
//
// Feature columns (float64): 0 n, 1 sigma, 2 jump_prev, 3 jump_next, 4 width,
// 5 line_residual, 6 circ_Sc, 7 radius, 8 boundary_len, 9 boundary_std,
// 10 sum_curvature, 11 mean_ang_diff, 12 line_k, 13 line_b, 14 xc, 15 yc.
// HBM: 4N read, 4N + 128*S written per scan -- latency bound, not a roofline kernel.
#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kFeat = 16;

struct SegArgs {
    const float *ranges;
    const double *tab;
    int N, max_seg;
    float jump;
    int32_t *seg_id, *num_seg;
    double *feat;
};

__device__ __forceinline__ double norm2(double x, double y) { return sqrt(x * x + y * y); }

__global__ __launch_bounds__(kThreads) void segment_kernel(SegArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N;
    double *sx = reinterpret_cast<double *>(smem);
    double *sy = sx + N;
    int *sid = reinterpret_cast<int *>(sy + N);
    int *sstart = sid + N;                 // [max_seg + 1]
    __shared__ int s_part[kThreads];
    __shared__ int s_nseg;

    const int b = blockIdx.x, tid = threadIdx.x;
    const float *r = a.ranges + (long long)b * N;
    const int per = (N + kThreads - 1) / kThreads;
    const int lo = min(tid * per, N), hi = min(lo + per, N);

    int cnt = 0;
    for (int i = lo; i < hi; ++i) {
        const double c = a.tab[N + 2 * i], s = a.tab[N + 2 * i + 1];
        sx[i] = (double)r[i] * c;
        sy[i] = (double)r[i] * s;
        const int flag = (i > 0) && (fabsf(r[i] - r[i - 1]) >= a.jump);
        sid[i] = flag;
        cnt += flag;
    }
    s_part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - cnt;  // exclusive prefix of this thread's chunk
    int32_t *gid = a.seg_id + (long long)b * N;
    for (int i = lo; i < hi; ++i) {
        const int flag = sid[i];
        run += flag;
        if ((flag || i == 0) && run <= a.max_seg) sstart[run] = i;
        sid[i] = run;
        gid[i] = run;
    }
    if (tid == kThreads - 1) s_nseg = s_part[kThreads - 1] + 1;
    __syncthreads();
    const int S = min(s_nseg, a.max_seg);
    if (tid == 0) {
        a.num_seg[b] = s_nseg;
        if (s_nseg <= a.max_seg) sstart[s_nseg] = N;
    }
    __syncthreads();

    for (int s = tid; s < S; s += kThreads) {
        const int p0 = sstart[s];
        const int p1 = (s + 1 < s_nseg) ? ((s + 1 <= a.max_seg) ? sstart[s + 1] : N) : N;
        const int n = p1 - p0;
        double *f = a.feat + ((long long)b * a.max_seg + s) * kFeat;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int c = 0; c < kFeat; ++c) f[c] = nan;
        f[0] = (double)n;
        // pass 1: mean
        double mx = 0.0, my = 0.0;
        for (int i = p0; i < p1; ++i) {
            mx += sx[i];
            my += sy[i];
        }
        mx /= (double)n;
        my /= (double)n;
        // pass 2: centred moments, boundary, curvature
        double suu = 0, svv = 0, suv = 0, suz = 0, svz = 0, sz = 0, sumx = 0, sumy = 0;
        double blen = 0, curv = 0, ang = 0;
        for (int i = p0; i < p1; ++i) {
            const double u = sx[i] - mx, v = sy[i] - my, z = u * u + v * v;
            suu += u * u;
            svv += v * v;
            suv += u * v;
            suz += u * z;
            svz += v * z;
            sz += z;
            sumx += sx[i];
            sumy += sy[i];
            if (i + 1 < p1) blen += norm2(sx[i + 1] - sx[i], sy[i + 1] - sy[i]);
            if (i + 2 < p1) {
                const double ax = sx[i], ay = sy[i], bx = sx[i + 1], by = sy[i + 1], cx = sx[i + 2], cy = sy[i + 2];
                const double dA = norm2(bx - ax, by - ay), dB = norm2(cx - bx, cy - by), dC = norm2(ax - cx, ay - cy);
                const double area = fabs(0.5 * (ax * (by - cy) + bx * (cy - ay) + cx * (ay - by)));
                curv += 4.0 * area / (dA * dB * dC);
                const double bax = ax - bx, bay = ay - by, bcx = cx - bx, bcy = cy - by;
                const double cosv = (bax * bcx + bay * bcy) / (norm2(bax, bay) * norm2(bcx, bcy));
                ang += acos(cosv);
            }
        }
        if (n > 1) f[1] = sqrt(sz) / (double)(n - 1);
        const int sp = max(s - 1, 0), sn = min(s + 1, s_nseg - 1);
        const int prev_last = (sp == s) ? p1 - 1 : p0 - 1;
        const int next_first = (sn == s) ? p0 : p1;
        f[2] = norm2(sx[prev_last] - sx[p0], sy[prev_last] - sy[p0]);
        f[3] = norm2(sx[p1 - 1] - sx[next_first], sy[p1 - 1] - sy[next_first]);
        f[4] = norm2(sx[p1 - 1] - sx[p0], sy[p1 - 1] - sy[p0]);
        f[8] = blen;
        double xc = 0, yc = 0, rc = 0;
        if (n >= 3) {
            // line y = k x + b: 2x2 normal equations in centred coordinates
            const double k = suv / suu, bb = my - k * mx;
            const double nrm = sqrt(k * k + 1.0);
            f[12] = k;
            f[13] = bb;
            f[5] = (k / nrm) * sumx + (-1.0 / nrm) * sumy - (double)n * fabs(bb / nrm);
            // circle: [[Suu,Suv],[Suv,Svv]] (uc,vc) = 0.5 (Suz,Svz); c' = -Sz/n
            const double det = suu * svv - suv * suv;
            const double uc = 0.5 * (suz * svv - svz * suv) / det;
            const double vc = 0.5 * (svz * suu - suz * suv) / det;
            rc = sqrt(uc * uc + vc * vc + sz / (double)n);
            xc = uc + mx;
            yc = vc + my;
            f[7] = rc;
            f[14] = xc;
            f[15] = yc;
            f[10] = curv;
            f[11] = ang / (double)(n - 2);
        }
        // pass 3: quantities that need the fit / the mean edge length
        if (n > 1) {
            const double me = blen / (double)(n - 1);
            double var = 0, sc = 0;
            for (int i = p0; i < p1; ++i) {
                if (i + 1 < p1) {
                    const double e = norm2(sx[i + 1] - sx[i], sy[i + 1] - sy[i]) - me;
                    var += e * e;
                }
                if (n >= 3) {
                    const double t = rc - sqrt(norm2(xc - sx[i], yc - sy[i]));
                    sc += t * t;
                }
            }
            f[9] = sqrt(var / (double)(n - 1));
            if (n >= 3) f[6] = sc;
        }
    }
}

}  // namespace

extern "C" int pof_segment_features(const float *ranges, const double *tab, int B, int N,
                                    double jump_dist, int max_seg, int32_t *seg_id, int32_t *num_seg,
                                    double *feat, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!ranges || !tab || !seg_id || !num_seg || !feat || B < 0 || N < 1 || max_seg < 1)
        return POF_E_BADARG;
    if (B == 0) return POF_OK;
    SegArgs a;
    a.ranges = ranges; a.tab = tab; a.N = N; a.max_seg = max_seg; a.jump = (float)jump_dist;
    a.seg_id = seg_id; a.num_seg = num_seg; a.feat = feat;
    const size_t lds = (size_t)N * (2 * sizeof(double) + sizeof(int)) + (size_t)(max_seg + 1) * sizeof(int);
    if (lds > 160 * 1024 - 2048) return POF_E_SHAPE;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(segment_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return POF_E_LAUNCH;
    }
    segment_kernel<<<B, kThreads, lds, pof_stream(stream)>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
