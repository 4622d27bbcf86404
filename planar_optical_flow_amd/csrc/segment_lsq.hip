// A13: jump-distance segmentation of a scan and per-segment least squares
//   src/depracted/model/adaboost_person_det.py:71-90   (cuts, labels)
//   src/depracted/model/adaboost_person_det.py:102-210 (features; line fit = 2x2
//   normal equations, circle fit = 3x3 normal equations of A=[-2x,-2y,1])
//
// One workgroup per scan.  Stage 1 (all lanes): points -> LDS (float64 xy), cut flags ->
// block scan -> segment ids and start offsets; a second block scan over the segments
// numbers the ones the reference keeps (more than two points, :53-55).
// Stage 2: ONE WAVE PER SEGMENT.  The lanes stride over the segment's (contiguous) points,
// every moment is a lane-partial sum closed by a __shfl_xor tree (wave_sum_f64), and the
// 2x2 / 3x3 normal equations are then solved in registers (every lane holds the reduced
// moments, lane 0 stores).  Centring the points first makes A^T A block diagonal
// ([[4Suu,4Suv,0],[4Suv,4Svv,0],[0,0,n]]), so the 3x3 circle solve reduces to one
// well-conditioned 2x2 solve plus a division; the result equals pinv(A) b for any segment
// with full column rank.  The per-axis median of the reference's "median deviation" is a
// rank selection inside the wave (each lane ranks its points against the segment in LDS).
//
// Two outputs:
//  feat     [B][max_seg][16], every segment: 0 n, 1 sigma, 2 jump_prev, 3 jump_next, 4 width,
//           5 line_residual, 6 circ_Sc, 7 radius, 8 boundary_len, 9 boundary_std,
//           10 sum_curvature, 11 mean_ang_diff, 12 line_k, 13 line_b, 14 xc, 15 yc
//           (jumps to the geometric neighbours).
//  ref_feat [B][max_seg][15], kept segments only, in the reference's column order and with
//           its data-set coupled definitions (compute_feature :102-210): 0 n, 1 sigma,
//           2 ||seg - median||_F / n, 3 jump to the previous KEPT segment, 4 jump to
//           kept[min(q+1, 3)] (NaN when that does not exist: the reference raises there),
//           5 width, 6 line residual, 7 Sc, 8 radius, 9 boundary length, 10 boundary std,
//           11 sum curvature, 12 mean angular difference, 13 mean speed over piece q of the
//           UNFILTERED split, 14 label.
// HBM: 4N (+4N next scan) read, 4N + 128*S (+120*K) written per scan -- latency bound.
#include "pof_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / POF_WAVE;
constexpr int kFeat = 16;
constexpr int kRef = 15;

struct SegArgs {
    const float *ranges, *next_ranges;
    const double *tab, *odom_dt, *wp_xy;
    const int32_t *wp_offsets;
    double radius_wp;
    int N, max_seg;
    int med_off;        // byte offset of the median sort scratch in dynamic LDS (reference rows only)
    float jump;
    int32_t *seg_id, *num_seg, *num_kept;
    double *feat, *ref_feat;
};

__device__ __forceinline__ double norm2(double x, double y) { return sqrt(x * x + y * y); }

// exclusive prefix of `cnt` over the workgroup; *total = sum.  s_part: kThreads ints.
__device__ __forceinline__ int block_exclusive(int cnt, int *s_part, int tid, int *total)
{
    s_part[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < kThreads; off <<= 1) {
        const int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    const int incl = s_part[tid];
    *total = s_part[kThreads - 1];
    __syncthreads();
    return incl - cnt;
}

// np.median of v[p0 .. p0+n) inside one wave.  Round 2 ranked every element against the whole segment
// (n^2 / 64 comparisons per lane: 28 000 instructions per axis for a 400-point wall, 0.9 ms per 4096 scans);
// round 3 sorts a copy in LDS with a bitonic network in its all-ascending form (the first step of every stage
// mirrors the upper half, so every compare-exchange puts the smaller value at the lower index): with that form
// the padding to a power of two is virtual -- a pair whose upper index is >= n is skipped, as if +inf sat there
// -- and the copy needs exactly the segment's own slots of a scratch row shared by the workgroup (segments are
// disjoint).  45 steps x 4 pairs per lane for n = 400.  One wave: its LDS operations are ordered, no barrier.
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_median(const double *v, double *scratch, int p0, int n, int lane)
{
    double *w = scratch + p0;
    for (int i = lane; i < n; i += POF_WAVE) w[i] = v[p0 + i];
    wave_lds_order();
    int npad = 1;
    while (npad < n) npad <<= 1;
    const int half = npad >> 1;
    auto cmpx = [&](int lo, int hi) {
        if (hi < n) {
            const double x = w[lo], y = w[hi];
            if (y < x) {
                w[lo] = y;
                w[hi] = x;
            }
        }
    };
    for (int k = 2; k <= npad; k <<= 1) {
        const int hk = k >> 1;
        for (int t = lane; t < half; t += POF_WAVE) {          // flip: i <-> block end - 1 - offset
            const int blk = t / hk, off = t - blk * hk;
            cmpx(blk * k + off, blk * k + k - 1 - off);
        }
        wave_lds_order();
        for (int j = k >> 2; j > 0; j >>= 1) {                 // half cleaners
            for (int t = lane; t < half; t += POF_WAVE) {
                const int lo = 2 * j * (t / j) + (t % j);
                cmpx(lo, lo + j);
            }
            wave_lds_order();
        }
    }
    const double m = (w[(n - 1) >> 1] + w[n >> 1]) / 2.0;      // NumPy: mean of the two middle elements
    wave_lds_order();
    return m;
}

__global__ __launch_bounds__(kThreads) void segment_kernel(SegArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N;
    double *sx = reinterpret_cast<double *>(smem);
    double *sy = sx + N;
    int *sid = reinterpret_cast<int *>(sy + N);
    int *sstart = sid + N;                 // [max_seg + 1]
    int *skidx = sstart + a.max_seg + 1;   // [max_seg] kept number of a segment (-1: dropped)
    int *skept = skidx + a.max_seg;        // [max_seg] segment of a kept number
    double *smed = reinterpret_cast<double *>(smem + a.med_off);   // [N] sort scratch of the medians (reference rows)
    __shared__ int s_part[kThreads];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & (POF_WAVE - 1);
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
    int n_cut;
    int run = block_exclusive(cnt, s_part, tid, &n_cut);
    const int nseg = n_cut + 1;
    int32_t *gid = a.seg_id + (long long)b * N;
    for (int i = lo; i < hi; ++i) {
        const int flag = sid[i];
        run += flag;
        if ((flag || i == 0) && run <= a.max_seg) sstart[run] = i;
        sid[i] = run;
        gid[i] = run;
    }
    const int S = min(nseg, a.max_seg);
    if (tid == 0) {
        a.num_seg[b] = nseg;
        if (nseg <= a.max_seg) sstart[nseg] = N;
    }
    __syncthreads();
    // segment s covers [sstart[s], seg_end(s))
    auto seg_end = [&](int s) { return (s + 1 < nseg && s + 1 <= a.max_seg) ? sstart[s + 1] : N; };

    // number the kept segments (more than two points)
    const int sper = (S + kThreads - 1) / kThreads;
    const int slo = min(tid * sper, S), shi = min(slo + sper, S);
    int kc = 0;
    for (int s = slo; s < shi; ++s) kc += (seg_end(s) - sstart[s]) > 2;
    int K;
    int q = block_exclusive(kc, s_part, tid, &K);
    for (int s = slo; s < shi; ++s) {
        const bool keep = (seg_end(s) - sstart[s]) > 2;
        skidx[s] = keep ? q : -1;
        if (keep) skept[q++] = s;
    }
    if (tid == 0 && a.num_kept) a.num_kept[b] = K;
    __syncthreads();

    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const float *rn = a.next_ranges ? a.next_ranges + (long long)b * N : nullptr;
    const double denom = (a.odom_dt ? a.odom_dt[b] : 0.0) + 1e-3;       // next_odom - odom + reg (:200-202)
    const int w0 = a.wp_offsets ? a.wp_offsets[b] : 0, w1 = a.wp_offsets ? a.wp_offsets[b + 1] : 0;

    for (int s = wave; s < S; s += kWaves) {
        const int p0 = sstart[s], p1 = seg_end(s), n = p1 - p0;
        // pass 1: mean
        double ax = 0.0, ay = 0.0;
        for (int i = p0 + lane; i < p1; i += POF_WAVE) {
            ax += sx[i];
            ay += sy[i];
        }
        const double sumx = wave_sum_f64(ax), sumy = wave_sum_f64(ay);
        const double mx = sumx / (double)n, my = sumy / (double)n;
        // pass 2: centred moments, boundary, curvature
        double suu = 0, svv = 0, suv = 0, suz = 0, svz = 0, sz = 0, blen = 0, curv = 0, ang = 0;
        for (int i = p0 + lane; i < p1; i += POF_WAVE) {
            const double u = sx[i] - mx, v = sy[i] - my, z = u * u + v * v;
            suu += u * u;
            svv += v * v;
            suv += u * v;
            suz += u * z;
            svz += v * z;
            sz += z;
            if (i + 1 < p1) blen += norm2(sx[i + 1] - sx[i], sy[i + 1] - sy[i]);
            if (i + 2 < p1) {
                const double ax_ = sx[i], ay_ = sy[i], bx = sx[i + 1], by = sy[i + 1], cx = sx[i + 2], cy = sy[i + 2];
                const double dA = norm2(bx - ax_, by - ay_), dB = norm2(cx - bx, cy - by), dC = norm2(ax_ - cx, ay_ - cy);
                const double area = fabs(0.5 * (ax_ * (by - cy) + bx * (cy - ay_) + cx * (ay_ - by)));
                curv += 4.0 * area / (dA * dB * dC);
                const double bax = ax_ - bx, bay = ay_ - by, bcx = cx - bx, bcy = cy - by;
                const double cosv = (bax * bcx + bay * bcy) / (norm2(bax, bay) * norm2(bcx, bcy));
                ang += acos(cosv);
            }
        }
        suu = wave_sum_f64(suu); svv = wave_sum_f64(svv); suv = wave_sum_f64(suv);
        suz = wave_sum_f64(suz); svz = wave_sum_f64(svz); sz = wave_sum_f64(sz);
        blen = wave_sum_f64(blen);
        if (n >= 3) {                      // wave-uniform
            curv = wave_sum_f64(curv);
            ang = wave_sum_f64(ang);
        }
        // the two solves (every lane holds the reduced moments)
        double k = nan, bb = nan, resid = nan, xc = nan, yc = nan, rc = nan;
        if (n >= 3) {
            // line y = k x + b: 2x2 normal equations in centred coordinates
            k = suv / suu;
            bb = my - k * mx;
            const double nrm = sqrt(k * k + 1.0);
            resid = (k / nrm) * sumx + (-1.0 / nrm) * sumy - (double)n * fabs(bb / nrm);
            // circle: [[Suu,Suv],[Suv,Svv]] (uc,vc) = 0.5 (Suz,Svz); c' = -Sz/n
            const double det = suu * svv - suv * suv;
            const double uc = 0.5 * (suz * svv - svz * suv) / det;
            const double vc = 0.5 * (svz * suu - suz * suv) / det;
            rc = sqrt(uc * uc + vc * vc + sz / (double)n);
            xc = uc + mx;
            yc = vc + my;
        }
        // pass 3: quantities that need the fit / the mean edge length
        double bstd = nan, scc = nan;
        if (n > 1) {
            const double me = blen / (double)(n - 1);
            double var = 0, sc = 0;
            for (int i = p0 + lane; i < p1; i += POF_WAVE) {
                if (i + 1 < p1) {
                    const double e = norm2(sx[i + 1] - sx[i], sy[i + 1] - sy[i]) - me;
                    var += e * e;
                }
                if (n >= 3) {
                    const double t = rc - sqrt(norm2(xc - sx[i], yc - sy[i]));
                    sc += t * t;
                }
            }
            bstd = sqrt(wave_sum_f64(var) / (double)(n - 1));
            if (n >= 3) scc = wave_sum_f64(sc);
        }
        const double sigma = (n > 1) ? sqrt(sz) / (double)(n - 1) : nan;
        const double width = norm2(sx[p1 - 1] - sx[p0], sy[p1 - 1] - sy[p0]);
        const double mang = (n >= 3) ? ang / (double)(n - 2) : nan;
        if (a.feat && lane == 0) {
            double *f = a.feat + ((long long)b * a.max_seg + s) * kFeat;
            const int sp = max(s - 1, 0), sn = min(s + 1, nseg - 1);
            const int prev_last = (sp == s) ? p1 - 1 : p0 - 1;
            const int next_first = (sn == s) ? p0 : p1;
            f[0] = (double)n;
            f[1] = sigma;
            f[2] = norm2(sx[prev_last] - sx[p0], sy[prev_last] - sy[p0]);
            f[3] = norm2(sx[p1 - 1] - sx[next_first], sy[p1 - 1] - sy[next_first]);
            f[4] = width;
            f[5] = resid;
            f[6] = scc;
            f[7] = rc;
            f[8] = blen;
            f[9] = bstd;
            f[10] = (n >= 3) ? curv : nan;
            f[11] = mang;
            f[12] = k;
            f[13] = bb;
            f[14] = xc;
            f[15] = yc;
        }
        const int kq = skidx[s];
        if (a.ref_feat && kq >= 0) {       // wave-uniform
            // 2: Frobenius norm of (segment - per-axis median) / n
            const double medx = wave_median(sx, smed, p0, n, lane), medy = wave_median(sy, smed, p0, n, lane);
            double fro = 0.0;
            for (int i = p0 + lane; i < p1; i += POF_WAVE) {
                const double dx = sx[i] - medx, dy = sy[i] - medy;
                fro += dx * dx + dy * dy;
            }
            fro = sqrt(wave_sum_f64(fro)) / (double)n;
            // 13: mean speed over piece kq of the unfiltered split
            double speed = nan;
            if (rn) {
                const int q0 = sstart[kq], q1 = seg_end(kq);
                double acc = 0.0;
                for (int i = q0 + lane; i < q1; i += POF_WAVE) acc += ((double)rn[i] - (double)r[i]) / denom;
                speed = wave_sum_f64(acc) / (double)(q1 - q0);
            }
            // 14: label: segment centre within radius_wp of an annotation
            int hit = 0;
            for (int w = w0 + lane; w < w1; w += POF_WAVE)
                hit |= norm2(mx - a.wp_xy[2 * w], my - a.wp_xy[2 * w + 1]) <= a.radius_wp;
            const bool pos = __any(hit);
            if (lane == 0) {
                double *f = a.ref_feat + ((long long)b * a.max_seg + kq) * kRef;
                const int pk = skept[max(kq - 1, 0)];
                const int pl = seg_end(pk) - 1;
                const int nq = min(kq + 1, 3);
                f[0] = (double)n;
                f[1] = sigma;
                f[2] = fro;
                f[3] = norm2(sx[pl] - sx[p0], sy[pl] - sy[p0]);
                if (nq < K) {
                    const int nf = sstart[skept[nq]];
                    f[4] = norm2(sx[p1 - 1] - sx[nf], sy[p1 - 1] - sy[nf]);
                } else {
                    f[4] = nan;
                }
                f[5] = width;
                f[6] = resid;
                f[7] = scc;
                f[8] = rc;
                f[9] = blen;
                f[10] = bstd;
                f[11] = curv;
                f[12] = mang;
                f[13] = speed;
                f[14] = pos ? 1.0 : -1.0;
            }
        }
    }
}

}  // namespace

extern "C" int pof_segment_features_ex(const float *ranges, const float *next_ranges, const double *tab, int B, int N,
                                       double jump_dist, const double *odom_dt, const int32_t *wp_offsets,
                                       const double *wp_xy, double radius_wp, int max_seg, int32_t *seg_id,
                                       int32_t *num_seg, int32_t *num_kept, double *feat, double *ref_feat,
                                       pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!ranges || !tab || !seg_id || !num_seg || (!feat && !ref_feat) || B < 0 || N < 1 || max_seg < 1)
        return POF_E_BADARG;
    if (wp_offsets && !wp_xy) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    SegArgs a;
    a.ranges = ranges; a.next_ranges = next_ranges; a.tab = tab; a.odom_dt = odom_dt;
    a.wp_offsets = wp_offsets; a.wp_xy = wp_xy; a.radius_wp = radius_wp;
    a.N = N; a.max_seg = max_seg; a.jump = (float)jump_dist;
    a.seg_id = seg_id; a.num_seg = num_seg; a.num_kept = num_kept; a.feat = feat; a.ref_feat = ref_feat;
    size_t lds = (size_t)N * (2 * sizeof(double) + sizeof(int)) + (size_t)(3 * max_seg + 1) * sizeof(int);
    lds = (lds + 7) & ~(size_t)7;
    a.med_off = (int)lds;
    if (a.ref_feat) lds += (size_t)N * sizeof(double);      // sort scratch of the per-axis medians
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

extern "C" int pof_segment_features(const float *ranges, const double *tab, int B, int N,
                                    double jump_dist, int max_seg, int32_t *seg_id, int32_t *num_seg,
                                    double *feat, pof_stream_t stream)
{
    if (!feat) return POF_E_BADARG;
    return pof_segment_features_ex(ranges, nullptr, tab, B, N, jump_dist, nullptr, nullptr, nullptr, 0.5, max_seg,
                                   seg_id, num_seg, nullptr, feat, nullptr, stream);
}
