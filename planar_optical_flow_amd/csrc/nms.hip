// A11: nms_predicted_center (src/utils/utils.py:535-571), one workgroup per scan.
//
//   1. regression offsets -> detection centres (canonical_to_global + rphi_to_xy)
//   2. sort by descending score (bitonic, in LDS).  The order is total: equal scores (a saturated
//      sigmoid gives many exact 1.0s in deployment) are visited by DESCENDING point index, i.e. like
//      np.argsort(kind="stable")[::-1]; the reference's argsort()[::-1] leaves ties to NumPy's
//      introsort, so only distinct scores are comparable with it bit for bit
//   3. greedy suppression in score order: a kept centre labels every centre
//      closer than min_dist with its instance id (later ids overwrite earlier
//      ones, as in the reference) and suppresses it
//   4. kept centres compacted in score order.
// Serial in the number of kept centres and latency bound: reported in
// microseconds, not GB/s.
//
// Round 3, N <= 512 (nms_small_kernel): the serial part no longer costs two workgroup barriers per POINT.
//   * order by counting: rank of a point = number of points that come before it in the total order (N
//     comparisons per point, all lanes busy, one barrier) instead of a 45-step bitonic network;
//   * the N x N suppression relation dist < min_dist is built ONCE, by all waves, as a bit matrix in LDS
//     (row i = ceil(N/32) words).  No square root: sqrt is monotone and correctly rounded, so
//     RN(sqrt(s)) < min_dist  <=>  s <= s*, the largest float64 with that property (host, nextafter);
//   * one wave then walks the kept centres only: the alive set lives in the lanes' registers (one word per
//     lane), the next kept index is a count-trailing-zeros, a kept centre clears its row from the set: a
//     handful of instructions and one LDS row read per KEPT centre, no barrier;
//   * instance ids afterwards, in parallel: the reference lets later kept centres overwrite earlier labels,
//     so a point's id is that of the LAST kept centre whose row holds it (row & kept, highest bit).
// 0.26 ms -> measured in profiles/r3_* for 1024 scans x 450 points.
#include <cmath>

#include "pof_common.h"

namespace {

constexpr int kThreads = 512;

struct NmsArgs {
    const float *ranges;
    const double *tab;
    const double *pred_cls;
    const double *pred_reg;
    double min_dist;
    int N, Npad;
    double *det_xy, *det_cls;
    int32_t *num_det, *instance_mask;
};

__global__ __launch_bounds__(kThreads) void nms_kernel(NmsArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int N = a.N, Np = a.Npad;
    double *s_key = reinterpret_cast<double *>(smem);  // score, sorted descending
    double *s_x = s_key + Np;
    double *s_y = s_x + Np;
    int *s_ord = reinterpret_cast<int *>(s_y + Np);    // original point index
    int *s_keep = s_ord + Np;
    int *s_scan = s_keep + Np;
    __shared__ int s_flag;

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float *r = a.ranges + (long long)b * N;
    const double *cls = a.pred_cls + (long long)b * N;
    const double *reg = a.pred_reg + (long long)b * N * 2;

    for (int i = tid; i < Np; i += kThreads) {
        if (i < N) {
            s_key[i] = cls[i];
            s_ord[i] = i;
        } else {
            s_key[i] = -INFINITY;  // padding sorts to the end
            s_ord[i] = -1;
        }
    }
    __syncthreads();
    // bitonic sort, descending by key
    for (int k = 2; k <= Np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Np; i += kThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const bool desc = (i & k) == 0;
                    const double ki = s_key[i], kl = s_key[l];
                    const int oi = s_ord[i], ol = s_ord[l];
                    // total order: (score, point index) descending; padding (-inf, ord -1) stays behind
                    // real entries
                    const bool l_first = (kl > ki) || (kl == ki && ol > oi);
                    const bool i_first = (ki > kl) || (ki == kl && oi > ol);
                    const bool swap = desc ? l_first : i_first;
                    if (swap) {
                        s_key[i] = kl;
                        s_key[l] = ki;
                        s_ord[i] = ol;
                        s_ord[l] = oi;
                    }
                }
            }
            __syncthreads();
        }
    }
    // centres in sorted order
    for (int i = tid; i < Np; i += kThreads) {
        if (i < N) {
            const int src = s_ord[i];
            const double ty = (double)r[src] + reg[2 * src + 1];
            const double tphi = atan2(reg[2 * src], ty);
            const double dphi = tphi + a.tab[src];
            const double dr = ty / cos(tphi);
            double s, c;
            sincos(dphi, &s, &c);
            s_x[i] = dr * c;
            s_y[i] = dr * s;
            s_keep[i] = 1;
        } else {
            s_keep[i] = 0;
        }
    }
    int32_t *inst = a.instance_mask + (long long)b * N;
    for (int i = tid; i < N; i += kThreads) inst[i] = 0;
    __syncthreads();

    int next_id = 1;
    for (int i = 0; i < N; ++i) {
        if (tid == 0) s_flag = s_keep[i];
        __syncthreads();
        const int live = s_flag;
        if (live) {
            const double xi = s_x[i], yi = s_y[i];
            for (int j = tid; j < N; j += kThreads) {
                const double dx = xi - s_x[j], dy = yi - s_y[j];
                const double dist = sqrt(dx * dx + dy * dy);
                if (dist < a.min_dist) {
                    s_keep[j] = (j == i);
                    inst[s_ord[j]] = next_id;
                }
            }
            ++next_id;
        }
        __syncthreads();
    }

    // compaction of kept centres (inclusive Hillis-Steele scan over Npad flags)
    for (int i = tid; i < Np; i += kThreads) s_scan[i] = s_keep[i];
    __syncthreads();
    for (int off = 1; off < Np; off <<= 1) {
        int v[8];
        int cnt = 0;
        for (int i = tid; i < Np; i += kThreads) v[cnt++] = (i >= off) ? s_scan[i - off] : 0;
        __syncthreads();
        cnt = 0;
        for (int i = tid; i < Np; i += kThreads) s_scan[i] += v[cnt++];
        __syncthreads();
    }
    double *oxy = a.det_xy + (long long)b * N * 2;
    double *ocl = a.det_cls + (long long)b * N;
    for (int i = tid; i < N; i += kThreads) {
        if (s_keep[i]) {
            const int pos = s_scan[i] - 1;
            oxy[2 * pos] = s_x[i];
            oxy[2 * pos + 1] = s_y[i];
            ocl[pos] = s_key[i];
        }
    }
    if (tid == 0) a.num_det[b] = s_scan[Np - 1];
}

constexpr int kSmallMaxN = 512;
constexpr int kSmallMaxW = kSmallMaxN / 32;

struct NmsSmallArgs {
    NmsArgs a;
    double s2_thr;      // dist < min_dist  <=>  dx*dx + dy*dy <= s2_thr
};

__global__ __launch_bounds__(kThreads) void nms_small_kernel(NmsSmallArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const NmsArgs &a = A.a;
    const int N = a.N, W = (N + 31) >> 5;
    double *s_key = reinterpret_cast<double *>(smem);          // [N] score in sorted order
    double *s_x = s_key + kSmallMaxN;                          // [N] centre, sorted order
    double *s_y = s_x + kSmallMaxN;
    double *s_raw = s_y + kSmallMaxN;                          // [N] score by point index
    int *s_ord = reinterpret_cast<int *>(s_raw + kSmallMaxN);  // [N] point index of sorted position
    unsigned *s_kept = reinterpret_cast<unsigned *>(s_ord + kSmallMaxN);   // [W] kept set
    int *s_base = reinterpret_cast<int *>(s_kept + kSmallMaxW);            // [W] kept centres before word w
    unsigned *s_mat = reinterpret_cast<unsigned *>(s_base + kSmallMaxW);   // [N][W] suppression relation

    const int b = blockIdx.x, tid = threadIdx.x;
    const float *r = a.ranges + (long long)b * N;
    const double *cls = a.pred_cls + (long long)b * N;
    const double *reg = a.pred_reg + (long long)b * N * 2;
    for (int i = tid; i < N; i += kThreads) s_raw[i] = cls[i];
    __syncthreads();
    // rank in the total order (score, point index) descending = position in the sorted sequence
    for (int i = tid; i < N; i += kThreads) {
        const double ki = s_raw[i];
        int rank = 0;
        for (int j = 0; j < N; ++j) {
            const double kj = s_raw[j];
            rank += ((kj > ki) || (kj == ki && j > i)) ? 1 : 0;
        }
        // a NaN score compares false both ways: it would collide with other ranks; give NaNs the tail in index order
        if (ki != ki) {
            rank = 0;
            for (int j = 0; j < N; ++j) rank += (s_raw[j] == s_raw[j] || j > i) ? 1 : 0;
        }
        s_ord[rank] = i;
        s_key[rank] = ki;
        const double ty = (double)r[i] + reg[2 * i + 1];
        const double tphi = atan2(reg[2 * i], ty);
        const double dphi = tphi + a.tab[i];
        const double dr = ty / cos(tphi);
        double sn, cs;
        sincos(dphi, &sn, &cs);
        s_x[rank] = dr * cs;
        s_y[rank] = dr * sn;
    }
    int32_t *inst = a.instance_mask + (long long)b * N;
    __syncthreads();
    // suppression relation: word (i, w) = bits j = 32 w .. of  (dx*dx + dy*dy <= s2_thr)
    for (int e = tid; e < N * W; e += kThreads) {
        const int i = e / W, w = e - i * W;
        const double xi = s_x[i], yi = s_y[i];
        unsigned bits = 0;
        const int j0 = 32 * w, jn = min(32, N - j0);
        for (int u = 0; u < jn; ++u) {
            const double dx = xi - s_x[j0 + u], dy = yi - s_y[j0 + u];
            bits |= (dx * dx + dy * dy <= A.s2_thr ? 1u : 0u) << u;
        }
        s_mat[e] = bits;
    }
    __syncthreads();
    if (tid < 64) {
        // the greedy walk, one wave, no barrier: lane w holds word w of the alive set
        const int lane = tid;
        unsigned alive = 0, kept = 0;
        if (lane < W) alive = (lane == W - 1 && (N & 31)) ? ((1u << (N & 31)) - 1u) : 0xffffffffu;
        int w = 0;
        unsigned lowmask = 0xffffffffu;         // bits of word w not yet passed
        while (w < W) {
            const unsigned m = (unsigned)__builtin_amdgcn_readlane((int)alive, w) & lowmask;
            if (m == 0) {
                ++w;
                lowmask = 0xffffffffu;
                continue;
            }
            const int bit = __builtin_ctz(m);
            const int idx = 32 * w + bit;
            const unsigned row = lane < W ? s_mat[idx * W + lane] : 0u;
            alive &= ~row;                       // everything within min_dist of the kept centre, itself included
            if (lane == w) kept |= 1u << bit;
            lowmask = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
        }
        if (lane < W) s_kept[lane] = kept;
        // kept centres before each word (exclusive prefix of the popcounts)
        int c = lane < W ? __builtin_popcount(kept) : 0, pre = c;
#pragma unroll
        for (int o = 1; o < kSmallMaxW; o <<= 1) {
            const int v = __shfl_up(pre, o, 64);
            if (lane >= o) pre += v;
        }
        if (lane < W) s_base[lane] = pre - c;
        if (lane == W - 1) a.num_det[b] = pre;
    }
    __syncthreads();
    double *oxy = a.det_xy + (long long)b * N * 2;
    double *ocl = a.det_cls + (long long)b * N;
    for (int j = tid; j < N; j += kThreads) {
        // id of a kept centre = its 1-based position among the kept ones; a point carries the id of the LAST kept
        // centre whose row holds it (later ids overwrite earlier ones in the reference); the relation is symmetric,
        // so that is the highest bit of row_j & kept
        int id = 0;
        for (int w = W - 1; w >= 0; --w) {
            const unsigned v = s_mat[j * W + w] & s_kept[w];
            if (v) {
                const int hb = 31 - __builtin_clz(v);
                id = s_base[w] + __builtin_popcount(s_kept[w] & ((hb == 31) ? 0xffffffffu : ((1u << (hb + 1)) - 1u)));
                break;
            }
        }
        inst[s_ord[j]] = id;
        const int wj = j >> 5, bj = j & 31;
        if ((s_kept[wj] >> bj) & 1u) {
            const int pos = s_base[wj] + __builtin_popcount(s_kept[wj] & ((1u << bj) - 1u));
            oxy[2 * pos] = s_x[j];
            oxy[2 * pos + 1] = s_y[j];
            ocl[pos] = s_key[j];
        }
    }
}

// largest float64 s with RN(sqrt(s)) < r (the reference compares the rounded distance with min_dist)
double nms_sq_threshold(double r)
{
    if (!(r > 0.0)) return -1.0;
    double c = r * r;
    for (int it = 0; it < 64 && !(std::sqrt(c) < r); ++it) c = std::nextafter(c, 0.0);
    for (int it = 0; it < 64; ++it) {
        const double up = std::nextafter(c, HUGE_VAL);
        if (!(std::sqrt(up) < r)) break;
        c = up;
    }
    return c;
}

int next_pow2(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace

extern "C" size_t pof_nms_workspace_bytes(int B, int N)
{
    (void)B;
    (void)N;
    return 0;  // everything lives in LDS; kept in the ABI for larger scans
}

extern "C" int pof_nms_predicted_center(const float *ranges, const double *tab, const double *pred_cls,
                                        const double *pred_reg, double min_dist, int B, int N,
                                        double *det_xy, double *det_cls, int32_t *num_det,
                                        int32_t *instance_mask, void *workspace, size_t workspace_bytes,
                                        pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    (void)workspace;
    (void)workspace_bytes;
    if (!ranges || !tab || !pred_cls || !pred_reg || !det_xy || !det_cls || !num_det || !instance_mask)
        return POF_E_BADARG;
    if (B < 0 || N < 1) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    NmsArgs a;
    a.ranges = ranges; a.tab = tab; a.pred_cls = pred_cls; a.pred_reg = pred_reg; a.min_dist = min_dist;
    a.N = N; a.Npad = next_pow2(N);
    a.det_xy = det_xy; a.det_cls = det_cls; a.num_det = num_det; a.instance_mask = instance_mask;
    if (N <= kSmallMaxN && std::isfinite(min_dist)) {
        NmsSmallArgs A;
        A.a = a;
        A.s2_thr = nms_sq_threshold(min_dist);
        const int W = (N + 31) / 32;
        const size_t lds_s = (size_t)kSmallMaxN * (4 * sizeof(double) + sizeof(int)) + 2 * kSmallMaxW * sizeof(int) +
                             (size_t)N * W * sizeof(unsigned);
        nms_small_kernel<<<B, kThreads, lds_s, pof_stream(stream)>>>(A);
        POF_CHECK_LAUNCH();
        return POF_OK;
    }
    if (a.Npad > 4096) return POF_E_SHAPE;  // 8 flags per thread in the scan
    a.det_xy = det_xy; a.det_cls = det_cls; a.num_det = num_det; a.instance_mask = instance_mask;
    const size_t lds = (size_t)a.Npad * (3 * sizeof(double) + 3 * sizeof(int));
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(nms_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return POF_E_LAUNCH;
    }
    nms_kernel<<<B, kThreads, lds, pof_stream(stream)>>>(a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
