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
