// N4: the boosted decision-stump detector of the legacy person-detection baseline
// (src/depracted/model/adaboost_person_det.py:212-378).
//
// pof_stump_search -- BoostedFeatureDetector.simple_classifier (:283-347), all feature dimensions in one
//   launch.  The reference sorts every feature column, takes as threshold candidates the midpoints
//   between sorted neighbours of opposite class, and for each candidate counts the misclassified samples
//   of "x > theta -> +1" with a Python loop over samples (D * T * n interpreted comparisons per call).
//   Here one workgroup owns one dimension: the (optionally index-gathered) column is sorted in LDS
//   (bitonic, key = (value, position), so equal values keep their sampled order), an inclusive scan
//   counts the +1 labels along the sorted order, and every candidate's error is
//       #(+1 with x <= theta) + #(-1 with x > theta)
//   read from the scan at the last position with value <= theta (found by bisection, which also covers
//   midpoints that round onto a neighbour and runs of equal values).  Per dimension the kernel reports the
//   smallest and the largest error count with the FIRST candidate reaching each, which is exactly what
//   the reference's min / argmin over `error / N` and over `1 - error / N` pick; the short cross-dimension
//   selection with its exact float comparisons stays on the host.
//
// pof_stump_vote -- BoostedFeatureDetector.eval (:349-378): result = sum_k alpha_k * (x[j_k] > theta_k ?
//   +1 : -1), accumulated in round order in float64 like the reference, label = sign(result).
#include "pof_common.h"

namespace {

constexpr int kStumpThreads = 256;
constexpr int kStumpMaxN = 2048;

struct StumpArgs {
    const double *X;        // [rows][D]
    const double *Y;        // [rows], +1 / -1
    const int *index;       // [n] row of each sample, or nullptr for 0..n-1
    long long rows;
    int n, D, P;            // P = power of two >= n
    int *min_err, *max_err, *n_thresh;
    double *theta_min, *theta_max;
};

__device__ __forceinline__ bool key_less(double a, int ia, double b, int ib)
{
    return a < b || (a == b && ia < ib);
}

// row of sample i; an index outside [0, rows) is clamped so that a bad index list cannot fault the device
// (the host wrapper rejects it beforehand)
__device__ __forceinline__ long long sample_row(const StumpArgs &a, int i)
{
    long long row = a.index ? a.index[i] : i;
    row = row < 0 ? 0 : row;
    return row >= a.rows ? a.rows - 1 : row;
}

__global__ __launch_bounds__(kStumpThreads) void stump_search_kernel(StumpArgs a)
{
    extern __shared__ unsigned char lds_raw[];
    double *s_val = reinterpret_cast<double *>(lds_raw);                 // [P]
    int *s_pos = reinterpret_cast<int *>(s_val + a.P);                   // [P]
    int *s_cum = s_pos + a.P;                                            // [P] inclusive count of +1 labels
    signed char *s_lab = reinterpret_cast<signed char *>(s_cum + a.P);   // [P] label along the sorted order
    __shared__ unsigned long long s_best_min, s_best_max;
    __shared__ int s_cnt;

    const int d = blockIdx.x, tid = threadIdx.x, n = a.n, P = a.P;
    for (int i = tid; i < P; i += kStumpThreads) {
        if (i < n) {
            s_val[i] = a.X[sample_row(a, i) * a.D + d];
        } else {
            s_val[i] = __builtin_huge_val();      // padding sorts behind every sample (position breaks ties)
        }
        s_pos[i] = i;
    }
    if (tid == 0) {
        s_best_min = ~0ull;
        s_best_max = 0ull;
        s_cnt = 0;
    }
    __syncthreads();
    // bitonic sort, ascending by (value, position)
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += kStumpThreads) {
                const int p = i ^ j;
                if (p > i) {
                    const double vi = s_val[i], vp = s_val[p];
                    const int qi = s_pos[i], qp = s_pos[p];
                    const bool up = (i & k) == 0;
                    const bool swap = up ? key_less(vp, qp, vi, qi) : key_less(vi, qi, vp, qp);
                    if (swap) {
                        s_val[i] = vp; s_val[p] = vi;
                        s_pos[i] = qp; s_pos[p] = qi;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < P; i += kStumpThreads) {
        int lab = 0;
        if (i < n) {
            const int q = s_pos[i];
            lab = a.Y[sample_row(a, q)] > 0.0 ? 1 : -1;
        }
        s_lab[i] = (signed char)lab;
        s_cum[i] = lab > 0 ? 1 : 0;
    }
    __syncthreads();
    // inclusive scan (Hillis-Steele over LDS; P <= 2048)
    for (int off = 1; off < P; off <<= 1) {
        int add[kStumpMaxN / kStumpThreads];
        int c = 0;
        for (int i = tid; i < P; i += kStumpThreads, ++c) add[c] = i >= off ? s_cum[i - off] : 0;
        __syncthreads();
        c = 0;
        for (int i = tid; i < P; i += kStumpThreads, ++c) s_cum[i] += add[c];
        __syncthreads();
    }
    const int total_pos = s_cum[n - 1];
    // candidates: sorted neighbours (c, c+1) of opposite class, c <= n-2
    unsigned long long best_min = ~0ull, best_max = 0ull;
    int cnt = 0;
    for (int c = tid; c + 1 < n; c += kStumpThreads) {
        if (s_lab[c] + s_lab[c + 1] != 0) continue;
        const double th = (s_val[c] + s_val[c + 1]) / 2;
        int lo = c, hi = n - 1;                   // last position with value <= th (>= c: s_val[c] <= th)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_val[mid] <= th) lo = mid; else hi = mid - 1;
        }
        const int pos_le = s_cum[lo];
        const int err = pos_le + ((n - 1 - lo) - (total_pos - pos_le));
        const unsigned long long kmin = ((unsigned long long)(unsigned)err << 32) | (unsigned)c;
        const unsigned long long kmax = ((unsigned long long)(unsigned)err << 32) | (unsigned)(0x7fffffff - c);
        best_min = kmin < best_min ? kmin : best_min;
        best_max = kmax > best_max ? kmax : best_max;
        ++cnt;
    }
    if (cnt) {
        atomicMin(&s_best_min, best_min);
        atomicMax(&s_best_max, best_max);
        atomicAdd(&s_cnt, cnt);
    }
    __syncthreads();
    if (tid == 0) {
        a.n_thresh[d] = s_cnt;
        if (s_cnt) {
            const int cmin = (int)(s_best_min & 0xffffffffu);
            const int cmax = 0x7fffffff - (int)(s_best_max & 0xffffffffu);
            a.min_err[d] = (int)(s_best_min >> 32);
            a.max_err[d] = (int)(s_best_max >> 32);
            a.theta_min[d] = (s_val[cmin] + s_val[cmin + 1]) / 2;
            a.theta_max[d] = (s_val[cmax] + s_val[cmax + 1]) / 2;
        } else {
            a.min_err[d] = a.max_err[d] = -1;
            a.theta_min[d] = a.theta_max[d] = 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void stump_vote_kernel(const double *__restrict__ X, long long N, int D,
                                                         const int *__restrict__ dim, const double *__restrict__ theta,
                                                         const double *__restrict__ alpha, int K,
                                                         double *__restrict__ result, double *__restrict__ label)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double *x = X + i * D;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
        int j = dim[k] - 1;                        // 1-based like `para[:, 0]`; 0 wraps to the last column
        if (j < 0) j += D;
        if (j < 0 || j >= D) continue;             // rejected by the host wrapper; never read out of the row
        acc += alpha[k] * (x[j] > theta[k] ? 1.0 : -1.0);
    }
    result[i] = acc;
    if (label) label[i] = acc > 0.0 ? 1.0 : (acc < 0.0 ? -1.0 : acc);     // np.sign (keeps NaN)
}

} // namespace

extern "C" int pof_stump_search(const double *X, const double *Y, long long rows, const int *index, int n, int D,
                                int *min_err, double *theta_min, int *max_err, double *theta_max,
                                int *n_thresh, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!X || !Y || !min_err || !theta_min || !max_err || !theta_max || !n_thresh) return POF_E_BADARG;
    if (n < 2 || D < 1 || rows < 1 || (!index && rows < n)) return POF_E_BADARG;
    if (n > kStumpMaxN || D > 65535) return POF_E_SHAPE;
    StumpArgs a;
    a.X = X; a.Y = Y; a.index = index; a.rows = rows; a.n = n; a.D = D;
    a.P = 2;
    while (a.P < n) a.P <<= 1;
    a.min_err = min_err; a.max_err = max_err; a.n_thresh = n_thresh;
    a.theta_min = theta_min; a.theta_max = theta_max;
    const size_t lds = (size_t)a.P * (sizeof(double) + 2 * sizeof(int) + 1);
    hipLaunchKernelGGL(stump_search_kernel, dim3(D), dim3(kStumpThreads), lds, pof_stream(stream), a);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_stump_vote(const double *X, long long N, int D, const int *dim, const double *theta,
                              const double *alpha, int K, double *result, double *label, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!X || !dim || !theta || !alpha || !result || N < 0 || D < 1 || K < 0) return POF_E_BADARG;
    if (N == 0) return POF_OK;
    if ((N + 255) / 256 > 2147483647LL) return POF_E_SHAPE;
    hipLaunchKernelGGL(stump_vote_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, pof_stream(stream),
                       X, N, D, dim, theta, alpha, K, result, label);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
