// N1: device-resident scan store -> batch windows.
//
// The reference keeps every sequence in host memory and builds each sample's
// window in Python (src/utils/dataset_dr_spaam.py:357-378):
//   inds_tmp  = (arange(num_scans + distance) * scan_stride)[::-1]
//   scan_inds = [max(0, scan_idx - i) for i in inds_tmp[:num_scans]]
//   scans     = vstack(scans[scan_inds], cur_scan)            -> (num_scans+1, N)
//   odom1_idx = argmin|odoms_t - scans_t[scan_idx]|,  odom0_idx for scans_t[scan_inds[-1]]
// Here all sequences live concatenated in HBM ([S_total][N] ranges, timestamps,
// odometry) and a batch of windows is gathered by two launches:
//   gather_windows_kernel      row copies, 16 B per lane, HBM-bound:
//                              (T+1)*N*4 bytes read + written per sample
//   associate_odometry_kernel  one wave per sample: float32 |dt| argmin with
//                              first-minimum ties over the sequence's odometry
#include "pof_common.h"

namespace {

__global__ __launch_bounds__(256) void gather_windows_kernel(const float *scans_all, const int32_t *seq_first,
                                                             const int32_t *scan_idx, int num_scans,
                                                             int distance, int stride, int N, float *out,
                                                             int32_t *row_cur, int32_t *row_prev)
{
    const int b = blockIdx.y, j = blockIdx.x;  // j in [0, num_scans]: template rows, then the current scan
    const int T1 = num_scans + 1;
    const int si = scan_idx[b];
    // template row j looks back (num_scans + distance - 1 - j) * stride scans, clamped at the sequence start
    const int back = (j < num_scans) ? (num_scans + distance - 1 - j) * stride : 0;
    const int local = max(0, si - back);
    const long long src_row = (long long)seq_first[b] + local;
    if (threadIdx.x == 0) {
        if (j == num_scans) row_cur[b] = (int32_t)src_row;
        if (j == num_scans - 1) row_prev[b] = (int32_t)src_row;  // scan_inds[-1]: the odom0 time stamp
    }
    const float *src = scans_all + src_row * N;
    float *dst = out + ((long long)b * T1 + j) * N;
    const bool vec = (N % 4 == 0) && (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 15) == 0);
    if (vec) {
        using F4V = float __attribute__((ext_vector_type(4)));
        for (int e = threadIdx.x; e < N / 4; e += blockDim.x)
            __builtin_nontemporal_store(reinterpret_cast<const F4V *>(src)[e], reinterpret_cast<F4V *>(dst) + e);
    } else if ((N % 2 == 0) && (((uintptr_t)src & 7) == 0) && (((uintptr_t)dst & 7) == 0)) {
        for (int e = threadIdx.x; e < N / 2; e += blockDim.x)
            reinterpret_cast<float2 *>(dst)[e] = reinterpret_cast<const float2 *>(src)[e];
    } else {
        for (int e = threadIdx.x; e < N; e += blockDim.x) dst[e] = src[e];
    }
}

// first-minimum argmin of |t[k] - t_ref| (float32) over k in [lo, hi)
__device__ __forceinline__ int wave_argmin_time(const float *t, int lo, int hi, float t_ref, int lane)
{
    float best = INFINITY;
    int bidx = 0x7fffffff;
    for (int k = lo + lane; k < hi; k += 64) {
        const float d = fabsf(t[k] - t_ref);
        if (d < best) {  // strictly smaller: keeps the earliest index this lane saw
            best = d;
            bidx = k;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    return bidx;
}

__global__ __launch_bounds__(256) void associate_odometry_kernel(const float *scans_t, const float *odoms_t,
                                                                 const float *odoms, const int32_t *odom_lo,
                                                                 const int32_t *odom_hi, const int32_t *row_cur,
                                                                 const int32_t *row_prev, int B, double *odom0,
                                                                 double *odom1, int32_t *idx0, int32_t *idx1)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const int lo = odom_lo[b], hi = odom_hi[b];
    const int i1 = wave_argmin_time(odoms_t, lo, hi, scans_t[row_cur[b]], lane);
    const int i0 = wave_argmin_time(odoms_t, lo, hi, scans_t[row_prev[b]], lane);
    if (lane < 3) {
        odom1[3 * b + lane] = (double)odoms[3 * (long long)i1 + lane];
        odom0[3 * b + lane] = (double)odoms[3 * (long long)i0 + lane];
    }
    if (lane == 0) {
        if (idx1) idx1[b] = i1 - lo;
        if (idx0) idx0[b] = i0 - lo;
    }
}

}  // namespace

extern "C" int pof_gather_windows(const float *scans_all, const int32_t *seq_first, const int32_t *scan_idx,
                                  int B, int num_scans, int distance, int stride, int N, float *out,
                                  int32_t *row_cur, int32_t *row_prev, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!scans_all || !seq_first || !scan_idx || !out || !row_cur || !row_prev) return POF_E_BADARG;
    if (B < 0 || num_scans < 1 || distance < 0 || stride < 1 || N < 1) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    gather_windows_kernel<<<dim3(num_scans + 1, B), 256, 0, pof_stream(stream)>>>(
        scans_all, seq_first, scan_idx, num_scans, distance, stride, N, out, row_cur, row_prev);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

extern "C" int pof_associate_odometry(const float *scans_t, const float *odoms_t, const float *odoms,
                                      const int32_t *odom_lo, const int32_t *odom_hi, const int32_t *row_cur,
                                      const int32_t *row_prev, int B, double *odom0, double *odom1,
                                      int32_t *idx0, int32_t *idx1, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!scans_t || !odoms_t || !odoms || !odom_lo || !odom_hi || !row_cur || !row_prev || !odom0 || !odom1)
        return POF_E_BADARG;
    if (B < 0) return POF_E_BADARG;
    if (B == 0) return POF_OK;
    associate_odometry_kernel<<<(B + 3) / 4, 256, 0, pof_stream(stream)>>>(scans_t, odoms_t, odoms, odom_lo, odom_hi,
                                                                          row_cur, row_prev, B, odom0, odom1, idx0,
                                                                          idx1);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
