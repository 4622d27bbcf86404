// A16: rotated-box IoU (2-D and 3-D), src/utils/rotate_iou.py:20-404.
//
// The reference is a numba.cuda kernel launched once per sample with N = 1
// (src/model/box_regression_fn.py:76-82) -- launch bound.  Here the whole
// evaluation set goes in one launch: G groups of (N boxes x K query boxes),
// one lane per (box, query) pair, with optional per-group valid counts so that
// ragged neighbour lists can be padded.
//
// Arithmetic is float32 in the reference's operation order (contraction off):
// corners -> 4+4 containment tests + 16 edge intersections -> angular
// insertion sort of <= 24 candidate points -> fan triangulation.
#include "pof_common.h"

namespace {

struct Pt {
    float x, y;
};

__device__ __forceinline__ void box_corners(const float *b, float *c)
{
    const float ang = b[4];
    // correctly rounded float32 cos / sin (float64 evaluation rounded once): what the reference's device
    // functions give when they run as plain Python (tools/gen_golden.py), and what the oracle computes.
    // Degenerate pairs (identical rotated boxes) flip between 0 and 1 on the last bit of these two values.
    const float ac = (float)cos((double)ang), as = (float)sin((double)ang);
    const float hx = b[2] / 2.0f, hy = b[3] / 2.0f;
    const float xs[4] = {-hx, -hx, hx, hx};
    const float ys[4] = {-hy, hy, hy, -hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = ac * xs[i] + as * ys[i] + b[0];
        c[2 * i + 1] = -as * xs[i] + ac * ys[i] + b[1];
    }
}

__device__ __forceinline__ bool in_quad(float px, float py, const float *q)
{
    const float ab0 = q[2] - q[0], ab1 = q[3] - q[1];
    const float ad0 = q[6] - q[0], ad1 = q[7] - q[1];
    const float ap0 = px - q[0], ap1 = py - q[1];
    const float abab = ab0 * ab0 + ab1 * ab1;
    const float abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1;
    const float adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0.0f && adad >= adap && adap >= 0.0f;
}

__device__ __forceinline__ bool edge_hit(const float *p1, const float *p2, int i, int j, float *ox, float *oy)
{
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    const float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    const float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1;
    const float DA0 = D0 - A0, CA0 = C0 - A0;
    const float DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const bool abc = CA1 * BA0 > BA1 * CA0;
        const bool abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1;
            const float CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1;
            const float Dx = ABBA * DC0 - BA0 * CDDC;
            const float Dy = ABBA * DC1 - BA1 * CDDC;
            *ox = Dx / DH;
            *oy = Dy / DH;
            return true;
        }
    }
    return false;
}

__device__ float inter_area(const float *b1, const float *b2)
{
    float c1[8], c2[8];
    box_corners(b1, c1);
    box_corners(b2, c2);
    // up to 8 containment points + 16 edge crossings
    float px[24], py[24], vs[24];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (in_quad(c1[2 * i], c1[2 * i + 1], c2)) {
            px[n] = c1[2 * i];
            py[n] = c1[2 * i + 1];
            ++n;
        }
        if (in_quad(c2[2 * i], c2[2 * i + 1], c1)) {
            px[n] = c2[2 * i];
            py[n] = c2[2 * i + 1];
            ++n;
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float hx, hy;
            if (edge_hit(c1, c2, i, j, &hx, &hy)) {
                px[n] = hx;
                py[n] = hy;
                ++n;
            }
        }
    if (n > 0) {
        float cx = 0.0f, cy = 0.0f;
        for (int i = 0; i < n; ++i) {
            cx += px[i];
            cy += py[i];
        }
        cx /= (float)n;
        cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float vx = px[i] - cx, vy = py[i] - cy;
            const float d = sqrtf(vx * vx + vy * vy);
            vx = vx / d;
            vy = vy / d;
            if (vy < 0.0f) vx = -2.0f - vx;
            vs[i] = vx;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float t = vs[i], tx = px[i], ty = py[i];
                int j = i;
                while (j > 0 && vs[j - 1] > t) {
                    vs[j] = vs[j - 1];
                    px[j] = px[j - 1];
                    py[j] = py[j - 1];
                    --j;
                }
                vs[j] = t;
                px[j] = tx;
                py[j] = ty;
            }
        }
    }
    float area = 0.0f;
    for (int i = 0; i < n - 2; ++i) {
        const float tri = ((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) -
                           (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) / 2.0f;
        area += fabsf(tri);
    }
    return area;
}

template <bool IS3D>
__device__ float iou_pair(const float *q, const float *b, int criterion)
{
    const float a1 = q[2] * q[3], a2 = b[2] * b[3];
    const float ai = inter_area(q, b);
    if (!IS3D) {
        if (criterion == -1) return ai / (a1 + a2 - ai);
        if (criterion == 0) return ai / a1;
        if (criterion == 1) return ai / a2;
        return ai;
    }
    const float v1 = a1 * q[6], v2 = a2 * b[6];
    float h;
    if (fabsf(q[5] - b[5]) >= 0.5f * (q[6] + b[6])) {
        h = 0.0f;
    } else {
        h = fminf(q[5] + 0.5f * q[6], b[5] + 0.5f * b[6]) - fmaxf(q[5] - 0.5f * q[6], b[5] - 0.5f * b[6]);
    }
    const float vi = ai * h;
    if (criterion == -1) return vi / (v1 + v2 - vi);
    if (criterion == 0) return vi / v1;
    if (criterion == 1) return vi / v2;
    return vi;
}

template <bool IS3D>
__global__ __launch_bounds__(64) void rotate_iou_kernel(const float *boxes, const float *query,
                                                        float *iou, int N, int K, const int32_t *n_valid,
                                                        const int32_t *k_valid, int criterion)
{
    constexpr int S = IS3D ? 7 : 5;
    const int g = blockIdx.z;
    const long long pair = (long long)blockIdx.x * 64 + threadIdx.x;
    if (pair >= (long long)N * K) return;
    const int n = (int)(pair / K), k = (int)(pair - (long long)n * K);
    float *o = iou + ((long long)g * N + n) * K + k;
    const int nv = n_valid ? n_valid[g] : N, kv = k_valid ? k_valid[g] : K;
    if (n >= nv || k >= kv) {
        *o = 0.0f;
        return;
    }
    float qb[S], bb[S];
    const float *bp = boxes + ((long long)g * N + n) * S;
    const float *qp = query + ((long long)g * K + k) * S;
#pragma unroll
    for (int c = 0; c < S; ++c) {
        bb[c] = bp[c];
        qb[c] = qp[c];
    }
    *o = iou_pair<IS3D>(qb, bb, criterion);
}

}  // namespace

extern "C" int pof_rotate_iou(const float *boxes, const float *query, float *iou, int G, int N, int K,
                              const int32_t *n_valid, const int32_t *k_valid, int criterion, int is_3d,
                              pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!boxes || !query || !iou || G < 0 || N < 0 || K < 0) return POF_E_BADARG;
    if (criterion < -1) return POF_E_BADARG;
    if (G == 0 || N == 0 || K == 0) return POF_OK;
    if (G > 65535) return POF_E_SHAPE;
    const long long pairs = (long long)N * K;
    dim3 grid((unsigned)((pairs + 63) / 64), 1, G);
    if (is_3d)
        rotate_iou_kernel<true><<<grid, 64, 0, pof_stream(stream)>>>(boxes, query, iou, N, K, n_valid, k_valid, criterion);
    else
        rotate_iou_kernel<false><<<grid, 64, 0, pof_stream(stream)>>>(boxes, query, iou, N, K, n_valid, k_valid, criterion);
    POF_CHECK_LAUNCH();
    return POF_OK;
}
