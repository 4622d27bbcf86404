// N2, training side: weight gradient of the trunk's Conv1d(kernel 3, padding 1)
// (src/depracted/model/dr_spaam.py:8-19), the third convolution pass of a training step next to the forward
// and data-gradient passes that run on conv3_kernel (conv_trunk.hip).
//
//   dw[co][ci][t] = sum over sequences s and positions l of  dy[s][co][l] * x[s][ci][l + t - 1]
//
// As a GEMM: M = output channels, N = (tap, input channel), K = (sequence, position) -- 2e5..1e6 long, so K is
// split over workgroups (no atomics: partial tiles, then one reduction pass; the result is deterministic).
// The library's kernel for this transposes both operands to channels-last first; here both are read as they lie
// ([S][C][L]): the rows of a channel tile of one sequence are ONE contiguous run in memory, staged into LDS with
// coalesced 16-byte loads, and the MFMA operands (v_mfma_f32_32x32x2_f32: A[m][k] = dy[co][l], B[k][n] = x[ci][l+t-1])
// are read from LDS with an odd row stride (conflict-free across the 32 channels of an operand).  The x row
// carries one zero in front and zeros behind, so the three taps are three reads of the same row at offsets
// +0 / +1 / +2 and the first tap of the next k-step is the third tap of this one (2 new B reads per step).
//
// Workgroup tile: 128 (or 64) output channels x 64 input channels x 3 taps; a wave owns 32 output channels and
// 6 (or 3) 32x32 accumulators.  One LDS stage = G whole sequences (G = 1 at L = 56 .. 5 at L = 7).  The next
// stage's global loads are issued into registers before the MFMAs of the stage in flight and written to LDS
// after them (one LDS buffer, two barriers per stage); two to three workgroups per CU cover the rest.
// Bound: float32 MFMA (256 FLOP/cycle/CU); flops = 2 * S * L * 3 * Ci * Co, the forward's.
#include "pof_common.h"

namespace {

constexpr int kWgThreads = 256;
constexpr int kWgCiT = 64;          // input channels per workgroup tile
constexpr int kWgLdsBudget = 48 * 1024;
// floats a thread holds per stage for the 128-row dy tile (64-row tiles: half), by load width: the narrow loads
// need more address / predicate state per float, so they get shorter stages
__host__ __device__ constexpr int stage_floats(int vec) { return vec == 4 ? 32 : vec == 2 ? 24 : 16; }

typedef float float16v __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float *x, *dy;
    float *partial;
    int S, Ci, Co, L;
    int Lh;             // k-steps per sequence = ceil(L / 2)
    int sd, sx;         // LDS row strides of the dy / x tiles (odd)
    int G;              // sequences per LDS stage
    int spw;            // sequences per workgroup (multiple of G)
    int nsplit;
    int ci_tiles, tiles;
    int vec, tpr_log2;  // global load width (floats) and threads per row (power of two >= L / vec)
};

template <int V> struct VecOf;
template <> struct VecOf<4> { typedef float4 type; };
template <> struct VecOf<2> { typedef float2 type; };
template <> struct VecOf<1> { typedef float type; };

__device__ __forceinline__ void lds_put(float *q, const float4 &v) { q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
__device__ __forceinline__ void lds_put(float *q, const float2 &v) { q[0] = v.x; q[1] = v.y; }
__device__ __forceinline__ void lds_put(float *q, const float &v) { q[0] = v; }

// One operand tile of a stage, global -> registers -> LDS.  Rows [row0, row0 + 2^ROWS_LOG2) of sequences
// [s0, s0 + G) go to LDS rows (g << ROWS_LOG2) + r, data from column `col0`.  A thread serves the same segment
// (V floats) of every (256 / threads-per-row)-th row: NP vectors at most, held in registers between `issue` (the
// global loads, started before the MFMAs of the stage in flight) and `commit` (the LDS writes, after them).
// Rows of absent sequences / channels are written as zeros.
template <int V, int NP, int ROWS_LOG2>
struct StageTile {
    typename VecOf<V>::type r[NP];

    __device__ __forceinline__ void issue(const float *__restrict__ src, int C, int L, int row0, int s0, int s_end,
                                          int G, int tpr_log2)
    {
        const int seg = threadIdx.x & ((1 << tpr_log2) - 1);
        const int rpp = kWgThreads >> tpr_log2, total = G << ROWS_LOG2;
        const bool seg_ok = seg * V < L;
        // uniform base (scalar registers) + 32-bit lane offset: the NP loads in flight cost one address register each
        const float *base = src + ((long long)s0 * C + row0) * L;
        const int g_ok = s_end - s0, c_ok = C - row0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rr = (threadIdx.x >> tpr_log2) + p * rpp;
            const int g = rr >> ROWS_LOG2, cr = rr & ((1 << ROWS_LOG2) - 1);
            typename VecOf<V>::type v = {};
            if (seg_ok && rr < total && g < g_ok && cr < c_ok)
                v = *reinterpret_cast<const typename VecOf<V>::type *>(base + ((g * C + cr) * L + seg * V));
            r[p] = v;
        }
    }

    __device__ __forceinline__ void commit(float *lds, int stride, int col0, int L, int G, int tpr_log2) const
    {
        const int seg = threadIdx.x & ((1 << tpr_log2) - 1);
        const int rpp = kWgThreads >> tpr_log2, total = G << ROWS_LOG2;
        if (seg * V >= L) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int rr = (threadIdx.x >> tpr_log2) + p * rpp;
            if (rr < total) lds_put(lds + rr * stride + col0 + seg * V, r[p]);
        }
    }
};

// WM = waves along the output channels: 4 -> tile 128 co, a wave owns both 32-channel halves of the ci tile;
// 2 -> tile 64 co, a wave owns one half.  V = floats per global load (4 / 2 / 1 by the row length's alignment).
// TAPS = 3: the kernel-3 convolution above.  TAPS = 1: the point-wise convolution (dw[co][ci] = sum dy[co][l] x[ci][l]):
// the x row carries no leading zero and a k-step is one MFMA per accumulator.
template <int WM, int V, int TAPS>
__global__ __launch_bounds__(kWgThreads, 2) void conv_wgrad_kernel(WgradArgs a)
{
    constexpr int MT_LOG2 = WM == 4 ? 7 : 6, MT = 1 << MT_LOG2;
    constexpr int NSUB = WM == 4 ? 2 : 1;
    constexpr int NPD = stage_floats(V) * MT / 128 / V, NPX = stage_floats(V) / 2 / V;
    extern __shared__ float lds[];
    float *lds_d = lds;
    float *lds_x = lds + a.G * MT * a.sd;
    const int lds_floats = a.G * (MT * a.sd + kWgCiT * a.sx);

    // 1-D grid, workgroup id -> (split, tile) so that the tiles of one split -- which read the same sequences --
    // are dispatched together on ONE XCD (ids go round-robin over the 8 XCDs) and share its L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = j % a.tiles, split = xcd + 8 * (j / a.tiles);
    if (split >= a.nsplit) return;
    const int co0 = (tile / a.ci_tiles) * MT, ci0 = (tile % a.ci_tiles) * kWgCiT;
    const int s_begin = split * a.spw;
    const int s_end = s_begin + a.spw < a.S ? s_begin + a.spw : a.S;

    StageTile<V, NPD, MT_LOG2> td;
    StageTile<V, NPX, 6> tx;
    td.issue(a.dy, a.Co, a.L, co0, s_begin, s_end, a.G, a.tpr_log2);
    tx.issue(a.x, a.Ci, a.L, ci0, s_begin, s_end, a.G, a.tpr_log2);

    for (int i = threadIdx.x; i < lds_floats; i += kWgThreads) lds[i] = 0.0f;   // the halo columns stay zero

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int mn = lane & 31, kk = lane >> 5;
    const int wm_idx = WM == 4 ? w : (w & 1), wn_idx = WM == 4 ? 0 : (w >> 1);

    float16v acc[NSUB][TAPS];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[sub][t][v] = 0.0f;

    for (int s0 = s_begin; s0 < s_end; s0 += a.G) {
        __syncthreads();        // the previous stage's operands are consumed (and, the first time, LDS is zeroed)
        td.commit(lds_d, a.sd, 0, a.L, a.G, a.tpr_log2);
        tx.commit(lds_x, a.sx, TAPS == 3 ? 1 : 0, a.L, a.G, a.tpr_log2);
        __syncthreads();
        if (s0 + a.G < s_end) {     // the next stage's loads fly during this stage's MFMAs
            td.issue(a.dy, a.Co, a.L, co0, s0 + a.G, s_end, a.G, a.tpr_log2);
            tx.issue(a.x, a.Ci, a.L, ci0, s0 + a.G, s_end, a.G, a.tpr_log2);
        }
        const int gv = s_end - s0 < a.G ? s_end - s0 : a.G;
        for (int g = 0; g < gv; ++g) {
            const float *pa = lds_d + (g * MT + 32 * wm_idx + mn) * a.sd + kk;
            const float *pb[NSUB];
            float b0[NSUB], b1[NSUB], b2[NSUB], b3[NSUB];
            float av = pa[0];
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                pb[sub] = lds_x + (g * kWgCiT + 32 * (wn_idx * NSUB + sub) + mn) * a.sx + kk;
                b0[sub] = pb[sub][0];
                b1[sub] = pb[sub][1];
                b2[sub] = pb[sub][2];
                b3[sub] = pb[sub][3];
            }
            if constexpr (TAPS == 3) {
                for (int i = 0; i < a.Lh; ++i) {
                    // operands of the NEXT k-step first (the row's zero tail makes the read past the last step
                    // harmless), then this step's MFMAs: the LDS latency hides behind them
                    const float an = pa[2 * i + 2];
                    float n2[NSUB], n3[NSUB];
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub) {
                        n2[sub] = pb[sub][2 * i + 4];
                        n3[sub] = pb[sub][2 * i + 5];
                    }
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub) {
                        acc[sub][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[sub], acc[sub][0], 0, 0, 0);
                        acc[sub][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[sub], acc[sub][1], 0, 0, 0);
                        acc[sub][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2[sub], acc[sub][2], 0, 0, 0);
                        b0[sub] = b2[sub];      // tap 0 of the next k-step (positions + 2) is tap 2 of this one
                        b1[sub] = b3[sub];
                        b2[sub] = n2[sub];
                        b3[sub] = n3[sub];
                    }
                    av = an;
                }
            } else {
                for (int i = 0; i < a.Lh; ++i) {
                    const float an = pa[2 * i + 2];
                    float n0[NSUB];
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub) n0[sub] = pb[sub][2 * i + 2];
#pragma unroll
                    for (int sub = 0; sub < NSUB; ++sub) {
                        acc[sub][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[sub], acc[sub][0], 0, 0, 0);
                        b0[sub] = n0[sub];
                    }
                    av = an;
                }
            }
        }
    }

    // partial[split][tap][co][ci]: the 32 lanes of a row write 128 contiguous bytes
    float *out = a.partial + (long long)split * TAPS * a.Co * a.Ci;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int ci = ci0 + 32 * (wn_idx * NSUB + sub) + mn;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = co0 + 32 * wm_idx + (v & 3) + 8 * (v >> 2) + 4 * kk;
                if (co < a.Co && ci < a.Ci) out[((long long)t * a.Co + co) * a.Ci + ci] = acc[sub][t][v];
            }
    }
}

// dw[co][ci][t] = sum over splits (float64 accumulation).  32 consecutive weights x 8 groups of splits per
// workgroup: a few hundred splits of a small layer (64 x 64 weights) would otherwise be one serial chain per thread
template <int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ partial, int nsplit, int Co,
                                                                int Ci, float *__restrict__ dw)
{
    __shared__ double s_part[8][TAPS][32];
    const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    const long long plane = (long long)Co * Ci;
    double s[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s[t] = 0.0;
    if (e < plane) {
        const float *p = partial + e + (long long)grp * TAPS * plane;
#pragma unroll 4
        for (int k = grp; k < nsplit; k += 8, p += 8 * TAPS * plane)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) s[t] += (double)p[t * plane];
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s_part[grp][t][el] = s[t];
    __syncthreads();
    if (threadIdx.x < 32 * TAPS) {
        const int t = threadIdx.x >> 5;
        double acc = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) acc += s_part[g][t][el];
        if (e < plane) dw[TAPS * (long long)e + t] = (float)acc;
    }
}

bool make_wgrad(int S, int Ci, int Co, int L, bool aligned, WgradArgs *a, int *wm, size_t *lds_bytes)
{
    if (S < 1 || Ci < 1 || Co < 1 || L < 1 || L > 256) return false;
    a->S = S; a->Ci = Ci; a->Co = Co; a->L = L;
    a->Lh = (L + 1) / 2;
    a->sd = 2 * a->Lh + 3;      // odd; two floats of zero tail for the look-ahead read
    a->sx = 2 * a->Lh + 5;
    *wm = Co > 64 ? 4 : 2;
    const int MT = *wm == 4 ? 128 : 64;
    const int per_seq = (MT * a->sd + kWgCiT * a->sx) * 4;
    a->vec = !aligned ? 1 : (L % 4 == 0) ? 4 : (L % 2 == 0) ? 2 : 1;
    const int per_row = (L + a->vec - 1) / a->vec;
    a->tpr_log2 = 0;
    while ((1 << a->tpr_log2) < per_row) ++a->tpr_log2;
    if (a->tpr_log2 > 8) return false;
    int G = kWgLdsBudget / per_seq;
    if (G < 1) G = 1;
    if (G > 8) G = 8;
    // a thread keeps a stage's loads in registers: at most stage_floats(vec) floats for the 128-row tile, i.e.
    // G * 128 rows / (256 >> tpr_log2 rows per pass) passes of `vec` floats
    const int max_g = (stage_floats(a->vec) / a->vec) * (kWgThreads >> a->tpr_log2) / 128;
    if (max_g < 1) return false;
    if (G > max_g) G = max_g;
    a->G = G;
    *lds_bytes = (size_t)G * per_seq;
    if (*lds_bytes > 64 * 1024) return false;          // L <= 256 with G = 1: 128 * 257 + 64 * 259 floats = 198 KB
    const int co_tiles = (Co + MT - 1) / MT;
    a->ci_tiles = (Ci + kWgCiT - 1) / kWgCiT;
    const long long tiles = (long long)co_tiles * a->ci_tiles;
    if (tiles > 65535) return false;
    a->tiles = (int)tiles;
    // sequences per workgroup: ~512 workgroups in all (two per CU, what the registers allow: one round), so that
    // the partial tiles -- nsplit * 3 * Co * Ci floats written and read once more -- stay below the operands' bytes
    long long spw = ((long long)S * tiles + 511) / 512;
    if (spw < 2 * G) spw = 2 * G;
    spw = (spw + G - 1) / G * G;
    a->spw = (int)spw;
    a->nsplit = (int)((S + spw - 1) / spw);
    return true;
}

}  // namespace

namespace {

int run_wgrad(const float *x, const float *dy, int S, int Ci, int Co, int L, int taps, float *dw, void *workspace,
              size_t workspace_bytes, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!x || !dy || !dw || !workspace) return POF_E_BADARG;
    if (S < 1 || Ci < 1 || Co < 1 || L < 1 || (taps != 1 && taps != 3)) return POF_E_BADARG;
    if ((long long)S * Ci * L >= (1LL << 31) * 4 || (long long)Co * Ci >= (1LL << 28)) return POF_E_SHAPE;
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    WgradArgs a;
    int wm;
    size_t lds;
    if (!make_wgrad(S, Ci, Co, L, aligned, &a, &wm, &lds)) return POF_E_SHAPE;
    if (workspace_bytes < (size_t)a.nsplit * taps * (size_t)Co * Ci * sizeof(float)) return POF_E_WORKSPACE;
    a.x = x; a.dy = dy; a.partial = static_cast<float *>(workspace);
    hipStream_t st = pof_stream(stream);
    const dim3 grid((unsigned)((a.nsplit + 7) / 8 * 8 * a.tiles));
#define POF_WGRAD(WM_, V_, T_) conv_wgrad_kernel<WM_, V_, T_><<<grid, kWgThreads, lds, st>>>(a)
#define POF_WGRAD_V(WM_, T_) \
    do { if (a.vec == 4) POF_WGRAD(WM_, 4, T_); else if (a.vec == 2) POF_WGRAD(WM_, 2, T_); else POF_WGRAD(WM_, 1, T_); } while (0)
    if (taps == 3) { if (wm == 4) POF_WGRAD_V(4, 3); else POF_WGRAD_V(2, 3); }
    else           { if (wm == 4) POF_WGRAD_V(4, 1); else POF_WGRAD_V(2, 1); }
#undef POF_WGRAD_V
#undef POF_WGRAD
    POF_CHECK_LAUNCH();
    if (taps == 3) conv_wgrad_reduce_kernel<3><<<(Co * Ci + 31) / 32, 256, 0, st>>>(a.partial, a.nsplit, Co, Ci, dw);
    else conv_wgrad_reduce_kernel<1><<<(Co * Ci + 31) / 32, 256, 0, st>>>(a.partial, a.nsplit, Co, Ci, dw);
    POF_CHECK_LAUNCH();
    return POF_OK;
}

size_t wgrad_workspace(int S, int Ci, int Co, int L, int taps)
{
    WgradArgs a;
    int wm;
    size_t lds;
    if ((taps != 1 && taps != 3) || !make_wgrad(S, Ci, Co, L, true, &a, &wm, &lds)) return 0;
    return (size_t)a.nsplit * taps * (size_t)Co * Ci * sizeof(float);
}

}  // namespace

extern "C" size_t pof_conv3_wgrad_workspace_bytes(int S, int Ci, int Co, int L) { return wgrad_workspace(S, Ci, Co, L, 3); }

extern "C" int pof_conv3_wgrad(const float *x, const float *dy, int S, int Ci, int Co, int L, float *dw,
                               void *workspace, size_t workspace_bytes, pof_stream_t stream)
{
    return run_wgrad(x, dy, S, Ci, Co, L, 3, dw, workspace, workspace_bytes, stream);
}

extern "C" size_t pof_conv1d_wgrad_workspace_bytes(int S, int Ci, int Co, int L, int kernel_size)
{
    return wgrad_workspace(S, Ci, Co, L, kernel_size);
}

extern "C" int pof_conv1d_wgrad(const float *x, const float *dy, int S, int Ci, int Co, int L, int kernel_size,
                                float *dw, void *workspace, size_t workspace_bytes, pof_stream_t stream)
{
    return run_wgrad(x, dy, S, Ci, Co, L, kernel_size, dw, workspace, workspace_bytes, stream);
}
