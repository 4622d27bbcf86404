// A9: Prototype._fusion (src/depracted/model/prototype.py:118-156), banded.
//
// feat1, feat2 [B][C][n] float32 -> out [B][D][n], D = 2*max_disp+1:
//   out[b,d,i] = sum_c sum_k f1[b,c,clamp(i+k-hk)] * f2[b,c,clamp(clamp(i+d-max_disp)+k-hk)]
// The reference forms the full n x n correlation with a GEMM and gathers the
// band; here only the band is computed (O(n*D) instead of O(n^2)), which is
// what keeps BASELINE config 5 (n = 450) HBM-bound: algorithmic bytes per
// sample 2*C*n*4 read + D*n*4 written (119 244 B at C=256, n=57), 8 flop/B.
//
// Forward: float32 MFMA on the per-position Gram matrix, band extracted from LDS (below).
#include "pof_common.h"

namespace {

// The banded correlation is a band of the per-position Gram matrix
//   P[a][b] = sum_c f1[c][a] * f2[c][b],     out[d][i] = sum_k P[A(i,k)][Bc(i,d,k)]
// with A = clamp(i+k-hk), Bc = clamp(clamp(i+d-md)+k-hk), and a 32 x 64 block of P is
// exactly what two float32 MFMAs per channel pair produce: v_mfma_f32_32x32x2_f32 takes ONE
// float per lane for each operand, lane l -> A[row l&31][k = l>>5], B[k = l>>5][col l&31], so
// the operands are plain coalesced global loads of the [C][n] rows (lanes 0-31 channel c,
// lanes 32-63 channel c+1) -- no LDS staging, no per-element VALU work.  Numerics: a k-ordered
// float32 fmaf chain over the channels (exact float32 products, one rounding per step).
//
// One wave = one sample x PTS = 32-(K-1) consecutive points: rows [i0-hk, i0-hk+32) of P
// against columns [i0-md-hk, i0-md-hk+64); positions outside [0, n) load as 0 and are never
// read back (the clamps stay inside the valid range).  The two accumulator tiles go to LDS
// and each lane sums K entries per output.  2.2x the band's flops, on the matrix pipe, which
// leaves the kernel HBM-bound: algorithmic bytes per sample 2*C*n*4 + D*n*4.
constexpr int kWavesPerBlock = 4;
constexpr int kMaxD = 15;    // max 2*max_disp+1
constexpr int kMaxK = 5;     // max kernel_size
constexpr int kPld = 65;     // LDS row stride of the staged 32 x 64 block of P

using f32x16 = float __attribute__((ext_vector_type(16)));

// T = float or _Float16 (BASELINE config 5: float16 storage, float32 products and accumulation --
// every float16 converts exactly, so the arithmetic after the load is the float32 kernel's).
template <int K, typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void band_corr_kernel(const T *f1, const T *f2,
                                                                        float *out, int B, int C, int n, int D,
                                                                        int nblk)
{
    constexpr int HK = K / 2;
    constexpr int PTS = 32 - (K - 1);
    __shared__ float s_p[kWavesPerBlock][32 * kPld];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit_raw = blockIdx.x * kWavesPerBlock + wave;  // (sample, point block)
    const bool live = unit_raw < B * nblk;                    // tail waves recompute the last unit, store nothing
    const int unit = live ? unit_raw : B * nblk - 1;
    const int b = unit / nblk, blk = unit - b * nblk;
    const int MD = D / 2;
    const int i0 = blk * PTS;
    const int ra = i0 - HK, cb = i0 - MD - HK;
    const int r = lane & 31, h = lane >> 5;
    // wave-uniform bases (scalar registers) + 32-bit per-lane offsets: global_load with an
    // SGPR base needs no 64-bit address VGPRs per load
    const T *g1 = f1 + (long long)b * C * n;
    const T *g2 = f2 + (long long)b * C * n;
    const int pa = ra + r, pb0 = cb + r, pb1 = cb + 32 + r;
    const bool va = pa >= 0 && pa < n, vb0 = pb0 >= 0 && pb0 < n, vb1 = pb1 >= 0 && pb1 < n;
    // clamped positions keep the masked lanes' (unused) loads inside the sample
    const int la = min(max(pa, 0), n - 1), l0 = min(max(pb0, 0), n - 1), l1 = min(max(pb1, 0), n - 1);
    const int hoff = h * n;   // upper half-wave: the odd channel of the pair

    f32x16 acc0 = {0}, acc1 = {0};
    // Channel pairs in blocks of U, register double-buffered: the loads of block k+1 are issued
    // before the 2U MFMAs of block k (one MFMA = 64 cycles, so a block covers ~1000 cycles of
    // memory latency per wave).  Indices past the end are clamped for the load and zeroed for
    // the MFMA; an odd last channel feeds zeros from the upper half-wave.
    constexpr int U = 8;
    const int csteps = (C + 1) >> 1;
    const long long step = 2LL * n;
    const bool odd_c = (C & 1) != 0;
    float xa[2][U], x0[2][U], x1[2][U];
    auto load_block = [&](int set, int cp0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cp = min(cp0 + u, csteps - 1);
            // the upper half-wave of an odd last step would read channel C: it re-reads C-1 (zeroed below)
            const int ho = (odd_c && cp == csteps - 1) ? 0 : hoff;
            const T *r1 = g1 + cp * step, *r2 = g2 + cp * step;   // uniform
            xa[set][u] = (float)r1[la + ho];
            x0[set][u] = (float)r2[l0 + ho];
            x1[set][u] = (float)r2[l1 + ho];
        }
    };
    auto mac_block = [&](int set, int cp0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cp = cp0 + u;
            const bool ok = cp < csteps && !(odd_c && cp == csteps - 1 && h == 1);
            const float a = (va && ok) ? xa[set][u] : 0.0f;
            const float b0 = (vb0 && ok) ? x0[set][u] : 0.0f;
            const float b1 = (vb1 && ok) ? x1[set][u] : 0.0f;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
        }
    };
    load_block(0, 0);
    for (int cp0 = 0; cp0 < csteps; cp0 += 2 * U) {
        load_block(1, cp0 + U);
        mac_block(0, cp0);
        if (cp0 + U < csteps) {
            load_block(0, cp0 + 2 * U);
            mac_block(1, cp0 + U);
        }
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *P = s_p[wave];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        P[row * kPld + r] = acc0[reg];
        P[row * kPld + 32 + r] = acc1[reg];
    }
    __syncthreads();
    float *o = out + (long long)b * D * n;
    for (int e = lane; e < D * PTS; e += 64) {
        const int d = e / PTS, il = e - d * PTS;
        const int i = i0 + il;
        if (i >= n || !live) continue;
        const int j = min(max(i + d - MD, 0), n - 1);
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int row = min(max(i + k - HK, 0), n - 1) - ra;
            const int col = min(max(j + k - HK, 0), n - 1) - cb;
            v += P[row * kPld + col];
        }
        o[(long long)d * n + i] = v;
    }
}

// n <= 64: the whole 64 x 64 Gram block of one sample in ONE wave.  Lane (r, h) loads the
// position pair (2r, 2r+1) of channel c+h with one 8-byte load per operand; the even and odd
// positions form two 32-row (32-column) MFMA operands, so each channel pair costs 2 loads and
// 4 MFMAs, every element of the sample is read exactly once, and P[row][col] is complete for
// any kernel size / displacement (all clamped indices lie in [0, n)).
constexpr int kSmallWaves = 4;
constexpr int kBandLd = 2 * (kMaxD / 2 + 2 * (kMaxK / 2)) + 2;   // 2 * 11 + 2: widest band, even stride

template <int K, typename T>
__global__ __launch_bounds__(64 * kSmallWaves) void band_corr_small_kernel(const T *f1, const T *f2,
                                                                           float *out, int B, int C, int n, int D)
{
    constexpr int HK = K / 2;
    __shared__ float s_p[kSmallWaves][64 * kBandLd];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b_raw = blockIdx.x * kSmallWaves + wave;
    const bool live = b_raw < B;
    const int b = live ? b_raw : B - 1;
    const int MD = D / 2;
    const int r = lane & 31, h = lane >> 5;
    const T *g1 = f1 + (long long)b * C * n;
    const T *g2 = f2 + (long long)b * C * n;
    // pair start clamped into the row; which half of the loaded pair is the even position
    const int p = max(min(2 * r, n - 2), 0);
    const bool ve = 2 * r < n, vo = 2 * r + 1 < n;
    const bool even_is_y = ve && (2 * r != p);       // odd n, last position: the pair is (n-2, n-1)
    const int hoff = h * n;

    f32x16 acc_ee = {0}, acc_eo = {0}, acc_oe = {0}, acc_oo = {0};
    constexpr int U = 8;
    const int csteps = (C + 1) >> 1;
    const long long step = 2LL * n;
    const bool odd_c = (C & 1) != 0;
    using F2 = float __attribute__((ext_vector_type(2)));
    F2 xa[2][U], xb[2][U];
    auto load_block = [&](int set, int cp0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cp = min(cp0 + u, csteps - 1);
            const int ho = (odd_c && cp == csteps - 1) ? 0 : hoff;
            const T *r1 = g1 + cp * step, *r2 = g2 + cp * step;   // uniform
            // element-aligned pair loads (odd rows of an odd-n tensor start on an element boundary)
            using T2 = T __attribute__((ext_vector_type(2)));
            T2 va, vb;
            __builtin_memcpy(&va, r1 + p + ho, sizeof(T2));
            __builtin_memcpy(&vb, r2 + p + ho, sizeof(T2));
            xa[set][u] = F2{(float)va.x, (float)va.y};
            xb[set][u] = F2{(float)vb.x, (float)vb.y};
        }
    };
    auto mac_block = [&](int set, int cp0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cp = cp0 + u;
            const bool ok = cp < csteps && !(odd_c && cp == csteps - 1 && h == 1);
            const F2 a = xa[set][u], bb = xb[set][u];
            const float ae = (ve && ok) ? (even_is_y ? a.y : a.x) : 0.0f;
            const float ao = (vo && ok) ? a.y : 0.0f;
            const float be = (ve && ok) ? (even_is_y ? bb.y : bb.x) : 0.0f;
            const float bo = (vo && ok) ? bb.y : 0.0f;
            acc_ee = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc_ee, 0, 0, 0);
            acc_eo = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, bo, acc_eo, 0, 0, 0);
            acc_oe = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, be, acc_oe, 0, 0, 0);
            acc_oo = __builtin_amdgcn_mfma_f32_32x32x2f32(ao, bo, acc_oo, 0, 0, 0);
        }
    };
    load_block(0, 0);
    for (int cp0 = 0; cp0 < csteps; cp0 += 2 * U) {
        load_block(1, cp0 + U);
        mac_block(0, cp0);
        if (cp0 + U < csteps) {
            load_block(0, cp0 + 2 * U);
            mac_block(1, cp0 + U);
        }
    }
    // Only the band |col - row| <= MD + 2*HK is ever read: keep P[row][col - row + HB] (HB-centred,
    // kBandLd wide) instead of the full 64 x 64 block -- 4 KB per wave, so LDS does not limit occupancy.
    // tile (pr, pc): row = 2 * rowidx + pr, col = 2 * r + pc, rowidx from the C/D layout
    float *P = s_p[wave];
    const int HB = MD + 2 * HK;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = 2 * ((reg & 3) + 8 * (reg >> 2) + 4 * h);
        const int c0 = 2 * r - row + HB;        // band slot of (row, 2r)
        if (c0 >= 0 && c0 < kBandLd) P[row * kBandLd + c0] = acc_ee[reg];
        if (c0 + 1 >= 0 && c0 + 1 < kBandLd) P[row * kBandLd + c0 + 1] = acc_eo[reg];
        if (c0 - 1 >= 0 && c0 - 1 < kBandLd) P[(row + 1) * kBandLd + c0 - 1] = acc_oe[reg];
        if (c0 >= 0 && c0 < kBandLd) P[(row + 1) * kBandLd + c0] = acc_oo[reg];
    }
    __syncthreads();
    float *o = out + (long long)b * D * n;
    for (int e = lane; e < D * n; e += 64) {
        const int d = e / n, i = e - d * n;
        if (!live) continue;
        const int j = min(max(i + d - MD, 0), n - 1);
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int row = min(max(i + k - HK, 0), n - 1);
            const int col = min(max(j + k - HK, 0), n - 1);
            v += P[row * kBandLd + col - row + HB];
        }
        o[e] = v;
    }
}

// Backward on the float32 MFMA.  With G[a][b] = dL/dP[a][b] (the output gradient scattered
// back onto the band of the Gram matrix, |b - a| <= HB = md + 2*hk):
//     d_f1[c][a] = sum_b f2[c][b] * G[a][b]          d_f2[c][b] = sum_a f1[c][a] * G[a][b]
// -- two [32 channels] x [32 positions] MFMA tiles per unit, contracted over the <= 32 + 2*HB
// positions of the band window.  The contraction runs over the contiguous axis of f, so the
// lanes of the A operand run along channels: lane (r, h) loads the 16 bytes
// f[c0 + r][q + 4h .. q + 4h + 3] (4 contraction steps per load; the k order inside an MFMA is
// free as long as both operands agree), the B operand is read from the band image of G in LDS.
//
// Phase 1 builds G deterministically (gather form: every band entry sums its contributions in a
// fixed order; no LDS atomics).  One workgroup = one sample; its 4 waves share G and split the
// (channel chunk, position block, which gradient) tiles.
constexpr int kBwdMaxN = 512;
constexpr int kBwdWaves = 4;

template <int K>
__global__ __launch_bounds__(64 * kBwdWaves) void band_corr_bwd_kernel(const float *f1, const float *f2,
                                                                      const float *g_out, float *d_f1, float *d_f2,
                                                                      int C, int n, int D)
{
    constexpr int HK = K / 2;
    extern __shared__ float s_g[];               // [n][W]: G[a][a - HB + s], then [D][n]: the output gradient
    const int MD = D / 2, HB = MD + 2 * HK, W = 2 * HB + 1;
    const int b = blockIdx.x;
    float *go = s_g + n * W;
    for (int e = threadIdx.x; e < D * n; e += 64 * kBwdWaves) go[e] = g_out[(long long)b * D * n + e];
    __syncthreads();

    // ---- phase 1: G[a][bb] = sum over (k, i, d) with A(i,k) = a and Bc(i,d,k) = bb of g[d][i] ----
    for (int e = threadIdx.x; e < n * W; e += 64 * kBwdWaves) {
        const int a = e / W, s = e - a * W;
        const int bb = a - HB + s;
        float acc = 0.0f;
        if (bb >= 0 && bb < n) {
            for (int k = 0; k < K; ++k) {
                // i with clamp(i + k - HK) == a: a single i in the interior, a run at either end
                int i_lo = a - k + HK, i_hi = i_lo;
                if (a == 0) i_lo = 0;
                if (a == n - 1) i_hi = n - 1;
                i_lo = max(i_lo, 0);
                i_hi = min(i_hi, n - 1);
                for (int i = i_lo; i <= i_hi; ++i) {
                    if (min(max(i + k - HK, 0), n - 1) != a) continue;
                    for (int d = 0; d < D; ++d) {
                        const int j = min(max(i + d - MD, 0), n - 1);
                        if (min(max(j + k - HK, 0), n - 1) == bb) acc += go[d * n + i];
                    }
                }
            }
        }
        s_g[e] = acc;
    }
    __syncthreads();

    // ---- phase 2: MFMA tiles -----------------------------------------------------------------
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int nchunk = (C + 31) >> 5, nblk = (n + 31) >> 5;
    const int units = nchunk * nblk * 2;
    for (int u = wave; u < units; u += kBwdWaves) {
        const int which = u & 1;                 // 0: d_f1 (reads f2), 1: d_f2 (reads f1)
        const int blk = (u >> 1) % nblk, chunk = (u >> 1) / nblk;
        const int c0 = chunk * 32, p0 = blk * 32;
        const float *src = (which == 0 ? f2 : f1) + (long long)b * C * n;
        float *dst = (which == 0 ? d_f1 : d_f2) + (long long)b * C * n;
        const int crow = min(c0 + r, C - 1);     // clamped row keeps masked lanes' loads in bounds
        const float *row = src + (long long)crow * n;
        const int w0 = max(p0 - HB, 0), w1 = min(p0 + 31 + HB, n - 1);
        f32x16 acc = {0};
        // all loads of the tile first (<= 8 groups of 8 positions: 32 + 2 * 11 + 7 < 64), then the MFMAs
        constexpr int kMaxQ = 8;
        float x[kMaxQ][4];
#pragma unroll
        for (int qi = 0; qi < kMaxQ; ++qi) {
            const int q = w0 + 8 * qi;
            const int pos0 = q + 4 * h;
            if (q <= w1) {
                if (q + 7 <= n - 1) {            // uniform: the whole 8-position group is inside the row
                    __builtin_memcpy(x[qi], row + pos0, 16);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) x[qi][t] = row[min(pos0 + t, n - 1)];
                }
            }
        }
        // B operands of the whole tile from the band image first (one LDS round trip), then the MFMAs.
        // G[a][bb] lives at s_g[a * W + bb - a + HB]: with the lane's fixed index `fx` (row a for d_f1,
        // column bb for d_f2) the address is linear in the contraction position `pos`
        //   d_f1: fx * (W - 1) + HB + pos            d_f2: pos * (W - 1) + HB + fx
        // and the entry exists iff |pos - fx| <= HB, pos < n and fx < n: one range test per element.
        float bvv[kMaxQ][4];
        const int fx = p0 + r;
        const int pos_lo = max(fx - HB, 0), pos_span = (fx < n ? min(fx + HB, n - 1) : -1) - pos_lo;   // < 0: no entry
        const int stride = which == 0 ? 1 : (W - 1);
        const int addr0 = (which == 0 ? fx * (W - 1) + HB : HB + fx) + (w0 + 4 * h) * stride;
#pragma unroll
        for (int qi = 0; qi < kMaxQ; ++qi) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = 8 * qi + t;                       // compile-time offset from this lane's first position
                const int pos = w0 + 4 * h + k;
                const bool ok = (unsigned)(pos - pos_lo) <= (unsigned)pos_span && pos_span >= 0;
                bvv[qi][t] = ok ? s_g[addr0 + k * stride] : 0.0f;
            }
        }
#pragma unroll
        for (int qi = 0; qi < kMaxQ; ++qi) {
            const int q = w0 + 8 * qi;
            if (q > w1) break;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // positions past the row end carry a zero B operand; rows c >= C are never stored
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[qi][t], bvv[qi][t], acc, 0, 0, 0);
            }
        }
        // C/D layout: col = lane & 31 (position), row = (reg & 3) + 8 * (reg >> 2) + 4 * h (channel)
        const int pcol = p0 + r;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int c = c0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (c < C && pcol < n) dst[(long long)c * n + pcol] = acc[reg];
        }
    }
}

// Backward for n <= 64 (the Prototype's feature maps: n = 57): the whole row of a channel is ONE wave-wide register
// (lane = position), so the band contraction needs no matrix unit and no operand staging:
//     d_f1[c][a] = sum_d G[a][a + d] * f2[c][a + d]          d_f2[c][b] = sum_d G[b + d][b] * f1[c][b + d]
// with |d| <= HB: a lane keeps its 2 * (2 HB + 1) band weights in registers for the whole sample (the row of G
// through a, the column through b), a channel costs two coalesced 256-byte row loads, 2 (W - 1) wave-wide DPP
// shifts, 2 W FMAs and two coalesced row stores.  The MFMA form above pads the <= 44-position contraction window to 64 and multiplies
// mostly structural zeros of the band (13-15 non-zeros per 44): 256 MFMAs + 3900 vector instructions per wave
// against ~4500 vector instructions here and no matrix work (profiles/r2_corr_bwd_pmc.txt).
// Phase 1 (the band image G in LDS) is the same gather as in band_corr_bwd_kernel.
constexpr int kBwdSmallMaxW = 19;       // 2 * HB + 1 <= 19: HB = max_disp + 2 * (K / 2) <= 9

template <int K, int HB>
__global__ __launch_bounds__(64 * kBwdWaves) void band_corr_bwd_small_kernel(const float *f1, const float *f2,
                                                                            const float *g_out, float *d_f1,
                                                                            float *d_f2, int C, int n, int D)
{
    constexpr int HK = K / 2, W = 2 * HB + 1;
    extern __shared__ float s_g[];               // [n][W]: G[a][a - HB + s], then [D][n]: the output gradient
    const int MD = D / 2;                        // HB == MD + 2 * HK (the launcher picks the instantiation)
    const int b = blockIdx.x;
    float *go = s_g + n * W;
    for (int e = threadIdx.x; e < D * n; e += 64 * kBwdWaves) go[e] = g_out[(long long)b * D * n + e];
    __syncthreads();
    for (int e = threadIdx.x; e < n * W; e += 64 * kBwdWaves) {
        const int a = e / W, s = e - a * W;
        const int bb = a - HB + s;
        float acc = 0.0f;
        if (bb >= 0 && bb < n) {
            for (int k = 0; k < K; ++k) {
                int i_lo = a - k + HK, i_hi = i_lo;
                if (a == 0) i_lo = 0;
                if (a == n - 1) i_hi = n - 1;
                i_lo = max(i_lo, 0);
                i_hi = min(i_hi, n - 1);
                for (int i = i_lo; i <= i_hi; ++i) {
                    if (min(max(i + k - HK, 0), n - 1) != a) continue;
                    for (int d = 0; d < D; ++d) {
                        const int j = min(max(i + d - MD, 0), n - 1);
                        if (min(max(j + k - HK, 0), n - 1) == bb) acc += go[d * n + i];
                    }
                }
            }
        }
        s_g[e] = acc;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = lane < n;
    // band weights of this lane's position p = lane, slot s <-> offset d = s - HB:
    //   wr[s] = G[p][p + d]  (row through p)       -> d_f1
    //   wc[s] = G[p + d][p]  (column through p)    -> d_f2;   G[a][b] lives at s_g[a * W + b - a + HB]
    float wr[W], wc[W];                         // slot HB + d <-> offset d
#pragma unroll
    for (int s = 0; s < W; ++s) {
        const int d = s - HB, q = lane + d;
        const bool ok = live && q >= 0 && q < n;
        wr[s] = ok ? s_g[lane * W + (HB + d)] : 0.0f;
        wc[s] = ok ? s_g[q * W + (HB - d)] : 0.0f;
    }
    // neighbours through the wave-wide DPP shifts of GFX9 (v_mov_b32_dpp wave_shl:1 / wave_shr:1: lane i reads lane
    // i + 1 / i - 1 across the row boundaries, the end lane reads 0 with bound_ctrl; tools/ubench/dpp_wave_shift.hip):
    // vector-pipe moves, no LDS traffic -- the ds_bpermute form of this loop is LDS-issue bound (0.46 ms)
    auto shl1 = [](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
    };
    auto shr1 = [](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
    };
    const int lc = live ? lane : n - 1;          // dead lanes load a valid address and are zeroed after the load
    const float *r1 = f1 + (long long)b * C * n + lc, *r2 = f2 + (long long)b * C * n + lc;
    float *o1 = d_f1 + (long long)b * C * n + lane, *o2 = d_f2 + (long long)b * C * n + lane;
    // next channel's rows are in flight while this one's shifts and FMAs run
    float nv1 = 0.0f, nv2 = 0.0f;
    if (wave < C) { nv1 = r1[(long long)wave * n]; nv2 = r2[(long long)wave * n]; }
    for (int c = wave; c < C; c += kBwdWaves) {
        const float v1 = live ? nv1 : 0.0f, v2 = live ? nv2 : 0.0f;
        const int cn = c + kBwdWaves < C ? c + kBwdWaves : c;
        nv1 = r1[(long long)cn * n];
        nv2 = r2[(long long)cn * n];
        float a1 = wr[HB] * v2, a2 = wc[HB] * v1;
        float p1 = v1, p2 = v2, m1 = v1, m2 = v2;       // values of lane + d / lane - d
#pragma unroll
        for (int d = 1; d <= HB; ++d) {
            p1 = shl1(p1); p2 = shl1(p2);
            m1 = shr1(m1); m2 = shr1(m2);
            a1 = fmaf(wr[HB + d], p2, a1);
            a1 = fmaf(wr[HB - d], m2, a1);
            a2 = fmaf(wc[HB + d], p1, a2);
            a2 = fmaf(wc[HB - d], m1, a2);
        }
        if (live) {     // ordinary stores: the 228-byte rows are partial lines that L2 merges (non-temporal: 0.255 -> 0.285 ms)
            o1[(long long)c * n] = a1;
            o2[(long long)c * n] = a2;
        }
    }
}

template <int K, typename T>
void launch_k(const T *f1, const T *f2, float *out, int B, int C, int n, int D, hipStream_t s)
{
    if (n >= 2 && n <= 64) {
        band_corr_small_kernel<K, T><<<(B + kSmallWaves - 1) / kSmallWaves, 64 * kSmallWaves, 0, s>>>(f1, f2, out, B,
                                                                                                   C, n, D);
        return;
    }
    constexpr int PTS = 32 - (K - 1);
    const int nblk = (n + PTS - 1) / PTS;
    const long long units = (long long)B * nblk;
    const unsigned grid = (unsigned)((units + kWavesPerBlock - 1) / kWavesPerBlock);
    band_corr_kernel<K, T><<<grid, 64 * kWavesPerBlock, 0, s>>>(f1, f2, out, B, C, n, D, nblk);
}

template <typename T>
int band_corr_entry(const T *feat1, const T *feat2, float *out, int B, int C, int n, int kernel_size, int max_disp,
                    pof_stream_t stream)
{
    if (!feat1 || !feat2 || !out || B < 0 || C < 1 || n < 1) return POF_E_BADARG;
    if (kernel_size < 1 || kernel_size > kMaxK || (kernel_size & 1) == 0) return POF_E_SHAPE;
    if (max_disp < 0 || 2 * max_disp + 1 > kMaxD) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    const int D = 2 * max_disp + 1;
    hipStream_t s = pof_stream(stream);
    switch (kernel_size) {
        case 1: launch_k<1, T>(feat1, feat2, out, B, C, n, D, s); break;
        case 3: launch_k<3, T>(feat1, feat2, out, B, C, n, D, s); break;
        default: launch_k<5, T>(feat1, feat2, out, B, C, n, D, s); break;
    }
    POF_CHECK_LAUNCH();
    return POF_OK;
}

}  // namespace

extern "C" int pof_band_correlation(const float *feat1, const float *feat2, float *out, int B, int C,
                                    int n, int kernel_size, int max_disp, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return band_corr_entry<float>(feat1, feat2, out, B, C, n, kernel_size, max_disp, stream);
}

extern "C" int pof_band_correlation_f16(const void *feat1_f16, const void *feat2_f16, float *out, int B, int C,
                                        int n, int kernel_size, int max_disp, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return band_corr_entry<_Float16>(static_cast<const _Float16 *>(feat1_f16),
                                     static_cast<const _Float16 *>(feat2_f16), out, B, C, n, kernel_size, max_disp,
                                     stream);
}

extern "C" int pof_band_correlation_backward(const float *feat1, const float *feat2, const float *g_out,
                                             float *d_feat1, float *d_feat2, int B, int C, int n,
                                             int kernel_size, int max_disp, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!feat1 || !feat2 || !g_out || !d_feat1 || !d_feat2 || B < 0 || C < 1 || n < 1) return POF_E_BADARG;
    if (kernel_size < 1 || kernel_size > kMaxK || (kernel_size & 1) == 0) return POF_E_SHAPE;
    if (max_disp < 0 || 2 * max_disp + 1 > kMaxD) return POF_E_SHAPE;
    if (n > kBwdMaxN) return POF_E_SHAPE;
    if (B == 0) return POF_OK;
    if (B > 65535) return POF_E_SHAPE;
    const int D = 2 * max_disp + 1;
    const int W = 2 * (max_disp + 2 * (kernel_size / 2)) + 1;
    const size_t lds = (size_t)n * (W + D) * sizeof(float);     // <= 512 * (23 + 15) * 4 = 78 KB
    if (lds > 64 * 1024) {
        const void *fn = kernel_size == 1 ? reinterpret_cast<const void *>(band_corr_bwd_kernel<1>)
                         : kernel_size == 3 ? reinterpret_cast<const void *>(band_corr_bwd_kernel<3>)
                                            : reinterpret_cast<const void *>(band_corr_bwd_kernel<5>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return POF_E_LAUNCH;
    }
    hipStream_t s = pof_stream(stream);
    if (n <= 64 && W <= kBwdSmallMaxW) {        // one wave holds a whole row: the shuffle form, one instantiation per band width
        const int HB = W / 2;
#define POF_BWD_SMALL(K_, HB_) band_corr_bwd_small_kernel<K_, HB_><<<B, 64 * kBwdWaves, lds, s>>>(feat1, feat2, g_out, d_feat1, d_feat2, C, n, D)
#define POF_BWD_SMALL_K(K_)                                                                   \
        switch (HB) {                                                                        \
            case 0: POF_BWD_SMALL(K_, 0); break; case 1: POF_BWD_SMALL(K_, 1); break;        \
            case 2: POF_BWD_SMALL(K_, 2); break; case 3: POF_BWD_SMALL(K_, 3); break;        \
            case 4: POF_BWD_SMALL(K_, 4); break; case 5: POF_BWD_SMALL(K_, 5); break;        \
            case 6: POF_BWD_SMALL(K_, 6); break; case 7: POF_BWD_SMALL(K_, 7); break;        \
            case 8: POF_BWD_SMALL(K_, 8); break; default: POF_BWD_SMALL(K_, 9); break;       \
        }
        if (kernel_size == 1) { POF_BWD_SMALL_K(1) } else if (kernel_size == 3) { POF_BWD_SMALL_K(3) } else { POF_BWD_SMALL_K(5) }
#undef POF_BWD_SMALL_K
#undef POF_BWD_SMALL
        POF_CHECK_LAUNCH();
        return POF_OK;
    }
    switch (kernel_size) {
        case 1: band_corr_bwd_kernel<1><<<B, 64 * kBwdWaves, lds, s>>>(feat1, feat2, g_out, d_feat1, d_feat2, C, n, D); break;
        case 3: band_corr_bwd_kernel<3><<<B, 64 * kBwdWaves, lds, s>>>(feat1, feat2, g_out, d_feat1, d_feat2, C, n, D); break;
        default: band_corr_bwd_kernel<5><<<B, 64 * kBwdWaves, lds, s>>>(feat1, feat2, g_out, d_feat1, d_feat2, C, n, D); break;
    }
    POF_CHECK_LAUNCH();
    return POF_OK;
}
