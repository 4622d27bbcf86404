// N2: the DROW / DR-SPAAM conv trunk layer for inference:
//   y = max_pool1d?( LeakyReLU_0.1( BatchNorm_eval( Conv1d(k = 3, pad = 1)(x) ) ), 2 )
// (src/depracted/model/dr_spaam.py:8-19 `_conv3x3`, :86-92 `_forward_conv`) on S short
// sequences at once: x [S][Ci][L] -> out [S][Co][L or L/2], float32.
//
// 450 x T x B sequences of 56 / 28 / 14 / 7 points: MIOpen falls to its naive / im2col paths
// on these shapes (6 TFLOP/s end to end).  Here the layer is an implicit GEMM on the float32
// MFMA (v_mfma_f32_32x32x2_f32): rows = output channels, columns = (sequence, position),
// K = (tap, input channel).
//   * B operand (activations): lane (n, h) needs x[seq(n)][ci + h][l(n) + tap - 1]: for a fixed
//     (tap, ci) the 32 columns are consecutive floats, so the operand is a coalesced global load
//     (zero at the sequence borders); the three taps re-read the same lines from L1.
//   * A operand (weights): pre-transposed on the host to [tap][ci][co]; a K-chunk (4 input
//     channels x 3 taps) x (128 output channels) is staged in LDS per workgroup and every lane
//     reads consecutive co.
//   * one wave = 32 columns x up to 128 output channels (4 accumulator tiles): one activation
//     load feeds 4 MFMAs; the next chunk's loads are in flight during the MFMAs; 116 VGPRs -> 4 waves per SIMD.
//   * epilogue: y = acc * scale[co] + shift[co] (BatchNorm folded with the conv bias), LeakyReLU,
//     optional max over position pairs (adjacent lanes), store.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "pof_common.h"

namespace {

constexpr int kCvWaves = 4;
constexpr int kCvCC = 4;             // input channels per LDS weight chunk
constexpr int kCvRows = 3 * kCvCC;   // K rows per chunk (tap-major), kernel width 3 (the split-K form)
constexpr long long kCvFillWorkgroups = 256;    // one workgroup = one wave per SIMD on each of the 256 CUs
using f32x16 = float __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float *x;       // [S][Ci][L]
    const float *wt;      // [3][Ci][Co]
    const float *scale;   // [Co]
    const float *shift;   // [Co]
    float *out;           // [S][Co][Lout]
    int S, Ci, Co, L, pool;
    int Lc;               // positions the convolution produces per sequence: L (stride 1), (L + 1) / 2 (stride 2)
    float slope;
    // FUSE1 (the trunk's first two layers in one launch): x is the single-channel input [S][L]; input channel c of
    // this convolution is computed on the fly, lrelu(a0 x[q-1] + a1 x[q] + a2 x[q+1] + b) with l1[c] = {a0, a1, a2, b}
    // (the first layer's taps times its folded BatchNorm scale, and its shift), slope1 its LeakyReLU slope
    const float *l1;      // [Ci][4]
    float slope1;
};

// Epilogue shared by both kernels.  C/D layout of v_mfma_f32_32x32x2_f32: column n = lane & 31,
// output channel = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) inside a 32-channel tile.
//   y = acc * scale[co] + shift[co] (BatchNorm folded with the conv bias; the two vectors of the workgroup's
//   channel tile sit in LDS, four consecutive channels per 16-byte read), LeakyReLU as max(y, slope * y) for
//   0 <= slope <= 1, the pooled pair on adjacent lanes through one DPP move, stores as uniform base + 32-bit
//   offset.  (The first version did two global loads, a 64-bit multiply-add, a select chain and a
//   ds_bpermute per output: ~1300 vector instructions per wave, 8 % of a 128 -> 128 layer.)
template <int CT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, const f32x16 (&acc)[CT], const float *s_scale,
                                              const float *s_shift, int co0, int seq, int l, int h, bool col_ok)
{
    const int Lout = a.pool ? a.Lc / 2 : a.Lc;
    const bool slope01 = a.slope >= 0.0f && a.slope <= 1.0f;                          // uniform
    const long long obase = (long long)seq * a.Co * Lout + (a.pool ? l / 2 : l);      // element offset of (seq, co = 0, l)
    const bool small = (long long)a.S * a.Co * Lout < (1LL << 30);                    // uniform: 32-bit byte offsets
    const bool writer = col_ok && (!a.pool || (!(l & 1) && l + 1 < a.Lc));
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const bool tile_full = co0 + t * 32 + 31 < a.Co;                               // uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = t * 32 + 8 * g + 4 * h;                                     // first of 4 consecutive channels
            const float4 sc = *reinterpret_cast<const float4 *>(s_scale + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(s_shift + cl);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = acc[t][4 * g + j] * scv[j] + shv[j];
                y = slope01 ? fmaxf(y, y * a.slope) : (y > 0.0f ? y : y * a.slope);
                if (a.pool) {
                    // the pair (2j, 2j+1) sits on adjacent lanes: quad_perm [1,0,3,2]
                    const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, y), 0xB1, 0xF, 0xF, true));
                    y = fmaxf(y, other);
                }
                const int co = co0 + cl + j;
                if (writer && (tile_full || co < a.Co)) {
                    if (small) a.out[(unsigned)(obase + (long long)co * Lout)] = y;
                    else a.out[obase + (long long)co * Lout] = y;
                }
            }
        }
    }
}

// KW taps (1 or 3, padding KW / 2), STRIDE 1 or 2 (round 3: the Prototype's stride-2 encoders and the point-wise heads
// leave the library as well); the DR-SPAAM trunk is <CT, 3, 1>.
// FUSE1: the DR-SPAAM trunk's first layer (1 -> 64, 0.3 ms of pure 1 GB write at B = 32, read back by the second
// layer) is folded into the second: the B operand of channel c is three FMAs, a multiply and a max on the lane's five
// input values instead of a global load -- a few hundred vector instructions per wave in the shadow of its MFMAs.
template <int CT, int KW, int STRIDE, bool FUSE1 = false>
__global__ __launch_bounds__(64 * kCvWaves, 4) void conv1d_kernel(ConvArgs a)
{
    static_assert(!FUSE1 || (KW == 3 && STRIDE == 1), "the fused first layer is a k = 3, stride 1 convolution");
    // input channels per LDS weight chunk: 4 x 3 taps = 12 K rows, or 16 x 1 tap -- the point-wise form would otherwise
    // meet a workgroup barrier every two k-steps (1024 -> 128, 64 positions x 256 sequences: 145 -> see DESIGN us)
    constexpr int CC = KW == 1 ? 16 : kCvCC;
    constexpr int kRows = KW * CC;                  // K rows per chunk (tap-major)
    constexpr int kPad = KW / 2;
    constexpr int COG = 32 * CT;                       // output channels per workgroup
    constexpr int NT = 64 * kCvWaves;
    constexpr int NP = CC / 2;                      // channel pairs (k-steps) per tap and chunk
    __shared__ __attribute__((aligned(16))) float s_w[2][kRows][COG];   // double-buffered weight chunk
    __shared__ __attribute__((aligned(16))) float s_scale[COG], s_shift[COG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int co0 = blockIdx.y * COG;
    if (threadIdx.x < COG) {      // visible to every wave after the first barrier of the K loop
        const int cs = min(co0 + (int)threadIdx.x, a.Co - 1);
        s_scale[threadIdx.x] = a.scale[cs];
        s_shift[threadIdx.x] = a.shift[cs];
    }
    constexpr int kL1Max = 128;                        // input channels the fused form takes
    __shared__ __attribute__((aligned(16))) float s_l1[FUSE1 ? kL1Max : 1][4];
    if (FUSE1) {
        for (int i = threadIdx.x; i < a.Ci * 4; i += NT) (&s_l1[0][0])[i] = a.l1[i];
    }
    const long long ncol = (long long)a.S * a.Lc;
    const long long n_g = ((long long)blockIdx.x * kCvWaves + wave) * 32 + r;   // this lane's column
    const bool col_ok = n_g < ncol;
    const long long nc = col_ok ? n_g : ncol - 1;
    const int seq = (int)(nc / a.Lc), l = (int)(nc - (long long)seq * a.Lc);    // l: output position
    bool tap_ok[KW];
#pragma unroll
    for (int tap = 0; tap < KW; ++tap) {
        const int li = l * STRIDE + tap - kPad;                                   // input position of this tap
        tap_ok[tap] = col_ok && li >= 0 && li < a.L;
    }
    float xs[FUSE1 ? 5 : 1];                           // FUSE1: x[l - 2 .. l + 2] of the lane's sequence (0 outside)
    if (FUSE1) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int q = l + j - 2;
            xs[j] = (col_ok && q >= 0 && q < a.L) ? a.x[(long long)seq * a.L + q] : 0.0f;
        }
        __syncthreads();                               // s_l1 is read by the first load_x below
    }
    // 32-bit element offsets of x[seq][h][l + tap - 1] relative to the (uniform) channel row base:
    // border / tail lanes point at a valid neighbour and are zeroed after the load
    // (BYTE offsets: SGPR base + zero-extended 32-bit VGPR offset is the global_load saddr form,
    // which needs no 64-bit address registers per load)
    const unsigned base_off = (unsigned)((long long)seq * a.Ci * a.L + l * STRIDE);
    unsigned off_h[KW], off_0[KW];
#pragma unroll
    for (int tap = 0; tap < KW; ++tap) {
        // a tap outside the sequence points at the lane's own centre element (always inside: l * STRIDE < L) and is
        // zeroed after the load
        off_0[tap] = (base_off + (tap_ok[tap] ? tap - kPad : 0)) * 4u;
        off_h[tap] = off_0[tap] + (unsigned)(h * a.L) * 4u;
    }
    const int nchunk = (a.Ci + CC - 1) / CC;

    // weight chunk -> registers (16-byte loads along co; uniform chunk base + per-thread byte offsets
    // that do not change from chunk to chunk), registers -> LDS
    constexpr int WV = (kRows * COG / 4 + NT - 1) / NT;       // float4 groups per thread and chunk
    using F4V = float __attribute__((ext_vector_type(4)));
    F4V wreg[WV];
    unsigned woff[WV];      // ((tap * Ci + cl) * Co + c) * 4 bytes
    int wcl[WV];            // cl, or CC when the group is outside the tile / the tensor
    const bool wvec = (a.Co & 3) == 0;
#pragma unroll
    for (int q = 0; q < WV; ++q) {
        const int e4 = threadIdx.x + q * NT;
        const int row = e4 / (COG / 4), c = (e4 - row * (COG / 4)) * 4;
        const int tap = row / CC, cl = row - tap * CC;
        const bool in = e4 < kRows * COG / 4 && co0 + c < a.Co;
        woff[q] = in ? (unsigned)(((long long)tap * a.Ci + cl) * a.Co + c) * 4u : 0u;
        wcl[q] = in ? cl : CC;
    }
    auto load_w = [&](int ci0) {
        const int cc = min(CC, a.Ci - ci0);
        const char *wbase = reinterpret_cast<const char *>(a.wt + (long long)ci0 * a.Co + co0);   // uniform
#pragma unroll
        for (int q = 0; q < WV; ++q) {
            F4V v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (wcl[q] < cc) {
                if (wvec) {
                    v = *reinterpret_cast<const F4V *>(wbase + woff[q]);
                } else {   // Co not a multiple of 4: element-wise with a column guard
                    const float *pw = reinterpret_cast<const float *>(wbase + woff[q]);
                    const int e4 = threadIdx.x + q * NT;
                    const int c = (e4 % (COG / 4)) * 4;
                    for (int j = 0; j < 4; ++j) v[j] = (co0 + c + j < a.Co) ? pw[j] : 0.0f;
                }
            }
            wreg[q] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int q = 0; q < WV; ++q) {
            const int e4 = threadIdx.x + q * NT;
            if (e4 < kRows * COG / 4) reinterpret_cast<F4V *>(&s_w[buf][0][0])[e4] = wreg[q];
        }
    };
    // activation operands of one chunk: 3 taps x NP channel pairs, uniform row base + lane offset
    float xb[2][KW][NP];
    auto load_x = [&](int set, int ci0) {
        const int cc = min(CC, a.Ci - ci0);
        if constexpr (FUSE1) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int c = min(ci0 + 2 * p + h, a.Ci - 1);
                const float4 k4 = *reinterpret_cast<const float4 *>(&s_l1[c][0]);
#pragma unroll
                for (int tap = 0; tap < KW; ++tap) {
                    const float y = fmaf(k4.z, xs[tap + 2], fmaf(k4.y, xs[tap + 1], fmaf(k4.x, xs[tap], k4.w)));
                    xb[set][tap][p] = fmaxf(y, y * a.slope1);          // LeakyReLU, 0 <= slope <= 1
                }
            }
            return;
        }
#pragma unroll
        for (int tap = 0; tap < KW; ++tap)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int c2 = min(2 * p, cc - 1);                               // uniform
                const char *rowp = reinterpret_cast<const char *>(a.x + (long long)(ci0 + c2) * a.L);  // uniform base
                const bool pair = 2 * p + 1 < cc;                                // uniform: the odd channel exists
                xb[set][tap][p] = *reinterpret_cast<const float *>(rowp + (pair ? off_h[tap] : off_0[tap]));
            }
    };

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0};

    load_w(0);
    load_x(0, 0);
    store_w(0);
    __syncthreads();
    auto chunk = [&](auto set_tag, const int ch) {
        constexpr int SET = decltype(set_tag)::value;          // activation set / LDS buffer of this chunk
        const int ci0 = ch * CC;
        const int cc = min(CC, a.Ci - ci0);
        const bool more = ch + 1 < nchunk;
        float4 k4n[FUSE1 ? NP : 1];       // FUSE1: the next chunk's first-unit coefficients of this lane's channels
        if (more) {                       // next chunk's weights and activations are in flight during the MFMAs
            load_w(ci0 + CC);
            if constexpr (FUSE1) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    k4n[p] = *reinterpret_cast<const float4 *>(&s_l1[min(ci0 + CC + 2 * p + h, a.Ci - 1)][0]);
            } else {
                load_x(SET ^ 1, ci0 + CC);
            }
        }
        // A operand one k-step ahead of its MFMAs
        float a_cur[CT], a_nxt[CT];
#pragma unroll
        for (int t = 0; t < CT; ++t) a_cur[t] = s_w[SET][h][t * 32 + r];
#pragma unroll
        for (int ks = 0; ks < KW * NP; ++ks) {
            const int tap = ks / NP, p = ks - tap * NP;
            if (ks + 1 < KW * NP) {
                const int tn = (ks + 1) / NP, pn = (ks + 1) - tn * NP;
#pragma unroll
                for (int t = 0; t < CT; ++t) a_nxt[t] = s_w[SET][tn * CC + 2 * pn + h][t * 32 + r];
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool ok = tap_ok[tap] && (2 * p + h < cc);
            const float bv = ok ? xb[SET][tap][p] : 0.0f;
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bv, acc[t], 0, 0, 0);
            if constexpr (FUSE1) {
                // one value of the NEXT chunk per k-step, issued behind this step's MFMAs (the matrix pipe runs them
                // while the vector ALU does these five instructions)
                if (more) {
                    const float y = fmaf(k4n[p].z, xs[tap + 2], fmaf(k4n[p].y, xs[tap + 1], fmaf(k4n[p].x, xs[tap], k4n[p].w)));
                    xb[SET ^ 1][tap][p] = fmaxf(y, y * a.slope1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < CT; ++t) a_cur[t] = a_nxt[t];
        }
        if (more) store_w(SET ^ 1);
        __syncthreads();      // (measured: the per-chunk barriers cost 1 % of the trunk)
    };
    for (int ch = 0; ch < nchunk; ch += 2) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < nchunk) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
    conv_epilogue<CT>(a, acc, s_scale, s_shift, co0, seq, l, h, col_ok);
}

// ---- split-K form for small launches (streaming inference: one scan per call) -------------------------
// With a few hundred column tiles the layer cannot give every SIMD a wave even at 32 output channels per
// workgroup, and a wave that owns a whole K loop (up to 768 dependent k-steps) runs at a quarter of the MFMA
// issue rate: nothing else is resident to hide its LDS / load latency (86-92 us per layer against an 18 us issue
// bound, profiles/r1i_conv_small_*).  Here the FOUR WAVES OF A WORKGROUP SHARE ONE 32-COLUMN TILE AND SPLIT K:
// wave w takes a contiguous quarter of the channel chunks, stages its own weight chunks in a private LDS
// region (no workgroup barrier inside the loop), and the four partial accumulators meet in LDS at the end,
// summed in wave order by wave 0, which runs the epilogue.  Four times the workgroups, a quarter of the
// serial chain.  (Summation order differs from conv3_kernel; exact on integer data, deterministic.)
template <int CT>
__global__ __launch_bounds__(64 * kCvWaves, 2) void conv3_splitk_kernel(ConvArgs a)
{
    constexpr int COG = 32 * CT;
    constexpr int NP = kCvCC / 2;
    constexpr int WV = (kCvRows * COG / 4 + 63) / 64;           // float4 groups per lane and chunk
    __shared__ __attribute__((aligned(16))) float s_w[kCvWaves][2][kCvRows][COG];
    __shared__ __attribute__((aligned(16))) float s_acc[kCvWaves - 1][CT * 16][64];
    __shared__ __attribute__((aligned(16))) float s_scale[COG], s_shift[COG];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int co0 = blockIdx.y * COG;
    if (threadIdx.x < COG) {      // visible to wave 0 after the barrier in front of the reduction
        const int cs = min(co0 + (int)threadIdx.x, a.Co - 1);
        s_scale[threadIdx.x] = a.scale[cs];
        s_shift[threadIdx.x] = a.shift[cs];
    }
    const long long ncol = (long long)a.S * a.L;
    const long long n_g = (long long)blockIdx.x * 32 + r;        // every wave of the workgroup: the same column
    const bool col_ok = n_g < ncol;
    const long long nc = col_ok ? n_g : ncol - 1;
    const int seq = (int)(nc / a.L), l = (int)(nc - (long long)seq * a.L);
    const bool tap_ok[3] = {col_ok && l > 0, col_ok, col_ok && l < a.L - 1};
    const unsigned base_off = (unsigned)((long long)seq * a.Ci * a.L + l);
    unsigned off_h[3], off_0[3];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        off_0[tap] = (base_off + (tap_ok[tap] ? tap - 1 : 0)) * 4u;
        off_h[tap] = off_0[tap] + (unsigned)(h * a.L) * 4u;
    }
    const int nchunk = (a.Ci + kCvCC - 1) / kCvCC;
    const int per = (nchunk + kCvWaves - 1) / kCvWaves;
    const int ch_lo = min(wave * per, nchunk), ch_hi = min(ch_lo + per, nchunk);

    using F4V = float __attribute__((ext_vector_type(4)));
    F4V wreg[WV];
    unsigned woff[WV];
    int wcl[WV];
    const bool wvec = (a.Co & 3) == 0;
#pragma unroll
    for (int q = 0; q < WV; ++q) {
        const int e4 = lane + q * 64;
        const int row = e4 / (COG / 4), c = (e4 - row * (COG / 4)) * 4;
        const int tap = row / kCvCC, cl = row - tap * kCvCC;
        const bool in = e4 < kCvRows * COG / 4 && co0 + c < a.Co;
        woff[q] = in ? (unsigned)(((long long)tap * a.Ci + cl) * a.Co + c) * 4u : 0u;
        wcl[q] = in ? cl : kCvCC;
    }
    auto load_w = [&](int ci0) {
        const int cc = min(kCvCC, a.Ci - ci0);
        const char *wbase = reinterpret_cast<const char *>(a.wt + (long long)ci0 * a.Co + co0);
#pragma unroll
        for (int q = 0; q < WV; ++q) {
            F4V v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (wcl[q] < cc) {
                if (wvec) {
                    v = *reinterpret_cast<const F4V *>(wbase + woff[q]);
                } else {
                    const float *pw = reinterpret_cast<const float *>(wbase + woff[q]);
                    const int e4 = lane + q * 64;
                    const int c = (e4 % (COG / 4)) * 4;
                    for (int j = 0; j < 4; ++j) v[j] = (co0 + c + j < a.Co) ? pw[j] : 0.0f;
                }
            }
            wreg[q] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int q = 0; q < WV; ++q) {
            const int e4 = lane + q * 64;
            if (e4 < kCvRows * COG / 4) reinterpret_cast<F4V *>(&s_w[wave][buf][0][0])[e4] = wreg[q];
        }
    };
    float xb[2][3][NP];
    auto load_x = [&](int set, int ci0) {
        const int cc = min(kCvCC, a.Ci - ci0);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int c2 = min(2 * p, cc - 1);
                const char *rowp = reinterpret_cast<const char *>(a.x + (long long)(ci0 + c2) * a.L);
                const bool pair = 2 * p + 1 < cc;
                xb[set][tap][p] = *reinterpret_cast<const float *>(rowp + (pair ? off_h[tap] : off_0[tap]));
            }
    };

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x16{0};

    if (ch_lo < ch_hi) {
        load_w(ch_lo * kCvCC);
        load_x(0, ch_lo * kCvCC);
        store_w(0);
    }
    auto chunk = [&](auto set_tag, const int ch) {
        constexpr int SET = decltype(set_tag)::value;
        const int ci0 = ch * kCvCC;
        const int cc = min(kCvCC, a.Ci - ci0);
        const bool more = ch + 1 < ch_hi;
        if (more) {
            load_w(ci0 + kCvCC);
            load_x(SET ^ 1, ci0 + kCvCC);
        }
#pragma unroll
        for (int ks = 0; ks < 3 * NP; ++ks) {
            const int tap = ks / NP, p = ks - tap * NP;
            float a_cur[CT];
#pragma unroll
            for (int t = 0; t < CT; ++t) a_cur[t] = s_w[wave][SET][tap * kCvCC + 2 * p + h][t * 32 + r];
            const bool ok = tap_ok[tap] && (2 * p + h < cc);
            const float bv = ok ? xb[SET][tap][p] : 0.0f;
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], bv, acc[t], 0, 0, 0);
        }
        if (more) store_w(SET ^ 1);       // the wave's own region: LDS operations of one wave stay in order
    };
    for (int ch = ch_lo; ch < ch_hi; ch += 2) {
        chunk(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < ch_hi) chunk(std::integral_constant<int, 1>{}, ch + 1);
    }
    // partial sums of waves 1..3 -> LDS, wave 0 adds them in wave order
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s_acc[wave - 1][t * 16 + reg][lane] = acc[t][reg];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < kCvWaves - 1; ++w)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[t][reg] += s_acc[w][t * 16 + reg][lane];
    conv_epilogue<CT>(a, acc, s_scale, s_shift, co0, seq, l, h, col_ok);
}

}  // namespace

namespace {

template <int KW, int STRIDE, bool FUSE1 = false>
void launch_conv1d(const ConvArgs &a, dim3 grid, int ct, hipStream_t s)
{
    if (ct == 1) conv1d_kernel<1, KW, STRIDE, FUSE1><<<grid, 64 * kCvWaves, 0, s>>>(a);
    else if (ct == 2) conv1d_kernel<2, KW, STRIDE, FUSE1><<<grid, 64 * kCvWaves, 0, s>>>(a);
    else conv1d_kernel<4, KW, STRIDE, FUSE1><<<grid, 64 * kCvWaves, 0, s>>>(a);
}

// l1 != nullptr: x is the single-channel input [S][L] and the Ci input channels of this (k = 3, stride 1) convolution
// are the first layer's outputs, computed in the kernel (ConvArgs::l1)
int conv1d_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift, int S, int Ci, int Co,
                    int L, int kernel, int stride, int pool, double negative_slope, float *out, pof_stream_t stream,
                    const float *l1 = nullptr, double slope1 = 0.0)
{
    if (!x || !wt || !scale || !shift || !out) return POF_E_BADARG;
    if (S < 0 || Ci < 1 || Co < 1 || L < 1) return POF_E_BADARG;
    if (l1 && (kernel != 3 || stride != 1 || Ci > 128 || !(slope1 >= 0.0 && slope1 <= 1.0))) return POF_E_SHAPE;
    if (!((kernel == 3 && (stride == 1 || stride == 2)) || (kernel == 1 && stride == 1))) return POF_E_SHAPE;
    const int Lc = stride == 1 ? L : (L + 1) / 2;         // (L + 2 pad - kernel) / stride + 1 with pad = kernel / 2
    if (pool && (stride != 1 || Lc < 2 || (Lc & 1))) return POF_E_SHAPE;   // pooled pairs sit on adjacent lanes: even L
    if (S == 0) return POF_OK;
    // 32-bit byte offsets per lane inside one launch: sequences go in chunks of < 2^30 input elements
    const long long per_seq = (long long)Ci * L;
    if (per_seq >= (1LL << 30)) return POF_E_SHAPE;
    const int s_max = (int)std::min<long long>(S, ((1LL << 30) - 1) / per_seq);
    const int Lout = pool ? Lc / 2 : Lc;
    hipStream_t s = pof_stream(stream);
    for (int s0 = 0; s0 < S; s0 += s_max) {
        ConvArgs a;
        a.S = std::min(s_max, S - s0);
        a.x = x + (long long)s0 * per_seq; a.wt = wt; a.scale = scale; a.shift = shift;
        a.out = out + (long long)s0 * Co * Lout;
        a.Ci = Ci; a.Co = Co; a.L = L; a.Lc = Lc; a.pool = pool ? 1 : 0; a.slope = (float)negative_slope;
        a.l1 = l1; a.slope1 = (float)slope1;
        if (l1) a.x = x + (long long)s0 * L;              // one input channel
        const long long ncol = (long long)a.S * Lc;
        const long long tiles = (ncol + 31) / 32;
        const long long gx = (tiles + kCvWaves - 1) / kCvWaves;
        if (gx > 0x7fffffffLL) return POF_E_SHAPE;
        // output channels per workgroup: 128 (64 for the narrow layers) when the launch gives every SIMD a wave
        // (one wave's four independent accumulators already keep its MFMA pipe busy).  A smaller launch
        // (streaming inference: a few dozen column tiles) leaves SIMDs idle and is bound by one wave's serial
        // K loop, so it takes narrower channel groups -- more and proportionally shorter workgroups, same
        // summation order, bit-identical results
        int ct = Co <= 32 ? 1 : (Co <= 64 ? 2 : 4);
        while (ct > 1 && gx * ((Co + 32 * ct - 1) / (32 * ct)) < kCvFillWorkgroups) ct >>= 1;
        // fewer than two resident rounds of 128-channel workgroups (4 per CU x 256 CUs): the last, partly filled round
        // costs a whole round -- 64-channel workgroups quantise finer (measured at S = 3600, one scan of a training
        // batch: 256 -> 256 L = 14 0.234 -> 0.218 ms, 512 -> 256 L = 14 0.425 -> 0.390, 256 -> 128 L = 28 0.214 -> 0.195),
        // at equal summation order
        if (ct == 4 && gx * ((Co + 127) / 128) < 8 * kCvFillWorkgroups) ct = 2;
        { static const int force = [] { const char *e = getenv("POF_CONV_CT"); return e ? atoi(e) : 0; }();
          if (force == 1 || force == 2 || force == 4) ct = (Co <= 32 && force > 1) ? 1 : (Co <= 64 && force > 2) ? 2 : force; }
        const long long wgs = gx * ((Co + 32 * ct - 1) / (32 * ct));
        const int nchunk = (Ci + kCvCC - 1) / kCvCC;
        if (!l1 && kernel == 3 && stride == 1 && wgs < 2 * kCvFillWorkgroups && Ci >= Co && nchunk >= 8 * kCvWaves &&
            tiles <= 0x7fffffffLL) {
            // still a launch that leaves most SIMDs with at most one wave, and a K loop long enough to pay for
            // the reduction (Ci >= 128; measured at one scan per call: 512->256 L=7 92 -> 58 us, 256->128 L=7
            // 43 -> 19 us, 256->256 L=14 52 -> 45 us; the widening layers Co = 2 Ci and Ci = 64 lose 10-50 %
            // to it and keep the one-wave-per-tile form): split K over the workgroup's waves
            // the partial sums go through LDS: 32 or 64 channels per workgroup.  With a very long K (Ci >= 512) the
            // 64-channel form wins even when the 32-channel one fills more SIMDs -- two MFMAs per A-operand read
            // instead of one over 96+ k-steps per wave (512 -> 256, L = 7, one scan: 57 -> 45 us)
            const int cts = (ct > 2 || (nchunk >= 32 * kCvWaves && Co >= 64)) ? 2 : ct;
            const dim3 grid((unsigned)tiles, (Co + 32 * cts - 1) / (32 * cts));
            if (cts == 1) conv3_splitk_kernel<1><<<grid, 64 * kCvWaves, 0, s>>>(a);
            else conv3_splitk_kernel<2><<<grid, 64 * kCvWaves, 0, s>>>(a);
        } else {
            const dim3 grid((unsigned)gx, (Co + 32 * ct - 1) / (32 * ct));
            if (l1) launch_conv1d<3, 1, true>(a, grid, ct, s);
            else if (kernel == 1) launch_conv1d<1, 1>(a, grid, ct, s);
            else if (stride == 2) launch_conv1d<3, 2>(a, grid, ct, s);
            else launch_conv1d<3, 1>(a, grid, ct, s);
        }
        POF_CHECK_LAUNCH();
    }
    return POF_OK;
}

}  // namespace

extern "C" int pof_conv3_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift,
                                  int S, int Ci, int Co, int L, int pool, double negative_slope, float *out,
                                  pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return conv1d_bn_lrelu(x, wt, scale, shift, S, Ci, Co, L, 3, 1, pool, negative_slope, out, stream);
}

extern "C" int pof_conv3_first_two(const float *x, const float *l1, double slope1, const float *wt, const float *scale,
                                   const float *shift, int S, int C1, int Co, int L, int pool, double negative_slope,
                                   float *out, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    if (!l1) return POF_E_BADARG;
    return conv1d_bn_lrelu(x, wt, scale, shift, S, C1, Co, L, 3, 1, pool, negative_slope, out, stream, l1, slope1);
}

extern "C" int pof_conv1d_bn_lrelu(const float *x, const float *wt, const float *scale, const float *shift,
                                   int S, int Ci, int Co, int L, int kernel_size, int stride, int pool,
                                   double negative_slope, float *out, pof_stream_t stream)
{
    POF_CLEAR_STALE_ERROR();
    return conv1d_bn_lrelu(x, wt, scale, shift, S, Ci, Co, L, kernel_size, stride, pool, negative_slope, out, stream);
}
