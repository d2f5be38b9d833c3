// Temporal attention of stride-1 windows of a series with the pair scores shared between the windows (SURVEY section 8f row 3,
// second half).  Reference: TemporalAttentionLayer.forward modules.py:166-193 (GATv2 scores :174-178, softmax :181, aggregation
// :191) under Predictor.get_score prediction.py:51-63, which scores every stride-1 window of the series.
//
// Window w is rows [w, w + K) of the series.  The node vector of local row i is the convolution output of series row w + i,
// and for pad <= i < K - pad that output does not depend on w (the window's zero padding, modules.py:14,20, reaches only its
// first / last `pad` rows).  So the raw score of a pair of such rows,
//     e(r, t) = c_r + d_t + sum_{k < P8} |L'_rk + R'_tk| - sum_{P8 <= k < PT} |L'_rk + R'_tk|          (the k_gat algebra)
// is a function of the two SERIES rows: it is computed once (k_tband_scores: a band |t - r| <= K - 1 - 2 pad around the
// diagonal) instead of once per window that contains both -- up to K - 2 pad times.  What stays per window
// (k_tband_att, one workgroup per window): the 2 pad (K - pad) + ... pairs that involve one of its own edge rows, the bias
// (K, K) that is indexed by the LOCAL positions (modules.py:179), the softmax rows and the aggregation att V.
// Everything is fp32 (projections: k_rowgemm on the fp32 MFMA; pair sums: VALU; aggregation: v_mfma_f32_16x16x4_f32).
#include <cstdlib>
#include "mtadgat_device.h"

namespace mtadgat {

namespace {

constexpr int TB_ROWS = 32;       // query rows per workgroup of the band kernel
constexpr int TB_EWMAX = 6;       // edge rows per window (kernel_size <= 7)
constexpr int TB_APITCH = 68;     // softmax rows restaged for the aggregation, 64 keys at a time

// sum over 4 columns of +-|l + r|
template <bool NEG>
__device__ __forceinline__ float abs4(float acc, const f32x4 l, const f32x4 r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = l[e] + r[e];
        acc = NEG ? acc - __builtin_fabsf(t) : acc + __builtin_fabsf(t);
    }
    return acc;
}

// ---- band of raw scores: BI[r][t - r + HB] for |t - r| <= HB.  A workgroup owns TB_ROWS query rows (their L' in LDS, read
// as wave-uniform 16-byte words) and one key per thread (its R' row straight from the row-major projections).
__global__ __launch_bounds__(256) void k_tband_scores(const TBandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * TB_ROWS;
    const int PT = a.ord ? a.ord[1] : a.PT, P8 = a.ord ? a.ord[0] : a.P8, lp = PT + 4;                  // LDS row: L' (PT) | c | pad
    for (int u = tid; u < TB_ROWS * (lp >> 2); u += 256) {
        const int i = u / (lp >> 2), c4 = (u - i * (lp >> 2)) * 4;
        const long r = r0 + i < a.Lrows ? r0 + i : a.Lrows - 1;
        *reinterpret_cast<f32x4*>(smem + i * lp + c4) = *reinterpret_cast<const f32x4*>(a.PJ + r * a.ldp + c4);     // (columns PT.. of the row: c and padding)
    }
    __syncthreads();
    const int nkeys = TB_ROWS + 2 * a.HB;
    for (int kk = tid; kk < nkeys; kk += 256) {
        const long t = r0 - a.HB + kk;
        const long tc = t < 0 ? 0 : (t >= a.Lrows ? a.Lrows - 1 : t);
        const float* __restrict__ rp = a.PJ + tc * a.ldp + a.ldl;
        float acc[TB_ROWS];
#pragma unroll
        for (int i = 0; i < TB_ROWS; ++i) acc[i] = 0.f;
        int k = 0;
        for (; k < P8; k += 4) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + k);
#pragma unroll
            for (int i = 0; i < TB_ROWS; ++i) acc[i] = abs4<false>(acc[i], *reinterpret_cast<const f32x4*>(smem + i * lp + k), rv);
        }
        for (; k < PT; k += 4) {
            const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + k);
#pragma unroll
            for (int i = 0; i < TB_ROWS; ++i) acc[i] = abs4<true>(acc[i], *reinterpret_cast<const f32x4*>(smem + i * lp + k), rv);
        }
        const float dv = rp[PT];
        if (t >= 0 && t < a.Lrows) {
#pragma unroll
            for (int i = 0; i < TB_ROWS; ++i) {
                const long r = r0 + i;
                const long dl = t - r + a.HB;
                if (r < a.Lrows && dl >= 0 && dl <= 2 * a.HB) a.BI[r * a.bp + dl] = acc[i] + smem[i * lp + PT] + dv;
            }
        }
    }
}

// ---- the pairs that involve an edge row of a window.  A wave per window; per instruction it takes one row p of the window
// whole: lanes 0-31 its key side R'_p (16 bytes each) against the EW edge queries -> EQ[w][q][p], lanes 32-63 its query side
// L'_p against the EW edge keys -> EK[w][p][q]: coalesced reads of the row-major projections, the edge rows' words stay in
// registers for the whole window, sums over the 32 lanes of a half by DPP + one cross-row exchange.
// (Tried instead: a workgroup per window, 32 embedding columns at a time transposed through LDS so that a thread owns a row and no
// reduction is needed -- 1.48 ms against the 1.33 ms of this version: four stage / barrier rounds per window cost more than the 36
// reduction instructions per row pair they save.)
__device__ __forceinline__ float half_sum(float v) {
    v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);    // row_half_mirror
    v += dpp_move<0x140>(v);    // row_mirror: every lane of a 16-lane row holds the row's sum
    return v + __shfl_xor(v, 16);
}
__global__ __launch_bounds__(256) void k_tband_edges(const TBandArgs a) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.n) return;
    const int K = a.K, PT = a.ord ? a.ord[1] : a.PT, P8 = a.ord ? a.ord[0] : a.P8, pad = a.pad, EW = 2 * a.pad;
    const int kq = ((K + 3) & ~3) + 4;
    const int half = lane >> 5, l = lane & 31;
    const int nu = PT >> 2;
    const int lc = l < nu ? l : nu - 1;
    const float sgn = (l >= nu) ? 0.f : (4 * l >= P8 ? -1.f : 1.f);
    // the edge rows' other side: half 0 (keys) needs the edge queries' L', half 1 the edge keys' R'
    const int eside = half ? a.ldl : 0, oside = half ? 0 : a.ldl;
    f32x4 e[TB_EWMAX];
    float ecd[TB_EWMAX];
#pragma unroll
    for (int q = 0; q < TB_EWMAX; ++q) {
        const int qc = q < EW ? q : EW - 1;
        const float* __restrict__ er = (qc < pad ? a.PJT : a.PJB) + (w * EW + qc) * (long)a.ldp + eside;
        e[q] = *reinterpret_cast<const f32x4*>(er + 4 * lc);
        ecd[q] = er[PT];
    }
    auto row_ptr = [&](int p) {
        return (p < pad ? a.PJT + (w * EW + p) * (long)a.ldp
                        : (p >= K - pad ? a.PJB + (w * EW + (p - (K - EW))) * (long)a.ldp : a.PJ + (w + p) * (long)a.ldp)) + oside;
    };
    constexpr int UN = 4;
    for (int p0 = 0; p0 < K; p0 += UN) {
        f32x4 ov[UN];
        float cd[UN];
#pragma unroll
        for (int x = 0; x < UN; ++x) {
            const float* __restrict__ op = row_ptr(p0 + x < K ? p0 + x : K - 1);
            ov[x] = *reinterpret_cast<const f32x4*>(op + 4 * lc);
            cd[x] = op[PT];
        }
#pragma unroll
        for (int x = 0; x < UN; ++x) {
            const int p = p0 + x;
            float mine = 0.f;
#pragma unroll
            for (int q = 0; q < TB_EWMAX; ++q)
                if (q < EW) {
                    const float t = half_sum(abs4<false>(0.f, e[q], ov[x]) * sgn) + cd[x] + ecd[q];
                    mine = l == q ? t : mine;
                }
            if (p < K && l < EW) {
                if (half) a.EK[(w * K + p) * (EW + 2) + l] = mine;
                else a.EQ[(w * EW + l) * kq + p] = mine;
            }
        }
    }
}

// ---- one workgroup per window, a wave per 16 query rows: raw scores (band / edge arrays) + bias, softmax, aggregation.
// LDS: Vs [16 NW][vld]   the window's node rows (convolution outputs; rows >= K and columns >= D zero)
//      att[NW][16][68]   a wave's softmax rows, 64 keys at a time (B operand of the aggregation)
// A wave asks for everything it reads -- its 16 x 2 raw scores, their bias words, its share of the node rows -- before it uses
// the first word: one memory round trip per window.
template <int DTMAX>
__global__ __launch_bounds__(512, 4) void k_tband_att(const TBandArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = nthr >> 6;
    const long w = blockIdx.x;
    const int K = a.K, D = a.D, pad = a.pad, EW = 2 * a.pad;
    const int vld = ((D + 15) & ~15) + 4, kq = ((K + 3) & ~3) + 4;
    float* __restrict__ Vs = smem;
    float* __restrict__ att = Vs + 16 * NW * vld;

    // ---- requests: raw scores of this wave's 16 rows (keys lane and lane + 64) and their bias
    const int i0 = 16 * wave;
    const int j0 = lane, j1 = lane + 64;
    const int j0c = j0 < K ? j0 : K - 1, j1c = j1 < K ? j1 : K - 1;
    const int qj0 = j0c < pad ? j0c : (j0c >= K - pad ? j0c - (K - EW) : -1);
    const int qj1 = j1c < pad ? j1c : (j1c >= K - pad ? j1c - (K - EW) : -1);
    float p0[16], p1[16], b0[16], b1[16];
    // (every base below is wave-uniform, the lane adds a 32-bit offset; rows of the band are read over their whole key range --
    // the words at this window's edge keys are other rows' band entries or row padding, replaced below)
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
        const int i = i0 + ii < K ? i0 + ii : K - 1;
        const int qi = i < pad ? i : (i >= K - pad ? i - (K - EW) : -1);
        const float* __restrict__ src = qi >= 0 ? a.EQ + (w * EW + qi) * kq : a.BI + (w + i) * a.bp + (a.HB - i);
        p0[ii] = (a.dbg & 4) ? (float)ii : src[j0c];
        p1[ii] = (a.dbg & 4) ? (float)ii : src[j1c];
        b0[ii] = (a.bias && !(a.dbg & 8)) ? a.bias[(long)i * K + j0c] : 0.f;
        b1[ii] = (a.bias && !(a.dbg & 8)) ? a.bias[(long)i * K + j1c] : 0.f;
    }
    // the wave's 16 rows of EK (scores against the edge keys): contiguous, through the wave's LDS slice to the lanes of those keys
    float* __restrict__ at = att + wave * (16 * TB_APITCH);
    {
        const int ekp = EW + 2, cnt = 16 * ekp;                     // <= 128 floats
        const long ekmax = (long)K * ekp - 1;
        const float* __restrict__ ekw = a.EK + w * K * ekp;
        const long o0 = (long)i0 * ekp + lane, o1 = o0 + 64;
        const float e0 = ekw[o0 < ekmax ? o0 : ekmax], e1 = ekw[o1 < ekmax ? o1 : ekmax];
        at[lane] = e0;
        if (lane + 64 < cnt) at[lane + 64] = e1;
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) {
            const int i = i0 + ii < K ? i0 + ii : K - 1;
            const bool interior = i >= pad && i < K - pad;          // (wave-uniform)
            const float k0 = at[ii * ekp + (qj0 >= 0 ? qj0 : 0)], k1 = at[ii * ekp + (qj1 >= 0 ? qj1 : 0)];
            p0[ii] = (interior && qj0 >= 0) ? k0 : p0[ii];
            p1[ii] = (interior && qj1 >= 0) ? k1 : p1[ii];
        }
    }
    // ---- stage the node rows
    if (a.dbg & 64) return;
    {
        const float* __restrict__ Vw = a.V + w * a.sv_w;
        const int v4 = vld >> 2, d4 = (D + 3) >> 2;
        for (int base = 0; base < ((a.dbg & 16) ? 0 : K * d4); base += 4 * nthr) {
            f32x4 v[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int u = base + tid + n * nthr;
                const int uc = u < K * d4 ? u : K * d4 - 1;
                const int r = uc / d4, c4 = (uc - r * d4) * 4;
                v[n] = *reinterpret_cast<const f32x4*>(Vw + (long)r * a.ldv + c4);        // (row pitch and base: multiples of 16 bytes; the words past column D belong to the row)
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int u = base + tid + n * nthr;
                if (u < K * d4) {
                    const int r = u / d4, c4 = (u - r * d4) * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[n][e] = c4 + e < D ? v[n][e] : 0.f;
                    *reinterpret_cast<f32x4*>(Vs + r * vld + c4) = v[n];
                }
            }
        }
        // zero padding: columns [4 d4, vld) of the rows < K, whole rows >= K
        for (int u = tid; u < 16 * NW * v4; u += nthr) {
            const int r = u / v4, c4 = (u - r * v4);
            if (r >= K || c4 >= d4) *reinterpret_cast<f32x4*>(Vs + r * vld + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    // ---- softmax
    if (!(a.dbg & 1))
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) {
        const float v0 = j0 < K ? p0[ii] + b0[ii] : -INFINITY;
        const float v1 = j1 < K ? p1[ii] + b1[ii] : -INFINITY;
        const float m = wave_max(fmaxf(v0, v1));
        const float e0 = j0 < K ? soft_exp(v0 - m) : 0.f;
        const float e1 = j1 < K ? soft_exp(v1 - m) : 0.f;
        const float inv = soft_rcp(wave_sum(e0 + e1));
        const bool rv = i0 + ii < K;
        p0[ii] = rv ? e0 * inv : 0.f;
        p1[ii] = rv ? e1 * inv : 0.f;
    }

    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j): out^T = V^T att^T on v_mfma_f32_16x16x4_f32, 64 keys of the softmax rows
    // at a time through this wave's LDS slice
    const int nr = lane & 15, kb = lane >> 4;
    const int DT = (D + 15) >> 4;
    f32x4 o[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (64 * half < K) {
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) at[ii * TB_APITCH + lane] = half ? p1[ii] : p0[ii];
            for (int g = 0; g < ((a.dbg & 2) ? 1 : 4); ++g) {
                const int kv = 64 * half + 16 * g;
                if (kv < K) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(at + nr * TB_APITCH + 16 * g + 4 * kb);
                    const float* __restrict__ vk = Vs + (kv + 4 * kb) * vld;
#pragma unroll
                    for (int dt = 0; dt < DTMAX; ++dt)
                        if (dt < DT) {
                            float av[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) av[t] = vk[t * vld + 16 * dt + nr];
#pragma unroll
                            for (int t = 0; t < 4; ++t) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bq[t], o[dt], 0, 0, 0);
                        }
                }
            }
        }
    }
    // ---- sigmoid; the 16 x D output tile goes through the wave's LDS slice so that a row leaves as one contiguous run
    // (lane-per-element stores of the MFMA layout are 4-byte writes to 64 different places per instruction)
    {
#pragma unroll
        for (int dt = 0; dt < DTMAX; ++dt)
            if (dt < DT) {
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = gate_sigmoid(o[dt][r]);
                *reinterpret_cast<f32x4*>(at + nr * TB_APITCH + 16 * dt + 4 * kb) = y;
            }
        float* __restrict__ ow = a.out + w * a.so_w;
        if (a.dbg & 32) { if (at[lane] == 12345.f) ow[0] = 1.f; return; }
        if (a.so_d == 1) {
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + r;
                if (row < K && lane < D) ow[(long)row * a.so_i + lane] = at[r * TB_APITCH + lane];
            }
        } else {
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + r;
                if (row < K && lane < D) ow[(long)row * a.so_i + (long)lane * a.so_d] = at[r * TB_APITCH + lane];
            }
        }
    }
}

size_t tband_att_lds(int K, int D) {
    const int NW = (K + 15) / 16, vld = ((D + 15) & ~15) + 4;
    return ((size_t)16 * NW * vld + (size_t)NW * 16 * TB_APITCH) * sizeof(float);
}

}  // namespace

// node counts the window kernel is laid out for (two keys per lane, 16 query rows per wave) and edge rows it holds
bool tband_applies(int K, int D, int PT, int pad, int ldl, int ldp) {
    if (K < 16 || K > 128 || D < 1 || D > 64 || PT < 8 || PT > 128 || (PT & 7) != 0) return false;      // (PT <= 128: a row side in one 32-lane read)
    if (pad < 1 || 2 * pad > TB_EWMAX || K < 4 * pad + 2) return false;
    if ((ldl & 3) != 0 || (ldp & 3) != 0 || ldl < PT + 4) return false;       // 16-byte reads of [L' | c | pad] and of the key side
    return tband_att_lds(K, D) <= 80 * 1024;
}

int launch_tband_scores(const TBandArgs& a, hipStream_t s) {
    if (a.Lrows <= 0) return 0;
    const size_t lds = (size_t)TB_ROWS * ((a.PTcap > a.PT ? a.PTcap : a.PT) + 4) * sizeof(float);
    hipLaunchKernelGGL(k_tband_scores, dim3((unsigned)((a.Lrows + TB_ROWS - 1) / TB_ROWS)), dim3(256), lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_tband_edges(const TBandArgs& a, hipStream_t s) {
    if (a.n <= 0) return 0;
    hipLaunchKernelGGL(k_tband_edges, dim3((unsigned)((a.n + 3) / 4)), dim3(256), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_tband_att(const TBandArgs& a, hipStream_t s) {
    if (a.n <= 0) return 0;
    const int NW = (a.K + 15) / 16;
    const size_t lds = tband_att_lds(a.K, a.D);
    const int DT = (a.D + 15) >> 4;
    if (const char* e_ = getenv("MTADGAT_TB_DBG")) const_cast<TBandArgs&>(a).dbg = atoi(e_);
#define TB_LAUNCH(N)                                                                                                    \
    do {                                                                                                                \
        if (lds > 64 * 1024) {                                                                                          \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tband_att<N>),                        \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
            if (e_ != hipSuccess) return (int)e_;                                                                       \
        }                                                                                                               \
        hipLaunchKernelGGL((k_tband_att<N>), dim3((unsigned)a.n), dim3(64 * NW), lds, s, a);                            \
    } while (0)
    if (DT <= 2) TB_LAUNCH(2);
    else if (DT <= 4) TB_LAUNCH(4);
    else return -2;                 // (tband_applies: D <= 64)
#undef TB_LAUNCH
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
