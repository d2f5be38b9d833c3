// k_gat2: fused GATv2 layer with a COLUMN-SLICED pair grid (inference, two-fp16-piece arithmetic) + its weight pack
//
// Reference: FeatureAttentionLayer.forward / TemporalAttentionLayer.forward (modules.py:65-95, :166-193) in the
// re-associated algebra of DESIGN.md section 3:  e_ij = c_i + d_j + sum_k s_k |L'_ik + R'_jk| + bias_ij.
//
// k_gat (mtadgat_gat.hip) splits the K x K pair grid of a window by query rows over the waves of a workgroup and lets
// every wave walk all embedding columns: row blocks that do not divide K (100 rows = 7 x 16 - 12), key padding,
// one idle wave and two barriers per 32 columns cost it half of the vector ALU.  This kernel turns the split around:
//
//   * every wave owns ALL K x K pairs of the window -- lane (li, lj) of an 8 x 8 lane grid holds IBL x IBL accumulators
//     (IBL = ceil(K / 8): 13 x 13 for 100 nodes, 7 x 7 for 55) -- and a SLICE of the embedding columns
//     (TC = E + 1 columns dealt out evenly: 14 of 111 per wave for the temporal layer);
//   * a wave projects its own columns (v_mfma_f32_16x16x32_f16 on two fp16 pieces per operand, node vectors split ONCE
//     when the window is staged) into its own LDS slice and then runs its pair loop with no barrier and no other
//     wave involved: per column 2 x IBL / 4 wide LDS reads feed 2 IBL^2 VALU instructions (profiles/ubench_gat2.hip:
//     2.7-2.9 cycles per instruction and SIMD at two to four waves per SIMD);
//   * the partial sums of the waves meet once, in a reduce-scatter through LDS in a fixed order (deterministic), after
//     which wave w owns the complete scores of the query rows {8 ii + li : ii = w, w + NWV, ...}: softmax in the 8-lane
//     DPP groups, attention rows as fp16 pieces into LDS, aggregation att V on the 16-bit matrix pipe, sigmoid.
//
// Node maps: query row of (li, ii) = 8 ii + li (cyclic: the 8 rows an owner wave ends up with per ii are consecutive
// nodes), key of (lj, jj) = IBL lj + jj (contiguous: bias rows and attention rows are read / written in runs).
// LDS slot of (g, idx) inside a projected column: g * 4A + idx for idx < 4A (A = IBL / 4: 16-byte reads), the remaining
// IBL % 4 values in an extra region (slot MAIN + g * EB + idx - 4A).
//
// The kernel serves the default fp32 arithmetic of large batches when the producing convolution recorded node values
// below 2^15 (GatArgs::vmax); otherwise it returns at once and k_gat (launched behind it with skip_h) does the work.
#include "mtadgat_device.h"

namespace mtadgat {

namespace {

typedef const __attribute__((address_space(3))) float* lds_cptr;
typedef const __attribute__((address_space(3))) f32x4* lds_c4;
typedef const __attribute__((address_space(3))) f32x2* lds_c2;
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N>
struct G2Slots {                                  // N = 4 A + B values per lane group
    static constexpr int A = N / 4, B = N % 4, EB = B == 3 ? 4 : B;
    static constexpr int MAIN = 8 * 4 * A, TOTAL = MAIN + 8 * EB;
};

// the N operands of a lane for the column at float offset `off`
template <int N>
__device__ __forceinline__ void g2_load_col(float (&v)[N], lds_cptr pm, lds_cptr pe, int off) {
    constexpr int A = G2Slots<N>::A, B = G2Slots<N>::B;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const f32x4 t = *(lds_c4)(pm + off + 4 * a);
        v[4 * a] = t[0]; v[4 * a + 1] = t[1]; v[4 * a + 2] = t[2]; v[4 * a + 3] = t[3];
    }
    if constexpr (B == 1) v[4 * A] = pe[off];
    if constexpr (B == 2) { const f32x2 t = *(lds_c2)(pe + off); v[4 * A] = t[0]; v[4 * A + 1] = t[1]; }
    if constexpr (B == 3) { const f32x4 t = *(lds_c4)(pe + off); v[4 * A] = t[0]; v[4 * A + 1] = t[1]; v[4 * A + 2] = t[2]; }
}

// one embedding column: acc_ij += s |l_i + r_j|  (s = +-1 in a VGPR; inline asm for the reasons given in mtadgat_gat.hip)
template <int N>
__device__ __forceinline__ void g2_pair_step(float (&acc)[N][N], const float (&l)[N], const float (&r)[N], float s) {
#pragma unroll
    for (int ii = 0; ii < N; ++ii) {
        float t[N];
#pragma unroll
        for (int jj = 0; jj < N; ++jj) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(t[jj]) : "v"(l[ii]), "v"(r[jj]));
#pragma unroll
        for (int jj = 0; jj < N; ++jj) asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(acc[ii][jj]) : "v"(t[jj]), "v"(s));
    }
}

__device__ __forceinline__ float g2_max8(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    return v;
}
__device__ __forceinline__ float g2_sum8(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

__device__ __forceinline__ f32x4 g2_mfma(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// w x ~= wh xl + wl xh + wh xh (two fp16 pieces per operand, mtadgat_device.h)
__device__ __forceinline__ f32x4 g2_mfma3(const f16x8 ah, const f16x8 al, const f16x8 bh, const f16x8 bl, f32x4 c) {
    c = g2_mfma(al, bh, c);
    c = g2_mfma(ah, bl, c);
    c = g2_mfma(ah, bh, c);
    return c;
}

constexpr int G2_IIP = 3;      // query-row indices per reduce-scatter pass

}  // namespace

template <int IBL, int NT>
__global__ __launch_bounds__(NT, NT / 256) void k_gat2(const Gat2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    using SL = G2Slots<IBL>;
    constexpr int NWV = NT / 64;
    constexpr int A4 = SL::A, EB = SL::EB, MAIN = SL::MAIN, TOTAL = SL::TOTAL;
    constexpr int CS = TOTAL + 4;                 // floats between the columns of a slice
    constexpr int NTL = (TOTAL + 15) / 16;        // 16-slot tiles of the projection
    constexpr int JP = IBL <= 8 ? 8 : 16;         // key positions per lane in the attention rows (pad positions hold zeros)
    constexpr int JQ = (IBL + 3) / 4;
    constexpr int NP = (IBL + G2_IIP - 1) / G2_IIP;
    constexpr int NOWN = (IBL + NWV - 1) / NWV;
    static_assert(IBL >= 4 && IBL <= 13, "accumulator block");

    if (a.vmax && __uint_as_float(*a.vmax) >= 32768.f) return;         // node values beyond the fp16 pieces: k_gat runs instead
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long win = blockIdx.x;
    const int K = a.K, D = a.D, pv = a.pv, pa = a.pa;
    const int li = lane >> 3, lj = lane & 7;
    const int n16 = lane & 15, kb = lane >> 4;    // MFMA roles
    _Float16* __restrict__ Vh = reinterpret_cast<_Float16*>(smem8 + a.off_v);
    _Float16* __restrict__ Vl = Vh + K * pv;
    float* __restrict__ cdL = reinterpret_cast<float*>(smem8 + a.off_cd);
    float* __restrict__ cdR = cdL + TOTAL;

    // this wave's embedding columns [c0, c1) of the TC = E + 1 (the last one carries the rank-1 terms c_i / d_j)
    const int TC = a.E + 1;
    const int c0 = (wave * TC) / NWV, c1 = ((wave + 1) * TC) / NWV;
    const int ncol = c1 - c0;
    const int nabs = (c1 < a.E ? c1 : a.E) - c0;

    // ---- stage the window: node vectors as two fp16 pieces, Vh/Vl[node][feature], feature D = 1 (the projection bias is
    // weight row D), features (D, KP) = 0.  vt == 0: source rows are the nodes; vt == 1: source columns are the nodes.
    // Unconditional clamped loads, all issued before the first LDS store.
    {
        const int srows = a.vt ? D : K, scols = a.vt ? K : D;
        const int UR = (scols + 3) >> 2;
        const int total = srows * UR;
        const float rinv = 1.0f / (float)UR;
        const float* __restrict__ vsrc = a.V + win * (long)srows * a.ldv;
        const int c4last = ((scols - 1) >> 2) << 2;
        constexpr int MAXU = 4;
        for (int base = 0; base < total; base += MAXU * NT) {
            f32x4 v[MAXU];
            int rr[MAXU], cc[MAXU];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = base + tid + n * NT;
                const int row = (int)(((float)u + 0.5f) * rinv), c4 = (u - row * UR) * 4;
                rr[n] = u < total ? row : -1;
                cc[n] = c4;
                const int rc = row < srows ? row : srows - 1, cl = c4 < scols ? c4 : c4last;
                v[n] = *reinterpret_cast<const f32x4*>(vsrc + (long)rc * a.ldv + cl);
            }
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int row = rr[n], c4 = cc[n];
                if (row >= 0) {
                    f32x4 t = v[n];
                    if (!a.vt) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = c4 + e < D ? t[e] : (c4 + e == D ? 1.f : 0.f);
                        unsigned h0, l0, h1, l1;
                        split_pair_h(t[0], t[1], h0, l0);
                        split_pair_h(t[2], t[3], h1, l1);
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2*>(Vh + row * pv + c4) = u32x2{h0, h1};
                        *reinterpret_cast<u32x2*>(Vl + row * pv + c4) = u32x2{l0, l1};
                    } else {
                        unsigned h0, l0, h1, l1;
                        split_pair_h(t[0], t[1], h0, l0);
                        split_pair_h(t[2], t[3], h1, l1);
                        const unsigned short hs[4] = {(unsigned short)h0, (unsigned short)(h0 >> 16), (unsigned short)h1, (unsigned short)(h1 >> 16)};
                        const unsigned short ls[4] = {(unsigned short)l0, (unsigned short)(l0 >> 16), (unsigned short)l1, (unsigned short)(l1 >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c4 + e < K) {
                                reinterpret_cast<unsigned short*>(Vh)[(c4 + e) * pv + row] = hs[e];
                                reinterpret_cast<unsigned short*>(Vl)[(c4 + e) * pv + row] = ls[e];
                            }
                    }
                }
            }
        }
        // ones column and zero padding (vt == 0: the columns the units above did not reach)
        const int f0 = a.vt ? D : 4 * UR, nf = a.KP - f0;
        if (nf > 0) {
            const float ninv = 1.0f / (float)nf;
            for (int u = tid; u < K * nf; u += NT) {
                const int node = (int)(((float)u + 0.5f) * ninv), f = f0 + (u - node * nf);
                reinterpret_cast<unsigned short*>(Vh)[node * pv + f] = f == D ? (unsigned short)0x3C00 : (unsigned short)0;
                reinterpret_cast<unsigned short*>(Vl)[node * pv + f] = 0;
            }
        }
    }
    __syncthreads();
    if (a.dbg_stop == 1) return;

    float* __restrict__ Lw = reinterpret_cast<float*>(smem8 + a.off_lr) + wave * a.lr_wave_floats;
    float* __restrict__ Rw = Lw + a.CW * CS;

    // ---- projection of this wave's columns: D[slot][col] = sum_f V[node(slot)][f] W[col][f] on v_mfma_f32_16x16x32_f16,
    // A = node vectors (16 slots of a tile; lane (m, kb): 8 features), B = weights (lane (n, kb): column c0 + n).
    // Result lane (n, mb): slots 16 T + 4 mb .. + 3 of column n -> one 16-byte LDS store.  The fp16 weights carry the
    // layer's power of two S; it is taken out of the scores at the end.
    if (ncol > 0) {
        int offL[NTL], offR[NTL];
#pragma unroll
        for (int T = 0; T < NTL; ++T) {
            int s = 16 * T + n16;
            s = s < TOTAL ? s : TOTAL - 1;
            int g, idx;
            if (s < MAIN) { g = s / (4 * A4 > 0 ? 4 * A4 : 1); idx = s - g * 4 * A4; }
            else { const int x = s - MAIN; g = x / (EB > 0 ? EB : 1); idx = 4 * A4 + (x - g * EB); }
            idx = idx < IBL ? idx : IBL - 1;
            int nl = idx * 8 + g, nr = g * IBL + idx;
            nl = nl < K ? nl : K - 1;
            nr = nr < K ? nr : K - 1;
            offL[T] = nl * pv + 8 * kb;
            offR[T] = nr * pv + 8 * kb;
        }
        f32x4 pacc[NTL][2];
#pragma unroll
        for (int T = 0; T < NTL; ++T) { pacc[T][0] = f32x4{0.f, 0.f, 0.f, 0.f}; pacc[T][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const _Float16* __restrict__ Wg = reinterpret_cast<const _Float16*>(a.W);
        const long sstride = (long)a.TCP * a.KP, pstride = 2 * sstride;
        const _Float16* __restrict__ wl = Wg + (long)(c0 + n16) * a.KP + 8 * kb;
#pragma unroll 1
        for (int c = 0; c < a.KC; ++c) {
            f16x8 bh[2], bl[2];
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                bh[sd] = *reinterpret_cast<const f16x8*>(wl + sd * sstride + 32 * c);
                bl[sd] = *reinterpret_cast<const f16x8*>(wl + pstride + sd * sstride + 32 * c);
            }
#pragma unroll
            for (int T = 0; T < NTL; ++T) {
                const f16x8 ahl = *reinterpret_cast<const f16x8*>(Vh + offL[T] + 32 * c), all_ = *reinterpret_cast<const f16x8*>(Vl + offL[T] + 32 * c);
                const f16x8 ahr = *reinterpret_cast<const f16x8*>(Vh + offR[T] + 32 * c), alr = *reinterpret_cast<const f16x8*>(Vl + offR[T] + 32 * c);
                pacc[T][0] = g2_mfma3(ahl, all_, bh[0], bl[0], pacc[T][0]);
                pacc[T][1] = g2_mfma3(ahr, alr, bh[1], bl[1], pacc[T][1]);
            }
        }
        if (n16 < ncol) {
#pragma unroll
            for (int T = 0; T < NTL; ++T)
                if (16 * T + 4 * kb < TOTAL) {
                    *reinterpret_cast<f32x4*>(Lw + n16 * CS + 16 * T + 4 * kb) = pacc[T][0];
                    *reinterpret_cast<f32x4*>(Rw + n16 * CS + 16 * T + 4 * kb) = pacc[T][1];
                }
        }
    }

    if (a.dbg_stop == 2) return;
    // ---- pair grid over this wave's columns (no barrier: the slice is private to the wave, LDS operations of a wave
    // execute in order).  Two operand sets alternate: the reads of the next column are in flight during a column.
    float acc[IBL][IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < IBL; ++jj) acc[ii][jj] = 0.f;
    {
        const lds_cptr lm = (lds_cptr)Lw + li * 4 * A4, le = (lds_cptr)Lw + MAIN + li * EB;
        const lds_cptr rm = (lds_cptr)Rw + lj * 4 * A4, re = (lds_cptr)Rw + MAIN + lj * EB;
        float lA[IBL], rA[IBL], lB[IBL], rB[IBL];
        if (nabs > 0) {
            g2_load_col<IBL>(lA, lm, le, 0);
            g2_load_col<IBL>(rA, rm, re, 0);
        }
        int k = 0, off = 0;
        const int npos_w = a.npos - c0;                 // columns [0, npos_w) of the slice have a' >= 0
#pragma unroll 1
        for (; k + 1 < nabs; k += 2) {
            float s0 = k < npos_w ? 1.f : -1.f, s1 = k + 1 < npos_w ? 1.f : -1.f;
            asm volatile("" : "+v"(s0), "+v"(s1));
            g2_load_col<IBL>(lB, lm, le, off + CS);
            g2_load_col<IBL>(rB, rm, re, off + CS);
            __builtin_amdgcn_sched_barrier(0);
            g2_pair_step<IBL>(acc, lA, rA, s0);
            __builtin_amdgcn_sched_barrier(0);
            g2_load_col<IBL>(lA, lm, le, off + 2 * CS);          // (past the last column of the slice: never consumed)
            g2_load_col<IBL>(rA, rm, re, off + 2 * CS);
            __builtin_amdgcn_sched_barrier(0);
            g2_pair_step<IBL>(acc, lB, rB, s1);
            __builtin_amdgcn_sched_barrier(0);
            off += 2 * CS;
        }
        if (k < nabs) {
            float s0 = k < npos_w ? 1.f : -1.f;
            asm volatile("" : "+v"(s0));
            g2_pair_step<IBL>(acc, lA, rA, s0);
        }
    }
    if (c1 == TC && ncol > 0) {                         // the wave that projected the rank-1 column keeps it for the softmax
        const int kc = a.E - c0;
        for (int s = lane; s < TOTAL; s += 64) {
            cdL[s] = Lw[kc * CS + s];
            cdR[s] = Rw[kc * CS + s];
        }
    }
    __syncthreads();                                    // every slice is consumed: the region becomes the exchange scratch
    if (a.dbg_stop == 3) { if (acc[0][0] == 12345.f) a.out[0] = acc[1][1]; return; }

    // ---- reduce-scatter of the partial sums: pass p moves the rows ii in [3p, 3p + 3); wave ii % NWV adds the NWV
    // partials in wave order
    float own[NOWN][IBL];
#pragma unroll
    for (int o = 0; o < NOWN; ++o)
#pragma unroll
        for (int jj = 0; jj < IBL; ++jj) own[o][jj] = 0.f;
    {
        f32x4* __restrict__ scr = reinterpret_cast<f32x4*>(smem8 + a.off_lr);
        static_for<0, NP>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            static_for<0, G2_IIP>([&](auto ic) {
                constexpr int iloc = decltype(ic)::value, ii = p * G2_IIP + iloc;
                if constexpr (ii < IBL) {
#pragma unroll
                    for (int q = 0; q < JQ; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 4 * q + e < IBL ? acc[ii][4 * q + e < IBL ? 4 * q + e : 0] : 0.f;
                        scr[((wave * G2_IIP + iloc) * JQ + q) * 64 + lane] = v;
                    }
                }
            });
            __syncthreads();
            static_for<0, G2_IIP>([&](auto ic) {
                constexpr int iloc = decltype(ic)::value, ii = p * G2_IIP + iloc;
                if constexpr (ii < IBL) {
                    constexpr int o = ii / NWV;
                    if (wave == ii % NWV) {
#pragma unroll 1
                        for (int src = 0; src < NWV; ++src) {
#pragma unroll
                            for (int q = 0; q < JQ; ++q) {
                                const f32x4 v = scr[((src * G2_IIP + iloc) * JQ + q) * 64 + lane];
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (4 * q + e < IBL) own[o][4 * q + e] += v[e];
                            }
                        }
                    }
                }
            });
            __syncthreads();
        });
    }

    if (a.dbg_stop == 4) { if (own[0][0] == 12345.f) a.out[0] = own[0][1]; return; }
    // ---- the region now takes the transposed node pieces VT[feature][key position] and the attention rows
    _Float16* __restrict__ VTh = reinterpret_cast<_Float16*>(smem8 + a.off_vt);
    _Float16* __restrict__ VTl = VTh + D * pa;
    _Float16* __restrict__ ATh = reinterpret_cast<_Float16*>(smem8 + a.off_att);
    _Float16* __restrict__ ATl = ATh + 8 * IBL * pa;
    {
        constexpr int PP = 4 * JP;                      // position pairs per row
        const int per_piece = D * PP;
        for (int u = tid; u < 2 * per_piece; u += NT) {
            const int piece = u >= per_piece ? 1 : 0;
            const int v = u - piece * per_piece;
            const int d = v / PP, p0 = 2 * (v - d * PP);
            const int g = p0 / JP, jj = p0 - g * JP;
            const int n0 = g * IBL + jj;
            const unsigned short* __restrict__ src = reinterpret_cast<const unsigned short*>(piece ? Vl : Vh) + d;
            const unsigned x0 = src[(n0 < K ? n0 : K - 1) * pv], x1 = src[(n0 + 1 < K ? n0 + 1 : K - 1) * pv];
            const unsigned w0 = (jj < IBL && n0 < K) ? x0 : 0u, w1 = (jj + 1 < IBL && n0 + 1 < K) ? x1 : 0u;
            *reinterpret_cast<unsigned*>((piece ? VTl : VTh) + d * pa + p0) = w0 | (w1 << 16);
        }
    }
    // scores -> softmax over the keys (reference modules.py:85-89 / :184-188): a query row lives in 8 adjacent lanes
    {
        float dv[IBL];
        g2_load_col<IBL>(dv, (lds_cptr)cdR + lj * 4 * A4, (lds_cptr)cdR + MAIN + lj * EB, 0);
        const float sinv = a.scale2[1];
        static_for<0, NOWN>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            const int ii = wave + o * NWV;
            if (ii < IBL) {
                const int irow = ii * 8 + li;
                const int irc = irow < K ? irow : K - 1;
                const float cv = cdL[ii < 4 * A4 ? li * 4 * A4 + ii : MAIN + li * EB + (ii - 4 * A4)];
                float e[IBL];
                float m = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < IBL; ++jj) {
                    const int j = lj * IBL + jj;
                    const float b = a.bias ? a.bias[(long)irc * K + (j < K ? j : K - 1)] : 0.f;
                    float v = __builtin_fmaf(own[o][jj] + cv + dv[jj], sinv, b);
                    v = j < K ? v : -INFINITY;
                    e[jj] = v;
                    m = fmaxf(m, v);
                }
                m = g2_max8(m);
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < IBL; ++jj) {
                    e[jj] = (lj * IBL + jj < K) ? soft_exp(e[jj] - m) : 0.f;
                    sum += e[jj];
                }
                sum = g2_sum8(sum);
                const float inv = irow < K ? soft_rcp(sum) : 0.f;
                // two fp16 pieces of the attention weights, JP positions per lane (the last JP - IBL hold zeros)
                unsigned hw[JP / 2], lw[JP / 2];
#pragma unroll
                for (int w2 = 0; w2 < JP / 2; ++w2) {
                    const float v0 = 2 * w2 < IBL ? e[2 * w2 < IBL ? 2 * w2 : 0] * inv : 0.f;
                    const float v1 = 2 * w2 + 1 < IBL ? e[2 * w2 + 1 < IBL ? 2 * w2 + 1 : 0] * inv : 0.f;
                    split_pair_h(v0, v1, hw[w2], lw[w2]);
                }
#pragma unroll
                for (int q = 0; q < JP / 8; ++q) {
                    *reinterpret_cast<u32x4*>(ATh + irow * pa + lj * JP + 8 * q) = u32x4{hw[4 * q], hw[4 * q + 1], hw[4 * q + 2], hw[4 * q + 3]};
                    *reinterpret_cast<u32x4*>(ATl + irow * pa + lj * JP + 8 * q) = u32x4{lw[4 * q], lw[4 * q + 1], lw[4 * q + 2], lw[4 * q + 3]};
                }
            }
        });
    }
    __syncthreads();
    if (a.dbg_stop == 5) return;
    if (wave >= IBL) return;                            // no query rows (no barrier below this point)

    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j) (raw node vectors: modules.py:93 / :191) for the 8 or 16 query rows
    // of this wave: tile row r -> node 8 wave + r (r < 8), 8 (wave + NWV) + r - 8 (a second owned index, else a repeat)
    {
        constexpr int KCP = JP / 4;                     // 32-position chunks
        const bool two = wave + NWV < IBL;
        auto tile_node = [&](int r) { return (r < 8 || !two) ? wave * 8 + (r & 7) : (wave + NWV) * 8 + (r - 8); };
        const int DT = (D + 15) >> 4;
        const int anode = tile_node(n16);
        f16x8 th[KCP], tl[KCP];                         // this lane's attention-row operand: 8 positions per chunk
#pragma unroll
        for (int c = 0; c < KCP; ++c) {
            th[c] = *reinterpret_cast<const f16x8*>(ATh + anode * pa + 32 * c + 8 * kb);
            tl[c] = *reinterpret_cast<const f16x8*>(ATl + anode * pa + 32 * c + 8 * kb);
        }
        const bool feat_regs = a.so_d == 1 || a.so_i != 1;       // registers of a lane = 4 consecutive features of one row
#pragma unroll 1
        for (int dt = 0; dt < DT; ++dt) {
            const int drow = 16 * dt + n16 < D ? 16 * dt + n16 : D - 1;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (feat_regs) {
                // out^T = VT att^T: A = VT (16 features), B = attention rows; lane (n = tile row, mb): features 16 dt + 4 mb + r
#pragma unroll
                for (int c = 0; c < KCP; ++c) {
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(VTh + drow * pa + 32 * c + 8 * kb), vl = *reinterpret_cast<const f16x8*>(VTl + drow * pa + 32 * c + 8 * kb);
                    o = g2_mfma3(vh, vl, th[c], tl[c], o);
                }
                const int d0 = 16 * dt + 4 * kb;
                const bool rv = anode < K && (n16 < 8 || two);
                float* __restrict__ orow = a.out + win * a.so_w + (long)anode * a.so_i;
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = gate_sigmoid(o[r]);
                if (a.so_d == 1 && rv && d0 + 3 < D) {
                    *reinterpret_cast<f32x4_a4*>(orow + d0) = y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rv && d0 + r < D) orow[(long)(d0 + r) * a.so_d] = y[r];
                }
            } else {
                // out = att VT^T: A = attention rows (16 tile rows), B = VT; lane (n = feature 16 dt + n, mb): tile rows 4 mb + r,
                // i.e. 4 consecutive nodes -> one 16-byte store along so_i == 1
#pragma unroll
                for (int c = 0; c < KCP; ++c) {
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(VTh + drow * pa + 32 * c + 8 * kb), vl = *reinterpret_cast<const f16x8*>(VTl + drow * pa + 32 * c + 8 * kb);
                    o = g2_mfma3(th[c], tl[c], vh, vl, o);
                }
                const int node0 = tile_node(4 * kb);
                const int d = 16 * dt + n16;
                const bool rv = d < D && (kb < 2 || two);
                float* __restrict__ op = a.out + win * a.so_w + (long)d * a.so_d + node0;
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = gate_sigmoid(o[r]);
                if (rv && node0 + 3 < K) {
                    *reinterpret_cast<f32x4_a4*>(op) = y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rv && node0 + r < K) op[r] = y[r];
                }
            }
        }
    }
}

// ---- weight pack of k_gat2, derived on the device from the fp32 tile pack of the fused projection (pack_tiles order:
// [tile][chunk of 8][lane (j, g)][4]: element (n, k) at tile n / 32, chunk k / 8, lane (n % 32) + 32 ((k % 8) / 4), k % 4):
//   W2[piece][side][row][KP] fp16, row r < E: the r-th column with a' != 0 (positive group first, the zero columns that pad
//   the groups of the tile pack to multiples of 8 are skipped), row E: the rank-1 column (c / d), rows above: zero.
__global__ void k_gat2_pack(const float* __restrict__ src, int NT_L, int Q, int D, int E, int npos, int P8, int PT, int TCP, int KP,
                            const float* __restrict__ scale, _Float16* __restrict__ dst) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = 2L * TCP * KP;
    if (idx >= total) return;
    const int k = (int)(idx % KP);
    const long r2 = idx / KP;
    const int row = (int)(r2 % TCP), side = (int)(r2 / TCP);
    float w = 0.f;
    if (row <= E && k <= D && k < 8 * Q) {
        const int n = row == E ? PT : (row < npos ? row : P8 + (row - npos));
        const long tile = (long)side * NT_L + n / 32;
        w = src[((tile * Q + k / 8) * 64 + (n % 32) + 32 * ((k % 8) / 4)) * 4 + (k % 4)];
    }
    w *= scale[0];
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    dst[idx] = h;
    dst[total + idx] = l;
}

int launch_gat2_pack(const float* src, int NT_L, int Q, int D, int E, int npos, int P8, int PT, int TCP, int KP, const float* scale,
                     void* dst, hipStream_t s) {
    const long total = 2L * TCP * KP;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_gat2_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, NT_L, Q, D, E, npos, P8, PT, TCP, KP, scale,
                       reinterpret_cast<_Float16*>(dst));
    LAUNCH_CHECK();
    return 0;
}

// LDS plan of k_gat2 for K nodes of dimension D and E embedding columns; returns false when the shape is not served
bool gat2_plan(int K, int D, int E, Gat2Plan& p) {
    p = Gat2Plan();
    if (K < 25 || K > 104 || D < 1 || D > 128 || E < 1) return false;
    const int IBL = (K + 7) / 8;
    const int NT = IBL <= 8 ? 1024 : (IBL <= 10 ? 768 : 512);
    const int NWV = NT / 64;
    const int TC = E + 1;
    const int CW = (TC + NWV - 1) / NWV;
    if (CW > 16) return false;
    const int A = IBL / 4, B = IBL % 4, EB = B == 3 ? 4 : B;
    const int TOTAL = 32 * A + 8 * EB, CS = TOTAL + 4;
    const int JP = IBL <= 8 ? 8 : 16, JQ = (IBL + 3) / 4;
    p.IBL = IBL; p.NT = NT; p.CW = CW;
    p.KC = (D + 1 + 31) / 32; p.KP = 32 * p.KC; p.pv = p.KP + 24;
    p.TCP = (TC + 16 + 7) & ~7;
    p.pa = 8 * JP + 16;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t off = 0;
    p.off_v = (int)off; off = al(off + (size_t)2 * K * p.pv * 2 + 64);       // (+ slack: the last row's chunk reads stay inside)
    p.off_cd = (int)off; off = al(off + (size_t)2 * TOTAL * 4);
    p.off_lr = (int)off;
    p.lr_wave_floats = 2 * CW * CS;
    const size_t slices = (size_t)NWV * p.lr_wave_floats * 4 + 16 * CS * 4;  // (+ one tile of slack: the prefetch of the column past a slice)
    const size_t scratch = (size_t)NWV * G2_IIP * JQ * 1024;
    const size_t vt = al((size_t)2 * D * p.pa * 2), att = al((size_t)2 * 8 * IBL * p.pa * 2);
    p.off_vt = p.off_lr; p.off_att = p.off_lr + (int)vt;
    size_t region = slices > scratch ? slices : scratch;
    if (vt + att > region) region = vt + att;
    off = al(off + region);
    p.lds_bytes = off;
    if (off > 160 * 1024) return false;
    p.ok = true;
    return true;
}

#define GAT2_CASE(I, T)                                                                                          \
    if (p.IBL == I && p.NT == T) {                                                                               \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat2<I, T>),                        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);       \
        if (e_ != hipSuccess) return (int)e_;                                                                    \
        hipLaunchKernelGGL((k_gat2<I, T>), dim3((unsigned)a.nwin), dim3(T), p.lds_bytes, s, a);                  \
        launched = true;                                                                                         \
    }

int launch_gat2(Gat2Args a, const Gat2Plan& p, hipStream_t s) {
    if (a.nwin <= 0) return 0;
    if (!p.ok) return -2;
    a.TCP = p.TCP; a.KP = p.KP; a.KC = p.KC; a.pv = p.pv; a.pa = p.pa; a.CW = p.CW;
    a.off_v = p.off_v; a.off_cd = p.off_cd; a.off_lr = p.off_lr; a.lr_wave_floats = p.lr_wave_floats; a.off_vt = p.off_vt; a.off_att = p.off_att;
    bool launched = false;
    GAT2_CASE(4, 1024) GAT2_CASE(5, 1024) GAT2_CASE(6, 1024) GAT2_CASE(7, 1024) GAT2_CASE(8, 1024)
    GAT2_CASE(9, 768) GAT2_CASE(10, 768)
    GAT2_CASE(11, 512) GAT2_CASE(12, 512) GAT2_CASE(13, 512)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}
#undef GAT2_CASE

}  // namespace mtadgat
