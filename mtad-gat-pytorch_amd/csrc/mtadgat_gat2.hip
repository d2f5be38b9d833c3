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
// Node map, the same for query rows and keys: node of (lane group g, index i) = IBL g + i (contiguous per lane: bias rows and
// attention rows are read / written in runs, and both sides of the projection read the node pieces in the same row order).
// LDS slot of (g, idx) inside a projected column: g * 4A + idx for idx < 4A (A = IBL / 4: 16-byte reads), the remaining
// IBL % 4 values in an extra region (slot MAIN + g * EB + idx - 4A).
//
// Columns beyond what a wave's LDS slice holds are taken in rounds (project a round, run its pair grid, next round), so that
// two workgroups fit a CU and the latency-bound phases of one hide behind the pair grid of the other.
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

// e^x for x <= 0 (softmax arguments after the maximum is taken out; -inf gives 0): the product x log2(e) in two pieces (the
// exact remainder of the leading piece), 5 VALU instructions instead of exp_fast's 9 (no clamp, no second piece of log2 e:
// 2e-8 |x| relative)
__device__ __forceinline__ float g2_exp(float x) {
    const float c_hi = 1.4426950216293335f;
    const float hi = x * c_hi;
    const float lo = __builtin_fmaf(x, c_hi, -hi);
    return __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(0.6931471805599453f, lo, 1.0f);
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

// sum of the NWV partial rows of one query-row index (JQ float4 per wave, waves IIP * JQ * 64 float4 apart), in wave order
template <int IBL, int NWV, int IIP>
__device__ __forceinline__ void g2_gather(float (&dst)[IBL], const f32x4* __restrict__ src) {
    constexpr int JQ = (IBL + 3) / 4, SB = 4;
#pragma unroll
    for (int s0 = 0; s0 < NWV; s0 += SB) {
        f32x4 v[SB][JQ];
#pragma unroll
        for (int w = 0; w < SB; ++w)
#pragma unroll
            for (int q = 0; q < JQ; ++q) v[w][q] = src[((s0 + w < NWV ? s0 + w : 0) * IIP * JQ + q) * 64];
#pragma unroll
        for (int w = 0; w < SB; ++w)
#pragma unroll
            for (int q = 0; q < JQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (s0 + w < NWV && 4 * q + e < IBL) dst[4 * q + e] += v[w][q][e];
    }
}

constexpr int G2_KCMAX = 5;    // 32-feature chunks of a node vector incl. the ones column (D <= 128)
__host__ __device__ constexpr int g2_iip(int ibl) { return ibl >= 9 ? 3 : 2; }      // query-row indices per reduce-scatter pass

}  // namespace

// Two workgroups per CU: while one is in its latency-bound phases (staging, projection, exchange, softmax, aggregation)
// the other one's pair grid keeps the vector ALU busy.  NT = 256 / 384 / 512 threads for accumulator blocks of
// 11-13 / 9-10 / 4-8 (two, three, four waves per SIMD by registers).
template <int IBL, int NT>
__global__ __launch_bounds__(NT, NT / 128) void k_gat2(const Gat2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    using SL = G2Slots<IBL>;
    constexpr int NWV = NT / 64;
    constexpr int A4 = SL::A, EB = SL::EB, MAIN = SL::MAIN, TOTAL = SL::TOTAL;
    constexpr int CS = TOTAL + 4;                 // floats between the columns of a slice
    constexpr int NTL = (TOTAL + 15) / 16;        // 16-slot tiles of the projection
    constexpr int KPOS = (8 * IBL + 31) / 32 * 32;    // key positions of an attention row: position = node index, zeros from K on
    constexpr int JQ = (IBL + 3) / 4;
    constexpr int IIP = g2_iip(IBL);
    constexpr int NP = (IBL + IIP - 1) / IIP;
    constexpr int NOWN = (IBL + NWV - 1) / NWV;
    static_assert(IBL >= 4 && IBL <= 13, "accumulator block");

    if (a.vmax && __uint_as_float(*a.vmax) >= 32768.f) return;         // node values beyond the fp16 pieces: k_gat runs instead
    // The first workgroups of a launch reach every CU at the same moment and would walk through their phases in step; a
    // one-time pseudo-random start delay (0..7 x ~3.5 us) spreads them, and since all workgroups take the same time the
    // spread persists: the pair grid of one workgroup then runs beside the latency-bound phases of its neighbour.
    if (blockIdx.x < (unsigned)a.stagger_blocks) {
        const unsigned nap = (blockIdx.x * 2654435761u) >> 29;
        for (unsigned i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(127);
    }
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

    // this wave's embedding columns [c0, c1) of the TC = E + 1 (the last one carries the rank-1 terms c_i / d_j), taken in
    // rounds of at most CW columns (the LDS slice of a wave holds one round)
    const int TC = a.E + 1;
    const int c0 = (wave * TC) / NWV, c1 = ((wave + 1) * TC) / NWV;

    const _Float16* __restrict__ Wg = reinterpret_cast<const _Float16*>(a.W);
    const long sstride = (long)a.TCP * a.KP, pstride = 2 * sstride;

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
        constexpr int MAXU = NT >= 512 ? 3 : 6;
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
        // the last node's chunk reads run past its row when the pitch is below KP: keep what they meet finite
        for (int u = tid; u < a.KP; u += NT) reinterpret_cast<unsigned short*>(Vl)[K * pv + u] = 0;
        // ones column and zero padding (vt == 0: the columns the units above did not reach)
        const int f0 = a.vt ? D : 4 * UR, nf = (a.KP < pv ? a.KP : pv) - f0;       // (a pitch below KP: the tail of the last chunk reads on into the next row)
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
    float acc[IBL][IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < IBL; ++jj) acc[ii][jj] = 0.f;

#pragma unroll 1
    for (int rd = 0; rd < a.NR; ++rd) {
        const int cb = c0 + rd * a.CW;                  // first column of the round
        int ncol = c1 - cb;
        ncol = ncol > a.CW ? a.CW : ncol;
        if (ncol <= 0) break;
        int nabs = (c1 < a.E ? c1 : a.E) - cb;
        nabs = nabs > a.CW ? a.CW : (nabs < 0 ? 0 : nabs);
        // ---- projection of the round's columns: D[slot][col] = sum_f V[node(slot)][f] W[col][f] on v_mfma_f32_16x16x32_f16,
        // A = node vectors (16 slots of a tile; lane (m, kb): 8 features), B = weights (lane (n, kb): 8 features of column
        // cb + n).  Result lane (n, mb): slots 16 T + 4 mb .. + 3 of column n -> one 16-byte LDS store.  The fp16 weights
        // carry the layer's power of two S; it is taken out of the scores at the end.  The accumulators of the pair grid are
        // live here (from the second round on), so the projection runs on few registers: one side (query / key) at a time,
        // one 32-feature chunk of weights at a time (the next one in flight), tiles in pairs whose MFMA chains alternate.
        {
            int offs[NTL];
#pragma unroll
            for (int T = 0; T < NTL; ++T) {
                int s = 16 * T + n16;
                s = s < TOTAL ? s : TOTAL - 1;
                int g, idx;
                if (s < MAIN) { g = s / (4 * A4 > 0 ? 4 * A4 : 1); idx = s - g * 4 * A4; }
                else { const int x = s - MAIN; g = x / (EB > 0 ? EB : 1); idx = 4 * A4 + (x - g * EB); }
                idx = idx < IBL ? idx : IBL - 1;
                int nd = g * IBL + idx;
                nd = nd < K ? nd : K - 1;
                offs[T] = nd * pv + 8 * kb;
            }
            f32x4 pacc[NTL];
#pragma unroll
            for (int T = 0; T < NTL; ++T) pacc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
            const _Float16* __restrict__ wl = Wg + (long)(cb + n16) * a.KP + 8 * kb;
            f16x8 wh = *reinterpret_cast<const f16x8*>(wl), wlo = *reinterpret_cast<const f16x8*>(wl + pstride);
            const int nit = 2 * a.KC;                 // (side, chunk) steps: the query side's chunks, then the key side's
            int c = 0, sd = 0;
#pragma unroll 1
            for (int it = 0; it < nit; ++it) {
                // weights of the next step in flight during this one (also across the change of sides)
                int cn = c + 1, sn = sd;
                if (cn == a.KC) { cn = 0; sn = 1; }
                if (it + 1 == nit) { cn = c; sn = sd; }
                const _Float16* __restrict__ wn = (a.dbg_flags & 1) ? wl : wl + sn * sstride + 32 * cn;      // (measurement: bit 0 = every step reads the same weights)
                f16x8 whn = *reinterpret_cast<const f16x8*>(wn), wln = *reinterpret_cast<const f16x8*>(wn + pstride);
#pragma unroll
                for (int T = 0; T < NTL; T += 2) {
                    const f16x8 ah0 = *reinterpret_cast<const f16x8*>(Vh + offs[T] + 32 * c), al0 = *reinterpret_cast<const f16x8*>(Vl + offs[T] + 32 * c);
                    if (T + 1 < NTL) {
                        const int T1 = T + 1 < NTL ? T + 1 : T;
                        const f16x8 ah1 = *reinterpret_cast<const f16x8*>(Vh + offs[T1] + 32 * c), al1 = *reinterpret_cast<const f16x8*>(Vl + offs[T1] + 32 * c);
                        pacc[T] = g2_mfma(al0, wh, pacc[T]);
                        pacc[T1] = g2_mfma(al1, wh, pacc[T1]);
                        pacc[T] = g2_mfma(ah0, wlo, pacc[T]);
                        pacc[T1] = g2_mfma(ah1, wlo, pacc[T1]);
                        pacc[T] = g2_mfma(ah0, wh, pacc[T]);
                        pacc[T1] = g2_mfma(ah1, wh, pacc[T1]);
                    } else {
                        pacc[T] = g2_mfma(al0, wh, pacc[T]);
                        pacc[T] = g2_mfma(ah0, wlo, pacc[T]);
                        pacc[T] = g2_mfma(ah0, wh, pacc[T]);
                    }
                }
                if (c + 1 == a.KC) {                    // a side is complete: its columns go to the slice
                    float* __restrict__ dstw = sd ? Rw : Lw;
                    if (n16 < ncol) {
#pragma unroll
                        for (int T = 0; T < NTL; ++T)
                            if (16 * T + 4 * kb < TOTAL) *reinterpret_cast<f32x4*>(dstw + n16 * CS + 16 * T + 4 * kb) = pacc[T];
                    }
#pragma unroll
                    for (int T = 0; T < NTL; ++T) pacc[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                wh = whn; wlo = wln;
                asm volatile("" : "+v"(wh), "+v"(wlo));      // keep the two register sets apart (the compiler otherwise folds the prefetch away)
                c = cn; sd = sn;
            }
        }
        if (a.dbg_stop == 2) continue;
        // ---- pair grid over the round's columns (no barrier: the slice is private to the wave, LDS operations of a wave
        // execute in order).  Two operand sets alternate: the reads of the next column are in flight during a column.
        {
            const lds_cptr lm = (lds_cptr)Lw + li * 4 * A4, le = (lds_cptr)Lw + MAIN + li * EB;
            const lds_cptr rm = (lds_cptr)Rw + lj * 4 * A4, re = (lds_cptr)Rw + MAIN + lj * EB;
            float lA[IBL], rA[IBL], lB[IBL], rB[IBL];
            if (nabs > 0) {
                g2_load_col<IBL>(lA, lm, le, 0);
                g2_load_col<IBL>(rA, rm, re, 0);
            }
            int k = 0, off = 0;
            const int npos_w = a.npos - cb;             // columns [0, npos_w) of the round have a' >= 0
#pragma unroll 1
            for (; k + 1 < nabs; k += 2) {
                float s0 = k < npos_w ? 1.f : -1.f, s1 = k + 1 < npos_w ? 1.f : -1.f;
                asm volatile("" : "+v"(s0), "+v"(s1));
                g2_load_col<IBL>(lB, lm, le, off + CS);
                g2_load_col<IBL>(rB, rm, re, off + CS);
                __builtin_amdgcn_sched_barrier(0);
                g2_pair_step<IBL>(acc, lA, rA, s0);
                __builtin_amdgcn_sched_barrier(0);
                g2_load_col<IBL>(lA, lm, le, off + 2 * CS);          // (past the round's last column: never consumed)
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
        if (cb <= a.E && a.E < cb + ncol) {             // the round that projected the rank-1 column keeps it for the softmax
            const int kc = a.E - cb;
            for (int s = lane; s < TOTAL; s += 64) {
                cdL[s] = Lw[kc * CS + s];
                cdR[s] = Rw[kc * CS + s];
            }
        }
    }
    __syncthreads();                                    // every slice is consumed: the region becomes the exchange scratch
    if (a.dbg_stop == 2) return;
    if (a.dbg_stop == 3) { if (acc[0][0] == 12345.f) a.out[0] = acc[1][1]; return; }

    // ---- reduce-scatter of the partial sums: pass p moves the rows ii in [IIP p, IIP p + IIP); wave ii % NWV adds the NWV
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
            static_for<0, IIP>([&](auto ic) {
                constexpr int iloc = decltype(ic)::value, ii = p * IIP + iloc;
                if constexpr (ii < IBL) {
#pragma unroll
                    for (int q = 0; q < JQ; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 4 * q + e < IBL ? acc[ii][4 * q + e < IBL ? 4 * q + e : 0] : 0.f;
                        scr[((wave * IIP + iloc) * JQ + q) * 64 + lane] = v;
                    }
                }
            });
            __syncthreads();
            // this wave's rows among those of the pass: ii = wave + o NWV (the accumulator index o is static, the position
            // of the row inside the pass is not)
            static_for<0, NOWN>([&](auto oc) {
                constexpr int o = decltype(oc)::value;
                const int ii = wave + o * NWV;
                if (ii >= p * IIP && ii < (p + 1) * IIP && ii < IBL) g2_gather<IBL, NWV, IIP>(own[o], scr + ((ii - p * IIP) * JQ) * 64 + lane);
            });
            __syncthreads();
        });
    }
    if (a.dbg_stop == 4) { if (own[0][0] == 12345.f) a.out[0] = own[0][1]; return; }

    // attention bias rows of this wave's query rows: requested now, consumed after the transposition below
    float bq[NOWN][IBL];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int ii = wave + o * NWV;
        const int irow = li * IBL + ii;
        const int irc = irow < K ? irow : K - 1;
#pragma unroll
        for (int jj = 0; jj < IBL; ++jj) {
            const int j = lj * IBL + jj;
            bq[o][jj] = (a.bias && ii < IBL) ? a.bias[(long)irc * K + (j < K ? j : K - 1)] : 0.f;
        }
    }

    // ---- the slice region now takes the transposed node pieces VT[feature][key position] (B / A operand of the aggregation)
    _Float16* __restrict__ VTh = reinterpret_cast<_Float16*>(smem8 + a.off_vt);
    _Float16* __restrict__ VTl = VTh + D * pa;
    {
        // lane = pair of adjacent nodes (positions 2 np, 2 np + 1; zeros from K on), wave = feature residue: VT[d][2 np ..] <-
        // (V[2 np][d], V[2 np + 1][d]) for d = wave, wave + NWV, ...; the reads of a batch of features are issued together
        constexpr int NPP = KPOS / 2, UN = 4;
        const float ninv = 1.0f / (float)NPP;
        const int fg = (int)(((float)tid + 0.5f) * ninv), np = tid - fg * NPP;      // (NT is not a multiple of NPP in general)
        const int NFG = NT / NPP;                                                   // feature residues served per pass
        const int n0 = 2 * np, n0c = n0 < K ? n0 : K - 1, n1c = n0 + 1 < K ? n0 + 1 : K - 1;
        const unsigned m0 = n0 < K ? 0xffffu : 0u, m1 = n0 + 1 < K ? 0xffffu : 0u;
        const unsigned short* __restrict__ sh = reinterpret_cast<const unsigned short*>(Vh);
        const unsigned short* __restrict__ sl = reinterpret_cast<const unsigned short*>(Vl);
        if (fg < NFG) {
            for (int d0 = fg; d0 < D; d0 += UN * NFG) {
                unsigned xh[UN], xl[UN];
#pragma unroll
                for (int n = 0; n < UN; ++n) {
                    const int d = d0 + n * NFG < D ? d0 + n * NFG : D - 1;
                    xh[n] = (sh[n0c * pv + d] & m0) | ((sh[n1c * pv + d] & m1) << 16);
                    xl[n] = (sl[n0c * pv + d] & m0) | ((sl[n1c * pv + d] & m1) << 16);
                }
#pragma unroll
                for (int n = 0; n < UN; ++n) {
                    const int d = d0 + n * NFG;
                    if (d < D) {
                        *reinterpret_cast<unsigned*>(VTh + d * pa + n0) = xh[n];
                        *reinterpret_cast<unsigned*>(VTl + d * pa + n0) = xl[n];
                    }
                }
            }
        }
    }
    __syncthreads();                                    // VT complete; the node pieces are dead: their region takes the attention rows
    if (a.dbg_stop == 5) return;
    // ---- per owned index ii (8 query rows: nodes IBL li + ii): scores -> softmax over the keys (reference modules.py:85-89 /
    // :184-188; a query row lives in 8 adjacent lanes) -> attention rows as two fp16 pieces into this wave's private LDS
    // rows -> h_i = sigmoid(sum_j att_ij V_j) (raw node vectors: modules.py:93 / :191) on the matrix pipe.
    // Output: so_d == 1 (features of a row contiguous: temporal layer) straight from the registers, 16 bytes per lane;
    // otherwise through an LDS tile [feature][node] that the whole workgroup then copies out in runs along so_i.
    const bool feat_regs = a.so_d == 1;
    float* __restrict__ otile = reinterpret_cast<float*>(smem8 + a.off_tile);
    const int tpitch = K | 1;                           // odd pitch: the 4-byte tile writes of a lane group spread over the banks
    if (wave < IBL) {
        _Float16* __restrict__ ATh = reinterpret_cast<_Float16*>(smem8 + a.off_att) + wave * (2 * 8 * pa);
        _Float16* __restrict__ ATl = ATh + 8 * pa;
        float dv[IBL];
        g2_load_col<IBL>(dv, (lds_cptr)cdR + lj * 4 * A4, (lds_cptr)cdR + MAIN + lj * EB, 0);
        const float sinv = a.scale2[1];
        constexpr int KCP = KPOS / 32;                  // 32-position chunks
        const int DT = (D + 15) >> 4;
        static_for<0, NOWN>([&](auto oc) {
            constexpr int o = decltype(oc)::value;
            const int ii = wave + o * NWV;
            if (ii < IBL) {
                const int irow = li * IBL + ii;
                const float cv = cdL[ii < 4 * A4 ? li * 4 * A4 + ii : MAIN + li * EB + (ii - 4 * A4)];
                float e[IBL];
                float m = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < IBL; ++jj) {
                    const int j = lj * IBL + jj;
                    float v = __builtin_fmaf(own[o][jj] + cv + dv[jj], sinv, bq[o][jj]);
                    v = j < K ? v : -INFINITY;
                    e[jj] = v;
                    m = fmaxf(m, v);
                }
                m = g2_max8(m);
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < IBL; ++jj) {
                    e[jj] = (lj * IBL + jj < K) ? g2_exp(e[jj] - m) : 0.f;
                    sum += e[jj];
                }
                sum = g2_sum8(sum);
                const float inv = irow < K ? soft_rcp(sum) : 0.f;
                // two fp16 pieces of the attention weights at positions IBL lj + jj of row li: 4-byte stores of position pairs
                // (the run of a lane starts at an odd position when IBL and lj are odd: its first element then goes alone, else
                // its last one -- IBL odd only), then the zero tail of the row
                {
                    unsigned short* __restrict__ rh = reinterpret_cast<unsigned short*>(ATh) + li * pa + lj * IBL;
                    unsigned short* __restrict__ rl = reinterpret_cast<unsigned short*>(ATl) + li * pa + lj * IBL;
                    constexpr bool ODD = (IBL & 1) != 0;
                    const bool shifted = ODD && (lj & 1);
#pragma unroll
                    for (int w2 = 0; w2 < IBL / 2; ++w2) {
                        const float v0 = (shifted ? e[2 * w2 + 1 < IBL ? 2 * w2 + 1 : 0] : e[2 * w2]) * inv;
                        const float v1 = (shifted ? e[2 * w2 + 2 < IBL ? 2 * w2 + 2 : 0] : e[2 * w2 + 1]) * inv;
                        unsigned hw, lw;
                        split_pair_h(v0, v1, hw, lw);
                        const int pos = 2 * w2 + (shifted ? 1 : 0);
                        *reinterpret_cast<unsigned*>(rh + pos) = hw;
                        *reinterpret_cast<unsigned*>(rl + pos) = lw;
                    }
                    if (ODD) {
                        const float vs = (shifted ? e[0] : e[IBL - 1]) * inv;
                        unsigned hw, lw;
                        split_pair_h(vs, 0.f, hw, lw);
                        const int pos = shifted ? 0 : IBL - 1;
                        rh[pos] = (unsigned short)hw;
                        rl[pos] = (unsigned short)lw;
                    }
                    if (8 * IBL < KPOS) {
                        for (int z = 8 * IBL + 2 * lj; z < KPOS; z += 16) {
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(ATh) + li * pa + z) = 0u;
                            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(ATl) + li * pa + z) = 0u;
                        }
                    }
                }
                // aggregation for the 8 rows (tile rows 8..15 repeat them).  LDS operations of a wave execute in order: the
                // reads below see the rows written above.  out^T = VT att^T: A = VT (16 features), B = attention rows;
                // result lane (n = tile row, mb): features 16 dt + 4 mb + r of node IBL (n & 7) + ii.  Two feature tiles in
                // flight: their MFMA chains alternate.
                const int arow = n16 & 7;
                f16x8 th[KCP], tl[KCP];
#pragma unroll
                for (int c = 0; c < KCP; ++c) {
                    th[c] = *reinterpret_cast<const f16x8*>(ATh + arow * pa + 32 * c + 8 * kb);
                    tl[c] = *reinterpret_cast<const f16x8*>(ATl + arow * pa + 32 * c + 8 * kb);
                }
                const int node = arow * IBL + ii;
                const bool rv = node < K && n16 < 8;
                float* __restrict__ orow = a.out + win * a.so_w + (long)node * a.so_i;
#pragma unroll 1
                for (int dt = 0; dt < DT; dt += 2) {
                    const int dra = 16 * dt + n16 < D ? 16 * dt + n16 : D - 1;
                    const int drb = 16 * dt + 16 + n16 < D ? 16 * dt + 16 + n16 : D - 1;
                    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < KCP; ++c) {
                        const f16x8 vha = *reinterpret_cast<const f16x8*>(VTh + dra * pa + 32 * c + 8 * kb), vla = *reinterpret_cast<const f16x8*>(VTl + dra * pa + 32 * c + 8 * kb);
                        const f16x8 vhb = *reinterpret_cast<const f16x8*>(VTh + drb * pa + 32 * c + 8 * kb), vlb = *reinterpret_cast<const f16x8*>(VTl + drb * pa + 32 * c + 8 * kb);
                        oa = g2_mfma(vla, th[c], oa);
                        ob = g2_mfma(vlb, th[c], ob);
                        oa = g2_mfma(vha, tl[c], oa);
                        ob = g2_mfma(vhb, tl[c], ob);
                        oa = g2_mfma(vha, th[c], oa);
                        ob = g2_mfma(vhb, th[c], ob);
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int d0 = 16 * (dt + half) + 4 * kb;
                        f32x4 y;
#pragma unroll
                        for (int r = 0; r < 4; ++r) y[r] = gate_sigmoid(half ? ob[r] : oa[r]);
                        if (feat_regs) {
                            if (rv && d0 + 3 < D) {
                                *reinterpret_cast<f32x4_a4*>(orow + d0) = y;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (rv && d0 + r < D) orow[d0 + r] = y[r];
                            }
                        } else if (a.off_tile >= 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (rv && d0 + r < D) otile[(d0 + r) * tpitch + node] = y[r];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (rv && d0 + r < D) orow[(long)(d0 + r) * a.so_d] = y[r];
                        }
                    }
                }
            }
        });
    }
    if (!feat_regs && a.off_tile >= 0) {
        __syncthreads();
        // tile [feature d][node] -> out[win][d * so_d + node * so_i]: consecutive threads take consecutive nodes
        const float kinv = 1.0f / (float)K;
        for (int u = tid; u < D * K; u += NT) {
            const int d = (int)(((float)u + 0.5f) * kinv), node = u - d * K;
            a.out[win * a.so_w + (long)d * a.so_d + (long)node * a.so_i] = otile[d * tpitch + node];
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

// LDS plan of k_gat2 for K nodes of dimension D and E embedding columns; returns false when the shape is not served.
// Layout: [column slices of the waves | rank-1 terms | node pieces]; after the pair grid the slice region holds the exchange
// scratch, then the transposed node pieces, and the (then dead) node-piece region the waves' private attention rows.
bool gat2_plan(int K, int D, int E, bool tile_out, Gat2Plan& p) {
    p = Gat2Plan();
    if (K < 25 || K > 104 || D < 1 || D > 128 || E < 1) return false;
    const int IBL = (K + 7) / 8;
    const int NT = IBL <= 8 ? 512 : (IBL <= 10 ? 384 : 256);
    const int NWV = NT / 64;
    const int TC = E + 1;
    const int per_wave = (TC + NWV - 1) / NWV;        // columns of the widest wave
    const int A = IBL / 4, B = IBL % 4, EB = B == 3 ? 4 : B;
    const int TOTAL = 32 * A + 8 * EB, CS = TOTAL + 4;
    const int KPOS = (8 * IBL + 31) / 32 * 32, JQ = (IBL + 3) / 4;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    p.IBL = IBL; p.NT = NT;
    p.KC = (D + 1 + 31) / 32; p.KP = 32 * p.KC;
    p.TCP = (TC + 16 + 7) & ~7;
    p.pa = KPOS + 16;
    const size_t cdbytes = al((size_t)2 * TOTAL * 4);
    const size_t scratch = (size_t)NWV * g2_iip(IBL) * JQ * 1024;
    const size_t vt = al((size_t)2 * D * p.pa * 2), att = al((size_t)NWV * 2 * 8 * p.pa * 2);
    const size_t tile = tile_out ? al((size_t)D * (K | 1) * 4) : 0;
    // fewest rounds (widest round) that leave room for two workgroups per CU; one per CU when even 4-column rounds do not.
    // Node-piece pitch: the padded feature count + 8 (rows start 4 banks apart); the bare feature count rounded to 8 when
    // that is what lets two workgroups fit (the last chunk's reads then run into the next row: finite values times zero weights)
    for (int budget : {80 * 1024, 160 * 1024}) {
        for (int nr = 1; nr <= per_wave; ++nr) {
            const int cw = (per_wave + nr - 1) / nr;
            if (cw > 16) continue;
            for (int pv : {p.KP + 8, ((D + 1 + 7) & ~7) + 8, (D + 1 + 7) & ~7}) {
                const size_t vbytes = al((size_t)2 * K * pv * 2 + 2 * (size_t)p.KP);
                const size_t slices = (size_t)NWV * 2 * cw * CS * 4;
                size_t region = slices > scratch ? slices : scratch;
                if (vt + tile > region) region = vt + tile;
                const size_t vreg = vbytes > att ? vbytes : att;
                const size_t total = al(region) + cdbytes + vreg;
                if (total <= (size_t)budget) {
                    p.CW = cw; p.NR = nr; p.pv = pv;
                    p.off_lr = 0; p.lr_wave_floats = 2 * cw * CS;
                    p.off_cd = (int)al(region);
                    p.off_v = p.off_cd + (int)cdbytes;
                    p.off_vt = 0; p.off_att = p.off_v;
                    p.off_tile = tile_out ? (int)vt : -1;
                    p.lds_bytes = total;
                    p.ok = true;
                    return true;
                }
            }
            if (cw <= 4) break;
        }
    }
    return false;
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
    a.TCP = p.TCP; a.KP = p.KP; a.KC = p.KC; a.pv = p.pv; a.pa = p.pa; a.CW = p.CW; a.NR = p.NR;
    a.off_v = p.off_v; a.off_cd = p.off_cd; a.off_lr = p.off_lr; a.lr_wave_floats = p.lr_wave_floats; a.off_vt = p.off_vt; a.off_att = p.off_att;
    a.off_tile = a.so_d == 1 ? -1 : p.off_tile;
    a.stagger_blocks = 1024;
    if (const char* e_ = getenv("MTADGAT_G2_FLAGS")) a.dbg_flags = atoi(e_);
    if (const char* e_ = getenv("MTADGAT_G2_ONLY")) if (atoi(e_) != a.K) return 0;      // (measurement: only the layer with that many nodes runs)
    if (const char* e_ = getenv("MTADGAT_G2_STAGGER")) a.stagger_blocks = atoi(e_);      // (measurement hook)
    bool launched = false;
    GAT2_CASE(4, 512) GAT2_CASE(5, 512) GAT2_CASE(6, 512) GAT2_CASE(7, 512) GAT2_CASE(8, 512)
    GAT2_CASE(9, 384) GAT2_CASE(10, 384)
    GAT2_CASE(11, 256) GAT2_CASE(12, 256) GAT2_CASE(13, 256)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}
#undef GAT2_CASE

}  // namespace mtadgat
