// k_gru (register-resident recurrence for large batches): kernel template + launch helpers, shared by the two
// translation units that instantiate it (fp32 build: mtadgat_gru_f32.hip, bf16 build: mtadgat_gru_bf16.hip) so that
// they compile in parallel.
#pragma once
#include "mtadgat_device.h"


namespace mtadgat {

// ---------------------------------------------------------------------------
// GRU: 32 (MW = 1) or 64 (MW = 2) windows per wave, hidden state resident in registers in F-layout for
// all T steps; W_ih / W_hh streamed from L2 in packed order; gates r|z|n as
// torch.nn.GRU (reference GRULayer.forward modules.py:235-238, RNNDecoder
// modules.py:255-257).  Optional per-step Linear on the new hidden state
// (ReconstructionModel.fc, modules.py:282).
//   XMODE 0: input rows from memory, X[(win*T + t)*ldx + k]
//   XMODE 1: the reference's decoder input h_end.repeat_interleave(W).view(b,W,-1)
//            (modules.py:279): x_t[j] = hin[(t*Hin + j) / T]; only NM <= 8*Qx distinct
//            hin entries m0[t] .. m0[t]+NM-1 occur at step t, and the packed "Wx" for
//            step t holds W_ih summed over the j that map to each of them.
// ---------------------------------------------------------------------------
// XMODE 0: input rows X[(win*T + t)*ldx + k], packed x part has Qxp = 3n chunks (zero chunks past Qx)
// XMODE 1: decoder input (see above) with exactly one 8-wide chunk per step (NM <= 8)
// XMODE 2: decoder input with Qxp = 3n chunks
// DROP   : trailing all-padding chunks of the recurrent part that are skipped (H <= 8*(4*NCG - DROP))
// Input rows must be 16-byte aligned (XMODE 0) and zero padded as far as the loads reach; every load
// in the loop nest is unconditional and the nest has no data-dependent control flow, so the compiler
// can count the outstanding loads exactly and waits with vmcnt(N > 0): the weight ring stays full.
// (With guarded loads it fell back to vmcnt(0..2) before every MFMA group: 79k instead of 31k cycles
// per hidden tile and step.)
// MW = 32-window groups per wave.  MW = 2 with one wave per SIMD beats two MW = 1 waves per SIMD (matrix
// pipe 85 % vs 82 % busy on the GRU layer, 74 % vs 69 % on the decoder): the MFMAs of one wave issue back to back, interleaving two
// waves leaves bubbles; each weight chunk is also fetched once for 64 windows.
// BF: bf16 operand build (v_mfma_f32_32x32x16_bf16, 16 features per chunk; fp32 accumulators, state and gates):
// the same chunk sequence with half as many, twice as wide chunks -- see mtadgat_device.h for the element order.
// MW = 2 launches 4 such waves per workgroup (one per SIMD): they are independent except that the per-step barrier
// keeps them within a few chunks of each other, so the packed-weight chunks one wave pulls from L2 are still in the
// CU's vector L1 when the other three ask for them (as separate one-wave workgroups they drift apart and every wave
// streams the whole 1.9 MB / 1 MB image from L2 each step).
// QXC > 0 (bf16 build, row input): the QXC packed input chunks of a step stay in registers for all hidden tiles and
// are replaced by the next step's while the last tile consumes them -- read once per step instead of once per tile.
// With the MFMAs 16x cheaper the five re-reads of x (5 x 67 KB per window: they miss the L2, the XCD's waves stream
// more than its 4 MB between two tiles) made the bf16 build HBM-bound at ~4 TB/s.
// X3 (with BF geometry): split-bf16 operands (mtadgat_device.h): every weight chunk comes as three bf16 pieces per
// gate, every activation chunk is split into three pieces where it is consumed, six bf16 MFMAs per gate and chunk --
// fp32-class results on the bf16 matrix pipe.  Two ring stages (a chunk is 36 MFMAs = 1.15 k cycles of cover).
template <int NCG, int XMODE, bool FC, int DROP, int MW, bool BF = false, int QXC = 0, bool X3 = false>
// (X3, MW = 1 at two waves per SIMD was tried: 256 registers per wave mean 60-140 spilled VGPRs; 32 768 windows 17.8 -> 22.6 ms)
__global__ __launch_bounds__((MW == 2 ? 256 : 64), ((NCG <= 6 && MW == 1 && !X3) ? 2 : 1)) void k_gru(const GruArgs a) {
    static_assert(!X3 || (BF && QXC == 0), "split-bf16 build uses the bf16 chunk geometry");
    extern __shared__ __attribute__((aligned(16))) float hn_dyn[];
    // the chunk-major kernel (mtadgat_gru_cm.hip) serves this launch when the input range allows two-piece operands
    if (X3 && a.skip_xh && a.vmax != nullptr && __uint_as_float(*a.vmax) < 32768.f) return;
    constexpr int WPB = MW == 2 ? 4 : 1;
    const int lane = threadIdx.x & 63;
    const int wv = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float (*hn_s)[NCG][16][64] = reinterpret_cast<float (*)[NCG][16][64]>(hn_dyn + (size_t)wv * MW * NCG * 1024);
    const int i = lane & 31, g = lane >> 5;
    long win[MW], winc[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) {
        win[w] = (((long)blockIdx.x * WPB + wv) * MW + w) * 32 + i;
        winc[w] = win[w] < a.B ? win[w] : a.B - 1;
    }
    const int T = a.T, Qx = a.Qx;
    const int Qxp = (XMODE == 1) ? 1 : ((BF && XMODE == 0 && QXC > 0) ? QXC : a.Qxp);
    constexpr int Qh = BF ? 2 * NCG : 4 * NCG;    // recurrent chunks that can be non-zero
    constexpr int Qhe = Qh - DROP;                // ... and as used
    // ring depth: 3 chunks of weights in flight (fp32: 36 MFMAs x 64 cycles ~ 2.3k cycles of cover).  A ring of 6 was
    // tried for the bf16 build, whose chunks are 8x shorter: no gain -- that build was bound by the input re-reads (XR)
    constexpr bool XR = BF && XMODE == 0 && QXC > 0;
    constexpr int R = X3 ? (MW == 1 ? 3 : 2) : 3;
    constexpr int NP = X3 ? 3 : 1;                // operand pieces per weight word
    constexpr int WN = 3 * NP;                    // 16-byte words per chunk and lane: [gate][piece]
    constexpr int XW = X3 ? 2 : 1;                // registers of an input chunk: the two fp32 halves (X3) or the operand itself
    constexpr int WNH = X3 ? 6 : WN;              // words of a recurrent chunk: X3 = two fp16 pieces per gate (|h| <= 1, scaled weights)
    // X3: all weights of the layer carry the power-of-two factor S (fp16 range of the recurrent pieces): the
    // accumulators hold S times the pre-activations
    const float wS = X3 ? a.scale[0] : 1.f, wInvS = X3 ? a.scale[1] : 1.f;
    // X3: the first qb input chunks (the convolution's unbounded outputs) come as three bf16 pieces, the others (sigmoid
    // outputs of the attention layers, a previous layer's or the encoder's state: all in [-1, 1]) as two fp16 pieces
    // (when the producing convolution recorded an output range that fits fp16, all input chunks go as two pieces: Wx2)
    const bool xh = X3 && a.vmax != nullptr && __uint_as_float(*a.vmax) < 32768.f;
    const int qb = X3 ? (xh ? 0 : a.qb3) : 0;
    const f32x4* __restrict__ Wx0 = (X3 && xh) ? a.Wx2 : a.Wx;
    constexpr int ROT = (XMODE == 1) ? (1 + Qhe) % R : Qhe % R;   // ring phase advance per hidden tile
    constexpr int Qf = 4 * NCG - (BF ? 0 : DROP); // fp32 8-feature chunks of h used by the per-step Linear (always fp32)
    const int S = Qxp + Qhe;

    f32x16 h[MW][NCG];
#pragma unroll
    for (int w = 0; w < MW; ++w)
#pragma unroll
        for (int c = 0; c < NCG; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[w][c][r] = 0.f;

    // ---- weight stream: one continuous sequence of chunks [tile c][x chunks 0..Qxp) [h chunks 0..Qhe)
    // per step, fetched through a 3-stage register ring that never drains: the cursor runs 3 chunks
    // (36 MFMAs ~ 2.3k cycles) ahead of the MFMAs across the x/h, tile and step boundaries.
    // (A variant with per-tile base pointers + compile-time offsets instead of the cursor needed ~20 more
    // VGPRs and measured slower.)
    // prefetch cursor: wave-uniform running pointers into the two packed streams (they stay in SGPRs; the
    // per-lane part of every weight address is the constant lane*16 bytes), advanced by one chunk per fetch
    int pc = 0, ps = 0, pt = 0;
    const f32x4* __restrict__ pwx = Wx0;           // next input-part chunk to fetch
    const f32x4* __restrict__ pwh = a.Wh;          // next recurrent-part chunk to fetch
    auto wload = [&](f32x4 (&dst)[WN]) {
        const bool isx = ps < Qxp;
        const f32x4* __restrict__ p = (isx ? pwx : pwh) + lane;
#pragma unroll
        for (int j = 0; j < WN; ++j) dst[j] = p[64 * j];
        pwx += isx ? (X3 ? (ps < qb ? 64 * WN : 64 * WNH) : 64 * WN) : 0;
        pwh += isx ? 0 : 64 * WNH;                 // (X3: the three extra words fetched for a recurrent chunk are the next chunk's first)
        const bool ws = (ps + 1 == S);             // end of this tile's chunk sequence
        ps = ws ? 0 : ps + 1;
        pwh += ws ? (a.whs - Qhe) * (64 * WNH) : 0; // skip the unused all-padding chunks of the tile
        const bool wc = ws && (pc + 1 == NCG);     // end of the step
        pc = ws ? (wc ? 0 : pc + 1) : pc;
        pt = wc ? pt + 1 : pt;
        pwh = wc ? a.Wh : pwh;
        // input-part weights: per step for the decoder ([t][c][Qxp], contiguous), shared by all steps otherwise
        pwx = wc ? ((XMODE == 0 || pt >= T) ? Wx0 : pwx) : pwx;
    };
    const float* xbase[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) xbase[w] = (XMODE == 0) ? a.X + winc[w] * T * a.ldx + 4 * g : a.X + winc[w] * a.ldx;
    auto loadx_t = [&](int w, int t, int q) -> f32x4 {
        const int qq = q < Qx ? q : Qx - 1;       // padded chunks re-read the last real one (their weights are zero)
        if (XMODE == 0) return *reinterpret_cast<const f32x4*>(xbase[w] + (long)t * a.ldx + 8 * qq);
        const int k0 = a.m0[t] + 8 * qq + 4 * g, kmax = (int)a.ldx - 1;     // stay inside the (zero padded) row
        f32x4 v;
        v[0] = xbase[w][min(k0, kmax)]; v[1] = xbase[w][min(k0 + 1, kmax)];
        v[2] = xbase[w][min(k0 + 2, kmax)]; v[3] = xbase[w][min(k0 + 3, kmax)];
        return v;
    };

    // B operand of packed input chunk q: fp32 build = the 8-feature chunk itself; bf16 build = chunks 2q, 2q+1 converted
    // (XMODE 1: the single folded chunk, upper half zero)
    auto loadxq = [&](f32x4 (&dst)[XW], int w, int t, int q) {
        if (X3) {                                   // the fp32 halves; split into pieces where consumed
            dst[0] = loadx_t(w, t, XMODE == 1 ? q : 2 * q);
            dst[XW - 1] = XMODE == 1 ? f32x4{0.f, 0.f, 0.f, 0.f} : loadx_t(w, t, 2 * q + 1);
        } else if (!BF) dst[0] = loadx_t(w, t, q);
        else if (XMODE == 1) dst[0] = cvt8(loadx_t(w, t, q), f32x4{0.f, 0.f, 0.f, 0.f});
        else dst[0] = cvt8(loadx_t(w, t, 2 * q), loadx_t(w, t, 2 * q + 1));
    };
    // one chunk into the three gate accumulators
    auto gates3 = [&](const f32x4 (&wv)[WN], const f32x4 (&xv)[XW], f32x16& g0, f32x16& g1, f32x16& g2) {
        if constexpr (X3) {
            f32x4 xs[3];
            split3(xv[0], xv[XW - 1], xs[0], xs[1], xs[2]);
            const f32x4 w0[3] = {wv[0], wv[1], wv[2]}, w1[3] = {wv[3], wv[4], wv[5]}, w2[3] = {wv[6], wv[7], wv[8]};
            // term by term across the gates: consecutive MFMAs never depend on each other
            g0 = mfma_bf(w0[0], xs[2], g0); g1 = mfma_bf(w1[0], xs[2], g1); g2 = mfma_bf(w2[0], xs[2], g2);
            g0 = mfma_bf(w0[2], xs[0], g0); g1 = mfma_bf(w1[2], xs[0], g1); g2 = mfma_bf(w2[2], xs[0], g2);
            g0 = mfma_bf(w0[1], xs[1], g0); g1 = mfma_bf(w1[1], xs[1], g1); g2 = mfma_bf(w2[1], xs[1], g2);
            g0 = mfma_bf(w0[0], xs[1], g0); g1 = mfma_bf(w1[0], xs[1], g1); g2 = mfma_bf(w2[0], xs[1], g2);
            g0 = mfma_bf(w0[1], xs[0], g0); g1 = mfma_bf(w1[1], xs[0], g1); g2 = mfma_bf(w2[1], xs[0], g2);
            g0 = mfma_bf(w0[0], xs[0], g0); g1 = mfma_bf(w1[0], xs[0], g1); g2 = mfma_bf(w2[0], xs[0], g2);
        } else {
            const f32x4 w3[3] = {wv[0], wv[1], wv[2]};
            mfma_x3<BF>(w3, xv[0], g0, g1, g2);
        }
    };
    // X3, two window groups per wave: software pipeline -- the 18 MFMAs of one (chunk, group) item run while the VALU
    // splits the operand of the next item (the wave issues in order: a split placed after the MFMAs of its own item
    // would wait behind them with the matrix pipe idle for ~220 cycles per item)
    constexpr bool PIPE = X3 && MW == 2;
    auto mfma18 = [&](const f32x4 (&wv)[WN], const f32x4 (&xs)[3], f32x16& g0, f32x16& g1, f32x16& g2) {
        if constexpr (X3) {
            g0 = mfma_bf(wv[0], xs[2], g0); g1 = mfma_bf(wv[3], xs[2], g1); g2 = mfma_bf(wv[6], xs[2], g2);
            g0 = mfma_bf(wv[2], xs[0], g0); g1 = mfma_bf(wv[5], xs[0], g1); g2 = mfma_bf(wv[8], xs[0], g2);
            g0 = mfma_bf(wv[1], xs[1], g0); g1 = mfma_bf(wv[4], xs[1], g1); g2 = mfma_bf(wv[7], xs[1], g2);
            g0 = mfma_bf(wv[0], xs[1], g0); g1 = mfma_bf(wv[3], xs[1], g1); g2 = mfma_bf(wv[6], xs[1], g2);
            g0 = mfma_bf(wv[1], xs[0], g0); g1 = mfma_bf(wv[4], xs[0], g1); g2 = mfma_bf(wv[7], xs[0], g2);
            g0 = mfma_bf(wv[0], xs[0], g0); g1 = mfma_bf(wv[3], xs[0], g1); g2 = mfma_bf(wv[6], xs[0], g2);
        }
    };
    // the product terms in triples (one MFMA per gate), the four value pairs of the next operand's split between them;
    // sched_barrier keeps the source order (the group-barrier solver gave up on whole stages).
    // CUR3: the current item is an input chunk (three bf16 pieces, 6 terms) or a recurrent chunk (two fp16 pieces, 3 terms);
    // NEXT: the item whose operand is split meanwhile -- 3 bf16 pieces, 2 fp16 pieces, or none (0)
    auto stage = [&](auto cur3_tag, auto next_tag, const f32x4 (&wv)[WN], const f32x4 (&xs)[3], f32x16& g0, f32x16& g1, f32x16& g2,
                     const f32x4 ra, const f32x4 rb, f32x4 (&xn)[3]) {
        if constexpr (X3) {
            constexpr bool CUR3 = decltype(cur3_tag)::value;
            constexpr int NEXT = decltype(next_tag)::value;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 hw = {0, 0, 0, 0}, mw = {0, 0, 0, 0}, lw = {0, 0, 0, 0};
            auto pair = [&](const int pp) {
                const float v0 = pp < 2 ? ra[2 * pp] : rb[2 * pp - 4], v1 = pp < 2 ? ra[2 * pp + 1] : rb[2 * pp - 3];
                if constexpr (NEXT == 3) {
                    const unsigned hh = pack_bf16(v0, v1);
                    const float r0 = v0 - __builtin_bit_cast(float, hh << 16), r1 = v1 - __builtin_bit_cast(float, hh & 0xffff0000u);
                    const unsigned mm = pack_bf16(r0, r1);
                    const float s0 = r0 - __builtin_bit_cast(float, mm << 16), s1 = r1 - __builtin_bit_cast(float, mm & 0xffff0000u);
                    hw[pp] = hh; mw[pp] = mm; lw[pp] = pack_bf16(s0, s1);
                } else if constexpr (NEXT == 2) {
                    unsigned hh, ll;
                    split_pair_h(v0, v1, hh, ll);
                    hw[pp] = hh; mw[pp] = ll;
                }
            };
            if constexpr (CUR3) {
                auto triple = [&](const int wi, const int xi) {
                    g0 = mfma_bf(wv[wi], xs[xi], g0); g1 = mfma_bf(wv[3 + wi], xs[xi], g1); g2 = mfma_bf(wv[6 + wi], xs[xi], g2);
                };
                triple(0, 2); __builtin_amdgcn_sched_barrier(0); pair(0); __builtin_amdgcn_sched_barrier(0);
                triple(2, 0); __builtin_amdgcn_sched_barrier(0); pair(1); __builtin_amdgcn_sched_barrier(0);
                triple(1, 1); __builtin_amdgcn_sched_barrier(0); pair(2); __builtin_amdgcn_sched_barrier(0);
                triple(0, 1); __builtin_amdgcn_sched_barrier(0); pair(3); __builtin_amdgcn_sched_barrier(0);
                triple(1, 0);
                triple(0, 0);
            } else {
                // recurrent chunk: words [gate][hi, lo]; terms w_h x_l, w_l x_h, w_h x_h
                auto triple = [&](const int wi, const int xi) {
                    g0 = mfma_h(wv[wi], xs[xi], g0); g1 = mfma_h(wv[2 + wi], xs[xi], g1); g2 = mfma_h(wv[4 + wi], xs[xi], g2);
                };
                triple(0, 1); __builtin_amdgcn_sched_barrier(0); pair(0); pair(1); __builtin_amdgcn_sched_barrier(0);
                triple(1, 0); __builtin_amdgcn_sched_barrier(0); pair(2); pair(3); __builtin_amdgcn_sched_barrier(0);
                triple(0, 0);
            }
            if constexpr (NEXT != 0) {
                xn[0] = __builtin_bit_cast(f32x4, hw); xn[1] = __builtin_bit_cast(f32x4, mw); xn[2] = __builtin_bit_cast(f32x4, lw);
            }
        }
    };
    using T3 = std::true_type;
    using T2 = std::false_type;
    using N0 = std::integral_constant<int, 0>;
    using N2 = std::integral_constant<int, 2>;
    using N3 = std::integral_constant<int, 3>;
    // one recurrent chunk, not pipelined (32-window waves): two fp16 pieces of h, three terms
    auto gates_h = [&](const f32x4 (&wv)[WN], const f32x4 lo, const f32x4 hi, f32x16& g0, f32x16& g1, f32x16& g2) {
        if constexpr (X3) {
            f32x4 xh, xl;
            split2h(lo, hi, xh, xl);
            g0 = mfma_h(wv[0], xl, g0); g1 = mfma_h(wv[2], xl, g1); g2 = mfma_h(wv[4], xl, g2);
            g0 = mfma_h(wv[1], xh, g0); g1 = mfma_h(wv[3], xh, g1); g2 = mfma_h(wv[5], xh, g2);
            g0 = mfma_h(wv[0], xh, g0); g1 = mfma_h(wv[2], xh, g1); g2 = mfma_h(wv[4], xh, g2);
        }
    };
    auto hraw = [&](int w, int q, f32x4& lo, f32x4& hi) {        // h chunk q of group w: the two fp32 halves
        const int cq = q >> 1, e0 = 8 * (q & 1);
        lo[0] = h[w][cq][e0 + 0]; lo[1] = h[w][cq][e0 + 1]; lo[2] = h[w][cq][e0 + 2]; lo[3] = h[w][cq][e0 + 3];
        hi[0] = h[w][cq][e0 + 4]; hi[1] = h[w][cq][e0 + 5]; hi[2] = h[w][cq][e0 + 6]; hi[3] = h[w][cq][e0 + 7];
    };
    constexpr int NXR = XR ? QXC : R;               // input operand registers: the whole step (XR) or the ring
    f32x4 wr[R][WN], xr[NXR][MW][XW];
#pragma unroll
    for (int st = 0; st < R; ++st) wload(wr[st]);
#pragma unroll
    for (int st = 0; st < NXR; ++st)
#pragma unroll
        for (int w = 0; w < MW; ++w) loadxq(xr[st][w], w, 0, st);

    for (int t = 0; t < T; ++t) {
        auto tile_body = [&](const int c, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            f32x16 ar[MW], az[MW], anx[MW], anh[MW];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * c + 8 * m + 4 * g;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + col);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + a.Hp + col);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias + 2 * a.Hp + col);
                const f32x4 b3 = *reinterpret_cast<const f32x4*>(a.bias + 3 * a.Hp + col);
#pragma unroll
                for (int w = 0; w < MW; ++w)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        ar[w][4 * m + s4] = X3 ? b0[s4] * wS : b0[s4];
                        az[w][4 * m + s4] = X3 ? b1[s4] * wS : b1[s4];
                        anx[w][4 * m + s4] = X3 ? b2[s4] * wS : b2[s4];
                        anh[w][4 * m + s4] = X3 ? b3[s4] * wS : b3[s4];
                    }
            }
            // ---- input part: W_i{r,z,n} x_t.  sched_barrier pins "MFMAs of chunk j, then the loads that
            // refill its ring stage": left alone the scheduler sinks all loads of an iteration below its
            // MFMAs and the next iteration waits for them.
            f32x4 xs0[3], xs1[3];                 // PIPE: split operands of the current / next item
            if constexpr (PIPE && XMODE == 1) {
                f32x4 lo, hi;
                split2h(xr[0][0][0], xr[0][0][XW - 1], xs0[0], xs0[1]);          // entries of the encoder's h_end: two fp16 pieces
                stage(T2{}, N2{}, wr[0], xs0, ar[0], az[0], anx[0], xr[0][MW - 1][0], xr[0][MW - 1][XW - 1], xs1);
                hraw(0, 0, lo, hi);
                stage(T2{}, N2{}, wr[0], xs1, ar[MW - 1], az[MW - 1], anx[MW - 1], lo, hi, xs0);
                wload(wr[0]);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (PIPE) {
                if (qb > 0) split3(xr[0][0][0], xr[0][0][XW - 1], xs0[0], xs0[1], xs0[2]);
                else split2h(xr[0][0][0], xr[0][0][XW - 1], xs0[0], xs0[1]);
                int q0 = 0;
                for (; q0 < qb; q0 += R) {             // three-piece chunks (qb is a multiple of R)
#pragma unroll
                    for (int st = 0; st < R; ++st) {
                        stage(T3{}, N3{}, wr[st], xs0, ar[0], az[0], anx[0], xr[st][MW - 1][0], xr[st][MW - 1][XW - 1], xs1);
                        if (st == R - 1 && q0 + R >= qb)   // the next chunk is the first two-piece one
                            stage(T3{}, N2{}, wr[st], xs1, ar[MW - 1], az[MW - 1], anx[MW - 1], xr[(st + 1) % R][0][0], xr[(st + 1) % R][0][XW - 1], xs0);
                        else
                            stage(T3{}, N3{}, wr[st], xs1, ar[MW - 1], az[MW - 1], anx[MW - 1], xr[(st + 1) % R][0][0], xr[(st + 1) % R][0][XW - 1], xs0);
                        wload(wr[st]);
#pragma unroll
                        for (int w = 0; w < MW; ++w) loadxq(xr[st][w], w, t, q0 + st + R);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                for (; q0 < Qxp; q0 += R) {            // two-piece chunks
#pragma unroll
                    for (int st = 0; st < R; ++st) {
                        stage(T2{}, N2{}, wr[st], xs0, ar[0], az[0], anx[0], xr[st][MW - 1][0], xr[st][MW - 1][XW - 1], xs1);
                        stage(T2{}, N2{}, wr[st], xs1, ar[MW - 1], az[MW - 1], anx[MW - 1], xr[(st + 1) % R][0][0], xr[(st + 1) % R][0][XW - 1], xs0);
                        wload(wr[st]);
#pragma unroll
                        for (int w = 0; w < MW; ++w) loadxq(xr[st][w], w, t, q0 + st + R);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                {   // the last stage split a pad chunk: the first recurrent item instead
                    f32x4 lo, hi;
                    hraw(0, 0, lo, hi);
                    split2h(lo, hi, xs0[0], xs0[1]);
                }
            } else if (XMODE == 1) {
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    if constexpr (X3) gates_h(wr[0], xr[0][w][0], xr[0][w][XW - 1], ar[w], az[w], anx[w]);
                    else gates3(wr[0], xr[0][w], ar[w], az[w], anx[w]);
                }
                wload(wr[0]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (XR) {
                const int tnx = t + 1 < T ? t + 1 : t;
#pragma unroll
                for (int q = 0; q < NXR; ++q) {
                    constexpr int dummy = 0; (void)dummy;
                    const int st = q % R;
#pragma unroll
                    for (int w = 0; w < MW; ++w) gates3(wr[st], xr[q][w], ar[w], az[w], anx[w]);
                    wload(wr[st]);
                    if (LAST) {                       // compile-time: only the last tile's code carries these loads
#pragma unroll
                        for (int w = 0; w < MW; ++w) loadxq(xr[q][w], w, tnx, q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int q0 = 0; q0 < Qxp; q0 += R) {
#pragma unroll
                    for (int st = 0; st < R; ++st) {
#pragma unroll
                        for (int w = 0; w < MW; ++w) {
                            if (X3 && q0 + st >= qb) gates_h(wr[st], xr[st][w][0], xr[st][w][XW - 1], ar[w], az[w], anx[w]);
                            else gates3(wr[st], xr[st][w], ar[w], az[w], anx[w]);
                        }
                        wload(wr[st]);
#pragma unroll
                        for (int w = 0; w < MW; ++w) loadxq(xr[st][w], w, t, q0 + st + R);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- recurrent part: W_h{r,z,n} h_{t-1}  (h_0 = 0 contributes nothing at t = 0; kept so the
            // weight stream stays continuous).  Ring stage of h chunk q is static: (x chunks + q) % 3.
            if constexpr (PIPE) {
#pragma unroll
                for (int q = 0; q < Qhe; ++q) {
                    constexpr int X0 = (XMODE == 1) ? 1 : 0;
                    const int st = (X0 + q) % R;
                    f32x4 lo, hi;
                    hraw(MW - 1, q, lo, hi);
                    stage(T2{}, N2{}, wr[st], xs0, ar[0], az[0], anh[0], lo, hi, xs1);
                    if (q + 1 < Qhe) {
                        hraw(0, q + 1, lo, hi);
                        stage(T2{}, N2{}, wr[st], xs1, ar[MW - 1], az[MW - 1], anh[MW - 1], lo, hi, xs0);
                    } else {
                        stage(T2{}, N0{}, wr[st], xs1, ar[MW - 1], az[MW - 1], anh[MW - 1], lo, hi, xs0);
                    }
                    wload(wr[st]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int q = 0; q < Qhe; ++q) {
                constexpr int X0 = (XMODE == 1) ? 1 : 0;
                const int st = (X0 + q) % R;
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    f32x4 hv[XW];
                    if (BF) {
                        const int cq = q >> 1, e0 = 8 * (q & 1);
                        f32x4 lo, hi;
                        lo[0] = h[w][cq][e0 + 0]; lo[1] = h[w][cq][e0 + 1]; lo[2] = h[w][cq][e0 + 2]; lo[3] = h[w][cq][e0 + 3];
                        hi[0] = h[w][cq][e0 + 4]; hi[1] = h[w][cq][e0 + 5]; hi[2] = h[w][cq][e0 + 6]; hi[3] = h[w][cq][e0 + 7];
                        if (X3) { gates_h(wr[st], lo, hi, ar[w], az[w], anh[w]); continue; }
                        hv[0] = cvt8(lo, hi);
                    } else {
                        const int cq = q >> 2, m = q & 3;
                        hv[0][0] = h[w][cq][4 * m + 0]; hv[0][1] = h[w][cq][4 * m + 1];
                        hv[0][2] = h[w][cq][4 * m + 2]; hv[0][3] = h[w][cq][4 * m + 3];
                    }
                    gates3(wr[st], hv, ar[w], az[w], anh[w]);
                }
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bring the ring back to phase 0 for the next tile: a compile-time register renaming (rotate by ROT stages)
            if (ROT != 0) {
                f32x4 tmp[R][WN];
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int u = 0; u < WN; ++u) tmp[st][u] = wr[(st + ROT) % R][u];
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int u = 0; u < WN; ++u) wr[st][u] = tmp[st][u];
            }
            // x chunks 0..2 of the next tile / step: their latency hides under the gate math
            if (!XR) {
                const int tn = LAST ? (t + 1 < T ? t + 1 : t) : t;
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int w = 0; w < MW; ++w) loadxq(xr[st][w], w, tn, st);
            }
            // ---- gates.  h_old for this tile comes back from LDS (written at the end of step t-1);
            // every lane reads and writes only its own slots -> no cross-lane hazard
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                if constexpr (X3 || BF) {
                    // two values per instruction where the ISA has a packed form (v_pk_mul / add / fma_f32): the wave's VALU
                    // issue is what bounds this build.  Same arithmetic as gate_sigmoid / gate_tanh with the power-of-two
                    // weight scale folded into the exponent constants, without their clamps (exp2 saturates to 0 / inf and
                    // 1 / (1 + inf) = 0: the limits come out right without them)
                    const float cs = -1.4426950408889634f * wInvS, ct = 2.8853900817779268f * wInvS;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 hold = {(t > 0) ? hn_s[w][c][r][lane] : 0.f, (t > 0) ? hn_s[w][c][r + 1][lane] : 0.f};
                        const f32x2 av = {ar[w][r], ar[w][r + 1]}, zv = {az[w][r], az[w][r + 1]};
                        const f32x2 xv = {anx[w][r], anx[w][r + 1]}, hv = {anh[w][r], anh[w][r + 1]};
                        const f32x2 ea = av * cs, ez = zv * cs;
                        const f32x2 da = f32x2{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])} + 1.0f;
                        const f32x2 dz = f32x2{__builtin_amdgcn_exp2f(ez[0]), __builtin_amdgcn_exp2f(ez[1])} + 1.0f;
                        const f32x2 rg = {__builtin_amdgcn_rcpf(da[0]), __builtin_amdgcn_rcpf(da[1])};
                        const f32x2 zg = {__builtin_amdgcn_rcpf(dz[0]), __builtin_amdgcn_rcpf(dz[1])};
                        const f32x2 en = (xv + rg * hv) * ct;
                        const f32x2 dn = f32x2{__builtin_amdgcn_exp2f(en[0]), __builtin_amdgcn_exp2f(en[1])} + 1.0f;
                        const f32x2 ng = f32x2{__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])} * -2.0f + 1.0f;
                        const f32x2 hn = zg * (hold - ng) + ng;
                        ar[w][r] = hn[0]; ar[w][r + 1] = hn[1];
                    }
                } else
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hold = (t > 0) ? hn_s[w][c][r][lane] : 0.f;
                    const float rg = gate_sigmoid(X3 ? ar[w][r] * wInvS : ar[w][r]);
                    const float zg = gate_sigmoid(X3 ? az[w][r] * wInvS : az[w][r]);
                    const float ng = gate_tanh(X3 ? (anx[w][r] + rg * anh[w][r]) * wInvS : anx[w][r] + rg * anh[w][r]);
                    ar[w][r] = __builtin_fmaf(zg, hold - ng, ng);          // (1 - z) n + z h
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) hn_s[w][c][r][lane] = ar[w][r];
            }
        };
        for (int c = 0; c + 1 < NCG; ++c) tile_body(c, std::false_type{});
        tile_body(NCG - 1, std::true_type{});
        __syncthreads();
#pragma unroll
        for (int w = 0; w < MW; ++w)
#pragma unroll
            for (int c = 0; c < NCG; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) h[w][c][r] = hn_s[w][c][r][lane];

#pragma unroll
        for (int w = 0; w < MW; ++w) {
            if (a.Seq && win[w] < a.B) {
                float* sp = a.Seq + (win[w] * T + t) * a.ldseq;
#pragma unroll
                for (int c = 0; c < NCG; ++c)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 v;
                        v[0] = h[w][c][4 * m + 0]; v[1] = h[w][c][4 * m + 1]; v[2] = h[w][c][4 * m + 2]; v[3] = h[w][c][4 * m + 3];
                        *reinterpret_cast<f32x4*>(sp + 32 * c + 8 * m + 4 * g) = v;
                    }
            }
            if (FC && (a.Yfc != nullptr || t == T - 1) && a.out_dim <= 4) {
                // few outputs (target dims of MSL / SMAP: 1): a 32-output MFMA tile per step would cost 4*Qhe
                // matrix instructions for one useful column.  Dot products on the VALU instead: lane (i, g)
                // covers its 16 features of every tile, the two halves meet through one cross-lane add.
                const f32x4* __restrict__ wf = a.Wfc;         // tile 0: [Qh][64 lanes][4], lane (o, g) = W[o][8q + 4g + s]
                float* yp = (a.Yfc && win[w] < a.B) ? a.Yfc + (win[w] * T + t) * (long)a.out_dim : nullptr;
                float* yl = (a.Ylast && t == T - 1 && win[w] < a.B) ? a.Ylast + win[w] * (long)a.out_dim : nullptr;
                for (int o = 0; o < a.out_dim; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < Qf; ++q) {
                        const int cq = q >> 2, m = q & 3;
                        const f32x4 wv = wf[q * 64 + o + 32 * g];
                        acc += wv[0] * h[w][cq][4 * m + 0] + wv[1] * h[w][cq][4 * m + 1] + wv[2] * h[w][cq][4 * m + 2] + wv[3] * h[w][cq][4 * m + 3];
                    }
                    acc += __shfl_xor(acc, 32);
                    const float y = acc + a.bfc[o];
                    if (g == 0) {
                        if (yp) yp[o] = y;
                        if (yl) yl[o] = y;
                    }
                }
            } else if (FC && (a.Yfc != nullptr || t == T - 1)) {
                for (int n = 0; n < a.NTfc; ++n) {
                    f32x16 y;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bfc + 32 * n + 8 * m + 4 * g);
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) y[4 * m + s4] = bv[s4];
                    }
                    const f32x4* __restrict__ wp = a.Wfc + ((long)n * (4 * NCG)) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < Qf; ++q) {
                        const int cq = q >> 2, m = q & 3;
                        f32x4 hv;
                        hv[0] = h[w][cq][4 * m + 0]; hv[1] = h[w][cq][4 * m + 1];
                        hv[2] = h[w][cq][4 * m + 2]; hv[3] = h[w][cq][4 * m + 3];
                        y = mfma4(wp[q * 64], hv, y);
                    }
                    if (win[w] < a.B) {
                        float* yp = a.Yfc ? a.Yfc + (win[w] * T + t) * (long)a.out_dim : nullptr;
                        float* yl = (a.Ylast && t == T - 1) ? a.Ylast + win[w] * (long)a.out_dim : nullptr;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int o = 32 * n + (r & 3) + 8 * (r >> 2) + 4 * g;
                            if (o < a.out_dim) {
                                if (yp) yp[o] = y[r];
                                if (yl) yl[o] = y[r];
                            }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < MW; ++w) {
        if (a.Hend && win[w] < a.B) {
            float* hp = a.Hend + win[w] * a.ldhe;
            if (a.ldhe >= a.Hp) {        // internal buffer: all Hp columns (the padding lanes of h are exact zeros)
#pragma unroll
                for (int c = 0; c < NCG; ++c)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 v;
                        v[0] = h[w][c][4 * m + 0]; v[1] = h[w][c][4 * m + 1]; v[2] = h[w][c][4 * m + 2]; v[3] = h[w][c][4 * m + 3];
                        *reinterpret_cast<f32x4*>(hp + 32 * c + 8 * m + 4 * g) = v;
                    }
            } else {
#pragma unroll
                for (int c = 0; c < NCG; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (j < a.H) hp[j] = h[w][c][r];
                    }
            }
        }
    }
}

template <int NCG, int XMODE, int MW, bool BF, bool X3 = false>
inline int launch_gru_mode(const GruArgs& a, bool fc, int drop, hipStream_t s) {
    constexpr int WPB = MW == 2 ? 4 : 1;
    const unsigned grid = (unsigned)((a.B + 32 * MW * WPB - 1) / (32 * MW * WPB));
    const size_t lds = (size_t)WPB * MW * NCG * 1024 * sizeof(float);
#define GRU_LAUNCH(FCV, DR)                                                                                            \
    {                                                                                                                  \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru<NCG, XMODE, FCV, DR, MW, BF, 0, X3>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e_ != hipSuccess) return (int)e_;                                                                      \
        }                                                                                                              \
        hipLaunchKernelGGL((k_gru<NCG, XMODE, FCV, DR, MW, BF, 0, X3>), dim3(grid), dim3(64 * WPB), lds, s, a);        \
    }
#define GRU_LAUNCH_XR(DR, QX)                                                                                          \
    {                                                                                                                  \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru<NCG, 0, false, DR, 2, true, QX>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e_ != hipSuccess) return (int)e_;                                                                      \
        }                                                                                                              \
        hipLaunchKernelGGL((k_gru<NCG, 0, false, DR, 2, true, QX>), dim3(grid), dim3(64 * WPB), lds, s, a);            \
    }
    if constexpr (BF && !X3 && XMODE == 0 && MW == 2 && NCG <= 5) {
        if (!fc && (a.Qxp == 6 || a.Qxp == 12)) {
            if (a.Qxp == 6) { if (drop == 0) GRU_LAUNCH_XR(0, 6) else GRU_LAUNCH_XR(1, 6) }
            else { if (drop == 0) GRU_LAUNCH_XR(0, 12) else GRU_LAUNCH_XR(1, 12) }
            LAUNCH_CHECK();
            return 0;
        }
    }
    if (!fc && drop == 0) GRU_LAUNCH(false, 0)
    else if (!fc) GRU_LAUNCH(false, 1)
    else if (drop == 0) GRU_LAUNCH(true, 0)
    else GRU_LAUNCH(true, 1)
#undef GRU_LAUNCH
#undef GRU_LAUNCH_XR
    LAUNCH_CHECK();
    return 0;
}

template <int NCG, bool BF, bool X3 = false>
inline int launch_gru_ncg(const GruArgs& a, int xmode, bool fc, bool two, hipStream_t s) {
    // trailing recurrent chunks that are pure padding: skip one when H <= 8*(4*NCG - 1)  (bf16: 16*(2*NCG - 1))
    const int drop = BF ? ((a.H <= 16 * (2 * NCG - 1)) ? 1 : 0) : ((a.H <= 8 * (4 * NCG - 1)) ? 1 : 0);
    if constexpr (NCG <= 5) {           // two 32-window groups per wave: 8 KB of LDS per group and tile, 4 waves per CU
        if (two) {
            if (xmode == 0) return launch_gru_mode<NCG, 0, 2, BF, X3>(a, fc, drop, s);
            if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 2, BF, X3>(a, fc, drop, s);
            return launch_gru_mode<NCG, 2, 2, BF, X3>(a, fc, drop, s);
        }
    }
    if (xmode == 0) return launch_gru_mode<NCG, 0, 1, BF, X3>(a, fc, drop, s);
    if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 1, BF, X3>(a, fc, drop, s);
    return launch_gru_mode<NCG, 2, 1, BF, X3>(a, fc, drop, s);
}


// split-bf16 build, in two translation units (hidden sizes up to 128 / up to 256)
template <bool UPPER>
inline int launch_gru_big_split(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) {
    if constexpr (!UPPER) {
        switch (ncg) {
            case 1: return launch_gru_ncg<1, true, true>(a, xmode, fc, two, s);
            case 2: return launch_gru_ncg<2, true, true>(a, xmode, fc, two, s);
            case 3: return launch_gru_ncg<3, true, true>(a, xmode, fc, two, s);
            case 4: return launch_gru_ncg<4, true, true>(a, xmode, fc, two, s);
            default: return -2;
        }
    } else {
        switch (ncg) {
            case 5: return launch_gru_ncg<5, true, true>(a, xmode, fc, two, s);
            case 6: return launch_gru_ncg<6, true, true>(a, xmode, fc, two, s);
            case 7: return launch_gru_ncg<7, true, true>(a, xmode, fc, two, s);
            case 8: return launch_gru_ncg<8, true, true>(a, xmode, fc, two, s);
            default: return -2;
        }
    }
}

template <bool BF>
inline int launch_gru_big_t(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s) {
    switch (ncg) {
        case 1: return launch_gru_ncg<1, BF>(a, xmode, fc, two, s);
        case 2: return launch_gru_ncg<2, BF>(a, xmode, fc, two, s);
        case 3: return launch_gru_ncg<3, BF>(a, xmode, fc, two, s);
        case 4: return launch_gru_ncg<4, BF>(a, xmode, fc, two, s);
        case 5: return launch_gru_ncg<5, BF>(a, xmode, fc, two, s);
        case 6: return launch_gru_ncg<6, BF>(a, xmode, fc, two, s);
        case 7: return launch_gru_ncg<7, BF>(a, xmode, fc, two, s);
        case 8: return launch_gru_ncg<8, BF>(a, xmode, fc, two, s);
        default: return -2;
    }
}

}  // namespace mtadgat
