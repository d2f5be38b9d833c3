// k_gru / k_gru_split: GRU layer and reconstruction decoder + launchers
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
template <int NCG, int XMODE, bool FC, int DROP, int MW, bool BF = false, int QXC = 0>
__global__ __launch_bounds__((MW == 2 ? 256 : 64), ((NCG <= 6 && MW == 1) ? 2 : 1)) void k_gru(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hn_dyn[];
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
    constexpr int R = 3;
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
    const f32x4* __restrict__ pwx = a.Wx;          // next input-part chunk to fetch
    const f32x4* __restrict__ pwh = a.Wh;          // next recurrent-part chunk to fetch
    auto wload = [&](f32x4 (&dst)[3]) {
        const bool isx = ps < Qxp;
        const f32x4* __restrict__ p = (isx ? pwx : pwh) + lane;
        dst[0] = p[0]; dst[1] = p[64]; dst[2] = p[128];
        pwx += isx ? 192 : 0;
        pwh += isx ? 0 : 192;
        const bool ws = (ps + 1 == S);             // end of this tile's chunk sequence
        ps = ws ? 0 : ps + 1;
        pwh += ws ? (a.whs - Qhe) * 192 : 0;       // skip the unused all-padding chunks of the tile
        const bool wc = ws && (pc + 1 == NCG);     // end of the step
        pc = ws ? (wc ? 0 : pc + 1) : pc;
        pt = wc ? pt + 1 : pt;
        pwh = wc ? a.Wh : pwh;
        // input-part weights: per step for the decoder ([t][c][Qxp], contiguous), shared by all steps otherwise
        pwx = wc ? ((XMODE == 0 || pt >= T) ? a.Wx : pwx) : pwx;
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
    auto loadxq = [&](int w, int t, int q) -> f32x4 {
        if (!BF) return loadx_t(w, t, q);
        if (XMODE == 1) return cvt8(loadx_t(w, t, q), f32x4{0.f, 0.f, 0.f, 0.f});
        return cvt8(loadx_t(w, t, 2 * q), loadx_t(w, t, 2 * q + 1));
    };
    constexpr int NXR = XR ? QXC : R;               // input operand registers: the whole step (XR) or the ring
    f32x4 wr[R][3], xr[NXR][MW];
#pragma unroll
    for (int st = 0; st < R; ++st) wload(wr[st]);
#pragma unroll
    for (int st = 0; st < NXR; ++st)
#pragma unroll
        for (int w = 0; w < MW; ++w) xr[st][w] = loadxq(w, 0, st);

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
                        ar[w][4 * m + s4] = b0[s4];
                        az[w][4 * m + s4] = b1[s4];
                        anx[w][4 * m + s4] = b2[s4];
                        anh[w][4 * m + s4] = b3[s4];
                    }
            }
            // ---- input part: W_i{r,z,n} x_t.  sched_barrier pins "MFMAs of chunk j, then the loads that
            // refill its ring stage": left alone the scheduler sinks all loads of an iteration below its
            // MFMAs and the next iteration waits for them.
            if (XMODE == 1) {
#pragma unroll
                for (int w = 0; w < MW; ++w) mfma_x3<BF>(wr[0], xr[0][w], ar[w], az[w], anx[w]);
                wload(wr[0]);
                __builtin_amdgcn_sched_barrier(0);
            } else if (XR) {
                const int tnx = t + 1 < T ? t + 1 : t;
#pragma unroll
                for (int q = 0; q < NXR; ++q) {
                    constexpr int dummy = 0; (void)dummy;
                    const int st = q % R;
#pragma unroll
                    for (int w = 0; w < MW; ++w) mfma_x3<BF>(wr[st], xr[q][w], ar[w], az[w], anx[w]);
                    wload(wr[st]);
                    if (LAST) {                       // compile-time: only the last tile's code carries these loads
#pragma unroll
                        for (int w = 0; w < MW; ++w) xr[q][w] = loadxq(w, tnx, q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (int q0 = 0; q0 < Qxp; q0 += R) {
#pragma unroll
                    for (int st = 0; st < R; ++st) {
#pragma unroll
                        for (int w = 0; w < MW; ++w) mfma_x3<BF>(wr[st], xr[st][w], ar[w], az[w], anx[w]);
                        wload(wr[st]);
#pragma unroll
                        for (int w = 0; w < MW; ++w) xr[st][w] = loadxq(w, t, q0 + st + R);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- recurrent part: W_h{r,z,n} h_{t-1}  (h_0 = 0 contributes nothing at t = 0; kept so the
            // weight stream stays continuous).  Ring stage of h chunk q is static: (x chunks + q) % 3.
#pragma unroll
            for (int q = 0; q < Qhe; ++q) {
                constexpr int X0 = (XMODE == 1) ? 1 : 0;
                const int st = (X0 + q) % R;
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    f32x4 hv;
                    if (BF) {
                        const int cq = q >> 1, e0 = 8 * (q & 1);
                        f32x4 lo, hi;
                        lo[0] = h[w][cq][e0 + 0]; lo[1] = h[w][cq][e0 + 1]; lo[2] = h[w][cq][e0 + 2]; lo[3] = h[w][cq][e0 + 3];
                        hi[0] = h[w][cq][e0 + 4]; hi[1] = h[w][cq][e0 + 5]; hi[2] = h[w][cq][e0 + 6]; hi[3] = h[w][cq][e0 + 7];
                        hv = cvt8(lo, hi);
                    } else {
                        const int cq = q >> 2, m = q & 3;
                        hv[0] = h[w][cq][4 * m + 0]; hv[1] = h[w][cq][4 * m + 1];
                        hv[2] = h[w][cq][4 * m + 2]; hv[3] = h[w][cq][4 * m + 3];
                    }
                    mfma_x3<BF>(wr[st], hv, ar[w], az[w], anh[w]);
                }
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bring the ring back to phase 0 for the next tile: a compile-time register renaming (rotate by ROT stages)
            if (ROT != 0) {
                f32x4 tmp[R][3];
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int u = 0; u < 3; ++u) tmp[st][u] = wr[(st + ROT) % R][u];
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int u = 0; u < 3; ++u) wr[st][u] = tmp[st][u];
            }
            // x chunks 0..2 of the next tile / step: their latency hides under the gate math
            if (!XR) {
                const int tn = LAST ? (t + 1 < T ? t + 1 : t) : t;
#pragma unroll
                for (int st = 0; st < R; ++st)
#pragma unroll
                    for (int w = 0; w < MW; ++w) xr[st][w] = loadxq(w, tn, st);
            }
            // ---- gates.  h_old for this tile comes back from LDS (written at the end of step t-1);
            // every lane reads and writes only its own slots -> no cross-lane hazard
#pragma unroll
            for (int w = 0; w < MW; ++w) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hold = (t > 0) ? hn_s[w][c][r][lane] : 0.f;
                    const float rg = gate_sigmoid(ar[w][r]);
                    const float zg = gate_sigmoid(az[w][r]);
                    const float ng = gate_tanh(anx[w][r] + rg * anh[w][r]);
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

// ---------------------------------------------------------------------------
// GRU, hidden-tile split: a workgroup owns 32 windows, wave c the 32 hidden units of tile c (all three
// gates).  Same packed weights, gate algebra and outputs as k_gru; what changes is where h lives: each
// wave keeps only its own tile in registers and publishes it in LDS once per step (F-layout, so a chunk of
// h_{t-1} is one 16-byte LDS read per lane).  A small batch then spreads over NCG times as many SIMDs (a
// 256-window batch occupies 8 waves in k_gru).
//   per step:  [MFMA: x chunks, then h chunks read from hs]  barrier B
//              [gates; own tile -> hs; per-step Linear partial -> ps]  barrier A
//              [wave t % NCG: reduce the Linear partials, store y_t]
// Two barriers per step keep hs / ps single-buffered: nobody overwrites h_{t-1} before all waves have
// consumed it (B), nobody reads h_t / the partials before they are complete (A).
// The chunk loops have the static shape of k_gru's (3-stage weight ring, unconditional loads,
// sched_barrier after each refill): the h part is padded with zero-weight chunks so that a step is a
// whole number of ring turns for any hidden size, which keeps NCG and H run-time values.
// ---------------------------------------------------------------------------
// SAVE (training): the gate activations r, z, n and q = W_hn h + b_hn of every step go to a.Gates for the backward.
template <int XMODE, bool FC, bool SAVE = false, bool BF = false>
__global__ __launch_bounds__(512, (XMODE == 3 ? 2 : 3)) void k_gru_split(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCG = blockDim.x >> 6;
    const int i = lane & 31, g = lane >> 5;
    const long win = (long)blockIdx.x * 32 + i;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Qx = a.Qx;
    const int Qxp = (XMODE == 1) ? 1 : (XMODE == 3 ? 0 : a.Qxp);      // XMODE 0/2: a multiple of 3; 3: no input chunks
    const int Qh = 4 * NCG;                        // fp32 chunks of h (per-step Linear tiles)
    const int Qhe = BF ? (a.H + 15) >> 4 : (a.H + 7) >> 3;     // recurrent chunks as needed (bf16 build: 16 features each)
    const int S3 = (Qxp + Qhe + 2) / 3 * 3;        // chunks per step: whole ring turns
    const int NH = S3 - Qxp;                       // h chunks per step incl. zero-weight padding
    f32x4* __restrict__ hs = reinterpret_cast<f32x4*>(gsm);                // [NCG][4][64] float4: h_{t-1}, F-layout
    float* __restrict__ ps = gsm + NCG * 1024;                              // [NCG][out_dim][32] Linear partials

    // ---- weight stream of this tile: [x chunks 0..Qxp) [h chunks 0..NH)] per step through a 3-stage ring
    // Running wave-uniform pointers, advanced by adds and scalar selects only (a branch inside the chunk
    // loops makes the compiler drain the ring with s_waitcnt vmcnt(0)); the h stream of a tile ends in
    // two all-zero chunks, so the padded chunks need no special case.
    const f32x4* __restrict__ whc = a.Wh + (long)c * a.whs * 192;
    const f32x4* __restrict__ wx0 = a.Wx + (long)c * Qxp * 192;
    const long wxskip = (XMODE == 0) ? 0 : (long)(NCG - 1) * Qxp * 192;    // decoder input weights are [t][c][Qxp]
    int ps_ = 0, pt = 0;
    const f32x4* __restrict__ pwx = wx0;
    const f32x4* __restrict__ pwh = whc;
    auto wload = [&](f32x4 (&dst)[3]) {
        const bool isx = ps_ < Qxp;
        const f32x4* __restrict__ p = (isx ? pwx : pwh) + lane;
        dst[0] = p[0]; dst[1] = p[64]; dst[2] = p[128];
        pwx += isx ? 192 : 0;
        pwh += isx ? 0 : 192;
        const bool ws = ps_ + 1 == S3;             // end of the step
        ps_ = ws ? 0 : ps_ + 1;
        pt = ws ? pt + 1 : pt;
        pwh = ws ? whc : pwh;
        const f32x4* __restrict__ nx = (XMODE == 0 || pt >= T) ? wx0 : pwx + wxskip;
        pwx = ws ? nx : pwx;
    };
    const float* __restrict__ xbase = (XMODE == 0 || XMODE == 3) ? a.X + winc * T * a.ldx + 4 * g : a.X + winc * a.ldx;
    auto loadx_t = [&](int t, int q) -> f32x4 {
        const int qq = q < Qx ? q : Qx - 1;       // padded chunks re-read the last real one (their weights are zero)
        if (XMODE == 0 || XMODE == 3) return *reinterpret_cast<const f32x4*>(xbase + (long)t * a.ldx + 8 * qq);
        const int k0 = a.m0[t] + 8 * qq + 4 * g, kmax = (int)a.ldx - 1;     // stay inside the (zero padded) row
        f32x4 v;
        v[0] = xbase[min(k0, kmax)]; v[1] = xbase[min(k0 + 1, kmax)];
        v[2] = xbase[min(k0 + 2, kmax)]; v[3] = xbase[min(k0 + 3, kmax)];
        return v;
    };
    auto hread = [&](int q) -> f32x4 { return hs[(q < Qhe ? q : Qhe - 1) * 64 + lane]; };   // padding: any finite chunk
    // XMODE 3: the input products W_i{r,z,n} x_t + b of all steps were computed beforehand by one throughput GEMM
    // (k_rowgemm over the b*T rows): X = (B*T, 3*Hp) [r | z | n]; a step starts from this tile's 3 x 16 values.
    // At small batches the recurrence is a latency chain -- the input chunks are half of its MFMAs.
    f32x4 xp[3][4];
    auto loadxp = [&](int t) {
        const float* __restrict__ p = a.X + (winc * T + t) * a.ldx + 32 * c + 4 * g;
#pragma unroll
        for (int b3 = 0; b3 < 3; ++b3)
#pragma unroll
            for (int m = 0; m < 4; ++m) xp[b3][m] = *reinterpret_cast<const f32x4*>(p + b3 * a.Hp + 8 * m);
    };

    f32x16 hown;                                   // this wave's tile of h
#pragma unroll
    for (int r = 0; r < 16; ++r) hown[r] = 0.f;
    // hs: h_{t-1} as MFMA B operands -- fp32 build [NCG][4][64] float4 (F-layout chunks), bf16 build [NCG][2][64]
    // 16-byte containers of 8 bf16 (converted once by the publishing wave)
    constexpr int HSC = BF ? 2 : 4;
#pragma unroll
    for (int m = 0; m < HSC; ++m) hs[(c * HSC + m) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto loadxq = [&](int t, int q) -> f32x4 {
        if (!BF) return loadx_t(t, q);
        if (XMODE == 1) return cvt8(loadx_t(t, q), f32x4{0.f, 0.f, 0.f, 0.f});
        return cvt8(loadx_t(t, 2 * q), loadx_t(t, 2 * q + 1));
    };
    f32x4 wr[3][3], xr[3];
    wload(wr[0]); wload(wr[1]); wload(wr[2]);
    if (XMODE == 3) {
        loadxp(0);
    } else {
#pragma unroll
        for (int st = 0; st < 3; ++st) xr[st] = loadxq(0, st);
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        f32x16 ar, az, anx, anh;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int col = 32 * c + 8 * m + 4 * g;
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(a.bias + 3 * a.Hp + col);
            if (XMODE == 3) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    ar[4 * m + s4] = xp[0][m][s4];
                    az[4 * m + s4] = xp[1][m][s4];
                    anx[4 * m + s4] = xp[2][m][s4];
                    anh[4 * m + s4] = b3[s4];
                }
            } else {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + col);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + a.Hp + col);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias + 2 * a.Hp + col);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    ar[4 * m + s4] = b0[s4];
                    az[4 * m + s4] = b1[s4];
                    anx[4 * m + s4] = b2[s4];
                    anh[4 * m + s4] = b3[s4];
                }
            }
        }
        // chunk q of h_{t-1} is requested one chunk ahead of its MFMAs (LDS latency under the previous group)
        f32x4 hv = hread(0);
        int qh = 0;                                // next h chunk to consume
        if (XMODE == 1) {
            // first ring turn: the single x chunk, then h chunks 0 and 1
            mfma_x3<BF>(wr[0], xr[0], ar, az, anx);
            wload(wr[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 1; st < 3; ++st) {
                const f32x4 hn = hread(qh + 1);
                mfma_x3<BF>(wr[st], hv, ar, az, anh);
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
                hv = hn; ++qh;
            }
        } else {
            for (int q0 = 0; q0 < Qxp; q0 += 3) {
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    mfma_x3<BF>(wr[st], xr[st], ar, az, anx);
                    wload(wr[st]);
                    xr[st] = loadxq(t, q0 + st + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        auto hturn = [&]() {                       // one ring turn of h chunks qh .. qh + 2
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const f32x4 hn = hread(qh + st + 1);
                mfma_x3<BF>(wr[st], hv, ar, az, anh);
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
                hv = hn;
            }
            qh += 3;
        };
        // The first turn is peeled so that the loop header is only reached from code with the same
        // outstanding-load pattern (9 weight loads in ring order): otherwise the wait counts at the header
        // are the conservative join with the x loop's and the ring is drained every turn.
        if (XMODE != 1) hturn();                   // NH >= 3 there
        while (qh < NH) hturn();
        // x chunks 0..2 (XMODE 3: the pre-projected tile) of the next step: their latency hides under the gate math
        {
            const int tn = t + 1 < T ? t + 1 : t;
            if (XMODE == 3) {
                loadxp(tn);
            } else {
#pragma unroll
                for (int st = 0; st < 3; ++st) xr[st] = loadxq(tn, st);
            }
        }
        // ---- gates (reference GRULayer / RNNDecoder: torch.nn.GRU equations, r|z|n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rg = gate_sigmoid(ar[r]);
            const float zg = gate_sigmoid(az[r]);
            const float ng = gate_tanh(anx[r] + rg * anh[r]);
            hown[r] = __builtin_fmaf(zg, hown[r] - ng, ng);       // (1 - z) n + z h
            if (SAVE) { ar[r] = rg; az[r] = zg; anx[r] = ng; }    // anh[r] already is q
        }
        if (SAVE && win < a.B) {
            float* gp = a.Gates + (win * T + t) * (4L * a.Hp) + 32 * c + 4 * g;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v0, v1, v2, v3;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) { v0[s4] = ar[4 * m + s4]; v1[s4] = az[4 * m + s4]; v2[s4] = anx[4 * m + s4]; v3[s4] = anh[4 * m + s4]; }
                *reinterpret_cast<f32x4*>(gp + 8 * m) = v0;
                *reinterpret_cast<f32x4*>(gp + a.Hp + 8 * m) = v1;
                *reinterpret_cast<f32x4*>(gp + 2 * a.Hp + 8 * m) = v2;
                *reinterpret_cast<f32x4*>(gp + 3 * a.Hp + 8 * m) = v3;
            }
        }
        f32x4 hvv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            hvv[m][0] = hown[4 * m + 0]; hvv[m][1] = hown[4 * m + 1]; hvv[m][2] = hown[4 * m + 2]; hvv[m][3] = hown[4 * m + 3];
        }
        __syncthreads();                           // B: every wave is done reading h_{t-1}
        if (BF) {
            hs[(c * 2 + 0) * 64 + lane] = cvt8(hvv[0], hvv[1]);
            hs[(c * 2 + 1) * 64 + lane] = cvt8(hvv[2], hvv[3]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) hs[(c * 4 + m) * 64 + lane] = hvv[m];
        }
        if (a.Seq && win < a.B) {
            float* sp = a.Seq + (win * T + t) * a.ldseq + 32 * c + 4 * g;
#pragma unroll
            for (int m = 0; m < 4; ++m) *reinterpret_cast<f32x4*>(sp + 8 * m) = hvv[m];
        }
        const bool fc_now = FC && (a.Yfc != nullptr || t == T - 1);
        if (fc_now) {
            // this tile's share of y_t = W_fc h_t (+ b): the 4 chunks of h_t held in registers
            for (int n = 0; n < a.NTfc; ++n) {
                f32x16 y;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = 0.f;
                const f32x4* __restrict__ wp = a.Wfc + ((long)n * Qh + 4 * c) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 4; ++m) y = mfma4(wp[m * 64], hvv[m], y);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * n + (r & 3) + 8 * (r >> 2) + 4 * g;
                    if (o < a.out_dim) ps[(c * a.out_dim + o) * 32 + i] = y[r];
                }
            }
        }
        __syncthreads();                           // A: h_t and the partials are complete
        if (fc_now && c == t % NCG && win < a.B) {
            float* yp = a.Yfc ? a.Yfc + (win * T + t) * (long)a.out_dim : nullptr;
            float* yl = (a.Ylast && t == T - 1) ? a.Ylast + win * (long)a.out_dim : nullptr;
            for (int o = g; o < a.out_dim; o += 2) {
                float y = a.bfc[o];
                for (int cc = 0; cc < NCG; ++cc) y += ps[(cc * a.out_dim + o) * 32 + i];
                if (yp) yp[o] = y;
                if (yl) yl[o] = y;
            }
        }
    }
    if (a.Hend && win < a.B) {
        float* hp = a.Hend + win * a.ldhe;
        if (a.ldhe >= a.Hp) {        // internal buffer: all Hp columns (the padding lanes of h are exact zeros)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v;
                v[0] = hown[4 * m + 0]; v[1] = hown[4 * m + 1]; v[2] = hown[4 * m + 2]; v[3] = hown[4 * m + 3];
                *reinterpret_cast<f32x4*>(hp + 32 * c + 8 * m + 4 * g) = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (j < a.H) hp[j] = hown[r];
            }
        }
    }
}

template <int NCG, int XMODE, int MW, bool BF>
static int launch_gru_mode(const GruArgs& a, bool fc, int drop, hipStream_t s) {
    constexpr int WPB = MW == 2 ? 4 : 1;
    const unsigned grid = (unsigned)((a.B + 32 * MW * WPB - 1) / (32 * MW * WPB));
    const size_t lds = (size_t)WPB * MW * NCG * 1024 * sizeof(float);
#define GRU_LAUNCH(FCV, DR)                                                                                            \
    {                                                                                                                  \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru<NCG, XMODE, FCV, DR, MW, BF>),   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e_ != hipSuccess) return (int)e_;                                                                      \
        }                                                                                                              \
        hipLaunchKernelGGL((k_gru<NCG, XMODE, FCV, DR, MW, BF>), dim3(grid), dim3(64 * WPB), lds, s, a);               \
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
    if constexpr (BF && XMODE == 0 && MW == 2 && NCG <= 5) {
        static const bool stream_x = std::getenv("MTADGAT_GRU_STREAM_X") != nullptr;      // A/B switch
        if (!fc && !stream_x && (a.Qxp == 6 || a.Qxp == 12)) {
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

template <int NCG, bool BF>
static int launch_gru_ncg(const GruArgs& a, int xmode, bool fc, bool two, hipStream_t s) {
    // trailing recurrent chunks that are pure padding: skip one when H <= 8*(4*NCG - 1)  (bf16: 16*(2*NCG - 1))
    const int drop = BF ? ((a.H <= 16 * (2 * NCG - 1)) ? 1 : 0) : ((a.H <= 8 * (4 * NCG - 1)) ? 1 : 0);
    if constexpr (NCG <= 5) {           // two 32-window groups per wave: 8 KB of LDS per group and tile, 4 waves per CU
        if (two) {
            if (xmode == 0) return launch_gru_mode<NCG, 0, 2, BF>(a, fc, drop, s);
            if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 2, BF>(a, fc, drop, s);
            return launch_gru_mode<NCG, 2, 2, BF>(a, fc, drop, s);
        }
    }
    if (xmode == 0) return launch_gru_mode<NCG, 0, 1, BF>(a, fc, drop, s);
    if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 1, BF>(a, fc, drop, s);
    return launch_gru_mode<NCG, 2, 1, BF>(a, fc, drop, s);
}

static int launch_gru_split(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + 31) / 32);
    const size_t lds = ((size_t)ncg * 1024 + (fc ? (size_t)ncg * a.out_dim * 32 : 0)) * sizeof(float);
    if (lds > 64 * 1024) return -2;
    const int xm = xmode == 3 ? 3 : (xmode == 0 ? 0 : (a.Qxp == 1 ? 1 : 2));
    const bool save = a.Gates != nullptr;
#define SPLIT_CASE(XM, F)                                                                                        \
    if (xm == XM && fc == F) {                                                                                   \
        if (a.bf16 && !save) hipLaunchKernelGGL((k_gru_split<XM, F, false, true>), dim3(grid), dim3(64 * ncg), lds, s, a); \
        else if (a.bf16) hipLaunchKernelGGL((k_gru_split<XM, F, true, true>), dim3(grid), dim3(64 * ncg), lds, s, a);      \
        else if (save) hipLaunchKernelGGL((k_gru_split<XM, F, true>), dim3(grid), dim3(64 * ncg), lds, s, a);    \
        else hipLaunchKernelGGL((k_gru_split<XM, F, false>), dim3(grid), dim3(64 * ncg), lds, s, a);             \
    }
    SPLIT_CASE(0, false) SPLIT_CASE(0, true) SPLIT_CASE(1, false) SPLIT_CASE(1, true) SPLIT_CASE(2, false) SPLIT_CASE(2, true)
    SPLIT_CASE(3, false) SPLIT_CASE(3, true)
#undef SPLIT_CASE
    LAUNCH_CHECK();
    return 0;
}

// training forward: the hidden-tile-split kernel at every batch size (it is the one that keeps the gates)
int launch_gru_train(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (xmode == 0 && ((a.ldx & 3) != 0 || a.Qxp % 3 != 0)) return -2;
    if (xmode == 3 && (a.ldx & 3) != 0) return -2;
    if (xmode != 0 && xmode != 3 && a.Qxp != 1 && a.Qxp % 3 != 0) return -2;
    if (ncg < 1) return -2;
    return launch_gru_split(a, ncg, xmode, fc, s);
}

// windows up to which the hidden-tile-split kernel is the faster one (a 32-window group per CU x 2)
long gru_split_max_windows() { return 64L * cu_count(); }

int launch_gru(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (xmode == 3) {                // pre-projected input: only the split kernel takes it
        if ((a.ldx & 3) != 0 || ncg < 1) return -2;
        return launch_gru_split(a, ncg, xmode, fc, s);
    }
    if (xmode == 0 && ((a.ldx & 3) != 0 || a.Qxp % 3 != 0)) return -2;
    if (xmode != 0 && a.Qxp != 1 && a.Qxp % 3 != 0) return -2;
    // Small batches: spread the 32-window groups over ncg waves each (k_gru_split) -- k_gru needs ~2 groups
    // per SIMD to fill the machine and leaves it mostly idle below that.  Measured on MI355X (W=100, F=55,
    // H=150, GRU + decoder): 256 windows 12.0 -> 4.9 ms, 16 k windows 12.2 -> 9.9 ms, 32 k windows 12.2 vs 19.6
    // (the 5 waves of a group land 2/1/1/1 on the SIMDs, so the split form loses once the machine is full).
    const int n_cu = cu_count();
    {
        const long groups = (a.B + 31) / 32;
        const size_t lds = ((size_t)ncg * 1024 + (fc ? (size_t)ncg * a.out_dim * 32 : 0)) * sizeof(float);
        if (ncg >= 2 && groups <= 2L * n_cu && lds <= 64 * 1024) return launch_gru_split(a, ncg, xmode, fc, s);
    }
    // two groups per wave once that still gives every SIMD a wave
    const bool two = (a.B + 31) / 32 >= 8L * n_cu;
    if (a.bf16) {
        switch (ncg) {
            case 1: return launch_gru_ncg<1, true>(a, xmode, fc, two, s);
            case 2: return launch_gru_ncg<2, true>(a, xmode, fc, two, s);
            case 3: return launch_gru_ncg<3, true>(a, xmode, fc, two, s);
            case 4: return launch_gru_ncg<4, true>(a, xmode, fc, two, s);
            case 5: return launch_gru_ncg<5, true>(a, xmode, fc, two, s);
            case 6: return launch_gru_ncg<6, true>(a, xmode, fc, two, s);
            case 7: return launch_gru_ncg<7, true>(a, xmode, fc, two, s);
            case 8: return launch_gru_ncg<8, true>(a, xmode, fc, two, s);
            default: return -2;
        }
    }
    switch (ncg) {
        case 1: return launch_gru_ncg<1, false>(a, xmode, fc, two, s);
        case 2: return launch_gru_ncg<2, false>(a, xmode, fc, two, s);
        case 3: return launch_gru_ncg<3, false>(a, xmode, fc, two, s);
        case 4: return launch_gru_ncg<4, false>(a, xmode, fc, two, s);
        case 5: return launch_gru_ncg<5, false>(a, xmode, fc, two, s);
        case 6: return launch_gru_ncg<6, false>(a, xmode, fc, two, s);
        case 7: return launch_gru_ncg<7, false>(a, xmode, fc, two, s);
        case 8: return launch_gru_ncg<8, false>(a, xmode, fc, two, s);
        default: return -2;
    }
}

}  // namespace mtadgat
