// k_gru_split (hidden-tile-split recurrence for small batches and the training forward) + the GRU launchers;
// k_gru itself lives in mtadgat_gru_impl.h / mtadgat_gru_{f32,bf16}.hip
#include "mtadgat_device.h"

namespace mtadgat {

int launch_gru_big_f32(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s);
int launch_gru_big_bf16(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s);
int launch_gru_big_x3_hi(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s);
int launch_gru_big_x3_lo(const GruArgs& a, int ncg, int xmode, bool fc, bool two, hipStream_t s);

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
// X3H (with BF's 16-feature chunk geometry): split operands, two fp16 pieces per value -- weights from the [gate][piece]
// packs scaled by S (six words per chunk), x_t split where it is consumed, h_t split ONCE per step by the wave that
// publishes its tile (hs holds the two pieces of every chunk), nine v_mfma_f32_32x32x16_f16 per chunk on the 16-bit pipe:
// fp32-class results (mtadgat_device.h) at a fifth of the fp32 MFMA time.  The mid-size batches' recurrence: a
// 32-window wave of k_gru_cm is a latency chain (one round of it takes 4 ms however few windows it holds), five waves
// per 32 windows are five times the SIMDs.  Needs |x| < 2^15 (recorded by the convolution; otherwise the launch returns
// at once and the caller's three-piece kernel serves it).
template <int XMODE, bool FC, bool SAVE = false, bool BF = false, bool X3H = false>
__global__ __launch_bounds__(512, (X3H || XMODE == 3 ? 2 : 3)) void k_gru_split(const GruArgs a) {
    static_assert(!X3H || (BF && XMODE != 3), "split operands: 16-feature chunks, in-kernel input part");
    if (X3H && a.vmax != nullptr && !(__uint_as_float(*a.vmax) < 32768.f)) return;
    if (!X3H && a.skip_xh && a.vmax != nullptr && __uint_as_float(*a.vmax) < 32768.f) return;
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
    constexpr int NWD = X3H ? 6 : 3;               // 1-KiB words per chunk: [gate] or [gate][piece]
    constexpr int CST = 64 * NWD;                  // chunk stride in float4
    const f32x4* __restrict__ whc = a.Wh + (long)c * a.whs * CST;
    const f32x4* __restrict__ wx0 = (X3H ? a.Wxq : a.Wx) + (long)c * Qxp * CST;
    const long wxskip = (XMODE == 0) ? 0 : (long)(NCG - 1) * Qxp * CST;    // decoder input weights are [t][c][Qxp]
    int ps_ = 0, pt = 0;
    const f32x4* __restrict__ pwx = wx0;
    const f32x4* __restrict__ pwh = whc;
    auto wload = [&](f32x4 (&dst)[NWD]) {
        const bool isx = ps_ < Qxp;
        const f32x4* __restrict__ p = (isx ? pwx : pwh) + lane;
#pragma unroll
        for (int k = 0; k < NWD; ++k) dst[k] = p[64 * k];
        pwx += isx ? CST : 0;
        pwh += isx ? 0 : CST;
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
    // (X3H: chunk q is the pair hs[2q], hs[2q + 1] = its hi and lo pieces)
    auto hread = [&](int q) -> f32x4 { return hs[(q < Qhe ? q : Qhe - 1) * (X3H ? 128 : 64) + lane]; };   // padding: any finite chunk
    auto hread_lo = [&](int q) -> f32x4 { return hs[(q < Qhe ? q : Qhe - 1) * 128 + 64 + lane]; };
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
    constexpr int HSC = (BF && !X3H) ? 2 : 4;
#pragma unroll
    for (int m = 0; m < HSC; ++m) hs[(c * HSC + m) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float wS = X3H ? a.scale[0] : 1.f, wInvS = X3H ? a.scale[1] : 1.f;
    f32x4 xr2[3];                                  // X3H: second raw half of the ring's input chunks
    auto loadxq = [&](int t, int q, f32x4& second) -> f32x4 {
        if (!BF) return loadx_t(t, q);
        if (X3H) {
            second = XMODE == 1 ? f32x4{0.f, 0.f, 0.f, 0.f} : loadx_t(t, 2 * q + 1);
            return XMODE == 1 ? loadx_t(t, q) : loadx_t(t, 2 * q);
        }
        if (XMODE == 1) return cvt8(loadx_t(t, q), f32x4{0.f, 0.f, 0.f, 0.f});
        return cvt8(loadx_t(t, 2 * q), loadx_t(t, 2 * q + 1));
    };
    // one chunk into the three gate accumulators
    auto mmx = [&](const f32x4 (&w)[NWD], const f32x4 x, const f32x4 x2, f32x16& g0, f32x16& g1, f32x16& g2, const bool is_h) {
        if constexpr (X3H) {
            f32x4 bh, bl;
            if (is_h) { bh = x; bl = x2; }         // pieces published by the tile's owner
            else split2h(x, x2, bh, bl);
            g0 = mfma_h(w[0], bl, g0); g1 = mfma_h(w[2], bl, g1); g2 = mfma_h(w[4], bl, g2);
            g0 = mfma_h(w[1], bh, g0); g1 = mfma_h(w[3], bh, g1); g2 = mfma_h(w[5], bh, g2);
            g0 = mfma_h(w[0], bh, g0); g1 = mfma_h(w[2], bh, g1); g2 = mfma_h(w[4], bh, g2);
        } else {
            f32x4 w3[3] = {w[0], w[1], w[2]};
            mfma_x3<BF>(w3, x, g0, g1, g2);
        }
    };
    f32x4 wr[3][NWD], xr[3];
    wload(wr[0]); wload(wr[1]); wload(wr[2]);
    if (XMODE == 3) {
        loadxp(0);
    } else {
#pragma unroll
        for (int st = 0; st < 3; ++st) xr[st] = loadxq(0, st, xr2[st]);
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
                for (int s4 = 0; s4 < 4; ++s4) {            // (X3H: the accumulators hold S x the pre-activations)
                    ar[4 * m + s4] = b0[s4] * wS;
                    az[4 * m + s4] = b1[s4] * wS;
                    anx[4 * m + s4] = b2[s4] * wS;
                    anh[4 * m + s4] = b3[s4] * wS;
                }
            }
        }
        // chunk q of h_{t-1} is requested one chunk ahead of its MFMAs (LDS latency under the previous group)
        f32x4 hv = hread(0), hv2 = hv;
        if constexpr (X3H) hv2 = hread_lo(0);
        int qh = 0;                                // next h chunk to consume
        if (XMODE == 1) {
            // first ring turn: the single x chunk, then h chunks 0 and 1
            mmx(wr[0], xr[0], xr2[0], ar, az, anx, false);
            wload(wr[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 1; st < 3; ++st) {
                const f32x4 hn = hread(qh + 1);
                f32x4 hn2 = hn;
                if constexpr (X3H) hn2 = hread_lo(qh + 1);
                mmx(wr[st], hv, hv2, ar, az, anh, true);
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
                hv = hn; hv2 = hn2; ++qh;
            }
        } else {
            for (int q0 = 0; q0 < Qxp; q0 += 3) {
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    mmx(wr[st], xr[st], xr2[st], ar, az, anx, false);
                    wload(wr[st]);
                    xr[st] = loadxq(t, q0 + st + 3, xr2[st]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        auto hturn = [&]() {                       // one ring turn of h chunks qh .. qh + 2
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const f32x4 hn = hread(qh + st + 1);
                f32x4 hn2 = hn;
                if constexpr (X3H) hn2 = hread_lo(qh + st + 1);
                mmx(wr[st], hv, hv2, ar, az, anh, true);
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
                hv = hn; hv2 = hn2;
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
                for (int st = 0; st < 3; ++st) xr[st] = loadxq(tn, st, xr2[st]);
            }
        }
        // ---- gates (reference GRULayer / RNNDecoder: torch.nn.GRU equations, r|z|n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (X3H) { ar[r] *= wInvS; az[r] *= wInvS; anx[r] *= wInvS; anh[r] *= wInvS; }
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
        if (X3H) {
            f32x4 ph, pl;
            split2h(hvv[0], hvv[1], ph, pl);
            hs[(2 * c + 0) * 128 + lane] = ph; hs[(2 * c + 0) * 128 + 64 + lane] = pl;
            split2h(hvv[2], hvv[3], ph, pl);
            hs[(2 * c + 1) * 128 + lane] = ph; hs[(2 * c + 1) * 128 + 64 + lane] = pl;
        } else if (BF) {
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

template <int XM, bool F>
static void launch_split_x3(const GruArgs& a, unsigned grid, int ncg, size_t lds, hipStream_t s) {
    if constexpr (XM != 3) {
        if (a.Gates) hipLaunchKernelGGL((k_gru_split<XM, F, true, true, true>), dim3(grid), dim3(64 * ncg), lds, s, a);
        else hipLaunchKernelGGL((k_gru_split<XM, F, false, true, true>), dim3(grid), dim3(64 * ncg), lds, s, a);
    }
}

static int launch_gru_split(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + 31) / 32);
    const size_t lds = ((size_t)ncg * 1024 + (fc ? (size_t)ncg * a.out_dim * 32 : 0)) * sizeof(float);
    if (lds > 64 * 1024) return -2;
    const int xm = xmode == 3 ? 3 : (xmode == 0 ? 0 : (a.Qxp == 1 ? 1 : 2));
    const bool save = a.Gates != nullptr;
    if (a.x3 && (xm == 3 || a.Wxq == nullptr || a.scale == nullptr)) return -2;
#define SPLIT_CASE(XM, F)                                                                                        \
    if (xm == XM && fc == F) {                                                                                   \
        if (a.x3) launch_split_x3<XM, F>(a, grid, ncg, lds, s);                                                  \
        else if (a.bf16 && !save) hipLaunchKernelGGL((k_gru_split<XM, F, false, true>), dim3(grid), dim3(64 * ncg), lds, s, a); \
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

// split-operand (two fp16 pieces) build of the hidden-tile-split kernel: a.x3 = 1, a.Wh the [gate][piece] pack, a.Wxq the
// two-piece input pack in [tile][chunk] order, a.Qxp in 16-feature chunks (a multiple of 3, or 1 for the decoder)
int launch_gru_split_x3(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (!a.x3 || ncg < 1 || xmode == 3) return -2;
    if (xmode == 0 && ((a.ldx & 3) != 0 || a.Qxp % 3 != 0)) return -2;
    if (xmode != 0 && a.Qxp != 1 && a.Qxp % 3 != 0) return -2;
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
        if (!a.x3 && ncg >= 2 && groups <= 2L * n_cu && lds <= 64 * 1024) return launch_gru_split(a, ncg, xmode, fc, s);
    }
    // two groups per wave once that still gives every SIMD a wave
    const bool two = (a.B + 31) / 32 >= 8L * n_cu;
    if (a.x3) {
        return ncg >= 5 ? launch_gru_big_x3_hi(a, ncg, xmode, fc, two, s) : launch_gru_big_x3_lo(a, ncg, xmode, fc, two, s);
    }
    return a.bf16 ? launch_gru_big_bf16(a, ncg, xmode, fc, two, s) : launch_gru_big_f32(a, ncg, xmode, fc, two, s);
}

}  // namespace mtadgat
