// k_conv_win: the window convolution (reference ConvLayer.forward, modules.py:18-22: zero padding per window, cross-correlation,
// bias, ReLU) on the 16-bit matrix pipe with fp32-class results -- one workgroup per window.
//
// k_conv_lds (mtadgat_kernels.hip) gives every wave 32 flat rows of the batch: it stages 38 input rows with one 4-byte load
// per lane and row, streams 98 KB of fp32 weights per 32 rows and spends 392 v_mfma_f32_32x32x2_f32 (64 cycles each) on them:
// 3.5 ms per 65 536 windows at (W = 100, F = 55), 61 % of that on the fp32 matrix pipe.  Here:
//   * the window (W x F floats, contiguous) is read with 16-byte loads, split ONCE into two fp16 pieces and kept in LDS with
//     its zero halo rows -- no per-tap masking, no per-chunk splitting.  The pieces are taken after scaling the window by a
//     power of two that puts its largest |x| into [2^13, 2^14): any input range is served (un-normalised series included),
//     every element keeps 22 significant bits down to 2^-28 of the window's maximum;
//   * the weights are two fp16 pieces of S * W (S: the layer's power of two, as for the other split-operand kernels),
//     [tile][16-channel chunk][piece][lane] words in the order of mtadgat_device.h: w x ~= wh xl + wl xh + wh xh, three
//     v_mfma_f32_32x32x16_f16 (32 cycles each) per 16 input channels, tap and 32 x 32 output tile;
//   * a wave owns two 32-row tiles of the window and both 32-channel tiles of the output: a weight word feeds four MFMAs.
// Output: h_cat[:, :F] (the fused front end's only consumer of the convolution), fp32, 16 bytes per lane and store; the
// largest value written is recorded for the attention layers' range guard (ConvArgs::vmax).
#include "mtadgat_device.h"

namespace mtadgat {

// CW_RT 32-row tiles per wave, NWV = 4 / CW_RT waves per window (W <= 128)
template <int NTB, int CW_RT>
__global__ __launch_bounds__(256 / CW_RT, CW_RT == 1 ? 4 : 3) void k_conv_win(const ConvArgs a) {
    constexpr int NTHR = 256 / CW_RT, NWV = NTHR / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    if ((a.dbg & 8) && blockIdx.x < 4096u) {           // (experiment: spread the first workgroups in time)
        const unsigned nap = (blockIdx.x * 2654435761u) >> 28;
        for (unsigned k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(100);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long win = blockIdx.x;
    const int W = a.W, F = a.F, Fq = a.Fq, taps = a.taps, pad = a.pad;
    // piece pitch in halfs: 4 x odd -> conflict-free 8-byte operand reads.  The smallest such pitch that holds the F channels: the
    // last chunk of a row then reads on into the next row (or the spare row) -- finite values against zero weights -- and a
    // workgroup takes 25.7 instead of 29.1 KB of LDS at (W = 100, F = 55): six of them per CU instead of five
    const int pvh = a.pvh;
    const int nrows = W + taps - 1;                    // staged rows: the window between its zero halos
    unsigned short* __restrict__ Xh = reinterpret_cast<unsigned short*>(smem8);
    unsigned short* __restrict__ Xl = Xh + (nrows + 2) * pvh;        // (two spare zero rows per piece: padding rows of the last tile, overrun of the last chunk)
    float* __restrict__ red = reinterpret_cast<float*>(Xl + (nrows + 2) * pvh);      // [4] wave maxima, [2] the window's scale and its inverse

    // ---- the window as a flat array of W F floats (16-byte loads when its base allows), largest |x| of the window
    const long s0 = a.gather ? (a.starts ? a.starts[win] : a.start0 + win * a.stride) : win * (long)W;
    const float* __restrict__ xw = a.X + s0 * F;
    const unsigned short* __restrict__ xw16 = reinterpret_cast<const unsigned short*>(a.X) + s0 * F;     // x_bf16: bfloat16 elements, read directly
    const int total = W * F;
    const bool vec = a.x_bf16 ? (reinterpret_cast<unsigned long>(xw16) & 7) == 0 : (reinterpret_cast<unsigned long>(xw) & 15) == 0;
    constexpr int MAXU = 6 * CW_RT;                    // 16-byte units per thread held in registers (W F <= 6144: plan)
    f32x4 v[MAXU];
    const int nunit = (total + 3) >> 2;
    float mx = 0.f;
    // the loader is picked once (wave-uniform) and issues all of a thread's units before anything looks at them: with the choice
    // inside the unit loop each unit was a load, a wait for it, and its maximum -- MAXU memory round trips in a row per window
    const bool whole = vec && (total & 3) == 0;        // every unit is one aligned 16-byte (bf16 input: 8-byte) word
    if (whole && !a.x_bf16) {
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * NTHR;
            v[n] = *reinterpret_cast<const f32x4*>(xw + 4 * (u < nunit ? u : nunit - 1));
        }
    } else if (whole) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        u32x2_ w2[MAXU];
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * NTHR;
            w2[n] = *reinterpret_cast<const u32x2_*>(xw16 + 4 * (u < nunit ? u : nunit - 1));
        }
#pragma unroll
        for (int n = 0; n < MAXU; ++n)
            v[n] = f32x4{__uint_as_float(w2[n][0] << 16), __uint_as_float(w2[n][0] & 0xffff0000u), __uint_as_float(w2[n][1] << 16), __uint_as_float(w2[n][1] & 0xffff0000u)};
    } else if (a.x_bf16) {
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * NTHR, uc = u < nunit ? u : nunit - 1;
            unsigned short h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = xw16[4 * uc + e < total ? 4 * uc + e : total - 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[n][e] = __uint_as_float((unsigned)h[e] << 16);
        }
    } else {
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * NTHR, uc = u < nunit ? u : nunit - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[n][e] = xw[4 * uc + e < total ? 4 * uc + e : total - 1];
        }
    }
#pragma unroll
    for (int n = 0; n < MAXU; ++n) {
        const int u = tid + n * NTHR;
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, (4 * u + e < total) ? fabsf(v[n][e]) : 0.f);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    // zero halo rows, the spare row behind them and the channel padding [F, Fq) of the window's rows
    {
        const int hw = pvh >> 1;                       // dwords per row
        for (int u = tid; u < 2 * pad * hw; u += NTHR) {
            const int r = u / hw, c = u - r * hw;
            const int row = r < pad ? r : nrows - 2 * pad + r;
            reinterpret_cast<unsigned*>(Xh + row * pvh)[c] = 0u;
            reinterpret_cast<unsigned*>(Xl + row * pvh)[c] = 0u;
        }
        for (int u = tid; u < 2 * hw; u += NTHR) {
            reinterpret_cast<unsigned*>(Xh + nrows * pvh)[u] = 0u;
            reinterpret_cast<unsigned*>(Xl + nrows * pvh)[u] = 0u;
        }
        const int npadc = (Fq < pvh ? Fq : pvh) - F;
        for (int u = tid; u < W * npadc; u += NTHR) {
            const int r = u / npadc, c = F + (u - r * npadc);
            Xh[(pad + r) * pvh + c] = 0;
            Xl[(pad + r) * pvh + c] = 0;
        }
    }
    __syncthreads();
    if (a.dbg & 4) return;
    if (tid == 0) {
        float m = red[0];
#pragma unroll
        for (int w2 = 1; w2 < NWV; ++w2) m = fmaxf(m, red[w2]);
        // sx = 2^(13 - floor(log2 m)): exponent field 267 - e (m = 0, denormal or not finite: 1)
        const unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
        const unsigned es = (e == 0u || e >= 254u) ? 127u : 267u - e;
        const unsigned ec = es < 1u ? 1u : (es > 253u ? 253u : es);
        red[4] = __uint_as_float(ec << 23);
        red[5] = __uint_as_float((254u - ec) << 23);
    }
    __syncthreads();
    const float sx = red[4], sxi = red[5];
    {
        const float finv = 1.0f / (float)F;
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * NTHR;
            if (u < nunit) {
                const int f0 = 4 * u;
                int row = (int)(((float)f0 + 0.5f) * finv), col = f0 - row * F;
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    // two elements at a time; a pair may straddle the end of a row
                    int r0 = row, c0 = col, r1 = row, c1 = col + 1;
                    if (c1 >= F) { r1 = row + 1; c1 = 0; }
                    unsigned hw_, lw_;
                    split_pair_h(v[n][2 * e2] * sx, v[n][2 * e2 + 1] * sx, hw_, lw_);
                    if (f0 + 2 * e2 < total) { Xh[(pad + r0) * pvh + c0] = (unsigned short)hw_; Xl[(pad + r0) * pvh + c0] = (unsigned short)lw_; }
                    if (f0 + 2 * e2 + 1 < total) { Xh[(pad + r1) * pvh + c1] = (unsigned short)(hw_ >> 16); Xl[(pad + r1) * pvh + c1] = (unsigned short)(lw_ >> 16); }
                    col += 2;
                    if (col >= F) { col -= F; ++row; if (col >= F) { col -= F; ++row; } }      // (F = 1: a pair is two rows)
                }
            }
        }
    }
    __syncthreads();

    // ---- out[t][o] = sum_{tap, c} w[o][tap][c] x[t + tap - pad][c]: A = weights (32 output channels), B = input rows (lane (i, g):
    // row of tile + i, channels 16 cb + 4 g .. + 3 and + 8 .. -- mtadgat_device.h), K runs over taps x 16-channel chunks
    const int i = lane & 31, g = lane >> 5;
    const int QF = Fq >> 4, Q = taps * QF;
    const f32x4* __restrict__ Wp = a.Wp;              // [tile][Q][2 pieces][64]
    f32x16 acc[CW_RT][NTB];
#pragma unroll
    for (int rt = 0; rt < CW_RT; ++rt)
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][nb][r] = 0.f;
    int xoff[CW_RT];                                   // half offset of (row of the tile, channels 4 g) in the staged pieces
#pragma unroll
    for (int rt = 0; rt < CW_RT; ++rt) {
        const int t = 32 * (CW_RT * wave + rt) + i;
        xoff[rt] = (t < W ? t : nrows - taps + 1) * pvh + 4 * g;        // rows past the window read the spare zero rows
    }
    // weight words of a chunk: [tile][piece]; a ring of four chunks: the words of chunk q + 3 are requested before the MFMAs
    // of chunk q (the L2 round trip is longer than the 12 MFMAs of a chunk: with one chunk of lookahead the loop ran at the
    // memory latency, 2.0 ms per 65 536 windows)
    constexpr int RING = 4;
    f32x4 wr[RING][NTB][2];
    auto wload = [&](f32x4 (&w)[NTB][2], int q) {
        const int qc = q < Q ? q : Q - 1;
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            w[nb][0] = Wp[(((long)nb * Q + qc) * 2) * 64 + lane];
            w[nb][1] = Wp[(((long)nb * Q + qc) * 2 + 1) * 64 + lane];
        }
    };
    wload(wr[0], 0);
    wload(wr[1], 1);
    wload(wr[2], 2);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    int tap = 0, cb = 0;
#pragma unroll 1
    for (int q0 = 0; q0 < ((a.dbg & 1) ? 0 : Q); q0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int q = q0 + u;
            wload(wr[(u + 3) % RING], q + 3);
            __builtin_amdgcn_sched_barrier(0);
            if (q < Q) {
                const int ko = tap * pvh + 16 * cb;
#pragma unroll
                for (int rt = 0; rt < CW_RT; ++rt) {
                    const unsigned short* __restrict__ ph = Xh + xoff[rt] + ko;
                    const unsigned short* __restrict__ pl = Xl + xoff[rt] + ko;
                    const u32x2 ha = *reinterpret_cast<const u32x2*>(ph), hb = *reinterpret_cast<const u32x2*>(ph + 8);
                    const u32x2 la = *reinterpret_cast<const u32x2*>(pl), lb = *reinterpret_cast<const u32x2*>(pl + 8);
                    const f32x4 xh = __builtin_bit_cast(f32x4, u4{ha[0], ha[1], hb[0], hb[1]});
                    const f32x4 xl = __builtin_bit_cast(f32x4, u4{la[0], la[1], lb[0], lb[1]});
#pragma unroll
                    for (int nb = 0; nb < NTB; ++nb) {
                        acc[rt][nb] = mfma_h(wr[u][nb][0], xl, acc[rt][nb]);
                        acc[rt][nb] = mfma_h(wr[u][nb][1], xh, acc[rt][nb]);
                        acc[rt][nb] = mfma_h(wr[u][nb][0], xh, acc[rt][nb]);
                    }
                }
                if (++cb == QF) { cb = 0; ++tap; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: 1 / (S sx), bias, ReLU; h_cat[:, :F] (+ the zero alignment padding of the row), range of the outputs
    const float osc = a.wscale[1] * sxi;               // both factors are powers of two
    if (a.dbg & 2) { if (acc[0][0][0] == 12345.f) a.HCAT[0] = osc; return; }
    float vmx = 0.f;
#pragma unroll
    for (int rt = 0; rt < CW_RT; ++rt) {
        const int t = 32 * (CW_RT * wave + rt) + i;
        if (t < W) {
            float* __restrict__ hrow = a.HCAT + (win * W + t) * (long)a.Dp;
            if (g == 0)
                for (int c = 3 * F; c < a.Dp; ++c) hrow[c] = 0.f;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int col = 32 * nb + 8 * m + 4 * g;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
                    f32x4 y;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        y[s4] = fmaxf(__builtin_fmaf(acc[rt][nb][4 * m + s4], osc, bv[s4]), 0.f);
                        vmx = (col + s4 < F) ? fmaxf(vmx, y[s4]) : vmx;
                    }
                    if (col + 3 < F) {
                        *reinterpret_cast<f32x4*>(hrow + col) = y;
                    } else {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            if (col + s4 < F) hrow[col + s4] = y[s4];
                    }
                }
        }
    }
    if (a.vmax) {
        vmx = wave_max(vmx);
        if (lane == 0 && !(vmx <= __uint_as_float(*a.vmax))) atomicMax(a.vmax, __float_as_uint(vmx));
    }
}

int conv_win_pitch(int F, int Fq) {
    int p = (F + 3) & ~3;
    if ((p >> 2) % 2 == 0) p += 4;                     // 4 x odd
    while (2 * p < Fq) p += 8;                         // a row's last chunk reads Fq - p halfs past its end: within the two spare rows
    return p < Fq + 4 ? p : Fq + 4;
}
size_t conv_win_lds(int W, int F, int Fq, int taps) {
    // (+ one row of slack per piece: the last row's last chunk reads up to Fq - pitch halfs past its end)
    return (size_t)2 * (W + taps + 1) * conv_win_pitch(F, Fq) * 2 + 8 * sizeof(float);          // (W + taps - 1 staged rows + 2 spare)
}

// window-per-workgroup convolution: fp32 or bfloat16 windows (materialised or gathered from a series), h_cat output only
bool conv_win_applies(const ConvArgs& a) {
    if (a.bf16 || !a.HCAT || a.XC || a.XCT || a.Y || !a.wscale) return false;
    if (a.taps != 2 * a.pad + 1 || a.NT > 2 || a.W > 128 || a.W < 1) return false;
    if ((long)a.W * a.F > 12L * 128 * 4 || (a.Fq & 15) != 0 || a.Fq < a.F) return false;
    if ((a.Dp & 3) != 0) return false;
    return conv_win_lds(a.W, a.F, a.Fq, a.taps) <= 64 * 1024;
}

int launch_conv_win(const ConvArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (!conv_win_applies(a)) return -2;
    const size_t lds = conv_win_lds(a.W, a.F, a.Fq, a.taps);
    ConvArgs& am = const_cast<ConvArgs&>(a);
    am.pvh = conv_win_pitch(a.F, a.Fq);
    if (getenv("MTADGAT_CONVW_WIDE")) am.pvh = a.Fq + 4;
    if (const char* e_ = getenv("MTADGAT_CONVW_DBG")) am.dbg = atoi(e_);
    static const int rt = getenv("MTADGAT_CONVW_RT") ? atoi(getenv("MTADGAT_CONVW_RT")) : 2;      // (measurement hook: row tiles per wave)
    if (a.NT >= 2) {
        if (rt == 1) hipLaunchKernelGGL((k_conv_win<2, 1>), dim3((unsigned)a.B), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_conv_win<2, 2>), dim3((unsigned)a.B), dim3(128), lds, s, a);
    } else {
        if (rt == 1) hipLaunchKernelGGL((k_conv_win<1, 1>), dim3((unsigned)a.B), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_conv_win<1, 2>), dim3((unsigned)a.B), dim3(128), lds, s, a);
    }
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
