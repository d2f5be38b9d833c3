// MI355X (gfx950 / CDNA4) kernels for the MTAD-GAT per-window forward path.
//
// Everything here is written for wave64 + the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fmaf chain) because the
// contract is <= 1e-5 parity with the reference's float32 forward.
//
// One idea carries all GEMM-shaped work -- the "F-layout".  A wave owns 32 data
// rows (windows, or (window, t) / (window, feature) pairs).  Feature vectors of
// those rows live in registers as 8-wide chunks: for chunk q lane (i = lane&31,
// g = lane>>5) holds the four features 8q+4g .. 8q+4g+3 of row i as a float4.
// With the weights as the MFMA "A" operand (rows = output features) and the
// activations as the "B" operand (columns = data rows), the MFMA k-step s of
// chunk q multiplies weight column 8q+4g+s by activation feature 8q+4g+s, and
// the 32x32 result tile comes out with lane (i, half) holding output features
// 32n + 8m + 4*half + {0..3} in accumulator registers 4m..4m+3 -- i.e. again in
// F-layout, chunk 4n+m.  So the output of one product is directly the "B"
// operand of the next one: the GRU's hidden state never leaves the register
// file between time steps, and no LDS transpose or barrier is needed.
//
// Weights are pre-packed on the host (mtadgat_pack.cpp) in exactly the order a
// wave consumes them, so every weight fetch is one coalesced 1 KiB
// global_load_dwordx4 per wave, served by the L2 (all packed weights of a model
// are ~2 MB and stay resident).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "mtadgat_kernels.h"

#ifndef MTADGAT_ATTEND_DEPTH
#define MTADGAT_ATTEND_DEPTH 2
#endif

namespace mtadgat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma4(const f32x4 w, const f32x4 x, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[3], acc, 0, 0, 0);
    return acc;
}

// one chunk for three gate accumulators, k-steps interleaved across the accumulators so consecutive
// MFMAs never depend on each other
__device__ __forceinline__ void mfma4x3(const f32x4 (&w)[3], const f32x4 x, f32x16& a0, f32x16& a1, f32x16& a2) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0][s], x[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1][s], x[s], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2][s], x[s], a2, 0, 0, 0);
    }
}

// four features k0..k0+3 of a row; zero beyond kvalid.  vec_ok: row base and k0 are 16-byte aligned
__device__ __forceinline__ f32x4 load_feat4(const float* __restrict__ row, int k0, int kvalid, bool vec_ok) {
    f32x4 v;
    if (vec_ok && k0 + 3 < kvalid) {
        v = *reinterpret_cast<const f32x4*>(row + k0);
    } else {
        v[0] = (k0 + 0 < kvalid) ? row[k0 + 0] : 0.f;
        v[1] = (k0 + 1 < kvalid) ? row[k0 + 1] : 0.f;
        v[2] = (k0 + 2 < kvalid) ? row[k0 + 2] : 0.f;
        v[3] = (k0 + 3 < kvalid) ? row[k0 + 3] : 0.f;
    }
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// GRU gate non-linearities on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each):
// ~8 VALU ops per gate instead of ~40 for the libm versions; the product x*log2(e) is formed in
// two pieces so the exponent keeps float accuracy for |x| up to ~40.
#ifndef MTADGAT_ACCURATE_GATES
__device__ __forceinline__ float exp_fast(float x) {   // e^x, argument clamped to [-88, 88] (no inf/NaN in, none out)
    x = __builtin_amdgcn_fmed3f(x, -88.0f, 88.0f);
    const float c_hi = 1.4426950216293335f;             // log2(e) rounded to float
    const float c_lo = 1.9259629911266175e-08f;          // log2(e) - c_hi
    const float hi = x * c_hi;
    const float lo = __builtin_fmaf(x, c_hi, -hi) + x * c_lo;
    return __builtin_amdgcn_exp2f(hi) * (1.0f + 0.6931471805599453f * lo);
}
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(exp_fast(2.0f * x) + 1.0f); }
__device__ __forceinline__ float soft_exp(float x) { return exp_fast(x); }
__device__ __forceinline__ float soft_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
__device__ __forceinline__ float gate_sigmoid(float x) { return sigmoidf_(x); }
__device__ __forceinline__ float gate_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ float soft_exp(float x) { return expf(x); }
__device__ __forceinline__ float soft_rcp(float x) { return 1.0f / x; }
#endif

// ---------------------------------------------------------------------------
// rowgemm: Y[r, :] = act(W * X[r, :] + bias) for R data rows, 32 rows per wave.
//   reference: every nn.Linear on the path -- the GAT `lin` projections
//   (modules.py:76-77, :81, :176-177, :181; re-associated as DESIGN.md section 3
//   describes) and Forecasting_Model (modules.py:307-311).
// ---------------------------------------------------------------------------
template <int NTB>
__global__ __launch_bounds__(64) void k_rowgemm(const RowGemmArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long rowc = row < a.R ? row : a.R - 1;
    const float* __restrict__ xrow = a.X + rowc * a.ldx;
    const bool xvec = (a.ldx & 3) == 0;
    const f32x4* __restrict__ Wp = a.Wp;
    const int Q = a.Q;

    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

        f32x4 xv = load_feat4(xrow, 4 * g, a.Kvalid, xvec);
        f32x4 w[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            w[nb] = Wp[((long)n * Q) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 xn = load_feat4(xrow, 8 * qn + 4 * g, a.Kvalid, xvec);
            f32x4 wn[NTB];
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
                wn[nb] = Wp[((long)n * Q + qn) * 64 + lane];
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w[nb], xv, acc[nb]);
            xv = xn;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w[nb] = wn[nb];
        }
        // epilogue
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
            const bool transposed = (n0 + nb) >= a.NT_rm;
            const long grp = row / a.group;
            const int member = (int)(row - grp * a.group);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
                f32x4 v;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float t = acc[nb][4 * m + s] + bv[s];
                    v[s] = a.relu ? fmaxf(t, 0.f) : t;
                }
                if (row < a.R) {
                    if (transposed) {   // lanes i <-> consecutive group members: coalesced 4-byte stores
                        float* tp = a.YT + (grp * a.YT_rows + (col - 32 * a.NT_rm)) * (long)a.YT_ld + member;
#pragma unroll
                        for (int s = 0; s < 4; ++s) tp[(long)s * a.YT_ld] = v[s];
                    } else {
                        float* yp = a.Y + row * a.ldy + col;
                        if (a.vec_store && col + 3 < a.Nvalid) {
                            *reinterpret_cast<f32x4*>(yp) = v;
                        } else {
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                if (col + s < a.Nvalid) yp[s] = v[s];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// conv: xc[b,t,o] = ReLU(bias[o] + sum_{i,j} w[o,i,j] * x[b, t+j-pad, i]), zero
// outside the window.  reference ConvLayer.forward, modules.py:18-22.
// Implicit GEMM: data rows = (window, t), K = taps x Fp, loaded straight from x.
// Writes xc (b*W, Fp), its transpose xcT (b*F, Wp) and columns [0,F) of h_cat.
// ---------------------------------------------------------------------------
template <int NTB>
__global__ __launch_bounds__(64) void k_conv(const ConvArgs a) {
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long row = (long)blockIdx.x * 32 + i;
    const long R = a.B * a.W;
    const long rowc = row < R ? row : R - 1;
    const long win = rowc / a.W;
    const int t = (int)(rowc - win * a.W);
    const float* __restrict__ xwin = a.X + win * (long)a.W * a.F;
    const int QF = a.Fp >> 3;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp;

    if (a.HCAT && row < R && g == 0)      // zero the alignment padding of the h_cat row (the GRU reads it unguarded)
        for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row * a.Dp + c] = 0.f;
    auto loadx = [&](int q) -> f32x4 {
        const int tap = q / QF;
        const int cb = q - tap * QF;
        const int tt = t + tap - a.pad;
        const int c0 = 8 * cb + 4 * g;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tt >= 0 && tt < a.W) {
            const float* p = xwin + (long)tt * a.F + c0;
            v[0] = (c0 + 0 < a.F) ? p[0] : 0.f;
            v[1] = (c0 + 1 < a.F) ? p[1] : 0.f;
            v[2] = (c0 + 2 < a.F) ? p[2] : 0.f;
            v[3] = (c0 + 3 < a.F) ? p[3] : 0.f;
        }
        return v;
    };

    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

        f32x4 xv = loadx(0);
        f32x4 w[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            w[nb] = Wp[((long)n * Q) * 64 + lane];
        }
        for (int q = 0; q < Q; ++q) {
            const int qn = (q + 1 < Q) ? q + 1 : q;
            const f32x4 xn = loadx(qn);
            f32x4 wn[NTB];
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
                wn[nb] = Wp[((long)n * Q + qn) * 64 + lane];
            }
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w[nb], xv, acc[nb]);
            xv = xn;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w[nb] = wn[nb];
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = col + s;
                    const float v = fmaxf(acc[nb][4 * m + s] + bv[s], 0.f);
                    if (row < R && o < a.F) {
                        if (a.XC) a.XC[row * a.Fp + o] = v;
                        if (a.XCT) a.XCT[(win * a.F + o) * (long)a.Wpad + t] = v;
                        if (a.HCAT) a.HCAT[row * a.Dp + o] = v;
                        if (a.Y) a.Y[row * a.F + o] = v;
                    }
                }
            }
        }
    }
}

// conv, LDS-staged variant (used when 32+taps-1 input rows of Fp floats fit a wave's LDS budget):
// the wave copies the input rows its 32 output rows touch into LDS once with coalesced loads and
// takes every MFMA B operand from there -- the straight-from-global version above re-reads each
// input row `taps` times with 4-byte gathers (PMC: 5-9x FETCH amplification, TA-bound).
template <int NTB>
__global__ __launch_bounds__(256) void k_conv_lds(const ConvArgs a) {
    // blockDim.x / 64 independent waves per workgroup, each with its own 32 output rows and LDS slice:
    // one-wave workgroups are launched too slowly to keep the matrix pipes fed at ~50 us per wave
    extern __shared__ __attribute__((aligned(16))) float xs_all[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    float* __restrict__ xs = xs_all + wv * ((32 + a.taps - 1) * (a.Fp + 4));
    const int i = lane & 31, g = lane >> 5;
    const long R = a.B * a.W;
    const long r0 = ((long)blockIdx.x * nwv + wv) * 32;
    const long row = r0 + i;
    const long rowc = row < R ? row : R - 1;
    const long win = rowc / a.W;
    const int t = (int)(rowc - win * a.W);
    const int Fld = a.Fp + 4;
    const int nrow = 32 + a.taps - 1;
    // source row of LDS row rr: flat row r0 - pad + rr of the (B*W, F) batch; in gather mode the windows
    // are views of the device-resident series (no (b, W, F) copy exists) and (window, t) advance together
    long gw = 0; int gt = 0;                          // window / position of the current flat row (gather mode)
    if (a.gather) {
        const long f0 = r0 - a.pad < 0 ? 0 : r0 - a.pad;
        gw = f0 / a.W;
        gt = (int)(f0 - gw * a.W);
    }
    auto src_row = [&](int rr, bool& ok) -> const float* {
        const long flat = r0 - a.pad + rr;
        ok = flat >= 0 && flat < R;
        long srow = ok ? flat : 0;
        if (a.gather) {
            const long wc = gw < a.B ? gw : a.B - 1;
            const long s0 = a.starts ? a.starts[wc] : a.start0 + wc * a.stride;
            srow = s0 + gt;
            if (flat >= 0 && ++gt == a.W) { gt = 0; ++gw; }      // rows are visited in increasing order, once each
        }
        return a.X + srow * a.F;
    };
    if (Fld <= 64) {
        // one element per lane and row; every load is unconditional (clamped column, a valid row for rows
        // outside the batch) and all of them are issued before the first LDS store: one memory round trip
        // per wave instead of one per row (a guarded load compiles to a branch + s_waitcnt vmcnt(0))
        constexpr int MAXR = 40;
        float v[MAXR];
        const int colc = lane < a.F ? lane : a.F - 1;
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr) {
            v[rr] = 0.f;
            if (rr < nrow) {                              // wave-uniform
                bool ok;
                const float* __restrict__ src = src_row(rr, ok);
                const float t = src[colc];
                v[rr] = (ok && lane < a.F) ? t : 0.f;
            }
        }
#pragma unroll
        for (int rr = 0; rr < MAXR; ++rr)
            if (rr < nrow && lane < Fld) xs[rr * Fld + lane] = v[rr];
        for (int rr = MAXR; rr < nrow; ++rr) {
            bool ok;
            const float* __restrict__ src = src_row(rr, ok);
            if (lane < Fld) xs[rr * Fld + lane] = (ok && lane < a.F) ? src[lane] : 0.f;
        }
    } else {
        for (int rr = 0; rr < nrow; ++rr) {
            bool ok;
            const float* __restrict__ src = src_row(rr, ok);
            for (int col = lane; col < Fld; col += 64) xs[rr * Fld + col] = (ok && col < a.F) ? src[col] : 0.f;
        }
    }
    __syncthreads();
    const int QF = a.Fp >> 3;
    const int Q = a.taps * QF;
    const f32x4* __restrict__ Wp = a.Wp;
    if (a.HCAT && row < R && g == 0)      // zero the alignment padding of the h_cat row (the GRU reads it unguarded)
        for (int c = 3 * a.F; c < a.Dp; ++c) a.HCAT[row * a.Dp + c] = 0.f;
    // B operand of chunk (tap, cb): input row i + tap of the staged block, columns 8 cb + 4 g .. +3; rows
    // of the neighbouring window count as zero padding (per window, modules.py:14,20)
    const float* __restrict__ xrow = static_cast<const float*>(__builtin_assume_aligned(xs, 16)) + i * Fld + 4 * g;
    auto loadx = [&](int tap, int cb) -> f32x4 {
        const int tt = t + tap - a.pad;
        f32x4 v = *reinterpret_cast<const f32x4*>(xrow + tap * Fld + 8 * cb);
        const bool ok = tt >= 0 && tt < a.W;
        v[0] = ok ? v[0] : 0.f; v[1] = ok ? v[1] : 0.f; v[2] = ok ? v[2] : 0.f; v[3] = ok ? v[3] : 0.f;
        return v;
    };
    for (int n0 = 0; n0 < a.NT; n0 += NTB) {
        f32x16 acc[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const f32x4* wq[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            const int n = (n0 + nb < a.NT) ? n0 + nb : a.NT - 1;
            wq[nb] = Wp + ((long)n * Q) * 64 + lane;
        }
        // two register sets in turn (no copies): the weights of chunk q + 1 are in flight during the MFMAs
        // of chunk q
        f32x4 w0[NTB], w1[NTB];
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) w0[nb] = wq[nb][0];
        int tap = 0, cb = 0;
        for (int q = 0; q < Q; q += 2) {
            const int q1 = q + 1 < Q ? q + 1 : q;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w1[nb] = wq[nb][(long)q1 * 64];
            const f32x4 xv0 = loadx(tap, cb);
            if (++cb == QF) { cb = 0; ++tap; }
            __builtin_amdgcn_sched_barrier(0);      // keep the next chunk's weight loads ahead of these MFMAs
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w0[nb], xv0, acc[nb]);
            __builtin_amdgcn_sched_barrier(0);
            const int q2 = q + 2 < Q ? q + 2 : Q - 1;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) w0[nb] = wq[nb][(long)q2 * 64];
            if (q + 1 < Q) {
                const f32x4 xv1 = loadx(tap, cb);
                if (++cb == QF) { cb = 0; ++tap; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb) acc[nb] = mfma4(w1[nb], xv1, acc[nb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) {
            if (n0 + nb >= a.NT) break;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * (n0 + nb) + 8 * m + 4 * g;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + col);
                f32x4 v;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) v[s4] = fmaxf(acc[nb][4 * m + s4] + bv[s4], 0.f);
                if (row < R) {
                    if (a.HCAT) {   // rows of h_cat are 16-byte aligned: one store per 4 channels
                        float* hp = a.HCAT + row * a.Dp + col;
                        if (col + 3 < a.F) {
                            *reinterpret_cast<f32x4*>(hp) = v;
                        } else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                if (col + s4 < a.F) hp[s4] = v[s4];
                        }
                    }
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int o = col + s4;
                        if (o < a.F) {
                            if (a.XC) a.XC[row * a.Fp + o] = v[s4];
                            if (a.XCT) a.XCT[(win * a.F + o) * (long)a.Wpad + t] = v[s4];
                            if (a.Y) a.Y[row * a.F + o] = v[s4];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// attend: complete-graph attention scores + softmax + aggregation + sigmoid for
// a block of query nodes of one window.  reference FeatureAttentionLayer.forward
// (modules.py:65-95) / TemporalAttentionLayer.forward (modules.py:166-193).
//
// GATv2 score, re-associated:
//   e_ij = c_i + d_j + sum_{k in P} |L'_ik + R'_jk| - sum_{k in N} |L'_ik + R'_jk| + bias_ij
// with L', R', c, d produced by k_rowgemm from the packed projection (columns
// [0,PT) = L', [PT,2PT) = R', 2PT = c, 2PT+1 = d of each node's row in LR).
// GAT (v1): e_ij = LeakyReLU(c_i + d_j) + bias_ij (PT = 0).
//
// lane <-> key node j (JPL nodes per lane), the query node i is wave-uniform so
// L'_i comes in through scalar loads and the inner loop is 2 VALU ops/element.
// The softmax'd rows are staged through LDS into MFMA B-operand order and the
// aggregation att @ V runs on the matrix pipe.
// ---------------------------------------------------------------------------
// compile-time loop (DPP controls must be immediates)
template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// value of lane N of this lane's 16-lane row (gfx90a+ DPP row_newbcast); folds into the consuming VALU op
template <int N>
__device__ __forceinline__ float row_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
}

// one 8-wide k tile of the pairwise term for all IB query rows.
//   r[jj][e]  : R'[k0+e][j]  for this lane's key nodes j (one VGPR each)
//   lt[x]     : L' tile, lane n of every 16-lane row holds L'[row 2x + (n>>3)][k0 + (n&7)]
// so L'_ik reaches all lanes through a DPP row broadcast fused into the add: 2 VALU ops/element
// (v_add_f32_dpp + v_add_f32 |t|), no scalar loads, no LDS.  Measured on MI355X (scratch
// microbenchmark): DPP add 4.3 cycles, |abs| accumulate 2.7 cycles per wave64
// instruction with >= 2 waves/SIMD.  A software-pipelined variant (no back-to-back dependent pair,
// no s_nop) measured slower because its extra live temporaries cost a wave of occupancy.
template <int JPL, int IB, bool NEG>
__device__ __forceinline__ void attend_tile(float (&acc)[IB][JPL], const float (&r)[JPL][8], const float (&lt)[IB / 2]) {
#pragma unroll
    for (int x = 0; x < IB / 2; ++x) {
        const float lv = lt[x];
        static_for<0, 16>([&](auto nn) {
            constexpr int N = decltype(nn)::value;
            constexpr int e = N & 7;
            const int ib = 2 * x + (N >> 3);
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const float t = row_bcast<N>(lv) + r[jj][e];
                if (NEG)
                    acc[ib][jj] -= fabsf(t);
                else
                    acc[ib][jj] += fabsf(t);
            }
        });
    }
}

// wave-wide all-reduce without LDS: butterfly inside each 16-lane row with DPP (fused into the
// v_max / v_add), then the four row results meet through v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_move<0x140>(v));   // row_mirror
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}

template <int JPL, int IB>
__global__ __launch_bounds__(64, (JPL <= 2 ? 3 : 2)) void k_attend(const AttendArgs a) {
    __shared__ __attribute__((aligned(16))) float att_s[32][68];
    static_assert(IB % 2 == 0 && IB <= 32, "IB");
    constexpr int NL = IB / 2;
    const int lane = threadIdx.x;
    // XCD-aware block -> (window, row block) map: dispatch ids b, b+8, b+16, ... run on the same
    // XCD (b % 8), so giving them the row blocks of ONE window lets that window's R'^T / V tiles be
    // fetched from HBM once and served to the other row blocks from that XCD's L2.
    const long blk = blockIdx.x;
    long win;
    int rb;
    if (a.xcd_map) {
        const long grp = blk / (8 * a.nblk);
        const int within = (int)(blk - grp * (8 * a.nblk));
        win = grp * 8 + (within & 7);
        rb = within >> 3;
    } else {
        win = blk / a.nblk;
        rb = (int)(blk - win * a.nblk);
    }
    if (win >= a.nwin) return;
    const int i0 = rb * a.rows_per_blk;
    const int nrows = min(a.rows_per_blk, a.K - i0);
    const int K = a.K, ldl = a.ldl, PT = a.PT, Kp = a.Kp;
    const float* __restrict__ Lrow0 = a.LC + (win * K + i0) * (long)ldl;
    const float* __restrict__ RTw = a.RT + win * (long)a.rt_rows * Kp;

    float acc[IB][JPL];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ib][jj] = 0.f;

    const float* Rp[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) {
        int j = jj * 64 + lane;
        j = j < K ? j : K - 1;
        Rp[jj] = RTw + j;
    }
    // rows past the end of the block are clamped duplicates; their results are dropped below
    int loff[NL];
    {
        const int n16 = lane & 15;
#pragma unroll
        for (int x = 0; x < NL; ++x) {
            const int i = 2 * x + (n16 >> 3);
            loff[x] = (i < nrows ? i : nrows - 1) * ldl + (n16 & 7);
        }
    }
    if (PT > 0) {
        // 3-deep register ring over the k tiles: tile t+2 is requested before tile t is consumed, so two
        // tiles of VALU work (~2.5k cycles) plus the other resident waves cover the HBM/L2 latency
        constexpr int DEPTH = JPL <= 2 ? MTADGAT_ATTEND_DEPTH : 2;   // JPL >= 4: the tiles themselves fill the register file
        float rr[DEPTH][JPL][8], lr[DEPTH][NL];
        const int ntile = PT >> 3;
        auto fetch = [&](int st, int tile) {
            const int k1 = (tile < ntile ? tile : ntile - 1) << 3;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj)
#pragma unroll
                for (int e = 0; e < 8; ++e) rr[st][jj][e] = Rp[jj][(long)(k1 + e) * Kp];
#pragma unroll
            for (int x = 0; x < NL; ++x) lr[st][x] = Lrow0[loff[x] + k1];
        };
#pragma unroll
        for (int st = 0; st < DEPTH - 1; ++st) fetch(st, st);
        const int ptile = a.P8 >> 3;
        for (int t0 = 0; t0 < ntile; t0 += DEPTH) {
#pragma unroll
            for (int st = 0; st < DEPTH; ++st) {
                const int t = t0 + st;
                if (t < ntile) {
                    fetch((st + DEPTH - 1) % DEPTH, t + DEPTH - 1);
                    if (t < ptile)
                        attend_tile<JPL, IB, false>(acc, rr[st], lr[st]);
                    else
                        attend_tile<JPL, IB, true>(acc, rr[st], lr[st]);
                }
            }
        }
    }

    // scores -> softmax over j (reference modules.py:85-89 / :184-188); branch-free over rows
    float dj[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) dj[jj] = Rp[jj][(long)PT * Kp];
    const float cvec = Lrow0[(long)(lane < nrows ? lane : nrows - 1) * ldl + PT];   // lane ib holds c_ib
    float bv[IB][JPL];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
        const int irow = i0 + (ib < nrows ? ib : nrows - 1);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            int j = jj * 64 + lane;
            j = j < K ? j : K - 1;
            bv[ib][jj] = a.bias ? a.bias[(long)irow * K + j] : 0.f;
        }
    }
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
        const float ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cvec), ib));
        float e[JPL];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = jj * 64 + lane;
            float v = acc[ib][jj] + ci + dj[jj];
            if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
            v += bv[ib][jj];
            v = j < K ? v : -INFINITY;
            e[jj] = v;
            m = fmaxf(m, v);
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            e[jj] = (jj * 64 + lane < K) ? soft_exp(e[jj] - m) : 0.f;
            sum += e[jj];
        }
        sum = wave_sum(sum);
        const float inv = soft_rcp(sum);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ib][jj] = ib < nrows ? e[jj] * inv : 0.f;
    }
    if (a.ATT) {  // optional dump of the attention matrix (tests)
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
            if (ib < nrows)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = jj * 64 + lane;
                    if (j < K) a.ATT[(win * K + i0 + ib) * (long)K + j] = acc[ib][jj];
                }
    }

    // aggregation h_i = sigmoid(sum_j att_ij * V_j) on the matrix pipe (modules.py:93 / :191)
    const int i = lane & 31, g = lane >> 5;
    if (IB < 32) {
        for (int r = IB + g; r < 32; r += 2)
            for (int c = i; c < 68; c += 32) att_s[r][c] = 0.f;
    }
    const int DT = (a.D + 31) >> 5;
    const float* __restrict__ Vw = a.V + win * (long)K * a.ldv;
    for (int dt0 = 0; dt0 < DT; dt0 += 2) {
        f32x16 o[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nb][r] = 0.f;
        int dcl[2];
        bool dok[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int d = 32 * (dt0 + nb) + i;
            dok[nb] = d < a.D;
            dcl[nb] = dok[nb] ? d : a.D - 1;
        }
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            if (jj * 64 < K) {
                __syncthreads();
#pragma unroll
                for (int ib = 0; ib < IB; ++ib) att_s[ib][lane] = acc[ib][jj];
                __syncthreads();
                // rolled loop over the chunks of this 64-node block, operands of chunk q+1 fetched before
                // chunk q's MFMAs (att is 0 past K; V loads are clamped + masked, no divergent control flow)
                const int jn = min(64, K - jj * 64);
                const int nq = (jn + 7) >> 3;
                auto fetch = [&](int q, f32x4& bq, f32x4 (&av)[2]) {
                    bq = *reinterpret_cast<const f32x4*>(&att_s[i][8 * q + 4 * g]);
                    const int jb = jj * 64 + 8 * q + 4 * g;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int jc = jb + s < K ? jb + s : K - 1;
                            const float v = Vw[(long)jc * a.ldv + dcl[nb]];
                            av[nb][s] = (jb + s < K && dok[nb]) ? v : 0.f;
                        }
                };
                f32x4 bq, av[2];
                fetch(0, bq, av);
#pragma unroll 1
                for (int q = 0; q < nq; ++q) {
                    f32x4 bn, an[2];
                    fetch(q + 1 < nq ? q + 1 : q, bn, an);
                    o[0] = mfma4(av[0], bq, o[0]);
                    o[1] = mfma4(av[1], bq, o[1]);
                    bq = bn; av[0] = an[0]; av[1] = an[1];
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * (dt0 + nb) + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i < nrows && d < a.D)
                    a.out[win * a.so_w + (long)(i0 + i) * a.so_i + (long)d * a.so_d] = gate_sigmoid(o[nb][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// gat (fused): one workgroup per window does the whole graph-attention layer -- projection,
// pairwise scores, softmax, aggregation, sigmoid -- with the node features V, the projected L' and
// R' never leaving the CU.  Same algebra and packed weights as k_rowgemm + k_attend (which remain
// the path for node counts / dims whose tiles do not fit in LDS).
//   LDS:  Ls [NWA*4*IBL][34]     L' columns of the current part, row-major (+ c when it is in the part)
//         Rs [K][34]             R' columns of the current part, row-major (+ d)
//         att[NWA][4*IBL][68]    softmax rows restaged for the aggregation MFMA (aliases Ls/Rs)
//         Vs [Kp16][vld]         node feature rows of this window, zero padded rows / columns, column D = 1
//                                (the projection bias is weight row D)
// NWA = ceil(K / (4*IBL)) waves own query rows; a workgroup may have more waves than that (they only take
// projection tiles).  The embedding is processed in parts of one 32-column tile per side: MFMA phase
// (projection of the part into Ls/Rs, one 32-node tile per wave, weights requested a phase early) ->
// barrier -> VALU phase -> barrier.  Several workgroups per CU are in different phases, so the matrix and
// vector pipes overlap across workgroups.
//
// VALU phase = 2-D register blocking of the K x K pair grid.  A wave owns 4*IBL query rows; lane
// (li = lane>>4, lj = lane&15) accumulates the IBL x JPL pairs {rows li + 4 ii} x {keys lj + 16 jj}.
// Per 2 embedding columns it reads IBL + JPL 8-byte LDS words (its rows of L', its keys of R') for
// 4*IBL*JPL VALU instructions -- v_add_f32 t, l, r; v_add_f32 acc, acc, |t| -- so the LDS feeds
// ~0.17 floats per VALU op (lane-per-key with wave-uniform broadcast rows needed 0.28-0.53 and was
// LDS-return bound), no lane is spent on padding beyond 16*JPL keys, and all addresses are
// base + immediate.  The two register sets A/B alternate: the loads of the next column pair are in
// flight while the current pair is consumed.  Row strides of 34 floats keep every ds_read_b64 wave
// access conflict-free (16 distinct keys x 2 banks each cover 32 bank pairs).
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(3))) float* lds_cptr;      // explicit LDS pointer (32-bit)
constexpr int GAT_LLD = 34;     // 32 columns + 2: rows 8-byte aligned, 16 consecutive rows start on 16 distinct bank pairs
constexpr int GAT_APITCH = 68;

// lp[ii]: one base pointer per query row.  The pointers are made opaque to the compiler on purpose:
// with a common base it merges row pairs into ds_read2_b64, which runs at half the LDS rate of two
// ds_read_b64 (MI355X: 8 vs 2 x 2 LDS cycles per wave instruction).
template <int IBL, int JPL>
__device__ __forceinline__ void gat_load(f32x2 (&l)[IBL], f32x2 (&r)[JPL], const lds_cptr (&lp)[IBL], lds_cptr rp, int col) {
    typedef const __attribute__((address_space(3))) f32x2* lds_c2;
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) l[ii] = *(lds_c2)(lp[ii] + col);
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) r[jj] = *(lds_c2)(rp + jj * 16 * GAT_LLD + col);
}

// The two instructions per pair are written as (volatile) inline asm: left to itself the compiler packs
// the column pair into v_pk_add_f32 (no faster) and schedules all sums of a step
// ahead of their uses, which costs > 100 VGPRs of temporaries and spills the accumulators.
template <int IBL, int JPL, bool NEG>
__device__ __forceinline__ void gat_step(float (&acc)[IBL][JPL], const f32x2 (&l)[IBL], const f32x2 (&r)[JPL]) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            float t[JPL];
            const float lv = l[ii][e];
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const float rv = r[jj][e];
                asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(t[jj]) : "v"(lv), "v"(rv));
            }
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                if (NEG)
                    asm volatile("v_sub_f32_e64 %0, %0, |%1|" : "+v"(acc[ii][jj]) : "v"(t[jj]));
                else
                    asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(acc[ii][jj]) : "v"(t[jj]));
            }
        }
}

// one 8-column k tile; on entry set A holds columns 0,1 of the tile (loads possibly still in flight),
// on exit it holds columns 0,1 of the next tile (pad columns past the end of a part: never consumed)
template <int IBL, int JPL, bool NEG>
__device__ __forceinline__ void gat_tile(float (&acc)[IBL][JPL], f32x2 (&lA)[IBL], f32x2 (&rA)[JPL], f32x2 (&lB)[IBL],
                                         f32x2 (&rB)[JPL], const lds_cptr (&lp)[IBL], lds_cptr rp) {
    gat_load<IBL, JPL>(lB, rB, lp, rp, 2);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lA, rA);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL>(lA, rA, lp, rp, 4);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lB, rB);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL>(lB, rB, lp, rp, 6);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lA, rA);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL>(lA, rA, lp, rp, 8);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lB, rB);
    __builtin_amdgcn_sched_barrier(0);
}

// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    return fmaxf(v, dpp_move<0x140>(v));
}
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v + dpp_move<0x140>(v);
}

template <int IBL, int JPL>
__global__ __launch_bounds__(512, 4) void k_gat(const GatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IBW = 4 * IBL;                       // query rows per wave
    constexpr int QB = 8;                              // weight chunks held in registers per task batch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const long win = blockIdx.x;
    const int K = a.K, D = a.D, PT = a.PT;
    const int vld = a.vld;
    const int Kp16 = (K + 15) & ~15;                   // rows of Vs: real nodes then zero rows
    const int NWA = (K + IBW - 1) / IBW;               // waves that own query rows (the rest only project)
    float* __restrict__ Ls = smem;
    float* __restrict__ Rs = Ls + NWA * IBW * GAT_LLD;     // K rows; lanes whose keys are >= K read on into Vs (never used)
    float* __restrict__ Vs = smem + a.lr_floats;
    const int i = lane & 31, g = lane >> 5;            // MFMA roles
    const int lj = lane & 15, li = lane >> 4;          // pair-grid roles

    const int NTn = (K + 31) >> 5;                    // node tiles
    const int ntask = 2 * NTn;                        // per part: query-side tiles then key-side tiles
    const int Q = a.Q;
    const int ptile = a.P8 >> 3, ntile = PT >> 3;
    const int nparts = (PT >> 5) + 1;                 // the part holding column PT (c, d) is the last one with content

    // The weights of this wave's first task of a part are requested one phase early -- before the barrier
    // that ends the previous VALU phase, for part 0 before the window is staged -- so the L2 round trip is
    // not on the critical path of the MFMA phase.  (The projection bias is row D of the packed weights,
    // multiplied by a constant-one column of Vs: no separate bias loads.)
    f32x4 w[QB];
    auto prefetch = [&](int part) {
        if (wave < ntask) {
            const int wtile = wave >= NTn ? a.NT_L + part : part;
            const f32x4* __restrict__ wp = a.Wp + ((long)wtile * Q) * 64 + lane;
#pragma unroll
            for (int u = 0; u < QB; ++u) w[u] = wp[(long)(u < Q ? u : Q - 1) * 64];
        }
    };

    // ---- stage the window's node rows (coalesced global reads), zero the padding rows / columns, set
    // the ones column D.  vt == 0: node rows are source rows (temporal layer: V = xc).  vt == 1: nodes
    // are the source's columns (feature layer: V = xc^T), transposed on the way into LDS.  All global
    // loads of a thread are issued before the first LDS store: one memory round trip per window, not one
    // per loop iteration.
    {
        const int nthr = blockDim.x;
        const int srows = a.vt ? D : K, scols = a.vt ? K : D;          // valid extent of the source block
        const int prow = a.vt ? vld : Kp16, pcol = a.vt ? Kp16 : vld;   // extent incl. the padding that must be written
        if ((a.ldv & 3) == 0 && ((scols + 3) & ~3) <= a.ldv) {
            // unit u = one float4 of a source row: row = u / p4 (exact through the float reciprocal: the
            // fractional part of (u + 0.5) / p4 stays >= 0.5 / p4 away from an integer).  Few, wide load
            // instructions: the cost of this phase is per load instruction, not per byte.
            const float* __restrict__ vsrc = a.V + win * (long)srows * a.ldv;
            constexpr int MAXU = 8;
            const int p4 = pcol >> 2, total = prow * p4;
            const float rinv = 1.0f / (float)p4;
            const int c4last = ((scols - 1) >> 2) << 2;
            f32x4 v[MAXU];
            int rr[MAXU], cc[MAXU];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr;
                const int row = (int)(((float)u + 0.5f) * rinv), c4 = (u - row * p4) * 4;
                rr[n] = u < total ? row : -1;
                cc[n] = c4;
                // unconditional load from a clamped (always valid) address, masked below: a guarded load
                // becomes a branch with s_waitcnt vmcnt(0) at the join, i.e. one round trip per unit
                const int rc = row < srows ? row : srows - 1, cl = c4 < scols ? c4 : c4last;
                v[n] = *reinterpret_cast<const f32x4*>(vsrc + (long)rc * a.ldv + cl);
            }
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int row = rr[n], c4 = cc[n];
                if (row >= 0) {
                    f32x4 t = v[n];
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int node = a.vt ? c4 + s4 : row, col = a.vt ? row : c4 + s4;
                        t[s4] = (node < K && col < D) ? t[s4] : ((node < K && col == D) ? 1.f : 0.f);
                    }
                    if (!a.vt) {
                        *reinterpret_cast<f32x4*>(Vs + row * vld + c4) = t;
                    } else {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) Vs[(c4 + s4) * vld + row] = t[s4];
                    }
                }
            }
            for (int u = tid + MAXU * nthr; u < total; u += nthr) {     // shapes beyond the register batch
                const int row = u / p4, c4 = (u - row * p4) * 4;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int node = a.vt ? c4 + s4 : row, col = a.vt ? row : c4 + s4;
                    const float t = (node < K && col < D) ? vsrc[(long)row * a.ldv + c4 + s4] : ((node < K && col == D) ? 1.f : 0.f);
                    Vs[node * vld + col] = t;
                }
            }
        } else {
            // unaligned caller tensor (stage entry point mtadgat_gat)
            const float* __restrict__ vsrc = a.V + win * (long)srows * a.ldv;
            for (int u = tid; u < Kp16 * vld; u += nthr) {
                const int node = u / vld, col = u - node * vld;
                float t = 0.f;
                if (node < K && col < D) t = a.vt ? vsrc[(long)col * a.ldv + node] : vsrc[(long)node * a.ldv + col];
                Vs[u] = (node < K && col == D) ? 1.f : t;
            }
        }
    }
    prefetch(0);
    __syncthreads();

    const bool rows_owner = wave < NWA;
    const int i0 = (rows_owner ? wave : 0) * IBW;
    lds_cptr lp[IBL];                                            // this lane's rows: i0 + li + 4 ii
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        lp[ii] = (lds_cptr)(Ls + (i0 + li + 4 * ii) * GAT_LLD);
        asm volatile("" : "+v"(lp[ii]));
    }
    const lds_cptr rp = (lds_cptr)(Rs + lj * GAT_LLD);           // this lane's keys: lj + 16 jj
    float acc[IBL][JPL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = 0.f;

    for (int part = 0; part < nparts; ++part) {
        // ---- MFMA phase: project this part's 32 + 32 columns for all nodes into Ls / Rs
        for (int task = wave; task < ntask; task += NW) {
            const bool keyside = task >= NTn;
            const int nt = keyside ? task - NTn : task;
            const int wtile = keyside ? a.NT_L + part : part;
            const int node = nt * 32 + i;
            const float* __restrict__ vrow = Vs + (node < K ? node : K - 1) * vld;
            const f32x4* __restrict__ wp = a.Wp + ((long)wtile * Q) * 64 + lane;
            if (task != wave) {                    // more tiles than waves: later tasks pay their own round trip
#pragma unroll
                for (int u = 0; u < QB; ++u) w[u] = wp[(long)(u < Q ? u : Q - 1) * 64];
            }
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            for (int qb = 0; qb < Q; qb += QB) {
#pragma unroll
                for (int u = 0; u < QB; ++u)
                    if (qb + u < Q) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(vrow + 8 * (qb + u) + 4 * g);
                        o = mfma4(w[u], xv, o);
                        // the chunk QB further on replaces this one as soon as it has been issued
                        if (qb + QB + u < Q) w[u] = wp[(long)(qb + QB + u) * 64];
                    }
            }
            if (node < K) {
                float* __restrict__ dst = (keyside ? Rs : Ls) + node * GAT_LLD + 4 * g;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x2 v0, v1;
                    v0[0] = o[4 * m + 0]; v0[1] = o[4 * m + 1]; v1[0] = o[4 * m + 2]; v1[1] = o[4 * m + 3];
                    *reinterpret_cast<f32x2*>(dst + 8 * m) = v0;
                    *reinterpret_cast<f32x2*>(dst + 8 * m + 2) = v1;
                }
            }
        }
        __syncthreads();
        // ---- VALU phase: pairwise term over this part's k tiles (positive group first, then negative)
        int ntl = ntile - 4 * part;
        ntl = ntl > 4 ? 4 : ntl;
        if (ntl > 0 && rows_owner) {
            f32x2 lA[IBL], rA[JPL], lB[IBL], rB[JPL];
            lds_cptr lq[IBL];
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii) lq[ii] = lp[ii];
            lds_cptr rq = rp;
            gat_load<IBL, JPL>(lA, rA, lq, rq, 0);
            int npos = ptile - 4 * part;
            npos = npos < 0 ? 0 : (npos > ntl ? ntl : npos);
            int kt = 0;
            // two back-to-back loops rather than a sign branch inside one: with the diamond the compiler
            // keeps two register copies of the accumulators (and spills)
#pragma unroll 1
            for (; kt < npos; ++kt) {
                gat_tile<IBL, JPL, false>(acc, lA, rA, lB, rB, lq, rq);
#pragma unroll
                for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                rq += 8;
            }
#pragma unroll 1
            for (; kt < ntl; ++kt) {
                gat_tile<IBL, JPL, true>(acc, lA, rA, lB, rB, lq, rq);
#pragma unroll
                for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                rq += 8;
            }
        }
        if (part + 1 < nparts) {
            prefetch(part + 1);
            __syncthreads();
        }
    }
    // rank-1 terms c_i (query column PT) and d_j (key column PT) sit in the last part
    float cv[IBL], dv[JPL];
    {
        const int col = PT & 31;
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) cv[ii] = lp[ii][col];
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) dv[jj] = rp[jj * 16 * GAT_LLD + col];
    }
    __syncthreads();
    if (!rows_owner) return;                           // no barrier below this point

    // ---- scores -> softmax over j (reference modules.py:85-89 / :184-188); a query row lives in
    // the 16 lanes of one DPP row (x JPL registers), so the reductions are row-local DPP butterflies
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        const int irow = i0 + li + 4 * ii;
        const int irc = irow < K ? irow : K - 1;
        float e[JPL];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = lj + 16 * jj;
            const float b = a.bias ? a.bias[(long)irc * K + (j < K ? j : K - 1)] : 0.f;
            float v = acc[ii][jj] + cv[ii] + dv[jj];
            if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
            v += b;
            v = j < K ? v : -INFINITY;
            e[jj] = v;
            m = fmaxf(m, v);
        }
        m = row_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            e[jj] = (lj + 16 * jj < K) ? soft_exp(e[jj] - m) : 0.f;
            sum += e[jj];
        }
        sum = row_sum(sum);
        const float inv = soft_rcp(sum);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = irow < K ? e[jj] * inv : 0.f;
    }

    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j) on the matrix pipe, as out^T = V^T att^T with
    // v_mfma_f32_16x16x4_f32 (M = 16 output features, N = this wave's 16 query rows, 4 keys per
    // instruction): no padding to 32 rows, and each lane ends up with 4 consecutive features of one row.
    // att is restaged, 64 keys at a time, through this wave's slice of the (now free) Ls/Rs region.
    //   B operand: lane (n = lane&15, kb = lane>>4) = att[row n][key 4 kb + t]   (16-byte LDS read = 4 steps t)
    //   A operand: lane (m = lane&15, kb)           = V[key 4 kb + t][16 dt + m]
    //   D: register r of lane (n, mb = lane>>4)     = out[row n][16 dt + 4 mb + r]
    static_assert(IBL == 4, "one 16-row MFMA group per wave");
    constexpr int DTMAX = 8;                           // D <= 128 (plan)
    float* __restrict__ att = Ls + wave * (IBW * GAT_APITCH);
    const int DT = (D + 15) >> 4;
    const int nr = lane & 15, kb = lane >> 4;
    constexpr int PASSES = (JPL + 3) / 4;
    f32x4 o[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int dcol[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) {
        const int d = 16 * dt + nr;
        dcol[dt] = d < vld ? d : vld - 1;              // columns > D of Vs are zero
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass * 64 < K) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4)
                    if (4 * pass + j4 < JPL) att[(li + 4 * ii) * GAT_APITCH + lj + 16 * j4] = acc[ii][4 * pass + j4];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int jn = min(64, K - pass * 64);
            const int ngrp = (jn + 15) >> 4;                   // rows < Kp16 of Vs: real or zero
            for (int grp = 0; grp < ngrp; ++grp) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(att + nr * GAT_APITCH + 16 * grp + 4 * kb);
                const float* __restrict__ vk = Vs + (pass * 64 + 16 * grp + 4 * kb) * vld;
#pragma unroll
                for (int dt = 0; dt < DTMAX; ++dt)
                    if (dt < DT) {
                        float av[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) av[t] = vk[t * vld + dcol[dt]];
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bq[t], o[dt], 0, 0, 0);
                    }
            }
        }
    }
    {
        const int row = i0 + nr;
        float* __restrict__ orow = a.out + win * a.so_w + (long)row * a.so_i;
#pragma unroll
        for (int dt = 0; dt < DTMAX; ++dt)
            if (dt < DT) {
                const int d0 = 16 * dt + 4 * kb;
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = gate_sigmoid(o[dt][r]);
                if (a.so_d == 1 && row < K && d0 + 3 < D) {
                    // 4 consecutive features of one row: one 16-byte store (dword aligned is enough for global memory)
                    typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
                    *reinterpret_cast<f32x4_a4*>(orow + d0) = y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row < K && d0 + r < D) orow[(long)(d0 + r) * a.so_d] = y[r];
                }
            }
    }
}

// ---------------------------------------------------------------------------
// GRU: 32 windows per wave, hidden state resident in registers in F-layout for
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
// pipe 87 % vs 82 % busy on the GRU layer): the MFMAs of one wave issue back to back, interleaving two
// waves leaves bubbles; each weight chunk is also fetched once for 64 windows.
template <int NCG, int XMODE, bool FC, int DROP, int MW>
__global__ __launch_bounds__(64, ((NCG <= 6 && MW == 1) ? 2 : 1)) void k_gru(const GruArgs a) {
    __shared__ float hn_s[MW][NCG][16][64];
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    long win[MW], winc[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) {
        win[w] = ((long)blockIdx.x * MW + w) * 32 + i;
        winc[w] = win[w] < a.B ? win[w] : a.B - 1;
    }
    const int T = a.T, Qx = a.Qx;
    const int Qxp = (XMODE == 1) ? 1 : a.Qxp;
    constexpr int Qh = 4 * NCG;                   // recurrent chunks that can be non-zero
    constexpr int Qhe = Qh - DROP;                // ... and as used
    constexpr int ROT = (XMODE == 1) ? (1 + Qhe) % 3 : Qhe % 3;   // ring phase advance per hidden tile
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

    f32x4 wr[3][3], xr[3][MW];
    wload(wr[0]); wload(wr[1]); wload(wr[2]);
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int w = 0; w < MW; ++w) xr[st][w] = loadx_t(w, 0, st);

    for (int t = 0; t < T; ++t) {
        for (int c = 0; c < NCG; ++c) {
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
                for (int w = 0; w < MW; ++w) mfma4x3(wr[0], xr[0][w], ar[w], az[w], anx[w]);
                wload(wr[0]);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                for (int q0 = 0; q0 < Qxp; q0 += 3) {
#pragma unroll
                    for (int st = 0; st < 3; ++st) {
#pragma unroll
                        for (int w = 0; w < MW; ++w) mfma4x3(wr[st], xr[st][w], ar[w], az[w], anx[w]);
                        wload(wr[st]);
#pragma unroll
                        for (int w = 0; w < MW; ++w) xr[st][w] = loadx_t(w, t, q0 + st + 3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- recurrent part: W_h{r,z,n} h_{t-1}  (h_0 = 0 contributes nothing at t = 0; kept so the
            // weight stream stays continuous).  Ring stage of h chunk q is static: (x chunks + q) % 3.
#pragma unroll
            for (int q = 0; q < Qhe; ++q) {
                constexpr int X0 = (XMODE == 1) ? 1 : 0;
                const int cq = q >> 2, m = q & 3, st = (X0 + q) % 3;
#pragma unroll
                for (int w = 0; w < MW; ++w) {
                    f32x4 hv;
                    hv[0] = h[w][cq][4 * m + 0]; hv[1] = h[w][cq][4 * m + 1];
                    hv[2] = h[w][cq][4 * m + 2]; hv[3] = h[w][cq][4 * m + 3];
                    mfma4x3(wr[st], hv, ar[w], az[w], anh[w]);
                }
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bring the ring back to phase 0 for the next tile: a compile-time register renaming
            if (ROT == 1) {
#pragma unroll
                for (int u = 0; u < 3; ++u) { const f32x4 t0 = wr[0][u]; wr[0][u] = wr[1][u]; wr[1][u] = wr[2][u]; wr[2][u] = t0; }
            } else if (ROT == 2) {
#pragma unroll
                for (int u = 0; u < 3; ++u) { const f32x4 t0 = wr[0][u]; wr[0][u] = wr[2][u]; wr[2][u] = wr[1][u]; wr[1][u] = t0; }
            }
            // x chunks 0..2 of the next tile / step: their latency hides under the gate math
            {
                const int tn = (c == NCG - 1) ? (t + 1 < T ? t + 1 : t) : t;
#pragma unroll
                for (int st = 0; st < 3; ++st)
#pragma unroll
                    for (int w = 0; w < MW; ++w) xr[st][w] = loadx_t(w, tn, st);
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
                    ar[w][r] = (1.0f - zg) * ng + zg * hold;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) hn_s[w][c][r][lane] = ar[w][r];
            }
        }
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
                    for (int q = 0; q < Qhe; ++q) {
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
                    const f32x4* __restrict__ wp = a.Wfc + ((long)n * Qh) * 64 + lane;
#pragma unroll
                    for (int q = 0; q < Qhe; ++q) {
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
template <int XMODE, bool FC>
__global__ __launch_bounds__(512, 3) void k_gru_split(const GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NCG = blockDim.x >> 6;
    const int i = lane & 31, g = lane >> 5;
    const long win = (long)blockIdx.x * 32 + i;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Qx = a.Qx;
    const int Qxp = (XMODE == 1) ? 1 : a.Qxp;      // XMODE 0/2: a multiple of 3
    const int Qh = 4 * NCG;                        // recurrent chunks of a tile that can be non-zero
    const int Qhe = (a.H + 7) >> 3;                // ... as needed
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
    const float* __restrict__ xbase = (XMODE == 0) ? a.X + winc * T * a.ldx + 4 * g : a.X + winc * a.ldx;
    auto loadx_t = [&](int t, int q) -> f32x4 {
        const int qq = q < Qx ? q : Qx - 1;       // padded chunks re-read the last real one (their weights are zero)
        if (XMODE == 0) return *reinterpret_cast<const f32x4*>(xbase + (long)t * a.ldx + 8 * qq);
        const int k0 = a.m0[t] + 8 * qq + 4 * g, kmax = (int)a.ldx - 1;     // stay inside the (zero padded) row
        f32x4 v;
        v[0] = xbase[min(k0, kmax)]; v[1] = xbase[min(k0 + 1, kmax)];
        v[2] = xbase[min(k0 + 2, kmax)]; v[3] = xbase[min(k0 + 3, kmax)];
        return v;
    };
    auto hread = [&](int q) -> f32x4 { return hs[(q < Qhe ? q : Qhe - 1) * 64 + lane]; };   // padding: any finite chunk

    f32x16 hown;                                   // this wave's tile of h
#pragma unroll
    for (int r = 0; r < 16; ++r) hown[r] = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) hs[(c * 4 + m) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wr[3][3], xr[3];
    wload(wr[0]); wload(wr[1]); wload(wr[2]);
#pragma unroll
    for (int st = 0; st < 3; ++st) xr[st] = loadx_t(0, st);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        f32x16 ar, az, anx, anh;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int col = 32 * c + 8 * m + 4 * g;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + col);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + a.Hp + col);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias + 2 * a.Hp + col);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(a.bias + 3 * a.Hp + col);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                ar[4 * m + s4] = b0[s4];
                az[4 * m + s4] = b1[s4];
                anx[4 * m + s4] = b2[s4];
                anh[4 * m + s4] = b3[s4];
            }
        }
        // chunk q of h_{t-1} is requested one chunk ahead of its MFMAs (LDS latency under the previous group)
        f32x4 hv = hread(0);
        int qh = 0;                                // next h chunk to consume
        if (XMODE == 1) {
            // first ring turn: the single x chunk, then h chunks 0 and 1
            mfma4x3(wr[0], xr[0], ar, az, anx);
            wload(wr[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 1; st < 3; ++st) {
                const f32x4 hn = hread(qh + 1);
                mfma4x3(wr[st], hv, ar, az, anh);
                wload(wr[st]);
                __builtin_amdgcn_sched_barrier(0);
                hv = hn; ++qh;
            }
        } else {
            for (int q0 = 0; q0 < Qxp; q0 += 3) {
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    mfma4x3(wr[st], xr[st], ar, az, anx);
                    wload(wr[st]);
                    xr[st] = loadx_t(t, q0 + st + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        auto hturn = [&]() {                       // one ring turn of h chunks qh .. qh + 2
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const f32x4 hn = hread(qh + st + 1);
                mfma4x3(wr[st], hv, ar, az, anh);
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
        // x chunks 0..2 of the next step: their latency hides under the gate math
        {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int st = 0; st < 3; ++st) xr[st] = loadx_t(tn, st);
        }
        // ---- gates (reference GRULayer / RNNDecoder: torch.nn.GRU equations, r|z|n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rg = gate_sigmoid(ar[r]);
            const float zg = gate_sigmoid(az[r]);
            const float ng = gate_tanh(anx[r] + rg * anh[r]);
            hown[r] = (1.0f - zg) * ng + zg * hown[r];
        }
        f32x4 hvv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            hvv[m][0] = hown[4 * m + 0]; hvv[m][1] = hown[4 * m + 1]; hvv[m][2] = hown[4 * m + 2]; hvv[m][3] = hown[4 * m + 3];
        }
        __syncthreads();                           // B: every wave is done reading h_{t-1}
#pragma unroll
        for (int m = 0; m < 4; ++m) hs[(c * 4 + m) * 64 + lane] = hvv[m];
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

// small helper: copy a (R, ncols) row-major matrix into a padded (R, ld) one, or back.
__global__ void k_copy2d(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long R, int ncols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = R * ncols;
    if (idx < total) {
        const long r = idx / ncols;
        const int c = (int)(idx - r * ncols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}
// transpose per window: src (B, R, C) with row stride lds -> dst (B, C, R) with row stride ldd
__global__ void k_transpose_win(const float* __restrict__ src, long lds, float* __restrict__ dst, long ldd, long B, int R,
                                int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = B * R * C;
    if (idx < total) {
        const long b = idx / ((long)R * C);
        const long rem = idx - b * (long)R * C;
        const int r = (int)(rem / C);
        const int c = (int)(rem - (long)r * C);
        dst[(b * C + c) * ldd + r] = src[(b * R + r) * lds + c];
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int launch_rowgemm(const RowGemmArgs& a, hipStream_t s) {
    if (a.R <= 0) return 0;
    const unsigned grid = (unsigned)((a.R + 31) / 32);
    if (a.NT >= 4)
        hipLaunchKernelGGL(k_rowgemm<4>, dim3(grid), dim3(64), 0, s, a);
    else if (a.NT >= 2)
        hipLaunchKernelGGL(k_rowgemm<2>, dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(k_rowgemm<1>, dim3(grid), dim3(64), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_conv(const ConvArgs& a, hipStream_t s) {
    const long R = a.B * a.W;
    if (R <= 0) return 0;
    const unsigned grid = (unsigned)((R + 31) / 32);
    const size_t lds = (size_t)(32 + a.taps - 1) * (a.Fp + 4) * sizeof(float);
    if (lds <= 20 * 1024) {       // >= 8 waves per CU keep their tile in LDS
        const unsigned wpb = (grid >= 4096 && 4 * lds <= 64 * 1024) ? 4 : 1;     // waves per workgroup
        const unsigned g4 = (grid + wpb - 1) / wpb;
        if (a.NT >= 2)
            hipLaunchKernelGGL(k_conv_lds<2>, dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
        else
            hipLaunchKernelGGL(k_conv_lds<1>, dim3(g4), dim3(64 * wpb), wpb * lds, s, a);
    } else if (a.NT >= 2)
        hipLaunchKernelGGL(k_conv<2>, dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(k_conv<1>, dim3(grid), dim3(64), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

// Split the K query nodes of a window into nblk blocks of <= rows_per_blk rows, one wave each,
// and pick the kernel's unrolled row count IB >= rows_per_blk that wastes the fewest rows.
void attend_plan(int K, int* rows_per_blk, int* nblk, int* IB) {
    const int jpl = (K + 63) / 64;
    const int ibmax = jpl <= 1 ? 32 : (jpl <= 2 ? 20 : (jpl <= 4 ? 16 : 8));   // register budget (no spills)
    const int step = jpl <= 2 ? 4 : 8;
    const int nb0 = (K + ibmax - 1) / ibmax;
    long best = -1;
    for (int nb = nb0; nb <= nb0 + 3; ++nb) {
        const int rows = (K + nb - 1) / nb;
        int ib = ((rows + step - 1) / step) * step;
        if (ib < 8) ib = 8;
        const long cost = (long)ib * nb;
        if (best < 0 || cost < best) {
            best = cost;
            *rows_per_blk = rows;
            *nblk = (K + rows - 1) / rows;
            *IB = ib;
        }
    }
}

#define ATTEND_CASE(J, I)                                                             \
    if (jpl == J && IB == I) {                                                        \
        hipLaunchKernelGGL((k_attend<J, I>), dim3(grid), dim3(64), 0, s, a);          \
        launched = true;                                                              \
    }

int launch_attend(const AttendArgs& a, int IB, hipStream_t s) {
    if (a.total_blocks <= 0) return 0;
    int jpl = (a.K + 63) / 64;
    if (jpl == 3) jpl = 4;
    if (jpl > 4 && jpl <= 8) jpl = 8;
    const unsigned grid = (unsigned)a.total_blocks;
    bool launched = false;
    ATTEND_CASE(1, 8) ATTEND_CASE(1, 12) ATTEND_CASE(1, 16) ATTEND_CASE(1, 20)
    ATTEND_CASE(1, 24) ATTEND_CASE(1, 28) ATTEND_CASE(1, 32)
    ATTEND_CASE(2, 8) ATTEND_CASE(2, 12) ATTEND_CASE(2, 16) ATTEND_CASE(2, 20)
    ATTEND_CASE(2, 24) ATTEND_CASE(2, 28) ATTEND_CASE(2, 32)
    ATTEND_CASE(4, 8) ATTEND_CASE(4, 16)
    ATTEND_CASE(8, 8)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}

#define GAT_CASE(I, J)                                                                          \
    if (IBL == I && JPL == J) {                                                                 \
        if (lds_bytes > 64 * 1024) {                                                            \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat<I, J>),    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e_ != hipSuccess) return (int)e_;                                               \
        }                                                                                       \
        hipLaunchKernelGGL((k_gat<I, J>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);          \
        launched = true;                                                                        \
    }

// IBL: query rows per lane (a wave owns 4*IBL rows), JPL: key nodes per lane (16*JPL >= K), nw waves
int launch_gat(const GatArgs& a, int IBL, int JPL, int nw, size_t lds_bytes, hipStream_t s) {
    if (a.nwin <= 0) return 0;
    if (16 * JPL < a.K || nw * 4 * IBL < a.K || nw > 8) return -2;     // nw may exceed the row-owning waves: the rest only project
    const unsigned grid = (unsigned)a.nwin;
    bool launched = false;
    GAT_CASE(4, 1) GAT_CASE(4, 2) GAT_CASE(4, 3) GAT_CASE(4, 4) GAT_CASE(4, 5) GAT_CASE(4, 6) GAT_CASE(4, 7) GAT_CASE(4, 8)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}

template <int NCG, int XMODE, int MW>
static int launch_gru_mode(const GruArgs& a, bool fc, int drop, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + 32 * MW - 1) / (32 * MW));
    if (!fc && drop == 0)
        hipLaunchKernelGGL((k_gru<NCG, XMODE, false, 0, MW>), dim3(grid), dim3(64), 0, s, a);
    else if (!fc)
        hipLaunchKernelGGL((k_gru<NCG, XMODE, false, 1, MW>), dim3(grid), dim3(64), 0, s, a);
    else if (drop == 0)
        hipLaunchKernelGGL((k_gru<NCG, XMODE, true, 0, MW>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((k_gru<NCG, XMODE, true, 1, MW>), dim3(grid), dim3(64), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

template <int NCG>
static int launch_gru_ncg(const GruArgs& a, int xmode, bool fc, bool two, hipStream_t s) {
    // trailing recurrent chunks that are pure padding: skip one when H <= 8*(4*NCG - 1)
    const int drop = (a.H <= 8 * (4 * NCG - 1)) ? 1 : 0;
    if constexpr (NCG <= 5) {           // two 32-window groups per wave: 8 KB of LDS per group and tile, 4 waves per CU
        if (two) {
            if (xmode == 0) return launch_gru_mode<NCG, 0, 2>(a, fc, drop, s);
            if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 2>(a, fc, drop, s);
            return launch_gru_mode<NCG, 2, 2>(a, fc, drop, s);
        }
    }
    if (xmode == 0) return launch_gru_mode<NCG, 0, 1>(a, fc, drop, s);
    if (a.Qxp == 1) return launch_gru_mode<NCG, 1, 1>(a, fc, drop, s);
    return launch_gru_mode<NCG, 2, 1>(a, fc, drop, s);
}

static int launch_gru_split(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + 31) / 32);
    const size_t lds = ((size_t)ncg * 1024 + (fc ? (size_t)ncg * a.out_dim * 32 : 0)) * sizeof(float);
    if (lds > 64 * 1024) return -2;
    const int xm = xmode == 0 ? 0 : (a.Qxp == 1 ? 1 : 2);
#define SPLIT_CASE(XM, F) if (xm == XM && fc == F) hipLaunchKernelGGL((k_gru_split<XM, F>), dim3(grid), dim3(64 * ncg), lds, s, a);
    SPLIT_CASE(0, false) SPLIT_CASE(0, true) SPLIT_CASE(1, false) SPLIT_CASE(1, true) SPLIT_CASE(2, false) SPLIT_CASE(2, true)
#undef SPLIT_CASE
    LAUNCH_CHECK();
    return 0;
}

int launch_gru(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (xmode == 0 && ((a.ldx & 3) != 0 || a.Qxp % 3 != 0)) return -2;
    if (xmode != 0 && a.Qxp != 1 && a.Qxp % 3 != 0) return -2;
    // Small batches: spread the 32-window groups over ncg waves each (k_gru_split) -- k_gru needs ~2 groups
    // per SIMD to fill the machine and leaves it mostly idle below that.  Measured on MI355X (W=100, F=55,
    // H=150, GRU + decoder): 256 windows 12.0 -> 4.9 ms, 16 k windows 12.2 -> 9.9 ms, 32 k windows 12.2 vs 19.6
    // (the 5 waves of a group land 2/1/1/1 on the SIMDs, so the split form loses once the machine is full).
    {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
                n_cu = 256;
        }
        const long groups = (a.B + 31) / 32;
        const size_t lds = ((size_t)ncg * 1024 + (fc ? (size_t)ncg * a.out_dim * 32 : 0)) * sizeof(float);
        if (ncg >= 2 && groups <= 2L * n_cu && lds <= 64 * 1024) return launch_gru_split(a, ncg, xmode, fc, s);
    }
    // two groups per wave once that still gives every SIMD a wave
    static int n_cu2 = 0;
    if (!n_cu2) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu2, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu2 <= 0)
            n_cu2 = 256;
    }
    const bool two = (a.B + 31) / 32 >= 8L * n_cu2;
    switch (ncg) {
        case 1: return launch_gru_ncg<1>(a, xmode, fc, two, s);
        case 2: return launch_gru_ncg<2>(a, xmode, fc, two, s);
        case 3: return launch_gru_ncg<3>(a, xmode, fc, two, s);
        case 4: return launch_gru_ncg<4>(a, xmode, fc, two, s);
        case 5: return launch_gru_ncg<5>(a, xmode, fc, two, s);
        case 6: return launch_gru_ncg<6>(a, xmode, fc, two, s);
        case 7: return launch_gru_ncg<7>(a, xmode, fc, two, s);
        case 8: return launch_gru_ncg<8>(a, xmode, fc, two, s);
        default: return -2;
    }
}

int launch_copy2d(const float* src, long lds, float* dst, long ldd, long R, int ncols, hipStream_t s) {
    const long total = R * ncols;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_copy2d, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds, dst, ldd, R, ncols);
    LAUNCH_CHECK();
    return 0;
}
int launch_transpose_win(const float* src, long lds, float* dst, long ldd, long B, int R, int C, hipStream_t s) {
    const long total = B * R * C;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_transpose_win, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lds, dst, ldd, B, R,
                       C);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
