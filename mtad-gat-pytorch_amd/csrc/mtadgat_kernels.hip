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

#include "mtadgat_kernels.h"

namespace mtadgat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma4(const f32x4 w, const f32x4 x, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[3], acc, 0, 0, 0);
    return acc;
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
template <int JPL, int IB, bool NEG>
__device__ __forceinline__ void attend_tile(float (&acc)[IB][JPL], const float* __restrict__ Lrow0, const int (&loff)[IB],
                                            const float* (&Rp)[JPL], int k0) {
    f32x4 r0[JPL], r1[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) {
        r0[jj] = *reinterpret_cast<const f32x4*>(Rp[jj] + k0);
        r1[jj] = *reinterpret_cast<const f32x4*>(Rp[jj] + k0 + 4);
    }
    // branch-free over the IB query rows (rows past the block end are clamped duplicates whose
    // results are dropped) so the scalar loads of L' can be scheduled ahead of the VALU work
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
        const float* __restrict__ Lp = Lrow0 + loff[ib] + k0;  // wave-uniform -> s_load_dwordx8
        float l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = Lp[e];
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t0 = l[e] + r0[jj][e];
                const float t1 = l[4 + e] + r1[jj][e];
                if (NEG) {
                    acc[ib][jj] -= fabsf(t0);
                    acc[ib][jj] -= fabsf(t1);
                } else {
                    acc[ib][jj] += fabsf(t0);
                    acc[ib][jj] += fabsf(t1);
                }
            }
        }
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int JPL, int IB>
__global__ __launch_bounds__(64) void k_attend(const AttendArgs a) {
    __shared__ __attribute__((aligned(16))) float att_s[32][68];
    const int lane = threadIdx.x;
    const long blk = blockIdx.x;
    const long win = blk / a.nblk;
    const int rb = (int)(blk - win * a.nblk);
    const int i0 = rb * a.rows_per_blk;
    const int nrows = min(a.rows_per_blk, a.K - i0);
    const int K = a.K, ldo = a.ldo, PT = a.PT;
    const float* __restrict__ LR = a.LR;
    const float* __restrict__ Lrow0 = LR + (win * K + i0) * (long)ldo;

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
        Rp[jj] = LR + (win * K + j) * (long)ldo + PT;
    }
    int loff[IB];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) loff[ib] = (ib < nrows ? ib : nrows - 1) * ldo;
    for (int k0 = 0; k0 < a.P8; k0 += 8) attend_tile<JPL, IB, false>(acc, Lrow0, loff, Rp, k0);
    for (int k0 = a.P8; k0 < PT; k0 += 8) attend_tile<JPL, IB, true>(acc, Lrow0, loff, Rp, k0);

    // scores -> softmax over j (reference modules.py:85-89 / :184-188)
    float dj[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) dj[jj] = Rp[jj][PT + 1];
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
        if (ib < nrows) {
            const float ci = Lrow0[(long)ib * ldo + 2 * PT];
            float e[JPL];
            float m = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = jj * 64 + lane;
                float v = acc[ib][jj] + ci + dj[jj];
                if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
                if (j < K) {
                    if (a.bias) v += a.bias[(long)(i0 + ib) * K + j];
                } else {
                    v = -INFINITY;
                }
                e[jj] = v;
                m = fmaxf(m, v);
            }
            m = wave_max(m);
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                e[jj] = expf(e[jj] - m);
                s += e[jj];
            }
            s = wave_sum(s);
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) acc[ib][jj] = e[jj] / s;
        } else {
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) acc[ib][jj] = 0.f;
        }
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
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            if (jj * 64 < K) {
                __syncthreads();
#pragma unroll
                for (int ib = 0; ib < IB; ++ib) att_s[ib][lane] = acc[ib][jj];
                __syncthreads();
                const int jn = min(64, K - jj * 64);
                const int nq = (jn + 7) >> 3;
                for (int q = 0; q < nq; ++q) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(&att_s[i][8 * q + 4 * g]);
                    const int jb = jj * 64 + 8 * q + 4 * g;
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int d = 32 * (dt0 + nb) + i;
                        f32x4 av;
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            av[s] = (jb + s < K && d < a.D) ? Vw[(long)(jb + s) * a.ldv + d] : 0.f;
                        o[nb] = mfma4(av, bv, o[nb]);
                    }
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * (dt0 + nb) + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (i < nrows && d < a.D)
                    a.out[win * a.so_w + (long)(i0 + i) * a.so_i + (long)d * a.so_d] = sigmoidf_(o[nb][r]);
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
template <int NCG, int XMODE, bool FC>
__global__ __launch_bounds__(64) void k_gru(const GruArgs a) {
    __shared__ float hn_s[NCG][16][64];
    const int lane = threadIdx.x;
    const int i = lane & 31, g = lane >> 5;
    const long win = (long)blockIdx.x * 32 + i;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Qx = a.Qx;
    constexpr int Qh = 4 * NCG;
    const bool xvec = (a.ldx & 3) == 0;
    const f32x4* __restrict__ Wh = a.Wh;

    f32x16 h[NCG];
#pragma unroll
    for (int c = 0; c < NCG; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[c][r] = 0.f;

    for (int t = 0; t < T; ++t) {
        const float* __restrict__ xrow = (XMODE == 0) ? a.X + (winc * T + t) * a.ldx : a.X + winc * a.ldx;
        const int m0 = (XMODE == 1) ? a.m0[t] : 0;
        const f32x4* __restrict__ Wx = a.Wx + ((XMODE == 1) ? (long)t * NCG * Qx * 3 * 64 : 0);
        auto loadx = [&](int q) -> f32x4 {
            if (XMODE == 0) return load_feat4(xrow, 8 * q + 4 * g, a.Kx, xvec);
            return load_feat4(xrow, m0 + 8 * q + 4 * g, a.Kx, false);
        };

        for (int c = 0; c < NCG; ++c) {
            f32x16 ar, az, anx, anh;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int col = 32 * c + 8 * m + 4 * g;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + col);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + a.Hp + col);
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias + 2 * a.Hp + col);
                const f32x4 b3 = *reinterpret_cast<const f32x4*>(a.bias + 3 * a.Hp + col);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    ar[4 * m + s] = b0[s];
                    az[4 * m + s] = b1[s];
                    anx[4 * m + s] = b2[s];
                    anh[4 * m + s] = b3[s];
                }
            }
            // ---- input part: W_i{r,z,n} x_t
            {
                const f32x4* __restrict__ wp = Wx + ((long)c * Qx) * 3 * 64 + lane;
                f32x4 xv = loadx(0);
                f32x4 w0 = wp[0], w1 = wp[64], w2 = wp[128];
                for (int q = 0; q < Qx; ++q) {
                    const int qn = (q + 1 < Qx) ? q + 1 : q;
                    const f32x4 xn = loadx(qn);
                    const f32x4* __restrict__ wq = wp + (long)qn * 3 * 64;
                    const f32x4 n0 = wq[0], n1 = wq[64], n2 = wq[128];
                    ar = mfma4(w0, xv, ar);
                    az = mfma4(w1, xv, az);
                    anx = mfma4(w2, xv, anx);
                    xv = xn; w0 = n0; w1 = n1; w2 = n2;
                }
            }
            // ---- recurrent part: W_h{r,z,n} h_{t-1}   (h_0 = 0: skipped at t = 0)
            if (t > 0) {
                const f32x4* __restrict__ wp = Wh + ((long)c * Qh) * 3 * 64 + lane;
                f32x4 w0 = wp[0], w1 = wp[64], w2 = wp[128];
#pragma unroll
                for (int cq = 0; cq < NCG; ++cq) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int q = 4 * cq + m;
                        const int qn = (q + 1 < Qh) ? q + 1 : q;
                        const f32x4* __restrict__ wq = wp + (long)qn * 3 * 64;
                        const f32x4 n0 = wq[0], n1 = wq[64], n2 = wq[128];
                        f32x4 hv;
                        hv[0] = h[cq][4 * m + 0]; hv[1] = h[cq][4 * m + 1];
                        hv[2] = h[cq][4 * m + 2]; hv[3] = h[cq][4 * m + 3];
                        ar = mfma4(w0, hv, ar);
                        az = mfma4(w1, hv, az);
                        anh = mfma4(w2, hv, anh);
                        w0 = n0; w1 = n1; w2 = n2;
                    }
                }
            }
            // ---- gates.  h_old for this tile comes back from LDS (written at the end of step t-1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float hold = (t > 0) ? hn_s[c][r][lane] : 0.f;
                const float rg = sigmoidf_(ar[r]);
                const float zg = sigmoidf_(az[r]);
                const float ng = tanhf(anx[r] + rg * anh[r]);
                ar[r] = (1.0f - zg) * ng + zg * hold;
            }
            // every lane reads and writes only its own slots -> no cross-lane hazard
#pragma unroll
            for (int r = 0; r < 16; ++r) hn_s[c][r][lane] = ar[r];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCG; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[c][r] = hn_s[c][r][lane];

        if (a.Seq && win < a.B) {
            float* sp = a.Seq + (win * T + t) * a.ldseq;
#pragma unroll
            for (int c = 0; c < NCG; ++c)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    f32x4 v;
                    v[0] = h[c][4 * m + 0]; v[1] = h[c][4 * m + 1]; v[2] = h[c][4 * m + 2]; v[3] = h[c][4 * m + 3];
                    *reinterpret_cast<f32x4*>(sp + 32 * c + 8 * m + 4 * g) = v;
                }
        }
        if (FC) {
            for (int n = 0; n < a.NTfc; ++n) {
                f32x16 y;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bfc + 32 * n + 8 * m + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) y[4 * m + s] = bv[s];
                }
                const f32x4* __restrict__ wp = a.Wfc + ((long)n * Qh) * 64 + lane;
#pragma unroll
                for (int cq = 0; cq < NCG; ++cq)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        f32x4 hv;
                        hv[0] = h[cq][4 * m + 0]; hv[1] = h[cq][4 * m + 1];
                        hv[2] = h[cq][4 * m + 2]; hv[3] = h[cq][4 * m + 3];
                        y = mfma4(wp[(4 * cq + m) * 64], hv, y);
                    }
                if (win < a.B) {
                    float* yp = a.Yfc + (win * T + t) * (long)a.out_dim;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = 32 * n + (r & 3) + 8 * (r >> 2) + 4 * g;
                        if (o < a.out_dim) yp[o] = y[r];
                    }
                }
            }
        }
    }
    if (a.Hend && win < a.B) {
        float* hp = a.Hend + win * a.ldhe;
#pragma unroll
        for (int c = 0; c < NCG; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * g;
                if (j < a.H) hp[j] = h[c][r];
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
    if (a.NT >= 2)
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
    const int ibmax = jpl <= 2 ? 32 : (jpl <= 4 ? 16 : 8);
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

template <int NCG>
static int launch_gru_ncg(const GruArgs& a, int xmode, bool fc, hipStream_t s) {
    const unsigned grid = (unsigned)((a.B + 31) / 32);
    if (xmode == 0 && !fc)
        hipLaunchKernelGGL((k_gru<NCG, 0, false>), dim3(grid), dim3(64), 0, s, a);
    else if (xmode == 0 && fc)
        hipLaunchKernelGGL((k_gru<NCG, 0, true>), dim3(grid), dim3(64), 0, s, a);
    else if (xmode == 1 && !fc)
        hipLaunchKernelGGL((k_gru<NCG, 1, false>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((k_gru<NCG, 1, true>), dim3(grid), dim3(64), 0, s, a);
    LAUNCH_CHECK();
    return 0;
}

int launch_gru(const GruArgs& a, int ncg, int xmode, bool fc, hipStream_t s) {
    if (a.B <= 0) return 0;
    switch (ncg) {
        case 1: return launch_gru_ncg<1>(a, xmode, fc, s);
        case 2: return launch_gru_ncg<2>(a, xmode, fc, s);
        case 3: return launch_gru_ncg<3>(a, xmode, fc, s);
        case 4: return launch_gru_ncg<4>(a, xmode, fc, s);
        case 5: return launch_gru_ncg<5>(a, xmode, fc, s);
        case 6: return launch_gru_ncg<6>(a, xmode, fc, s);
        case 7: return launch_gru_ncg<7>(a, xmode, fc, s);
        case 8: return launch_gru_ncg<8>(a, xmode, fc, s);
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
