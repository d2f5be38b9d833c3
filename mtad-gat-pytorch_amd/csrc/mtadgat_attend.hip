// k_attend: un-fused graph attention for node counts beyond the fused kernel (K > 128) + launcher
#include "mtadgat_device.h"

namespace mtadgat {

#ifndef MTADGAT_ATTEND_DEPTH
#define MTADGAT_ATTEND_DEPTH 2
#endif

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
    const int K = a.K, ldl = a.ldl, PT = a.ord ? a.ord[1] : a.PT, Kp = a.Kp;
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
        const int ptile = (a.ord ? a.ord[0] : a.P8) >> 3;
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

}  // namespace mtadgat
