// k_gat: fused graph-attention layer, one workgroup per window + launcher
#include "mtadgat_gat_impl.h"

namespace mtadgat {

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
// (li = lane / RJ, lj = lane % RJ; RJ = 16, RI = 4 in the numbers below) accumulates the IBL x JPL pairs
// {rows li + RI ii} x {keys lj + RJ jj}.
// Per 2 embedding columns it reads IBL + JPL 8-byte LDS words (its rows of L', its keys of R') for
// 4*IBL*JPL VALU instructions -- v_add_f32 t, l, r; v_add_f32 acc, acc, |t| -- so the LDS feeds
// ~0.17 floats per VALU op (lane-per-key with wave-uniform broadcast rows needed 0.28-0.53 and was
// LDS-return bound), no lane is spent on padding beyond 16*JPL keys, and all addresses are
// base + immediate.  The two register sets A/B alternate: the loads of the next column pair are in
// flight while the current pair is consumed.  Row strides of 34 floats keep every ds_read_b64 wave
// access conflict-free (16 distinct keys x 2 banks each cover 32 bank pairs).
// ---------------------------------------------------------------------------
// RJ = lanes along the key axis (16, or 8 when that pads K less: 55 features -> 56 instead of 64 keys);
// RI = 64 / RJ lanes along the row axis; a wave owns RI*IBL = 16 query rows either way.
// BF: bf16 operand build of the projection (16 features per chunk, fp32 accumulation); the pair grid, softmax and
// aggregation are fp32 either way.
#ifndef MTADGAT_GAT_MINW
#define MTADGAT_GAT_MINW 4
#endif
// X3 (with BF): split-bf16 operands for the projection (mtadgat_device.h) -- three weight pieces per chunk, the node rows
// split where they are consumed: fp32-class L' / R' from the bf16 matrix pipe, which (unlike the fp32 MFMA) runs beside
// the pair grid of the other waves.
template <int IBL, int JPL, int RJ, bool BF = false, bool X3 = false>
__global__ __launch_bounds__(512, MTADGAT_GAT_MINW) void k_gat(const GatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RI = 64 / RJ;
    constexpr int IBW = RI * IBL;                      // query rows per wave
#ifndef MTADGAT_GAT_QB
#define MTADGAT_GAT_QB 8
#endif
#ifndef MTADGAT_GAT_QB3
#define MTADGAT_GAT_QB3 2
#endif
    constexpr int QB = X3 ? MTADGAT_GAT_QB3 : MTADGAT_GAT_QB;       // weight chunks held in registers per task batch
    constexpr int NP = X3 ? 3 : 1;                     // 16-byte words per weight chunk and lane
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const long win = blockIdx.x;
    const int K = a.K, D = a.D;
    int P8, PT;
    gat_load_order(a.ord, a.P8, a.PT, P8, PT);
    const int vld = a.vld;
    const int Kp16 = (K + 15) & ~15;                   // rows of Vs: real nodes then zero rows
    const int NWA = (K + IBW - 1) / IBW;               // waves that own query rows (the rest only project)
    float* __restrict__ Ls = smem;
    float* __restrict__ Rs = Ls + NWA * IBW * GAT_LLD;     // K rows; lanes whose keys are >= K read on into Vs (never used)
    float* __restrict__ Vs = smem + a.lr_floats;
    const int i = lane & 31, g = lane >> 5;            // MFMA roles
    const int lj = lane % RJ, li = lane / RJ;          // pair-grid roles

    const int NTn = (K + 31) >> 5;                    // node tiles
    const int ntask = 2 * NTn;                        // per part: query-side tiles then key-side tiles
    const int Q = a.Q;
    const int ptile = P8 >> 3, ntile = PT >> 3;
    const int nparts = (PT >> 5) + 1;                 // the part holding column PT (c, d) is the last one with content

    // The weights of this wave's first task of a part are requested one phase early -- before the barrier
    // that ends the previous VALU phase, for part 0 before the window is staged -- so the L2 round trip is
    // not on the critical path of the MFMA phase.  (The projection bias is row D of the packed weights,
    // multiplied by a constant-one column of Vs: no separate bias loads.)
    // X3: when the node values are known to fit fp16 (the producing convolution recorded their maximum), two fp16 pieces
    // per operand and three MFMA terms instead of three bf16 pieces and six; the weights then carry the layer's power of two
    bool useh = false;
    if constexpr (X3) useh = a.vmax != nullptr && __uint_as_float(*a.vmax) < 32768.f;
    if (a.winflag) {                                   // behind a CONV launch of k_gath: only the windows it flagged (uniform per workgroup)
        if (!a.winflag[win]) return;
    } else if (X3 && a.skip_h && useh) return;         // k_gath has served this launch (uniform over the grid: no barrier is split)
    const int npw = X3 ? (useh ? 2 : 3) : 1;           // words per weight chunk and lane
    const f32x4* __restrict__ Wbase = (X3 && useh) ? a.Wp2 : a.Wp;
    f32x4 w[QB][NP];
    auto wfetch = [&](const f32x4* __restrict__ wp, int u, int q) {
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
            if (pc < npw) w[u][pc] = wp[((long)q * npw + pc) * 64];
    };
    auto prefetch = [&](int part) {
        if (wave < ntask) {
            const int wtile = wave >= NTn ? a.NT_L + part : part;
            const f32x4* __restrict__ wp = Wbase + ((long)wtile * Q) * (64 * npw) + lane;
#pragma unroll
            for (int u = 0; u < QB; ++u) wfetch(wp, u, u < Q ? u : Q - 1);
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
        lp[ii] = (lds_cptr)(Ls + (i0 + li + RI * ii) * GAT_LLD);
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
            const f32x4* __restrict__ wp = Wbase + ((long)wtile * Q) * (64 * npw) + lane;
            if (task != wave) {                    // more tiles than waves: later tasks pay their own round trip
#pragma unroll
                for (int u = 0; u < QB; ++u) wfetch(wp, u, u < Q ? u : Q - 1);
            }
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            for (int qb = 0; qb < Q; qb += QB) {
#pragma unroll
                for (int u = 0; u < QB; ++u)
                    if (qb + u < Q) {
                        if (BF) {
                            // the upper half of the last chunk may lie past the row's zero padding: its weights are
                            // zero, the read is clamped into the row so that it stays finite
                            const int c1 = 16 * (qb + u) + 8 + 4 * g;
                            const f32x4 lo = *reinterpret_cast<const f32x4*>(vrow + 16 * (qb + u) + 4 * g);
                            f32x4 hi = *reinterpret_cast<const f32x4*>(vrow + (c1 + 3 < vld ? c1 : vld - 4));
                            if constexpr (X3) {
                                // (a clamped upper half would carry real values here: its weights are zero, but the three
                                // pieces of a wrong value are still finite -- nothing to mask)
                                if (useh) {
                                    f32x4 xh, xl;
                                    split2h(lo, hi, xh, xl);
                                    o = mfma_h(w[u][0], xl, o);
                                    o = mfma_h(w[u][1], xh, o);
                                    o = mfma_h(w[u][0], xh, o);
                                } else {
                                    f32x4 xs[3];
                                    split3(lo, hi, xs[0], xs[1], xs[2]);
                                    o = mfma_s3(w[u], xs, o);
                                }
                            } else {
                                o = mfma_bf(w[u][0], cvt8(lo, hi), o);
                            }
                        } else {
                            const f32x4 xv = *reinterpret_cast<const f32x4*>(vrow + 8 * (qb + u) + 4 * g);
                            o = mfma4(w[u][0], xv, o);
                        }
                        // the chunk QB further on replaces this one as soon as it has been issued
                        if (qb + QB + u < Q) wfetch(wp, u, qb + QB + u);
                    }
            }
            if (X3 && useh) {                      // the fp16 weights carry S
                const float inv = a.scale2[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] *= inv;
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
            gat_load<IBL, JPL, RJ>(lA, rA, lq, rq, 0);
            int npos = ptile - 4 * part;
            npos = npos < 0 ? 0 : (npos > ntl ? ntl : npos);
            int kt = 0;
            // two back-to-back loops rather than a sign branch inside one: with the diamond the compiler
            // keeps two register copies of the accumulators (and spills)
#pragma unroll 1
            for (; kt < npos; ++kt) {
                gat_tile<IBL, JPL, RJ, false>(acc, lA, rA, lB, rB, lq, rq);
#pragma unroll
                for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                rq += 8;
            }
#pragma unroll 1
            for (; kt < ntl; ++kt) {
                gat_tile<IBL, JPL, RJ, true>(acc, lA, rA, lB, rB, lq, rq);
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
        for (int jj = 0; jj < JPL; ++jj) dv[jj] = rp[jj * RJ * GAT_LLD + col];
    }
    __syncthreads();
    if (!rows_owner) return;                           // no barrier below this point

    // ---- scores -> softmax over j (reference modules.py:85-89 / :184-188); a query row lives in
    // the 16 lanes of one DPP row (x JPL registers), so the reductions are row-local DPP butterflies
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        const int irow = i0 + li + RI * ii;
        const int irc = irow < K ? irow : K - 1;
        float e[JPL];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = lj + RJ * jj;
            const float b = a.bias ? a.bias[(long)irc * K + (j < K ? j : K - 1)] : 0.f;
            float v = acc[ii][jj] + cv[ii] + dv[jj];
            if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
            v += b;
            v = j < K ? v : -INFINITY;
            e[jj] = v;
            m = fmaxf(m, v);
        }
        m = row_max<RJ>(m);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            e[jj] = (lj + RJ * jj < K) ? soft_exp(e[jj] - m) : 0.f;
            sum += e[jj];
        }
        sum = row_sum<RJ>(sum);
        const float inv = soft_rcp(sum);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = irow < K ? e[jj] * inv : 0.f;
        if (a.ATT) {
            // training: keep the softmax row for the backward, then drop attention entries (reference
            // F.dropout(attention, p, training), modules.py:90 / :189) -- the aggregation below sees att * mask / (1 - p)
            float* __restrict__ ap = a.ATT + (win * K + irc) * (long)K;
            const unsigned key = drop_window_key(a.drop, a.drop_stream, win);
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = lj + RJ * jj;
                if (irow < K && j < K) ap[j] = acc[ii][jj];
                if (a.drop.thresh)
                    acc[ii][jj] = drop_keep(key, (unsigned)(irc * K + j), a.drop.thresh) ? acc[ii][jj] * a.drop.keep_scale : 0.f;
            }
        }
    }

    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j) on the matrix pipe, as out^T = V^T att^T with
    // v_mfma_f32_16x16x4_f32 (M = 16 output features, N = this wave's 16 query rows, 4 keys per
    // instruction): no padding to 32 rows, and each lane ends up with 4 consecutive features of one row.
    // att is restaged, 64 keys at a time, through this wave's slice of the (now free) Ls/Rs region.
    //   B operand: lane (n = lane&15, kb = lane>>4) = att[row n][key 4 kb + t]   (16-byte LDS read = 4 steps t)
    //   A operand: lane (m = lane&15, kb)           = V[key 4 kb + t][16 dt + m]
    //   D: register r of lane (n, mb = lane>>4)     = out[row n][16 dt + 4 mb + r]
    static_assert(IBW == 16, "one 16-row MFMA group per wave");
    constexpr int DTMAX = 8;                           // D <= 128 (plan)
    float* __restrict__ att = Ls + wave * (IBW * GAT_APITCH);
    const int DT = (D + 15) >> 4;
    const int nr = lane & 15, kb = lane >> 4;
    constexpr int JPP = 64 / RJ;                       // key registers per 64-key pass
    constexpr int PASSES = (JPL + JPP - 1) / JPP;
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
                for (int j4 = 0; j4 < JPP; ++j4)       // all 64 key columns of the pass: the MFMA below reads whole 16-key groups, and
                                                        // whatever the previous owner of this LDS left there (NaN patterns) times a zero row of V is not zero
                    att[(li + RI * ii) * GAT_APITCH + lj + RJ * j4] = (JPP * pass + j4 < JPL) ? acc[ii][(JPP * pass + j4 < JPL) ? JPP * pass + j4 : 0] : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int jn = min(64, K - pass * 64);
            const int ngrp = (jn + 15) >> 4;                   // rows < Kp16 of Vs: real or zero
            for (int grp = 0; grp < ngrp; ++grp) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(att + nr * GAT_APITCH + 16 * grp + 4 * kb);
                const float* __restrict__ vk = Vs + (pass * 64 + 16 * grp + 4 * kb) * vld;
                if (X3 && useh) {
                    // two fp16 pieces of the attention weights (in [0, 1 / (1 - p)]) and of the node values (below 2^15): the
                    // four keys of a lane are one v_mfma_f32_16x16x16_f16 operand, three terms per product -- on the pipe
                    // that runs beside the other waves' pair grids (the fp32 form below does not)
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    unsigned h0, l0, h1, l1;
                    split_pair_h(bq[0], bq[1], h0, l0);
                    split_pair_h(bq[2], bq[3], h1, l1);
                    const f16x4 bhh = __builtin_bit_cast(f16x4, u2{h0, h1}), bll = __builtin_bit_cast(f16x4, u2{l0, l1});
#pragma unroll
                    for (int dt = 0; dt < DTMAX; ++dt)
                        if (dt < DT) {
                            unsigned a0, c0, a1, c1;
                            split_pair_h(vk[dcol[dt]], vk[vld + dcol[dt]], a0, c0);
                            split_pair_h(vk[2 * vld + dcol[dt]], vk[3 * vld + dcol[dt]], a1, c1);
                            const f16x4 ahh = __builtin_bit_cast(f16x4, u2{a0, a1}), all_ = __builtin_bit_cast(f16x4, u2{c0, c1});
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahh, bll, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(all_, bhh, o[dt], 0, 0, 0);
                            o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahh, bhh, o[dt], 0, 0, 0);
                        }
                    continue;
                }
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


// ---- staging of one 32-column part of the wide kernels (round 6): the block's L' rows (row-major source) and the R' columns of KJ
// keys (key-minor source) as 16-byte loads -- a thread takes 8 consecutive columns of one L' row (2 loads, 4 ds_write_b64) and, per
// unit, 4 consecutive keys of a column PAIR (2 loads, 4 ds_write_b64: the 4 x 2 transpose is register naming) -- issued together
// (one memory round trip per part) and stored without masks when the tile is interior (uniform test).  Before: 4-byte loads, one
// element per load, ~17 vector instructions of index arithmetic and masking per element -- 414 per thread and part at config 4,
// a tenth of the pair grid's count.  LC rows / RT rows are whole 16-byte words (ldl a multiple of 32, Kp of 4, bases aligned).
template <int NTHR, int KJ>
struct WideStage {
    static constexpr int QN = KJ / 4;                  // key quads of the tile
    static constexpr int RU = 4 * KJ / NTHR;           // (16 column pairs x QN quads) / threads
    f32x4 l0, l1, ra[RU], rb[RU];
    __device__ __forceinline__ void issue(const float* __restrict__ LCw, int ldl, const float* __restrict__ RTw, int Kp, int i0b, int kb0,
                                          int c0, int K, int tid) {
        int tl = tid;
        asm volatile("" : "+v"(tl));                   // (indices re-derived per part: as loop invariants they were spilled)
        {
            const int r = tl >> 2, cq = (tl & 3) * 8;
            const int row = i0b + r < K ? i0b + r : K - 1;
            const float* __restrict__ p = LCw + (unsigned)(row * ldl + c0 + cq);
            l0 = *reinterpret_cast<const f32x4*>(p);
            l1 = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int n = 0; n < RU; ++n) {
            const int u = tl + n * NTHR;
            const int cp = u / QN, q = u - cp * QN;
            const int j0 = kb0 + 4 * q < Kp - 4 ? kb0 + 4 * q : Kp - 4;       // (a quad past the keys is masked when it is stored)
            const float* __restrict__ p = RTw + (unsigned)((c0 + 2 * cp) * Kp + j0);
            ra[n] = *reinterpret_cast<const f32x4*>(p);
            rb[n] = *reinterpret_cast<const f32x4*>(p + Kp);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ Ls, float* __restrict__ Rs, int i0b, int kb0, int c0, int K, int PT, int tid) const {
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const bool interior = i0b + NTHR / 4 <= K && kb0 + KJ <= K && c0 + 32 <= PT;      // (uniform)
        {
            const int r = tl >> 2, cq = (tl & 3) * 8;
            f32x2* __restrict__ d = reinterpret_cast<f32x2*>(Ls + r * GAT_LLD + cq);
            if (interior) {
                d[0] = f32x2{l0[0], l0[1]}; d[1] = f32x2{l0[2], l0[3]}; d[2] = f32x2{l1[0], l1[1]}; d[3] = f32x2{l1[2], l1[3]};
            } else {
                const bool rok = i0b + r < K;
                float e[8] = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = (rok && c0 + cq + k < PT) ? e[k] : 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = f32x2{e[2 * k], e[2 * k + 1]};
            }
        }
#pragma unroll
        for (int n = 0; n < RU; ++n) {
            const int u = tl + n * NTHR;
            const int cp = u / QN, q = u - cp * QN;
            float* __restrict__ d = Rs + (4 * q) * GAT_LLD + 2 * cp;
            if (interior) {
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x2*>(d + k * GAT_LLD) = f32x2{ra[n][k], rb[n][k]};
            } else {
                const bool cok = c0 + 2 * cp < PT;     // (PT is even: both columns of the pair are on one side of it)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = cok && kb0 + 4 * q + k < K;
                    *reinterpret_cast<f32x2*>(d + k * GAT_LLD) = f32x2{ok ? ra[n][k] : 0.f, ok ? rb[n][k] : 0.f};
                }
            }
        }
    }
};

// ---- aggregation o += att V of the wide kernels, software-pipelined (round 6).  The V rows of the window go through LDS in
// tiles of VK keys (32; 16 when the node dimension exceeds 256 so that two tiles fit beside the softmax slices), TWO tile
// buffers filled by LDS-DMA (global_load_lds_dwordx4: a 1-KiB piece of a V row per wave instruction, no staging registers):
// while the workgroup's waves run the MFMAs of tile t, tile t + 1 is on its way from memory into the other buffer; ONE
// s_waitcnt + barrier per tile publishes it.  Before: every tile was loaded element by element behind guards (a branch and a
// memory round trip per 16 bytes and thread, 4..9 in a row) between two barriers, with the matrix pipe idle.
// `fast` (uniform): V rows 16-byte aligned with a pitch and a node dimension that are multiples of four; otherwise the tile is
// staged by guarded scalar loads in front of its barrier as before (no prefetch).
template <int DTMAX>
struct WideAgg {
    static constexpr int VK = DTMAX > 16 ? 16 : 32;                     // keys per V tile
};

// 16 bytes per lane from (uniform base + per-lane byte offset) to LDS byte address ldsdst + 16 * lane (M0 is compiler-reserved:
// written and restored inside the statement; as mtadgat_gru_cm.hip's glds_burst)
__device__ __forceinline__ void gat_glds16(const void* sbase, unsigned voff, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
}

// request tile kv .. kv + VK - 1 into Vt ([VK][vld2]); rows past K are zero-filled with ordinary stores.  Not waited for.
template <int DTMAX, int NTHR>
__device__ __forceinline__ void wide_vtile_dma(float* __restrict__ Vt, const float* __restrict__ Vw, int ldv, int kv, int K, int D, int vld2,
                                               int wave, int lane, bool fast) {
    if (!fast) return;
    constexpr int VK = WideAgg<DTMAX>::VK, NWV = NTHR / 64;
    const int npc = (D + 255) >> 8;                                     // 1-KiB pieces per row
    const int col = lane * 4;
    // the window's V base as an SGPR pair (it is uniform, but a 64-bit product computed on the vector ALU does not satisfy the
    // "s" constraint by itself); the row and piece go into the 32-bit lane offset
    const unsigned long vb = reinterpret_cast<unsigned long>(Vw);
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)vb), bhi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
    const void* sbase = reinterpret_cast<const void*>(((unsigned long)bhi << 32) | blo);
    for (int r = wave; r < VK; r += NWV)                                // (wave-uniform)
        for (int pc = 0; pc < npc; ++pc) {
            float* __restrict__ dst = Vt + r * vld2 + pc * 256;
            if (pc * 256 + col < D) {
                if (kv + r < K) gat_glds16(sbase, (unsigned)(((kv + r) * ldv + pc * 256 + col) * 4), (unsigned)(size_t)(lds_cptr)dst);
                else *reinterpret_cast<f32x4*>(dst + col) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
}
// the slow path of a tile (not `fast`): guarded scalar loads, stored at once
template <int DTMAX, int NTHR>
__device__ __forceinline__ void wide_vtile_slow(float* __restrict__ Vt, const float* __restrict__ Vw, int ldv, int kv, int K, int D, int vld2, int tid) {
    constexpr int VK = WideAgg<DTMAX>::VK;
    const int vq = vld2 >> 2;
    for (int u = tid; u < VK * vq; u += NTHR) {
        const int r = u / vq, c4 = (u - r * vq) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kv + r < K) {
            const float* src = Vw + (long)(kv + r) * ldv + c4;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) v[s4] = (c4 + s4 < D) ? src[s4] : 0.f;
        }
        *reinterpret_cast<f32x4*>(Vt + r * vld2 + c4) = v;
    }
}

// o += att V over the keys [kbeg, kbeg + 128 NKP) of the window (those below K).  acc: the (dropped-out) softmax rows of this
// wave's 16 query rows, pair-grid layout; att: this wave's private LDS slice [16][GAT_APITCH]; Vs2: two tile buffers of
// VK x vld2 floats shared by the workgroup, the first tile (keys kbeg ..) already requested into buffer 0 by wide_vtile_dma.
// On entry every wave of the workgroup is past its last use of the LDS that att aliases (the caller's barrier).
template <int NKP, int DTMAX, int NTHR>
__device__ __forceinline__ void wide_aggregate(const float (&acc)[NKP][4][8], f32x4 (&o)[DTMAX], float* __restrict__ att,
                                               float* __restrict__ Vs2, const float* __restrict__ Vw, int ldv, int kbeg, int K, int D,
                                               int vld2, int tid, int wave, bool fast) {
    constexpr int RJ = 16, RI = 4, VK = WideAgg<DTMAX>::VK, TPH = 64 / VK;          // V tiles per 64-key half
    const int lane = tid & 63, lj = lane % RJ, li = lane / RJ;
    const int nr = lane & 15, kb = lane >> 4;
    const int DT = (D + 15) >> 4;
    const int kend = kbeg + 128 * NKP < K ? kbeg + 128 * NKP : K;
    int buf = 0;
    static_for<0, 2 * NKP>([&](auto khc) {
        constexpr int kp = decltype(khc)::value >> 1, half = decltype(khc)::value & 1;
        const int k0 = kbeg + kp * 128 + half * 64;                 // 64 keys of att at a time
        if (k0 < kend) {
            // this wave's slice: its own earlier reads are behind it in program order (LDS operations of a wave complete in order)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) att[(li + RI * ii) * GAT_APITCH + lj + RJ * j4] = acc[kp][ii][4 * half + j4];
#pragma unroll 1
            for (int kq = 0; kq < TPH; ++kq) {
                const int kv = k0 + VK * kq;
                if (kv < kend) {                                    // (uniform)
                    float* __restrict__ Vt = Vs2 + buf * (VK * vld2);
                    if (!fast) wide_vtile_slow<DTMAX, NTHR>(Vt, Vw, ldv, kv, K, D, vld2, tid);
                    // tile t has landed for every wave; every wave is done with the MFMAs of tile t - 1, i.e. with the other buffer
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (kv + VK < kend) wide_vtile_dma<DTMAX, NTHR>(Vs2 + (buf ^ 1) * (VK * vld2), Vw, ldv, kv + VK, K, D, vld2, wave, lane, fast);
#pragma unroll
                    for (int grp = 0; grp < VK / 16; ++grp) {       // 16 keys per MFMA group
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(att + nr * GAT_APITCH + VK * kq + 16 * grp + 4 * kb);
                        const float* __restrict__ vk = Vt + (16 * grp + 4 * kb) * vld2 + nr;
                        // Output tiles in pairs behind ONE uniform branch each (no second code path: two versions of the loop
                        // made the compiler shuffle all of o between two register assignments); the four V words of tile
                        // dt + 1 are requested before the MFMAs of tile dt (left alone the compiler waits for each word right
                        // in front of its MFMA: an LDS round trip per 32-cycle instruction).  An odd DT computes one tile of
                        // finite garbage (the next row's first words) into an o that is never stored.
                        const float* __restrict__ vk1 = vk + vld2;
                        const float* __restrict__ vk2 = vk1 + vld2;
                        const float* __restrict__ vk3 = vk2 + vld2;
                        float avA[4], avB[4];
                        avA[0] = vk[0]; avA[1] = vk1[0]; avA[2] = vk2[0]; avA[3] = vk3[0];
                        static_for<0, DTMAX / 2>([&](auto dc) {
                            constexpr int d0 = 2 * decltype(dc)::value;
                            if (d0 < DT) {
                                avB[0] = vk[16 * (d0 + 1)]; avB[1] = vk1[16 * (d0 + 1)]; avB[2] = vk2[16 * (d0 + 1)]; avB[3] = vk3[16 * (d0 + 1)];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int t = 0; t < 4; ++t) o[d0] = __builtin_amdgcn_mfma_f32_16x16x4f32(avA[t], bq[t], o[d0], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (d0 + 2 < DTMAX) {
                                    avA[0] = vk[16 * (d0 + 2)]; avA[1] = vk1[16 * (d0 + 2)]; avA[2] = vk2[16 * (d0 + 2)]; avA[3] = vk3[16 * (d0 + 2)];
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int t = 0; t < 4; ++t) o[d0 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(avB[t], bq[t], o[d0 + 1], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        });
                    }
                    buf ^= 1;
                }
            }
        }
    });
}

// ---------------------------------------------------------------------------
// gat (wide): graph attention for node counts beyond the per-window fused kernel (128 < K <= 512, BASELINE
// config 4: 512 features / 256 time steps).  The projected L', R' come from k_rowgemm through HBM (LC row-major
// per query node, RT key-node-minor -- what k_attend consumed); everything after that is the fused kernel's
// machinery: a workgroup owns a block of up to 128 query rows of one window, stages the L' rows of its block and
// the R' columns of ALL keys of one 32-column part in LDS, runs the same 2-D register-blocked pair grid (4 x 8
// pairs per lane and 128-key pass, KP passes -> all K scores of a row stay in registers), then softmax and the
// aggregation att V on the 16x16x4 MFMA with V staged through LDS 32 keys at a time for all waves.
//   LDS: Ls [128][34] | Rs [KP*128][34]  (aliased afterwards by: att [NW][16][68] | Vs2 [32][D + 4])
// ---------------------------------------------------------------------------
struct GatWideArgs {
    const float* LC;     // (B*K, ldl): [L'(PT) | c | pad]
    const float* RT;     // (B, rt_rows, Kp): rows [0, PT) = R' (key minor), row PT = d
    int ldl, rt_rows, Kp, PT, P8;
    const int* ord;      // device-side [P8, PT] (overrides PT / P8 when set)
    const float* bias;   // (K, K) or null
    const float* V;      // (B*K, ldv) node rows
    int ldv, D, K;
    float* out;          // out[win*so_w + i*so_i + d*so_d]
    long so_w, so_i, so_d;
    long nwin;
    int nblk;            // row blocks per window
    int v1;
    float alpha;
    // training mode (as GatArgs): the softmax rows (before dropout) are kept for the backward, dropout (modules.py:90 / :189) is
    // applied to the attention matrix inside the kernel
    float* ATT;          // (B, K, K) or null
    DropArgs drop;
    unsigned drop_stream;
};

// KP = 4 (385..512 keys) keeps 128 score registers per lane: those workgroups have 4 waves (one per SIMD, 512 VGPRs)
template <int KP, int DTMAX>
__global__ __launch_bounds__((KP == 4 ? 256 : 512), 1) void k_gat_wide(const GatWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IBL = 4, JPL = 8, RJ = 16, RI = 4, IBW = 16;
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = nthr >> 6;
    // XCD-aware map: the row blocks of one window run on the same XCD (dispatch ids b, b + 8, ...), so its R' / V
    // are fetched from HBM once and served to the other row blocks from that XCD's L2
    const long blk = blockIdx.x;
    const long grp = blk / (8 * a.nblk);
    const int within = (int)(blk - grp * (8 * a.nblk));
    const long win = grp * 8 + (within & 7);
    const int rb = within >> 3;
    if (win >= a.nwin) return;
    const int K = a.K, D = a.D;
    int P8, PT;
    gat_load_order(a.ord, a.P8, a.PT, P8, PT);
    const int i0b = rb * (NW * IBW);                       // first query row of this workgroup
    const int KJ = KP * 128;                               // key slots
    float* __restrict__ Ls = smem;                         // [NW*16][34]
    float* __restrict__ Rs = Ls + NW * IBW * GAT_LLD;      // [KJ][34]
    const int lj = lane % RJ, li = lane / RJ;
    const float* __restrict__ LCw = a.LC + (win * K) * (long)a.ldl;
    const float* __restrict__ RTw = a.RT + win * (long)a.rt_rows * a.Kp;

    const int i0 = i0b + wave * IBW;
    lds_cptr lp[IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        lp[ii] = (lds_cptr)(Ls + (wave * IBW + li + RI * ii) * GAT_LLD);
        asm volatile("" : "+v"(lp[ii]));
    }
    const lds_cptr rp0 = (lds_cptr)(Rs + lj * GAT_LLD);
    float acc[KP][IBL][JPL];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) acc[kp][ii][jj] = 0.f;

    const int ntile = PT >> 3, ptile = P8 >> 3;
    const int nparts = (ntile + 3) >> 2;
    for (int part = 0; part < nparts; ++part) {
        // ---- stage this part (WideStage): all loads issued together, one memory round trip
        {
            WideStage<(KP == 4 ? 256 : 512), KP * 128> st;
            st.issue(LCw, a.ldl, RTw, a.Kp, i0b, 0, 32 * part, K, tid);
            st.store(Ls, Rs, i0b, 0, 32 * part, K, PT, tid);
        }
        __syncthreads();
        int ntl = ntile - 4 * part;
        ntl = ntl > 4 ? 4 : ntl;
        int npos = ptile - 4 * part;
        npos = npos < 0 ? 0 : (npos > ntl ? ntl : npos);
        static_for<0, KP>([&](auto kpc) {
            constexpr int kp = decltype(kpc)::value;
            if (kp * 128 < K) {
                f32x2 lA[IBL], rA[JPL], lB[IBL], rB[JPL];
                lds_cptr lq[IBL];
#pragma unroll
                for (int ii = 0; ii < IBL; ++ii) lq[ii] = lp[ii];
                lds_cptr rq = rp0 + kp * 128 * GAT_LLD;
                gat_load<IBL, JPL, RJ>(lA, rA, lq, rq, 0);
                int kt = 0;
#pragma unroll 1
                for (; kt < npos; ++kt) {
                    gat_tile<IBL, JPL, RJ, false>(acc[kp], lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                    rq += 8;
                }
#pragma unroll 1
                for (; kt < ntl; ++kt) {
                    gat_tile<IBL, JPL, RJ, true>(acc[kp], lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                    rq += 8;
                }
            }
        });
        __syncthreads();
    }
    // ---- the first V tile of the aggregation is requested here (the part loop ended with a barrier: every wave is done with
    // Ls / Rs, which the tile buffers alias): its round trip runs under the softmax
    constexpr int NTHRC = KP == 4 ? 256 : 512;
    const int vld2 = ((D + 15) & ~15) + 4;
    const float* __restrict__ Vw = a.V + (win * K) * (long)a.ldv;
    const bool vfast = (a.ldv & 3) == 0 && (D & 3) == 0 && (reinterpret_cast<size_t>(a.V) & 15) == 0;
    float* __restrict__ Vs2 = smem + NW * IBW * GAT_APITCH;      // 2 x [VK][vld2]
    wide_vtile_dma<DTMAX, NTHRC>(Vs2, Vw, a.ldv, 0, K, D, vld2, wave, lane, vfast);
    // ---- scores -> softmax over all K keys of a row (16 lanes x KP*8 registers)
    // The score's rank-1 terms and the bias are loaded in BATCHES, unconditionally from clamped addresses: the keys' d_j once
    // (they do not depend on the row), the rows' c_i together, a row's KP*8 bias values together.  (Left as `acc + c + RT[..]`
    // and `if (bias) v += bias[..]` per element, each score waited for two memory round trips of its own -- a guarded load is a
    // branch with s_waitcnt vmcnt(0) at the join: 2 x KP*32 serial round trips per lane.)
    float dv[KP][JPL], cvr[IBL];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = kp * 128 + lj + RJ * jj;
            dv[kp][jj] = RTw[(unsigned)(PT * a.Kp + (j < K ? j : K - 1))];
        }
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        const int irow = i0 + li + RI * ii;
        cvr[ii] = LCw[(unsigned)((irow < K ? irow : K - 1) * a.ldl + PT)];
    }
    const bool has_bias = a.bias != nullptr;
    const float* __restrict__ bsrc = has_bias ? a.bias : LCw;          // (no bias: every lane reads one valid word and drops it)
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        const int irow = i0 + li + RI * ii;
        const int irc = irow < K ? irow : K - 1;
        const float cv = cvr[ii];
        float bv[KP][JPL];
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = kp * 128 + lj + RJ * jj;
                bv[kp][jj] = bsrc[has_bias ? (unsigned)(irc * K + (j < K ? j : K - 1)) : 0u];
            }
        float m = -INFINITY;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = kp * 128 + lj + RJ * jj;
                float v = acc[kp][ii][jj] + cv + dv[kp][jj];
                if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
                v += has_bias ? bv[kp][jj] : 0.f;
                v = j < K ? v : -INFINITY;
                acc[kp][ii][jj] = v;
                m = fmaxf(m, v);
            }
        m = row_max<RJ>(m);
        float sum = 0.f;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = kp * 128 + lj + RJ * jj;
                const float e = j < K ? soft_exp(acc[kp][ii][jj] - m) : 0.f;
                acc[kp][ii][jj] = e;
                sum += e;
            }
        sum = row_sum<RJ>(sum);
        const float inv = soft_rcp(sum);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) acc[kp][ii][jj] = irow < K ? acc[kp][ii][jj] * inv : 0.f;
    }
    if (a.ATT || a.drop.thresh) {                          // training: keep the softmax rows, drop attention entries (counter-based mask)
        const unsigned key = drop_window_key(a.drop, a.drop_stream, win);
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + li + RI * ii;
#pragma unroll
            for (int kp = 0; kp < KP; ++kp)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = kp * 128 + lj + RJ * jj;
                    if (irow < K && j < K) {
                        if (a.ATT) a.ATT[(win * K + irow) * (long)K + j] = acc[kp][ii][jj];
                        if (a.drop.thresh) acc[kp][ii][jj] *= drop_keep(key, (unsigned)(irow * K + j), a.drop.thresh) ? a.drop.keep_scale : 0.f;
                    }
                }
        }
    }
    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j): out^T = V^T att^T on v_mfma_f32_16x16x4_f32 (as k_gat); the
    // softmax rows go through this wave's LDS slice 64 keys at a time, V through two shared LDS tiles (wide_aggregate)
    float* __restrict__ att = smem + wave * (IBW * GAT_APITCH);
    const int nr = lane & 15, kb = lane >> 4;
    const int DT = (D + 15) >> 4;
    f32x4 o[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    wide_aggregate<KP, DTMAX, NTHRC>(acc, o, att, Vs2, Vw, a.ldv, 0, K, D, vld2, tid, wave, vfast);
    {
        const int row = i0 + nr;
        float* __restrict__ orow = a.out + win * a.so_w + (long)row * a.so_i;
#pragma unroll
        for (int dt = 0; dt < DTMAX; ++dt)
            if (dt < DT) {
                const int d0 = 16 * dt + 4 * kb;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row < K && d0 + r < D) orow[(long)(d0 + r) * a.so_d] = gate_sigmoid(o[dt][r]);
            }
    }
}

// ---------------------------------------------------------------------------
// gat (wide), more than 256 keys, inference: the key axis in blocks of 256 with a running softmax (round 5).
// k_gat_wide keeps ALL scores of a query row in registers -- 128 per lane at 385..512 keys, i.e. a 512-register wave, one wave
// per SIMD: the pair grid then issues one VALU instruction per ~4.7 cycles and nothing covers the staging round trips (config 4's
// feature layer, 512 nodes: two thirds of that shape's pair-grid work at 0.2 of the vector ALU).  Here a workgroup walks the keys in
// blocks of 256: pair grid of the block (all embedding parts; 64 score registers per lane), then
//     m' = max(m, max_j s_j),  l = l e^(m - m') + sum_j e^(s_j - m'),  o = o e^(m - m') + sum_j e^(s_j - m') V_j
// (the aggregation of the block on the 16x16x4 MFMA as in k_gat_wide), and at the end h = sigmoid(o / l): eight waves per workgroup,
// two per SIMD.  Same inputs, same output; the softmax rows never exist as a whole, so the training forward (which keeps them)
// stays on k_gat_wide.
// ---------------------------------------------------------------------------
// NW waves per workgroup (8: one workgroup per CU, 128 query rows).
// (Measured and not kept, round 6: key blocks of 128 -- 32 score registers per lane, no spills -- with the L' / R' words of part
// p + 1 requested before part p's pair grid and stored behind it: 6.75-6.86 against 6.50-6.53 ms per 896-window chunk; the L' tile
// is staged twice as often and the staging round trip is not what the kernel waits for.)
template <int DTMAX, int NW>
__global__ __launch_bounds__(64 * NW, (NW == 4 ? 2 : 1)) void k_gat_wide_os(const GatWideArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int IBL = 4, JPL = 8, RJ = 16, RI = 4, IBW = 16, KPB = 2, KB = KPB * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int nthr = 64 * NW;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long blk = blockIdx.x;
    const long grp = blk / (8 * a.nblk);
    const int within = (int)(blk - grp * (8 * a.nblk));
    const long win = grp * 8 + (within & 7);
    const int rb = within >> 3;
    if (win >= a.nwin) return;
    const int K = a.K, D = a.D;
    int P8, PT;
    gat_load_order(a.ord, a.P8, a.PT, P8, PT);
    const int i0b = rb * (NW * IBW);
    float* __restrict__ Ls = smem;                         // [128][34]
    float* __restrict__ Rs = Ls + NW * IBW * GAT_LLD;      // [256][34]
    const int lj = lane % RJ, li = lane / RJ;
    const float* __restrict__ LCw = a.LC + (win * K) * (long)a.ldl;
    const float* __restrict__ RTw = a.RT + win * (long)a.rt_rows * a.Kp;
    const int i0 = i0b + wave * IBW;
    lds_cptr lp[IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        lp[ii] = (lds_cptr)(Ls + (wave * IBW + li + RI * ii) * GAT_LLD);
        asm volatile("" : "+v"(lp[ii]));
    }
    const lds_cptr rp0 = (lds_cptr)(Rs + lj * GAT_LLD);
    const int ntile = PT >> 3, ptile = P8 >> 3;
    const int nparts = (ntile + 3) >> 2;
    // aggregation-side layout (as k_gat_wide): the wave's softmax rows through its LDS slice, V through a shared 32-key tile
    const int vld2 = ((D + 15) & ~15) + 4;
    float* __restrict__ att = smem + wave * (IBW * GAT_APITCH);
    // the two V tile buffers do NOT alias the pair-grid region: a key block's first tile is requested (LDS-DMA) before the
    // block's softmax, while other waves may still be in the pair grid
    const int att_floats = NW * IBW * GAT_APITCH, pair_floats = (NW * IBW + KB) * GAT_LLD;
    float* __restrict__ Vs2 = smem + (att_floats > pair_floats ? att_floats : pair_floats);
    float* __restrict__ scl = Vs2 + 2 * 32 * vld2 + wave * IBW;     // [16] per wave: row factors
    const int nr = lane & 15, kbq = lane >> 4;
    const int DT = (D + 15) >> 4;
    const float* __restrict__ Vw = a.V + (win * K) * (long)a.ldv;
    f32x4 o[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool vfast = (a.ldv & 3) == 0 && (D & 3) == 0 && (reinterpret_cast<size_t>(a.V) & 15) == 0;
    float mrow[IBL], lrow[IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) { mrow[ii] = -INFINITY; lrow[ii] = 0.f; }

    for (int kb0 = 0; kb0 < K; kb0 += KB) {
        float acc[KPB][IBL][JPL];
#pragma unroll
        for (int kp = 0; kp < KPB; ++kp)
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) acc[kp][ii][jj] = 0.f;
        for (int part = 0; part < nparts; ++part) {
            __syncthreads();                               // the previous users of Ls / Rs (pair grid, att slices) are done
            {
                WideStage<nthr, KB> st;                    // all loads of the part issued together: one memory round trip
                st.issue(LCw, a.ldl, RTw, a.Kp, i0b, kb0, 32 * part, K, tid);
                st.store(Ls, Rs, i0b, kb0, 32 * part, K, PT, tid);
            }
            __syncthreads();
            int ntl = ntile - 4 * part;
            ntl = ntl > 4 ? 4 : ntl;
            int npos = ptile - 4 * part;
            npos = npos < 0 ? 0 : (npos > ntl ? ntl : npos);
            static_for<0, KPB>([&](auto kpc) {
                constexpr int kp = decltype(kpc)::value;
                if (kb0 + kp * 128 < K) {
                    f32x2 lA[IBL], rA[JPL], lB[IBL], rB[JPL];
                    lds_cptr lq[IBL];
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] = lp[ii];
                    lds_cptr rq = rp0 + kp * 128 * GAT_LLD;
                    gat_load<IBL, JPL, RJ>(lA, rA, lq, rq, 0);
                    int kt = 0;
#pragma unroll 1
                    for (; kt < npos; ++kt) {
                        gat_tile<IBL, JPL, RJ, false>(acc[kp], lA, rA, lB, rB, lq, rq);
#pragma unroll
                        for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                        rq += 8;
                    }
#pragma unroll 1
                    for (; kt < ntl; ++kt) {
                        gat_tile<IBL, JPL, RJ, true>(acc[kp], lA, rA, lB, rB, lq, rq);
#pragma unroll
                        for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                        rq += 8;
                    }
                }
            });
        }
        // ---- the block's first V tile is requested here: its round trip runs under the softmax of the block (buffer 0 was last
        // read by the previous block's aggregation, which every wave left before the barriers of this block's part loop)
        wide_vtile_dma<DTMAX, nthr>(Vs2, Vw, a.ldv, kb0, K, D, vld2, wave, lane, vfast);
        // ---- scores of the block, running softmax statistics; acc becomes e^(s - m'), scl the factor of what was summed before
        // (rank-1 terms and bias in batches of unconditional loads, as in k_gat_wide: one round trip each instead of two per score)
        float dv[KPB][JPL], cvr4[IBL];
#pragma unroll
        for (int kp = 0; kp < KPB; ++kp)
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = kb0 + kp * 128 + lj + RJ * jj;
                dv[kp][jj] = RTw[(unsigned)(PT * a.Kp + (j < K ? j : K - 1))];
            }
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + li + RI * ii;
            cvr4[ii] = LCw[(unsigned)((irow < K ? irow : K - 1) * a.ldl + PT)];
        }
        const bool has_bias = a.bias != nullptr;
        const float* __restrict__ bsrc = has_bias ? a.bias : LCw;
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + li + RI * ii;
            const int irc = irow < K ? irow : K - 1;
            const float cvr = cvr4[ii];
            float bv[KPB][JPL];
#pragma unroll
            for (int kp = 0; kp < KPB; ++kp)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = kb0 + kp * 128 + lj + RJ * jj;
                    bv[kp][jj] = bsrc[has_bias ? (unsigned)(irc * K + (j < K ? j : K - 1)) : 0u];
                }
            float mb = -INFINITY;
#pragma unroll
            for (int kp = 0; kp < KPB; ++kp)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = kb0 + kp * 128 + lj + RJ * jj;
                    float v = acc[kp][ii][jj] + cvr + dv[kp][jj];
                    if (a.v1) v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f);
                    v += has_bias ? bv[kp][jj] : 0.f;
                    v = j < K ? v : -INFINITY;
                    acc[kp][ii][jj] = v;
                    mb = fmaxf(mb, v);
                }
            mb = row_max<RJ>(mb);
            const float mnew = fmaxf(mrow[ii], mb);           // (finite: the block holds at least one real key)
            const float f = mrow[ii] == -INFINITY ? 0.f : soft_exp(mrow[ii] - mnew);
            float sum = 0.f;
#pragma unroll
            for (int kp = 0; kp < KPB; ++kp)
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = kb0 + kp * 128 + lj + RJ * jj;
                    const float e = (j < K && irow < K) ? soft_exp(acc[kp][ii][jj] - mnew) : 0.f;
                    acc[kp][ii][jj] = e;
                    sum += e;
                }
            sum = row_sum<RJ>(sum);
            lrow[ii] = lrow[ii] * f + sum;
            mrow[ii] = mnew;
            if (lj == 0) scl[li + RI * ii] = f;
        }
        __syncthreads();                                   // every wave is done with Ls / Rs; scl is visible
        {
            const float f = scl[nr];
#pragma unroll
            for (int dt = 0; dt < DTMAX; ++dt)
                if (dt < DT) { o[dt][0] *= f; o[dt][1] *= f; o[dt][2] *= f; o[dt][3] *= f; }
        }
        // ---- o += e V over the block's keys (wide_aggregate: two V tile buffers, one barrier per tile)
        wide_aggregate<KPB, DTMAX, nthr>(acc, o, att, Vs2, Vw, a.ldv, kb0, K, D, vld2, tid, wave, vfast);
    }
    // ---- h = sigmoid(o / l)
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
        if (lj == 0) scl[li + RI * ii] = soft_rcp(lrow[ii]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    {
        const float inv = scl[nr];
        const int row = i0 + nr;
        float* __restrict__ orow = a.out + win * a.so_w + (long)row * a.so_i;
#pragma unroll
        for (int dt = 0; dt < DTMAX; ++dt)
            if (dt < DT) {
                const int d0 = 16 * dt + 4 * kbq;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row < K && d0 + r < D) orow[(long)(d0 + r) * a.so_d] = gate_sigmoid(o[dt][r] * inv);
            }
    }
}

size_t gat_wide_os_lds(int D, int nw) {
    const size_t pair = (size_t)(nw * 16 + 256) * GAT_LLD;
    const size_t att = (size_t)nw * 16 * GAT_APITCH;
    // [pair grid | softmax slices] then two V tiles (WideAgg; D <= 256 here) that alias neither, then the row factors
    return ((pair > att ? pair : att) + (size_t)2 * 32 * (((D + 15) & ~15) + 4) + nw * 16) * sizeof(float);
}

size_t gat_wide_lds(int K, int D, int nw) {
    const int KP = (K + 127) / 128;
    const size_t pair = (size_t)(nw * 16 + KP * 128) * GAT_LLD;
    const size_t agg = (size_t)nw * 16 * GAT_APITCH + (size_t)2 * (D <= 256 ? 32 : 16) * (((D + 15) & ~15) + 4);     // two V tiles (WideAgg)
    return (pair > agg ? pair : agg) * sizeof(float);
}

int launch_gat_wide(const float* LC, const float* RT, int ldl, int rt_rows, int Kp, int PT, int P8, const float* bias,
                    const float* V, int ldv, int D, int K, float* out, long so_w, long so_i, long so_d, long nwin, int v1,
                    float alpha, hipStream_t s, float* att, const DropArgs* drop, unsigned drop_stream, const int* ord) {
    if (nwin <= 0) return 0;
    if (K > 512 || D > 512) return -2;
    const int KP = (K + 127) / 128;
    const int nw = KP == 4 ? 4 : 8;          // (eight waves also below 128 keys: the staging helpers take the thread count at compile time; waves past K idle)
    GatWideArgs a{};
    a.LC = LC; a.RT = RT; a.ldl = ldl; a.rt_rows = rt_rows; a.Kp = Kp; a.PT = PT; a.P8 = P8; a.bias = bias;
    a.V = V; a.ldv = ldv; a.D = D; a.K = K; a.out = out; a.so_w = so_w; a.so_i = so_i; a.so_d = so_d; a.nwin = nwin;
    a.nblk = (K + nw * 16 - 1) / (nw * 16);
    a.v1 = v1; a.alpha = alpha;
    a.ATT = att; a.drop_stream = drop_stream; a.ord = ord;
    if (drop) a.drop = *drop;
    const int dtmax = D <= 256 ? 16 : 32;
    static const bool os_off = getenv("MTADGAT_WIDE_OS") && atoi(getenv("MTADGAT_WIDE_OS")) == 0;     // (measurement hook)
    if (K > 256 && dtmax == 16 && !att && !(drop && drop->thresh) && !os_off) {
        // more than 256 keys, node dimension up to 256, inference: key blocks of 256 with a running softmax (eight waves per
        // workgroup, two per SIMD; with 32 output tiles per lane -- D up to 512 -- the kernel spills and k_gat_wide stays)
        // (NW = 4 -- 64 query rows per workgroup, two workgroups per CU, so that one's staging runs under the other's pair grid --
        // was measured in round 6: config 4's attention 131 instead of 120 ms per 8 192 windows; each workgroup stages all of R')
        constexpr int nw2 = 8;
        a.nblk = (K + nw2 * 16 - 1) / (nw2 * 16);
        const size_t lds2 = gat_wide_os_lds(D, nw2);
        if (lds2 <= 160 * 1024) {
            const unsigned grid2 = (unsigned)(((nwin + 7) / 8 * 8) * a.nblk);
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_wide_os<16, nw2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            if (e_ != hipSuccess) return (int)e_;
            hipLaunchKernelGGL((k_gat_wide_os<16, nw2>), dim3(grid2), dim3(64 * nw2), lds2, s, a);
            LAUNCH_CHECK();
            return 0;
        }
        a.nblk = (K + nw * 16 - 1) / (nw * 16);
    }
    const size_t lds = gat_wide_lds(K, D, nw);
    if (lds > 160 * 1024) return -2;
    const unsigned grid = (unsigned)(((nwin + 7) / 8 * 8) * a.nblk);
#define WIDE_CASE(N, DTM)                                                                                              \
    if (KP == N && dtmax == DTM) {                                                                                     \
        if (lds > 64 * 1024) {                                                                                         \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gat_wide<N, DTM>),                    \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
            if (e_ != hipSuccess) return (int)e_;                                                                      \
        }                                                                                                              \
        hipLaunchKernelGGL((k_gat_wide<N, DTM>), dim3(grid), dim3(64 * nw), lds, s, a);                                \
    }
    WIDE_CASE(1, 16) WIDE_CASE(2, 16) WIDE_CASE(3, 16) WIDE_CASE(4, 16) WIDE_CASE(1, 32) WIDE_CASE(2, 32) WIDE_CASE(3, 32) WIDE_CASE(4, 32)
#undef WIDE_CASE
    LAUNCH_CHECK();
    return 0;
}

#define GAT_CASE(I, J, RJ)                                                                      \
    if (IBL == I && JPL == J && rj == RJ) {                                                     \
        const void* fn_ = a.bf16 == 2 ? reinterpret_cast<const void*>(&k_gat<I, J, RJ, true, true>)  \
                          : a.bf16 ? reinterpret_cast<const void*>(&k_gat<I, J, RJ, true>)      \
                                   : reinterpret_cast<const void*>(&k_gat<I, J, RJ, false>);    \
        if (lds_bytes > 64 * 1024) {                                                            \
            hipError_t e_ = hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e_ != hipSuccess) return (int)e_;                                               \
        }                                                                                       \
        if (a.bf16 == 2) hipLaunchKernelGGL((k_gat<I, J, RJ, true, true>), dim3(grid), dim3(64 * nw), lds_bytes, s, a); \
        else if (a.bf16) hipLaunchKernelGGL((k_gat<I, J, RJ, true>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);  \
        else hipLaunchKernelGGL((k_gat<I, J, RJ, false>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);        \
        launched = true;                                                                        \
    }

// rj lanes along the key axis (16 or 8), IBL query rows per lane (a wave owns (64/rj)*IBL = 16 rows), JPL key
// nodes per lane (rj*JPL >= K), nw waves
int launch_gat(const GatArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, hipStream_t s) {
    if (a.nwin <= 0) return 0;
    if (rj * JPL < a.K || nw * 16 < a.K || nw > 8) return -2;     // nw may exceed the row-owning waves: the rest only project
    const unsigned grid = (unsigned)a.nwin;
    bool launched = false;
    GAT_CASE(4, 1, 16) GAT_CASE(4, 2, 16) GAT_CASE(4, 3, 16) GAT_CASE(4, 4, 16) GAT_CASE(4, 5, 16) GAT_CASE(4, 6, 16) GAT_CASE(4, 7, 16) GAT_CASE(4, 8, 16)
    GAT_CASE(2, 1, 8) GAT_CASE(2, 3, 8) GAT_CASE(2, 5, 8) GAT_CASE(2, 7, 8)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
