// k_gath: the fused graph-attention layer on two fp16 pieces per operand (inference, node values below 2^15) + launcher.
// Reference: FeatureAttentionLayer.forward modules.py:65-95, TemporalAttentionLayer.forward modules.py:166-193.
#include "mtadgat_gat_impl.h"

namespace mtadgat {

// ---------------------------------------------------------------------------
// k_gath: k_gat for the two-fp16-piece arithmetic (inference, node values below 2^15) on a vector-ALU diet.  k_gat is bound by
// VALU issue (profiles/r03_pmc_summary.txt: 72.9 k VALU instructions per temporal window, 44.9 k of them pair-grid), and 28 k of
// its instructions are not the pair grid.  Here the node vectors are split into their two fp16 pieces ONCE, when the window is
// staged (k_gat: again for every 32-column part, side and aggregation group -- 8 + 7 times per value), and kept in LDS as
// pieces, row-major [node][feature]:
//   * projection: the B operand of v_mfma_f32_32x32x16_f16 is two 8-byte LDS reads per piece, no VALU;
//   * the power-of-two weight scale S stays in L', R', c, d and leaves with one multiply per score;
//   * aggregation: the A operand (4 keys x 1 feature per lane) comes from the row-major pieces through the LDS transpose read
//     (ds_read_b64_tr_b16, round 5; rounds 3-4 gathered it with eight 16-bit reads and four merges per tile);
//     the softmax rows are split per 16-key group as before;
//   * staging: one index computation per 16-byte unit, exp with one rounding-error term.
// Same LDS budget as k_gat (pieces: 2 x 2 bytes per value), same pair grid (gat_tile), same launch geometry.  Round 5: also the
// training forward's kernel from 4096 windows (keeps the softmax rows in a.ATT, applies the attention dropout).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float gath_exp(float x) {      // e^x, x <= 0 (or -inf): the product x log2(e) in two pieces
    const float c_hi = 1.4426950216293335f;
    const float hi = x * c_hi;
    const float lo = __builtin_fmaf(x, c_hi, -hi);
    return __builtin_amdgcn_exp2f(hi) * __builtin_fmaf(0.6931471805599453f, lo, 1.0f);
}
// Developer build -DMTADGAT_GATH_STAMP (profiles/gath_timeline.py): lane 0 of every wave of GATH_STAMP_WINS mid-launch workgroups
// records s_memtime at the phase boundaries below; read back through mtadgat_debug_gath_stamps.  Not in the regular build.
#ifdef MTADGAT_GATH_STAMP
constexpr int GATH_STAMP_WINS = 64;
__device__ unsigned long long g_gath_stamp[2 * GATH_STAMP_WINS * 8 * 32];
#define GATH_STAMP(id)                                                                                                   \
    do {                                                                                                                 \
        const long sw_ = (long)blockIdx.x - (long)(a.nwin >> 1);                                                         \
        if (sw_ >= 0 && sw_ < GATH_STAMP_WINS && (threadIdx.x & 63) == 0)                                                \
            g_gath_stamp[(((CONV ? 0 : GATH_STAMP_WINS) + sw_) * 8 + (threadIdx.x >> 6)) * 32 + (id)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define GATH_STAMP(id) do { } while (0)
#endif
// (A <= 80-VGPR build with one operand register set -- three 8-wave workgroups per CU -- was measured at 12.0 against 9.6 ms for
// the two layers and is gone; so are staggered workgroup starts, which changed nothing: DESIGN.md section 4.)
//
// CONV (round 5): the workgroup of the TEMPORAL layer (nodes = the window's time steps, vt == 0) computes its window's convolution
// itself (reference ConvLayer.forward, modules.py:18-22, on the way into TemporalAttentionLayer.forward -- mtad_gat.py:67-70 is
// one dataflow).  k_conv_win's arithmetic, instruction for instruction (mtadgat_convw.hip: the window scaled by a power of two,
// two fp16 pieces between zero halos, three v_mfma_f32_32x32x16_f16 per 16 input channels and tap), spread over the workgroup's
// eight waves (one 32-row x 32-channel output tile each); the staged input pieces borrow the L' / R' region, which the
// projection only claims afterwards.  The epilogue writes h_cat[:, :F] -- the feature layer and the recurrence read it -- and
// drops the window's own pieces straight into Vh / Vl: the convolution's output never comes back from memory for this layer, the
// convolution kernel and its launch are gone from the forward, and its matrix work runs beside the pair grid of the CU's other
// workgroup.  The fp16 range guard becomes per window: a window whose largest convolution output reaches 2^15 sets its flag and
// leaves the layer to k_gat's bf16-piece build, which is enqueued behind this kernel and looks only at flagged windows.
template <int IBL, int JPL, int RJ, bool CONV>
__global__ __launch_bounds__(512, MTADGAT_GAT_MINW) void k_gath(const GatArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RI = 64 / RJ;
    constexpr int IBW = RI * IBL;                      // query rows per wave
#ifndef MTADGAT_GATH_QB2
#define MTADGAT_GATH_QB2 2
#endif
    // weight chunks held in registers per task batch (requested before the pair grid of the previous part, so they are there when
    // the projection phase begins).  The 4-row blocking (temporal layer at the flagship shape) has 128 registers in use with two;
    // the 2-row blocking (feature layer: 91 registers, 7 chunks per tile at W = 100) has room for six, but six measured slower than
    // two (both layers 10.70 vs 10.55 ms per 65 536 windows, same box: -DMTADGAT_GATH_QB2=6).
    constexpr int QB = IBL == 2 ? MTADGAT_GATH_QB2 : MTADGAT_GAT_QB3;
    if constexpr (!CONV)
        if (!(a.vmax != nullptr && __uint_as_float(*a.vmax) < 32768.f)) return;     // k_gat's bf16-piece build serves this launch
    // Wave priorities by phase (round 5): the kernel's waves share a SIMD's issue slots with the other workgroup's, and the
    // arbiter serves the older wave first -- usually one that sits in its VALU-bound pair grid, while a wave of the other workgroup
    // that is in a latency-bound phase (staging, convolution, projection, softmax, aggregation: 60 % of the kernel's time by the
    // knock-outs) waits for a slot between its round trips.  Everything but the pair grid runs at priority 3, the pair grid at 0:
    // the chains get their few instructions at once and the pair grid soaks up the rest.  Measured, both layers per 65 536
    // windows, same process: 10.65 -> 10.43 ms (8 runs each way); the reverse assignment 10.42 -> 10.79.
    __builtin_amdgcn_s_setprio(3);
    GATH_STAMP(0);
#ifdef MTADGAT_GATH_STAMP
    {
        const long sw_ = (long)blockIdx.x - (long)(a.nwin >> 1);
        if (sw_ >= 0 && sw_ < GATH_STAMP_WINS && (threadIdx.x & 63) == 0) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_gath_stamp[(((CONV ? 0 : GATH_STAMP_WINS) + sw_) * 8 + (threadIdx.x >> 6)) * 32 + 31] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    const long win = blockIdx.x;
    const int K = a.K, D = a.D;
    // the column order of the pack: GATv2 layers come in k_gath's compact order (a.E > 0: the non-negative group padded to 2 columns,
    // PB = the boundary, a multiple of 2; mtadgat_packdev.hip), everything else in whole 8-column tiles of one sign (PB = P8)
    int PB, PT;
    if (a.E > 0 && a.ord) {
        int P8_, PT_, npos_;
        gat_load_order3(a.ord, P8_, PT_, npos_);
        PB = (npos_ + 1) & ~1;
        PT = (PB + (a.E - npos_) + 7) & ~7;
    } else {
        gat_load_order(a.ord, a.P8, a.PT, PB, PT);
    }
    const int pvh = a.vld;                             // piece pitch in halfs
    const int KR = K + 1;                              // rows of the pieces: the nodes and one zero row (keys past K of a 16-key group)
    // waves that own query rows (the rest only project): n_full of them 4 RI = 16 rows (IBL per lane), n_short one row per lane
    // less (IBL - 1: 12 rows, or 8 with 8 lanes along the keys) -- 100 rows = 4 x 16 + 3 x 12, 55 = 3 x 16 + 8: no padded rows
    const int NWA = a.n_full + a.n_short;
    // Round 6, "run-ahead projection" (a.lr_buf > 0: the launcher found an idle wave, Q <= 4 and room for a second L' / R' buffer):
    // part p's pair grid reads buffer p & 1 while the workgroup's last wave -- it owns no query rows and used to wait at the
    // barrier -- projects part p + 1 into the other buffer, with the part's weight words (2 x Q chunks x 2 pieces = 64 registers
    // at Q = 4) fetched once per part instead of once per 32-node tile.  Parts 1 .. nparts - 1 lose their projection phase and one
    // of their two barriers: 3 x (2.6-3.1 k + ~0.3 k) of the temporal workgroup's ~114 k cycles per window (r06 timelines).
    const int lr_buf = CONV ? a.lr_buf : 0;            // (the projector's code only exists in the CONV build: the temporal layer's inference launch)
    const bool ahead = lr_buf > 0;
    const int Lrows = ahead ? K : NWA * IBW;           // (two buffers only fit with K rows of L': padded rows are never read by the pair grid)
    float* __restrict__ Ls = smem;
    float* __restrict__ Rs = Ls + Lrows * GAT_LLD;
    unsigned short* __restrict__ Vh = reinterpret_cast<unsigned short*>(smem + a.lr_floats);
    unsigned short* __restrict__ Vl = Vh + KR * pvh + 16;          // (+ 16 zero halfs: chunk reads of the last row run past its end when the pitch is below 16 Q)
    const int i = lane & 31, g = lane >> 5;            // MFMA roles
    const int lj = lane % RJ, li = lane / RJ;          // pair-grid roles

    const int NTn = (K + 31) >> 5;                    // node tiles
    const int ntask = 2 * NTn;                        // per part: query-side tiles then key-side tiles
    const int Q = a.Q;                                // 16-feature chunks incl. the ones column
    const int ptile = PB >> 3, ntile = PT >> 3;        // whole non-negative tiles (the next one may be mixed: PB & 7 columns of it)
    const int pmix = (PB & 7) >> 1;                    // non-negative 2-column steps of the mixed tile, 0 = there is none
    const int nparts = (PT >> 5) + 1;

    auto fresh_lane = [&]() -> int {                  // the lane id, recomputed where it is used: see the note at the tail
        int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };
    const f32x4* __restrict__ Wbase = a.Wp2;
    f32x4 w[QB][2];
    const bool rows_owner = wave < NWA;
    const bool full = wave < a.n_full;
    const int i0 = !rows_owner ? 0 : (full ? wave * IBW : a.n_full * IBW + (wave - a.n_full) * (IBW - RI));
    const int iblw = full ? IBL : IBL - 1;             // rows per lane of this wave
    float acc[IBL][JPL];
    auto wfetch = [&](const f32x4* __restrict__ wp, int u, int q) {
        w[u][0] = wp[((long)q * 2) * 64];
        w[u][1] = wp[((long)q * 2 + 1) * 64];
    };
    auto prefetch = [&](int part) {
        if (wave < ntask) {
            const int wtile = wave >= NTn ? a.NT_L + part : part;
            const f32x4* __restrict__ wp = Wbase + ((long)wtile * Q) * (64 * 2) + fresh_lane();
#pragma unroll
            for (int u = 0; u < QB; ++u) wfetch(wp, u, u < Q ? u : Q - 1);
        }
    };

    // ---- stage the window as fp16 pieces: Vh/Vl[node][feature], feature D = 1 (the projection bias is weight row D), the
    // other features up to 16 Q and the rows K .. Kp16 zero.  vt == 0: source rows are the nodes; vt == 1: source columns.
    const int nthr = blockDim.x;
    const int FP = 16 * Q < pvh ? 16 * Q : pvh;        // features of a piece row that exist
    int f0;                                            // first feature the generic fill below has to write
    if constexpr (CONV) {
        const GatConvIn& c = a.cv;
        const int W = K, F = D, Fq = c.Fq, taps = c.taps, pad = c.pad, pvx = c.pvx;
        const int nrows = W + taps - 1;                // staged input rows: the window between its zero halos
        unsigned short* __restrict__ Xh = reinterpret_cast<unsigned short*>(smem);
        unsigned short* __restrict__ Xl = Xh + (nrows + 2) * pvx;
        float* __restrict__ red = reinterpret_cast<float*>(Xl + (nrows + 2) * pvx);      // [8] input maxima, [2] scale, [8] output maxima
        const long s0 = c.gather ? (c.starts ? c.starts[win] : c.start0 + win * c.stride) : win * (long)W;
        const float* __restrict__ xw = c.X + s0 * F;
        const unsigned short* __restrict__ xw16 = reinterpret_cast<const unsigned short*>(c.X) + s0 * F;
        const int total = W * F;
        const bool vec = c.x_bf16 ? (reinterpret_cast<unsigned long>(xw16) & 7) == 0 : (reinterpret_cast<unsigned long>(xw) & 15) == 0;
        constexpr int MAXU = 3;                        // 16-byte units per thread (W F <= 6144, 512 threads: launcher)
        f32x4 v[MAXU];
        const int nunit = (total + 3) >> 2;
        float mx = 0.f;
        // the loader is picked once (wave-uniform) and issues all of a thread's units before anything looks at them: with the choice
        // inside the unit loop each unit was a load, a wait for it, and its maximum -- three memory round trips in a row per window
        const bool whole = vec && (total & 3) == 0;    // every unit is one aligned 16-byte (bf16 input: 8-byte) word
        if (whole && !c.x_bf16) {
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr;
                v[n] = *reinterpret_cast<const f32x4*>(xw + 4 * (u < nunit ? u : nunit - 1));
            }
        } else if (whole) {
            typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
            u32x2_ w2[MAXU];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr;
                w2[n] = *reinterpret_cast<const u32x2_*>(xw16 + 4 * (u < nunit ? u : nunit - 1));
            }
#pragma unroll
            for (int n = 0; n < MAXU; ++n)
                v[n] = f32x4{__uint_as_float(w2[n][0] << 16), __uint_as_float(w2[n][0] & 0xffff0000u), __uint_as_float(w2[n][1] << 16), __uint_as_float(w2[n][1] & 0xffff0000u)};
        } else if (c.x_bf16) {
            unsigned short h[MAXU][4];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr, uc = u < nunit ? u : nunit - 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[n][e] = xw16[4 * uc + e < total ? 4 * uc + e : total - 1];
            }
#pragma unroll
            for (int n = 0; n < MAXU; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] = __uint_as_float((unsigned)h[n][e] << 16);
        } else {
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr, uc = u < nunit ? u : nunit - 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[n][e] = xw[4 * uc + e < total ? 4 * uc + e : total - 1];
            }
        }
#pragma unroll
        for (int n = 0; n < MAXU; ++n) {
            const int u = tid + n * nthr;
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, (4 * u + e < total) ? fabsf(v[n][e]) : 0.f);
        }
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        GATH_STAMP(1);
        {   // zero halo rows, the two spare rows behind them and the channel padding [F, Fq) of the window's rows
            const int hw = pvx >> 1;                   // dwords per row
            for (int u = tid; u < 2 * pad * hw; u += nthr) {
                const int r = u / hw, cc = u - r * hw;
                const int row = r < pad ? r : nrows - 2 * pad + r;
                reinterpret_cast<unsigned*>(Xh + row * pvx)[cc] = 0u;
                reinterpret_cast<unsigned*>(Xl + row * pvx)[cc] = 0u;
            }
            for (int u = tid; u < 2 * hw; u += nthr) {
                reinterpret_cast<unsigned*>(Xh + nrows * pvx)[u] = 0u;
                reinterpret_cast<unsigned*>(Xl + nrows * pvx)[u] = 0u;
            }
            const int npadc = (Fq < pvx ? Fq : pvx) - F;
            for (int u = tid; u < W * npadc; u += nthr) {
                const int r = u / npadc, cc = F + (u - r * npadc);
                Xh[(pad + r) * pvx + cc] = 0;
                Xl[(pad + r) * pvx + cc] = 0;
            }
        }
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
            for (int w2 = 1; w2 < NW; ++w2) m = fmaxf(m, red[w2]);
            // sx = 2^(13 - floor(log2 m)): exponent field 267 - e (m = 0, denormal or not finite: 1)
            const unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
            const unsigned es = (e == 0u || e >= 254u) ? 127u : 267u - e;
            const unsigned ec = es < 1u ? 1u : (es > 253u ? 253u : es);
            red[8] = __uint_as_float(ec << 23);
            red[9] = __uint_as_float((254u - ec) << 23);
        }
        __syncthreads();
        GATH_STAMP(2);
        const float sx = red[8], sxi = red[9];
        {
            const float finv = 1.0f / (float)F;
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = tid + n * nthr;
                if (u < nunit) {
                    const int fl0 = 4 * u;
                    int row = (int)(((float)fl0 + 0.5f) * finv), col = fl0 - row * F;
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        // two elements at a time; a pair may straddle the end of a row
                        int r0 = row, c0 = col, r1 = row, c1 = col + 1;
                        if (c1 >= F) { r1 = row + 1; c1 = 0; }
                        unsigned hw_, lw_;
                        split_pair_h(v[n][2 * e2] * sx, v[n][2 * e2 + 1] * sx, hw_, lw_);
                        if (fl0 + 2 * e2 < total) { Xh[(pad + r0) * pvx + c0] = (unsigned short)hw_; Xl[(pad + r0) * pvx + c0] = (unsigned short)lw_; }
                        if (fl0 + 2 * e2 + 1 < total) { Xh[(pad + r1) * pvx + c1] = (unsigned short)(hw_ >> 16); Xl[(pad + r1) * pvx + c1] = (unsigned short)(lw_ >> 16); }
                        col += 2;
                        if (col >= F) { col -= F; ++row; if (col >= F) { col -= F; ++row; } }      // (F = 1: a pair is two rows)
                    }
                }
            }
        }
        __syncthreads();
        GATH_STAMP(3);
        // out[t][o] = sum_{tap, ch} w[o][tap][ch] x[t + tap - pad][ch]: A = weights (32 output channels), B = input rows (lane (i, g):
        // row of the tile + i, channels 16 cb + 4 g .. + 3 and + 8 .. -- mtadgat_device.h), K runs over taps x 16-channel chunks
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const int QF = Fq >> 4, QC = taps * QF;
        const int RT = (W + 31) >> 5;
        const float osc = c.wscale[1] * sxi;           // both factors are powers of two
        const int FPc = FP < 32 * c.NT ? FP : 32 * c.NT;
        float vmx = 0.f;
#ifndef MTADGAT_GATH_CONV_RPW
#define MTADGAT_GATH_CONV_RPW 1
#endif
        // A wave takes RPW row tiles of one 32-channel output tile (each tile's sum keeps its order: same bits).  RPW = 2 halves the
        // weight words the workgroup pulls from L2 (448 -> 224 KB per window) with four waves working -- and changes nothing: the
        // phase is a latency chain per wave (28 chunks, three in flight), 13.1 k cycles with RPW = 1 and 14.9 k with RPW = 2 in the
        // stamped build; both layers 10.51-10.54 (RPW 1) vs 10.55-10.61 ms (RPW 2) per 65 536 windows, same box; a ring of seven
        // chunks instead of four: 10.52-10.62 (profiles/r06_gath_experiments.txt).
        if (a.dbg & 8) __builtin_amdgcn_s_sleep(78);          // sensitivity probe: ~5 k idle cycles ahead of the convolution (results unchanged)
        constexpr int RPW = MTADGAT_GATH_CONV_RPW;
        const int RTP = (RT + RPW - 1) / RPW;
        for (int ctask = wave; ctask < RTP * c.NT; ctask += NW) {
            const int nb = ctask / RTP, rtp = ctask - nb * RTP;
            int tt[RPW], xoff[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                tt[r] = 32 * (RPW * rtp + r) + i;
                xoff[r] = (tt[r] < W ? tt[r] : nrows - taps + 1) * pvx + 4 * g;       // rows past the window read the spare zero rows
            }
            const f32x4* __restrict__ Wc = c.Wp + ((long)nb * QC) * (2 * 64) + lane;      // [tile][QC][2 pieces][64]
            f32x16 acc[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
            f32x4 cbias[4];                            // the epilogue's bias words, requested ahead of the weight stream
#pragma unroll
            for (int m = 0; m < 4; ++m) cbias[m] = *reinterpret_cast<const f32x4*>(c.bias + 32 * nb + 8 * m + 4 * g);
            // a ring of four chunks: the words of chunk q + 3 are requested before the MFMAs of chunk q (mtadgat_convw.hip)
#ifndef MTADGAT_GATH_CONV_RING
#define MTADGAT_GATH_CONV_RING 4
#endif
            constexpr int RING = MTADGAT_GATH_CONV_RING;
            f32x4 wr[RING][2];
            auto wload = [&](f32x4 (&wq)[2], int q) {
                const int qc = q < QC ? q : QC - 1;
                wq[0] = Wc[((long)qc * 2) * 64];
                wq[1] = Wc[((long)qc * 2 + 1) * 64];
            };
#pragma unroll
            for (int u = 0; u < RING - 1; ++u) wload(wr[u], u);
            int tap = 0, cb = 0;
#pragma unroll 1
            for (int q0 = 0; q0 < QC; q0 += RING) {
#pragma unroll
                for (int u = 0; u < RING; ++u) {
                    const int q = q0 + u;
                    wload(wr[(u + RING - 1) % RING], q + RING - 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q < QC) {
                        const int ko = tap * pvx + 16 * cb;
#pragma unroll
                        for (int r = 0; r < RPW; ++r) {
                            const unsigned short* __restrict__ ph = Xh + xoff[r] + ko;
                            const unsigned short* __restrict__ pl = Xl + xoff[r] + ko;
                            const u32x2 ha = *reinterpret_cast<const u32x2*>(ph), hb = *reinterpret_cast<const u32x2*>(ph + 8);
                            const u32x2 la = *reinterpret_cast<const u32x2*>(pl), lb = *reinterpret_cast<const u32x2*>(pl + 8);
                            const f32x4 xh = __builtin_bit_cast(f32x4, u4{ha[0], ha[1], hb[0], hb[1]});
                            const f32x4 xl = __builtin_bit_cast(f32x4, u4{la[0], la[1], lb[0], lb[1]});
                            acc[r] = mfma_h(wr[u][0], xl, acc[r]);
                            acc[r] = mfma_h(wr[u][1], xh, acc[r]);
                            acc[r] = mfma_h(wr[u][0], xh, acc[r]);
                        }
                        if (++cb == QF) { cb = 0; ++tap; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // epilogue: 1 / (S sx), bias, ReLU -> h_cat[:, :F] (+ the zero alignment padding of the row) and the window's pieces
            GATH_STAMP(4);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int t = tt[r];
                if (t < W) {
                    float* __restrict__ hrow = c.HCAT + (win * W + t) * (long)c.Dp;
                    if (nb == 0 && g == 0)
                        for (int cc = 3 * F; cc < c.Dp; ++cc) hrow[cc] = 0.f;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int col = 32 * nb + 8 * m + 4 * g;
                        const f32x4 bv = cbias[m];
                        f32x4 y;
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            y[s4] = fmaxf(__builtin_fmaf(acc[r][4 * m + s4], osc, bv[s4]), 0.f);
                            vmx = (col + s4 < F) ? fmaxf(vmx, y[s4]) : vmx;
                        }
                        if (col + 3 < F) {
                            *reinterpret_cast<f32x4*>(hrow + col) = y;
                        } else {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4)
                                if (col + s4 < F) hrow[col + s4] = y[s4];
                        }
                        if (col < FPc) {
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) y[s4] = col + s4 < D ? y[s4] : (col + s4 == D ? 1.f : 0.f);
                            unsigned h0, l0, h1, l1;
                            split_pair_h(y[0], y[1], h0, l0);
                            split_pair_h(y[2], y[3], h1, l1);
                            *reinterpret_cast<u32x2*>(Vh + t * pvh + col) = u32x2{h0, h1};
                            *reinterpret_cast<u32x2*>(Vl + t * pvh + col) = u32x2{l0, l1};
                        }
                    }
                }
            }
        }
        vmx = wave_max(vmx);
        if (lane == 0) red[10 + wave] = vmx;
        GATH_STAMP(5);
        __syncthreads();                               // every wave is done with the staged input: the region is the projection's now
        float wmax = red[10];
        for (int w2 = 1; w2 < NW; ++w2) wmax = fmaxf(wmax, red[10 + w2]);
        const bool big = !(wmax < 32768.f);
        if (tid == 0) {
            if (c.vmax && !(wmax <= __uint_as_float(*c.vmax))) atomicMax(c.vmax, __float_as_uint(wmax));
            if (c.flag) c.flag[win] = big ? 1 : 0;
        }
        GATH_STAMP(6);
        if (big) return;                               // (uniform) k_gat's bf16-piece build, enqueued behind this kernel, takes the window
        f0 = FPc;
    } else {
        const int srows = a.vt ? D : K, scols = a.vt ? K : D;
        const int UR = (scols + 3) >> 2;
        const int total = srows * UR;
        const float rinv = 1.0f / (float)UR;
        const float* __restrict__ vsrc = a.V + win * (long)srows * a.ldv;
        const int c4last = ((scols - 1) >> 2) << 2;
        constexpr int MAXU = 3;
        for (int base = 0; base < total; base += MAXU * nthr) {
            f32x4 v[MAXU];
            int rr[MAXU], cc[MAXU];
#pragma unroll
            for (int n = 0; n < MAXU; ++n) {
                const int u = base + tid + n * nthr;
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
                    unsigned h0, l0, h1, l1;
                    if (!a.vt) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = c4 + e < D ? t[e] : (c4 + e == D ? 1.f : 0.f);
                        split_pair_h(t[0], t[1], h0, l0);
                        split_pair_h(t[2], t[3], h1, l1);
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<u32x2*>(Vh + row * pvh + c4) = u32x2{h0, h1};
                        *reinterpret_cast<u32x2*>(Vl + row * pvh + c4) = u32x2{l0, l1};
                    } else {
                        split_pair_h(t[0], t[1], h0, l0);
                        split_pair_h(t[2], t[3], h1, l1);
                        const unsigned short hs[4] = {(unsigned short)h0, (unsigned short)(h0 >> 16), (unsigned short)h1, (unsigned short)(h1 >> 16)};
                        const unsigned short ls[4] = {(unsigned short)l0, (unsigned short)(l0 >> 16), (unsigned short)l1, (unsigned short)(l1 >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c4 + e < K) {
                                Vh[(c4 + e) * pvh + row] = hs[e];
                                Vl[(c4 + e) * pvh + row] = ls[e];
                            }
                    }
                }
            }
        }
        f0 = a.vt ? D : 4 * UR;
    }
    {
        // ones column and zero features of the real nodes (up to the pitch; chunk reads beyond it meet the next row: finite values
        // against zero weights), the zero row K and the 16 halfs behind each piece
        const int nf = FP - f0;
        if (nf > 0) {
            const float ninv = 1.0f / (float)nf;
            for (int u = tid; u < K * nf; u += nthr) {
                const int node = (int)(((float)u + 0.5f) * ninv), f = f0 + (u - node * nf);
                Vh[node * pvh + f] = f == D ? (unsigned short)0x3C00 : (unsigned short)0;
                Vl[node * pvh + f] = 0;
            }
        }
        for (int u = tid; u < ((pvh + 16) >> 1); u += nthr) {
            reinterpret_cast<unsigned*>(Vh + K * pvh)[u] = 0u;
            reinterpret_cast<unsigned*>(Vl + K * pvh)[u] = 0u;
        }
    }
    prefetch(0);
    __syncthreads();
    GATH_STAMP(7);

    lds_cptr lp[IBL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        lp[ii] = (lds_cptr)(Ls + (i0 + li + RI * ii) * GAT_LLD);
        asm volatile("" : "+v"(lp[ii]));
    }
    const lds_cptr rp = (lds_cptr)(Rs + lj * GAT_LLD);
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = 0.f;

    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    for (int part = 0; part < nparts; ++part) {
        // ---- MFMA phase: project this part's 32 + 32 columns for all nodes into L' / R' (scaled by S: the weights carry it)
        if (!ahead || part == 0) {
        for (int task = wave; task < ntask && !(a.dbg & 2); task += NW) {
            const bool keyside = task >= NTn;
            const int nt = keyside ? task - NTn : task;
            const int wtile = keyside ? a.NT_L + part : part;
            const int lane_p = fresh_lane();
            const int i = lane_p & 31, g = lane_p >> 5;          // MFMA roles
            const int node = nt * 32 + i;
            const unsigned short* __restrict__ vrh = Vh + (node < K ? node : K - 1) * pvh + 4 * g;
            const unsigned short* __restrict__ vrl = Vl + (node < K ? node : K - 1) * pvh + 4 * g;
            const f32x4* __restrict__ wp = Wbase + ((long)wtile * Q) * (64 * 2) + lane_p;
            if (task != wave) {
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
                        // the lane's eight features of the chunk: 16 q + 4 g .. + 3 and 16 q + 8 + 4 g .. + 3 (mtadgat_device.h)
                        const u32x2 ha = *reinterpret_cast<const u32x2*>(vrh + 16 * (qb + u)), hb = *reinterpret_cast<const u32x2*>(vrh + 16 * (qb + u) + 8);
                        const u32x2 la = *reinterpret_cast<const u32x2*>(vrl + 16 * (qb + u)), lb = *reinterpret_cast<const u32x2*>(vrl + 16 * (qb + u) + 8);
                        typedef unsigned u4 __attribute__((ext_vector_type(4)));
                        const f32x4 xh = __builtin_bit_cast(f32x4, u4{ha[0], ha[1], hb[0], hb[1]});
                        const f32x4 xl = __builtin_bit_cast(f32x4, u4{la[0], la[1], lb[0], lb[1]});
                        o = mfma_h(w[u][0], xl, o);
                        o = mfma_h(w[u][1], xh, o);
                        o = mfma_h(w[u][0], xh, o);
                        if (qb + QB + u < Q) wfetch(wp, u, qb + QB + u);
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
        }
        GATH_STAMP(8 + 3 * (part < 5 ? part : 4));
        if (!ahead || part == 0) __syncthreads();
        GATH_STAMP(9 + 3 * (part < 5 ? part : 4));
        const int boff = ahead ? (part & 1) * lr_buf : 0;        // this part's L' / R' buffer (floats)
        if constexpr (CONV) {
        if (ahead && wave == NW - 1 && part + 1 < nparts && !(a.dbg & 2)) {
            // (the projector keeps priority 3 like every latency phase: 0 / 1 / 2 measured the same, 9.96-10.11 vs 10.07-10.11 ms)
            // the run-ahead projector: all 2 NTn tiles of part + 1 into the other buffer -- a side at a time with that side's weight
            // words of the whole part in registers (Q x 2 pieces: 32 at Q = 4), every tile's products in the order of the projection
            // phase (same bits).  ~100 MFMAs + the LDS traffic of eight tiles: well inside a pair-grid part (10-16 k cycles)
            const int lane_p = fresh_lane();
            const int i = lane_p & 31, g = lane_p >> 5;
            constexpr int QA = 4;                      // (launcher: Q <= 4)
#pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                float* __restrict__ Db = (side ? Rs : Ls) + ((part + 1) & 1) * lr_buf;
                const f32x4* __restrict__ wsd = Wbase + ((long)((side ? a.NT_L : 0) + part + 1) * Q) * (64 * 2) + lane_p;
                f32x4 wS[QA][2];
#pragma unroll
                for (int q = 0; q < QA; ++q) {
                    const int qc = q < Q ? q : Q - 1;
                    wS[q][0] = wsd[((long)qc * 2) * 64];
                    wS[q][1] = wsd[((long)qc * 2 + 1) * 64];
                }
#pragma unroll 1
                for (int nt = 0; nt < NTn; ++nt) {
                    const int node = nt * 32 + i;
                    const unsigned short* __restrict__ vrh = Vh + (node < K ? node : K - 1) * pvh + 4 * g;
                    const unsigned short* __restrict__ vrl = Vl + (node < K ? node : K - 1) * pvh + 4 * g;
                    f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                    for (int q = 0; q < QA; ++q)
                        if (q < Q) {
                            const u32x2 ha = *reinterpret_cast<const u32x2*>(vrh + 16 * q), hb = *reinterpret_cast<const u32x2*>(vrh + 16 * q + 8);
                            const u32x2 la = *reinterpret_cast<const u32x2*>(vrl + 16 * q), lb = *reinterpret_cast<const u32x2*>(vrl + 16 * q + 8);
                            typedef unsigned u4 __attribute__((ext_vector_type(4)));
                            const f32x4 xh = __builtin_bit_cast(f32x4, u4{ha[0], ha[1], hb[0], hb[1]});
                            const f32x4 xl = __builtin_bit_cast(f32x4, u4{la[0], la[1], lb[0], lb[1]});
                            o = mfma_h(wS[q][0], xl, o);
                            o = mfma_h(wS[q][1], xh, o);
                            o = mfma_h(wS[q][0], xh, o);
                        }
                    if (node < K) {
                        float* __restrict__ dst = Db + node * GAT_LLD + 4 * g;
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            f32x2 v0, v1;
                            v0[0] = o[4 * m + 0]; v0[1] = o[4 * m + 1]; v1[0] = o[4 * m + 2]; v1[1] = o[4 * m + 3];
                            *reinterpret_cast<f32x2*>(dst + 8 * m) = v0;
                            *reinterpret_cast<f32x2*>(dst + 8 * m + 2) = v1;
                        }
                    }
                }
            }
        }
        }
        // ---- VALU phase: pairwise term over this part's k tiles (positive group first, then negative)
        int ntl = ntile - 4 * part;
        ntl = ntl > 4 ? 4 : ntl;
        if ((a.dbg & 16) && part == 0) {                      // sensitivity probe: 1 000 extra vector-ALU instructions per wave (results unchanged)
            float d0 = (float)part, d1 = 1.f, d2 = 2.f, d3 = 3.f;
#pragma unroll 1
            for (int it = 0; it < 250; ++it) {
                asm volatile("v_add_f32_e32 %0, %0, %0\n\tv_add_f32_e32 %1, %1, %1\n\tv_add_f32_e32 %2, %2, %2\n\tv_add_f32_e32 %3, %3, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            }
            if (d0 + d1 + d2 + d3 == 12345.f) a.out[0] = d0;
        }
        if ((a.dbg & 32) && part == 0) __builtin_amdgcn_s_sleep(78);      // sensitivity probe: ~5 k idle cycles in front of the first pair grid
        if (ntl > 0 && rows_owner && !(a.dbg & 1)) {
            __builtin_amdgcn_s_setprio(0);                       // the pair grid takes the issue slots nobody else wants (see the kernel's head)
            int npos = ptile - 4 * part;
            const bool mixed_here = pmix != 0 && npos >= 0 && npos < ntl;      // tile `ptile` lies in this part
            npos = npos < 0 ? 0 : (npos > ntl ? ntl : npos);
            lds_cptr rq = rp + boff;
            int kt = 0;
            if (full) {
                f32x2 lA[IBL], rA[JPL], lB[IBL], rB[JPL];
                lds_cptr lq[IBL];
#pragma unroll
                for (int ii = 0; ii < IBL; ++ii) lq[ii] = lp[ii] + boff;
                gat_load<IBL, JPL, RJ>(lA, rA, lq, rq, 0);
#pragma unroll 1
                for (; kt < npos; ++kt) {
                    gat_tile<IBL, JPL, RJ, false>(acc, lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                    rq += 8;
                }
                if (mixed_here) {                                // the tile on the sign boundary (compact order): sign per step, as a scalar
                    gat_tile_s<IBL, JPL, RJ>(acc, lA, rA, lB, rB, lq, rq, pmix);
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                    rq += 8;
                    ++kt;
                }
#pragma unroll 1
                for (; kt < ntl; ++kt) {
                    gat_tile<IBL, JPL, RJ, true>(acc, lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IBL; ++ii) lq[ii] += 8;
                    rq += 8;
                }
            } else {
                constexpr int IS = IBL - 1;             // the short block: the lane's last row belongs to the next wave
                float (&accs)[IS][JPL] = reinterpret_cast<float (&)[IS][JPL]>(acc);
                f32x2 lA[IS], rA[JPL], lB[IS], rB[JPL];
                lds_cptr lq[IS];
#pragma unroll
                for (int ii = 0; ii < IS; ++ii) lq[ii] = lp[ii] + boff;
                gat_load<IS, JPL, RJ>(lA, rA, lq, rq, 0);
#pragma unroll 1
                for (; kt < npos; ++kt) {
                    gat_tile<IS, JPL, RJ, false>(accs, lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IS; ++ii) lq[ii] += 8;
                    rq += 8;
                }
                if (mixed_here) {
                    gat_tile_s<IS, JPL, RJ>(accs, lA, rA, lB, rB, lq, rq, pmix);
#pragma unroll
                    for (int ii = 0; ii < IS; ++ii) lq[ii] += 8;
                    rq += 8;
                    ++kt;
                }
#pragma unroll 1
                for (; kt < ntl; ++kt) {
                    gat_tile<IS, JPL, RJ, true>(accs, lA, rA, lB, rB, lq, rq);
#pragma unroll
                    for (int ii = 0; ii < IS; ++ii) lq[ii] += 8;
                    rq += 8;
                }
            }
        }
        __builtin_amdgcn_s_setprio(3);
        GATH_STAMP(10 + 3 * (part < 5 ? part : 4));
        if (part + 1 < nparts) {
            if (!ahead) prefetch(part + 1);
            __syncthreads();
        }
    }
    // Round 6: the (K, K) attention bias (modules.py:85 / :184), all 4 x JPL words of the lane requested HERE in one batch -- the
    // pair grid's operand registers are free now -- so that they travel while the c / d columns are read and the workgroup meets
    // at the barrier below.  Requested inside the softmax's row loop (rounds 1-5) the compiler waited for them row by row: the
    // r06a timeline (profiles/gath_timeline.py) put 13.7 k (temporal) / 6.3 k (feature) cycles on the softmax phase.
    float bv[IBL][JPL];
    if (rows_owner) {
        const int lb = fresh_lane();                   // (roles from a fresh lane id: held across the pair grid they would spill)
        const int lj = lb % RJ, li = lb / RJ;
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + li + RI * ii;
            const int irc = irow < K ? irow : K - 1;
#pragma unroll
            for (int jj = 0; jj < JPL; ++jj) {
                const int j = lj + RJ * jj;
                bv[ii][jj] = a.bias ? a.bias[(long)irc * K + (j < K ? j : K - 1)] : 0.f;
            }
        }
    }
    float cv[IBL], dv[JPL];
    {
        const int col = (PT & 31) + (ahead ? ((nparts - 1) & 1) * lr_buf : 0);
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) cv[ii] = lp[ii][col];
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) dv[jj] = rp[jj * RJ * GAT_LLD + col];
    }
    __syncthreads();
    GATH_STAMP(23);
    if (a.dbg & 4) { if (acc[0][0] == 12345.f) a.out[0] = cv[0] + dv[0]; return; }       // (uniform)
    static_assert(IBW == 16, "one 16-row MFMA group per wave");
    constexpr int DTMAX = 8;                           // D <= 128 (plan)
    const int DT = (D + 15) >> 4;
    // The lane's roles for the tail, from a fresh lane id: held across the pair grid they cost registers the 128-VGPR budget does
    // not have -- the compiler spilled them (11 dwords per lane = 22 KB of scratch writes per window, which the round-4 counters
    // showed as "45 KB written for a 22 KB output")
    int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane_t));
    const int ljt = lane_t % RJ, lit = lane_t / RJ;
    const int nr = lane_t & 15, kb = lane_t >> 4;
    f32x4 o[DTMAX];
#pragma unroll
    for (int dt = 0; dt < DTMAX; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rows_owner) {                                  // (the waves that only project wait at the barrier below)
    // ---- scores -> softmax over j (reference modules.py:85-89 / :184-188); S leaves the scores here
    const float sinv = a.scale2[1];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        const int irow = i0 + lit + RI * ii;
        const bool rowok = ii < iblw && irow < K;
        const int irc = irow < K ? irow : K - 1;
        float e[JPL];
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const int j = ljt + RJ * jj;
            const float b = bv[ii][jj];
            float v;
            if (a.v1) {
                v = (acc[ii][jj] + cv[ii] + dv[jj]) * sinv;
                v = fmaxf(v, 0.f) + a.alpha * fminf(v, 0.f) + b;
            } else {
                v = __builtin_fmaf(acc[ii][jj] + cv[ii] + dv[jj], sinv, b);
            }
            v = j < K ? v : -INFINITY;
            e[jj] = v;
            m = fmaxf(m, v);
        }
        m = row_max<RJ>(m);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            e[jj] = (ljt + RJ * jj < K) ? gath_exp(e[jj] - m) : 0.f;
            sum += e[jj];
        }
        sum = row_sum<RJ>(sum);
        const float inv = soft_rcp(sum);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = rowok ? e[jj] * inv : 0.f;
    }

    GATH_STAMP(24);
    if (a.ATT || a.drop.thresh) {                      // training forward: keep the softmax rows, drop attention entries (counter-based mask)
        const unsigned key = drop_window_key(a.drop, a.drop_stream, win);
#pragma unroll
        for (int ii = 0; ii < IBL; ++ii) {
            const int irow = i0 + lit + RI * ii;
            if (ii < iblw && irow < K) {
                float* __restrict__ ap = a.ATT ? a.ATT + (win * K + irow) * (long)K : nullptr;
#pragma unroll
                for (int jj = 0; jj < JPL; ++jj) {
                    const int j = ljt + RJ * jj;
                    if (j < K) {
                        if (ap) ap[j] = acc[ii][jj];
                        if (a.drop.thresh) acc[ii][jj] *= drop_keep(key, (unsigned)(irow * K + j), a.drop.thresh) ? a.drop.keep_scale : 0.f;
                    }
                }
            }
        }
    }

    // ---- aggregation h_i = sigmoid(sum_j att_ij V_j) as out^T = V^T att^T on v_mfma_f32_16x16x16_f16, three terms per product
    // (k_gat); the softmax rows go through this wave's slice of the (now free) Ls / Rs region 64 keys at a time and are split per
    // 16-key group, the node values come as packed fp16 pieces straight from LDS
    constexpr int APP = 36;                            // pitch of the restaged rows: 32 keys per pass
    float* __restrict__ att = Ls + wave * (IBW * APP);
    constexpr int JPP = 32 / RJ;                       // key registers per 32-key pass
    constexpr int PASSES = (JPL + JPP - 1) / JPP;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int lo_off = (int)(Vl - Vh);
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass * 32 < K) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
                for (int j4 = 0; j4 < JPP; ++j4)
                    att[(lit + RI * ii) * APP + ljt + RJ * j4] = (JPP * pass + j4 < JPL) ? acc[ii][(JPP * pass + j4 < JPL) ? JPP * pass + j4 : 0] : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int jn = min(32, K - pass * 32);
            const int ngrp = (jn + 15) >> 4;
            for (int grp = 0; grp < ngrp; ++grp) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(att + nr * APP + 16 * grp + 4 * kb);
                unsigned h0, l0, h1, l1;
                split_pair_h(bq[0], bq[1], h0, l0);
                split_pair_h(bq[2], bq[3], h1, l1);
                const f16x4 bhh = __builtin_bit_cast(f16x4, u32x2{h0, h1}), bll = __builtin_bit_cast(f16x4, u32x2{l0, l1});
                // A operand = four keys x one feature per lane: V[key0 + 0..3][16 dt + nr].  The pieces lie [node][feature], so
                // this is a transposed access: ds_read_b64_tr_b16 (gfx950) hands lane nr column nr of the 4-key x 16-feature
                // block whose 8-byte row segments the 16 lanes of the group address -- lane nr: key (nr >> 2), features
                // 4 (nr & 3) .. + 3 -- one LDS instruction per piece and feature tile where the 16-bit gather took four reads and
                // two merges.  Keys past K read the zero row.
                const int key0 = pass * 32 + 16 * grp + 4 * kb;
                unsigned r[DTMAX][4];
                {
                    const int kr = key0 + (nr >> 2);
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) s16x4* lds_s4;
                    const unsigned short* __restrict__ pr = Vh + (kr < K ? kr : K) * pvh + 4 * (nr & 3);
#pragma unroll
                    for (int dt = 0; dt < DTMAX; ++dt)
                        if (dt < DT) {
                            const u32x2 th = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pr + 16 * dt)));
                            const u32x2 tl = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pr + lo_off + 16 * dt)));
                            r[dt][0] = th[0]; r[dt][1] = th[1]; r[dt][2] = tl[0]; r[dt][3] = tl[1];
                        }
                }
#pragma unroll
                for (int dt = 0; dt < DTMAX; ++dt)
                    if (dt < DT) {
                        const f16x4 ahh = __builtin_bit_cast(f16x4, u32x2{r[dt][0], r[dt][1]}), all_ = __builtin_bit_cast(f16x4, u32x2{r[dt][2], r[dt][3]});
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahh, bll, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(all_, bhh, o[dt], 0, 0, 0);
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ahh, bhh, o[dt], 0, 0, 0);
                    }
            }
        }
    }
    }   // rows_owner
    GATH_STAMP(25);
    // ---- output.  The layer's K x D result leaves through LDS (round 5): every wave drops its sigmoid values into one tile laid
    // out like the destination -- rows along the output's long stride, the unit-stride index inside a row -- and the workgroup
    // then writes whole rows, 256 consecutive bytes per wave instruction.  Before, a lane stored the four values it held: 16-byte
    // pieces that start 56 bytes into a 64-byte sector for the temporal layer (45 KB written for a 22 KB output) and, for the
    // feature layer -- whose node index is the output's unit-stride index -- four separate 4-byte stores 704 bytes apart.
    // The tile takes the L' / R' region and runs on into the pieces: nobody reads either after the barrier.
    __syncthreads();
    {
        float* __restrict__ otile = smem;
        const bool nodes_minor = a.so_i == 1;            // feature layer: out[.. + node + d * so_d]; temporal: out[.. + node * so_i + d]
        const int R = nodes_minor ? D : K, C = nodes_minor ? K : D;      // rows x unit-stride columns of the destination block
        if (rows_owner) {
            const int row = i0 + nr;
            const bool rv = nr < RI * iblw && row < K;
#pragma unroll
            for (int dt = 0; dt < DTMAX; ++dt)
                if (dt < DT) {
                    const int d0 = 16 * dt + 4 * kb;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rv && d0 + r < D) otile[nodes_minor ? (d0 + r) * C + row : row * C + d0 + r] = gate_sigmoid(o[dt][r]);
                }
        }
        GATH_STAMP(26);
        __syncthreads();
        GATH_STAMP(27);
        const long rstride = nodes_minor ? a.so_d : a.so_i;
        float* __restrict__ obase = a.out + win * a.so_w;
        // a wave instruction stays inside one destination row (its 64-byte sectors are touched by one request each)
        for (int r = wave; r < R; r += NW)
            for (int c = lane_t; c < C; c += 64) obase[(long)r * rstride + c] = otile[r * C + c];
    }
    GATH_STAMP(28);
}

#ifdef MTADGAT_GATH_STAMP
extern "C" int mtadgat_debug_gath_stamps(unsigned long long* dst, size_t n) {
    if (n > sizeof(g_gath_stamp) / sizeof(unsigned long long)) n = sizeof(g_gath_stamp) / sizeof(unsigned long long);
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gath_stamp), n * sizeof(unsigned long long));
}
#endif

#define GATH_CASE(I, J, RJ)                                                                     \
    if (IBL == I && JPL == J && rj == RJ) {                                                     \
        const void* fn_ = conv ? reinterpret_cast<const void*>(&k_gath<I, J, RJ, true>) : reinterpret_cast<const void*>(&k_gath<I, J, RJ, false>); \
        if (lds_bytes > 64 * 1024) {                                                            \
            hipError_t e_ = hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            if (e_ != hipSuccess) return (int)e_;                                               \
        }                                                                                       \
        if (conv) hipLaunchKernelGGL((k_gath<I, J, RJ, true>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);  \
        else hipLaunchKernelGGL((k_gath<I, J, RJ, false>), dim3(grid), dim3(64 * nw), lds_bytes, s, a);      \
        launched = true;                                                                        \
    }

// can the workgroup compute its window's convolution itself?  (temporal layer: nodes = time steps; what k_conv_win needs, the
// staged input inside the L' / R' region, one 16-byte unit in three per thread of eight waves)
bool gath_conv_applies(const GatArgs& a, int nw, int F, int W) {
    const GatConvIn& c = a.cv;
    if (a.vt != 0 || a.K != W || a.D != F || nw != 8 || !c.X || !c.HCAT || !c.Wp || !c.bias || !c.wscale) return false;
    if (c.taps != 2 * c.pad + 1 || c.NT > 2 || c.NT < 1 || W > 128 || W < 1 || (long)W * F > 6144) return false;
    if ((c.Fq & 15) != 0 || c.Fq < F || (c.Dp & 3) != 0 || 32 * c.NT < F) return false;
    if (c.pvx != conv_win_pitch(F, c.Fq)) return false;
    return conv_win_lds(W, F, c.Fq, c.taps) + 24 * sizeof(float) <= (size_t)a.lr_floats * sizeof(float);
}

// the fp16-piece build of the fused layer (a.vld = piece pitch in halfs, a.lr_floats as for k_gat, a.Q = 16-feature chunks);
// conv: the workgroup computes the window's convolution first (a.cv; a.V is not read)
int launch_gath(const GatArgs& a, int IBL, int JPL, int rj, int nw, size_t lds_bytes, bool conv, hipStream_t s) {
    if (a.nwin <= 0) return 0;
    if (rj * JPL < a.K || nw > 8 || a.n_full + a.n_short > nw || a.n_full * 16 + a.n_short * (16 - 64 / rj) < a.K) return -2;
    if (conv && !gath_conv_applies(a, nw, a.D, a.K)) return -2;
    const unsigned grid = (unsigned)a.nwin;
    bool launched = false;
    GATH_CASE(4, 1, 16) GATH_CASE(4, 2, 16) GATH_CASE(4, 3, 16) GATH_CASE(4, 4, 16) GATH_CASE(4, 5, 16) GATH_CASE(4, 6, 16) GATH_CASE(4, 7, 16) GATH_CASE(4, 8, 16)
    GATH_CASE(2, 1, 8) GATH_CASE(2, 3, 8) GATH_CASE(2, 5, 8) GATH_CASE(2, 7, 8) GATH_CASE(2, 9, 8) GATH_CASE(2, 11, 8) GATH_CASE(2, 13, 8) GATH_CASE(2, 15, 8)
    if (!launched) return -2;
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
