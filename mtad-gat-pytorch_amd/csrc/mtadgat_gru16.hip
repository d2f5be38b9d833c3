// k_gru16 / k_gru16_bwd: the GRU recurrences at small batches (the reference's own: Predictor batch 256,
// prediction.py:31; --bs 256, args.py:47), forward and back-propagation through time.  fp32.
//
// At 256 windows the 100-step recurrence is a latency chain on a handful of CUs.  k_gru_split gives a 32-window
// group to one CU with one wave per 32-unit hidden tile: 5 tiles on 4 SIMDs (one SIMD carries two), and every wave
// streams its 3 x 20 weight chunks from L2 in every step.  Here a workgroup owns 16 windows and a wave a 16-unit
// tile on v_mfma_f32_16x16x4_f32 (A = weights [unit][k], B = h [k][window], D = [unit][window]):
//   * twice as many workgroups (CUs) work on a batch, ten waves spread 3/3/2/2 over the SIMDs;
//   * the wave's weights -- 3 gates x Hp16/4 k-steps, one float per lane and instruction -- fit the register file
//     (120 VGPRs at H = 150) and are loaded ONCE for all steps: nothing is streamed inside the step loop;
//   * h_{t-1} is exchanged through a double-buffered LDS image hs[k][window] (one barrier per step); lane (n, kb)
//     reads hs[4 s + kb][n]: 64 consecutive floats per instruction, conflict free;
//   * the input products of all steps come pre-computed (k_rowgemm for the GRU layer, k_xproj_dec for the decoder's
//     folded input, modules.py:279) as the accumulators' initial values, the per-step Linear of the decoder runs as
//     one row GEMM over the stored states afterwards.
// Same gate arithmetic and saved quantities (r, z, n, q) as k_gru_split; results differ from it by summation order only.
#include "mtadgat_device.h"

namespace mtadgat {

// KSM: k-steps of 4 held in registers (>= KS; instantiated for a few sizes, 38 = the reference's default hidden size 150)

// lane (n = lane & 15, rb = lane >> 4) of wave `tile`: window n of the group, units 16 tile + 4 rb .. + 3
template <bool SAVE, int KSM>
__global__ __launch_bounds__(640) void k_gru16(const Gru16Args a) {
    extern __shared__ __attribute__((aligned(16))) float hs_all[];       // [2][Hp16][16]
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, rb = lane >> 4;
    const long win = (long)blockIdx.x * 16 + n;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Hp = a.Hp, KS = a.KS, Hp16 = 16 * a.NT16;
    const int unit0 = 16 * tile + 4 * rb;
    const bool padw = tile == a.NT16 - 1 && Hp > Hp16;       // Hp is a multiple of 32: one more 16-column group of zeros
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // weights of this wave, once: wreg[g][s] = W_h{g}[16 tile + (lane & 15)][4 s + (lane >> 4)]
    float wreg[3][KSM];
    {
        const float* __restrict__ wp = a.W16 + ((long)tile * 3) * KS * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < KSM; ++s) wreg[g][s] = s < KS ? wp[((long)g * KS + s) * 64] : 0.f;
    }
    const f32x4 bhn = *reinterpret_cast<const f32x4*>(a.bias + 3 * Hp + unit0);
    float* __restrict__ hs0 = hs_all;
    float* __restrict__ hs1 = hs_all + Hp16 * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) hs0[(unit0 + r) * 16 + n] = 0.f;
    f32x4 hown = {0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xrow = a.XP + winc * (long)T * (3L * Hp) + unit0;
    f32x4 xr = *reinterpret_cast<const f32x4*>(xrow);
    f32x4 xz = *reinterpret_cast<const f32x4*>(xrow + Hp);
    f32x4 xn = *reinterpret_cast<const f32x4*>(xrow + 2 * Hp);
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const float* __restrict__ hcur = (t & 1) ? hs1 : hs0;
        float* __restrict__ hnxt = (t & 1) ? hs0 : hs1;
        f32x4 ar = xr, az = xz, anh = bhn;
        const f32x4 anx = xn;
        {   // next step's input products: in flight during this step's MFMAs
            const int tn = t + 1 < T ? t + 1 : t;
            const float* __restrict__ p = xrow + (long)tn * (3L * Hp);
            xr = *reinterpret_cast<const f32x4*>(p);
            xz = *reinterpret_cast<const f32x4*>(p + Hp);
            xn = *reinterpret_cast<const f32x4*>(p + 2 * Hp);
        }
        const float* __restrict__ hb = hcur + rb * 16 + n;
#pragma unroll
        for (int s = 0; s < KSM; ++s)
            if (s < KS) {
                const float b = hb[s * 64];
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][s], b, ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[1][s], b, az, 0, 0, 0);
                anh = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[2][s], b, anh, 0, 0, 0);
            }
        f32x4 rg, zg, ng;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rg[r] = gate_sigmoid(ar[r]);
            zg[r] = gate_sigmoid(az[r]);
            ng[r] = gate_tanh(anx[r] + rg[r] * anh[r]);
            hown[r] = __builtin_fmaf(zg[r], hown[r] - ng[r], ng[r]);          // (1 - z) n + z h
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hnxt[(unit0 + r) * 16 + n] = hown[r];
        if (win < a.B) {
            if (a.Seq) {
                float* __restrict__ sp = a.Seq + (win * T + t) * (long)Hp + unit0;
                *reinterpret_cast<f32x4*>(sp) = hown;
                if (padw) *reinterpret_cast<f32x4*>(sp + 16) = zero4;          // columns [Hp16, Hp): consumers read whole rows
            }
            if (SAVE) {
                float* __restrict__ gp = a.Gates + (win * T + t) * (4L * Hp) + unit0;
                *reinterpret_cast<f32x4*>(gp) = rg;
                *reinterpret_cast<f32x4*>(gp + Hp) = zg;
                *reinterpret_cast<f32x4*>(gp + 2 * Hp) = ng;
                *reinterpret_cast<f32x4*>(gp + 3 * Hp) = anh;
                if (padw) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(gp + g * Hp + 16) = zero4;
                }
            }
        }
        __syncthreads();
    }
    if (a.Hend && win < a.B) {
        // the caller's rows may be unpadded (ldhe = H): element stores, columns below ncol only
        float* __restrict__ hp = a.Hend + win * a.ldhe;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (unit0 + r < a.ncol) hp[unit0 + r] = hown[r];
        if (padw)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (unit0 + 16 + r < a.ncol) hp[unit0 + 16 + r] = 0.f;
    }
}

// BPTT with the same decomposition: W_hh^T register resident (A = W_hh^T [unit j][k = u] per gate block), the three
// gate-gradient blocks of all units exchanged through a double-buffered LDS image.
template <int KSM>
__global__ __launch_bounds__(640) void k_gru16_bwd(const Gru16BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float das_all[];      // [2][3][Hp16][16]
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, rb = lane >> 4;
    const long win = (long)blockIdx.x * 16 + n;
    const long winc = win < a.B ? win : a.B - 1;
    const int T = a.T, Hp = a.Hp, KS = a.KS, Hp16 = 16 * a.NT16;
    const int unit0 = 16 * tile + 4 * rb;

    float wreg[3][KSM];
    {
        const float* __restrict__ wp = a.W16T + ((long)tile * 3) * KS * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < KSM; ++s) wreg[g][s] = s < KS ? wp[((long)g * KS + s) * 64] : 0.f;
    }
    f32x4 dh = {0.f, 0.f, 0.f, 0.f};
    if (a.DHend) dh = *reinterpret_cast<const f32x4*>(a.DHend + winc * a.ldde + unit0);
    const int blk = Hp16 * 16;

    for (int t = T - 1; t >= 0; --t) {
        const long row = winc * T + t;
        const float* __restrict__ gp = a.Gates + row * (4L * Hp) + unit0;
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(gp);
        const f32x4 z4 = *reinterpret_cast<const f32x4*>(gp + Hp);
        const f32x4 n4 = *reinterpret_cast<const f32x4*>(gp + 2 * Hp);
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(gp + 3 * Hp);
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(a.Seq + (row - (t > 0 ? 1 : 0)) * (long)Hp + unit0);
        if (a.DHseq) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.DHseq + row * a.lddh + unit0);
#pragma unroll
            for (int r = 0; r < 4; ++r) dh[r] += v[r];
        }
        const float hmask = t > 0 ? 1.f : 0.f;
        f32x4 dan, dar, daz, dnh, dhz;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dv = dh[r], rr = r4[r], z = z4[r], nn = n4[r], q = q4[r], hprev = h4[r] * hmask;
            dan[r] = dv * (1.f - z) * (1.f - nn * nn);
            daz[r] = dv * (hprev - nn) * z * (1.f - z);
            dar[r] = dan[r] * q * rr * (1.f - rr);
            dnh[r] = dan[r] * rr;
            dhz[r] = dv * z;
        }
        if (win < a.B) {
            float* __restrict__ op = a.DA + row * (4L * Hp) + unit0;
            *reinterpret_cast<f32x4*>(op) = dan;
            *reinterpret_cast<f32x4*>(op + Hp) = dar;
            *reinterpret_cast<f32x4*>(op + 2 * Hp) = daz;
            *reinterpret_cast<f32x4*>(op + 3 * Hp) = dnh;
            if (tile == a.NT16 - 1 && Hp > Hp16) {       // columns [Hp16, Hp): the weight-gradient and input-gradient GEMMs read whole rows
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(op + g * Hp + 16) = zero4;
            }
        }
        float* __restrict__ dcur = das_all + ((T - 1 - t) & 1) * 3 * blk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dcur[(unit0 + r) * 16 + n] = dar[r];
            dcur[blk + (unit0 + r) * 16 + n] = daz[r];
            dcur[2 * blk + (unit0 + r) * 16 + n] = dnh[r];
        }
        __syncthreads();
        f32x4 a0 = dhz, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
        const float* __restrict__ db = dcur + rb * 16 + n;
#pragma unroll
        for (int s = 0; s < KSM; ++s)
            if (s < KS) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[0][s], db[s * 64], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[1][s], db[blk + s * 64], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[2][s], db[2 * blk + s * 64], a2, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[r] = a0[r] + a1[r] + a2[r];
        // the next step writes the other buffer; this one is overwritten two steps later, after the next barrier
    }
}

// decoder input products of all steps: XP[(b, t)][g Hp + u] = bias_g[u] + sum_{c < 8} fold[t][g][u][c] * hend[b][m0[t] + c]
// (reference modules.py:279: x_t[j] = h_end[(t H + j) / W]; the per-step column sums of W_ih are folded on the host)
__global__ void k_xproj_dec(const float* __restrict__ hend, long ldh, int Hin, const float* __restrict__ fold, const int* __restrict__ m0,
                            const float* __restrict__ bias, int Hp, int T, long B, float* __restrict__ XP) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = B * T * Hp;
    if (idx >= total) return;
    const long row = idx / Hp;
    const int u = (int)(idx - row * Hp);
    const long b = row / T;
    const int t = (int)(row - b * T);
    const int lo = m0[t];
    float hv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) hv[c] = hend[b * ldh + (lo + c < Hin ? lo + c : Hin - 1)];      // columns past Hin carry zero weights
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float* __restrict__ f = fold + (((long)t * 3 + g) * Hp + u) * 8;
        float acc = bias[g * Hp + u];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc = __builtin_fmaf(f[c], hv[c], acc);
        XP[row * (3L * Hp) + g * Hp + u] = acc;
    }
}

static size_t g16_lds(int NT16, bool bwd) { return (size_t)(bwd ? 6 : 2) * (16 * NT16) * 16 * sizeof(float); }

int launch_gru16(const Gru16Args& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (a.KS < 1 || a.KS > 40 || a.NT16 < 1 || a.NT16 > 10 || 4 * a.KS > 16 * a.NT16) return -2;
    const unsigned grid = (unsigned)((a.B + 15) / 16);
    const size_t lds = g16_lds(a.NT16, false);
    auto go = [&](auto ksm) {
        constexpr int KSM = decltype(ksm)::value;
        if (a.Gates)
            hipLaunchKernelGGL((k_gru16<true, KSM>), dim3(grid), dim3(64 * a.NT16), lds, s, a);
        else
            hipLaunchKernelGGL((k_gru16<false, KSM>), dim3(grid), dim3(64 * a.NT16), lds, s, a);
    };
    if (a.KS <= 16) go(std::integral_constant<int, 16>());
    else if (a.KS <= 32) go(std::integral_constant<int, 32>());
    else if (a.KS <= 38) go(std::integral_constant<int, 38>());
    else go(std::integral_constant<int, 40>());
    LAUNCH_CHECK();
    return 0;
}

int launch_gru16_bwd(const Gru16BwdArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (a.KS < 1 || a.KS > 40 || a.NT16 < 1 || a.NT16 > 10 || 4 * a.KS > 16 * a.NT16) return -2;
    const size_t lds = g16_lds(a.NT16, true);       // at most 61 440 bytes (10 tiles)
    auto go = [&](auto ksm) {
        constexpr int KSM = decltype(ksm)::value;
        hipLaunchKernelGGL((k_gru16_bwd<KSM>), dim3((unsigned)((a.B + 15) / 16)), dim3(64 * a.NT16), lds, s, a);
    };
    if (a.KS <= 16) go(std::integral_constant<int, 16>());
    else if (a.KS <= 32) go(std::integral_constant<int, 32>());
    else if (a.KS <= 38) go(std::integral_constant<int, 38>());
    else go(std::integral_constant<int, 40>());
    LAUNCH_CHECK();
    return 0;
}

int launch_xproj_dec(const float* hend, long ldh, int Hin, const float* fold, const int* m0, const float* bias, int Hp, int T, long B,
                     float* XP, hipStream_t s) {
    const long total = B * T * Hp;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_xproj_dec, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hend, ldh, Hin, fold, m0, bias, Hp, T, B, XP);
    LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------
// k_gru1 / k_gru1_bwd: one window per workgroup (persistent over the batch), for batches of up to a few windows per
// CU.  The fp32 MFMA rate equals the fp32 VALU rate on this chip (64 FLOP / clk / SIMD), and the vector pipe has no
// minimum N: a workgroup keeps W_hh in the registers of its (at most 8) waves -- lane = one of the 3H gate rows, one
// VGPR per hidden index -- and multiplies it with the ONE state vector of its window, so 256 windows use 256 CUs for
// 100 short steps instead of 16 CUs for 100 long ones.  The state reaches the multiply through DPP: every 16-lane
// row holds h[16 c .. 16 c + 15] in one VGPR per c, and `v_fmac_f32_dpp acc, h_c, w_k row_newbcast:i` feeds lane i
// of the row to all its lanes -- one instruction per hidden index, no LDS read per index.
//   per step: FMA phase (pre-activations of the 3H rows -> LDS) | barrier | gate phase (lanes < Hp: one hidden unit
//   each, state in a register; h_t -> LDS, global stores) | barrier.
// ---------------------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void fmac_bcast(float& acc, float h, float w) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(w), "n"(I));
}
template <int KSM, int C = 0>
__device__ __forceinline__ void fma_rows(float (&acc)[4], const float (&hreg)[KSM], const float (&w)[16 * KSM]) {
    if constexpr (C < KSM) {
        fmac_bcast<0>(acc[0], hreg[C], w[16 * C + 0]);   fmac_bcast<1>(acc[1], hreg[C], w[16 * C + 1]);
        fmac_bcast<2>(acc[2], hreg[C], w[16 * C + 2]);   fmac_bcast<3>(acc[3], hreg[C], w[16 * C + 3]);
        fmac_bcast<4>(acc[0], hreg[C], w[16 * C + 4]);   fmac_bcast<5>(acc[1], hreg[C], w[16 * C + 5]);
        fmac_bcast<6>(acc[2], hreg[C], w[16 * C + 6]);   fmac_bcast<7>(acc[3], hreg[C], w[16 * C + 7]);
        fmac_bcast<8>(acc[0], hreg[C], w[16 * C + 8]);   fmac_bcast<9>(acc[1], hreg[C], w[16 * C + 9]);
        fmac_bcast<10>(acc[2], hreg[C], w[16 * C + 10]); fmac_bcast<11>(acc[3], hreg[C], w[16 * C + 11]);
        fmac_bcast<12>(acc[0], hreg[C], w[16 * C + 12]); fmac_bcast<13>(acc[1], hreg[C], w[16 * C + 13]);
        fmac_bcast<14>(acc[2], hreg[C], w[16 * C + 14]); fmac_bcast<15>(acc[3], hreg[C], w[16 * C + 15]);
        fma_rows<KSM, C + 1>(acc, hreg, w);
    }
}

// KSM = ceil(H / 16) groups of 16 hidden indices held in registers
template <bool SAVE, int KSM>
__global__ __launch_bounds__(512) void k_gru1(const Gru16Args a) {
    __shared__ float hs[2][16 * KSM];
    __shared__ float pre[512];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, Hp = a.Hp, H = a.H, HK = 16 * KSM;
    // FMA role: gate row R = tid of the 3H rows [r | z | n]
    const int R = tid;
    const bool rvalid = R < 3 * H;
    const int gate = rvalid ? R / H : 0, unit = rvalid ? R - gate * H : 0;
    float w[16 * KSM];
    {
        const float* __restrict__ wp = a.W16 + (long)wave * HK * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16 * KSM; ++k) w[k] = wp[k * 64];
    }
    const float bhn = a.bias[3 * Hp + unit];
    const int xoff = gate * Hp + unit;                 // input product of this row (gates r, z)
    // gate role: hidden unit u = tid
    const int u = tid;
    const bool gl = u < Hp, gv = u < H;
    const int uc = gv ? u : 0;

    for (long win = blockIdx.x; win < a.B; win += gridDim.x) {
        const float* __restrict__ xrow = a.XP + win * (long)T * (3L * Hp);
        for (int i = tid; i < 2 * HK; i += blockDim.x) (&hs[0][0])[i] = 0.f;      // h_0 = 0 (and the padding of both buffers)
        float hown = 0.f;
        float xa = rvalid && gate < 2 ? xrow[xoff] : 0.f;       // step 0
        float xn = gv ? xrow[2 * Hp + uc] : 0.f;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const float* __restrict__ hc = hs[t & 1];
            float* __restrict__ hn = hs[(t & 1) ^ 1];
            float hreg[KSM];
#pragma unroll
            for (int c = 0; c < KSM; ++c) hreg[c] = hc[16 * c + (lane & 15)];
            float acc[4] = {gate == 2 ? bhn : xa, 0.f, 0.f, 0.f};
            const int tn = t + 1 < T ? t + 1 : t;
            const float xa_n = xrow[(long)tn * (3L * Hp) + xoff];              // next step's values: in flight under the FMAs
            const float xn_n = xrow[(long)tn * (3L * Hp) + 2 * Hp + uc];
            fma_rows<KSM>(acc, hreg, w);
            pre[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
            if (gl) {
                float rg = 0.f, zg = 0.f, ng = 0.f, q = 0.f;
                if (gv) {
                    rg = gate_sigmoid(pre[u]);
                    zg = gate_sigmoid(pre[H + u]);
                    q = pre[2 * H + u];
                    ng = gate_tanh(xn + rg * q);
                    hown = __builtin_fmaf(zg, hown - ng, ng);
                }
                if (u < HK) hn[u] = hown;
                const long row = win * T + t;
                if (a.Seq) a.Seq[row * Hp + u] = hown;
                if (SAVE) {
                    float* __restrict__ gp = a.Gates + row * (4L * Hp) + u;
                    gp[0] = rg; gp[Hp] = zg; gp[2 * Hp] = ng; gp[3 * Hp] = q;
                }
            }
            xa = rvalid && gate < 2 ? xa_n : 0.f;
            xn = xn_n;
            __syncthreads();
        }
        if (a.Hend && u < a.ncol) a.Hend[win * a.ldhe + u] = gl ? hown : 0.f;
    }
}

// BPTT: 16-lane row rr of the workgroup = (gate block p = rr % 3, 16 outputs j of the hidden state); lane holds
// W_hh[p H + k][j] for all k.  dh_{t-1}[j] = dh_t[j] z_t[j] + sum_p sum_k W_hh[p H + k][j] da_p[k].
template <int KSM>
__global__ __launch_bounds__(512) void k_gru1_bwd(const Gru16BwdArgs a) {
    __shared__ float das[3][16 * KSM];
    __shared__ float part[3][16 * KSM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, Hp = a.Hp, H = a.H, HK = 16 * KSM;
    const int rr = tid >> 4, p = rr % 3, jb = rr / 3;
    const int j = 16 * jb + (tid & 15);
    const bool jvalid = jb < (H + 15) / 16;
    float w[16 * KSM];
    {
        const float* __restrict__ wp = a.W16T + (long)wave * HK * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16 * KSM; ++k) w[k] = wp[k * 64];
    }
    const int u = tid;
    const bool gl = u < Hp, gv = u < H;
    const int uc = gl ? u : 0;

    for (long win = blockIdx.x; win < a.B; win += gridDim.x) {
        float dhz = (a.DHend && gl) ? a.DHend[win * a.ldde + uc] : 0.f;         // carries dh into the step below
        for (int i = tid; i < 3 * HK; i += blockDim.x) { (&part[0][0])[i] = 0.f; (&das[0][0])[i] = 0.f; }
        // operands of step T-1
        long row = win * T + (T - 1);
        float r_ = 0.f, z_ = 0.f, n_ = 0.f, q_ = 0.f, hp_ = 0.f, ds_ = 0.f;
        auto fetch = [&](long rw, int t) {
            const float* __restrict__ gp = a.Gates + rw * (4L * Hp) + uc;
            r_ = gp[0]; z_ = gp[Hp]; n_ = gp[2 * Hp]; q_ = gp[3 * Hp];
            hp_ = a.Seq[(rw - (t > 0 ? 1 : 0)) * (long)Hp + uc];
            hp_ = t > 0 ? hp_ : 0.f;
            ds_ = a.DHseq ? a.DHseq[rw * a.lddh + uc] : 0.f;
        };
        fetch(row, T - 1);
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            row = win * T + t;
            if (gl) {
                float dh = dhz + (u < HK ? (part[0][u] + part[1][u]) + part[2][u] : 0.f) + ds_;
                dh = gv ? dh : 0.f;
                const float dan = dh * (1.f - z_) * (1.f - n_ * n_);
                const float daz = dh * (hp_ - n_) * z_ * (1.f - z_);
                const float dar = dan * q_ * r_ * (1.f - r_);
                const float dnh = dan * r_;
                dhz = dh * z_;
                float* __restrict__ op = a.DA + row * (4L * Hp) + u;
                op[0] = gv ? dan : 0.f; op[Hp] = gv ? dar : 0.f; op[2 * Hp] = gv ? daz : 0.f; op[3 * Hp] = gv ? dnh : 0.f;
                if (u < HK) { das[0][u] = gv ? dar : 0.f; das[1][u] = gv ? daz : 0.f; das[2][u] = gv ? dnh : 0.f; }
            }
            __syncthreads();
            if (t > 0) fetch(row - 1, t - 1);                 // next step's operands: in flight under the FMAs
            float hreg[KSM];
#pragma unroll
            for (int c = 0; c < KSM; ++c) hreg[c] = das[p][16 * c + (lane & 15)];
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            fma_rows<KSM>(acc, hreg, w);
            if (jvalid) part[p][j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            __syncthreads();
        }
    }
}

template <class F>
static int g1_dispatch(int H, F&& go) {
    const int ks = (H + 15) / 16;
    if (ks <= 3) go(std::integral_constant<int, 3>());
    else if (ks <= 6) go(std::integral_constant<int, 6>());
    else if (ks <= 8) go(std::integral_constant<int, 8>());
    else if (ks <= 10) go(std::integral_constant<int, 10>());
    else return -2;
    return 0;
}

// W16 / W16T of the args point at the k_gru1 packs here: [waves][16 * g1_ksm(H)][64]
int launch_gru1(const Gru16Args& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    const int nw = g1_waves(a.H, a.Hp, false);
    if (a.H < 1 || a.H > 160 || nw > 8) return -2;
    const long cap = cu_count();                      // one workgroup per CU (its registers hold W_hh), persistent over the batch
    const unsigned grid = (unsigned)(a.B < cap ? a.B : cap);
    int rc = g1_dispatch(a.H, [&](auto ksm) {
        constexpr int KSM = decltype(ksm)::value;
        if (a.Gates)
            hipLaunchKernelGGL((k_gru1<true, KSM>), dim3(grid), dim3(64 * nw), 0, s, a);
        else
            hipLaunchKernelGGL((k_gru1<false, KSM>), dim3(grid), dim3(64 * nw), 0, s, a);
    });
    if (rc) return rc;
    LAUNCH_CHECK();
    return 0;
}

int launch_gru1_bwd(const Gru16BwdArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    const int nw = g1_waves(a.H, a.Hp, true);
    if (a.H < 1 || a.H > 160 || nw > 8) return -2;
    const long cap = cu_count();
    const unsigned grid = (unsigned)(a.B < cap ? a.B : cap);
    int rc = g1_dispatch(a.H, [&](auto ksm) {
        constexpr int KSM = decltype(ksm)::value;
        hipLaunchKernelGGL((k_gru1_bwd<KSM>), dim3(grid), dim3(64 * nw), 0, s, a);
    });
    if (rc) return rc;
    LAUNCH_CHECK();
    return 0;
}

}  // namespace mtadgat
