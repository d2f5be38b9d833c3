// k_gru_cm: the large-batch recurrence (GRU layer / reconstruction decoder) in "chunk-major" form, split-fp16 operands
// (two fp16 pieces per fp32 value, three v_mfma_f32_32x32x16_f16 per product: mtadgat_device.h).
// Reference: GRULayer.forward modules.py:235-238, RNNDecoder modules.py:255-257, ReconstructionModel modules.py:276-283.
//
// What the tile-major k_gru (mtadgat_gru_impl.h) could not do, and why this kernel is laid out differently:
//   * it walked the hidden tiles one after the other, so every tile re-read x_t (5 x 672 B per window and step: 3.5 TB/s
//     of L2 misses) and re-split x_t and h_{t-1} into pieces (4.4 k of its 9.8 k VALU instructions per step);
//   * its new state needed a second home until the step ended (a 160 KB LDS mirror), its gate math ran with the matrix
//     pipe idle.
// Here a wave owns 32 windows and the accumulators of a whole GROUP of hidden tiles (r, z, n_x of up to three tiles:
// 9 x 16 registers; H = 150 is two groups, tiles {0,1,2} and {3,4}), so
//   * the input part runs chunk-major within a group: x_t is read once per group and step (the second read is L2
//     resident), split once per read (20 VALU per 16 features), and each piece feeds the 3 x NT gate tiles of the group;
//   * the recurrent part runs tile-major from the register-resident PIECES of h_{t-1} (split once, when h_t is born):
//     no VALU in the product loop at all;
//   * the gate math of tile c is issued between the MFMAs of tile c + 1 (a group's last tile: between those of the next
//     group's first input chunk) -- the f16 MFMA pipe runs beside the VALU (profiles/r02_mfma_valu_overlap.txt);
//   * the new pieces of tile c live in registers the tile's dead accumulators free: no LDS mirror of the state.
// With 32 windows per wave every weight word feeds ONE MFMA, i.e. a wave needs 1 KiB of weights per 32 cycles -- twice
// what the vector L1 delivers to four waves.  So the four waves of a workgroup share the weight stream through LDS:
// a ring of S granules (up to 6 NCG words of 1 KiB: the six [gate][piece] words of consecutive items), filled by LDS-DMA
// (global_load_lds_dwordx4, no staging registers; each wave issues a quarter of a granule), one workgroup barrier per
// granule.  x_t arrives the same way into a per-wave landing buffer (read back by the lane that asked for it).
// All DMA is inline asm and all waits on it are counted by hand (s_waitcnt vmcnt(N)): the main loop contains no
// compiler-visible global load, so hipcc inserts no vmcnt waits of its own (stores only make ours conservative).
//
#include "mtadgat_device.h"

namespace mtadgat {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const unsigned char* cbytes;

// the arguments the kernel needs (a slim copy of GruArgs: the wave-uniform state of the loop nest has to fit the SGPRs)
struct CmArgs {
    const float* X;
    long ldx;
    int Qx;
    const int* m0;
    cbytes Wx;           // XMODE 0: [chunk][tile][gate][piece] words; XMODE 1: [t][tile][gate][piece]
    cbytes Wh;           // [tile][whs chunks][gate][piece]
    int whs;
    const float* bias;
    const float* scale;
    const unsigned* vmax;
    int H, T;
    long B;
    float* Hend;
    long ldhe;
    float* Seq;
    long ldseq;
    const f32x4* Wfc;
    const float* bfc;
    float* Yfc;
    float* Ylast;
    int out_dim;
};

// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from (uniform base + per-lane 32-bit offset) to LDS byte address
// M0 + 16 * lane.  The instruction's immediate offset moves the global source AND the LDS destination
// (profiles/ubench_glds.hip), so a burst of up to four consecutive KiB needs one M0 / base set-up.
// M0 is compiler-reserved: it is written and restored inside the statement.
template <int CNT>
__device__ __forceinline__ void glds_burst(const void* sbase, unsigned voff, unsigned ldsdst) {
    unsigned keep;
    if constexpr (CNT == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
    else if constexpr (CNT == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
    else if constexpr (CNT == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
}
// 4 bytes per lane from a per-lane 64-bit address to ldsdst + 4 * lane
__device__ __forceinline__ void glds4_v(const void* vaddr, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vaddr), "s"(ldsdst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier without the vmcnt(0) drain of __syncthreads(): the DMA of later granules stays in flight across it.
// lgkmcnt(0): this wave's LDS reads of the slot that is recycled after the barrier have returned.
#ifdef MTADGAT_CM_NOBARRIER     // knock-out (results invalid: the waves race for the ring): what the granule barriers cost
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

#define CM_SB() __builtin_amdgcn_sched_barrier(0)

// Geometry shared with the launcher
template <int NCG, int XMODE>
struct CmGeom {
    static constexpr int NGRP = NCG > 3 ? 2 : 1;                 // tile groups per step
    static constexpr int NT0 = NCG > 3 ? (NCG + 1) / 2 : NCG;    // tiles of group 0 (the larger one)
    static constexpr int GMAX = 6 * NCG;                         // weight words (1 KiB) of the largest granule (NCG recurrent chunks)
    static constexpr int NPMAX = (GMAX + 3) / 4;                 // DMA pieces per wave of the largest granule
    static constexpr int SLOTB = 4 * NPMAX * 1024;               // bytes per ring slot
    static constexpr int S = 3;                                  // ring slots: one being consumed, two in flight
    static constexpr int DX = XMODE == 0 ? 4 : 2;                // x landing slots per wave (chunks in flight)
    static constexpr int XSLOT = XMODE == 0 ? 2048 : 1024;
    static constexpr int Hp = 32 * NCG;
    static constexpr int L_X = S * SLOTB;
    static constexpr int L_BIAS = L_X + 4 * DX * XSLOT;
    static constexpr int L_FC = L_BIAS + 4 * Hp * 4;
    static constexpr int L_BFC = L_FC + 4 * (4 * NCG) * 2 * 16;  // 4 floats
    static constexpr int L_M0 = L_BFC + 16;                      // T ints (XMODE 1), T <= 512
    static constexpr int L_END = L_M0 + (XMODE == 1 ? 512 * 4 : 0);
    static constexpr int c0(int grp) { return grp == 0 ? 0 : NT0; }
    static constexpr int nt(int grp) { return grp == 0 ? NT0 : NCG - NT0; }
    // accumulator sets (r, z, n_x of one tile each).  Tile j of a group owns set set_of(grp, j); a group walks its tiles in the
    // order xord(grp, .) through an input granule.  Two rules fix both maps: the set of group 0's LAST tile is not used by
    // group 1 (that tile's gate math runs all through group 1's input part), and the set of the last group's last tile is the
    // one group 0 touches last in its first granule (that tile's gate math has only the first items of the next step).
    static constexpr int NTM = NCG >= 3 ? 3 : 2;
    static constexpr int set_of(int grp, int j) { return (NCG == 4 && grp == 1) ? (j == 0 ? 2 : 0) : j; }
    static constexpr int xord(int grp, int k) {
        if (NCG == 5 && grp == 0) return k == 0 ? 0 : (k == 1 ? 2 : 1);
        if (NCG == 4 && grp == 0) return k == 0 ? 1 : 0;
        return k;
    }
    static constexpr int npx(int grp) { return (6 * nt(grp) + 3) / 4; }     // DMA pieces per wave: input granule of a group
    static constexpr int NPH = (6 * NCG + 3) / 4;                           // ... recurrent granule
};

// XMODE 0: input rows X[(win*T + t)*ldx + k] (fp32, 16-byte aligned rows, zero padded), weights [chunk][tile][gate][piece];
//          at least two 16-feature chunks (the launcher checks)
// XMODE 1: the reference's decoder input (modules.py:279): x_t[j] = hin[(t*Hin + j) / T], folded into one 8-wide chunk per
//          step (see mtadgat_gru_impl.h); weights [t][tile][gate][piece]
// FC: per-step Linear with out_dim <= 4 (ReconstructionModel.fc, modules.py:282) as dot products on the new state
//
// The order of a step: for each tile group  [nqx input granules (chunk q into the group's tiles)]  [2 recurrent granules per
// tile (NCG chunks each)].  While granule g is consumed, granule g + 2 is requested into the ring slot g - 1 left; every
// site knows at compile time (input granules: up to a two-way choice on q) which granule that is and how many DMA pieces a
// wave issues for it, so the boundary wait counts are immediates.
template <int NCG, int XMODE, bool FC>
__global__ __launch_bounds__(256, 1) void k_gru_cm(const CmArgs a) {
    static_assert(NCG >= 2 && NCG <= 5, "hidden sizes 33 .. 160");
    using GEO = CmGeom<NCG, XMODE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NGRP = GEO::NGRP, NT0 = GEO::NT0, SLOTB = GEO::SLOTB, DX = GEO::DX, XSLOT = GEO::XSLOT, Hp = GEO::Hp;
    constexpr int L_X = GEO::L_X, L_BIAS = GEO::L_BIAS, L_FC = GEO::L_FC, L_BFC = GEO::L_BFC, L_M0 = GEO::L_M0;
    constexpr int NTM = GEO::NTM;                                 // accumulator sets
    constexpr int QH = 2 * NCG;                                   // recurrent 16-feature chunks
    constexpr int XG = XMODE == 0 ? 2 : 4;                        // x DMA pieces per input granule and wave
    constexpr int NPH = GEO::NPH;
    constexpr int PPI = (8 + QH - 1) / QH;                        // gate value pairs per item of the following tile

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, g = lane >> 5;
    const long win = ((long)blockIdx.x * 4 + wv) * 32 + i;
    // lanes past the batch repeat its last window: same inputs, same results, same store addresses (benign duplicates) --
    // no lane predicate anywhere in the loop nest
    const long winc = win < a.B ? win : a.B - 1;
    const long wave0 = ((long)blockIdx.x * 4 + wv) * 32;
    const long wb = wave0 < a.B ? wave0 : a.B - 1;               // first window of the wave (clamped): uniform part of the x addresses
    const int T = a.T;
    const int nqx = XMODE == 0 ? (a.Qx + 1) / 2 : 1;            // 16-feature input chunks with non-zero weights
    // the two-piece input arithmetic needs |x| < 2^15 (fp16 pieces): when the producing convolution recorded more, the
    // tile-major kernel with its three-bf16-piece chunks serves the launch instead (it skips in the other case)
    if (XMODE == 0 && a.vmax != nullptr && !(__uint_as_float(*a.vmax) < 32768.f)) return;

    const float wInvS = a.scale[1];
    {
        const float wS = a.scale[0];
        float* lb = reinterpret_cast<float*>(lds + L_BIAS);
        for (int k = threadIdx.x; k < 4 * Hp; k += 256) lb[k] = a.bias[k] * wS;
        if (FC) {
            f32x4* lf = reinterpret_cast<f32x4*>(lds + L_FC);
            for (int k = threadIdx.x; k < a.out_dim * 8 * NCG; k += 256) {
                const int o = k / (8 * NCG), r = k - o * 8 * NCG;
                lf[k] = a.Wfc[(r >> 1) * 64 + o + 32 * (r & 1)];          // [o][8-chunk q][g]: W[o][8q + 4g + s]
            }
            if (threadIdx.x < 4) reinterpret_cast<float*>(lds + L_BFC)[threadIdx.x] = (int)threadIdx.x < a.out_dim ? a.bfc[threadIdx.x] : 0.f;
        }
        if (XMODE == 1)
            for (int k = threadIdx.x; k < T; k += 256) reinterpret_cast<int*>(lds + L_M0)[k] = a.m0[k];
    }
    __syncthreads();

    // ---- input addressing: uniform base + a constant 32-bit offset per lane (XMODE 0); per-lane rows (XMODE 1)
    const float* xwave = a.X + wb * T * a.ldx;
    const unsigned xvoff = (unsigned)(((winc - wb) * T * a.ldx + 4 * g) * 4);
    const float* xrow1 = a.X + winc * a.ldx;
    const int kmax = (int)a.ldx - 1;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- wave-uniform state
    int cslot = 0;                             // ring slot being consumed
    int xq = 0, xpass = 0, xt = 0, xps = 0, xcs = 0;   // x prefetch: chunk, pass (group), step, landing slot; landing slot read next
    cbytes pf_src = nullptr;                   // this wave's quarter of the granule being requested
    unsigned pf_dst = 0;

    // sources of the weight words of a granule
    auto src_x = [&](const int c0, const int q, const int t) -> cbytes {
        const long w0 = XMODE == 0 ? ((long)q * NCG + c0) * 6 : ((long)(t < T ? t : T - 1) * NCG + c0) * 6;
        return a.Wx + w0 * 1024;
    };
    auto src_h = [&](const int c, const int half) -> cbytes { return a.Wh + (long)((c * a.whs + half * NCG) * 6) * 1024; };
    // request set-up: the granule at `src` (NP pieces per wave) goes to the slot two granules ahead = the one left last
    auto pf_setup = [&](auto np_tag, cbytes src) {
        constexpr int NP = decltype(np_tag)::value;
        const int pslot = cslot == 0 ? 2 : cslot - 1;
        pf_src = src + wv * (NP * 1024);
        pf_dst = (unsigned)(pslot * SLOTB + wv * (NP * 1024));
    };
    // burst b of the request: pieces 4b .. min(4b + 3, NP - 1).  (A wave's last pieces may run past the granule's words:
    // they copy the words that follow in memory into the unused tail of the slot.)
    auto pf_burst = [&](auto np_tag, auto b_tag) {
        constexpr int NP = decltype(np_tag)::value, B = decltype(b_tag)::value;
        if constexpr (4 * B < NP) {
            constexpr int CNT = NP - 4 * B < 4 ? NP - 4 * B : 4;
            glds_burst<CNT>(pf_src + B * 4096, lane16, pf_dst + B * 4096);
        }
    };
    auto xglds = [&]() {                       // the DMA pieces of input chunk (xq, xt) into landing slot xps; cursor advanced
        const int tt = xt < T ? xt : T - 1;
        const unsigned dst = (unsigned)(L_X + (wv * DX + xps) * XSLOT);
        if constexpr (XMODE == 0) {
            const int c0 = 2 * xq, c1 = 2 * xq + 1 < a.Qx ? 2 * xq + 1 : a.Qx - 1;     // padded half chunk: its weights are zero
            const float* row = xwave + (long)tt * a.ldx;
            glds_burst<1>(row + 8 * c0, xvoff, dst);
            glds_burst<1>(row + 8 * c1, xvoff, dst + 1024u);
        } else {
            const int k0 = reinterpret_cast<const int*>(lds + L_M0)[tt] + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) glds4_v(xrow1 + (k0 + e < kmax ? k0 + e : kmax), dst + (unsigned)e * 256u);
        }
        const bool wrapq = xq + 1 == nqx;
        xq = wrapq ? 0 : xq + 1;
        const bool wrapp = wrapq && xpass + 1 == NGRP;
        xpass = wrapq ? (wrapp ? 0 : xpass + 1) : xpass;
        xt = wrapp ? xt + 1 : xt;
        xps = xps + 1 == DX ? 0 : xps + 1;
    };
    auto xread = [&](f32x4& xa, f32x4& xb) {   // landing slot xcs -> registers
        const unsigned char* p = lds + L_X + (wv * DX + xcs) * XSLOT;
        if constexpr (XMODE == 0) {
            xa = *reinterpret_cast<const f32x4*>(p + lane * 16);
            xb = *reinterpret_cast<const f32x4*>(p + 1024 + lane * 16);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[e] = *reinterpret_cast<const float*>(p + e * 256 + lane * 4);
            xb = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        xcs = xcs + 1 == DX ? 0 : xcs + 1;
    };
    // the six [gate][piece] words of item `item` of the granule in ring slot cslot
    auto aread = [&](f32x4 (&w)[6], const int item) {
        const unsigned char* p = lds + cslot * SLOTB + item * 6144 + lane * 16;
#pragma unroll
        for (int k = 0; k < 6; ++k) w[k] = *reinterpret_cast<const f32x4*>(p + k * 1024);
    };
    // granule boundary, executed at the start of a granule's LAST item (whose words are in registers already): the next
    // granule has landed for every wave, the slot just finished is free.  N: the DMA pieces this wave issued after the
    // awaited granule's = everything it issued during the granule that ends (its x pieces + the request of the granule
    // after the awaited one).
    auto boundary = [&](auto n_tag) {
        wait_vm<decltype(n_tag)::value>();
        ring_barrier();
        cslot = cslot == 2 ? 0 : cslot + 1;
    };
    auto init_acc = [&](f32x16& acc, const int row, const int c) {     // S * bias into an accumulator (LDS -> registers)
        const unsigned char* p = lds + L_BIAS + (row * Hp + 32 * c + 4 * g) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + m * 32);
            acc[4 * m + 0] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
        }
    };

    // ---- state
    f32x16 ar[NTM], az[NTM], anx[NTM], anh[2];
    f32x4 Ph[QH], Pl[QH], Pnh[QH], Pnl[QH];        // pieces of h_{t-1} per recurrent chunk (MFMA B operands); of h_t
#pragma unroll
    for (int q = 0; q < QH; ++q) {
        Ph[q] = f32x4{0.f, 0.f, 0.f, 0.f}; Pl[q] = Ph[q]; Pnh[q] = Ph[q]; Pnl[q] = Ph[q];
    }
    f32x4 xh, xl, xnh, xnl;                        // pieces of the current / next input chunk
    f32x4 xa, xb;                                  // raw halves of the next input chunk
    f32x4 wA[6], wB[6];                            // weight words of the current / next item (roles alternate)
    float yacc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 hv;                                      // four finished values of the state waiting for their store
    const float cs = -1.4426950408889634f * wInvS, ct2 = 2.8853900817779268f * wInvS;

    // ---- gate math of one value pair p = accumulator registers (2p, 2p + 1) of tile C (set K), two values per instruction
    // where the ISA has a packed form, cut into GS = 18 steps of one to five instructions.  Measured on this chip
    // (profiles/ubench_gate.hip, one wave per SIMD): a plain VALU instruction issues in 4.7 cycles, a transcendental in 8.7,
    // a packed one in 5.0, v_accvgpr_read in 5.7; an MFMA occupies the matrix pipe for 32 cycles and hides ~28 cycles of
    // whatever the wave issues behind it.  The first version of this kernel issued the 9 MFMAs of an item in triples and the
    // gate math of a pair in one piece: 56 % of its time was gate math beside an idle matrix pipe.  So every item is nine
    // MFMAs with a "gap" after each, and the 8 x 18 steps of a tile are dealt out evenly over ALL the gaps that host it.
    constexpr int GS = 18;
    // (plain one-value instructions on purpose: the packed fp32 forms -- v_pk_mul / add / fma_f32 -- do not run beside the
    // MFMAs; with them the gate math cost the same ~500 cycles per pair however finely it was interleaved)
    float g_a0, g_a1, g_z0, g_z1, g_r0, g_r1, g_u0, g_u1, g_x0, g_x1, g_v0, g_v1, g_n0, g_n1, g_h0, g_h1, g_d0, g_d1;      // the pair in flight
    unsigned g_nh = 0;
    auto gate_step = [&](auto c_tag, auto k_tag, auto p_tag, auto s_tag, const int tg) {
        constexpr int C = decltype(c_tag)::value, K = decltype(k_tag)::value, p = decltype(p_tag)::value, S = decltype(s_tag)::value;
        constexpr int r = 2 * p, qq = 2 * C + (p >> 2), d = p & 3, m = p >> 1, h0 = 2 * (p & 1);
        if constexpr (S == 0) { g_a0 = ar[K][r] * cs; g_a1 = ar[K][r + 1] * cs; }
        else if constexpr (S == 1) { g_z0 = az[K][r] * cs; g_z1 = az[K][r + 1] * cs; }
        else if constexpr (S == 2) { g_a0 = __builtin_amdgcn_exp2f(g_a0); g_a1 = __builtin_amdgcn_exp2f(g_a1); }
        else if constexpr (S == 3) { g_z0 = __builtin_amdgcn_exp2f(g_z0); g_z1 = __builtin_amdgcn_exp2f(g_z1); }
        else if constexpr (S == 4) { g_a0 += 1.0f; g_a1 += 1.0f; g_z0 += 1.0f; g_z1 += 1.0f; }
        else if constexpr (S == 5) { g_r0 = __builtin_amdgcn_rcpf(g_a0); g_r1 = __builtin_amdgcn_rcpf(g_a1); }
        else if constexpr (S == 6) { g_u0 = __builtin_amdgcn_rcpf(g_z0); g_u1 = __builtin_amdgcn_rcpf(g_z1); }
        else if constexpr (S == 7) { g_x0 = anx[K][r] * ct2; g_x1 = anx[K][r + 1] * ct2; }
        else if constexpr (S == 8) { const f32x16& AH = anh[C & 1]; g_v0 = AH[r] * ct2; g_v1 = AH[r + 1] * ct2; }
        else if constexpr (S == 9) { g_n0 = __builtin_fmaf(g_r0, g_v0, g_x0); g_n1 = __builtin_fmaf(g_r1, g_v1, g_x1); }
        else if constexpr (S == 10) { g_n0 = __builtin_amdgcn_exp2f(g_n0); g_n1 = __builtin_amdgcn_exp2f(g_n1); }
        else if constexpr (S == 11 || S == 13) {
            // (the element goes through a scalar first: __builtin_bit_cast applied directly to a vector-element expression
            // compiles to element 0 whatever the index -- hipcc 7.2)
            const u32x4 p4 = __builtin_bit_cast(u32x4, S == 11 ? Ph[qq] : Pl[qq]);
            const unsigned pd = p4[d];
            const f16x2 o2 = __builtin_bit_cast(f16x2, pd);
            if constexpr (S == 11) { g_n0 += 1.0f; g_n1 += 1.0f; g_h0 = (float)o2[0]; g_h1 = (float)o2[1]; }
            else { g_h0 += (float)o2[0]; g_h1 += (float)o2[1]; }
        }
        else if constexpr (S == 12) { g_n0 = __builtin_amdgcn_rcpf(g_n0); g_n1 = __builtin_amdgcn_rcpf(g_n1); }
        else if constexpr (S == 14) {
            g_n0 = __builtin_fmaf(g_n0, -2.0f, 1.0f); g_n1 = __builtin_fmaf(g_n1, -2.0f, 1.0f);
            hv[h0] = __builtin_fmaf(g_u0, g_h0 - g_n0, g_n0);           // (1 - z) n + z h
            hv[h0 + 1] = __builtin_fmaf(g_u1, g_h1 - g_n1, g_n1);
        } else if constexpr (S == 15) {
            g_nh = pack_f16(hv[h0], hv[h0 + 1]);
            const f16x2 hh = __builtin_bit_cast(f16x2, g_nh);
            g_d0 = hv[h0] - (float)hh[0]; g_d1 = hv[h0 + 1] - (float)hh[1];
        } else if constexpr (S == 16) {
            Pnh[qq][d] = __builtin_bit_cast(float, g_nh);
            Pnl[qq][d] = __builtin_bit_cast(float, pack_f16(g_d0, g_d1));
        } else if constexpr ((p & 1) == 1) {        // S == 17: values 4m .. 4m+3 = hidden units 32C + 8m + 4g + {0..3} are complete
            if (FC) {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (o < a.out_dim) {
                        const f32x4 wf = *reinterpret_cast<const f32x4*>(lds + L_FC + ((o * 4 * NCG + 4 * C + m) * 2 + g) * 16);
                        yacc[o] += wf[0] * hv[0] + wf[1] * hv[1] + wf[2] * hv[2] + wf[3] * hv[3];
                    }
            }
            if (a.Seq) *reinterpret_cast<f32x4*>(a.Seq + (winc * T + tg) * a.ldseq + 32 * C + 8 * m + 4 * g) = hv;
            if (a.Hend && tg == T - 1) *reinterpret_cast<f32x4*>(a.Hend + winc * a.ldhe + 32 * C + 8 * m + 4 * g) = hv;   // ldhe >= Hp
        }
    };
    // the share of gap GI of NGAP of the 8 GS steps of tile C (set K)
    auto gate_gap = [&](auto c_tag, auto k_tag, auto ngap_tag, auto gi_tag, const int tg) {
        constexpr int NGAP = decltype(ngap_tag)::value, GI = decltype(gi_tag)::value;
        constexpr int LO = GI * 8 * GS / NGAP, HI = (GI + 1) * 8 * GS / NGAP;
        static_for<LO, HI>([&](auto s_tag) {
            constexpr int st = decltype(s_tag)::value;
            gate_step(c_tag, k_tag, std::integral_constant<int, st / GS>{}, std::integral_constant<int, st % GS>{}, tg);
        });
    };
    // the per-step Linear's result once every tile of step tg has contributed
    auto fc_finish = [&](const int tg) {
        if (FC) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (o < a.out_dim) {
                    const float y = yacc[o] + __shfl_xor(yacc[o], 32) + reinterpret_cast<const float*>(lds + L_BFC)[o];
                    yacc[o] = 0.f;
                    if (a.Yfc) a.Yfc[(winc * T + tg) * (long)a.out_dim + o] = y;        // both half-wave lanes of a window hold y
                    if (a.Ylast && tg == T - 1) a.Ylast[winc * (long)a.out_dim + o] = y;
                }
        }
    };
    // one item: nine MFMAs = (weight hi x operand lo, weight lo x operand hi, weight hi x operand hi) x three gates, smallest
    // terms first; consecutive MFMAs never share an accumulator.  gap(G) runs in the shadow of MFMA number G
    auto item9 = [&](const f32x4 (&w)[6], const f32x4 bh, const f32x4 bl, f32x16& g0, f32x16& g1, f32x16& g2, auto&& gap) {
        g0 = mfma_h(w[0], bl, g0); CM_SB(); gap(std::integral_constant<int, 0>{}); CM_SB();
        g1 = mfma_h(w[2], bl, g1); CM_SB(); gap(std::integral_constant<int, 1>{}); CM_SB();
        g2 = mfma_h(w[4], bl, g2); CM_SB(); gap(std::integral_constant<int, 2>{}); CM_SB();
        g0 = mfma_h(w[1], bh, g0); CM_SB(); gap(std::integral_constant<int, 3>{}); CM_SB();
        g1 = mfma_h(w[3], bh, g1); CM_SB(); gap(std::integral_constant<int, 4>{}); CM_SB();
        g2 = mfma_h(w[5], bh, g2); CM_SB(); gap(std::integral_constant<int, 5>{}); CM_SB();
        g0 = mfma_h(w[0], bh, g0); CM_SB(); gap(std::integral_constant<int, 6>{}); CM_SB();
        g1 = mfma_h(w[2], bh, g1); CM_SB(); gap(std::integral_constant<int, 7>{}); CM_SB();
        g2 = mfma_h(w[4], bh, g2); CM_SB(); gap(std::integral_constant<int, 8>{}); CM_SB();
    };
    // word k of the six [gate][piece] words of item `item` of the granule in ring slot cslot
    auto aread1 = [&](f32x4 (&w)[6], const int item, const int k) {
        w[k] = *reinterpret_cast<const f32x4*>(lds + cslot * SLOTB + item * 6144 + lane * 16 + k * 1024);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: first x chunks and the first two granules on their way, accumulators of tile 0 primed
    {
#pragma unroll
        for (int k = 0; k < DX; ++k) xglds();
        constexpr int NP0 = GEO::npx(0);
        cslot = 1;                              // (pf_setup targets the slot before cslot)
        pf_setup(std::integral_constant<int, NP0>{}, src_x(0, 0, 0));
        pf_burst(std::integral_constant<int, NP0>{}, I0{}); pf_burst(std::integral_constant<int, NP0>{}, I1{});
        cslot = 2;
        if constexpr (XMODE == 0) {
            pf_setup(std::integral_constant<int, NP0>{}, src_x(0, 1, 0));
            pf_burst(std::integral_constant<int, NP0>{}, I0{}); pf_burst(std::integral_constant<int, NP0>{}, I1{});
            wait_vm<NP0>();                     // everything but the second granule has landed
        } else {
            pf_setup(std::integral_constant<int, NPH>{}, src_h(0, 0));
            pf_burst(std::integral_constant<int, NPH>{}, I0{}); pf_burst(std::integral_constant<int, NPH>{}, I1{});
            wait_vm<NPH>();
        }
        ring_barrier();
        cslot = 0;
        xread(xa, xb);
    }
    split2h(xa, xb, xh, xl);
    aread(wA, GEO::xord(0, 0));
    {
        constexpr int J0 = GEO::xord(0, 0), K0i = GEO::set_of(0, J0);
        init_acc(ar[K0i], 0, J0); init_acc(az[K0i], 1, J0); init_acc(anx[K0i], 2, J0);
    }

    // ---- which gate math rides where.  Within a group the gates of tile j - 1 ride in the items of tile j's recurrent part.
    // A group's LAST tile has no such successor:
    //   * group 0's (two groups) rides through the first NXD input granules of group 1, which does not use its accumulator set;
    //   * the last group's is "squeezed" into the first items of the next step's first granule -- those before the item that
    //     re-uses its set (group 0 walks its tiles in an order that puts that item last).
    constexpr int LG = NGRP - 1;
    constexpr int CSQ = GEO::c0(LG) + GEO::nt(LG) - 1, KSQ = GEO::set_of(LG, GEO::nt(LG) - 1);      // squeezed tile (= NCG - 1), its set
    constexpr int CD1 = NT0 - 1, KD1 = GEO::set_of(0, NT0 - 1);                                   // group 0's last tile, its set
    constexpr int NXD = XMODE == 0 ? 4 : 1;
    static_assert(NGRP == 1 || GEO::set_of(1, 0) != KD1, "group 1 must not use the set of group 0's last tile");
    static_assert(NGRP == 1 || GEO::nt(1) < 2 || GEO::set_of(1, 1) != KD1, "group 1 must not use the set of group 0's last tile");
    static_assert(GEO::set_of(0, GEO::xord(0, NT0 - 1)) == KSQ, "group 0 touches the squeezed tile's set last");
    using CSQt = std::integral_constant<int, CSQ>;
    using KSQt = std::integral_constant<int, KSQ>;
    using CD1t = std::integral_constant<int, CD1>;
    using KD1t = std::integral_constant<int, KD1>;

    // one input granule: chunk q of step t into the NT tiles of group GRP, one item per tile in the group's order.
    // QI: the granule's number when it is one of the first (compile-time: it hosts gate math), -1 otherwise
    auto x_granule = [&](auto grp_tag, auto qi_tag, const int q, const int t) {
        constexpr int GRP = decltype(grp_tag)::value, QI = decltype(qi_tag)::value;
        constexpr bool FIRST = QI == 0;
        constexpr int C0 = GEO::c0(GRP), NT = GEO::nt(GRP), NPX = GEO::npx(GRP);
        constexpr bool HOST_SQ = GRP == 0 && QI == 0;
        constexpr bool HOST_D1 = NGRP == 2 && GRP == 1 && QI >= 0 && QI < NXD;
        const bool have_sq = t > 0;
        const bool lastq = q + 1 == nqx;
        // the granule requested meanwhile (two ahead): this group's chunk q + 2, or the first tile's recurrent granule q + 2 - nqx
        const bool tgt_h = XMODE == 1 || q + 2 >= nqx;
        using NPXt = std::integral_constant<int, NPX>;
        using NPHt = std::integral_constant<int, NPH>;
        static_for<0, NT>([&](auto k_tag) {
            constexpr int k = decltype(k_tag)::value;
            constexpr int j = GEO::xord(GRP, k), K = GEO::set_of(GRP, j);
            constexpr bool LASTI = k == NT - 1;
            f32x4 (&wc)[6] = (k & 1) ? wB : wA;
            f32x4 (&wn)[6] = (k & 1) ? wA : wB;
            // the item after this one: the next tile of the order; after the last: the same group's next chunk, or chunk 0 of
            // the first tile's recurrent part
            const int nitem = LASTI ? (lastq ? 0 : GEO::xord(GRP, 0)) : GEO::xord(GRP, LASTI ? 0 : k + 1);
            // what runs in the shadow of MFMA number G of the item
            auto gap = [&](auto g_tag) {
                constexpr int G = decltype(g_tag)::value;
                if constexpr (LASTI && G == 0) {
                    if (tgt_h) boundary(std::integral_constant<int, XG + NPH>{});
                    else boundary(std::integral_constant<int, XG + NPX>{});
                }
                if constexpr (G < 6) aread1(wn, nitem, G);        // (after boundary(): of the granule that begins)
                if constexpr (k == 0) {
                    // requests: the x chunk DX chunks ahead, then the weight granule two granules ahead in up to two bursts
                    if constexpr (G == 1) xglds();
                    if constexpr (G == 3) {
                        if (tgt_h) pf_setup(NPHt{}, src_h(C0, XMODE == 1 ? 1 : q + 2 - nqx));
                        else pf_setup(NPXt{}, src_x(C0, q + 2, t));
                    }
                    if constexpr (G == 4) { if (tgt_h) pf_burst(NPHt{}, I0{}); else pf_burst(NPXt{}, I0{}); }
                    if constexpr (G == 6) { if (tgt_h) pf_burst(NPHt{}, I1{}); else pf_burst(NPXt{}, I1{}); }
                }
                if constexpr (LASTI) {
                    // the next chunk's raw halves (landed: it was requested DX chunks ago) and their split
                    if constexpr (G == 6) xread(xa, xb);
                    if constexpr (G == 8) split2h(xa, xb, xnh, xnl);
                }
                if constexpr (HOST_SQ && !LASTI) {
                    if (have_sq) gate_gap(CSQt{}, KSQt{}, std::integral_constant<int, (NT - 1) * 9>{}, std::integral_constant<int, k * 9 + G>{}, t - 1);
                }
                if constexpr (HOST_D1)
                    gate_gap(CD1t{}, KD1t{}, std::integral_constant<int, NXD * NT * 9>{}, std::integral_constant<int, (QI * NT + k) * 9 + G>{}, t);
            };
            item9(wc, xh, xl, ar[K], az[K], anx[K], gap);
            if constexpr (FIRST) {
                // accumulators of the next tile of the order (the last one's set only now: the squeezed gates are done)
                if constexpr (k + 1 < NT) {
                    if constexpr (HOST_SQ && k + 1 == NT - 1) {
                        if (have_sq) fc_finish(t - 1);
                    }
                    constexpr int jn = GEO::xord(GRP, k + 1), Kn = GEO::set_of(GRP, jn);
                    init_acc(ar[Kn], 0, C0 + jn); init_acc(az[Kn], 1, C0 + jn); init_acc(anx[Kn], 2, C0 + jn);
                }
                if constexpr (LASTI) {
                    if constexpr (HOST_SQ) {
                        if (have_sq) {          // the squeezed tile's two chunks of the state: the previous step is complete
                            Ph[2 * CSQ] = Pnh[2 * CSQ]; Pl[2 * CSQ] = Pnl[2 * CSQ];
                            Ph[2 * CSQ + 1] = Pnh[2 * CSQ + 1]; Pl[2 * CSQ + 1] = Pnl[2 * CSQ + 1];
                        }
                    }
                    init_acc(anh[C0 & 1], 3, C0);           // free here (group 0: the squeezed gates are done; group 1: the hosted tile uses the other)
                }
                CM_SB();
            }
        });
        xh = xnh; xl = xnl;
        if constexpr (NT & 1) {                     // odd item count: bring the words of the next item back into wA
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) wA[kk] = wB[kk];
        }
    };

    for (int t = 0; t < T; ++t) {
        static_for<0, NGRP>([&](auto grp_tag) {
            constexpr int GRP = decltype(grp_tag)::value;
            constexpr int C0 = GEO::c0(GRP), NT = GEO::nt(GRP);
            constexpr int G2 = (GRP + 1) % NGRP;                 // the group after this one (of the next step after the last)
            constexpr int NPEEL = (NGRP == 2 && GRP == 1) ? NXD : 1;      // granules with a compile-time number
            const int t2 = GRP == NGRP - 1 ? t + 1 : t;
            // ---------------- input part of the group, chunk-major
            static_for<0, NPEEL>([&](auto qi_tag) {
                constexpr int QI = decltype(qi_tag)::value;
                if (QI < nqx) x_granule(grp_tag, qi_tag, QI, t);
            });
            for (int q = NPEEL; q < nqx; ++q) x_granule(grp_tag, std::integral_constant<int, -1>{}, q, t);
            if constexpr (NGRP == 2 && GRP == 1) {
                // fewer input chunks than hosting granules: the rest of the hosted gate math, unhidden
                static_for<0, NXD>([&](auto qi_tag) {
                    constexpr int QI = decltype(qi_tag)::value;
                    if (QI >= nqx)
                        static_for<0, NT * 9>([&](auto u_tag) {
                            gate_gap(CD1t{}, KD1t{}, std::integral_constant<int, NXD * NT * 9>{},
                                     std::integral_constant<int, QI * NT * 9 + decltype(u_tag)::value>{}, t);
                        });
                });
            }
            // ---------------- recurrent part, tile-major; the gates of tile c - 1 between the MFMAs of tile c
            static_for<0, NT>([&](auto j_tag) {
                constexpr int jt = decltype(j_tag)::value;
                constexpr int c = C0 + jt, K = GEO::set_of(GRP, jt);
                static_for<0, QH>([&](auto q_tag) {
                    constexpr int qh = decltype(q_tag)::value;
                    constexpr int gi = qh % NCG;                 // item within its granule
                    constexpr bool LASTI = gi == NCG - 1;
                    constexpr int it = jt * QH + qh;             // item number within the group's recurrent part (even count per tile)
                    // the granule requested during this one: two ahead in the step's order
                    constexpr int hg = 2 * jt + qh / NCG, k2 = hg + 2;
                    constexpr bool T_H = k2 < 2 * NT;                                   // a later recurrent granule of this group
                    constexpr bool T_X = !T_H && (XMODE == 0 || k2 - 2 * NT == 0);      // an input granule of the next group
                    constexpr int NPT = T_X ? GEO::npx(G2) : NPH;                       // (else: the next group's first recurrent granule)
                    using NPTt = std::integral_constant<int, NPT>;
                    // the item after this one; after the group's last: the next group's first tile of its order
                    constexpr int nitem = !LASTI ? gi + 1 : ((jt == NT - 1 && qh == QH - 1) ? GEO::xord(G2, 0) : 0);
                    f32x4 (&wc)[6] = (it & 1) ? wB : wA;
                    f32x4 (&wn)[6] = (it & 1) ? wA : wB;
                    auto gap = [&](auto g_tag) {
                        constexpr int G = decltype(g_tag)::value;
                        if constexpr (LASTI && G == 0) boundary(NPTt{});
                        if constexpr (G < 6) aread1(wn, nitem, G);
                        if constexpr (gi == 0) {
                            if constexpr (G == 6) {
                                if constexpr (T_H) pf_setup(NPTt{}, src_h(C0 + k2 / 2, k2 % 2));
                                else if constexpr (T_X) pf_setup(NPTt{}, src_x(GEO::c0(G2), k2 - 2 * NT, t2));
                                else pf_setup(NPTt{}, src_h(GEO::c0(G2), 0));
                            }
                            if constexpr (G == 7) pf_burst(NPTt{}, I0{});
                            if constexpr (G == 8) pf_burst(NPTt{}, I1{});
                        }
                        if constexpr (jt > 0)
                            gate_gap(std::integral_constant<int, c - 1>{}, std::integral_constant<int, GEO::set_of(GRP, jt - 1)>{},
                                     std::integral_constant<int, QH * 9>{}, std::integral_constant<int, qh * 9 + G>{}, t);
                    };
                    item9(wc, Ph[qh], Pl[qh], ar[K], az[K], anh[c & 1], gap);
                    if constexpr (qh == QH - 1) {
                        // the other n_h accumulator is free again (tile c - 1's gates are done): prime it for tile c + 1;
                        // after the group's last tile: the accumulator set of the next group's first tile
                        if constexpr (jt + 1 < NT) {
                            init_acc(anh[(c + 1) & 1], 3, c + 1);
                        } else {
                            constexpr int J2 = GEO::xord(G2, 0), K2 = GEO::set_of(G2, J2), C2 = GEO::c0(G2) + J2;
                            init_acc(ar[K2], 0, C2); init_acc(az[K2], 1, C2); init_acc(anx[K2], 2, C2);
                        }
                        CM_SB();
                    }
                });
            });
        });
        // every tile but the squeezed one has its new pieces: they replace the old ones (nothing reads those any more)
#pragma unroll
        for (int kk = 0; kk < QH - 2; ++kk) { Ph[kk] = Pnh[kk]; Pl[kk] = Pnl[kk]; }
    }
    // ---------------- the last step's last tile
    static_for<0, 8 * GS>([&](auto s_tag) {
        constexpr int st = decltype(s_tag)::value;
        gate_step(CSQt{}, KSQt{}, std::integral_constant<int, st / GS>{}, std::integral_constant<int, st % GS>{}, T - 1);
    });
    fc_finish(T - 1);
    wait_vm<0>();
}

// ---- re-ordering of the two-piece input pack: [tile][chunk][gate][piece] words -> [chunk][tile][gate][piece]
__global__ void k_reorder_xq(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int ncg, int Qd) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)ncg * Qd * 6 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int w = (int)(r % 6);
    const long r2 = r / 6;
    const int c = (int)(r2 % ncg);
    const int q = (int)(r2 / ncg);
    dst[idx] = src[(((long)c * Qd + q) * 6 + w) * 64 + lane];
}

template <int NCG, int XMODE, bool FC>
int launch_cm_one(const CmArgs& a, hipStream_t s) {
    using GEO = CmGeom<NCG, XMODE>;
    const size_t lds = (size_t)GEO::L_END;
    {   // (per device, and cheap: set on every launch -- a process-wide "done" flag would leave a second GPU of the process without it)
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_cm<NCG, XMODE, FC>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
    }
    const unsigned grid = (unsigned)((a.B + 127) / 128);
    hipLaunchKernelGGL((k_gru_cm<NCG, XMODE, FC>), dim3(grid), dim3(256), lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

template <int NCG>
int launch_cm_ncg(const CmArgs& a, int xmode, bool fc, hipStream_t s) {
    if (xmode == 0) return launch_cm_one<NCG, 0, false>(a, s);
    return fc ? launch_cm_one<NCG, 1, true>(a, s) : launch_cm_one<NCG, 1, false>(a, s);
}

}  // namespace

int launch_reorder_xq(const float* src, float* dst, int ncg, int Qd, hipStream_t s) {
    const long total = (long)ncg * Qd * 6 * 64;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(k_reorder_xq, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), ncg, Qd);
    LAUNCH_CHECK();
    return 0;
}

bool gru_cm_supported(int ncg, int xmode, bool fc, int out_dim) {
    if (ncg < 2 || ncg > 5) return false;
    if (xmode != 0 && xmode != 1) return false;
    if (fc && (xmode != 1 || out_dim > 4)) return false;
    return true;
}

int launch_gru_cm(const GruArgs& g, int ncg, int xmode, bool fc, hipStream_t s) {
    if (g.B <= 0) return 0;
    if (!gru_cm_supported(ncg, xmode, fc, g.out_dim)) return -2;
    if (xmode == 0 && ((g.ldx & 3) != 0 || g.Qx < 3)) return -2;        // two 16-feature input chunks at least
    if (g.Hp != 32 * ncg || g.Wxq == nullptr || g.T > 512) return -2;
    if (g.Hend != nullptr && (g.ldhe < g.Hp || (g.ldhe & 3) != 0)) return -2;
    if (g.Seq != nullptr && (g.ldseq & 3) != 0) return -2;
    CmArgs a{};
    a.X = g.X; a.ldx = g.ldx; a.Qx = g.Qx; a.m0 = g.m0;
    a.Wx = reinterpret_cast<cbytes>(g.Wxq); a.Wh = reinterpret_cast<cbytes>(g.Wh); a.whs = g.whs;
    a.bias = g.bias; a.scale = g.scale; a.vmax = g.vmax; a.H = g.H; a.T = g.T; a.B = g.B;
    a.Hend = g.Hend; a.ldhe = g.ldhe; a.Seq = g.Seq; a.ldseq = g.ldseq;
    a.Wfc = g.Wfc; a.bfc = g.bfc; a.Yfc = g.Yfc; a.Ylast = g.Ylast; a.out_dim = g.out_dim;
    switch (ncg) {
#ifdef MTADGAT_CM_ONLY                      // developer builds: one hidden-size class only (compile time)
        case MTADGAT_CM_ONLY: return launch_cm_ncg<MTADGAT_CM_ONLY>(a, xmode, fc, s);
#else
        case 2: return launch_cm_ncg<2>(a, xmode, fc, s);
        case 3: return launch_cm_ncg<3>(a, xmode, fc, s);
        case 4: return launch_cm_ncg<4>(a, xmode, fc, s);
        case 5: return launch_cm_ncg<5>(a, xmode, fc, s);
#endif
        default: return -2;
    }
}

}  // namespace mtadgat
