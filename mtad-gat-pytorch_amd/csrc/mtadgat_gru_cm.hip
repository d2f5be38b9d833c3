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
// WMODE 0 is the same arithmetic with every operand loaded straight from memory by ordinary loads (no LDS ring, no
// barriers, no DMA): the bring-up / cross-check build, bounded by the vector L1 (64 B/clk per CU).
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

// LDS-DMA: 16 bytes per lane from (uniform base + per-lane 32-bit offset) to LDS byte address ldsdst + 16 * lane.
// M0 carries the LDS address; it is compiler-reserved, so it is written and restored inside the statement.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(ldsdst) : "memory");
}
// ... from a per-lane 64-bit address
__device__ __forceinline__ void glds16_v(const void* vaddr, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vaddr), "s"(ldsdst) : "memory");
}
// 4 bytes per lane to ldsdst + 4 * lane
__device__ __forceinline__ void glds4_v(const void* vaddr, unsigned ldsdst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vaddr), "s"(ldsdst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n DMA pieces still in flight (n is wave-uniform, 0 <= n <= 15)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
        case 0: wait_vm<0>(); break;   case 1: wait_vm<1>(); break;   case 2: wait_vm<2>(); break;   case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;   case 5: wait_vm<5>(); break;   case 6: wait_vm<6>(); break;   case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break;   case 9: wait_vm<9>(); break;   case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
    }
}
// workgroup barrier without the vmcnt(0) drain of __syncthreads(): the DMA of later granules stays in flight across it.
// lgkmcnt(0): this wave's LDS reads of the slot that is recycled after the barrier have returned.
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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
};

// XMODE 0: input rows X[(win*T + t)*ldx + k] (fp32, 16-byte aligned rows, zero padded), weights [chunk][tile][gate][piece]
// XMODE 1: the reference's decoder input (modules.py:279): x_t[j] = hin[(t*Hin + j) / T], folded into one 8-wide chunk per
//          step (see mtadgat_gru_impl.h); weights [t][tile][gate][piece]
// FC: per-step Linear with out_dim <= 4 (ReconstructionModel.fc, modules.py:282) as dot products on the new state
template <int NCG, int XMODE, bool FC, int WMODE>
__global__ __launch_bounds__(256, 1) void k_gru_cm(const CmArgs a) {
    static_assert(NCG >= 2 && NCG <= 5, "hidden sizes 33 .. 160");
    using GEO = CmGeom<NCG, XMODE>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NGRP = GEO::NGRP, NT0 = GEO::NT0, SLOTB = GEO::SLOTB, S = GEO::S, DX = GEO::DX, XSLOT = GEO::XSLOT, Hp = GEO::Hp;
    constexpr int L_X = GEO::L_X, L_BIAS = GEO::L_BIAS, L_FC = GEO::L_FC, L_BFC = GEO::L_BFC, L_M0 = GEO::L_M0;
    constexpr int NTM = NT0;                                      // accumulator sets (tiles of the larger group)
    constexpr int QH = 2 * NCG;                                   // recurrent 16-feature chunks
    constexpr int XG = WMODE == 1 ? (XMODE == 0 ? 2 : 4) : 0;     // x DMA pieces per input granule and wave
    constexpr int NPH = (6 * NCG + 3) / 4;                        // DMA pieces per wave of a recurrent granule
    constexpr int PPI = (8 + QH - 1) / QH;                        // gate value pairs per item of the following tile

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, g = lane >> 5;
    const long win = ((long)blockIdx.x * 4 + wv) * 32 + i;
    const long winc = win < a.B ? win : a.B - 1;
    // lanes past the batch repeat its last window: same inputs, same results, same store addresses (benign duplicates) --
    // no lane predicate anywhere in the loop nest
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

    // ---- per-lane input addressing
    const float* xrow = XMODE == 0 ? a.X + winc * T * a.ldx + 4 * g : a.X + winc * a.ldx;
    const int kmax = (int)a.ldx - 1;

    // ---- the granule sequence of a step: for each tile group [nqx input granules][2 recurrent granules per tile]
    // granule -> (source of its weight words, number of words)
    auto gran_desc = [&](int grp, int k, int t, cbytes& src, int& words) {
        const int C0 = grp == 0 ? 0 : NT0, NT = grp == 0 ? NT0 : NCG - NT0;
        if (k < nqx) {
            const long w0 = XMODE == 0 ? ((long)k * NCG + C0) * 6 : ((long)(t < T ? t : T - 1) * NCG + C0) * 6;
            src = a.Wx + w0 * 1024;
            words = 6 * NT;
        } else {
            const int kk = k - nqx, c = C0 + (kk >> 1);
            src = a.Wh + (long)((c * a.whs + (kk & 1) * NCG) * 6) * 1024;
            words = 6 * NCG;
        }
    };
    auto gran_count = [&](int grp) { return nqx + 2 * (grp == 0 ? NT0 : NCG - NT0); };

    // ---- cursors (wave-uniform)
    int pgrp = 0, pk = 0, pt = 0, pslot = 0;   // weight prefetch: group, granule within the group, step, ring slot
    int pf_np = 0;                             // DMA pieces per wave of the granule being prefetched
    int pf_words = 0;
    cbytes pf_src = nullptr;
    unsigned pf_dst = 0;
    int cslot = 0;                             // ring slot being consumed
    int xq = 0, xpass = 0, xt = 0, xps = 0, xcs = 0;   // x prefetch: chunk, pass (group), step, landing slot; landing slot read next
    int cgrp = 0, ck = 0, ct = 0;              // WMODE 0: the granule being consumed and the base of its words
    cbytes c_src = nullptr;

    auto pf_begin = [&]() {                    // next granule to fetch -> (pf_src, pf_dst, pf_np), cursor advanced
        gran_desc(pgrp, pk, pt, pf_src, pf_words);
        pf_np = (pf_words + 3) >> 2;
        pf_dst = (unsigned)(pslot * SLOTB);
        const bool wrapk = pk + 1 == gran_count(pgrp);
        pk = wrapk ? 0 : pk + 1;
        const bool wrapg = wrapk && pgrp + 1 == NGRP;
        pgrp = wrapk ? (wrapg ? 0 : pgrp + 1) : pgrp;
        pt = wrapg ? pt + 1 : pt;
        pslot = pslot + 1 == S ? 0 : pslot + 1;
    };
    auto wglds = [&](const int n) {            // piece n of this wave's quarter of the granule (if it has that many)
        if constexpr (WMODE == 1) {
            if (n < pf_np) {
                const int j = wv * pf_np + n;
                const int js = j < pf_words ? j : pf_words - 1;      // the quarter's padding re-reads the last word
                glds16_s(pf_src + (size_t)js * 1024, (unsigned)lane * 16u, pf_dst + (unsigned)j * 1024u);
            }
        }
    };
    auto xglds = [&](const int e) {            // DMA piece e of input chunk (xq, xt) into landing slot xps
        if constexpr (WMODE == 1) {
            const int tt = xt < T ? xt : T - 1;
            const unsigned dst = (unsigned)(L_X + (wv * DX + xps) * XSLOT);
            if constexpr (XMODE == 0) {
                const int c8 = 2 * xq + e < a.Qx ? 2 * xq + e : a.Qx - 1;     // padded half chunk: its weights are zero
                glds16_v(xrow + (long)tt * a.ldx + 8 * c8, dst + (unsigned)e * 1024u);
            } else {
                const int k0 = reinterpret_cast<const int*>(lds + L_M0)[tt] + 4 * g + e;
                glds4_v(xrow + (k0 < kmax ? k0 : kmax), dst + (unsigned)e * 256u);
            }
        }
    };
    auto x_advance = [&]() {
        const bool wrapq = xq + 1 == nqx;
        xq = wrapq ? 0 : xq + 1;
        const bool wrapp = wrapq && xpass + 1 == NGRP;
        xpass = wrapq ? (wrapp ? 0 : xpass + 1) : xpass;
        xt = wrapp ? xt + 1 : xt;
        xps = xps + 1 == DX ? 0 : xps + 1;
    };
    // WMODE 0: an x chunk straight from memory (step tq, chunk q)
    auto xload_direct = [&](int q, int tq, f32x4& xa, f32x4& xb) {
        const int tt = tq < T ? tq : T - 1;
        if constexpr (XMODE == 0) {
            const int c0 = 2 * q < a.Qx ? 2 * q : a.Qx - 1, c1 = 2 * q + 1 < a.Qx ? 2 * q + 1 : a.Qx - 1;
            xa = *reinterpret_cast<const f32x4*>(xrow + (long)tt * a.ldx + 8 * c0);
            xb = *reinterpret_cast<const f32x4*>(xrow + (long)tt * a.ldx + 8 * c1);
        } else {
            const int k0 = a.m0[tt] + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[e] = xrow[k0 + e < kmax ? k0 + e : kmax];
            xb = f32x4{0.f, 0.f, 0.f, 0.f};
        }
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
    // the six [gate][piece] words of item `item` of the granule in ring slot cslot (WMODE 0: at c_src)
    auto aread = [&](f32x4 (&w)[6], const int item) {
        if constexpr (WMODE == 1) {
            const unsigned char* p = lds + cslot * SLOTB + item * 6144 + lane * 16;
#pragma unroll
            for (int k = 0; k < 6; ++k) w[k] = *reinterpret_cast<const f32x4*>(p + k * 1024);
        } else {
            const f32x4* p = reinterpret_cast<const f32x4*>(c_src) + item * 6 * 64 + lane;
#pragma unroll
            for (int k = 0; k < 6; ++k) w[k] = p[k * 64];
        }
    };
    // granule boundary, executed at the start of a granule's LAST item (whose words are in registers already): the next
    // granule has landed for every wave, the slot just finished is free.  xc: x pieces issued during the granule that ends.
    // At this point the DMA issued after the awaited granule's is: this granule's own x pieces and the weight pieces of
    // the granule after the awaited one (pf_np of them).
    auto boundary = [&](const int xc) {
        if constexpr (WMODE == 1) {
            wait_vm_dyn(xc + pf_np);
            ring_barrier();
        }
        cslot = cslot + 1 == S ? 0 : cslot + 1;
        if constexpr (WMODE == 0) {
            const bool wrapk = ck + 1 == gran_count(cgrp);
            ck = wrapk ? 0 : ck + 1;
            const bool wrapg = wrapk && cgrp + 1 == NGRP;
            cgrp = wrapk ? (wrapg ? 0 : cgrp + 1) : cgrp;
            ct = wrapg ? ct + 1 : ct;
            int words;
            gran_desc(cgrp, ck, ct, c_src, words);
        }
    };
    auto init_acc = [&](f32x16& acc, const int row, const int c) {     // S * bias into an accumulator (LDS -> registers)
        const unsigned char* p = lds + L_BIAS + (row * Hp + 32 * c + 4 * g) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p + m * 32);
            acc[4 * m + 0] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
        }
    };

    // ---- state.  Accumulator set of tile c of group grp: index c - C0 + (NTM - NT), so that a group's last tile always
    // sits in the last set -- the one the next group's first granule touches last (its gate math is still reading it)
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

    // ---- gate math, in micro-operations that are spread over the issue points between MFMA triples:
    // gate_val: one hidden value (accumulator register r, set K, tile C) -> hv[r & 3];
    // gate_fin: pair p = registers (2p, 2p + 1) -> its two new pieces; every second pair: four values complete -> stores
    auto gate_val = [&](auto c_tag, auto k_tag, auto r_tag) {
        constexpr int C = decltype(c_tag)::value, K = decltype(k_tag)::value, r = decltype(r_tag)::value;
        constexpr int p = r >> 1, qq = 2 * C + (p >> 2), d = p & 3, u = r & 1;
        const f32x16& AH = anh[C & 1];
        // (the element goes through a scalar first: __builtin_bit_cast applied directly to a vector-element expression
        // compiles to element 0 whatever the index -- hipcc 7.2)
        const u32x4 ph4 = __builtin_bit_cast(u32x4, Ph[qq]), pl4 = __builtin_bit_cast(u32x4, Pl[qq]);
        const unsigned phd = ph4[d], pld = pl4[d];
        const f16x2 oh = __builtin_bit_cast(f16x2, phd), ol = __builtin_bit_cast(f16x2, pld);
        const float hold = (float)oh[u] + (float)ol[u];
        const float rg = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(ar[K][r] * cs) + 1.0f);
        const float zg = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(az[K][r] * cs) + 1.0f);
        const float en = __builtin_fmaf(rg, AH[r], anx[K][r]) * ct2;
        const float ng = __builtin_fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(en) + 1.0f), -2.0f, 1.0f);
        hv[r & 3] = __builtin_fmaf(zg, hold - ng, ng);                  // (1 - z) n + z h
    };
    auto gate_fin = [&](auto c_tag, auto p_tag, const int tg) {
        constexpr int C = decltype(c_tag)::value, p = decltype(p_tag)::value;
        constexpr int qq = 2 * C + (p >> 2), d = p & 3, m = p >> 1;
        unsigned nh, nl;
        split_pair_h(hv[2 * (p & 1)], hv[2 * (p & 1) + 1], nh, nl);
        Pnh[qq][d] = __builtin_bit_cast(float, nh);
        Pnl[qq][d] = __builtin_bit_cast(float, nl);
        if constexpr ((p & 1) == 1) {               // values 4m .. 4m+3 = hidden units 32C + 8m + 4g + {0..3} are complete
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
    // the micro-operations of pairs P0 .. P0 + NP - 1 of tile C (val, val, fin each), third h of three
    auto gate_ops = [&](auto c_tag, auto k_tag, auto p0_tag, auto np_tag, const int h, const int tg) {
        constexpr int P0 = decltype(p0_tag)::value, NP = decltype(np_tag)::value;
        static_for<0, 3 * NP>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int pp = P0 + j / 3, kind = j % 3;
            if constexpr (pp < 8) {
                if (j / NP == h) {
                    if constexpr (kind == 0) gate_val(c_tag, k_tag, std::integral_constant<int, 2 * pp>{});
                    else if constexpr (kind == 1) gate_val(c_tag, k_tag, std::integral_constant<int, 2 * pp + 1>{});
                    else gate_fin(c_tag, std::integral_constant<int, pp>{}, tg);
                }
            }
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
    // three MFMAs: one term of the product for the three gates (consecutive MFMAs never share an accumulator)
    auto triple = [&](const f32x4 (&w)[6], const int piece, const f32x4 b, f32x16& g0, f32x16& g1, f32x16& g2) {
        g0 = mfma_h(w[0 + piece], b, g0); g1 = mfma_h(w[2 + piece], b, g1); g2 = mfma_h(w[4 + piece], b, g2);
    };

    // ---- prologue: first x chunks and the first S - 1 granules on their way, accumulators of tile 0 primed
    if constexpr (WMODE == 1) {
#pragma unroll
        for (int k = 0; k < DX; ++k) {
#pragma unroll
            for (int e = 0; e < XG; ++e) xglds(e);
            x_advance();
        }
        pf_begin();
#pragma unroll
        for (int n = 0; n < GEO::NPMAX; ++n) wglds(n);
        pf_begin();
#pragma unroll
        for (int n = 0; n < GEO::NPMAX; ++n) wglds(n);
        wait_vm_dyn(pf_np);                    // everything but the second granule has landed
        ring_barrier();
        xread(xa, xb);
    } else {
        int words;
        gran_desc(0, 0, 0, c_src, words);
        xload_direct(0, 0, xa, xb);
    }
    split2h(xa, xb, xh, xl);
    aread(wA, 0);
    init_acc(ar[NTM - NT0], 0, 0); init_acc(az[NTM - NT0], 1, 0); init_acc(anx[NTM - NT0], 2, 0);

    // one input granule: chunk q of step t into the NT tiles of group GRP; item j = the chunk's product into tile C0 + j.
    // FIRST (q == 0) also carries what is left of the previous group: the gate math of its last tile
    auto x_granule = [&](auto grp_tag, auto first_tag, const int q, const int t) {
        constexpr int GRP = decltype(grp_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int C0 = GRP == 0 ? 0 : NT0, NT = GRP == 0 ? NT0 : NCG - NT0, K0 = NTM - NT;
        constexpr int NPX_NEXT_MAX = GEO::NPMAX;                                  // the granule prefetched meanwhile may be any kind
        constexpr int NHOOK = 3 * (NT - 1);
        constexpr int PERH = (XG + NPX_NEXT_MAX + NHOOK - 1) / NHOOK;            // DMA pieces per issue point
        constexpr int PPX = (8 + NT - 2) / (NT - 1);                              // deferred gate pairs per item
        // the tile whose gates are still to do: the previous group's last one (of the previous step for group 0)
        constexpr int CD = GRP == 0 ? NCG - 1 : NT0 - 1;
        const int td = GRP == 0 ? t - 1 : t;
        const bool have_d = GRP != 0 || t > 0;
        const bool lastq = q + 1 == nqx;
        static_for<0, NT>([&](auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int K = K0 + j;
            constexpr bool LASTI = j == NT - 1;
            f32x4 (&wc)[6] = (j & 1) ? wB : wA;
            f32x4 (&wn)[6] = (j & 1) ? wA : wB;
            if constexpr (j == 0) pf_begin();
            if constexpr (LASTI) boundary(XG);
            aread(wn, LASTI ? 0 : j + 1);       // (after boundary(): slot / base of the granule that begins)
            CM_SB();
            auto hook = [&](const int h) {
                if constexpr (!LASTI && WMODE == 1) {
                    const int hi = 3 * j + h;
#pragma unroll
                    for (int u = 0; u < PERH; ++u) {
                        const int n = hi * PERH + u;
                        if (n < XG) { xglds(n); if (n == XG - 1) x_advance(); }
                        else wglds(n - XG);
                    }
                }
                if constexpr (FIRST && !LASTI) {
                    if (have_d)
                        gate_ops(std::integral_constant<int, CD>{}, std::integral_constant<int, NTM - 1>{},
                                 std::integral_constant<int, j * PPX>{}, std::integral_constant<int, PPX>{}, h, td);
                }
            };
            triple(wc, 0, xl, ar[K], az[K], anx[K]);
            CM_SB(); hook(0); CM_SB();
            if constexpr (LASTI) {
                // the next chunk's raw halves (landed: it was requested DX chunks ago)
                if constexpr (WMODE == 1) xread(xa, xb);
                else xload_direct(lastq ? 0 : q + 1, (lastq && GRP == NGRP - 1) ? t + 1 : t, xa, xb);
                CM_SB();
            }
            triple(wc, 1, xh, ar[K], az[K], anx[K]);
            CM_SB(); hook(1); CM_SB();
            triple(wc, 0, xh, ar[K], az[K], anx[K]);
            CM_SB(); hook(2); CM_SB();
            if constexpr (FIRST) {
                // accumulators of the next tile of the group (the last set only after the deferred gates are done)
                if constexpr (j + 1 < NT) {
                    if constexpr (j + 1 == NT - 1 && GRP == 0) {
                        if (have_d) fc_finish(td);
                    }
                    init_acc(ar[K + 1], 0, C0 + j + 1); init_acc(az[K + 1], 1, C0 + j + 1); init_acc(anx[K + 1], 2, C0 + j + 1);
                }
                if constexpr (LASTI) {
                    if constexpr (GRP == 0) {
                        if (have_d) {           // the state's last two chunks: the previous step is complete
                            Ph[QH - 2] = Pnh[QH - 2]; Pl[QH - 2] = Pnl[QH - 2];
                            Ph[QH - 1] = Pnh[QH - 1]; Pl[QH - 1] = Pnl[QH - 1];
                        }
                    }
                    init_acc(anh[C0 & 1], 3, C0);           // both n_h accumulators are free here: prime the group's first
                }
            }
            if constexpr (LASTI) {
                split2h(xa, xb, xnh, xnl);
                CM_SB();
            }
        });
        xh = xnh; xl = xnl;
        if constexpr (NT & 1) {                     // odd item count: bring the words of the next item back into wA
#pragma unroll
            for (int k = 0; k < 6; ++k) wA[k] = wB[k];
        }
    };

    for (int t = 0; t < T; ++t) {
        static_for<0, NGRP>([&](auto grp_tag) {
            constexpr int GRP = decltype(grp_tag)::value;
            constexpr int C0 = GRP == 0 ? 0 : NT0, NT = GRP == 0 ? NT0 : NCG - NT0, K0 = NTM - NT;
            // ---------------- input part of the group, chunk-major
            x_granule(grp_tag, std::true_type{}, 0, t);
            for (int q = 1; q < nqx; ++q) x_granule(grp_tag, std::false_type{}, q, t);
            // ---------------- recurrent part, tile-major; the gates of tile c - 1 between the MFMAs of tile c
            static_for<0, NT>([&](auto j_tag) {
                constexpr int jt = decltype(j_tag)::value;
                constexpr int c = C0 + jt, K = K0 + jt;
                static_for<0, QH>([&](auto q_tag) {
                    constexpr int qh = decltype(q_tag)::value;
                    constexpr int gi = qh % NCG;                 // item within its granule
                    constexpr bool LASTI = gi == NCG - 1;
                    constexpr int it = jt * QH + qh;             // item number within the group's recurrent part (even count per tile)
                    constexpr int NHOOK = 3 * (NCG - 1);
                    constexpr int PERH = (GEO::NPMAX + NHOOK - 1) / NHOOK;
                    f32x4 (&wc)[6] = (it & 1) ? wB : wA;
                    f32x4 (&wn)[6] = (it & 1) ? wA : wB;
                    if constexpr (gi == 0) pf_begin();
                    if constexpr (LASTI) boundary(0);
                    aread(wn, LASTI ? 0 : gi + 1);
                    CM_SB();
                    auto hook = [&](const int h) {
                        if constexpr (!LASTI && WMODE == 1) {
                            const int hi = 3 * gi + h;
#pragma unroll
                            for (int u = 0; u < PERH; ++u) wglds(hi * PERH + u);
                        }
                        if constexpr (jt > 0)
                            gate_ops(std::integral_constant<int, c - 1>{}, std::integral_constant<int, K - 1>{},
                                     std::integral_constant<int, qh * PPI>{}, std::integral_constant<int, PPI>{}, h, t);
                    };
                    f32x16& AH = anh[c & 1];
                    triple(wc, 0, Pl[qh], ar[K], az[K], AH);
                    CM_SB(); hook(0); CM_SB();
                    triple(wc, 1, Ph[qh], ar[K], az[K], AH);
                    CM_SB(); hook(1); CM_SB();
                    triple(wc, 0, Ph[qh], ar[K], az[K], AH);
                    CM_SB(); hook(2); CM_SB();
                    if constexpr (qh == QH - 1) {
                        // the other n_h accumulator is free again (tile c - 1's gates are done): prime it for tile c + 1;
                        // after the group's last tile: the first accumulator set of the next group
                        if constexpr (jt + 1 < NT) {
                            init_acc(anh[(c + 1) & 1], 3, c + 1);
                        } else {
                            constexpr int NG2 = (GRP + 1) % NGRP;
                            constexpr int C2 = NG2 == 0 ? 0 : NT0, K2 = NTM - (NG2 == 0 ? NT0 : NCG - NT0);
                            init_acc(ar[K2], 0, C2); init_acc(az[K2], 1, C2); init_acc(anx[K2], 2, C2);
                        }
                        CM_SB();
                    }
                });
            });
        });
        // every tile but the last has its new pieces: they replace the old ones (nothing reads those any more)
#pragma unroll
        for (int k = 0; k < QH - 2; ++k) { Ph[k] = Pnh[k]; Pl[k] = Pnl[k]; }
    }
    // ---------------- the last step's last tile
    static_for<0, 3>([&](auto h_tag) {
        gate_ops(std::integral_constant<int, NCG - 1>{}, std::integral_constant<int, NTM - 1>{}, std::integral_constant<int, 0>{},
                 std::integral_constant<int, 8>{}, decltype(h_tag)::value, T - 1);
    });
    fc_finish(T - 1);
    if constexpr (WMODE == 1) wait_vm<0>();
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

template <int NCG, int XMODE, bool FC, int WMODE>
int launch_cm_one(const CmArgs& a, hipStream_t s) {
    using GEO = CmGeom<NCG, XMODE>;
    const size_t lds = (size_t)GEO::L_END;     // same map in both modes (the ring is unused in mode 0)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_cm<NCG, XMODE, FC, WMODE>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
        attr_set = true;
    }
    const unsigned grid = (unsigned)((a.B + 127) / 128);
    hipLaunchKernelGGL((k_gru_cm<NCG, XMODE, FC, WMODE>), dim3(grid), dim3(256), lds, s, a);
    LAUNCH_CHECK();
    return 0;
}

template <int NCG>
int launch_cm_ncg(const CmArgs& a, int xmode, bool fc, int wmode, hipStream_t s) {
    if (wmode == 1) {
        if (xmode == 0) return launch_cm_one<NCG, 0, false, 1>(a, s);
        return fc ? launch_cm_one<NCG, 1, true, 1>(a, s) : launch_cm_one<NCG, 1, false, 1>(a, s);
    }
#ifdef MTADGAT_CM_WMODE0
    if (xmode == 0) return launch_cm_one<NCG, 0, false, 0>(a, s);
    return fc ? launch_cm_one<NCG, 1, true, 0>(a, s) : launch_cm_one<NCG, 1, false, 0>(a, s);
#else
    return -2;
#endif
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

int launch_gru_cm(const GruArgs& g, int ncg, int xmode, bool fc, int wmode, hipStream_t s) {
    if (g.B <= 0) return 0;
    if (!gru_cm_supported(ncg, xmode, fc, g.out_dim)) return -2;
    if (xmode == 0 && (g.ldx & 3) != 0) return -2;
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
        case MTADGAT_CM_ONLY: return launch_cm_ncg<MTADGAT_CM_ONLY>(a, xmode, fc, wmode, s);
#else
        case 2: return launch_cm_ncg<2>(a, xmode, fc, wmode, s);
        case 3: return launch_cm_ncg<3>(a, xmode, fc, wmode, s);
        case 4: return launch_cm_ncg<4>(a, xmode, fc, wmode, s);
        case 5: return launch_cm_ncg<5>(a, xmode, fc, wmode, s);
#endif
        default: return -2;
    }
}

}  // namespace mtadgat
