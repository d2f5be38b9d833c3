// Device-side helpers shared by the kernel translation units (everything is inline).
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
#ifndef MTADGAT_DEVICE_H
#define MTADGAT_DEVICE_H
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "mtadgat_kernels.h"

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

// ---- bf16 operand mode (opt-in; fp32 accumulation, fp32 state / softmax / gates) ---------------------------
// v_mfma_f32_32x32x16_bf16 contracts 16 features per instruction: lane (i, g) supplies 8 of them.  A "bf16
// chunk" q is the pair of F-layout chunks 2q, 2q+1: element e of lane g is feature
//     16 q + (e < 4 ? 4 g + e : 8 + 4 g + (e - 4))
// i.e. exactly the eight fp32 values the lane already holds for those two chunks (its accumulator registers
// 8(q&1)..8(q&1)+7 of tile q/2, or two float4 loads) -- converting them with v_cvt_pk_bf16_f32 gives the B
// operand with no cross-lane traffic, and the weights are packed on the host in the same element order.
// The 16-byte operand travels in an f32x4 container so that the weight ring code is shared with the fp32 build.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));        // v_cvt_pk_bf16_f32, round to nearest even
}
__device__ __forceinline__ f32x4 cvt8(const f32x4 lo, const f32x4 hi) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 r = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
    return __builtin_bit_cast(f32x4, r);
}
__device__ __forceinline__ f32x16 mfma_bf(const f32x4 w, const f32x4 x, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
// ---- split-bf16 operand mode ("x3"): fp32-class accuracy on the bf16 matrix pipe ---------------------------------
// On gfx950 the fp32 MFMA shares the fp32 vector datapath: it runs at the VALU rate (64 FLOP/clk/SIMD) and does NOT
// overlap with VALU work of other waves (profiles/r02_mfma_valu_overlap.txt: 1-10 %), while the bf16 MFMA is 16x
// faster per flop and overlaps (~80 %).  Every fp32 value is the exact sum of three bf16 pieces to 2^-24 relative
// (x = hi + mid + lo, each piece the bf16 rounding of the remaining residual), so
//     w x  ~=  wh xh + wh xm + wm xh + wm xm + wh xl + wl xh          (dropped terms <= 2^-24 |w x| each)
// is six v_mfma_f32_32x32x16_bf16 (16 features, 6 x 32 cycles) in place of eight v_mfma_f32_32x32x2_f32 (8 x 64
// cycles) for the same 16 features: 2.7x less pipe time, on a pipe that runs beside the VALU.  Products of bf16 pairs
// are exact in fp32 and the accumulation is the same fp32 accumulation, so the result differs from the fp32 MFMA's by
// a few 2^-24 per product.  Weights are split once per load (k_split3), activations where they are consumed.
__device__ __forceinline__ void split3(const f32x4 a, const f32x4 b, f32x4& hi, f32x4& mid, f32x4& lo) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 h, m, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float v0 = p < 2 ? a[2 * p] : b[2 * p - 4], v1 = p < 2 ? a[2 * p + 1] : b[2 * p - 3];
        const unsigned hw = pack_bf16(v0, v1);
        const float r0 = v0 - __builtin_bit_cast(float, hw << 16), r1 = v1 - __builtin_bit_cast(float, hw & 0xffff0000u);
        const unsigned mw = pack_bf16(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, mw << 16), s1 = r1 - __builtin_bit_cast(float, mw & 0xffff0000u);
        h[p] = hw; m[p] = mw; l[p] = pack_bf16(s0, s1);
    }
    hi = __builtin_bit_cast(f32x4, h); mid = __builtin_bit_cast(f32x4, m); lo = __builtin_bit_cast(f32x4, l);
}
// ---- two fp16 pieces: for operands of known range (the recurrent state, |h| <= 1; weights scaled by a power of two into
// fp16's normal range): x = hi + lo to 2^-22 relative (or 2^-25 absolute, fp16's subnormal spacing), fp16 x fp16 products
// are exact in fp32, and   w x ~= wh xh + wh xl + wl xh   is THREE v_mfma_f32_32x32x16_f16 per 16 features, with 4 bytes
// per weight instead of 6 and a 5-instruction split per value pair instead of 11.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));       // v_cvt_pk_f16_f32, round to nearest even
}
__device__ __forceinline__ void split_pair_h(float v0, float v1, unsigned& hw, unsigned& lw) {
    hw = pack_f16(v0, v1);
    const f16x2 hh = __builtin_bit_cast(f16x2, hw);
    lw = pack_f16(v0 - (float)hh[0], v1 - (float)hh[1]);
}
__device__ __forceinline__ void split2h(const f32x4 a, const f32x4 b, f32x4& hi, f32x4& lo) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hw, lw;
        split_pair_h(p < 2 ? a[2 * p] : b[2 * p - 4], p < 2 ? a[2 * p + 1] : b[2 * p - 3], hw, lw);
        h[p] = hw; l[p] = lw;
    }
    hi = __builtin_bit_cast(f32x4, h); lo = __builtin_bit_cast(f32x4, l);
}
__device__ __forceinline__ f32x16 mfma_h(const f32x4 w, const f32x4 x, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}

// w: the three pieces of the weight word (hi, mid, lo); x likewise
__device__ __forceinline__ f32x16 mfma_s3(const f32x4 (&w)[3], const f32x4 (&x)[3], f32x16 acc) {
    acc = mfma_bf(w[0], x[2], acc);
    acc = mfma_bf(w[2], x[0], acc);
    acc = mfma_bf(w[1], x[1], acc);
    acc = mfma_bf(w[0], x[1], acc);
    acc = mfma_bf(w[1], x[0], acc);
    acc = mfma_bf(w[0], x[0], acc);
    return acc;
}

// one chunk for three gate accumulators: fp32 build 4 x 3 MFMAs (8 features), bf16 build 3 MFMAs (16 features)
template <bool BF>
__device__ __forceinline__ void mfma_x3(const f32x4 (&w)[3], const f32x4 x, f32x16& a0, f32x16& a1, f32x16& a2);

// four features k0..k0+3 of a row; zero beyond kvalid.  Branch-free: the loads are unconditional from addresses inside the row's
// valid features and masked afterwards -- with a guarded load per lane (the two halves of a wave hold different k0) every call was
// a divergent branch whose loads waited for everything in flight before them (the weight words of the same chunk), one memory
// round trip after another.  XV (kvalid >= 4, k0 a multiple of 4): one 16-byte load; the group that straddles kvalid is read as
// the row's last four valid features and shifted into place.
// Two halves, so that a software-pipelined loop can issue the raw load one chunk ahead and look at the values only when it uses them
// (feat4_fix right behind the load would wait for it, and for every load issued before it).
template <bool XV>
__device__ __forceinline__ f32x4 feat4_raw(const float* __restrict__ row, int k0, int kvalid) {
    f32x4 v;
    if constexpr (XV) {
        v = *reinterpret_cast<const f32x4*>(row + (k0 + 3 < kvalid ? k0 : kvalid - 4));
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = row[k0 + e < kvalid ? k0 + e : kvalid - 1];
    }
    return v;
}
template <bool XV>
__device__ __forceinline__ f32x4 feat4_fix(const f32x4 w, int k0, int kvalid) {
    f32x4 v = w;
    if constexpr (XV) {
        if (kvalid & 3) {                                   // wave-uniform: a straddling group exists
            const int sh = (k0 + 3 < kvalid) ? 0 : (k0 - kvalid) & 3;      // = k0 - (kvalid - 4): 1..3 for the straddling group
            const bool s1 = sh & 1, s2 = sh & 2;
            const float a0 = s1 ? w[1] : w[0], a1 = s1 ? w[2] : w[1], a2 = s1 ? w[3] : w[2];
            v[0] = s2 ? a2 : a0;
            v[1] = s2 ? w[3] : a1;
            v[2] = a2;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k0 + e < kvalid) ? v[e] : 0.f;
    return v;
}

template <>
__device__ __forceinline__ void mfma_x3<false>(const f32x4 (&w)[3], const f32x4 x, f32x16& a0, f32x16& a1, f32x16& a2) {
    mfma4x3(w, x, a0, a1, a2);
}
template <>
__device__ __forceinline__ void mfma_x3<true>(const f32x4 (&w)[3], const f32x4 x, f32x16& a0, f32x16& a1, f32x16& a2) {
    a0 = mfma_bf(w[0], x, a0);
    a1 = mfma_bf(w[1], x, a1);
    a2 = mfma_bf(w[2], x, a2);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// GRU gate non-linearities on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each):
// a handful of VALU ops instead of ~40 for the libm versions.  exp_fast (softmax, output sigmoid) forms the
// product x*log2(e) in two pieces so the exponent keeps float accuracy for |x| up to ~40.
#ifndef MTADGAT_ACCURATE_GATES
__device__ __forceinline__ float exp_fast(float x) {   // e^x, argument clamped to [-88, 88] (no inf/NaN in, none out)
    x = __builtin_amdgcn_fmed3f(x, -88.0f, 88.0f);
    const float c_hi = 1.4426950216293335f;             // log2(e) rounded to float
    const float c_lo = 1.9259629911266175e-08f;          // log2(e) - c_hi
    const float hi = x * c_hi;
    const float lo = __builtin_fmaf(x, c_hi, -hi) + x * c_lo;
    return __builtin_amdgcn_exp2f(hi) * (1.0f + 0.6931471805599453f * lo);
}
// Gates: the one-piece product x*log2(e) is enough.  Its rounding error (|x| * 1.44 * 2^-24 relative in e^x)
// is multiplied by sigma(1 - sigma) resp. (1 - tanh^2), which decay faster than |x| grows: the result moves by
// < 2e-8 (sigmoid) / < 4e-8 (tanh), below the fp32 spacing of the gate values themselves.  5-6 VALU ops per gate.
__device__ __forceinline__ float gate_sigmoid(float x) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -60.0f, 60.0f) * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gate_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -30.0f, 30.0f) * 2.8853900817779268f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
__device__ __forceinline__ float soft_exp(float x) { return exp_fast(x); }
__device__ __forceinline__ float soft_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
__device__ __forceinline__ float gate_sigmoid(float x) { return sigmoidf_(x); }
__device__ __forceinline__ float gate_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ float soft_exp(float x) { return expf(x); }
__device__ __forceinline__ float soft_rcp(float x) { return 1.0f / x; }
#endif

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

// ---- dropout: counter-based, stateless.  The keep decision of element `idx` of window `win` in dropout
// stream `stream` is a pure function of (seed, stream, win, idx), so the forward, a re-computed forward and
// the backward see the same mask whatever the chunking / sharding of the batch (reference: F.dropout on the
// attention matrices, modules.py:90 / :189, and nn.Dropout in Forecasting_Model, modules.py:310 -- same
// distribution, not the same random stream).
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned drop_window_key(const DropArgs& d, unsigned stream, long win_local) {
    const unsigned long w = (unsigned long)(d.win0 + win_local);
    unsigned h = mix32((unsigned)w ^ d.seed_lo);
    h = mix32(h + (unsigned)(w >> 32) * 0x9E3779B9U + d.seed_hi + stream * 0x85EBCA6BU);
    return h;
}
__device__ __forceinline__ bool drop_keep(unsigned key, unsigned idx, unsigned thresh) {
    return mix32(key ^ (idx * 0x9E3779B1U + 0x7F4A7C15U)) >= thresh;
}

#define LAUNCH_CHECK()                          \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

}  // namespace mtadgat
#endif  // MTADGAT_DEVICE_H
