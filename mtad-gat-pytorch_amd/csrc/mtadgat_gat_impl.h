// Shared pieces of the fused graph-attention kernels (mtadgat_gat.hip: k_gat, k_gat_wide; mtadgat_gath.hip: k_gath): the 2-D
// register-blocked pair grid over LDS-resident L' / R' rows and the row reductions of the softmax.  See the header comment of
// mtadgat_gat.hip for the layout.
#pragma once
#include "mtadgat_device.h"

namespace mtadgat {

#ifndef MTADGAT_GAT_MINW
#define MTADGAT_GAT_MINW 4
#endif
#ifndef MTADGAT_GAT_QB3
#define MTADGAT_GAT_QB3 2
#endif

typedef const __attribute__((address_space(3))) float* lds_cptr;      // explicit LDS pointer (32-bit)
constexpr int GAT_LLD = 34;     // 32 columns + 2: rows 8-byte aligned, 16 consecutive rows start on 16 distinct bank pairs
constexpr int GAT_APITCH = 68;

// The layer's column-order record {P8, PT} (k_gat_colorder writes it on the device when the weights change): both words in ONE
// scalar load.  Written as asm because the compiler does not know that nothing writes the record while the kernel runs and
// reads it with two vector loads, each waited for in turn -- two memory round trips at the head of every workgroup.
__device__ __forceinline__ void gat_load_order(const int* ord, int P8_host, int PT_host, int& P8, int& PT) {
    P8 = P8_host; PT = PT_host;
    if (ord) {
        typedef int i32x2_ __attribute__((ext_vector_type(2)));
        i32x2_ pr;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr) : "s"(ord) : "memory");
        P8 = pr[0]; PT = pr[1];
    }
}

// ... and with the number of non-negative columns (k_gath's compact order): [P8, PT, npos, 0] in ONE 16-byte scalar load
__device__ __forceinline__ void gat_load_order3(const int* ord, int& P8, int& PT, int& npos) {
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    i32x4_ pr;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pr) : "s"(ord) : "memory");
    P8 = pr[0]; PT = pr[1]; npos = pr[2];
}

// lp[ii]: one base pointer per query row.  The pointers are made opaque to the compiler on purpose:
// with a common base it merges row pairs into ds_read2_b64, which runs at half the LDS rate of two
// ds_read_b64 (MI355X: 8 vs 2 x 2 LDS cycles per wave instruction).
template <int IBL, int JPL, int RJ>
__device__ __forceinline__ void gat_load(f32x2 (&l)[IBL], f32x2 (&r)[JPL], const lds_cptr (&lp)[IBL], lds_cptr rp, int col) {
    typedef const __attribute__((address_space(3))) f32x2* lds_c2;
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) l[ii] = *(lds_c2)(lp[ii] + col);
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) r[jj] = *(lds_c2)(rp + jj * RJ * GAT_LLD + col);
}

// The two instructions per pair are written as (volatile) inline asm: left to itself the compiler packs
// the column pair into v_pk_add_f32 (no faster -- neither its form with shuffles nor, round 3, a hand-placed
// v_pk_add_f32 on the 8-byte words as loaded, 3 instructions per 2 pair-columns: 10.91 -> 10.82 ms, the pair grid is
// not bound by VALU issue alone) and schedules all sums of a step
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
template <int IBL, int JPL, int RJ, bool NEG>
__device__ __forceinline__ void gat_tile(float (&acc)[IBL][JPL], f32x2 (&lA)[IBL], f32x2 (&rA)[JPL], f32x2 (&lB)[IBL],
                                         f32x2 (&rB)[JPL], const lds_cptr (&lp)[IBL], lds_cptr rp) {
    gat_load<IBL, JPL, RJ>(lB, rB, lp, rp, 2);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lA, rA);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lA, rA, lp, rp, 4);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lB, rB);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lB, rB, lp, rp, 6);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lA, rA);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lA, rA, lp, rp, 8);
    __builtin_amdgcn_sched_barrier(0);
    gat_step<IBL, JPL, NEG>(acc, lB, rB);
    __builtin_amdgcn_sched_barrier(0);
}

// k_gath's tile (round 6): the sign of a 2-column step is a wave-uniform scalar (+1 / -1) that rides in the accumulate --
// v_fma_f32 acc, |t|, sgn, acc is the same one full-rate instruction as v_add / v_sub with the |t| modifier and rounds the same --
// so one loop serves non-negative, negative and the tile on the sign boundary of the compact column order (whose first steps add
// and whose last steps subtract), without a third copy of the unrolled tile body in the kernel.
template <int IBL, int JPL>
__device__ __forceinline__ void gat_step_s(float (&acc)[IBL][JPL], const f32x2 (&l)[IBL], const f32x2 (&r)[JPL], float sgn) {
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
            for (int jj = 0; jj < JPL; ++jj)
                asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(acc[ii][jj]) : "v"(t[jj]), "s"(sgn));
        }
}
// one 8-column tile; npl = its non-negative 2-column steps (4: all add, 0: all subtract)
template <int IBL, int JPL, int RJ>
__device__ __forceinline__ void gat_tile_s(float (&acc)[IBL][JPL], f32x2 (&lA)[IBL], f32x2 (&rA)[JPL], f32x2 (&lB)[IBL],
                                           f32x2 (&rB)[JPL], const lds_cptr (&lp)[IBL], lds_cptr rp, int npl) {
    const float s0 = npl > 0 ? 1.f : -1.f, s1 = npl > 1 ? 1.f : -1.f, s2 = npl > 2 ? 1.f : -1.f, s3 = npl > 3 ? 1.f : -1.f;
    gat_load<IBL, JPL, RJ>(lB, rB, lp, rp, 2);
    __builtin_amdgcn_sched_barrier(0);
    gat_step_s<IBL, JPL>(acc, lA, rA, s0);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lA, rA, lp, rp, 4);
    __builtin_amdgcn_sched_barrier(0);
    gat_step_s<IBL, JPL>(acc, lB, rB, s1);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lB, rB, lp, rp, 6);
    __builtin_amdgcn_sched_barrier(0);
    gat_step_s<IBL, JPL>(acc, lA, rA, s2);
    __builtin_amdgcn_sched_barrier(0);
    gat_load<IBL, JPL, RJ>(lA, rA, lp, rp, 8);
    __builtin_amdgcn_sched_barrier(0);
    gat_step_s<IBL, JPL>(acc, lB, rB, s3);
    __builtin_amdgcn_sched_barrier(0);
}

// all-reduce over the RJ (16 or 8) adjacent lanes that hold one query row
template <int RJ>
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    if (RJ == 16) v = fmaxf(v, dpp_move<0x140>(v));
    return v;
}
template <int RJ>
__device__ __forceinline__ float row_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    if (RJ == 16) v += dpp_move<0x140>(v);
    return v;
}

}  // namespace mtadgat
