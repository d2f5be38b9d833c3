// Issue cost of the instruction kinds in k_gru_cm's gate math on gfx950, ONE wave per SIMD (the kernel's occupancy):
// cycles per instruction (s_memtime, wave 0 of block 0) for independent streams and for dependent chains, alone and in
// the shadow of v_mfma_f32_32x32x16_f16.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/bin/ubench_gate profiles/ubench_gate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// 8 independent registers a0..a7; OP is applied to each in turn (independent stream) -- 8 instructions per expansion
#define IND8(OP) asm volatile(OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define DEP8(OP) asm volatile(OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" OP " %0, %0\n" : "+v"(a0));
#define IND8_2(OP) asm volatile(OP " %0, %8, %0\n" OP " %1, %8, %1\n" OP " %2, %8, %2\n" OP " %3, %8, %3\n" OP " %4, %8, %4\n" OP " %5, %8, %5\n" OP " %6, %8, %6\n" OP " %7, %8, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define DEP8_2(OP) asm volatile(OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" OP " %0, %1, %0\n" : "+v"(a0) : "v"(b));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = 0.6f, a2 = 0.7f, a3 = 0.8f, a4 = 0.9f, a5 = 1.0f, a6 = 1.1f, a7 = 1.2f;
    float b = 0.999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b, b};
    f32x16 acc0, acc1, acc2;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; }
    f32x4 wv = {a0, a1, a2, a3}, xv = {a4, a5, a6, a7};
    const f16x8 wh = __builtin_bit_cast(f16x8, wv), xh = __builtin_bit_cast(f16x8, xv);
    const long long s = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) { REP16(IND8("v_exp_f32")) }
        else if constexpr (MODE == 1) { REP16(IND8("v_rcp_f32")) }
        else if constexpr (MODE == 2) { REP16(DEP8("v_exp_f32")) }
        else if constexpr (MODE == 3) { REP16(IND8_2("v_mul_f32")) }
        else if constexpr (MODE == 4) { REP16(DEP8_2("v_mul_f32")) }
        else if constexpr (MODE == 5) {
            REP16(asm volatile("v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3\n"
                               "v_pk_mul_f32 %0, %4, %0\n v_pk_mul_f32 %1, %4, %1\n v_pk_mul_f32 %2, %4, %2\n v_pk_mul_f32 %3, %4, %3\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if constexpr (MODE == 6) {   // accumulator file reads
            REP16(asm volatile("v_accvgpr_read_b32 %0, %8\n v_accvgpr_read_b32 %1, %8\n v_accvgpr_read_b32 %2, %8\n v_accvgpr_read_b32 %3, %8\n"
                               "v_accvgpr_read_b32 %4, %8\n v_accvgpr_read_b32 %5, %8\n v_accvgpr_read_b32 %6, %8\n v_accvgpr_read_b32 %7, %8\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "a"(b));)
        } else if constexpr (MODE == 7) {   // bare MFMA, three accumulators round robin
            REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc1, 0, 0, 0);
                  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);)
        } else if constexpr (MODE >= 8 && MODE <= 12) {
            // one MFMA, then N fillers: 8 = 6 v_mul (independent), 9 = 4 v_exp + 2 v_mul, 10 = 8 v_mul, 11 = 2 v_exp + 2 v_rcp + 2 v_mul dependent pairs,
            // 12 = 6 dependent v_mul
            REP16(
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc0, 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                if constexpr (MODE == 8) asm volatile("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
                                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b));
                if constexpr (MODE == 9) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
                                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b));
                if constexpr (MODE == 10) asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                                                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
                if constexpr (MODE == 11) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_mul_f32 %0, %2, %0\n v_mul_f32 %1, %2, %1\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n"
                                                       : "+v"(a0), "+v"(a1) : "v"(b));
                if constexpr (MODE == 12) asm volatile("v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n v_mul_f32 %0, %1, %0\n"
                                                       : "+v"(a0) : "v"(b));
                __builtin_amdgcn_sched_barrier(0);)
        } else if constexpr (MODE == 13) {  // fp16 <-> fp32 conversions as used by the piece split
            REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n v_cvt_f32_f16 %1, %0\n v_cvt_f32_f16_sdwa %2, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_cvt_pk_f16_f32 %3, %4, %5\n v_cvt_f32_f16 %4, %3\n v_cvt_f32_f16_sdwa %5, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                               "v_sub_f32 %6, %1, %2\n v_sub_f32 %7, %4, %5\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 14) {  // s_ scalar adds
            int s0 = it, s1 = it + 1;
            REP16(asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n" : "+s"(s0), "+s"(s1));)
            a0 += (float)(s0 + s1);
        }
    }
    const long long e = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = e - s;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, int per_iter_instr, int per_iter_mfma) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on some parts; report raw ticks too
    std::printf("%-58s ticks %10lld  per instr %7.3f", name, c, (double)c / ((double)iters * per_iter_instr));
    if (per_iter_mfma) std::printf("  per MFMA(+fillers) %7.3f", (double)c / ((double)iters * per_iter_mfma));
    std::printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<3>("v_mul_f32 independent (8 streams)", 128, 0);
    run<4>("v_mul_f32 dependent chain", 128, 0);
    run<0>("v_exp_f32 independent", 128, 0);
    run<2>("v_exp_f32 dependent chain", 128, 0);
    run<1>("v_rcp_f32 independent", 128, 0);
    run<5>("v_pk_mul_f32 independent (4 streams)", 128, 0);
    run<6>("v_accvgpr_read_b32", 128, 0);
    run<13>("cvt_pk_f16 / cvt_f32_f16 / sdwa / sub mix", 128, 0);
    run<14>("s_add_i32 (2 streams)", 128, 0);
    run<7>("v_mfma_f32_32x32x16_f16 bare (3 accumulators)", 48, 48);
    run<8>("MFMA + 6 independent v_mul", 16 * 7, 16);
    run<10>("MFMA + 8 independent v_mul", 16 * 9, 16);
    run<9>("MFMA + 4 v_exp + 2 v_mul", 16 * 7, 16);
    run<11>("MFMA + (exp,exp,mul,mul,rcp,rcp) two dependent chains", 16 * 7, 16);
    run<12>("MFMA + 6 dependent v_mul", 16 * 7, 16);
    return 0;
}
