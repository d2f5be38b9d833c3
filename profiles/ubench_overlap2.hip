// Does work issued by the SAME wave overlap with v_mfma_f32_32x32x16_f16 (one wave per SIMD)?  Variants of where the MFMA's
// accumulator lives and what the fillers are.  ticks = s_memtime cycles per MFMA + fillers.
//   hipcc --offload-arch=gfx950 -O2 [-mllvm -amdgpu-mfma-vgpr-form] -o profiles/bin/ubench_overlap2 profiles/ubench_overlap2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters, const float* gbuf) {
    __shared__ f32x4 ldsbuf[2048];
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = 0.6f, a2 = 0.7f, a3 = 0.8f, a4 = 0.9f, a5 = 1.0f;
    const float b = 0.999f;
    f32x16 acc0, acc1, acc2;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; }
    f32x4 wv = {a0, a1, a2, a3}, xv = {a4, a5, a0, a1};
    ldsbuf[threadIdx.x] = wv; ldsbuf[threadIdx.x + 256] = xv;
    __syncthreads();
    const f32x4* lp = &ldsbuf[threadIdx.x & 63];
    float rb = b;
    float ta;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ta) : "v"(a0));
    int sa = 0, sb = 1, cnt = 0;
    const unsigned voff = (threadIdx.x & 63) * 16;
    const unsigned ldst = 8192 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 4096;
    const float* gsrc = gbuf + (blockIdx.x & 7) * 4096;
    f32x4 l0 = wv, l1 = wv, l2 = wv, l3 = wv;
    const long long s = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        REP16(
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv), __builtin_bit_cast(f16x8, xv), acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == 1) asm volatile("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
                                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b));
            if constexpr (MODE == 2) {      // fillers read another accumulator (v_accvgpr_read when it lives in the AGPR file)
                a0 += acc1[0]; a1 += acc1[1]; a2 += acc1[2]; a3 += acc1[3]; a4 += acc1[4]; a5 += acc1[5];
            }
            if constexpr (MODE == 5) {      // six reads of an accumulator-file register that no MFMA touches
                asm volatile("v_accvgpr_read_b32 %0, %6\n v_accvgpr_read_b32 %1, %6\n v_accvgpr_read_b32 %2, %6\n v_accvgpr_read_b32 %3, %6\n v_accvgpr_read_b32 %4, %6\n v_accvgpr_read_b32 %5, %6\n"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5) : "a"(ta));
            }
            if constexpr (MODE == 6) {      // three v_exp + three v_rcp
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
            }
            if constexpr (MODE == 7) {      // six scalar instructions
                asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n" : "+s"(sa), "+s"(sb));
            }
            if constexpr (MODE == 8) {      // four LDS reads whose results are not needed soon
                l0 = lp[0]; l1 = lp[64]; l2 = lp[128]; l3 = lp[192];
            }
            if constexpr (MODE == 9) {      // one LDS-DMA piece (1 KiB per wave) per MFMA, as k_gru_cm issues them
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(ldst) : "memory");
            }
            if constexpr (MODE == 10) {     // a burst of four pieces every fourth MFMA
                if ((++cnt & 3) == 0) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(ldst) : "memory");
                }
            }
            if constexpr (MODE == 11 || MODE == 12) {     // one piece every 4th (11) / 8th (12) MFMA: below the 64 B/clk of the CU's memory path
                if ((++cnt & (MODE == 11 ? 3 : 7)) == 0) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(voff), "s"(gsrc), "s"(ldst) : "memory");
                }
            }
            if constexpr (MODE == 3) {      // the A operand of the next MFMA comes from LDS
                wv = lp[(it & 1) * 64];
            }
            if constexpr (MODE == 4) {      // two more MFMAs on other accumulators, then 6 v_mul
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv), __builtin_bit_cast(f16x8, xv), acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5) : "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long e = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = e - s;
    float r = a0 + a1 + a2 + a3 + a4 + a5 + rb + wv[0] + (float)(sa + sb) + l0[0] + l1[1] + l2[2] + l3[3];
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
static void run(const char* name, int mfma_per_iter) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 200;
    float* gbuf; (void)hipMalloc(&gbuf, 1 << 20); (void)hipMemset(gbuf, 0, 1 << 20);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters, gbuf);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters, gbuf);
    (void)hipDeviceSynchronize();
    long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    std::printf("%-70s ticks per MFMA %7.2f\n", name, (double)c / ((double)iters * mfma_per_iter));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("MFMA alone (dependent chain on one accumulator)", 16);
    run<1>("MFMA + 6 v_mul", 16);
    run<2>("MFMA + 6 v_add reading another accumulator", 16);
    run<3>("MFMA + its A operand from ds_read_b128", 16);
    run<4>("2 MFMAs (two accumulators) + 6 v_mul after the second", 32);
    run<5>("MFMA + 6 v_accvgpr_read (register no MFMA touches)", 16);
    run<6>("MFMA + 3 v_exp + 3 v_rcp", 16);
    run<7>("MFMA + 6 s_add", 16);
    run<8>("MFMA + 4 ds_read_b128 (results unused)", 16);
    run<9>("MFMA + 1 LDS-DMA piece (L2 resident source)", 16);
    run<10>("MFMA, a burst of 4 LDS-DMA pieces every 4th", 16);
    run<11>("MFMA, one LDS-DMA piece every 4th", 16);
    run<12>("MFMA, one LDS-DMA piece every 8th", 16);
    return 0;
}
