// Do the matrix pipe and the vector pipe of a gfx950 SIMD run concurrently when they are fed by DIFFERENT waves?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run an MFMA-only loop, waves 4-7 a VALU-only loop of
// about the same stand-alone duration.  Times: MFMA waves alone, VALU waves alone, both.  both ~ max -> concurrent;
// both ~ sum -> the pipes exclude each other.   hipcc --offload-arch=gfx950 -O3 ubench_overlap.hip -o ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define REP8(x) x x x x x x x x
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int mfma_on, int valu_on, int valu_kind, int mfma_kind) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (!mfma_on) return;
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        const float x = threadIdx.x * 0.001f, y = 0.5f;
        if (mfma_kind == 1) {              // bf16 operands: v_mfma_f32_32x32x16_bf16 (8 passes), twice as many per iteration
            bf16x8 xb, yb;
            for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)(x + e); yb[e] = (__bf16)(y * e); }
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a3, 0, 0, 0);
                }
            }
        } else
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!valu_on) return;
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
        const float b = threadIdx.x * 0.5f;
        for (int i = 0; i < iters; ++i) {
            if (valu_kind == 0) {          // v_add_f32, 8 independent chains, 192 per iteration
                REP8(asm volatile("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                                  "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                                  "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                                  "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                                  "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"
                                  "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
            } else {                       // the k_gat pair: add then |.|-accumulate
                REP8(asm volatile("v_add_f32 %0, %8, %4\n v_add_f32 %1, %8, %5\n v_add_f32 %2, %8, %6\n v_add_f32 %3, %8, %7\n"
                                  "v_add_f32_e64 %4, %4, |%0|\n v_add_f32_e64 %5, %5, |%1|\n v_add_f32_e64 %6, %6, |%2|\n v_add_f32_e64 %7, %7, |%3|\n"
                                  "v_add_f32 %0, %8, %4\n v_add_f32 %1, %8, %5\n v_add_f32 %2, %8, %6\n v_add_f32 %3, %8, %7\n"
                                  "v_add_f32_e64 %4, %4, |%0|\n v_add_f32_e64 %5, %5, |%1|\n v_add_f32_e64 %6, %6, |%2|\n v_add_f32_e64 %7, %7, |%3|\n"
                                  "v_add_f32 %0, %8, %4\n v_add_f32 %1, %8, %5\n v_add_f32 %2, %8, %6\n v_add_f32 %3, %8, %7\n"
                                  "v_add_f32_e64 %4, %4, |%0|\n v_add_f32_e64 %5, %5, |%1|\n v_add_f32_e64 %6, %6, |%2|\n v_add_f32_e64 %7, %7, |%3|\n"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
static float run(float* d, int iters, int m, int v, int kind, int mk) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<256, 512>>>(d, iters, m, v, kind, mk); hipDeviceSynchronize();
    hipEventRecord(a); k<<<256, 512>>>(d, iters, m, v, kind, mk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 4000;
    for (int mk = 0; mk < 2; ++mk)
    for (int kind = 0; kind < 2; ++kind) {
        const float tm = run(d, iters, 1, 0, kind, mk), tv = run(d, iters, 0, 1, kind, mk), tb = run(d, iters, 1, 1, kind, mk);
        printf("%s | %s: MFMA waves alone %.3f ms (%.1f cyc per instr), VALU waves alone %.3f ms (%.2f cyc per instr), both %.3f ms"
               "  -> sum %.3f, max %.3f, overlap %.0f %% of the shorter\n", mk == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_32x32x16_bf16",
               kind == 0 ? "v_add_f32 x 192" : "k_gat pair (add, |.|-acc) x 192", tm, tm * 2.4e6 / (iters * (mk ? 32.0 : 16.0)), tv, tv * 2.4e6 / (iters * 192.0), tb,
               tm + tv, tm > tv ? tm : tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
    }
    // the same question for two waves per SIMD of each kind
    return 0;
}
