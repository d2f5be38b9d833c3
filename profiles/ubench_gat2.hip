// Pair-grid microbenchmark for the column-sliced attention kernel (k_gat2): every wave owns ALL
// K x K node pairs of a window (lane (li, lj) of an 8 x 8 lane grid: IBL consecutive query rows x JPL consecutive keys,
// IBL*JPL accumulators) and a slice of the embedding columns; per column it reads its IBL + JPL operands from LDS with
// (IBL + JPL) / 4 wide reads and issues 2*IBL*JPL VALU instructions.  Question: how many cycles does a column cost per
// wave at 2 waves per SIMD (13 x 13, K = 100) and at 4 waves per SIMD (7 x 7, K = 55)?
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize ubench_gat2.hip -o bin/ubench_gat2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) float* lds_cptr;
typedef const __attribute__((address_space(3))) f32x4* lds_c4;
typedef const __attribute__((address_space(3))) f32x2* lds_c2;

template <int N>
struct Slots {                                  // N = 4 A + B values per lane group: A 16-byte reads + one read of the rest
    static constexpr int A = N / 4, B = N % 4, EB = B == 3 ? 4 : B;
    static constexpr int MAIN = 8 * 4 * A, TOTAL = MAIN + 8 * EB;
};

template <int N>
__device__ __forceinline__ void load_col(float (&v)[N], lds_cptr pm, lds_cptr pe, int off) {
    constexpr int A = Slots<N>::A, B = Slots<N>::B;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        const f32x4 t = *(lds_c4)(pm + off + 4 * a);
        v[4 * a] = t[0]; v[4 * a + 1] = t[1]; v[4 * a + 2] = t[2]; v[4 * a + 3] = t[3];
    }
    if constexpr (B == 1) v[4 * A] = pe[off];
    if constexpr (B == 2) { const f32x2 t = *(lds_c2)(pe + off); v[4 * A] = t[0]; v[4 * A + 1] = t[1]; }
    if constexpr (B == 3) { const f32x4 t = *(lds_c4)(pe + off); v[4 * A] = t[0]; v[4 * A + 1] = t[1]; v[4 * A + 2] = t[2]; }
}

template <int IBL, int JPL, int MODE>
__device__ __forceinline__ void pair_step(float (&acc)[IBL][JPL], const float (&l)[IBL], const float (&r)[JPL], float s) {
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii) {
        float t[JPL];
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(t[jj]) : "v"(l[ii]), "v"(r[jj]));
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            if (MODE == 0) asm volatile("v_add_f32_e64 %0, %0, |%1|" : "+v"(acc[ii][jj]) : "v"(t[jj]));
            else asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(acc[ii][jj]) : "v"(t[jj]), "v"(s));
        }
    }
}

// CW columns per wave and window, CS floats between columns; L and R slices of the wave back to back
template <int IBL, int JPL, int MODE, int NT, int MINW>
__global__ __launch_bounds__(NT, MINW) void k_pair(float* out, int iters, int CW, int CS, int lds_floats_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane >> 3, lj = lane & 7;
    float* mine = smem + wave * lds_floats_per_wave;
    for (int u = lane; u < lds_floats_per_wave; u += 64) mine[u] = (float)((u * 2654435761u + wave) & 1023) * (1.f / 512.f) - 1.f;
    __syncthreads();
    lds_cptr Lw = (lds_cptr)mine, Rw = (lds_cptr)(mine + CW * CS);
    lds_cptr lm = Lw + li * 4 * Slots<IBL>::A, le = Lw + Slots<IBL>::MAIN + li * Slots<IBL>::EB;
    lds_cptr rm = Rw + lj * 4 * Slots<JPL>::A, re = Rw + Slots<JPL>::MAIN + lj * Slots<JPL>::EB;
    float acc[IBL][JPL];
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) acc[ii][jj] = 0.f;
    float s = 1.f;
    asm volatile("" : "+v"(s));
    for (int it = 0; it < iters; ++it) {
        float lA[IBL], rA[JPL], lB[IBL], rB[JPL];
        load_col<IBL>(lA, lm, le, 0);
        load_col<JPL>(rA, rm, re, 0);
        int off = 0;
#pragma unroll 1
        for (int k = 0; k < CW; k += 2) {
            load_col<IBL>(lB, lm, le, off + CS);
            load_col<JPL>(rB, rm, re, off + CS);
            __builtin_amdgcn_sched_barrier(0);
            pair_step<IBL, JPL, MODE>(acc, lA, rA, s);
            __builtin_amdgcn_sched_barrier(0);
            load_col<IBL>(lA, lm, le, off + 2 * CS);          // (the last one reads past the slice: never consumed)
            load_col<JPL>(rA, rm, re, off + 2 * CS);
            __builtin_amdgcn_sched_barrier(0);
            pair_step<IBL, JPL, MODE>(acc, lB, rB, s);
            __builtin_amdgcn_sched_barrier(0);
            off += 2 * CS;
        }
    }
    float r = 0.f;
#pragma unroll
    for (int ii = 0; ii < IBL; ++ii)
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) r += acc[ii][jj];
    out[(size_t)blockIdx.x * blockDim.x + tid] = r;
}

template <int IBL, int JPL, int MODE, int NT, int MINW>
static void run(const char* name, float* d, int wg_per_cu, int CW, int iters) {
    const int nw = NT / 64;
    const int CS = ((Slots<IBL>::TOTAL > Slots<JPL>::TOTAL ? Slots<IBL>::TOTAL : Slots<JPL>::TOTAL) + 4);
    const int per_wave = 2 * (CW + 2) * CS;
    size_t lds = (size_t)nw * per_wave * 4;
    const size_t want = (size_t)(160 * 1024) / wg_per_cu - 1024;        // pad the allocation so that exactly wg_per_cu fit
    if (lds > want) { printf("%s: LDS %zu > %zu, skipped\n", name, lds, want); return; }
    if (wg_per_cu > 1 && lds < want * 3 / 4) lds = want * 3 / 4 + 1024;
    if (wg_per_cu == 1 && lds <= 80 * 1024) lds = 81 * 1024;
    auto fn = k_pair<IBL, JPL, MODE, NT, MINW>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * nw), lds, 0, d, iters, CW, CS, per_wave);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * nw), lds, 0, d, iters, CW, CS, per_wave);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipError_t e = hipGetLastError();
    const double cols = (double)iters * CW;                       // columns per wave
    const double ns_col = ms * 1e6 / cols;
    const double valu = 2.0 * IBL * JPL;
    const int waves_simd = nw * wg_per_cu / 4;
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.1f ns per column and wave = %6.0f cyc @2.4GHz  -> %.2f cyc per VALU instr and SIMD  (%s)\n",
           name, waves_simd, ms, ns_col, ns_col * 2.4, ns_col * 2.4 / (valu * waves_simd), hipGetErrorString(e));
}

int main() {
    float* d; hipMalloc(&d, (size_t)1024 * 1024 * 4 * 4);
    const int iters = 300;
    run<13, 13, 0, 512, 2>("13x13 abs-add, 8 waves x 1 WG", d, 1, 14, iters);
    run<13, 13, 1, 512, 2>("13x13 fma-sign, 8 waves x 1 WG", d, 1, 14, iters);
    run<13, 13, 0, 256, 2>("13x13 abs-add, 4 waves x 2 WG", d, 2, 14, iters);
    run<13, 13, 0, 256, 1>("13x13 abs-add, 4 waves x 1 WG", d, 1, 14, iters);
    run<7, 7, 0, 1024, 4>("7x7 abs-add, 16 waves x 1 WG", d, 1, 14, iters);
    run<7, 7, 1, 1024, 4>("7x7 fma-sign, 16 waves x 1 WG", d, 1, 14, iters);
    run<7, 7, 0, 512, 4>("7x7 abs-add, 8 waves x 2 WG", d, 2, 14, iters);
    run<7, 7, 0, 512, 2>("7x7 abs-add, 8 waves x 1 WG", d, 1, 14, iters);
    run<7, 7, 0, 768, 3>("7x7 abs-add, 12 waves x 1 WG", d, 1, 14, iters);
    run<8, 8, 0, 1024, 4>("8x8 abs-add, 16 waves x 1 WG", d, 1, 14, iters);
    run<8, 8, 0, 768, 3>("8x8 abs-add, 12 waves x 1 WG", d, 1, 14, iters);
    run<10, 10, 0, 768, 3>("10x10 abs-add, 12 waves x 1 WG", d, 1, 14, iters);
    run<16, 16, 0, 256, 1>("16x16 abs-add, 4 waves x 1 WG", d, 1, 14, iters);
    return 0;
}
