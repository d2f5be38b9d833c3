// Probe of the LDS-DMA addressing rules k_gru_cm relies on (gfx950): does M0 reach LDS addresses >= 64 KiB, and does the
// instruction's immediate offset move the LDS destination as well as the global source?
//   hipcc --offload-arch=gfx950 -O2 -o profiles/bin/ubench_glds profiles/ubench_glds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* __restrict__ src, unsigned dst, float* __restrict__ out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = -1.f;
    __syncthreads();
    const unsigned voff = threadIdx.x * 16;
    unsigned keep;
    if (mode == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    __syncthreads();
    // report: first float of every KiB of LDS
    for (int i = threadIdx.x; i < 160; i += blockDim.x) out[i] = reinterpret_cast<float*>(lds)[i * 256];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 160 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned dsts[] = {0u, 32768u, 65536u, 98304u, 131072u, 150u * 1024u};
    for (int mode = 0; mode < 2; ++mode)
        for (unsigned d : dsts) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, d, out, mode);
            std::vector<float> o(160);
            hipMemcpy(o.data(), out, 160 * 4, hipMemcpyDeviceToHost);
            std::printf("mode %d (imm offset %d) M0 = %6u KiB %3u: landed at KiB", mode, mode ? 2048 : 0, d, d / 1024);
            for (int i = 0; i < 160; ++i)
                if (o[i] != -1.f) std::printf(" %d(first value %.0f)", i, o[i]);
            std::printf("\n");
        }
    return 0;
}
