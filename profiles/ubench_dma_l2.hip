// Sustained LDS-DMA (global_load_lds_dwordx4) throughput of one CU when the source streams from L2 (a buffer far larger than
// the 32 KB vector L1, shared by all workgroups -- the weight stream of k_gru_cm), against the same from an L1-resident source.
//   hipcc --offload-arch=gfx950 -O2 -o profiles/bin/ubench_dma_l2 profiles/ubench_dma_l2.hip
// One workgroup of four waves per CU; each wave issues 1-KiB pieces (its quarter of consecutive 4-KiB blocks), keeping at most
// DEPTH pieces in flight (s_waitcnt vmcnt), for `iters` passes over `span` bytes.  Reports bytes per cycle and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* __restrict__ src, unsigned span, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lane16 = (threadIdx.x & 63) * 16u;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nblk = span / 4096u;
    __syncthreads();
    const long long s = __builtin_amdgcn_s_memtime();
    unsigned slot = 0;
    for (int it = 0; it < iters; ++it)
        for (unsigned b = 0; b < nblk; ++b) {
            const unsigned char* p = src + (size_t)b * 4096u + wv * 1024u;
            const unsigned dst = (slot * 4u + wv) * 1024u;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(%4)"
                         : "=&s"(keep) : "v"(lane16), "s"(p), "s"(dst), "n"(DEPTH - 1) : "memory");
            slot = slot + 1 == 24 ? 0 : slot + 1;        // 24 x 4 KiB of LDS as landing area
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long e = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = e - s;
}

template <int DEPTH>
static void run(const unsigned char* src, unsigned span, int iters, int grid, long long* cyc, const char* what) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipLaunchKernelGGL(k<DEPTH>, dim3(grid), dim3(256), 96 * 1024, 0, src, span, iters, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (long long v : h) { sum += (double)v; mx = mx > (double)v ? mx : (double)v; }
    const double bytes = (double)span * iters;
    std::printf("%-44s depth %2d grid %4d  span %7u B: %.1f B/clk/CU (mean), %.1f (slowest workgroup)\n", what, DEPTH, grid, span, bytes / (sum / grid), bytes / mx);
}

int main() {
    unsigned char* src; long long* cyc;
    hipMalloc(&src, 8 << 20); hipMemset(src, 1, 8 << 20); hipMalloc(&cyc, 1024 * sizeof(long long));
    for (int grid : {1, 256}) {
        run<8>(src, 4096, 4000, grid, cyc, "L1-resident source (4 KiB)");
        run<8>(src, 640 * 1024, 30, grid, cyc, "L2 stream (640 KiB, as the GRU weights)");
        run<16>(src, 640 * 1024, 30, grid, cyc, "L2 stream (640 KiB)");
        run<32>(src, 640 * 1024, 30, grid, cyc, "L2 stream (640 KiB)");
        run<48>(src, 640 * 1024, 30, grid, cyc, "L2 stream (640 KiB)");
    }
    return 0;
}
