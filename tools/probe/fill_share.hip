// Probe: L2 -> LDS fill rate when the workgroups of an XCD read the SAME rows (the tile kernel's Pd slices) at the same
// time: identical order vs an order rotated per workgroup.  8 loader waves, 8 pieces per wave and "slice" (64 KiB), a
// barrier per slice, the slice's source = 64 KiB shared by all workgroups of the XCD (block % 8), advancing through a
// 16 MiB region.   mode 0: same piece order everywhere; mode 1: piece order rotated by workgroup; mode 2: private rows
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* src, int slices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3;
    const size_t region = (size_t)16 << 20;
    for (int s = 0; s < slices; ++s) {
        size_t base = (size_t)xcd * region + ((size_t)s * 65536) % region;
        if (MODE == 2) base = ((size_t)blockIdx.x * 977 + s) * 65536 % ((size_t)1 << 30);
        unsigned char* slot = lds + (s & 1) * 65536;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int p = wave + 8 * i;
            if (MODE == 1) p = (p + wg * 5) & 63;
            __builtin_amdgcn_global_load_lds((const void*)(src + base + (size_t)p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(slot + p * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
template <int MODE> void run(const unsigned char* src, const char* name) {
    const int slices = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 128 * 1024, 0, src, slices);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 128 * 1024, 0, src, slices);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)slices * 65536 / (ms * 1e-3) / 1e9;
    printf("%-34s %6.1f GB/s/CU %6.2f TB/s  (%.0f cycles per 64 KiB slice at 2.1 GHz)\n", name, gbs, gbs * 256 / 1e3, ms * 1e-3 / slices * 2.1e9);
}
int main() {
    unsigned char* src; (void)hipMalloc(&src, (size_t)1 << 30); (void)hipMemset(src, 1, (size_t)1 << 30);
    run<0>(src, "shared rows, same order");
    run<1>(src, "shared rows, rotated order");
    run<2>(src, "private rows (HBM stream)");
    return 0;
}
