// Probe: L2 -> LDS fill rate per CU as a function of the bytes in flight.  One 512-thread workgroup per CU; LW of its
// waves are loaders that issue 1 KiB global_load_lds_dwordx4 pieces back to back and keep at most DEPTH of them
// outstanding each (s_waitcnt vmcnt(DEPTH - 1) before every issue).  Source: a buffer of `foot` MiB cycled by all
// workgroups (small = L2-resident per XCD, 64 MiB = Infinity Cache, 2 GiB = HBM); destination: a 128 KiB LDS window.
//   fill_depth                 -> table of GB/s per CU and TB/s chip-wide
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DEPTH>
__global__ __launch_bounds__(512) void k(const unsigned char* src, size_t foot_bytes, int pieces, int loaders) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= loaders) return;
    // every XCD (block % 8) walks its own part of the footprint so that the L2s hold disjoint data
    size_t off = ((size_t)(blockIdx.x & 7) * 8191 + (size_t)(blockIdx.x >> 3) * 131 + wave * 17) * 1024 % foot_bytes;
    for (int i = 0; i < pieces; ++i) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        unsigned char* dst = lds + ((size_t)(wave * 16 + (i & 15)) * 1024);
        __builtin_amdgcn_global_load_lds((const void*)(src + off + lane * 16), (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        off += 8 * 1024;                       // the 8 waves of a workgroup interleave
        if (off >= foot_bytes) off -= foot_bytes;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int DEPTH> float run(const unsigned char* src, size_t foot, int pieces, int loaders) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), 128 * 1024, 0, src, foot, pieces, loaders);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(256), dim3(512), 128 * 1024, 0, src, foot, pieces, loaders);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const size_t big = (size_t)2 << 30;
    unsigned char* src; hipMalloc(&src, big); hipMemset(src, 1, big);
    const int pieces = 4000;
    for (size_t foot_mb : {8, 24, 128, 2048})
        for (int loaders : {4, 8}) {
            printf("footprint %4zu MiB, %d loader waves/CU:", foot_mb, loaders);
            const size_t foot = foot_mb << 20;
            float ms[5] = {run<2>(src, foot, pieces, loaders), run<4>(src, foot, pieces, loaders), run<8>(src, foot, pieces, loaders),
                           run<16>(src, foot, pieces, loaders), run<32>(src, foot, pieces, loaders)};
            const int d[5] = {2, 4, 8, 16, 32};
            for (int i = 0; i < 5; ++i) {
                const double gbs = (double)pieces * loaders * 1024 / (ms[i] * 1e-3) / 1e9;
                printf("  depth %2d (%3d KiB in flight): %5.1f GB/s/CU %5.2f TB/s |", d[i], d[i] * loaders, gbs, gbs * 256 / 1e3);
            }
            printf("\n");
        }
    return 0;
}
