// How long does the chip take to START and retire N tiny workgroups?  (pose set-up: 4096 one-wave workgroups, 9 KiB of LDS each)
//   hipcc -O3 --offload-arch=gfx950 tools/probe/dispatch_rate.hip -o tools/probe/dispatch_rate && tools/probe/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSB>
__global__ void tiny(float* out, int spin) {
    __shared__ float s[LDSB / 4 > 0 ? LDSB / 4 : 1];
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;     // a dependent chain of `spin` FMAs
    s[threadIdx.x % (LDSB / 4 > 0 ? LDSB / 4 : 1)] = v;
    __syncthreads();
    if (v == -1.f) out[blockIdx.x] = s[0];
}
template <int LDSB>
static void run(const char* name, int blocks, int threads, int spin, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(tiny<LDSB>, dim3(blocks), dim3(threads), 0, 0, d, spin);
    hipEventRecord(e0);
    const int n = 400;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny<LDSB>, dim3(blocks), dim3(threads), 0, 0, d, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks %5d x %4d threads, spin %5d: %.2f us per launch (back to back)\n", name, blocks, threads, spin, 1e3 * ms / n);
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    for (int spin : {0, 2000}) {
        run<0>("no LDS", 4096, 64, spin, d);
        run<9216>("9 KiB LDS", 4096, 64, spin, d);
        run<9216>("9 KiB LDS", 2048, 64, spin, d);
        run<9216>("9 KiB LDS", 1024, 64, spin, d);
        run<18432>("18 KiB LDS", 2048, 64, spin, d);
        run<36864>("36 KiB LDS", 1024, 256, spin, d);
        run<9216>("9 KiB LDS", 256, 64, spin, d);
    }
    return 0;
}
