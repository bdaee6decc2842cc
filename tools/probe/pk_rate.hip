// What a wave's own instruction stream sustains for scalar and packed fp32 FMAs (gfx950): tools/probe/pk_rate.hip
// build: hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip ; run: ./pk_rate
// 12 independent accumulators per lane (a 3x3-product-like stream), N rounds; one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int n, float s) {
    float a[12];
    for (int i = 0; i < 12; ++i) a[i] = threadIdx.x * 0.001f + i;
    f2 p[6];
    for (int i = 0; i < 6; ++i) p[i] = {a[2 * i], a[2 * i + 1]};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(p[(i + 1) % 6]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 12; ++i) r += a[i];
    for (int i = 0; i < 6; ++i) r += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main() {
    float* d; hipMalloc(&d, 1 << 22);
    const int n = 2000;
    for (int waves = 1; waves <= 2; ++waves)
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const dim3 grid(256), block(256 * waves);           // waves per SIMD = waves
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, n, 1.0001f);
                else hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, n, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fmas_per_lane = (double)n * 96;
            printf("%s, %d wave(s)/SIMD: %.3f ms for %d x 96 FMAs per lane -> %.2f ns per FMA-per-lane-per-wave (%.2f cyc at 2.4 GHz)\n",
                   mode ? "v_pk_fma_f32 (48 instr)" : "v_fma_f32 (96 instr)", waves, ms, n, ms * 1e6 / fmas_per_lane, ms * 1e6 / fmas_per_lane * 2.4);
        }
    return 0;
}
