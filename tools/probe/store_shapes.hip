// Probe: what does the per-CU store path sustain for different store shapes?  One 512-thread workgroup per CU,
// every wave issues `iters` store instructions of one shape back to back (no loads, no compute), to an output of
// `rows` x rowbytes laid out like the LBS output ([frame][V][3] floats, V = 6890).
//   shape 0: dwordx3, 16 lanes contiguous (192 B) x 4 frame rows     (the tile kernel's store)
//   shape 1: dwordx3, 32 lanes contiguous (384 B) x 2 frame rows     (the 128 x 64 kernel's store)
//   shape 2: dwordx4, 64 lanes contiguous (1024 B of one frame row)
//   shape 3: dwordx4, 24 + 24 + 16 lanes: runs of 384 / 384 / 256 B in three frame rows
//   shape 4: dword,   64 lanes contiguous (256 B)
//   shape 5: dwordx4, each lane its own row (row-per-lane)
// footprint: `frames` distinct frame rows are cycled (small = L2-resident, large = HBM stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int V = 6890;
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int frames, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row = (size_t)V * 3;                         // floats per frame
    typedef float f3 __attribute__((ext_vector_type(3)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 val = {(float)lane, (float)wave, (float)blockIdx.x, 1.f};
    const f3 val3 = {(float)lane, (float)wave, (float)blockIdx.x};
    int f = (blockIdx.x * 8 + wave) * 4 % frames;
    int v = (blockIdx.x * 37 + wave * 5) % 40 * 128;           // column block inside the row
    for (int it = 0; it < iters; ++it) {
        float* p;
        if (SHAPE == 0) {
            p = out + (size_t)((f + (lane >> 4)) % frames) * row + (size_t)(v + (lane & 15)) * 3;
            asm volatile("global_store_dwordx3 %0, %1, off" ::"v"(p), "v"(val3) : "memory");
        } else if (SHAPE == 1) {
            p = out + (size_t)((f + (lane >> 5)) % frames) * row + (size_t)(v + (lane & 31)) * 3;
            asm volatile("global_store_dwordx3 %0, %1, off" ::"v"(p), "v"(val3) : "memory");
        } else if (SHAPE == 2) {
            p = out + (size_t)f * row + (size_t)v * 3 + lane * 4;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(val) : "memory");
        } else if (SHAPE == 3) {
            const int r = lane < 24 ? 0 : lane < 48 ? 1 : 2, c = lane < 24 ? lane : lane < 48 ? lane - 24 : lane - 48;
            p = out + (size_t)((f + r) % frames) * row + (size_t)v * 3 + c * 4;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(val) : "memory");
        } else if (SHAPE == 4) {
            p = out + (size_t)f * row + (size_t)v * 3 + lane;
            asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(val.x) : "memory");
        } else {
            p = out + (size_t)((f + lane) % frames) * row + (size_t)v * 3;
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(val) : "memory");
        }
        f += 4; if (f >= frames) f -= frames;
        v += 128; if (v >= 40 * 128) v -= 40 * 128;
    }
}
int main() {
    const int frames_big = 8192, cus = 256;
    float* out;
    hipMalloc(&out, (size_t)frames_big * V * 3 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int bytes[6] = {768, 768, 1024, 1024, 256, 1024};
    for (int frames : {32, 8192})
        for (int shape = 0; shape < 6; ++shape) {
            const int iters = 2000;
            auto launch = [&]() {
                switch (shape) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(512), 0, 0, out, frames, iters); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); launch(); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double total = (double)cus * 8 * iters * bytes[shape];
            const double ns_per_instr_cu = ms * 1e6 / (8.0 * iters);
            printf("frames %5d shape %d: %.3f ms  %.2f TB/s  %.1f ns per store instruction per CU (%.1f B/ns/CU)\n", frames, shape, ms,
                   total / ms / 1e9, ns_per_instr_cu, bytes[shape] / ns_per_instr_cu);
        }
    return 0;
}
