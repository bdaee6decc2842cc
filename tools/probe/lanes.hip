// Lane-mapping probe for the LDS-free exchange primitives used by k2b_fit.hip (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ float lane_xor4(float v) {
    const int x = __builtin_bit_cast(int, v);
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xa, true);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, x, 0x104, 0xf, 0x5, true));
}
__global__ void k(float* out) {
    const int l = threadIdx.x;
    float a = (float)l, b = 100.f + l;
    float a1 = a, b1 = b;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a1), "+v"(b1));
    out[0 * 64 + l] = a1;
    out[1 * 64 + l] = b1;
    float a2 = a, b2 = b;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a2), "+v"(b2));
    out[2 * 64 + l] = a2;
    out[3 * 64 + l] = b2;
    out[4 * 64 + l] = dpp<0x128>(a);   // row_ror:8
    out[5 * 64 + l] = lane_xor4(a);
    out[6 * 64 + l] = dpp<0x4E>(a);    // quad_perm [2,3,0,1]
    out[7 * 64 + l] = dpp<0xB1>(a);    // quad_perm [1,0,3,2]
}
int main() {
    float* d; hipMalloc(&d, 8 * 64 * 4);
    k<<<1, 64>>>(d);
    float h[8 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[8] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1", "row_ror:8", "xor4", "quad[2,3,0,1]", "quad[1,0,3,2]"};
    for (int i = 0; i < 8; ++i) { printf("%-14s", names[i]); for (int l = 0; l < 64; ++l) printf(" %g", h[i * 64 + l]); printf("\n"); }
    return 0;
}
