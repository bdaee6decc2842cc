// LDS-free cross-lane primitives shared by the fit kernels (gfx950 / CDNA4, wave64): DPP row operations,
// v_permlane16/32_swap exchanges, reductions and scans built from them.
#pragma once
#include <hip/hip_runtime.h>

namespace k2b {

// Inclusive prefix sum inside each 32-lane half with DPP (no LDS traffic): Hillis-Steele inside each
// row of 16 (row_shr 1, 2, 4, 8; out-of-row sources read 0), then row_bcast:15 into rows 1 and 3.
// Accumulated in DOUBLE: the subtree sums are differences of two prefixes, and in fp32 their
// absolute error (eps x the largest prefix) was visible after Adam's per-parameter normalisation
// (parity 3e-6 -> up to 9e-5); in double the differences are exact to fp32.
__device__ __forceinline__ double half_wave_inclusive_scan(float v) {
    double s = (double)v;
#define K2B_DPP_ADD64(ctrl, row_mask)                                                                   \
    {                                                                                                   \
        const long long bits = __builtin_bit_cast(long long, s);                                        \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), ctrl, row_mask, 0xf, true); \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), ctrl, row_mask, 0xf, true);    \
        s += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);                      \
    }
    K2B_DPP_ADD64(0x111, 0xf);   // row_shr:1
    K2B_DPP_ADD64(0x112, 0xf);   // row_shr:2
    K2B_DPP_ADD64(0x114, 0xf);   // row_shr:4
    K2B_DPP_ADD64(0x118, 0xf);   // row_shr:8
    K2B_DPP_ADD64(0x142, 0xa);   // row_bcast:15 -> rows 1, 3 (lanes 16..31 and 48..63)
#undef K2B_DPP_ADD64
    return s;
}

__device__ __forceinline__ double bperm64(int byte_addr, double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(bits & 0xffffffffll));
    const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, (int)(bits >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float bperm(int byte_addr, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

// ---- LDS-free cross-lane exchanges (gfx950) ------------------------------------------------------
// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b`.
// (Inline asm: the clang builtin of ROCm 7.2 returns the updated first register in BOTH result
//  elements - tools/probe/lanes.hip.  The s_nop covers the two wait states between a VALU write of an
//  operand and the swap reading it, which hipcc does not insert inside asm.)
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// v_permlane16_swap: the odd 16-lane rows of `a` trade places with the even rows of `b`.
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp<0xB1>(v); }    // quad_perm [1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp<0x4E>(v); }    // quad_perm [2,3,0,1]
__device__ __forceinline__ float lane_xor8(float v) { return dpp<0x128>(v); }   // row_ror:8
__device__ __forceinline__ float lane_xor4(float v) {                           // row_shr:4 into banks 1,3 ; row_shl:4 into banks 0,2
    const int x = __builtin_bit_cast(int, v);
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xa, true);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, x, 0x104, 0xf, 0x5, true));
}
// sum of v over the lane and its partner lane ^ 32 / ^ 16 (every lane gets the pair sum)
__device__ __forceinline__ float pair_sum32(float v) { float a = v, b = v; swap32(a, b); return a + b; }
__device__ __forceinline__ float pair_sum16(float v) { float a = v, b = v; swap16(a, b); return a + b; }

// v_min_f32 as asm: fminf() costs a canonicalising v_max_f32 per operand on top; NaNs lose against numbers either way
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// minimum over the lane's group of eight {l ^ 1, l ^ 2, l ^ 4}: quad_perm twice, then the half-row mirror (the s_nop covers the
// two wait states between a VALU write and a DPP read of the same register, which hipcc does not insert inside asm)
__device__ __forceinline__ float group8_min(float v) {
    float r;
    asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ float wave_sum_fast(float v) {
    v = pair_sum32(v);
    v = pair_sum16(v);
    v += lane_xor8(v);
    v += lane_xor4(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}

// 16 per-lane values -> lane l ends with the wave-wide sum of v[(l>>2)&15].
__device__ __forceinline__ float butterfly16_sum(const float (&v)[16], int lane) {
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = v[i], b = v[8 + i];
        swap32(a, b);
        w8[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = w8[i], b = w8[4 + i];
        swap16(a, b);
        w4[i] = a + b;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float s0 = w4[i] + lane_xor8(w4[i]), s1 = w4[2 + i] + lane_xor8(w4[2 + i]);
        w2[i] = (lane & 8) ? s1 : s0;
    }
    const float t0 = w2[0] + lane_xor4(w2[0]), t1 = w2[1] + lane_xor4(w2[1]);
    float r = (lane & 4) ? t1 : t0;
    r += lane_xor2(r);
    r += lane_xor1(r);
    return r;
}

// 4 per-lane values -> every lane of the 16-lane row r ends with the wave-wide sum of v[r]
__device__ __forceinline__ float butterfly4_sum(const float (&v)[4]) {
    float w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float a = v[i], b = v[2 + i];
        swap32(a, b);
        w[i] = a + b;                // lanes < 32: v[i] over the lane pair; lanes >= 32: v[2 + i]
    }
    float a = w[0], b = w[1];
    swap16(a, b);
    float r = a + b;                 // even rows: w[0]; odd rows: w[1]
    r += lane_xor8(r);
    r += lane_xor4(r);
    r += lane_xor2(r);
    r += lane_xor1(r);
    return r;
}

// inclusive prefix sum over all 64 lanes, in double: the two half-wave scans, then the lower half's total (lane 31) added to the
// upper half with one more DPP step (row_bcast:31 into rows 2, 3)
__device__ __forceinline__ double wave_inclusive_scan(float v, int lane) {
    (void)lane;
    double s = half_wave_inclusive_scan(v);
    const long long bits = __builtin_bit_cast(long long, s);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), 0x143, 0xc, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), 0x143, 0xc, 0xf, true);
    s += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    return s;
}

}  // namespace k2b
