// Device-side helpers shared by the fit and LBS kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace k2b {

constexpr int kWave = 64;

struct Mat3 {
    float m[9];  // row-major
};
struct Vec3 {
    float x, y, z;
};

__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Vec3 mul(const Mat3& A, Vec3 v) {
    return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
            A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
__device__ __forceinline__ Vec3 mulT(const Mat3& A, Vec3 v) {  // A^T v
    return {A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z,
            A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z};
}
__device__ __forceinline__ Mat3 mul(const Mat3& A, const Mat3& B) {
    Mat3 C;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            C.m[3 * r + c] = A.m[3 * r] * B.m[c] + A.m[3 * r + 1] * B.m[3 + c] + A.m[3 * r + 2] * B.m[6 + c];
    return C;
}

// Wave-local ordering point for LDS traffic exchanged between lanes of ONE wave: DS
// instructions of a wave execute in issue order, so only the compiler must be stopped
// from moving a read above another lane's write.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ float read_lane(float v, int lane) {  // lane must be wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32): the IEEE-exact division and
// sqrt sequences cost ~10 dependent instructions each, and this kernel is latency-bound.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// sin and cos of a non-negative angle with ONE shared range reduction (k = round(a 2/pi), two-term
// Cody-Waite) and the fdlibm float kernels on [-pi/4, pi/4]; ~1-2 ulp.  Rotation angles are O(1);
// beyond 1e3 rad the two-term reduction loses accuracy and the library functions take over.
__device__ __forceinline__ void sincos_small(float a, float& s, float& c) {
    if (a > 1.0e3f) { s = sinf(a); c = cosf(a); return; }
    const float kf = rintf(a * 0.63661977236758134308f);
    const int k = (int)kf;
    float r = fmaf(kf, -1.57079625129699707031f, a);      // pi/2 high part
    r = fmaf(kf, -7.54978941586159635335e-08f, r);        // pi/2 low part
    const float z = r * r;
    const float ps = r + r * z * (-0.166666666416265235595f + z * (0.0083333293858894631756f +
                     z * (-0.000198393348360966317347f + z * 0.0000027183114939898219064f)));
    const float pc = 1.0f + z * (-0.499999997251031003120f + z * (0.0416666233237390631894f +
                     z * (-0.00138867637746099294692f + z * 0.0000243904487962774090654f)));
    const float s0 = (k & 1) ? pc : ps, c0 = (k & 1) ? ps : pc;
    s = (k & 2) ? -s0 : s0;
    c = ((k + 1) & 2) ? -c0 : c0;
}

// Rodrigues' formula exactly as smplx.lbs.batch_rodrigues evaluates it
// (oracle/smpl_torch.py::batch_rodrigues): angle = ||theta + 1e-8||, u = theta / angle,
// R = I + sin(angle) K + (1 - cos(angle)) K K with K = skew(u).
struct Rodrigues {
    Mat3 R;
    Vec3 u;
    float angle, inv_angle, s, c;
};

__device__ __forceinline__ Rodrigues rodrigues_fwd(Vec3 th) {
    Rodrigues o;
    const float ex = th.x + 1e-8f, ey = th.y + 1e-8f, ez = th.z + 1e-8f;
    o.angle = fast_sqrt(ex * ex + ey * ey + ez * ez);
    o.inv_angle = fast_rcp(o.angle);
    o.u = {th.x * o.inv_angle, th.y * o.inv_angle, th.z * o.inv_angle};
    sincos_small(o.angle, o.s, o.c);
    const float omc = 1.0f - o.c;
    const float ux = o.u.x, uy = o.u.y, uz = o.u.z;
    // K K = u u^T - (u.u) I restricted to what the matrix product gives
    const float xx = ux * ux, yy = uy * uy, zz = uz * uz, xy = ux * uy, xz = ux * uz, yz = uy * uz;
    o.R.m[0] = 1.0f + omc * (-(yy + zz));
    o.R.m[1] = o.s * (-uz) + omc * xy;
    o.R.m[2] = o.s * uy + omc * xz;
    o.R.m[3] = o.s * uz + omc * xy;
    o.R.m[4] = 1.0f + omc * (-(xx + zz));
    o.R.m[5] = o.s * (-ux) + omc * yz;
    o.R.m[6] = o.s * (-uy) + omc * xz;
    o.R.m[7] = o.s * ux + omc * yz;
    o.R.m[8] = 1.0f + omc * (-(xx + yy));
    return o;
}

}  // namespace k2b
