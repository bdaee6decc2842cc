// k2b_metrics.hip — pose-error metric of the evaluation path (MPJAE) on the GPU.
//
// One thread per pair of axis-angle rotations: geodesic angle in degrees between them, following the
// reference's float32 arithmetic step by step (keypoints2body/cli/eval.py: rotvec_to_rotmat :88-128,
// compute_angular_error_deg :131-140): R = I + a K + b K^2 with a = sin t / t, b = (1 - cos t) / t^2
// (Taylor series for t <= 1e-8), trace of the elementwise product of the two matrices,
// cos = (trace - 1) / 2 clipped to [-1 + 1e-6, 1 - 1e-6], arccos, degrees.
// Streaming kernel: 24 B in + 4 B out per pair, HBM-bound; precise sinf / cosf / acosf on purpose (the
// metric is compared with the reference's, not timed against a roofline that matters).
#include "k2b_internal.h"

namespace k2b {

namespace {

struct M9 { float m[9]; };

__device__ __forceinline__ M9 rotmat_of(float x, float y, float z) {
    const float t2 = x * x + y * y + z * z;
    const float t = sqrtf(t2);
    float a, b;
    if (t > 1e-8f) {
        a = sinf(t) / t;
        b = (1.0f - cosf(t)) / (t * t);
    } else {
        a = 1.0f - t2 / 6.0f + (t2 * t2) / 120.0f;
        b = 0.5f - t2 / 24.0f + (t2 * t2) / 720.0f;
    }
    const float xy = x * y, xz = x * z, yz = y * z, xx = x * x, yy = y * y, zz = z * z;
    M9 r;
    r.m[0] = 1.0f - b * (yy + zz); r.m[1] = b * xy - a * z;        r.m[2] = b * xz + a * y;
    r.m[3] = b * xy + a * z;        r.m[4] = 1.0f - b * (xx + zz); r.m[5] = b * yz - a * x;
    r.m[6] = b * xz - a * y;        r.m[7] = b * yz + a * x;        r.m[8] = 1.0f - b * (xx + yy);
    return r;
}

__global__ __launch_bounds__(256) void k2b_angular_error_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const M9 p = rotmat_of(pred[3 * i], pred[3 * i + 1], pred[3 * i + 2]);
    const M9 g = rotmat_of(gt[3 * i], gt[3 * i + 1], gt[3 * i + 2]);
    float tr = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) tr += p.m[k] * g.m[k];
    float c = (tr - 1.0f) * 0.5f;
    c = fminf(fmaxf(c, -1.0f + 1e-6f), 1.0f - 1e-6f);
    out[i] = acosf(c) * 57.29577951308232f;
}

}  // namespace

hipError_t launch_angular_error(const float* pred, const float* gt, float* out, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k2b_angular_error_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pred, gt, out, n);
    return hipGetLastError();
}

}  // namespace k2b
