// k2b_precompute.hip — the dense J x V joint-regressor contraction on the matrix cores.
//
// smplx evaluates J = J_regressor . (v_template + shapedirs . beta) every forward
// (oracle/smpl_torch.py step 2).  J is linear in beta, so the engine contracts the
// regressor with v_template and with shapedirs ONCE per model:
//     J_template[J][3]     = J_regressor[J][V] . v_template[V][3]
//     J_dirs[J][3*NB]      = J_regressor[J][V] . shapedirs[V][3*NB]
// and the per-iteration kernel only evaluates J_template + J_dirs . beta.
//
// This is the one GEMM-shaped piece of the path, so it runs on MFMA:
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain), one wave per 16x16 output
// tile and K-slice of V, partial tiles summed in a fixed order by a second kernel
// (bitwise reproducible, no atomics).
#include "k2b_internal.h"

namespace k2b {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = (J tiles, N tiles, splits); block = 64
__global__ __launch_bounds__(64) void k2b_jreg_mfma_kernel(const float* __restrict__ jreg, const float* __restrict__ rhs,
                                                           float* __restrict__ partial, int J, int V, int N,
                                                           int chunk) {
    const int lane = threadIdx.x;
    const int row = blockIdx.x * 16 + (lane & 15);   // A operand: A[row][k = lane >> 4]
    const int col = blockIdx.y * 16 + (lane & 15);   // B operand: B[k = lane >> 4][col]
    const int kq = lane >> 4;
    const int k0 = blockIdx.z * chunk;
    const int k1 = min(k0 + chunk, V);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; k += 4) {
        const int kk = k + kq;
        const float av = (row < J && kk < k1) ? jreg[(size_t)row * V + kk] : 0.f;
        const float bv = (col < N && kk < k1) ? rhs[(size_t)kk * N + col] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    // C/D map: col = lane & 15, row = (lane >> 4) * 4 + i
    const int Jp = gridDim.x * 16, Np = gridDim.y * 16;
    float* out = partial + (size_t)blockIdx.z * Jp * Np;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = blockIdx.x * 16 + (lane >> 4) * 4 + i;
        out[(size_t)r * Np + blockIdx.y * 16 + (lane & 15)] = acc[i];
    }
}

__global__ void k2b_jreg_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int J, int N,
                                       int Jp, int Np, int splits) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= J * N) return;
    const int r = idx / N, c = idx % N;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(size_t)z * Jp * Np + (size_t)r * Np + c];
    out[idx] = s;
}

hipError_t launch_jreg_contract(const float* j_regressor, const float* rhs, float* out, int J, int V, int N,
                                float* partial_ws, int num_splits, hipStream_t stream) {
    const int jt = (J + 15) / 16, nt = (N + 15) / 16;
    int chunk = (V + num_splits - 1) / num_splits;
    chunk = (chunk + 3) / 4 * 4;
    hipLaunchKernelGGL(k2b_jreg_mfma_kernel, dim3(jt, nt, num_splits), dim3(64), 0, stream, j_regressor, rhs,
                       partial_ws, J, V, N, chunk);
    hipLaunchKernelGGL(k2b_jreg_reduce_kernel, dim3((J * N + 255) / 256), dim3(256), 0, stream, partial_ws, out, J, N,
                       jt * 16, nt * 16, num_splits);
    return hipGetLastError();
}

}  // namespace k2b
