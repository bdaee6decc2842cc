// k2b_lbs.hip — full SMPL forward (pose set-up + vertex skinning) for gfx950.
//
// Replaces the body-model call `self.smpl(**kwargs)` of the reference
// (keypoints2body/core/fitters/world_space.py:34,192,278), i.e. smplx's SMPL.forward /
// lbs(): shape blend, pose-corrective blend, kinematic chain, linear blend skinning,
// vertex-selected extra joints, translation.  CPU twin: oracle/smpl_torch.py.
//
// The two contractions of LBS are GEMMs over (frames x vertices):
//   v_posed[f][v][c] = sum_k X[f][k] * Pd[k][v][c]      K = 9(J-1) + NB + 2   (224 for SMPL)
//        X = [vec(R_1..R_{J-1} - I) | beta | 1 | 1],  Pd = [posedirs ; shapedirs ; v_template ; residual]
//   T[f][v][e]       = sum_j W[v][j] * A[f][j][e]       K = J (24), e = 12 entries of the 3x4 transform
// and the result is v[f][v] = T[f][v] [v_posed; 1] + transl[f].
//
// Both run on the matrix cores.  fp32-input MFMA issues at the fp32 VALU rate, so operands are
// split into two f16 terms (x = hi + lo, ~22 mantissa bits) and each product takes three
// v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate): 3/16 of the fp32 MFMA time
// at fp32-level accuracy (|error| ~ 1e-6 m on metre-scale vertices; tests/test_gpu_parity.py).
//
// Tiling: one wave owns a 32-frame x 32-vertex tile.  MFMA orientation D[frame][vertex]: the
// accumulator of lane l holds 16 frames of ONE vertex (column l & 31), so the skinning epilogue
// (T applied to v_posed) is a per-lane computation with no cross-lane traffic.  Operand
// fragments are stored k-step-major ([ks][row][16 halfs]) so that a wave's fragment load is one
// contiguous 1 KiB read.
#include <hip/hip_fp16.h>

#include "k2b_internal.h"

namespace k2b {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// ---------------------------------------------------------------------------------------------
// Pose set-up: one 64-lane workgroup per frame, lane j = joint j.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k2b_pose_setup_kernel(const PoseArgs a) {
    __shared__ float sR[kMaxJoints][9];
    __shared__ float sd[kMaxJoints][3];
    __shared__ int spar[kMaxJoints];
    const int f = blockIdx.x;
    const int j = threadIdx.x;
    const int J = a.num_joints, NB = a.num_betas;
    const bool act = j < J;
    const int bp = a.frames_padded;

    Vec3 th = {0.f, 0.f, 0.f};
    Vec3 Jj = {0.f, 0.f, 0.f}, Jp = {0.f, 0.f, 0.f};
    int par = -1;
    if (act) {
        const float* src = j == 0 ? a.go + (size_t)f * 3 : a.bp + (size_t)f * 3 * (J - 1) + 3 * (j - 1);
        th = {src[0], src[1], src[2]};
        par = a.parents[j];
        float e[3], p[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < 3; ++c) {
            float s = a.j_template[j * 3 + c];
            for (int k = 0; k < NB; ++k) s += a.j_dirs[(j * 3 + c) * NB + k] * a.be[(size_t)f * NB + k];
            e[c] = s;
            if (par >= 0) {
                float q = a.j_template[par * 3 + c];
                for (int k = 0; k < NB; ++k) q += a.j_dirs[(par * 3 + c) * NB + k] * a.be[(size_t)f * NB + k];
                p[c] = q;
            }
        }
        Jj = {e[0], e[1], e[2]};
        Jp = {p[0], p[1], p[2]};
    }
    const Rodrigues rod = rodrigues_fwd(th);
    if (act) {
        for (int i = 0; i < 9; ++i) sR[j][i] = rod.R.m[i];
        const Vec3 d = Jj - Jp;
        sd[j][0] = d.x; sd[j][1] = d.y; sd[j][2] = d.z;
        spar[j] = par;
    }
    __syncthreads();

    // X operand of the vertex GEMM, f16 hi/lo, fragment layout [ks][frame][16]
    auto put_x = [&](int k, float x) {
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        const size_t o = ((size_t)(k >> 4) * bp + f) * 16 + (k & 15);
        a.xh[o] = hi;
        a.xl[o] = lo;
    };
    const int P = 9 * (J - 1);
    if (act && j > 0) {
        for (int i = 0; i < 9; ++i) put_x((j - 1) * 9 + i, rod.R.m[i] - ((i % 4 == 0) ? 1.f : 0.f));
    }
    for (int k = P + j; k < a.k_steps_x * 16; k += 64) {   // betas, the two constant-1 features, zero padding
        const float x = k < P + NB ? a.be[(size_t)f * NB + (k - P)] : (k < P + NB + 2 ? 1.f : 0.f);
        put_x(k, x);
    }
    // zero rows of the A operand for the padded joints J .. 16*k_steps_a - 1
    for (int idx = j; idx < (a.k_steps_a * 16 - J) * 12; idx += 64) {
        const int jj = J + idx / 12, e = idx % 12;
        const size_t o = (((size_t)e * a.k_steps_a + (jj >> 4)) * bp + f) * 16 + (jj & 15);
        a.ah[o] = (_Float16)0.f;
        a.al[o] = (_Float16)0.f;
    }
    if (!act) return;

    // global transform: compose towards the root
    Mat3 Rg = rod.R;
    Vec3 pg = {sd[j][0], sd[j][1], sd[j][2]};
    for (int anc = par; anc >= 0; anc = spar[anc]) {
        Mat3 Ra;
        for (int i = 0; i < 9; ++i) Ra.m[i] = sR[anc][i];
        const Vec3 da = {sd[anc][0], sd[anc][1], sd[anc][2]};
        pg = mul(Ra, pg) + da;
        Rg = mul(Ra, Rg);
    }
    // A_j = [Rg | pg - Rg J_j], f16 hi/lo, fragment layout [e][ks][frame][16] with k = joint
    const Vec3 rj = mul(Rg, Jj);
    const float At[12] = {Rg.m[0], Rg.m[1], Rg.m[2], pg.x - rj.x, Rg.m[3], Rg.m[4], Rg.m[5], pg.y - rj.y,
                          Rg.m[6], Rg.m[7], Rg.m[8], pg.z - rj.z};
    for (int e = 0; e < 12; ++e) {
        _Float16 hi, lo;
        split_f16(At[e], hi, lo);
        const size_t o = (((size_t)e * a.k_steps_a + (j >> 4)) * bp + f) * 16 + (j & 15);
        a.ah[o] = hi;
        a.al[o] = lo;
    }
    if (a.joints_out) {
        float* o = a.joints_out + ((size_t)f * a.num_out_joints + j) * 3;
        const float tx = a.tr ? a.tr[(size_t)f * 3] : 0.f, ty = a.tr ? a.tr[(size_t)f * 3 + 1] : 0.f,
                    tz = a.tr ? a.tr[(size_t)f * 3 + 2] : 0.f;
        o[0] = pg.x + tx; o[1] = pg.y + ty; o[2] = pg.z + tz;
    }
}

// ---------------------------------------------------------------------------------------------
// Vertex kernel: one wave per (32 frames x 32 vertices) tile, 4 waves per workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int kChunkPairs = 8;   // frame-tile pairs (64 frames each) per L2-resident chunk

__device__ __forceinline__ half8 ld_frag(const _Float16* base, size_t row_index, int h) {
    return *reinterpret_cast<const half8*>(base + row_index * 16 + 8 * h);
}

__global__ __launch_bounds__(256, 2) void k2b_lbs_mfma_kernel(const SkinArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    // workgroup = 2 frame tiles x 2 vertex tiles.  Rasterisation for the 8 per-XCD L2s (blocks are
    // dealt round-robin over the XCDs, so b % 8 labels blocks that share an L2): each label owns
    // every 8th vertex-tile pair, and inside a chunk of kChunkPairs frame-tile pairs the frame
    // index runs fastest - the 172 KB of vertex operands of one pair stay L2-resident while the
    // chunk's frames sweep over them, and the chunk's per-frame operands (1.2 MB) stay resident
    // while the vertex pairs advance.  Placement affects speed only.
    const int vpairs = (a.v_tiles + 1) / 2, fpairs = (a.f_tiles + 1) / 2;
    const int vl = (vpairs + 7) / 8;                       // vertex-tile pairs per label
    const int label = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int chunk = i / (vl * kChunkPairs), rem = i % (vl * kChunkPairs);
    const int vpair = (rem / kChunkPairs) * 8 + label, fpair = chunk * kChunkPairs + rem % kChunkPairs;
    if (vpair >= vpairs || fpair >= fpairs) return;
    const int vt = vpair * 2 + (wave & 1);
    const int ft = fpair * 2 + (wave >> 1);
    if (vt >= a.v_tiles || ft >= a.f_tiles) return;
    const int vp = a.v_tiles * 32;          // padded vertex count of this vertex set
    const int bp = a.frames_padded;
    const int KS = a.k_steps_x, KA = a.k_steps_a;
    const size_t frow = (size_t)ft * 32 + col;      // A-operand row (frame) of this lane
    const size_t vrow = (size_t)vt * 32 + col;      // B-operand column (vertex) of this lane

    // ---- 1. v_posed * 256 = X . Pd  (three coordinates) -------------------------------------------
    floatx16 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    {
        half8 xh = ld_frag(a.xh, frow, h), xl = ld_frag(a.xl, frow, h);
        half8 ph[3], pl[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ph[c] = ld_frag(a.pdh, (size_t)c * vp + vrow, h);
            pl[c] = ld_frag(a.pdl, (size_t)c * vp + vrow, h);
        }
        for (int ks = 0; ks < KS; ++ks) {
            half8 nxh = xh, nxl = xl, nph[3], npl[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { nph[c] = ph[c]; npl[c] = pl[c]; }
            if (ks + 1 < KS) {
                nxh = ld_frag(a.xh, (size_t)(ks + 1) * bp + frow, h);
                nxl = ld_frag(a.xl, (size_t)(ks + 1) * bp + frow, h);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    nph[c] = ld_frag(a.pdh, ((size_t)(ks + 1) * 3 + c) * vp + vrow, h);
                    npl[c] = ld_frag(a.pdl, ((size_t)(ks + 1) * 3 + c) * vp + vrow, h);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, ph[c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, pl[c], acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, ph[c], acc[c], 0, 0, 0);
            }
            xh = nxh; xl = nxl;
#pragma unroll
            for (int c = 0; c < 3; ++c) { ph[c] = nph[c]; pl[c] = npl[c]; }
        }
    }
    const float inv_scale = 1.0f / kPdScale;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] *= inv_scale;

    // ---- 2. per output coordinate r: T[4r..4r+3] = A . W^T, then out_r = T . [v_posed; 1] -----------
    const int v = vt * 32 + col;                       // index inside this vertex set
    const bool v_ok = v < a.num_out;
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
        floatx16 t4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) t4[e][i] = 0.f;
        for (int ks = 0; ks < KA; ++ks) {
            const half8 wh = ld_frag(a.wth, (size_t)ks * vp + vrow, h), wl = ld_frag(a.wtl, (size_t)ks * vp + vrow, h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t row = ((size_t)(4 * r + e) * KA + ks) * bp + frow;
                const half8 ah = ld_frag(a.ah, row, h), al = ld_frag(a.al, row, h);
                t4[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, t4[e], 0, 0, 0);
                t4[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, t4[e], 0, 0, 0);
                t4[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, t4[e], 0, 0, 0);
            }
        }
        // C/D map of 32x32 MFMA: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int f = ft * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (v_ok && f < a.num_frames) {
                float o = t4[0][i] * acc[0][i] + t4[1][i] * acc[1][i] + t4[2][i] * acc[2][i] + t4[3][i];
                if (a.tr) o += a.tr[(size_t)f * 3 + r];
                a.out[((size_t)f * a.out_stride + a.out_row0 + v) * 3 + r] = o;
            }
        }
    }
}

// joints J..J+E-1 := vertices[extra ids] (when the full mesh has just been produced)
__global__ void k2b_gather_joints_kernel(const float* __restrict__ verts, const int* __restrict__ ids, float* joints,
                                         int num_frames, int V, int J, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_frames * E) return;
    const int f = i / E, e = i % E;
    const float* s = verts + ((size_t)f * V + ids[e]) * 3;
    float* d = joints + ((size_t)f * (J + E) + J + e) * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

int lbs_frames_padded(int num_frames) { return (num_frames + 31) / 32 * 32; }

hipError_t launch_pose_setup(const PoseArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k2b_pose_setup_kernel, dim3(a.num_frames), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_skin(const SkinArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0 || a.num_out <= 0) return hipSuccess;
    const int vpairs = (a.v_tiles + 1) / 2, fpairs = (a.f_tiles + 1) / 2;
    const int vl = (vpairs + 7) / 8, chunks = (fpairs + kChunkPairs - 1) / kChunkPairs;
    const dim3 grid(8 * vl * kChunkPairs * chunks);
    hipLaunchKernelGGL(k2b_lbs_mfma_kernel, grid, dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_gather_joints(const float* verts, const int* ids, float* joints, int num_frames, int V, int J, int E,
                                hipStream_t stream) {
    if (num_frames <= 0 || E <= 0) return hipSuccess;
    const int n = num_frames * E;
    hipLaunchKernelGGL(k2b_gather_joints_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, verts, ids, joints,
                       num_frames, V, J, E);
    return hipGetLastError();
}

}  // namespace k2b
