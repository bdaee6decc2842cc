// k2b_lbs.hip — full SMPL forward (pose set-up + vertex skinning) for gfx950.
//
// Replaces the body-model call `self.smpl(**kwargs)` of the reference
// (keypoints2body/core/fitters/world_space.py:34,192,278), i.e. smplx's SMPL.forward /
// lbs(): shape blend, pose-corrective blend, kinematic chain, linear blend skinning,
// vertex-selected extra joints, translation.  CPU twin: oracle/smpl_torch.py.
//
// Two kernels per call:
//   k2b_pose_setup_kernel  one 64-lane workgroup per frame, lane j = joint j: Rodrigues,
//                          J(beta), global transform by walking the ancestor chain from LDS,
//                          relative transforms A_j and the pose feature, both stored
//                          frame-minor ([.][Bpad]) so the skinning kernel can fetch a group
//                          of frames with one scalar load.
//   k2b_skin_kernel        thread = vertex, kSkinFrames frames per thread: every posedirs /
//                          shapedirs / weight element fetched once per frame group and reused
//                          from registers; per-frame operands arrive through the scalar path.
#include "k2b_internal.h"

namespace k2b {

constexpr int kSkinFrames = 8;

__global__ __launch_bounds__(64) void k2b_pose_setup_kernel(const PoseArgs a, int bpad) {
    __shared__ float sR[kMaxJoints][9];
    __shared__ float sd[kMaxJoints][3];
    __shared__ int spar[kMaxJoints];
    const int f = blockIdx.x;
    const int j = threadIdx.x;
    const int J = a.num_joints, NB = a.num_betas;
    const bool act = j < J;

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
    // A_j = [Rg | pg - Rg J_j]
    const Vec3 rj = mul(Rg, Jj);
    const float At[12] = {Rg.m[0], Rg.m[1], Rg.m[2], pg.x - rj.x, Rg.m[3], Rg.m[4], Rg.m[5], pg.y - rj.y,
                          Rg.m[6], Rg.m[7], Rg.m[8], pg.z - rj.z};
    for (int e = 0; e < 12; ++e) a.A[((size_t)j * 12 + e) * bpad + f] = At[e];
    if (j > 0) {
        for (int i = 0; i < 9; ++i)
            a.feat[((size_t)(j - 1) * 9 + i) * bpad + f] = rod.R.m[i] - ((i % 4 == 0) ? 1.f : 0.f);
    }
    if (a.joints_out) {
        float* o = a.joints_out + ((size_t)f * a.num_out_joints + j) * 3;
        const float tx = a.tr ? a.tr[(size_t)f * 3] : 0.f, ty = a.tr ? a.tr[(size_t)f * 3 + 1] : 0.f,
                    tz = a.tr ? a.tr[(size_t)f * 3 + 2] : 0.f;
        o[0] = pg.x + tx; o[1] = pg.y + ty; o[2] = pg.z + tz;
    }
}

template <int J>
__global__ __launch_bounds__(256) void k2b_skin_kernel(const SkinArgs a, int bpad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.num_out) return;
    const int v = a.vertex_ids ? a.vertex_ids[i] : i;
    const int f0 = blockIdx.y * kSkinFrames;
    const int NB = a.num_betas, P = a.num_pose_feats, V = a.num_vertices;
    const float* __restrict__ feat = a.feat;
    const float* __restrict__ Am = a.A;

    float w[J];
#pragma unroll
    for (int j = 0; j < J; ++j) w[j] = a.lbs_weights[(size_t)v * J + j];

    float vp[kSkinFrames][3];
    {
        const float t0 = a.v_template[v * 3], t1 = a.v_template[v * 3 + 1], t2 = a.v_template[v * 3 + 2];
#pragma unroll
        for (int fr = 0; fr < kSkinFrames; ++fr) { vp[fr][0] = t0; vp[fr][1] = t1; vp[fr][2] = t2; }
        for (int k = 0; k < NB; ++k) {
            const float s0 = a.shapedirs[((size_t)v * 3 + 0) * NB + k], s1 = a.shapedirs[((size_t)v * 3 + 1) * NB + k],
                        s2 = a.shapedirs[((size_t)v * 3 + 2) * NB + k];
#pragma unroll
            for (int fr = 0; fr < kSkinFrames; ++fr) {
                const int f = f0 + fr < a.num_frames ? f0 + fr : a.num_frames - 1;
                const float b = a.be[(size_t)f * NB + k];
                vp[fr][0] += s0 * b; vp[fr][1] += s1 * b; vp[fr][2] += s2 * b;
            }
        }
    }
    // pose-corrective blend: v_posed += feat . posedirs[:, 3v..3v+2]
    const float* __restrict__ pd = a.posedirs + (size_t)3 * v;
#pragma unroll 4
    for (int k = 0; k < P; ++k) {
        const float p0 = pd[(size_t)k * 3 * V], p1 = pd[(size_t)k * 3 * V + 1], p2 = pd[(size_t)k * 3 * V + 2];
        const float* __restrict__ fk = feat + (size_t)k * bpad + f0;   // wave-uniform address
#pragma unroll
        for (int fr = 0; fr < kSkinFrames; ++fr) {
            const float ff = fk[fr];
            vp[fr][0] += ff * p0; vp[fr][1] += ff * p1; vp[fr][2] += ff * p2;
        }
    }
    // skinning: T = sum_j w_j A_j ; out = T [v_posed; 1] + transl
#pragma unroll
    for (int fr = 0; fr < kSkinFrames; ++fr) {
        const int f = f0 + fr;
        if (f >= a.num_frames) break;
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] += w[j] * Am[((size_t)j * 12 + e) * bpad + f];
        }
        float ox = T[0] * vp[fr][0] + T[1] * vp[fr][1] + T[2] * vp[fr][2] + T[3];
        float oy = T[4] * vp[fr][0] + T[5] * vp[fr][1] + T[6] * vp[fr][2] + T[7];
        float oz = T[8] * vp[fr][0] + T[9] * vp[fr][1] + T[10] * vp[fr][2] + T[11];
        if (a.tr) { ox += a.tr[(size_t)f * 3]; oy += a.tr[(size_t)f * 3 + 1]; oz += a.tr[(size_t)f * 3 + 2]; }
        float* o = a.out + ((size_t)f * a.out_stride + a.out_row0 + i) * 3;
        o[0] = ox; o[1] = oy; o[2] = oz;
    }
}

// generic-J fallback (weights streamed from memory)
__global__ __launch_bounds__(256) void k2b_skin_kernel_anyj(const SkinArgs a, int bpad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.num_out) return;
    const int v = a.vertex_ids ? a.vertex_ids[i] : i;
    const int NB = a.num_betas, P = a.num_pose_feats, V = a.num_vertices, J = a.num_joints;
    for (int fr = 0; fr < kSkinFrames; ++fr) {
        const int f = blockIdx.y * kSkinFrames + fr;
        if (f >= a.num_frames) break;
        float vp[3];
        for (int c = 0; c < 3; ++c) {
            float s = a.v_template[v * 3 + c];
            for (int k = 0; k < NB; ++k) s += a.shapedirs[((size_t)v * 3 + c) * NB + k] * a.be[(size_t)f * NB + k];
            vp[c] = s;
        }
        for (int k = 0; k < P; ++k) {
            const float ff = a.feat[(size_t)k * bpad + f];
            for (int c = 0; c < 3; ++c) vp[c] += ff * a.posedirs[(size_t)k * 3 * V + 3 * v + c];
        }
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < J; ++j) {
            const float wj = a.lbs_weights[(size_t)v * J + j];
            for (int e = 0; e < 12; ++e) T[e] += wj * a.A[((size_t)j * 12 + e) * bpad + f];
        }
        float o3[3];
        for (int r = 0; r < 3; ++r)
            o3[r] = T[4 * r] * vp[0] + T[4 * r + 1] * vp[1] + T[4 * r + 2] * vp[2] + T[4 * r + 3] +
                    (a.tr ? a.tr[(size_t)f * 3 + r] : 0.f);
        float* o = a.out + ((size_t)f * a.out_stride + a.out_row0 + i) * 3;
        o[0] = o3[0]; o[1] = o3[1]; o[2] = o3[2];
    }
}

int skin_bpad(int num_frames) { return (num_frames + kSkinFrames - 1) / kSkinFrames * kSkinFrames; }

hipError_t launch_pose_setup(const PoseArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0) return hipSuccess;
    hipLaunchKernelGGL(k2b_pose_setup_kernel, dim3(a.num_frames), dim3(64), 0, stream, a, skin_bpad(a.num_frames));
    return hipGetLastError();
}

hipError_t launch_skin(const SkinArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0 || a.num_out <= 0) return hipSuccess;
    const dim3 grid((a.num_out + 255) / 256, (a.num_frames + kSkinFrames - 1) / kSkinFrames);
    const int bpad = skin_bpad(a.num_frames);
    if (a.num_joints == 24)
        hipLaunchKernelGGL(k2b_skin_kernel<24>, grid, dim3(256), 0, stream, a, bpad);
    else
        hipLaunchKernelGGL(k2b_skin_kernel_anyj, grid, dim3(256), 0, stream, a, bpad);
    return hipGetLastError();
}

}  // namespace k2b
