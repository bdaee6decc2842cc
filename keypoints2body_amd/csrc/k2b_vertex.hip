// k2b_vertex.hip — joint-loss term of VERTEX-SELECTED joints (model joint index >= J), and the
// stand-alone Adam step that goes with it.
//
// The fused fit kernel (k2b_fit.hip) fits kinematic joints only: its loss never touches a vertex.  smplx
// appends "extra" joints that are single mesh vertices (nose, eyes, ears, toes, heels, finger tips:
// `vertex_joint_selector`), and the reference lets a caller fit them through `target_model_indices`
// (core/fitters/world_space.py:198-201).  This file is the path for that case: k2b_fit_world then queues, per Adam
// iteration, (1) the fused kernel in evaluate-only mode for the kinematic targets and all priors and (2)
// `k2b_vertex_term_kernel` with its Adam tail: vertex term, sum of both gradients, the frame's optimiser step
// (k2b_api.hip::fit_world_vertex_joints; no host work between the launches).  Without the tail the kernel is the
// stand-alone term behind k2b_vertex_term (L-BFGS closures, tests).
//
// Vertex term, one 64-lane workgroup per frame (a handful of vertices: written for clarity, not speed).
// For a selected vertex with rest position, shape / pose blend rows and skinning weights (t, S, Pd, w):
//   vp = t + S beta + Pd vec(R_1..R_23 - I)                       (smplx blend shapes)
//   x  = sum_j w_j [ Rg_j (vp - Jr_j) + p_j ] + transl           (linear blend skinning, Jr = rest joints)
//   L  = w_joint^2 conf^2 sum_xyz gmof(x - y)                     (core/losses.py:6-10,49-51)
// Backward by hand.  With g = dL/dx, F_j = sum_e w_ej g_e and M_j = sum_e w_ej q_ej x g_e (q_ej the point of
// vertex e rigidly attached to joint j), a rotation of joint k moves everything attached at or below k, so the
// torque about p_k is sum_{j in subtree(k)} M_j - p_k x sum F_j, pulled back to theta_k exactly as in the fused
// kernel (left Jacobian of the rotation vector); joint offsets get Rgp_k^T sum F_j; the blend shapes get
// g_vp = sum_j w_j Rg_j^T g, which reaches beta through S and the local rotations through Pd (as an extra
// torque axial(G_R R^T) on each joint); the rest joints inside the skinning transform get -Rg_j^T F_j.
#include "k2b_internal.h"

namespace k2b {

namespace {

constexpr int VE = 32;                  // selected vertices per launch (one lane each)

__device__ __forceinline__ Vec3 axial_of_GRt(const Mat3& G, const Mat3& R) {
    // axial(G R^T): M = G R^T, result (M32 - M23, M13 - M31, M21 - M12)
    float M[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M[3 * r + c] = G.m[3 * r] * R.m[3 * c] + G.m[3 * r + 1] * R.m[3 * c + 1] + G.m[3 * r + 2] * R.m[3 * c + 2];
    return {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
}

// VJM / NBM: capacity in joints / shape coefficients (24 / 16: SMPL, the arithmetic of round 1 unchanged; 64 / 32: SMPL-H, SMPL-X)
template <int VJM, int NBM>
__global__ __launch_bounds__(64) void k2b_vertex_term_kernel(const VertexTermArgs a) {
    constexpr int kMaxBetas = NBM;          // (shadows the 24-joint kernel's constant inside this kernel)
    const int VJ = a.num_joints;
    __shared__ float sR[VJM][9], sRg[VJM][9], sp[VJM][3], sJr[VJM][3];
    __shared__ float sX[9 * (VJM - 1) + 1];
    __shared__ float svp[VE][3], sg[VE][3], sgvp[VE][3];
    __shared__ float sF[VJM][3], sM[VJM][3];
    __shared__ float sGX[9 * (VJM - 1) + 1];
    __shared__ float sGrad[3 + 3 * (VJM - 1) + NBM + 3];
    __shared__ int spar[VJM];

    const int f = blockIdx.x;
    const int lane = threadIdx.x;
    const int NB = a.num_betas, E = a.num_sel, V = a.num_vertices;
    const int PF = 9 * (VJ - 1);                   // 207 pose features
    const int D = 3 * (VJ - 1);
    const bool isJ = lane < VJ, isE = lane < E;

    // ---- joints: local rotation, rest joint, chain ---------------------------------------------------
    Vec3 th = {0.f, 0.f, 0.f};
    int par = -1;
    Vec3 Jr = {0.f, 0.f, 0.f};
    if (isJ) {
        const float* src = lane == 0 ? a.go + (size_t)f * 3 : a.bp + (size_t)f * D + 3 * (lane - 1);
        th = {src[0], src[1], src[2]};
        par = a.parents[lane];
        float e[3];
        for (int c = 0; c < 3; ++c) {
            float s = a.j_template[lane * 3 + c];
            for (int k = 0; k < NB; ++k) s += a.j_dirs[(lane * 3 + c) * NB + k] * a.be[(size_t)f * NB + k];
            e[c] = s;
        }
        Jr = {e[0], e[1], e[2]};
    }
    const Rodrigues rod = rodrigues_fwd(th);
    if (isJ) {
        for (int i = 0; i < 9; ++i) sR[lane][i] = rod.R.m[i];
        sJr[lane][0] = Jr.x; sJr[lane][1] = Jr.y; sJr[lane][2] = Jr.z;
        spar[lane] = par < 0 ? -1 : par;
        if (lane > 0)
            for (int i = 0; i < 9; ++i) sX[(lane - 1) * 9 + i] = rod.R.m[i] - ((i % 4 == 0) ? 1.f : 0.f);
    }
    __syncthreads();
    Mat3 Rg = rod.R;
    Vec3 pg = Jr;
    if (isJ) {
        // p_j = p_par + Rg_par (Jr_j - Jr_par): walk towards the root
        if (par >= 0) pg = Jr - Vec3{sJr[par][0], sJr[par][1], sJr[par][2]};
        for (int anc = par; anc >= 0; anc = spar[anc]) {
            Mat3 Ra;
            for (int i = 0; i < 9; ++i) Ra.m[i] = sR[anc][i];
            const int pa = spar[anc];
            const Vec3 da = pa >= 0 ? Vec3{sJr[anc][0] - sJr[pa][0], sJr[anc][1] - sJr[pa][1], sJr[anc][2] - sJr[pa][2]}
                                    : Vec3{sJr[anc][0], sJr[anc][1], sJr[anc][2]};
            pg = mul(Ra, pg) + da;
            Rg = mul(Ra, Rg);
        }
        for (int i = 0; i < 9; ++i) sRg[lane][i] = Rg.m[i];
        sp[lane][0] = pg.x; sp[lane][1] = pg.y; sp[lane][2] = pg.z;
    }
    __syncthreads();

    // ---- selected vertices: forward, loss, dL/dx, dL/dvp ---------------------------------------------------
    const int vid = isE ? a.extra_ids[a.sel[lane]] : 0;
    float loss_e = 0.f;
    Vec3 g = {0.f, 0.f, 0.f};
    if (isE) {
        float vp[3];
        for (int c = 0; c < 3; ++c) {
            float s = a.v_template[(size_t)vid * 3 + c];
            for (int k = 0; k < NB; ++k) s += a.shapedirs[((size_t)vid * 3 + c) * NB + k] * a.be[(size_t)f * NB + k];
            for (int k = 0; k < PF; ++k) s += a.posedirs[(size_t)k * 3 * V + (size_t)vid * 3 + c] * sX[k];
            vp[c] = s;
        }
        Vec3 x = {0.f, 0.f, 0.f};
        for (int j = 0; j < VJ; ++j) {
            const float w = a.lbs_weights[(size_t)vid * VJ + j];
            Mat3 R;
            for (int i = 0; i < 9; ++i) R.m[i] = sRg[j][i];
            const Vec3 q = mul(R, Vec3{vp[0] - sJr[j][0], vp[1] - sJr[j][1], vp[2] - sJr[j][2]}) + Vec3{sp[j][0], sp[j][1], sp[j][2]};
            x.x += w * q.x; x.y += w * q.y; x.z += w * q.z;
        }
        const float tx = a.tr[(size_t)f * 3], ty = a.tr[(size_t)f * 3 + 1], tz = a.tr[(size_t)f * 3 + 2];
        const int kcol = a.sel_k[lane];
        const float* y = a.targets + ((size_t)f * a.num_targets + kcol) * 3;
        const float ex = x.x + tx - y[0], ey = x.y + ty - y[1], ez = x.z + tz - y[2];
        const float cf = a.conf ? a.conf[(a.conf_per_frame ? (size_t)f * a.num_targets : 0) + kcol] : 1.0f;
        const float wc = (a.joint_w * a.joint_w) * (cf * cf);
        const float s2 = a.sigma * a.sigma;
        const float x2 = ex * ex, y2 = ey * ey, z2 = ez * ez;
        const float dx = s2 + x2, dy = s2 + y2, dz = s2 + z2;
        loss_e = wc * ((s2 * x2) / dx + (s2 * y2) / dy + (s2 * z2) / dz);
        const float k2 = 2.f * wc * (s2 * s2);
        g = {k2 * ex / (dx * dx), k2 * ey / (dy * dy), k2 * ez / (dz * dz)};
        // dL/dvp = sum_j w_j Rg_j^T g
        Vec3 gvp = {0.f, 0.f, 0.f};
        for (int j = 0; j < VJ; ++j) {
            const float w = a.lbs_weights[(size_t)vid * VJ + j];
            Mat3 R;
            for (int i = 0; i < 9; ++i) R.m[i] = sRg[j][i];
            const Vec3 t = mulT(R, g);
            gvp.x += w * t.x; gvp.y += w * t.y; gvp.z += w * t.z;
        }
        svp[lane][0] = vp[0]; svp[lane][1] = vp[1]; svp[lane][2] = vp[2];
        sg[lane][0] = g.x; sg[lane][1] = g.y; sg[lane][2] = g.z;
        sgvp[lane][0] = gvp.x; sgvp[lane][1] = gvp.y; sgvp[lane][2] = gvp.z;
    }
    __syncthreads();
    const float loss = wave_sum(loss_e);
    const float gtx = wave_sum(g.x), gty = wave_sum(g.y), gtz = wave_sum(g.z);

    // ---- joints: force and moment of the vertices attached to each joint ----------------------------------
    if (isJ) {
        Vec3 F = {0.f, 0.f, 0.f}, M = {0.f, 0.f, 0.f};
        for (int e = 0; e < E; ++e) {
            const int ve = a.extra_ids[a.sel[e]];
            const float w = a.lbs_weights[(size_t)ve * VJ + lane];
            const Vec3 ge = {sg[e][0], sg[e][1], sg[e][2]};
            const Vec3 q = mul(Rg, Vec3{svp[e][0] - Jr.x, svp[e][1] - Jr.y, svp[e][2] - Jr.z}) + pg;
            const Vec3 m = cross(q, ge);
            F.x += w * ge.x; F.y += w * ge.y; F.z += w * ge.z;
            M.x += w * m.x; M.y += w * m.y; M.z += w * m.z;
        }
        sF[lane][0] = F.x; sF[lane][1] = F.y; sF[lane][2] = F.z;
        sM[lane][0] = M.x; sM[lane][1] = M.y; sM[lane][2] = M.z;
    }
    // pose-blend gradient: G_X[k] = sum_e sum_c Pd[k][vid_e, c] g_vp[e][c]
    for (int k = lane; k < PF; k += 64) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) {
            const int ve = a.extra_ids[a.sel[e]];
            for (int c = 0; c < 3; ++c) s += a.posedirs[(size_t)k * 3 * V + (size_t)ve * 3 + c] * sgvp[e][c];
        }
        sGX[k] = s;
    }
    __syncthreads();

    // ---- joints: subtree sums, torque, pull-back ---------------------------------------------------------
    Vec3 gth = {0.f, 0.f, 0.f};
    float gbeta_part[kMaxBetas];
#pragma unroll
    for (int k = 0; k < kMaxBetas; ++k) gbeta_part[k] = 0.f;
    if (isJ) {
        Vec3 aj = {0.f, 0.f, 0.f}, tj = {0.f, 0.f, 0.f};
        for (int k = 0; k < VJ; ++k) {          // k in subtree(lane)  <=>  lane is k or an ancestor of k
            bool below = false;
            for (int t = k; t >= 0; t = spar[t])
                if (t == lane) { below = true; break; }
            if (below) {
                aj.x += sF[k][0]; aj.y += sF[k][1]; aj.z += sF[k][2];
                tj.x += sM[k][0]; tj.y += sM[k][1]; tj.z += sM[k][2];
            }
        }
        const Vec3 torque = tj - cross(pg, aj);
        // Rg = Rgp R  =>  Rgp^T v = R (Rg^T v)
        Vec3 w = mul(rod.R, mulT(Rg, torque));
        const Vec3 gd = mul(rod.R, mulT(Rg, aj));                   // dL/d(Jr_j - Jr_par)
        if (lane > 0) {
            Mat3 G;
            for (int i = 0; i < 9; ++i) G.m[i] = sGX[(lane - 1) * 9 + i];
            w = w + axial_of_GRt(G, rod.R);
        }
        const float a1 = rod.s * rod.inv_angle, a3 = (1.0f - rod.c) * rod.inv_angle;
        const float uw = rod.u.x * w.x + rod.u.y * w.y + rod.u.z * w.z;
        const float a2uw = (1.0f - a1) * uw;
        const Vec3 uxw = cross(rod.u, w);
        gth = {a1 * w.x + a2uw * rod.u.x - a3 * uxw.x, a1 * w.y + a2uw * rod.u.y - a3 * uxw.y, a1 * w.z + a2uw * rod.u.z - a3 * uxw.z};
        // betas through the joint offsets (gd) and through the rest joints inside the skinning transform
        const Vec3 F = {sF[lane][0], sF[lane][1], sF[lane][2]};
        const Vec3 gJ = mulT(Rg, F);                                 // dL/dJr_j = -Rg_j^T F_j
        for (int k = 0; k < NB; ++k) {
            float s = 0.f;
            for (int c = 0; c < 3; ++c) {
                const float dj = a.j_dirs[(lane * 3 + c) * NB + k];
                const float dp = par >= 0 ? a.j_dirs[(par * 3 + c) * NB + k] : 0.f;
                const float gdc = c == 0 ? gd.x : (c == 1 ? gd.y : gd.z);
                const float gjc = c == 0 ? gJ.x : (c == 1 ? gJ.y : gJ.z);
                s += gdc * (dj - dp) - gjc * dj;
            }
            gbeta_part[k] = s;
        }
    }
    // betas through the shape blend of the vertices
    float gbeta_v[kMaxBetas];
#pragma unroll
    for (int k = 0; k < kMaxBetas; ++k) gbeta_v[k] = 0.f;
    if (isE) {
        for (int k = 0; k < NB; ++k) {
            float s = 0.f;
            for (int c = 0; c < 3; ++c) s += a.shapedirs[((size_t)vid * 3 + c) * NB + k] * sgvp[lane][c];
            gbeta_v[k] = s;
        }
    }

    // ---- outputs ---------------------------------------------------------------------------------------------
    const int P = 3 + D + NB + 3;
    if (isJ) {
        float* dst = lane == 0 ? sGrad : sGrad + 3 + 3 * (lane - 1);
        dst[0] = gth.x; dst[1] = gth.y; dst[2] = gth.z;
    }
#pragma unroll
    for (int k = 0; k < kMaxBetas; ++k) {
        const float s = wave_sum(gbeta_part[k] + gbeta_v[k]);
        if (lane == 0 && k < NB) sGrad[3 + D + k] = s;
    }
    if (lane == 0) {
        sGrad[3 + D + NB] = gtx; sGrad[3 + D + NB + 1] = gty; sGrad[3 + D + NB + 2] = gtz;
        a.loss_out[f] = loss + (a.loss_in ? a.loss_in[f] : 0.f);
    }
    __syncthreads();
    if (!a.grad_in) {
        for (int p = lane; p < P; p += 64) a.grad_out[(size_t)f * P + p] = sGrad[p];
        return;
    }
    // Adam tail: total gradient (parameters outside the optimiser get none), torch.optim.Adam's single-tensor step
    const float2 co = *a.adam_coef;
    const float inv_bc2 = fast_rcp(co.y);
    for (int p = lane; p < P; p += 64) {
        const int group = p < 3 ? 0 : (p < 3 + D ? 1 : (p < 3 + D + NB ? 2 : 3));
        const bool opt = ((a.opt_mask >> group) & 1) && !(group == 2 && p - 3 - D < a.frozen_shape);
        const float g = opt ? a.grad_in[(size_t)f * P + p] + sGrad[p] : 0.f;
        if (a.grad_out) a.grad_out[(size_t)f * P + p] = g;
        float* x = group == 0 ? a.go_w + (size_t)f * 3 + p
                 : (group == 1 ? a.bp_w + (size_t)f * D + (p - 3)
                 : (group == 2 ? a.be_w + (size_t)f * NB + (p - 3 - D) : a.tr_w + (size_t)f * 3 + (p - 3 - D - NB)));
        const size_t i = (size_t)f * P + p;
        const float mi = a.adam_m[i] + a.one_minus_beta1 * (g - a.adam_m[i]);
        const float vi = a.adam_v[i] * a.beta2 + a.one_minus_beta2 * g * g;
        const float denom = fast_sqrt(vi) * inv_bc2 + a.eps;
        a.adam_m[i] = mi;
        a.adam_v[i] = vi;
        *x = *x - co.x * (mi * fast_rcp(denom));
    }
}

// torch.optim.Adam single-tensor step on a flat parameter block (same arithmetic as the fused kernel's)
__global__ __launch_bounds__(256) void k2b_adam_kernel(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float lr_over_bc1, float sqrt_bc2,
                                                       float one_minus_beta1, float beta2, float one_minus_beta2, float eps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + one_minus_beta1 * (gi - m[i]);
    const float vi = v[i] * beta2 + one_minus_beta2 * gi * gi;
    const float denom = fast_sqrt(vi) * fast_rcp(sqrt_bc2) + eps;
    m[i] = mi;
    v[i] = vi;
    x[i] = x[i] - lr_over_bc1 * (mi * fast_rcp(denom));
}

}  // namespace

hipError_t launch_vertex_term(const VertexTermArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0 || a.num_sel <= 0) return hipSuccess;
    if (a.num_sel > VE || a.num_joints < 1 || a.num_joints > 64 || a.num_betas > 32) return hipErrorInvalidValue;
    if (a.num_joints <= kFitJoints && a.num_betas <= 16)
        hipLaunchKernelGGL((k2b_vertex_term_kernel<kFitJoints, 16>), dim3(a.num_frames), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((k2b_vertex_term_kernel<64, 32>), dim3(a.num_frames), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_adam(float* x, const float* g, float* m, float* v, long long n, float lr_over_bc1, float sqrt_bc2,
                       float one_minus_beta1, float beta2, float one_minus_beta2, float eps, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k2b_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, g, m, v, n, lr_over_bc1,
                       sqrt_bc2, one_minus_beta1, beta2, one_minus_beta2, eps);
    return hipGetLastError();
}

}  // namespace k2b
