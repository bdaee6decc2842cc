// k2b_fit.hip — fused world-space SMPLify fit for gfx950 (MI355X, CDNA4).
//
// One launch runs ALL Adam iterations of `WorldSpaceFitter.fit_frame`'s Adam branch
// (reference keypoints2body/core/fitters/world_space.py:248-256) for a batch of independent
// frames.  One frame per 64-lane wavefront; a workgroup of up to 8 waves shares the GMM
// precision matrices, which stay resident in LDS (141 KB of the CU's 160 KB) for the whole
// launch.  Per iteration and frame nothing is read from or written to HBM.
//
// Lane roles of a wave
//   "row layout"   lane l holds optimiser state for flat parameter p = l (register set A)
//                  and p = 64 + l (set B); p runs over [global_orient 3 | body_pose 69 |
//                  betas NB | transl 3].  The GMM rows use this layout too: lane l <-> body
//                  pose index l-3 (set A), lanes 0..7 of set B <-> indices 61..68.
//   "joint layout" lane j < 24 owns joint j of the kinematic tree: Rodrigues, the chain
//                  down-sweep (parent records staged in wave-private LDS), the loss
//                  gradient, the up-sweep of subtree force/torque sums and the Rodrigues
//                  reverse mode.
// The two layouts exchange values through a 96-float wave-private LDS staging strip.
//
// Arithmetic restated (see oracle/fit_torch.py for the CPU twin and the reference lines):
//   joints  p_j = p_par + Rg_par (J_j(beta) - J_par(beta)),  Rg_j = Rg_par R_j   (smplx chain)
//   loss    w_j^2 sum_k c_k^2 gmof(p_k + t - y_k) + w_pp^2 min_m(0.5 d_m^T P_m d_m - log nllw_m)
//           + w_a^2 sum exp(s_i th_i)^2 + w_s^2 |beta|^2 + w_pr^2 |th - th_0|^2   (losses.py:49-66)
//   Adam    torch.optim.Adam single-tensor update (torch/optim/adam.py), bias terms from host.
#include "k2b_internal.h"

namespace k2b {

namespace {

constexpr int NJ = kFitJoints;          // 24
constexpr int D = kPriorDim;            // 69
constexpr int MAXW = kFitMaxWaves;      // waves (frames) per workgroup
constexpr int PA_FLOATS = kPriorMaxGauss * (17 * 256 + 64);  // 35328
constexpr int XS = 96;                  // staging strip: go@0, body@4, betas@76, transl@92
constexpr int XS_BODY = 4, XS_BETA = 76, XS_TRANSL = 92;
constexpr int TREE_REC = 16;            // Rg(9) p(3) pad(4)
constexpr int UP_REC = 8;               // a(3) tau(3) pad(2)
constexpr int WAVE_LDS = XS + NJ * TREE_REC + NJ * UP_REC;  // 672 floats
static_assert((PA_FLOATS + MAXW * WAVE_LDS) * 4 <= 163840, "LDS budget");

// 8 per-lane values -> one value per lane: lane l ends with the sum over the 8 lanes
// {l&7 + 8 s} of v[(l>>3)&7].  7 cross-lane moves instead of 24.
__device__ __forceinline__ float butterfly8(const float (&v)[8], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    float w[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b5 ? v[4 + i] : v[i];
        const float send = b5 ? v[i] : v[4 + i];
        w[i] = keep + __shfl_xor(send, 32, kWave);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b4 ? w[2 + i] : w[i];
        const float send = b4 ? w[i] : w[2 + i];
        u[i] = keep + __shfl_xor(send, 16, kWave);
    }
    const float keep = b3 ? u[1] : u[0];
    const float send = b3 ? u[0] : u[1];
    return keep + __shfl_xor(send, 8, kWave);
}

// 16 per-lane values -> lane l ends with the wave-wide sum of v[(l>>2)&15].
__device__ __forceinline__ float butterfly16_sum(const float (&v)[16], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b5 ? v[8 + i] : v[i];
        const float send = b5 ? v[i] : v[8 + i];
        w8[i] = keep + __shfl_xor(send, 32, kWave);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b4 ? w8[4 + i] : w8[i];
        const float send = b4 ? w8[i] : w8[4 + i];
        w4[i] = keep + __shfl_xor(send, 16, kWave);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b3 ? w4[2 + i] : w4[i];
        const float send = b3 ? w4[i] : w4[2 + i];
        w2[i] = keep + __shfl_xor(send, 8, kWave);
    }
    const float keep = b2 ? w2[1] : w2[0];
    const float send = b2 ? w2[0] : w2[1];
    float r = keep + __shfl_xor(send, 4, kWave);
    r += __shfl_xor(r, 2, kWave);
    r += __shfl_xor(r, 1, kWave);
    return r;
}

}  // namespace

__global__ __launch_bounds__(MAXW * 64) void k2b_fit_world_kernel(const FitArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[PA_FLOATS + MAXW * WAVE_LDS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int waves = blockDim.x >> 6;
    const int M = a.num_gauss;

    // ---- 0. GMM precision image -> LDS (shared by every wave of the workgroup) ----------
    {
        const float4* src = reinterpret_cast<const float4*>(a.pa_image);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < PA_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int f = blockIdx.x * waves + wave;   // frame of this wave
    if (f >= a.num_frames) return;             // no further workgroup-wide sync below

    float* xs = lds + PA_FLOATS + wave * WAVE_LDS;
    float* tree = xs + XS;
    float* up = tree + NJ * TREE_REC;
    const float* pa = lds;                                    // [m][17][64][4]
    const float* pa68 = lds + kPriorMaxGauss * 17 * 256;      // [m][64]

    const int NB = a.num_betas;
    const int nparamB = 8 + NB + 3;            // lanes of set B that hold a parameter

    // ---- 1. per-lane constants ----------------------------------------------------------
    // row layout: staging offsets of this lane's two parameters
    const int offA = lane < 3 ? lane : XS_BODY + (lane - 3);
    const int offB = lane < 8 ? XS_BODY + 61 + lane : (lane < 8 + NB ? XS_BETA + (lane - 8) : XS_TRANSL + (lane - 8 - NB));
    const bool actB = lane < nparamB;
    const bool bodyA = lane >= 3, bodyB = lane < 8;
    const bool betaB = lane >= 8 && lane < 8 + NB;
    const bool optB = actB && !(betaB && a.freeze_betas);

    float muA[kPriorMaxGauss], cA[kPriorMaxGauss];
#pragma unroll
    for (int m = 0; m < kPriorMaxGauss; ++m) {
        muA[m] = m < M ? a.row_const[(0 * kPriorMaxGauss + m) * 64 + lane] : 0.f;
        cA[m] = m < M ? a.row_const[(1 * kPriorMaxGauss + m) * 64 + lane] : 0.f;
    }
    // B rows after the butterfly: lane (r = l&7, s = l>>3) owns row 61+r of component s
    const float muB = a.row_const[2 * kPriorMaxGauss * 64 + lane];
    const float cB = a.row_const[2 * kPriorMaxGauss * 64 + 64 + lane];

    // angle prior: sign (0 = not a prior index) for the set-A parameter of this lane
    float angA = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane == 3 + a.angle_index[i]) angA = a.angle_sign[i];

    // joint layout constants
    const bool isJ = lane < NJ;
    const int jl = isJ ? lane : 0;
    const int parent = a.tree[jl * 8 + 0];
    const int depth = isJ ? a.tree[jl * 8 + 1] : -1;
    const int ch0 = a.tree[jl * 8 + 2], ch1 = a.tree[jl * 8 + 3], ch2 = a.tree[jl * 8 + 4];
    float dt[3], dd[3][kMaxBetas];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dt[c] = a.dt[jl * 3 + c];
#pragma unroll
        for (int k = 0; k < kMaxBetas; ++k) dd[c][k] = a.dd[(jl * 3 + c) * kMaxBetas + k];
    }
    const int thoff = jl == 0 ? 0 : XS_BODY + 3 * (jl - 1);

    // targets: lane j holds the target of its joint (or none)
    const int tk = isJ ? a.lane_target[jl] : -1;
    Vec3 tgt = {0.f, 0.f, 0.f};
    float wconf = 0.f;  // w_j^2 c^2
    if (tk >= 0) {
        const float* y = a.j3d + ((size_t)f * a.num_targets + tk) * 3;
        tgt = {y[0], y[1], y[2]};
        const float c = a.conf ? a.conf[(a.conf_per_frame ? (size_t)f * a.num_targets : 0) + tk] : 1.0f;
        wconf = (a.joint_w * a.joint_w) * (c * c);
    }

    // ---- 2. parameters and optimiser state (row layout) -----------------------------------
    auto load_param = [&](int p, const float* go, const float* bp, const float* be, const float* tr) -> float {
        if (p < 3) return go[(size_t)f * 3 + p];
        if (p < 3 + D) return bp[(size_t)f * D + (p - 3)];
        if (p < 3 + D + NB) return be[(size_t)f * NB + (p - 3 - D)];
        return tr[(size_t)f * 3 + (p - 3 - D - NB)];
    };
    float x0 = load_param(lane, a.go_in, a.bp_in, a.be_in, a.tr_in);
    float x1 = actB ? load_param(64 + lane, a.go_in, a.bp_in, a.be_in, a.tr_in) : 0.f;
    const float* prsrc = a.preserve ? a.preserve : a.bp_in;
    const float pr0 = bodyA ? prsrc[(size_t)f * D + (lane - 3)] : 0.f;
    const float pr1 = bodyB ? prsrc[(size_t)f * D + 61 + lane] : 0.f;
    float m0 = 0.f, v0 = 0.f, m1 = 0.f, v1 = 0.f;

    const float s2 = a.sigma * a.sigma;
    const float wpp2 = a.pose_prior_w * a.pose_prior_w;
    const float wa2 = a.angle_w * a.angle_w;
    const float ws2 = a.shape_w * a.shape_w;
    const float wpr2 = a.preserve_w * a.preserve_w;
    const float om_b1 = a.one_minus_beta1;       // lerp weight float(1 - beta1), formed in double on host
    const float om_b2 = a.one_minus_beta2;       // float(1 - beta2) computed in double on host

    float loss_total = 0.f;
    float g0 = 0.f, g1 = 0.f;

    for (int it = 0; it < a.num_iters; ++it) {
        const bool last = it == a.num_iters - 1;
        // ---- a. parameters -> staging strip ------------------------------------------------
        xs[offA] = x0;
        if (actB) xs[offB] = x1;
        wave_sync();

        // ---- b. joint-layout reads -----------------------------------------------------------
        const Vec3 th = {xs[thoff], xs[thoff + 1], xs[thoff + 2]};
        float beta[kMaxBetas];
#pragma unroll
        for (int k = 0; k < kMaxBetas; ++k) beta[k] = k < NB ? xs[XS_BETA + k] : 0.f;
        const Vec3 tr = {xs[XS_TRANSL], xs[XS_TRANSL + 1], xs[XS_TRANSL + 2]};

        // ---- c. GMM prior: y_m = P_m theta - P_m mu_m for every component ----------------------
        float acc[kPriorMaxGauss];
#pragma unroll
        for (int m = 0; m < kPriorMaxGauss; ++m) acc[m] = -cA[m];
#pragma unroll 1
        for (int jb = 0; jb < 17; ++jb) {
            const float4 t4 = *reinterpret_cast<const float4*>(xs + XS_BODY + 4 * jb);
#pragma unroll
            for (int m = 0; m < kPriorMaxGauss; ++m) {
                const float4 p4 = *reinterpret_cast<const float4*>(pa + ((m * 17 + jb) * 64 + lane) * 4);
                acc[m] += p4.x * t4.x;
                acc[m] += p4.y * t4.y;
                acc[m] += p4.z * t4.z;
                acc[m] += p4.w * t4.w;
            }
        }
        {
            const float t68 = xs[XS_BODY + 68];
#pragma unroll
            for (int m = 0; m < kPriorMaxGauss; ++m) acc[m] += pa68[m * 64 + lane] * t68;
        }
        // rows 61..68: lane (r, s) sums columns 9s..9s+8 of row 61+r for every component
        const int rB = lane & 7, sB = lane >> 3;
        float pb[kPriorMaxGauss];
        {
            // keep these 72 L1-resident loads inside the loop: hoisted, they would pin 72 VGPRs
            const float* pbp = a.pb;
            asm volatile("" : "+s"(pbp));
            float tb[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int col = 9 * sB + c;
                tb[c] = col < D ? xs[XS_BODY + col] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < kPriorMaxGauss; ++m) {
                float s = 0.f;
                if (m < M) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) s += pbp[(m * 9 + c) * 64 + lane] * tb[c];
                }
                pb[m] = s;
            }
        }
        const float yB = butterfly8(pb, lane) - cB;          // component sB, row 61+rB
        const float xB = xs[XS_BODY + 61 + rB];
        float zB = (xB - muB) * yB;
        zB += __shfl_xor(zB, 4, kWave);
        zB += __shfl_xor(zB, 2, kWave);
        zB += __shfl_xor(zB, 1, kWave);                      // sum over r, component sB
        float z[kPriorMaxGauss];
#pragma unroll
        for (int m = 0; m < kPriorMaxGauss; ++m) z[m] = bodyA ? (x0 - muA[m]) * acc[m] : 0.f;
        float q = butterfly8(z, lane);
        q += __shfl_xor(q, 4, kWave);
        q += __shfl_xor(q, 2, kWave);
        q += __shfl_xor(q, 1, kWave);                        // lanes 8m..8m+7: d^T P d of component m
        const float val = 0.5f * (q + zB) + a.neg_log_nllw[sB < M ? sB : 0];
        float best = read_lane(val, 0);
        int mstar = 0;
#pragma unroll
        for (int m = 1; m < kPriorMaxGauss; ++m) {
            const float vm = read_lane(val, 8 * m);
            if (m < M && vm < best) { best = vm; mstar = m; }
        }
        float yA = acc[0];
#pragma unroll
        for (int m = 1; m < kPriorMaxGauss; ++m) yA = (mstar == m) ? acc[m] : yA;
        const float yBs = __shfl(yB, rB + 8 * mstar, kWave);  // row 61+lane for lanes < 8

        // ---- d. forward kinematics (joint layout) -----------------------------------------------
        Vec3 dj;
        {
            float e[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = dt[c];
#pragma unroll
                for (int k = 0; k < kMaxBetas; ++k) s += dd[c][k] * beta[k];
                e[c] = s;
            }
            dj = {e[0], e[1], e[2]};
        }
        const Rodrigues rod = rodrigues_fwd(isJ ? th : Vec3{0.f, 0.f, 0.f});
        Mat3 Rgp, Rg;                      // parent's and own global rotation
        Vec3 pj = dj;                      // posed joint (without transl)
#pragma unroll
        for (int i = 0; i < 9; ++i) { Rgp.m[i] = (i % 4 == 0) ? 1.f : 0.f; Rg.m[i] = rod.R.m[i]; }
        if (depth == 0) {
            float* rec = tree + lane * TREE_REC;
            *reinterpret_cast<float4*>(rec) = {Rg.m[0], Rg.m[1], Rg.m[2], Rg.m[3]};
            *reinterpret_cast<float4*>(rec + 4) = {Rg.m[4], Rg.m[5], Rg.m[6], Rg.m[7]};
            *reinterpret_cast<float4*>(rec + 8) = {Rg.m[8], pj.x, pj.y, pj.z};
        }
        for (int lev = 1; lev <= a.max_depth; ++lev) {
            wave_sync();
            if (depth == lev) {
                const float* prec = tree + parent * TREE_REC;
                const float4 r0 = *reinterpret_cast<const float4*>(prec);
                const float4 r1 = *reinterpret_cast<const float4*>(prec + 4);
                const float4 r2 = *reinterpret_cast<const float4*>(prec + 8);
                Rgp.m[0] = r0.x; Rgp.m[1] = r0.y; Rgp.m[2] = r0.z; Rgp.m[3] = r0.w;
                Rgp.m[4] = r1.x; Rgp.m[5] = r1.y; Rgp.m[6] = r1.z; Rgp.m[7] = r1.w;
                Rgp.m[8] = r2.x;
                const Vec3 pp = {r2.y, r2.z, r2.w};
                Rg = mul(Rgp, rod.R);
                pj = pp + mul(Rgp, dj);
                float* rec = tree + lane * TREE_REC;
                *reinterpret_cast<float4*>(rec) = {Rg.m[0], Rg.m[1], Rg.m[2], Rg.m[3]};
                *reinterpret_cast<float4*>(rec + 4) = {Rg.m[4], Rg.m[5], Rg.m[6], Rg.m[7]};
                *reinterpret_cast<float4*>(rec + 8) = {Rg.m[8], pj.x, pj.y, pj.z};
            }
        }

        // ---- e. joint loss, its gradient, subtree force / torque sums ------------------------------
        Vec3 gj = {0.f, 0.f, 0.f};
        float part = 0.f;                  // per-lane partial of the loss (summed at the end)
        if (tk >= 0) {
            const float ex = pj.x + tr.x - tgt.x, ey = pj.y + tr.y - tgt.y, ez = pj.z + tr.z - tgt.z;
            const float x2 = ex * ex, y2 = ey * ey, z2 = ez * ez;
            const float dx = s2 + x2, dy = s2 + y2, dz = s2 + z2;
            if (last) part = wconf * ((s2 * x2) / dx + (s2 * y2) / dy + (s2 * z2) / dz);
            gj = {wconf * 2.f * ex * (s2 * s2) / (dx * dx), wconf * 2.f * ey * (s2 * s2) / (dy * dy),
                  wconf * 2.f * ez * (s2 * s2) / (dz * dz)};
        }
        Vec3 aj = gj;                      // sum of joint-loss gradients over the subtree
        Vec3 tj = cross(pj, gj);           // sum of p x g over the subtree
        if (isJ) {
            float* rec = up + lane * UP_REC;
            *reinterpret_cast<float4*>(rec) = {aj.x, aj.y, aj.z, tj.x};
            *reinterpret_cast<float2*>(rec + 4) = {tj.y, tj.z};
        }
        for (int lev = a.max_depth - 1; lev >= 0; --lev) {
            wave_sync();
            if (depth == lev && ch0 >= 0) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int c = s == 0 ? ch0 : (s == 1 ? ch1 : ch2);
                    if (c >= 0) {
                        const float4 r0 = *reinterpret_cast<const float4*>(up + c * UP_REC);
                        const float2 r1 = *reinterpret_cast<const float2*>(up + c * UP_REC + 4);
                        aj.x += r0.x; aj.y += r0.y; aj.z += r0.z;
                        tj.x += r0.w; tj.y += r1.x; tj.z += r1.y;
                    }
                }
                float* rec = up + lane * UP_REC;
                *reinterpret_cast<float4*>(rec) = {aj.x, aj.y, aj.z, tj.x};
                *reinterpret_cast<float2*>(rec + 4) = {tj.y, tj.z};
            }
        }
        wave_sync();
        // torque about this joint of every force below it, expressed in the parent frame
        const Vec3 torque = tj - cross(pj, aj);
        const Vec3 w = mulT(Rgp, torque);
        Mat3 G;   // 0.5 [w]x R : the tangent-space cotangent of R_j
        {
            const float* R = rod.R.m;
            G.m[0] = 0.5f * (-w.z * R[3] + w.y * R[6]); G.m[1] = 0.5f * (-w.z * R[4] + w.y * R[7]); G.m[2] = 0.5f * (-w.z * R[5] + w.y * R[8]);
            G.m[3] = 0.5f * (w.z * R[0] - w.x * R[6]);  G.m[4] = 0.5f * (w.z * R[1] - w.x * R[7]);  G.m[5] = 0.5f * (w.z * R[2] - w.x * R[8]);
            G.m[6] = 0.5f * (-w.y * R[0] + w.x * R[3]); G.m[7] = 0.5f * (-w.y * R[1] + w.x * R[4]); G.m[8] = 0.5f * (-w.y * R[2] + w.x * R[5]);
        }
        const Vec3 gth = rodrigues_bwd(rod, th, G);
        const Vec3 gd = mulT(Rgp, aj);     // dL/d(J_j - J_parent)
        float gb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) gb[k] = isJ ? gd.x * dd[0][k] + gd.y * dd[1][k] + gd.z * dd[2][k] : 0.f;
        const float gbeta = butterfly16_sum(gb, lane);   // lanes 4k..4k+3 hold d joint-loss / d beta_k
        const Vec3 groot = {read_lane(aj.x, 0), read_lane(aj.y, 0), read_lane(aj.z, 0)};  // = d/d transl

        // ---- f. joint layout -> row layout, then the priors (already in row layout) -------------------
        if (isJ) { xs[thoff] = gth.x; xs[thoff + 1] = gth.y; xs[thoff + 2] = gth.z; }
        if ((lane & 3) == 0 && (lane >> 2) < NB) xs[XS_BETA + (lane >> 2)] = gbeta;
        if (lane == 63) { xs[XS_TRANSL] = groot.x; xs[XS_TRANSL + 1] = groot.y; xs[XS_TRANSL + 2] = groot.z; }
        wave_sync();
        g0 = xs[offA];
        g1 = actB ? xs[offB] : 0.f;
        wave_sync();
        if (bodyA) {
            g0 += wpp2 * yA + 2.f * wpr2 * (x0 - pr0);
            if (angA != 0.f) {
                const float e = expf(x0 * angA);
                g0 += wa2 * 2.f * angA * e * e;
                if (last) part += wa2 * e * e;
            }
            if (last) part += wpr2 * (x0 - pr0) * (x0 - pr0);
        }
        if (bodyB) {
            g1 += wpp2 * yBs + 2.f * wpr2 * (x1 - pr1);
            if (last) part += wpr2 * (x1 - pr1) * (x1 - pr1);
        }
        if (betaB) {
            g1 += 2.f * ws2 * x1;
            if (last) part += ws2 * x1 * x1;
        }
        if (last) loss_total = wave_sum(part) + wpp2 * best;

        // ---- g. Adam (torch.optim.Adam, single-tensor path) ------------------------------------------
        const float2 co = a.adam_coef[it];     // {lr / (1 - b1^t), sqrt(1 - b2^t)}
        {
            m0 = m0 + om_b1 * (g0 - m0);
            v0 = v0 * a.beta2 + om_b2 * g0 * g0;
            const float denom = sqrtf(v0) / co.y + a.eps;
            x0 = x0 - co.x * (m0 / denom);
        }
        if (optB) {
            m1 = m1 + om_b1 * (g1 - m1);
            v1 = v1 * a.beta2 + om_b2 * g1 * g1;
            const float denom = sqrtf(v1) / co.y + a.eps;
            x1 = x1 - co.x * (m1 / denom);
        }
    }

    // ---- 4. results -----------------------------------------------------------------------------------
    auto store_param = [&](int p, float v) {
        if (p < 3) a.go_out[(size_t)f * 3 + p] = v;
        else if (p < 3 + D) a.bp_out[(size_t)f * D + (p - 3)] = v;
        else if (p < 3 + D + NB) a.be_out[(size_t)f * NB + (p - 3 - D)] = v;
        else a.tr_out[(size_t)f * 3 + (p - 3 - D - NB)] = v;
    };
    store_param(lane, x0);
    if (actB) store_param(64 + lane, x1);
    if (lane == 0 && a.loss_out) a.loss_out[f] = loss_total;
    if (a.grad_out) {
        const int P = 3 + D + NB + 3;
        a.grad_out[(size_t)f * P + lane] = g0;
        if (actB) a.grad_out[(size_t)f * P + 64 + lane] = (betaB && a.freeze_betas) ? 0.f : g1;
    }
}

hipError_t launch_fit_world(const FitArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0) return hipSuccess;
    // enough waves per workgroup to cover the frames with one workgroup per CU, at most MAXW
    int waves = (a.num_frames + a.num_cus - 1) / a.num_cus;
    waves = waves < 1 ? 1 : (waves > MAXW ? MAXW : waves);
    const int blocks = (a.num_frames + waves - 1) / waves;
    hipLaunchKernelGGL(k2b_fit_world_kernel, dim3(blocks), dim3(waves * 64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace k2b
