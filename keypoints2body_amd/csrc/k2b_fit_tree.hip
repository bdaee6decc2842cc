// k2b_fit_tree.hip — fused fit for LARGE kinematic trees (25..64 joints: SMPL-H / SMPL-X) on gfx950.
//
// Same work as k2b_fit.hip (all Adam iterations of `WorldSpaceFitter.fit_frame`'s Adam branch, reference
// keypoints2body/core/fitters/world_space.py:248-256, in one launch; nothing touches HBM inside the loop), for
// models whose tree does not fit the 24-lane layout of that kernel:
//   * one wavefront per frame, lane l = the l-th joint in DFS pre-order (every subtree a contiguous lane range);
//     the optimiser state lives in the lane of its joint: theta_j (3), and on the low lanes the shape
//     coefficients (betas | expression, one per lane) and the translation (lanes 0..2);
//   * forward kinematics level by level with cross-lane moves (12 ds_bpermute per level, depth <= 15), the
//     analytic backward exactly as in k2b_fit.hip: subtree sums of the joint-loss gradient g and of p x g as
//     differences of ONE inclusive prefix scan (double precision: the differences must not cancel), the torque
//     about each joint pulled back to its rotation vector with the closed-form left Jacobian of SO(3);
//   * max-mixture pose prior (core/prior.py:182-195) over the first D_v <= 64 body-pose dimensions, lane i = prior
//     dimension i.  The reference's SMPL-X handling feeds a 69-D body pose to a 69-D mixture although smplx's
//     SMPL-X has 63 body dimensions (SURVEY.md note N3); this engine defines it as the 69-D mixture evaluated
//     at [theta_body(63) | 0 x 6].  With the six padded dimensions constant the quadratic form folds, ON THE HOST,
//     into a D_v-dimensional one:  q_m = d^T A_m d + 2 b_m^T d + c_m  (d = theta_v - mu_v; A = P_vv, b = P_vc d_c,
//     c = d_c^T P_cc d_c, d_c = -mu_c), and y' = A theta_v + h (h = b - A mu_v) is both the gradient of 0.5 q and,
//     through q = d.(y' + b) + c, the value.  A theta_v runs on the MATRIX CORES, for all frames of the workgroup at once:
//     A_m = the leading block of the 64 x 64 core of P_m, whose f16 hi/lo fragments k2b_fit.hip already uses (with
//     theta_j = 0 for j >= D_v the core's extra columns contribute nothing, its extra rows are never read).  The eight
//     components' fragments (128 KiB) are resident in LDS; wave w owns the components w, w + tw, ...: it multiplies them
//     with the f16 hi/lo strips of theta_v that every frame's wave published (three v_mfma_f32_16x16x32_f16 products
//     per tile, one frame per MFMA column), adds h, reduces q per frame and hands y' and q back through LDS; two
//     workgroup barriers per iteration.  (The first version did it on the vector ALU, per frame: 128 ds_read_b128 +
//     512 FMAs + 64 v_readlane per iteration - half of the kernel's instructions.)
// Loss terms, Adam arithmetic and the "loss of the last iteration before its step" convention are those of
// k2b_fit.hip (oracle: oracle/fit_torch.py; goldens: tests/golden/smplx_fit_*.npz).
#include "k2b_internal.h"
#include "k2b_lanes.h"

namespace k2b {

namespace {

constexpr int TW = 8;                    // frames (waves) per workgroup
constexpr int TMG = kPriorMaxGauss;      // 8 mixture components
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ double shfl64(double v, int src) { return bperm64(src << 2, v); }

template <int NS, bool CHAIN>            // NS: capacity for shape coefficients (betas | expression): 16, 20 or 32; CHAIN: warm-start chains
__global__ __launch_bounds__(64 * TW) void k2b_fit_tree_kernel(const FitTreeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tlds[];
    // [8][16][64] half8 A fragments (128 KiB) | [M][64] h | [M][64] b | [M][64] mu | [TW][64] theta_v fp32 |
    // [TW][hi 64 | lo 64] theta_v f16 | [TW][M][64] y' | [TW][M] q | [M] constants | [M] 1 / scale
    const half8* const sFrag = reinterpret_cast<const half8*>(tlds);
    float* const sH = tlds + TMG * 16 * 64 * 4;
    float* const sB = sH + TMG * 64;
    float* const sMu = sB + TMG * 64;
    float* const sTh = sMu + TMG * 64;
    _Float16* const sTh16 = reinterpret_cast<_Float16*>(sTh + TW * 64);
    float* const sY = sTh + TW * 64 + TW * 64;           // (TW x 128 halfs = TW x 64 floats)
    float* const sQ = sY + TW * TMG * 64;                // 0.5 q + (0.5 c_m - log nll weight): what the arg-min compares
    float* const sPcl = sQ + TW * TMG;                   // [M] that constant | [M] 1 / fragment scale - read from LDS, not from
    float* const sIsc = sPcl + TMG;                      //  global memory, inside the iteration (k2b_fit.hip has the story)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int J = a.num_joints, NB = a.num_shape, Dv = a.prior_dims;
    // Two shapes.  Plain: every wave carries a frame, and between two barriers every wave also runs its share of the
    // mixture's components for all frames of the workgroup.  Component-wave shape (<= 4 frames per CU, i.e. at most one
    // frame per SIMD): four EXTRA waves - one per SIMD, beside the frame wave - own two components each, keep their
    // fragments in registers for the whole launch (no fragment image in LDS) and evaluate them for every frame WHILE the frame
    // waves run kinematics and the backward; the frame waves only wait for y' and q before the priors.  In the plain shape the
    // component phase sat between the barriers as a latency chain (fragment reads, six dependent MFMAs per tile, LDS round
    // trips: 44 % of an iteration at 1024 frames).
    const int ncw = a.comp_waves;                                  // 0 or 4
    const int tw = (blockDim.x >> 6) - ncw;                        // frames (waves) of this workgroup: 1..TW, by batch size
    const bool comp_role = wave >= tw;
    const int fr_raw = blockIdx.x * tw + (comp_role ? 0 : wave);
    const bool frame_ok = !comp_role && fr_raw < a.num_frames;
    const int fr = frame_ok ? fr_raw : a.num_frames - 1;           // idle waves shadow the last frame and write nothing

    // ---- prior image -> LDS (whole workgroup) ---------------------------------------------------------------------------
    if (ncw == 0)
        for (int i = threadIdx.x; i < TMG * 16 * 64; i += blockDim.x)
            reinterpret_cast<float4*>(tlds)[i] = reinterpret_cast<const float4*>(a.pfrag)[i];
    for (int i = threadIdx.x; i < TMG * 64; i += blockDim.x) { sH[i] = a.ph[i]; sB[i] = a.pb[i]; sMu[i] = a.pmu[i]; }
    if (threadIdx.x < TMG) { sPcl[threadIdx.x] = a.pcl[threadIdx.x]; sIsc[threadIdx.x] = a.inv_scale[threadIdx.x]; }
    __syncthreads();

    if (comp_role) {
        // ---- component waves: components cw and cw + 4, fragments resident in registers (128 VGPRs) ------------------------------
        if (!(a.pose_prior_w * a.pose_prior_w > 0.f)) return;      // no mixture term: the frame waves take no barriers either
        const int cw = wave - tw;
        const int cn = lane & 15, cg = lane >> 4;                  // MFMA lane: column (frame slot) and k / row group
        const int cslot = cn < tw ? cn : tw - 1;
        half8 fh[2][4][2], fl[2][4][2];                            // [own component][row tile][k-step]
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const half8* fi = reinterpret_cast<const half8*>(a.pfrag) + (size_t)(((cw + 4 * o) * 4 + t) * 4) * 64 + lane;
                fh[o][t][0] = fi[0]; fh[o][t][1] = fi[64]; fl[o][t][0] = fi[128]; fl[o][t][1] = fi[192];
            }
        const float isc[2] = {sIsc[cw], sIsc[cw + 4]}, pcl[2] = {sPcl[cw], sPcl[cw + 4]};
        const int steps = CHAIN ? a.chain_len : 1;
        for (int step = 0; step < steps; ++step) {
            const int nit = step == 0 ? a.num_iters : a.chain_iters;
            for (int it = 0; it < nit; ++it) {
                __syncthreads();                                   // theta_v of every frame published
                const half8 bh0 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 8 * cg);
                const half8 bh1 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 32 + 8 * cg);
                const half8 bl0 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 64 + 8 * cg);
                const half8 bl1 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 96 + 8 * cg);
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int c = cw + 4 * o;
                    // the four row tiles are independent chains: issued step-major, small terms first (as in the plain shape)
                    floatx4 acc[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[o][t][0], bh0, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[o][t][1], bh1, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[o][t][0], bl0, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[o][t][1], bl1, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[o][t][0], bh0, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[o][t][1], bh1, acc[t], 0, 0, 0);
                    const float inv_scale = isc[o];
                    float qp = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r0 = 16 * t + 4 * cg;
                        const floatx4 th4 = *reinterpret_cast<const floatx4*>(sTh + cslot * 64 + r0);
                        const floatx4 mu4 = *reinterpret_cast<const floatx4*>(sMu + c * 64 + r0);
                        const floatx4 h4 = *reinterpret_cast<const floatx4*>(sH + c * 64 + r0);
                        const floatx4 b4 = *reinterpret_cast<const floatx4*>(sB + c * 64 + r0);
                        floatx4 y;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            y[i] = acc[t][i] * inv_scale + h4[i];
                            qp += (th4[i] - mu4[i]) * (y[i] + b4[i]);
                        }
                        *reinterpret_cast<floatx4*>(sY + (cslot * TMG + c) * 64 + r0) = y;
                    }
                    qp = pair_sum32(qp);
                    qp = pair_sum16(qp);
                    sQ[cslot * TMG + c] = 0.5f * qp + pcl[o];
                }
                asm volatile("" ::"v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));   // (B fragments live past the last MFMA, see k2b_fit.hip)
                __syncthreads();                                   // y' and q published
            }
        }
        return;
    }

    // ---- per-lane constants ---------------------------------------------------------------------------------------------
    const int* tb = a.tab + lane * 8;
    const int joint = tb[0], ssize = tb[2];
    int anc[4];                                  // byte address (lane x 4) of the ancestor 2^r levels up, or of lane 63
#pragma unroll
    for (int r = 0; r < 4; ++r) anc[r] = a.anc[lane * 4 + r] << 2;
    const int psrc = tb[4], pcomp = tb[5];       // prior layout: lane i takes theta[pcomp] of lane psrc (or -1)
    const int pd0 = tb[6];                       // tree layout: this joint's first prior dimension (or -1)
    const bool isJ = lane < J;
    const float dtx = a.dt[lane * 3], dty = a.dt[lane * 3 + 1], dtz = a.dt[lane * 3 + 2];
    float dd[3][NS];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < NS; ++k) dd[c][k] = k < NB ? a.dd[(lane * 3 + c) * 32 + k] : 0.f;

    // target of this joint
    const int tk = isJ ? a.lane_target[lane] : -1;
    float ty0 = 0.f, ty1 = 0.f, ty2 = 0.f, wconf = 0.f;
    if (tk >= 0) {
        const size_t f0 = (size_t)fr * (CHAIN ? a.chain_len : 1);
        const float* y = a.j3d + (f0 * a.num_targets + tk) * 3;
        ty0 = y[0]; ty1 = y[1]; ty2 = y[2];
        const float cf = a.conf ? a.conf[(a.conf_per_frame ? f0 * a.num_targets : 0) + tk] : 1.0f;
        wconf = (a.joint_w * a.joint_w) * (cf * cf);
    }
    const float s2 = a.sigma * a.sigma;

    // ---- parameters and Adam state -----------------------------------------------------------------------------------------
    const int D = 3 * (J - 1);
    float th[3] = {0.f, 0.f, 0.f};
    if (isJ) {
        const float* src = joint == 0 ? a.go_in + (size_t)fr * 3 : a.bp_in + (size_t)fr * D + 3 * (joint - 1);
        th[0] = src[0]; th[1] = src[1]; th[2] = src[2];
    }
    float sh = lane < NB ? a.be_in[(size_t)fr * NB + lane] : 0.f;
    float tr = lane < 3 ? a.tr_in[(size_t)fr * 3 + lane] : 0.f;
    float mth[3] = {0.f, 0.f, 0.f}, vth[3] = {0.f, 0.f, 0.f}, msh = 0.f, vsh = 0.f, mtr = 0.f, vtr = 0.f;
    // which parameters the optimiser owns
    const bool opt_th = isJ && (joint == 0 ? (a.opt_mask & 1) : (a.opt_mask & 2));
    const bool opt_sh = lane < NB && (a.opt_mask & 4) && !(a.freeze_betas && lane < a.num_betas_prior);
    const bool opt_tr = lane < 3 && (a.opt_mask & 8);
    // prior-layout constants (lane i = prior dimension i)
    const bool isP = lane < Dv;
    float pres0 = isP ? (a.preserve ? a.preserve[(size_t)fr * D + lane] : a.bp_in[(size_t)fr * D + lane]) : 0.f;
    float ang_s = 0.f;                           // sign of the bending prior on this dimension (0: none)
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (a.angle_index[i] == lane && isP) ang_s = a.angle_sign[i];
    // warm-start chain (a.chain_len > 1): this wave carries a SEQUENCE; step 0 is its first frame (no preserve term, num_iters
    // iterations), every later step starts from the previous result, preserves it and runs chain_iters iterations with a
    // fresh optimiser state (reference api/sequence.py:214-281, world_space.py:159,211,214); frame rows fr * chain_len + step
    // (a template parameter: the plain launch keeps its register budget - with the chain's mutable state in the same
    //  instantiation the SMPL-X kernel went from 247 registers to 256 + 40 bytes of scratch)
    const int chain = CHAIN ? a.chain_len : 1;
    const float wpp = a.pose_prior_w * a.pose_prior_w, wang = a.angle_w * a.angle_w, wsh = a.shape_w * a.shape_w;
    float wpr = CHAIN ? 0.f : a.preserve_w * a.preserve_w;

    float loss_total = 0.f;
    float gth[3] = {0.f, 0.f, 0.f}, gsh = 0.f, gtr = 0.f;

    for (int step = 0; step < chain; ++step) {
    const int nit = step == 0 ? a.num_iters : a.chain_iters;
    if (CHAIN && step > 0) {
        if (tk >= 0) {                           // targets of this step's frame
            const size_t ft = (size_t)fr * chain + step;
            const float* y = a.j3d + (ft * a.num_targets + tk) * 3;
            ty0 = y[0]; ty1 = y[1]; ty2 = y[2];
            const float cf = a.conf ? a.conf[(a.conf_per_frame ? ft * a.num_targets : 0) + tk] : 1.0f;
            wconf = (a.joint_w * a.joint_w) * (cf * cf);
        }
        {                                        // preserve the previous result (prior layout), fresh Adam state
            const float c0 = shfl(th[0], psrc < 0 ? 0 : psrc), c1 = shfl(th[1], psrc < 0 ? 0 : psrc), c2 = shfl(th[2], psrc < 0 ? 0 : psrc);
            pres0 = psrc < 0 ? 0.f : (pcomp == 0 ? c0 : (pcomp == 1 ? c1 : c2));
        }
        wpr = a.preserve_w * a.preserve_w;
#pragma unroll
        for (int c = 0; c < 3; ++c) mth[c] = vth[c] = 0.f;
        msh = vsh = mtr = vtr = 0.f;
    }
    const bool use_gmm = wpp > 0.f;              // (uniform over the launch: the barriers below are taken by every wave or by none)
    const int cn = lane & 15, cg = lane >> 4;    // MFMA lane: column (frame slot of the workgroup) and k / row group
    const int cslot = cn < tw ? cn : tw - 1;     // columns beyond the workgroup's frames repeat the last slot
    for (int it = 0; it < nit; ++it) {
        const float2 co = a.adam_coef[it];       // requested a whole iteration ahead of its use (at the use it was a global load with a
                                                 // full wait on the iteration's critical path)
        // ---- body pose in prior layout (lane i = prior dimension i); published for the component role ------------------------------
        float thv;
        {
            const float c0 = shfl(th[0], psrc < 0 ? 0 : psrc), c1 = shfl(th[1], psrc < 0 ? 0 : psrc), c2 = shfl(th[2], psrc < 0 ? 0 : psrc);
            thv = psrc < 0 ? 0.f : (pcomp == 0 ? c0 : (pcomp == 1 ? c1 : c2));
        }
        if (use_gmm) {
            const _Float16 hh = (_Float16)thv;
            sTh[wave * 64 + lane] = thv;
            sTh16[wave * 128 + lane] = hh;
            sTh16[wave * 128 + 64 + lane] = (_Float16)(thv - (float)hh);
            __syncthreads();
          if (ncw == 0) {
            // component role: y' = A theta_v + h and q = d . (y' + b) of the components w, w + tw, ... for every frame slot
            const half8 bh0 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 8 * cg);
            const half8 bh1 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 32 + 8 * cg);
            const half8 bl0 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 64 + 8 * cg);
            const half8 bl1 = *reinterpret_cast<const half8*>(sTh16 + cslot * 128 + 96 + 8 * cg);
            for (int c = wave; c < TMG; c += tw) {
                const float inv_scale = sIsc[c], pcl_c = sPcl[c];
                float qp = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const half8* fr4 = sFrag + ((c * 4 + t) * 4) * 64 + lane;
                    const half8 ph0 = fr4[0], ph1 = fr4[64], pl0 = fr4[128], pl1 = fr4[192];
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};              // small terms first, as k2b_fit.hip does
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl0, bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl1, bh1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph0, bl0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph1, bl1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph0, bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph1, bh1, acc, 0, 0, 0);
                    // accumulator layout: lane (n = l & 15, g = l >> 4), register i <-> row 16 t + 4 g + i, column n
                    const int r0 = 16 * t + 4 * cg;
                    const floatx4 th4 = *reinterpret_cast<const floatx4*>(sTh + cslot * 64 + r0);
                    const floatx4 mu4 = *reinterpret_cast<const floatx4*>(sMu + c * 64 + r0);
                    const floatx4 h4 = *reinterpret_cast<const floatx4*>(sH + c * 64 + r0);
                    const floatx4 b4 = *reinterpret_cast<const floatx4*>(sB + c * 64 + r0);
                    floatx4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        y[i] = acc[i] * inv_scale + h4[i];
                        qp += (th4[i] - mu4[i]) * (y[i] + b4[i]);     // (rows >= D_v: theta = mu = h = b = 0)
                    }
                    *reinterpret_cast<floatx4*>(sY + (cslot * TMG + c) * 64 + r0) = y;
                }
                qp = pair_sum32(qp);
                qp = pair_sum16(qp);                                  // summed over the four row groups
                sQ[cslot * TMG + c] = 0.5f * qp + pcl_c;
            }
            // keep the B fragments live past the last MFMA (ROCm 7.2 register allocation, see k2b_fit.hip)
            asm volatile("" ::"v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
            __syncthreads();
          }
        }
        // ---- rest offset from the parent: d = dt + dd . shape -------------------------------------------------------------
        float dx = dtx, dy = dty, dz = dtz;
#pragma unroll
        for (int k = 0; k < NS; ++k) {               // (directions beyond NB are zero: no branch on the runtime count)
            const float sk = read_lane(sh, k);
            dx = fmaf(dd[0][k], sk, dx); dy = fmaf(dd[1][k], sk, dy); dz = fmaf(dd[2][k], sk, dz);
        }
        const Rodrigues rod = rodrigues_fwd({th[0], th[1], th[2]});

        // ---- forward kinematics by pointer doubling: after round r a lane holds its transform relative to the ancestor
        // 2^(r+1) levels up (global once it runs out of ancestors).  Lanes without an ancestor at that distance fetch
        // lane 63, which is no joint and holds the identity throughout: no select in the rounds.
        Mat3 Rg = rod.R;
        Vec3 pg = {dx, dy, dz};
        auto round = [&](int r) __attribute__((always_inline)) {
            Mat3 pR;
#pragma unroll
            for (int i = 0; i < 9; ++i) pR.m[i] = bperm(anc[r], Rg.m[i]);
            const Vec3 pp = {bperm(anc[r], pg.x), bperm(anc[r], pg.y), bperm(anc[r], pg.z)};
            pg = mul(pR, pg) + pp;
            Rg = mul(pR, Rg);
        };
        if (a.num_rounds == 4) {                 // the full SMPL-X / SMPL-H tree: straight-line code (a conditional round ends in twelve
            round(0); round(1); round(2); round(3);   // register copies where its results join the skipped path)
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < a.num_rounds) round(r);
        }

        // ---- joint term (losses.py:6-10,49-51) ---------------------------------------------------------------------------------------
        const float t0 = read_lane(tr, 0), t1 = read_lane(tr, 1), t2 = read_lane(tr, 2);
        float lj = 0.f;
        Vec3 g = {0.f, 0.f, 0.f};
        if (tk >= 0) {
            const float ex = pg.x + t0 - ty0, ey = pg.y + t1 - ty1, ez = pg.z + t2 - ty2;
            const float x2 = ex * ex, y2 = ey * ey, z2 = ez * ez;
            const float qx = s2 + x2, qy = s2 + y2, qz = s2 + z2;
            if (it == nit - 1) lj = wconf * ((s2 * x2) / qx + (s2 * y2) / qy + (s2 * z2) / qz);   // (the loss that leaves is the last iteration's)
            const float k2 = 2.f * wconf * (s2 * s2);
            g = {k2 * ex * fast_rcp(qx * qx), k2 * ey * fast_rcp(qy * qy), k2 * ez * fast_rcp(qz * qz)};   // (1-ulp reciprocal, as k2b_fit.hip:
                                                                                                           //  three IEEE divisions were 33 instructions)
        }

        // ---- subtree sums: S = sum g, Mo = sum p x g over the subtree of every joint ----------------------------------------------------
        const Vec3 pxg = cross(pg, g);
        const float v6[6] = {g.x, g.y, g.z, pxg.x, pxg.y, pxg.z};
        float sub[6];
        const int hi = lane + ssize - 1 < 63 ? lane + ssize - 1 : 63;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float v = isJ ? v6[i] : 0.f;
            asm volatile("" : "+v"(v));                  // (select on the float: behind the conversion it is two selects on the double)
            const double pre = wave_inclusive_scan(v, lane);
            const double up = shfl64(pre, hi);           // prefix at the last lane of the subtree
            sub[i] = (float)(up - (pre - (double)v));    // minus the prefix just before its first lane (this lane): no second fetch
        }
        const Vec3 S = {sub[0], sub[1], sub[2]};
        const Vec3 torque = Vec3{sub[3], sub[4], sub[5]} - cross(pg, S);
        // pull back: Rg = Rgp R  =>  Rgp^T v = R (Rg^T v)
        const Vec3 w = mul(rod.R, mulT(Rg, torque));
        const Vec3 gd = mul(rod.R, mulT(Rg, S));                    // d loss / d (rest offset of this joint)
        {
            const float a1 = rod.s * rod.inv_angle, a3 = (1.0f - rod.c) * rod.inv_angle;
            const float uw = rod.u.x * w.x + rod.u.y * w.y + rod.u.z * w.z;
            const float a2uw = (1.0f - a1) * uw;
            const Vec3 uxw = cross(rod.u, w);
            gth[0] = isJ ? a1 * w.x + a2uw * rod.u.x - a3 * uxw.x : 0.f;
            gth[1] = isJ ? a1 * w.y + a2uw * rod.u.y - a3 * uxw.y : 0.f;
            gth[2] = isJ ? a1 * w.z + a2uw * rod.u.z - a3 * uxw.z : 0.f;
        }
        // translation: the whole tree's force (root subtree = everything), one component per lane 0..2
        {
            const float s0 = read_lane(S.x, 0), s1 = read_lane(S.y, 0), s2r = read_lane(S.z, 0);
            gtr = lane == 0 ? s0 : (lane == 1 ? s1 : s2r);
        }
        // shape coefficients through the rest offsets: g_k = sum_l gd_l . dd_l[:, k]  (+ shape prior on the betas)
        gsh = 0.f;
#pragma unroll
        for (int blk = 0; blk < (NS + 15) / 16; ++blk) { // 16 coefficients per butterfly: lane l ends with the total of k = (l >> 2) & 15
            if constexpr (NS % 16 != 0 && NS % 16 <= 4) {
                if (blk == NS / 16) {                    // a last block of at most four coefficients (SMPL-X: 16 + 4): a four-value butterfly
                    float part4[4];                      // instead of a sixteen-value one over twelve zeros
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int kk = 16 * blk + k < NS ? 16 * blk + k : 0;
                        part4[k] = (16 * blk + k < NS && isJ) ? gd.x * dd[0][kk] + gd.y * dd[1][kk] + gd.z * dd[2][kk] : 0.f;
                    }
                    const float tot = butterfly4_sum(part4);                  // row r of 16 lanes: coefficient 16 blk + r
                    const float mine = shfl(tot, (lane & 3) << 4);
                    if ((lane >> 4) == blk) gsh = mine;                       // (lanes 16 blk + 4 .. hold no coefficient: masked below)
                    continue;
                }
            }
            float part[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int kk = 16 * blk + k < NS ? 16 * blk + k : 0;                 // (beyond the capacity: compile-time zeros)
                part[k] = (16 * blk + k < NS && isJ) ? gd.x * dd[0][kk] + gd.y * dd[1][kk] + gd.z * dd[2][kk] : 0.f;
            }
            const float tot = butterfly16_sum(part, lane);
            const float mine = shfl(tot, (lane & 15) << 2);          // coefficient 16 blk + (lane & 15) sits in lanes 4 k .. 4 k + 3
            if ((lane >> 4) == blk) gsh = mine;
        }
        float lsh = 0.f;
        if (lane < a.num_betas_prior) { lsh = wsh * sh * sh; gsh += 2.f * wsh * sh; }

        // ---- priors on the body pose, prior layout (lane i = prior dimension i) ----------------------------------------------------------
        float gv = 0.f, lv = 0.f;                // gradient and loss contributions in prior layout
        float lpr = 0.f;
        if (use_gmm) {
            if (ncw != 0) __syncthreads();       // component-wave shape: y' and q arrive while this wave ran the tree
            // arg-min over the components (those beyond M carry a 3e38 constant - never the arg-min): lanes 0..7 take one value
            // each, an 8-lane DPP minimum, then the lowest lane that holds it - the first minimum wins, as torch.min does and as
            // the chain of eight compare / select steps this replaces
            const float ell = lane < TMG ? sQ[wave * TMG + lane] : 3.0e38f;
            const float best = read_lane(vmin(group8_min(ell), 3.0e38f), 0);
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < TMG && ell == best);
            const int bm = hit ? (int)__builtin_ctzll(hit) : 0;
            const float yb = sY[(wave * TMG + bm) * 64 + lane];
            gv = isP ? wpp * yb : 0.f;
            lpr = wpp * best;
        }
        if (ang_s != 0.f) {                       // losses.py:13-21,54: w^2 exp(s theta)^2
            const float e = __expf(ang_s * thv);
            lv += wang * e * e;
            gv += 2.f * wang * ang_s * e * e;
        }
        if (wpr > 0.f && isP) {
            const float df = thv - pres0;
            lv += wpr * df * df;
            gv += 2.f * wpr * df;
        }
        // back to the tree layout: joint lane l takes dimensions pd0 .. pd0 + 2
        {
            const int s0i = pd0 < 0 ? 0 : pd0;
            const float a0 = shfl(gv, s0i), a1 = shfl(gv, s0i + 1 < 64 ? s0i + 1 : 63), a2 = shfl(gv, s0i + 2 < 64 ? s0i + 2 : 63);
            if (pd0 >= 0) { gth[0] += a0; gth[1] += a1; gth[2] += a2; }
        }
        if (it == nit - 1) loss_total = wave_sum_fast(lj + lsh + lv) + lpr;

        // ---- Adam (torch.optim.Adam, single-tensor path; bias terms from the host table) ----------------------------------------------
        auto adam = [&](float& x, float& mm, float& vv, float gi, bool on) {
            if (!on) return;
            mm = mm + a.one_minus_beta1 * (gi - mm);
            vv = vv * a.beta2 + a.one_minus_beta2 * gi * gi;
            const float denom = fast_sqrt(vv) * fast_rcp(co.y) + a.eps;
            x = x - co.x * (mm * fast_rcp(denom));
        };
#pragma unroll
        for (int c = 0; c < 3; ++c) adam(th[c], mth[c], vth[c], gth[c], opt_th);
        adam(sh, msh, vsh, gsh, opt_sh);
        adam(tr, mtr, vtr, gtr, opt_tr);
    }
    // results of this step's frame (a plain launch has one step and row fr)
    if (frame_ok) {
        const size_t fo = CHAIN ? (size_t)fr * chain + step : (size_t)fr;
        if (isJ) {
            float* dst = joint == 0 ? a.go_out + fo * 3 : a.bp_out + fo * D + 3 * (joint - 1);
            dst[0] = th[0]; dst[1] = th[1]; dst[2] = th[2];
        }
        if (lane < NB) a.be_out[fo * NB + lane] = sh;
        if (lane < 3) a.tr_out[fo * 3 + lane] = tr;
        if (lane == 0 && a.loss_out) a.loss_out[fo] = loss_total;
    }
    }  // chain steps

    if (!frame_ok) return;
    if (a.grad_out) {
        const int P = 3 + D + NB + 3;
        float* go = a.grad_out + (size_t)fr * P;
        if (isJ) {
            float* dst = joint == 0 ? go : go + 3 + 3 * (joint - 1);
            dst[0] = opt_th ? gth[0] : 0.f; dst[1] = opt_th ? gth[1] : 0.f; dst[2] = opt_th ? gth[2] : 0.f;
        }
        if (lane < NB) go[3 + D + lane] = opt_sh ? gsh : 0.f;
        if (lane < 3) go[3 + D + NB + lane] = opt_tr ? gtr : 0.f;
    }
}

}  // namespace

hipError_t launch_fit_tree(const FitTreeArgs& a_in, hipStream_t stream) {
    if (a_in.num_frames <= 0) return hipSuccess;
    if (a_in.num_joints > 64 || a_in.num_shape > 32 || a_in.prior_dims > 64 || a_in.num_gauss > TMG) return hipErrorInvalidValue;
    const size_t lds = (size_t)(TMG * 16 * 64 * 4 + 3 * TMG * 64 + 2 * TW * 64 + TW * TMG * 64 + TW * TMG + 2 * TMG) * sizeof(float);
    // frames per workgroup: enough to cover the batch with one workgroup per CU (up to 8: two waves per SIMD); small
    // batches get fewer waves per CU, so that every SIMD hosts at most one frame and all CUs work
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int tw = (a_in.num_frames + cus - 1) / cus;
    tw = tw < 1 ? 1 : (tw > TW ? TW : tw);
    FitTreeArgs a = a_in;
    // at most one frame per SIMD: four component waves ride along (one per SIMD) and take the mixture off the frame waves
    a.comp_waves = a.debug_shape == 1 ? 0 : ((tw <= 4 || a.debug_shape == 2) ? 4 : 0);
    if (a.debug_shape == 2 && tw > 4) tw = 4;
    const dim3 grid((a.num_frames + tw - 1) / tw), block(64 * (tw + a.comp_waves));
    hipError_t e;
#define K2B_TREE(NS_, CH_)                                                                                             \
    do {                                                                                                               \
        static std::atomic<unsigned long long> lds_set{0};                                                             \
        e = ensure_dynamic_lds(k2b_fit_tree_kernel<NS_, CH_>, lds_set, lds);                                    \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((k2b_fit_tree_kernel<NS_, CH_>), grid, block, lds, stream, a);                              \
    } while (0)
    const bool chain = a.chain_len > 1;
    // (capacity 20 = SMPL-X's betas | expression: twelve coefficients of padding cost 36 registers in the 32-wide instantiation)
    if (a.num_shape <= 16) { if (chain) K2B_TREE(16, true); else K2B_TREE(16, false); }
    else if (a.num_shape <= 20) { if (chain) K2B_TREE(20, true); else K2B_TREE(20, false); }
    else { if (chain) K2B_TREE(32, true); else K2B_TREE(32, false); }
#undef K2B_TREE
    return hipGetLastError();
}

}  // namespace k2b
