// k2b_fit.hip — fused world-space SMPLify fit for gfx950 (MI355X, CDNA4).
//
// One launch runs ALL Adam iterations of `WorldSpaceFitter.fit_frame`'s Adam branch
// (reference keypoints2body/core/fitters/world_space.py:248-256) for a batch of independent
// frames.  One frame per 64-lane wavefront; a workgroup of up to 8 waves shares the GMM
// precision matrices, which stay resident in LDS (156 KB of the CU's 160 KB) for the whole
// launch.  Per iteration and frame nothing is read from or written to HBM.
//
// Lane roles of a wave
//   "row layout"   lane l holds optimiser state for flat parameter p = l (register set A)
//                  and p = 64 + l (set B); p runs over [global_orient 3 | body_pose 69 |
//                  betas NB | transl 3].  The GMM rows use this layout too: lane l <-> body
//                  pose index l-3 (set A), lanes 0..7 of set B <-> indices 61..68.
//   "tree layout"  lane t < 24 owns the t-th joint of the kinematic tree in DFS pre-order, so
//                  every subtree is a contiguous lane range.  Global transforms come from a
//                  pointer-doubling down-sweep (log2(depth) rounds of cross-lane moves instead
//                  of one round per level); subtree force / torque sums come from windowed
//                  sums over lane ranges (no level loop, no cancellation).  All tree traffic is
//                  ds_bpermute cross-lane moves: the "tree in LDS" is the LDS crossbar, with no
//                  LDS allocation.
// The two layouts exchange values through a 96-float wave-private LDS staging strip.
//
// Two execution shapes (template parameter SPLIT):
//   unified  one wave does everything for its frame (large batches: >= 2 waves per SIMD hide latency);
//   split    small batches (<= 4 frames per CU) leave SIMDs idle, so each frame gets TWO waves that
//            run concurrently on different SIMDs: the "row" wave (GMM prior + Adam, owns the
//            optimiser state) and the "tree" wave (kinematics, joint loss, analytic backward).
//            The two parts of an iteration are independent given the parameters, so the critical
//            path is max(GMM, tree) instead of their sum; they meet at two workgroup barriers per
//            iteration and exchange parameters / gradients through the LDS strips.
//
// Latency, not throughput, bounds this kernel at one wave per SIMD (1024 frames on 1024
// SIMDs), so every phase is written to keep many independent LDS operations in flight:
// the GMM matrix-vector products are software-pipelined by hand (sched_barrier pins the
// order: next block's 9 ds_read_b128 are issued before the current block's 32 FMAs).
//
// Arithmetic restated (see oracle/fit_torch.py for the CPU twin and the reference lines):
//   joints  p_j = p_par + Rg_par (J_j(beta) - J_par(beta)),  Rg_j = Rg_par R_j   (smplx chain)
//   loss    w_j^2 sum_k c_k^2 gmof(p_k + t - y_k) + w_pp^2 min_m(0.5 d_m^T P_m d_m - log nllw_m)
//           + w_a^2 sum exp(s_i th_i)^2 + w_s^2 |beta|^2 + w_pr^2 |th - th_0|^2   (losses.py:49-66)
//   Adam    torch.optim.Adam single-tensor update (torch/optim/adam.py), bias terms from host.
#include "k2b_internal.h"

namespace k2b {

// Diagnostic build only (-DK2B_FIT_STAMPS, tools/stamp_build.sh): s_memtime stamps of one
// iteration of wave 0 / block 0, written to a buffer no other code reads.  The shipped
// library is built without it: no stamp executes there.
#ifdef K2B_FIT_STAMPS
__device__ unsigned long long* g_k2b_stamps = nullptr;
// stamps stay in registers during the iteration and are stored once after the loop, so that a stamp
// costs one s_memtime + lgkmcnt wait and no memory traffic
#define K2B_STAMP(i)                                                                         \
    do {                                                                                     \
        if (stamp_on) {                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                               \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_reg[i])::"memory"); \
            __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                    \
    } while (0)
#else
#define K2B_STAMP(i)
#endif

namespace {

static_assert(kFitJoints == 24, "lane tables are built for the SMPL tree");
constexpr int D = kPriorDim;            // 69
constexpr int MAXW = kFitMaxWaves;      // waves (frames) per workgroup
constexpr int MG = kPriorMaxGauss;      // 8
constexpr int PA_FLOATS = MG * (17 * 256 + 64);   // rows 0..60 (lanes 3..63): [m][17][64][4] + [m][64]
constexpr int PB_FLOATS = MG * 9 * 64;            // rows 61..68: [m][9][64]
constexpr int P_FLOATS = PA_FLOATS + PB_FLOATS;   // 39936 floats = 159744 B
constexpr int XS = 96;                  // staging strip: go@0, body@4, betas@76, transl@92
constexpr int XS_BODY = 4, XS_BETA = 76, XS_TRANSL = 92;
static_assert(P_FLOATS == kPriorImageFloats, "host image size");
static_assert((P_FLOATS + MAXW * XS) * 4 <= 163840, "LDS budget");

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
//  elements - tools/probe/lanes.hip.  The s_nop covers the VALU-write -> cross-lane-read wait states
//  hipcc does not insert inside asm.)
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// v_permlane16_swap: the odd 16-lane rows of `a` trade places with the even rows of `b`.
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
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

__device__ __forceinline__ float wave_sum_fast(float v) {
    v = pair_sum32(v);
    v = pair_sum16(v);
    v += lane_xor8(v);
    v += lane_xor4(v);
    v += lane_xor2(v);
    v += lane_xor1(v);
    return v;
}

// 8 per-lane values -> one value per lane: lane l ends with the sum over the 8 lanes
// {l&7 + 8 s} of v[(l>>3)&7].  Halving exchanges: after swap32 of (v[i], v[4+i]) the two registers
// hold, in every lane, its own and its partner's copy of the value that lane keeps.
__device__ __forceinline__ float butterfly8(const float (&v)[8], int lane) {
    float w[4], u[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float a = v[i], b = v[4 + i];
        swap32(a, b);
        w[i] = a + b;              // lanes < 32: v[i] summed over the pair; lanes >= 32: v[4+i]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float a = w[i], b = w[2 + i];
        swap16(a, b);
        u[i] = a + b;              // even rows: w[i]; odd rows: w[2+i]
    }
    const float s0 = u[0] + lane_xor8(u[0]), s1 = u[1] + lane_xor8(u[1]);
    return (lane & 8) ? s1 : s0;
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

}  // namespace

template <int NBT, bool SPLIT>
__global__ __launch_bounds__(MAXW * 64) void k2b_fit_world_kernel(const FitArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[P_FLOATS + MAXW * XS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int waves = blockDim.x >> 6;
    const int M = a.num_gauss;

    // ---- 0. GMM precision image -> LDS (shared by every wave of the workgroup) ----------
    {
        const float4* src = reinterpret_cast<const float4*>(a.pa_image);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < P_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    // frame and role of this wave
    // (split: waves 0..F-1 are the row waves of frames 0..F-1, waves F..2F-1 their tree waves.  Waves w
    //  and w + 4 of a workgroup share a SIMD, so with F = 4 every SIMD hosts one row wave and one tree
    //  wave - complementary instruction mixes - instead of two of a kind.)
    const int frames_per_wg = SPLIT ? waves >> 1 : waves;
    const int fslot = SPLIT ? (wave < frames_per_wg ? wave : wave - frames_per_wg) : wave;
    const bool do_row = !SPLIT || wave < frames_per_wg;     // GMM prior, priors in row layout, Adam, results
    const bool do_tree = !SPLIT || wave >= frames_per_wg;   // kinematics, joint loss, analytic backward
    const int f_raw = blockIdx.x * frames_per_wg + fslot;
    if (!SPLIT && f_raw >= a.num_frames) return;      // unified: no further workgroup-wide sync below
    // split: waves of a padding slot must still reach every barrier; they recompute the last frame
    // and skip the final stores
    const bool f_valid = f_raw < a.num_frames;
    const int f = f_valid ? f_raw : a.num_frames - 1;

    float* xs = lds + P_FLOATS + (SPLIT ? 2 * fslot : wave) * XS;     // parameters, row wave -> tree wave
    float* gs = SPLIT ? xs + XS : xs;                                  // gradients, tree wave -> row wave
    const float4* pa4 = reinterpret_cast<const float4*>(lds);  // [m][17][64] float4
    const float* pa68 = lds + MG * 17 * 256;                   // [m][64]
    const float* pbl = lds + PA_FLOATS;                        // [m][9][64]

    const int NB = a.num_betas;
    const int nparamB = 8 + NB + 3;            // lanes of set B that hold a parameter

    // ---- 1. per-lane constants ----------------------------------------------------------
    // row layout: staging offsets of this lane's two parameters
    const int offA = lane < 3 ? lane : XS_BODY + (lane - 3);
    const int offB = lane < 8 ? XS_BODY + 61 + lane : (lane < 8 + NB ? XS_BETA + (lane - 8) : XS_TRANSL + (lane - 8 - NB));
    const bool actB = lane < nparamB;
    const bool bodyA = lane >= 3, bodyB = lane < 8;
    const bool betaB = lane >= 8 && lane < 8 + NB;
    const bool translB = actB && !bodyB && !betaB;
    // optimiser membership of this lane's parameters (k2b_fit_config.optimize_mask)
    const bool optA = (a.opt_mask >> (lane < 3 ? 0 : 1)) & 1;
    const bool optB = actB && ((a.opt_mask >> (bodyB ? 1 : (betaB ? 2 : 3))) & 1);

    float muA[MG], cA[MG];
#pragma unroll
    for (int m = 0; m < MG; ++m) {
        muA[m] = m < M ? a.row_const[(0 * MG + m) * 64 + lane] : 0.f;
        cA[m] = m < M ? a.row_const[(1 * MG + m) * 64 + lane] : 0.f;
    }
    // B rows after the butterfly: lane (r = l&7, s = l>>3) owns row 61+r of component s
    const float muB = a.row_const[2 * MG * 64 + lane];
    const float cB = a.row_const[2 * MG * 64 + 64 + lane];
    const int rB = lane & 7, sB = lane >> 3;

    // angle prior: sign (0 = not a prior index) for the set-A parameter of this lane
    float angA = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane == 3 + a.angle_index[i]) angA = a.angle_sign[i];

    // tree layout constants (host tables, DFS pre-order)
    const int* lt = a.lane_tab + lane * kLaneTabStride;
    const int joint = lt[0];                    // joint of this lane, -1 beyond the tree
    const bool isJ = joint >= 0;
    const bool has_par = lt[1] >= 0;
    int anc_addr[kMaxRounds];
    bool anc_ok[kMaxRounds];
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) {
        anc_ok[r] = lt[2 + r] >= 0;
        anc_addr[r] = (anc_ok[r] ? lt[2 + r] : lane) * 4;
    }
    // subtree lane range of this lane's joint, mirrored into the upper half-wave (lane 32 + t)
    bool sub_ok;
    int sub_end_addr;
    {
        const int* lt2 = a.lane_tab + (lane & 31) * kLaneTabStride;
        const int t = lane & 31, size = lt2[2 + kMaxRounds];      // subtree size in lanes (0 beyond the tree)
        sub_ok = lt2[0] >= 0 && size > 0;
        const int first = (lane & 32) + t, lastl = first + (sub_ok ? size - 1 : 0);
        sub_end_addr = lastl * 4;
    }
    float dt[3], dd[3][NBT];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dt[c] = a.dt[lane * 3 + c];
#pragma unroll
        for (int k = 0; k < NBT; ++k) dd[c][k] = a.dd[(lane * 3 + c) * kMaxBetas + k];
    }
    const int jj = isJ ? joint : 0;
    const int thoff = jj == 0 ? 0 : XS_BODY + 3 * (jj - 1);

    // targets: the lane of joint j holds the target of that joint (or none)
    const int tk = isJ ? a.lane_target[jj] : -1;
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
    const float tp1 = translB ? a.tr_prior[(size_t)f * 3 + (lane - 8 - NB)] : 0.f;   // centre of the transl prior
    float m0 = 0.f, v0 = 0.f, m1 = 0.f, v1 = 0.f;

    const float s2 = a.sigma * a.sigma;
    const float wpp2 = a.pose_prior_w * a.pose_prior_w;
    const float wa2 = a.angle_w * a.angle_w;
    const float ws2 = a.shape_w * a.shape_w;
    const float wpr2 = a.preserve_w * a.preserve_w;
    const float wt2 = a.transl_prior_w * a.transl_prior_w;
    const float om_b1 = a.one_minus_beta1;       // lerp weight float(1 - beta1), formed in double on host
    const float om_b2 = a.one_minus_beta2;       // float(1 - beta2) computed in double on host

    float loss_total = 0.f;
    float g0 = 0.f, g1 = 0.f;
#ifdef K2B_FIT_STAMPS
    unsigned long long stamp_reg[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    for (int it = 0; it < a.num_iters; ++it) {
        const bool last = it == a.num_iters - 1;
#ifdef K2B_FIT_STAMPS
        const bool stamp_on = (it == 5) && blockIdx.x == 0 && fslot == 0;   // frame slot 0: its row wave and, in split mode, its tree wave
#endif
        K2B_STAMP(0);
        // ---- a. parameters -> staging strip ------------------------------------------------
        if (do_row) {
            xs[offA] = x0;
            if (actB) xs[offB] = x1;
        }
        if (SPLIT) __syncthreads(); else wave_sync();

        K2B_STAMP(1);
        float yA = 0.f, yBs = 0.f, best = 0.f;
        if (do_row && wpp2 != 0.f) {   // a zero pose-prior weight (camera stage 1) skips the mixture entirely
        // ---- c. GMM prior: y_m = P_m theta - P_m mu_m for every component ----------------------
        // rows 0..60 (set A).  Hand-pipelined: block jb+1's nine ds_read_b128 are in flight
        // while block jb's 32 FMAs issue.
        float acc[MG];
#pragma unroll
        for (int m = 0; m < MG; ++m) acc[m] = -cA[m];
        {
            const float4* xb4 = reinterpret_cast<const float4*>(xs + XS_BODY);
            float4 tq[2], pq[2][MG];
            tq[0] = xb4[0];
#pragma unroll
            for (int m = 0; m < MG; ++m) pq[0][m] = pa4[(m * 17 + 0) * 64 + lane];
#pragma unroll
            for (int jb = 0; jb < 17; ++jb) {
                const int cur = jb & 1, nxt = cur ^ 1;
                if (jb + 1 < 17) {
                    tq[nxt] = xb4[jb + 1];
#pragma unroll
                    for (int m = 0; m < MG; ++m) pq[nxt][m] = pa4[(m * 17 + jb + 1) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MG; ++m) {
                    acc[m] += pq[cur][m].x * tq[cur].x;
                    acc[m] += pq[cur][m].y * tq[cur].y;
                    acc[m] += pq[cur][m].z * tq[cur].z;
                    acc[m] += pq[cur][m].w * tq[cur].w;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        K2B_STAMP(2);
        // column 68 of rows 0..60, then rows 61..68: lane (r, s) sums columns 9s..9s+8 of row
        // 61+r for every component (same pipelining, one component ahead)
        float pb[MG];
        {
            float tb[9], p68[MG], pv[2][9];
            const float t68 = xs[XS_BODY + 68];
#pragma unroll
            for (int m = 0; m < MG; ++m) p68[m] = pa68[m * 64 + lane];
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const int col = 9 * sB + c;
                tb[c] = xs[XS_BODY + (col < D ? col : 0)];
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) pv[0][c] = pbl[(0 * 9 + c) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MG; ++m) acc[m] += p68[m] * t68;
#pragma unroll
            for (int m = 0; m < MG; ++m) {
                const int cur = m & 1, nxt = cur ^ 1;
                if (m + 1 < MG) {
#pragma unroll
                    for (int c = 0; c < 9; ++c) pv[nxt][c] = pbl[((m + 1) * 9 + c) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0);
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 9; ++c) s += pv[cur][c] * tb[c];   // image is zero for columns >= 69
                pb[m] = s;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        K2B_STAMP(3);
        const float yB = butterfly8(pb, lane) - cB;          // component sB, row 61+rB
        const float xB = xs[XS_BODY + 61 + rB];
        const float zB = (xB - muB) * yB;
        float z[MG];
#pragma unroll
        for (int m = 0; m < MG; ++m) z[m] = bodyA ? (x0 - muA[m]) * acc[m] : 0.f;
        float q = butterfly8(z, lane) + zB;                  // partial of component sB over lanes {r + 8 s}
        q += lane_xor4(q);
        q += lane_xor2(q);
        q += lane_xor1(q);                                   // lanes 8m..8m+7: d^T P d of component m
        const float val = 0.5f * q + a.neg_log_nllw[sB < M ? sB : 0];
        best = read_lane(val, 0);
        int mstar = 0;
#pragma unroll
        for (int m = 1; m < MG; ++m) {
            const float vm = read_lane(val, 8 * m);
            if (m < M && vm < best) { best = vm; mstar = m; }
        }
        yA = acc[0];
#pragma unroll
        for (int m = 1; m < MG; ++m) yA = (mstar == m) ? acc[m] : yA;
        yBs = bperm((rB + 8 * mstar) * 4, yB);   // row 61+lane for lanes < 8
        }  // do_row: GMM

        K2B_STAMP(4);
        if (do_tree) {
        // ---- b/d. tree-layout reads, J(beta), Rodrigues ---------------------------------------------
        const Vec3 th = {xs[thoff], xs[thoff + 1], xs[thoff + 2]};
        const Vec3 tr = {xs[XS_TRANSL], xs[XS_TRANSL + 1], xs[XS_TRANSL + 2]};
        Vec3 dj;
        {
            float beta[NBT];
#pragma unroll
            for (int k = 0; k < NBT; ++k) beta[k] = xs[XS_BETA + (k < NB ? k : 0)];
            float e[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = dt[c];
#pragma unroll
                for (int k = 0; k < NBT; ++k) s += dd[c][k] * beta[k];   // dd is zero for k >= NB
                e[c] = s;
            }
            dj = {e[0], e[1], e[2]};
        }
        const Rodrigues rod = rodrigues_fwd(isJ ? th : Vec3{0.f, 0.f, 0.f});
        K2B_STAMP(5);
        // ---- pointer-doubling down-sweep: after round r a lane is composed with 2^(r+1) ancestors ----
        Mat3 Rg = rod.R;                   // becomes the global rotation
        Vec3 pj = dj;                      // becomes the posed joint (without transl)
#pragma unroll
        for (int r = 0; r < kMaxRounds; ++r) {
            if (r < a.num_rounds) {
                Mat3 Ra;
#pragma unroll
                for (int i = 0; i < 9; ++i) Ra.m[i] = bperm(anc_addr[r], Rg.m[i]);
                const Vec3 da = {bperm(anc_addr[r], pj.x), bperm(anc_addr[r], pj.y), bperm(anc_addr[r], pj.z)};
                if (anc_ok[r]) {
                    pj = mul(Ra, pj) + da;
                    Rg = mul(Ra, Rg);
                }
            }
        }
        // parent's global rotation without cross-lane traffic: Rg = Rgp R  =>  Rgp = Rg R^T
        // (identity at the root; lanes beyond the needed depth hold unused values)
        Mat3 Rgp;
        {
            const float* R = rod.R.m;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = Rg.m[3 * r] * R[3 * c] + Rg.m[3 * r + 1] * R[3 * c + 1] + Rg.m[3 * r + 2] * R[3 * c + 2];
                    Rgp.m[3 * r + c] = has_par ? v : ((r == c) ? 1.f : 0.f);
                }
        }

        K2B_STAMP(6);
        // ---- e. joint loss, its gradient, subtree force / torque sums ------------------------------
        Vec3 gj = {0.f, 0.f, 0.f};
        float part = 0.f;                  // per-lane partial of the joint loss
        if (tk >= 0) {
            const float ex = pj.x + tr.x - tgt.x, ey = pj.y + tr.y - tgt.y, ez = pj.z + tr.z - tgt.z;
            const float x2 = ex * ex, y2 = ey * ey, z2 = ez * ez;
            const float dx = s2 + x2, dy = s2 + y2, dz = s2 + z2;
            if (last) part = wconf * ((s2 * x2) / dx + (s2 * y2) / dy + (s2 * z2) / dz);
            // d gmof / d e = 2 e s^4 / (s^2 + e^2)^2
            const float k2 = 2.f * wconf * (s2 * s2);
            gj = {k2 * ex * fast_rcp(dx * dx), k2 * ey * fast_rcp(dy * dy), k2 * ez * fast_rcp(dz * dz)};
        }
        K2B_STAMP(7);
        // subtree sums of g and p x g.  A subtree is the lane range [t, t + size_t) (DFS order), so its
        // sum is a difference of two inclusive prefix sums.  The two triples ride in the two 32-lane
        // halves (g in lanes t, p x g in lanes 32 + t); the scan is five DPP steps per half-wave in
        // double (no LDS round trip), then ONE round of cross-lane fetches for the range ends.
        float sums[6];
        {
            const Vec3 pxg = cross(pj, gj);
            float w3[3];
            {
                const float lo[3] = {isJ ? gj.x : 0.f, isJ ? gj.y : 0.f, isJ ? gj.z : 0.f};
                const float hi[3] = {isJ ? pxg.x : 0.f, isJ ? pxg.y : 0.f, isJ ? pxg.z : 0.f};
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float x = lo[i], y = hi[i];
                    swap32(x, y);                                   // x: lanes t keep g, lanes 32 + t receive p x g of joint t
                    w3[i] = x;
                }
            }
            float s3[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double scan = half_wave_inclusive_scan(w3[i]);
                const double hi_end = bperm64(sub_end_addr, scan);  // prefix at the last lane of the subtree
                const double lo_end = scan - (double)w3[i];         // prefix just before its first lane (this lane)
                s3[i] = sub_ok ? (float)(hi_end - lo_end) : 0.f;
            }
            // bring the p x g sums back to the joint's own lane
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float x = s3[i], y = s3[i];
                swap32(x, y);                                       // y: lanes t receive the value of lane 32 + t
                sums[i] = s3[i];
                sums[3 + i] = y;
            }
        }
        const Vec3 aj = {sums[0], sums[1], sums[2]};
        const Vec3 tj = {sums[3], sums[4], sums[5]};
        K2B_STAMP(8);
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
        K2B_STAMP(9);
        float gb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int kk = k < NBT ? k : 0;
            gb[k] = (isJ && k < NBT) ? gd.x * dd[0][kk] + gd.y * dd[1][kk] + gd.z * dd[2][kk] : 0.f;
        }
        const float gbeta = butterfly16_sum(gb, lane);   // lanes 4k..4k+3 hold d joint-loss / d beta_k
        const Vec3 groot = {read_lane(aj.x, 0), read_lane(aj.y, 0), read_lane(aj.z, 0)};  // root = lane 0: d/d transl

        K2B_STAMP(10);
        // ---- f. tree layout -> gradient strip (unified: the parameter strip is reused) --------------
        const float jloss = last ? wave_sum_fast(part) : 0.f;
        wave_sync();
        if (isJ) { gs[thoff] = gth.x; gs[thoff + 1] = gth.y; gs[thoff + 2] = gth.z; }
        if ((lane & 3) == 0 && (lane >> 2) < NB) gs[XS_BETA + (lane >> 2)] = gbeta;
        if (lane == 63) { gs[XS_TRANSL] = groot.x; gs[XS_TRANSL + 1] = groot.y; gs[XS_TRANSL + 2] = groot.z; gs[XS - 1] = jloss; }
        }  // do_tree
        if (SPLIT) __syncthreads(); else wave_sync();

        if (do_row) {
        // ---- row layout: gradients of the joint term, then the priors ---------------------------------
        g0 = gs[offA];
        g1 = actB ? gs[offB] : 0.f;
        const float jloss = gs[XS - 1];
        wave_sync();
        float part = 0.f;                  // per-lane partial of the prior losses
        if (bodyA) {
            g0 += wpp2 * yA + 2.f * wpr2 * (x0 - pr0);
            if (angA != 0.f) {
                const float e = __expf(x0 * angA);
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
        if (translB) {
            g1 += 2.f * wt2 * (x1 - tp1);
            if (last) part += wt2 * (x1 - tp1) * (x1 - tp1);
        }
        if (last) loss_total = wave_sum_fast(part) + wpp2 * best + jloss;

        K2B_STAMP(11);
        // ---- g. Adam (torch.optim.Adam, single-tensor path) ------------------------------------------
        const float2 co = a.adam_coef[it];     // {lr / (1 - b1^t), sqrt(1 - b2^t)}
        const float inv_bc2 = fast_rcp(co.y);
        if (optA) {
            m0 = m0 + om_b1 * (g0 - m0);
            v0 = v0 * a.beta2 + om_b2 * g0 * g0;
            const float denom = fast_sqrt(v0) * inv_bc2 + a.eps;
            x0 = x0 - co.x * (m0 * fast_rcp(denom));
        }
        if (optB) {
            m1 = m1 + om_b1 * (g1 - m1);
            v1 = v1 * a.beta2 + om_b2 * g1 * g1;
            const float denom = fast_sqrt(v1) * inv_bc2 + a.eps;
            x1 = x1 - co.x * (m1 * fast_rcp(denom));
        }
        }  // do_row
        K2B_STAMP(12);
    }
#ifdef K2B_FIT_STAMPS
    if (blockIdx.x == 0 && fslot == 0 && lane == 0 && g_k2b_stamps != nullptr)
        for (int i = 0; i < 13; ++i) g_k2b_stamps[(do_row ? 0 : 1) * 16 + i] = stamp_reg[i];
#endif
    if (!do_row || !f_valid) return;

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
        a.grad_out[(size_t)f * P + lane] = optA ? g0 : 0.f;
        if (actB) a.grad_out[(size_t)f * P + 64 + lane] = optB ? g1 : 0.f;
    }
}

#ifdef K2B_FIT_STAMPS
extern "C" int k2b_debug_set_stamp_buffer(void* dev_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k2b_stamps), &dev_ptr, sizeof(void*));
}
#endif

hipError_t launch_fit_world(const FitArgs& a, hipStream_t stream) {
    if (a.num_frames <= 0) return hipSuccess;
    // frames per workgroup: enough to cover the batch with one workgroup per CU
    int fpw = (a.num_frames + a.num_cus - 1) / a.num_cus;
    const bool split = fpw <= MAXW / 2;            // SIMDs would idle: give every frame two cooperating waves
    const int cap = split ? MAXW / 2 : MAXW;
    fpw = fpw < 1 ? 1 : (fpw > cap ? cap : fpw);
    const int blocks = (a.num_frames + fpw - 1) / fpw;
    const dim3 block((split ? 2 * fpw : fpw) * 64);
    if (a.num_betas <= 10) {
        if (split) hipLaunchKernelGGL((k2b_fit_world_kernel<10, true>), dim3(blocks), block, 0, stream, a);
        else hipLaunchKernelGGL((k2b_fit_world_kernel<10, false>), dim3(blocks), block, 0, stream, a);
    } else {
        if (split) hipLaunchKernelGGL((k2b_fit_world_kernel<16, true>), dim3(blocks), block, 0, stream, a);
        else hipLaunchKernelGGL((k2b_fit_world_kernel<16, false>), dim3(blocks), block, 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace k2b
