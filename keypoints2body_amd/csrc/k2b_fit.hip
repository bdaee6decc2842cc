// k2b_fit.hip — fused world-space SMPLify fit for gfx950 (MI355X, CDNA4).
//
// One launch runs ALL Adam iterations of `WorldSpaceFitter.fit_frame`'s Adam branch
// (reference keypoints2body/core/fitters/world_space.py:248-256) for a batch of independent
// frames.  A workgroup is always 8 wavefronts and carries up to 16 frames; nothing is read from or
// written to HBM per iteration.
//
// GMM prior on the matrix cores.  y_m = P_m (theta - mu_m) for the 8 mixture components is a
// (8 x 69 x 69) x (69 x frames) product.  Wave w of the workgroup owns component w: the 64 x 64 core
// of P_w lives in ITS REGISTERS for the whole launch as MFMA A fragments (4 row tiles x K = 32 + 32),
// split into two f16 terms (hi + lo, ~22 mantissa bits, power-of-two scaled per component) so that
// each product is three f16 MFMAs (hi.hi + hi.lo + lo.hi) with fp32 accumulation.  The B operand is
// theta[0..63] of ALL frames of the workgroup (one frame per MFMA column), published by the frames'
// row roles as f16 hi/lo strips in LDS.  Each wave reduces its component's quadratic form per frame in
// the accumulator layout and publishes y and q through LDS.  The rim of P (rows / columns 64..68) stays
// on the vector ALU and uses the symmetry of P: the five rim rows give y_64..68 and, through
// theta_B^T (P_BA d_A), the rim's share of every quadratic form; the rim columns are added to y only
// for the arg-min component.
//
// Lane roles of a wave
//   "row layout"   optimiser state: register set A, lane l <-> body_pose[l] (l = 0..63, the MFMA core
//                  rows); set B, lane l <-> body_pose[64 + l] (l < 5), global_orient (5..7), betas
//                  (8..8+NB-1), transl (next 3).
//   "tree layout"  lane t < 24 owns the t-th joint of the kinematic tree in DFS pre-order, so
//                  every subtree is a contiguous lane range.  Global transforms come from a
//                  pointer-doubling down-sweep (log2(depth) rounds of cross-lane moves instead
//                  of one round per level); subtree force / torque sums are differences of one
//                  prefix scan.  All tree traffic is ds_bpermute / DPP cross-lane moves.
// The layouts exchange values through per-frame LDS strips.
//
// Three execution shapes (template parameter MODE), chosen by frames per CU:
//   split         <= 4 frames per CU: each frame gets TWO waves on one SIMD, the "row" wave (two mixture
//                 components, rim of the prior, Adam; owns the optimiser state) and the "tree" wave
//                 (kinematics, joint loss, analytic backward), in separate specialised loops, so the
//                 critical path of an iteration is the tree alone;
//   split-paired  <= 8: the same, with two frames per row wave and both their trees in one pass of the
//                 tree wave (one tree per 32-lane half);
//   paired        <= 16: wave w does everything for frames 2w and 2w + 1 and carries component w.
// Every wave of the workgroup meets at two barriers per iteration (parameters published /
// gradients, y and q published), in every shape.
//
// Arithmetic restated (see oracle/fit_torch.py for the CPU twin and the reference lines):
//   joints  p_j = p_par + Rg_par (J_j(beta) - J_par(beta)),  Rg_j = Rg_par R_j   (smplx chain)
//   loss    w_j^2 sum_k c_k^2 gmof(p_k + t - y_k) + w_pp^2 min_m(0.5 d_m^T P_m d_m - log nllw_m)
//           + w_a^2 sum exp(s_i th_i)^2 + w_s^2 |beta|^2 + w_pr^2 |th - th_0|^2   (losses.py:49-66)
//   Adam    torch.optim.Adam single-tensor update (torch/optim/adam.py), bias terms from host.
#include <cstdlib>
#include <cstring>

#include "k2b_internal.h"
#include "k2b_lanes.h"
#include "k2b_lbfgs_device.h"

// Per-phase s_memtime stamps of the split shape's two roles live outside this file: tools/build_fit_stamps.sh compiles it with
// -DK2B_FIT_DIAG_HEADER=<tools/fit_diag.h>, which fills the hooks below (iteration 50 of one workgroup, device printf behind the loop).
#ifdef K2B_FIT_DIAG_HEADER
#include K2B_FIT_DIAG_HEADER
#else
#define K2B_FSTAMP_DECL ((void)0)
#define K2B_FSTAMP(i) ((void)0)
#define K2B_FSTAMP_TREE_PRINT ((void)0)
#define K2B_FSTAMP_ROW_PRINT ((void)0)
#endif

namespace k2b {

namespace {

static_assert(kFitJoints == 24, "lane tables are built for the SMPL tree");
constexpr int D = kPriorDim;            // 69
constexpr int MAXW = kFitMaxWaves;      // waves per workgroup
constexpr int MAXS = 16;                // frame slots per workgroup (= MFMA columns)
constexpr int MG = kPriorMaxGauss;      // 8
constexpr int NC = 64;                  // core rows / columns of a component handled by the matrix cores
constexpr int NR = D - NC;              // 5 rim rows / columns
// LDS (floats): rim rows of P as [m][5 rim rows][64 columns] (read transposed - lane = column - for the arg-min component's
// rim columns), then mu | c = P mu of the core rows [m][2][64], then the per-frame-slot blocks:
// parameters (row role -> tree / component roles), gradients (tree -> row), theta[0..63] as f16 hi | lo.
constexpr int RIM_FLOATS = 2 * NR * 64 * 4;
constexpr int CMU_FLOATS = MG * 2 * NC;
constexpr int RF_FLOATS = MG * kPriorRimFragEntries * 4;   // rim rows 64..68 as a fifth MFMA row tile, compact (k2b_api.hip)
static_assert(RIM_FLOATS + CMU_FLOATS + RF_FLOATS == kPriorImageFloats, "host image size");
constexpr int XS = 96;                  // strip: go@0, body@4, betas@76, transl@92, joint loss@95
constexpr int XS_BODY = 4, XS_BETA = 76, XS_TRANSL = 92;
constexpr int SLOT = 2 * XS + NC + 4;   // 260 floats: the +4 spreads the 16 frame columns of the MFMA-side reads over the banks
constexpr int YX_STRIDE = MG * NC + 4;  // y exchange: [slot][m][64] (+4: b128 stores of the 16 frame columns hit distinct banks)
constexpr int DD_STRIDE_MAX = 52;        // per-lane stride of the J_dirs table in LDS (see the kernel)
constexpr int PLO_FLOATS = MG * 4 * 2 * 64 * 4;   // lo fragments of all components (paired shape): [m][tile][ks][64 lanes][8 halfs]
constexpr int WX_FLOATS = MAXS * 64;                 // rim rows' products: [slot][lane 8 m + s] = (P_m[64 + s][0..63] theta)[s < 5], zeros for s >= 5
constexpr int RC_FLOATS = 8 * 64;                    // per-lane rim constants (FitArgs::row_const) for the 16-wave shapes, which have no registers for them
constexpr int LDS_FLOATS = RIM_FLOATS + CMU_FLOATS + RF_FLOATS + MAXS * SLOT + MAXS * MG + MAXS * YX_STRIDE + 64 * DD_STRIDE_MAX + PLO_FLOATS + WX_FLOATS + RC_FLOATS;
static_assert(LDS_FLOATS * 4 <= 163840, "LDS budget");

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// sum over the 8 lanes {l ^ 1, l ^ 2, l ^ 4 ...} of a group of eight: quad_perm twice, then the
// half-row mirror (lane i <-> 7 - i), which after the quad steps delivers the other quad's total
__device__ __forceinline__ float group8_sum(float v) {
    v += lane_xor1(v);
    v += lane_xor2(v);
    v += dpp<0x141>(v);   // row_half_mirror
    return v;
}

// minimum over the wave of a value that is uniform inside each group of eight lanes
__device__ __forceinline__ float group8_wave_min(float v) {
    float r;
    asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    { float a = r, b = r; swap16(a, b); r = vmin(a, b); }
    { float a = r, b = r; swap32(a, b); r = vmin(a, b); }
    return r;
}

// sum over the 32-lane half of the lane
__device__ __forceinline__ float half_sum_fast(float v) {
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

// 16 per-lane values -> lane l ends with the sum over ITS 32-lane half of v[k],
// k = 8 bit4(l) + 4 bit3(l) + 2 bit2(l) + bit1(l).
__device__ __forceinline__ float butterfly16_half_sum(const float (&v)[16], int lane) {
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = v[i], b = v[8 + i];
        swap16(a, b);
        w8[i] = a + b;             // even 16-lane rows: v[i] over the row pair; odd rows: v[8 + i]
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s0 = w8[i] + lane_xor8(w8[i]), s1 = w8[4 + i] + lane_xor8(w8[4 + i]);
        w4[i] = (lane & 8) ? s1 : s0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float t0 = w4[i] + lane_xor4(w4[i]), t1 = w4[2 + i] + lane_xor4(w4[2 + i]);
        w2[i] = (lane & 4) ? t1 : t0;
    }
    const float u0 = w2[0] + lane_xor2(w2[0]), u1 = w2[1] + lane_xor2(w2[1]);
    float r = (lane & 2) ? u1 : u0;
    r += lane_xor1(r);
    return r;
}

}  // namespace

// Round 4: a fourth shape with SIXTEEN waves per workgroup for 9..16 frames per CU (the 4096-frame headline).  A wave on its own
// issues a vector instruction every ~4.5 cycles while its SIMD can take one every ~2.5 from several (MI355X_MICROARCH.md,
// "vector-instruction ISSUE cost"; tools/probe/pk_rate.hip), and every role of this kernel is a dependent chain with LDS round
// trips and cross-lane moves: the `paired` shape's two waves per SIMD left the vector pipe half idle.
//   wide   <= 16 frames per CU: eight ROW waves (one mixture component + the optimiser state of two frames each) and eight TREE
//          waves (two frames each, one tree per 32-lane half) - four waves per SIMD, at most 128 registers per lane.  The row
//          role has its own lean code path below (set B of both frames packed into one register set, lo / rim fragments streamed
//          from LDS tile by tile, a tile consumed under the next tile's products); the tree role is the split-paired one with
//          its J_dirs table in LDS.  4096 frames: 0.457 -> 0.38-0.40 ms.
// (Measured and NOT kept in round 4: `split` with eight row waves of one component each (12 waves: the row waves' component
//  chain is halved but runs in the tree wave's shadow either way - 0.2465 against 0.2428 ms at 1024 frames), and a 16-wave shape
//  with one frame per wave (twice the tree instructions per frame: 0.399 against 0.294 ms at 2048 frames).)
enum { MODE_SPLIT = 0, MODE_SPLIT_PAIRED = 1, MODE_PAIRED = 2, MODE_WIDE = 3, MODE_SPLIT_LBFGS = 4 };   // 4: split + the L-BFGS step as a prologue

template <int MODE> struct Shape {
    static constexpr bool SPLIT = MODE != MODE_PAIRED;
    static constexpr bool PAIR = MODE != MODE_SPLIT && MODE != MODE_SPLIT_LBFGS;
    static constexpr int NROW = MODE == MODE_WIDE ? 8 : 4;                                   // row waves (split shapes)
    static constexpr int NWAVES = SPLIT ? 2 * NROW : MAXW;                                   // as many tree waves as row waves
    static constexpr int CPW = MG / NROW;                                                    // mixture components per row wave
};

// torch.optim.Adam, single-tensor path, for one parameter; the fused operations are pinned so that every code path of the
// kernel rounds alike whatever the compiler would contract in its context (co = {lr / (1 - b1^t), sqrt(1 - b2^t)})
__device__ __forceinline__ void adam_update(float& x, float& m, float& v, float g, float2 co, float inv_bc2, float om_b1, float beta2,
                                            float om_b2, float eps) {
    m = __builtin_fmaf(om_b1, g - m, m);
    v = __builtin_fmaf(om_b2 * g, g, v * beta2);
    const float denom = __builtin_fmaf(fast_sqrt(v), inv_bc2, eps);
    x = __builtin_fmaf(-co.x, m * fast_rcp(denom), x);
}

template <int NBT, int MODE>
__global__ __launch_bounds__(Shape<MODE>::NWAVES * 64) void k2b_fit_world_kernel(const FitArgs a) {
    constexpr bool SPLIT = Shape<MODE>::SPLIT, PAIR = Shape<MODE>::PAIR;
    constexpr int NROW = Shape<MODE>::NROW, CPW = Shape<MODE>::CPW;
    constexpr bool WIDE16 = Shape<MODE>::NWAVES == 16;       // four waves per SIMD: 128 registers per lane
    constexpr int FW = PAIR ? 2 : 1;         // frames a wave carries in its row and tree roles
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ int row_sync_cell;                    // split shape: meeting point of the four row waves

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;               // 8, 12 or 16 waves by shape
    const int M = a.num_gauss;
    const int F = a.frames_per_wg;           // frame slots of this workgroup (split <= 4, split-paired <= 8, paired <= 16)
    int* row_sync = &row_sync_cell;
    if (tid == 0) *row_sync = 0;                 // visible to every wave after the first barrier of the loop
    __shared__ int lb_done_cell[2];               // persistent L-BFGS: slot s's optimiser has finished (written between barriers 3 and 4)
    if (tid < 2) lb_done_cell[tid] = 1;

    // ---- 0. rim of the precisions, mu and c of the core rows -> LDS (shared by the workgroup) ----------
    {
        const float4* src = reinterpret_cast<const float4*>(a.pa_image);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < (RIM_FLOATS + CMU_FLOATS + RF_FLOATS) / 4; i += blockDim.x) dst[i] = src[i];
    }

    // frame slots and roles of this wave
    //   split    waves 0..3 are the row waves of slots 0..3, waves 4..7 their tree waves (waves w and
    //            w + 4 share a SIMD, so every SIMD hosts one row wave and one tree wave);
    //   paired   two frame slots per wave (2w, 2w + 1): the tree of the first lives in lanes 0..31, that of
    //            the second in lanes 32..63; the row work runs once per slot.  Without split, wave w does
    //            both roles for its two slots.
    // (split shape with one or two frames: the tree wave of slot s is wave 4 + ((s + 2) & 3), i.e. it sits on the SIMD of an idle
    //  row slot instead of sharing its own row wave's SIMD - the two roles of a frame are co-critical and issue-bound together)
    constexpr bool SPLIT1 = MODE == MODE_SPLIT || MODE == MODE_SPLIT_LBFGS;     // one frame per row / tree wave
    const bool spread = SPLIT1 && F <= 2 && wave >= 4;
    const int slot0 = (PAIR ? 2 : 1) * (SPLIT ? (spread ? ((wave + 2) & 3) : (wave < NROW ? wave : wave - NROW)) : wave);
    const bool do_row = slot0 < F && (!SPLIT || wave < NROW);     // rim of the prior, priors in row layout, Adam, results
    const bool do_tree = slot0 < F && (!SPLIT || wave >= NROW);   // kinematics, joint loss, analytic backward
    // every wave must reach every barrier: a padding slot recomputes the last frame and skips the final stores
    int f[FW];
    bool f_valid[FW];
#pragma unroll
    for (int h = 0; h < FW; ++h) {
        const int f_raw = blockIdx.x * F + slot0 + h;
        f_valid[h] = slot0 + h < F && f_raw < a.num_frames;
        f[h] = f_raw < a.num_frames ? f_raw : a.num_frames - 1;
    }

    const float* cmu = lds + RIM_FLOATS;                                // [m][mu | c][64]
    const half8* rimfrag = reinterpret_cast<const half8*>(lds + RIM_FLOATS + CMU_FLOATS);   // [m][fragment 4][21 entries]
    float* slots = lds + RIM_FLOATS + CMU_FLOATS + RF_FLOATS;
    for (int i = tid; i < MAXS * SLOT; i += blockDim.x) slots[i] = 0.f;      // (strip tails are read in 16-byte pieces: keep them finite)
    __syncthreads();                                                          // (before the first publish)
    float* qx = slots + MAXS * SLOT;                           // [slot][m]   core part of d^T P_m d
    float* yx = qx + MAXS * MG;                                // [slot][m][64] (stride YX_STRIDE)  core part of y_m, rows 0..63
    // J_dirs differences of every tree lane, [lane][3][NBT] at a stride that spreads the lanes' b128 reads
    // over the banks (36 / 52 floats): 30-48 registers per lane would otherwise stay live across the loop
    constexpr int DDN = 3 * NBT, DDQ = (DDN + 3) / 4, DD_STRIDE = 4 * DDQ + 4;
    static_assert(DD_STRIDE <= DD_STRIDE_MAX, "LDS budget of the J_dirs table");
    float* ddl = yx + MAXS * YX_STRIDE;
    half8* plo = reinterpret_cast<half8*>(ddl + 64 * DD_STRIDE_MAX);   // paired shape: lo fragments, [m][tile][ks][lane]
    float* wx = reinterpret_cast<float*>(plo) + PLO_FLOATS;           // [slot][64]
    float* rcl = wx + WX_FLOATS;                                      // [8][64]: row_const (16-wave shapes)
    constexpr bool RC_IN_LDS = WIDE16;
    if (RC_IN_LDS)
        for (int i = tid; i < RC_FLOATS; i += blockDim.x) rcl[(i & 63) * 8 + (i >> 6)] = a.row_const[i];   // [lane][8]: two 16-byte reads per lane
    constexpr bool DD_IN_LDS = (PAIR && !SPLIT) || WIDE16;
    if (DD_IN_LDS) {
        for (int i = tid; i < 64 * DDN; i += blockDim.x) {
            const int l = i / DDN, r = i % DDN;
            ddl[l * DD_STRIDE + r] = a.dd[(l * 3 + r / NBT) * kMaxBetas + r % NBT];
        }
    }

    const int NB = a.num_betas;
    const int nparamB = 8 + NB + 3;            // lanes of set B that hold a parameter

    // ---- component role: the 64 x 64 core of P_wave as MFMA A fragments, resident in registers ----------
    // accumulator layout of v_mfma_f32_16x16x32_f16: lane (n = l & 15, g = l >> 4), register i of tile t
    // <-> row 16 t + 4 g + i of the component, column n = frame slot n.
    const int cn = lane & 15, cg = lane >> 4;
    const int cslot = cn < F ? cn : F - 1;                     // columns beyond the workgroup's frames repeat the last slot
    const float* cxs = slots + cslot * SLOT;                   // slot this lane's MFMA column reads
    const int ridx = cn < NR ? cg * NR + cn : 4 * NR;          // this lane's entry in a compact rim fragment (as an A operand lane: row cn, k-group cg)
    const _Float16* cth_hi = reinterpret_cast<const _Float16*>(cxs + 2 * XS);
    const _Float16* cth_lo = cth_hi + NC;

    // ---- 1. per-lane constants ----------------------------------------------------------
    // row layout: which parameter the two register sets of this lane hold, and where it sits in the strips
    const bool actB = lane < nparamB;
    const bool bodyB = lane < NR, goB = lane >= NR && lane < 8;
    const bool betaB = lane >= 8 && lane < 8 + NB;
    const bool translB = actB && !bodyB && !goB && !betaB;
    const int offA = XS_BODY + lane;
    const int offB = bodyB ? XS_BODY + NC + lane : (goB ? lane - NR : (betaB ? XS_BETA + (lane - 8) : XS_TRANSL + (lane - 8 - NB)));
    // flat parameter index in [global_orient 3 | body_pose 69 | betas NB | transl 3]
    const int pA = 3 + lane;
    const int pB = bodyB ? 3 + NC + lane : (goB ? lane - NR : (betaB ? 3 + D + (lane - 8) : 3 + D + NB + (lane - 8 - NB)));
    // optimiser membership of this lane's parameters (k2b_fit_config.optimize_mask)
    const bool optA = (a.opt_mask >> 1) & 1;
    const bool optB = actB && ((a.opt_mask >> (bodyB ? 1 : (goB ? 0 : (betaB ? 2 : 3)))) & 1);

    // rim of the prior: lane (gm = l >> 3, gs = l & 7) works on component gm, columns 8 gs .. 8 gs + 7;
    // lanes gs < 5 finish rim row 64 + gs of that component
    const int gm = lane >> 3, gs = lane & 7;
    float pbb_r[NR], cB_r = 0.f, kB_r = 0.f, muB_r = 0.f;
    if (!RC_IN_LDS) {
#pragma unroll
        for (int k = 0; k < NR; ++k) pbb_r[k] = a.row_const[k * 64 + lane];     // P_gm[64 + gs][64 + k]
        cB_r = a.row_const[5 * 64 + lane];    // (P mu)[64 + gs]
        kB_r = a.row_const[6 * 64 + lane];    // (P_BA mu_A)[gs]
        muB_r = a.row_const[7 * 64 + lane];   // mu[64 + gs]
    }

    // angle prior: sign (0 = not a prior index) for the set-A parameter of this lane
    float angA = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane == a.angle_index[i]) angA = a.angle_sign[i];

    // tree layout constants (host tables, DFS pre-order; paired: one tree per 32-lane half)
    const int hb = PAIR ? lane >> 5 : 0;         // which of the wave's frames this lane's tree belongs to
    const int tl = PAIR ? (lane & 31) : lane;    // lane inside that tree
    float* xs_t = slots + (slot0 + hb) * SLOT;   // strips of this lane's tree
    float* gs_t = xs_t + XS;
    const int* lt = a.lane_tab + tl * kLaneTabStride;
    const int joint = lt[0];                    // joint of this lane, -1 beyond the tree
    const bool isJ = joint >= 0;
    // ancestor fetched in doubling round r.  A lane without one fetches lane 31 of its half, which is
    // beyond the tree (24 joints) and therefore holds the identity transform (theta = 0, offset = 0)
    // in every round: composing with it is a no-op, so the rounds need no per-lane select.
    int anc_addr[kMaxRounds];
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) anc_addr[r] = ((lt[2 + r] >= 0 ? lt[2 + r] : 31) + 32 * hb) * 4;
    // subtree lane range of this lane's joint inside its 32-lane half (unpaired: the upper half mirrors
    // the lower one and carries the p x g sums)
    bool sub_ok;
    int sub_end_addr;
    {
        const int* lt2 = a.lane_tab + (lane & 31) * kLaneTabStride;
        const int t = lane & 31, size = lt2[2 + kMaxRounds];      // subtree size in lanes (0 beyond the tree)
        sub_ok = lt2[0] >= 0 && size > 0;
        const int first = (lane & 32) + t, lastl = first + (sub_ok ? size - 1 : 0);
        sub_end_addr = lastl * 4;
    }
    float dt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dt[c] = a.dt[tl * 3 + c];
    // J_dirs differences of this lane: registers in the split shapes (the LDS reads would sit on
    // the tree's critical path), LDS in the paired shape (two frames of optimiser state per wave leave no
    // room for 30-48 more loop-invariant registers)
    // (the 16-wave shapes have 128 registers per lane: LDS there too)
    float ddr[DD_IN_LDS ? 1 : 4 * DDQ];
    if (!DD_IN_LDS) {
#pragma unroll
        for (int r = 0; r < 4 * DDQ; ++r) ddr[r] = r < DDN ? a.dd[(tl * 3 + r / NBT) * kMaxBetas + r % NBT] : 0.f;
    }
    const float4* ddl4 = reinterpret_cast<const float4*>(ddl + tl * DD_STRIDE);
    auto read_dd = [&](float (&dd)[4 * DDQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < DDQ; ++q) {
            if (DD_IN_LDS) {
                const float4 v = ddl4[q];
                dd[4 * q] = v.x; dd[4 * q + 1] = v.y; dd[4 * q + 2] = v.z; dd[4 * q + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dd[4 * q + e] = ddr[DD_IN_LDS ? 0 : 4 * q + e];
            }
        }
    };
    const int jj = isJ ? joint : 0;
    const int thoff = jj == 0 ? 0 : XS_BODY + 3 * (jj - 1);

    // targets: the lane of joint j holds the target of that joint (or none)
    const int tk = isJ ? a.lane_target[jj] : -1;
    Vec3 tgt = {0.f, 0.f, 0.f};
    float wconf = 0.f;  // w_j^2 c^2
    if (tk >= 0) {
        const size_t ft = (size_t)(PAIR ? (hb ? f[FW - 1] : f[0]) : f[0]) * (a.chain_len > 1 ? a.chain_len : 1);
        const float* y = a.j3d + (ft * a.num_targets + tk) * 3;
        tgt = {y[0], y[1], y[2]};
        const float c = a.conf ? a.conf[(a.conf_per_frame ? ft * a.num_targets : 0) + tk] : 1.0f;
        wconf = (a.joint_w * a.joint_w) * (c * c);
    }

    // ---- L-BFGS step as a prologue (MODE_SPLIT_LBFGS; FitArgs::lb_mode): the frame's row wave consumes the previous launch's
    // closure result and writes the point this launch evaluates into the parameter arrays it is about to read.  One launch per
    // round instead of two (a closure launch is ~8 us, a step launch ~5-7 us, and the host paces the 80 launches of a fit).
    // Staging LDS for the history pairs: everything between the y exchange and the rim products (y, J_dirs table, lo fragments,
    // rim products: 116 KiB that nothing has written yet - a barrier separates the step from their first use), shared out over the
    // workgroup's frames (one frame: 116 KiB, four: 29 KiB = 28 pairs of a 128-float parameter vector).
    if constexpr (MODE == MODE_SPLIT_LBFGS) {
        if (a.lb_mode == 1 || a.lb_mode == 2) {
            if (do_row && f_valid[0]) {
                LbfgsArgs la = a.lbv;
                la.finalize = a.lb_mode == 2 ? 1 : 0;
                constexpr int kFree = (MAXS * YX_STRIDE + 64 * DD_STRIDE_MAX + PLO_FLOATS + WX_FLOATS) * 4;
                const int per_wave = (kFree / F) & ~15;
                const int PL = (la.P + 63) / 64 * 64;
                const int head = 2 * la.H * (int)sizeof(double);
                int pairs = (per_wave - head) / (2 * PL * (int)sizeof(float));
                pairs = pairs < 0 ? 0 : pairs;
                unsigned char* stage = reinterpret_cast<unsigned char*>(yx) + wave * per_wave;
                lbfgs_dev::lbfgs_step_frame(la, f[0], lane, stage, pairs);
            }
            // the parameters written above are read back below by other lanes of the same wave (and by nobody else in this launch)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
        }
    }

    // persistent L-BFGS (lb_mode 3): see the tree waves' loop
    const bool lb_loop = MODE == MODE_SPLIT_LBFGS && a.lb_mode == 3;
    bool lb_step = false;
    int lb_frame = 0, lb_pairs = 0;
    unsigned char* lb_stage = nullptr;
    if constexpr (MODE == MODE_SPLIT_LBFGS) {
        if (lb_loop && wave >= NROW && wave - NROW < F) {
            const int sl = wave - NROW;                           // slot whose optimiser this (idle) tree wave is
            lb_frame = blockIdx.x * F + sl;
            lb_step = lb_frame < a.num_frames;
            // staging LDS: the half of the lo-fragment area the split shape never writes (components 0..3 keep theirs in registers)
            constexpr int kFree = PLO_FLOATS * 4 / 2;
            const int per = (kFree / F) & ~15;
            const int P_ = 3 + D + NB + 3, PL = (P_ + 63) / 64 * 64, head = 2 * a.lb_history * (int)sizeof(double);
            lb_pairs = (per - head) / (2 * PL * (int)sizeof(float));
            lb_pairs = lb_pairs < 0 ? 0 : lb_pairs;
            lb_stage = reinterpret_cast<unsigned char*>(plo) + sl * per;
        }
    }

    // ---- 2. parameters and optimiser state (row layout) -----------------------------------
    auto param_ptr = [&](int fr, int p, const float* go, const float* bp, const float* be, const float* tr) -> const float* {
        if (p < 3) return go + (size_t)fr * 3 + p;
        if (p < 3 + D) return bp + (size_t)fr * D + (p - 3);
        if (p < 3 + D + NB) return be + (size_t)fr * NB + (p - 3 - D);
        return tr + (size_t)fr * 3 + (p - 3 - D - NB);
    };
    const float* prsrc = a.preserve ? a.preserve : a.bp_in;
    float x0[FW], x1[FW], pr0[FW], pr1[FW], tp1[FW];
    float m0[FW], v0[FW], m1[FW], v1[FW], g0[FW], g1[FW], loss_total[FW];
#pragma unroll
    for (int h = 0; h < FW; ++h) {
        x0[h] = *param_ptr(f[h], pA, a.go_in, a.bp_in, a.be_in, a.tr_in);
        x1[h] = actB ? *param_ptr(f[h], pB, a.go_in, a.bp_in, a.be_in, a.tr_in) : 0.f;
        pr0[h] = prsrc[(size_t)f[h] * D + lane];
        pr1[h] = bodyB ? prsrc[(size_t)f[h] * D + NC + lane] : 0.f;
        tp1[h] = translB ? a.tr_prior[(size_t)f[h] * 3 + (lane - 8 - NB)] : 0.f;   // centre of the transl prior
        m0[h] = v0[h] = m1[h] = v1[h] = g0[h] = g1[h] = loss_total[h] = 0.f;
    }

    const float s2 = a.sigma * a.sigma;
    const float wpp2 = a.pose_prior_w * a.pose_prior_w;
    const float wa2 = a.angle_w * a.angle_w;
    const float ws2 = a.shape_w * a.shape_w;
    // warm-start chain (a.chain_len > 1, split shape only): every slot is a SEQUENCE; step 0 is its first frame (no
    // preserve term, a.num_iters iterations), the later steps start from the previous step's result, preserve it and
    // run a.chain_iters iterations with a fresh optimiser state (reference api/sequence.py:214-281, world_space.py:159,211,214)
    const bool chain = !PAIR && a.chain_len > 1;
    float wpr2 = chain ? 0.f : a.preserve_w * a.preserve_w;
    const float wt2 = a.transl_prior_w * a.transl_prior_w;
    // set-B coefficients of the priors: gradient c_y y + 2 c_q (x - ref), loss share c_q (x - ref)^2
    const float cyB = bodyB ? wpp2 : 0.f;
    float cqB = bodyB ? wpr2 : (betaB ? ws2 : (translB ? wt2 : 0.f));
    float refB[FW];
#pragma unroll
    for (int h = 0; h < FW; ++h) refB[h] = bodyB ? pr1[h] : (translB ? tp1[h] : 0.f);
    const float om_b1 = a.one_minus_beta1;       // lerp weight float(1 - beta1), formed in double on host
    const float om_b2 = a.one_minus_beta2;       // float(1 - beta2) computed in double on host

    const bool use_gmm = wpp2 != 0.f;   // a zero pose-prior weight (camera stage 1) skips the mixture entirely

    // ---- a. parameters -> staging strips (fp32 for the tree, the rim and the quadratic forms; f16 hi | lo for the MFMA) ----
    auto publish = [&]() __attribute__((always_inline)) {
        {
#pragma unroll
            for (int h = 0; h < FW; ++h) {
                float* xs = slots + (slot0 + h) * SLOT;
                xs[offA] = x0[h];
                if (actB) xs[offB] = x1[h];
                if (use_gmm) {
                    _Float16* th_hi = reinterpret_cast<_Float16*>(xs + 2 * XS);
                    const _Float16 hh = (_Float16)x0[h];
                    th_hi[lane] = hh;
                    th_hi[NC + lane] = (_Float16)(x0[h] - (float)hh);
                }
            }
        }
    };

    // ---- c. GMM prior, 64 x 64 core of one component for every frame slot: three f16 MFMA products ----
    // (small terms first; the results are consumed later in the iteration, so the matrix pipe runs under
    //  the vector work issued in between)
    // (paired shape: the lo fragments are read from LDS each iteration instead of occupying 32 registers
    //  next to two frames of optimiser state)
    auto comp_issue = [&](const half8 (&ph)[4][2], const half8 (&plr)[4][2], floatx4 (&yacc)[5], int lds_comp, int comp) __attribute__((always_inline)) {
        // lds_comp >= 0: the lo fragments of that component are read from LDS instead of registers
        if constexpr (WIDE16) {
            // 128 registers per lane: the lo fragments of a tile are requested one tile ahead (8 registers in flight instead of 32),
            // the rim fragments behind the third tile; tile-major, every tile's own chain in the usual order (bit-identical)
            const half8* plc = plo + (size_t)lds_comp * 4 * 2 * 64 + lane;
            const half8 bh0 = *reinterpret_cast<const half8*>(cth_hi + 8 * cg);
            const half8 bh1 = *reinterpret_cast<const half8*>(cth_hi + 32 + 8 * cg);
            const half8 bl0 = *reinterpret_cast<const half8*>(cth_lo + 8 * cg);
            const half8 bl1 = *reinterpret_cast<const half8*>(cth_lo + 32 + 8 * cg);
            const half8* rf = rimfrag + comp * kPriorRimFragEntries + ridx;
            half8 nl0 = plc[0], nl1 = plc[64];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const half8 l0 = nl0, l1 = nl1;
                if (t < 3) { nl0 = plc[((t + 1) * 2) * 64]; nl1 = plc[((t + 1) * 2 + 1) * 64]; }
                else { nl0 = rf[21]; nl1 = rf[63]; }                       // rl0, rl1
                __builtin_amdgcn_sched_barrier(0);
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, bh0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, bh1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bl0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bl1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bh0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bh1, acc, 0, 0, 0);
                yacc[t] = acc;
                __builtin_amdgcn_sched_barrier(0);
            }
            const half8 rh0 = rf[0], rh1 = rf[42];
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(nl0, bh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(nl1, bh1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bl0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bl1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bh1, acc, 0, 0, 0);
            yacc[4] = acc;
            asm volatile("" ::"v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
            return;
        }
        half8 pl[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) pl[t][ks] = lds_comp >= 0 ? plo[((lds_comp * 4 + t) * 2 + ks) * 64 + lane] : plr[t][ks];
        const half8 bh0 = *reinterpret_cast<const half8*>(cth_hi + 8 * cg);
        const half8 bh1 = *reinterpret_cast<const half8*>(cth_hi + 32 + 8 * cg);
        const half8 bl0 = *reinterpret_cast<const half8*>(cth_lo + 8 * cg);
        const half8 bl1 = *reinterpret_cast<const half8*>(cth_lo + 32 + 8 * cg);
        const half8* rf = rimfrag + comp * kPriorRimFragEntries + ridx;
        const half8 rh0 = rf[0], rl0 = rf[21], rh1 = rf[42], rl1 = rf[63];   // rim rows 64..68 as a fifth row tile (rows 69..79 are zero)
        if constexpr (SPLIT1) {
            // Step-major over the five row tiles: five independent accumulation chains advance together, so a dependent MFMA never
            // waits for its predecessor's result.  The split shapes' row waves carry two components and their chain - not the tree's -
            // is the iteration's critical path at <= 4 frames per CU (stamps: DESIGN.md 4.1): -2.7 % at 1024 frames; in the two-frames-
            // per-wave shapes the same order is +0.8 % (paired) and +2.2 % (split-paired), so they keep the tile-major one.  Every tile's
            // own chain has the same order in both: the results are bit-identical.
            const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[t][0], bh0, z4, 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl0, bh0, z4, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[t][1], bh1, yacc[t], 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl1, bh1, yacc[4], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bl0, yacc[t], 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bl0, yacc[4], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bl1, yacc[t], 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bl1, yacc[4], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bh0, yacc[t], 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bh0, yacc[4], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) yacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bh1, yacc[t], 0, 0, 0);
            yacc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bh1, yacc[4], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[t][0], bh0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl[t][1], bh1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bl0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bl1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bh0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bh1, acc, 0, 0, 0);
                yacc[t] = acc;
            }
            floatx4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl0, bh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl1, bh1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bl0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bl1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bh0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bh1, acc, 0, 0, 0);
            yacc[4] = acc;
        }
        // keep the B fragments live past the last MFMA so that no MFMA destination is allocated on
        // top of its own B operand
        asm volatile("" ::"v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
    };

    auto tree_pass = [&](bool last) __attribute__((always_inline)) {
        // ---- b/d. tree-layout reads, J(beta), Rodrigues ---------------------------------------------
        const Vec3 th = {xs_t[thoff], xs_t[thoff + 1], xs_t[thoff + 2]};
        const float4 tr4 = *reinterpret_cast<const float4*>(xs_t + XS_TRANSL);          // transl | (joint loss slot)
        const Vec3 tr = {tr4.x, tr4.y, tr4.z};
        Vec3 dj;
        {
            float beta[NBT];
#pragma unroll
            for (int q = 0; q < (NBT + 3) / 4; ++q) {      // 16-byte reads; entries beyond NB are zeros (strips cleared at kernel start)
                const float4 b4 = *reinterpret_cast<const float4*>(xs_t + XS_BETA + 4 * q);
                if (4 * q < NBT) beta[4 * q] = b4.x;
                if (4 * q + 1 < NBT) beta[4 * q + 1] = b4.y;
                if (4 * q + 2 < NBT) beta[4 * q + 2] = b4.z;
                if (4 * q + 3 < NBT) beta[4 * q + 3] = b4.w;
            }
            float dd[4 * DDQ];
            read_dd(dd);
            float e[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s = dt[c];
#pragma unroll
                for (int k = 0; k < NBT; ++k) s += dd[c * NBT + k] * beta[k];   // dd is zero for k >= NB
                e[c] = s;
            }
            dj = {e[0], e[1], e[2]};
        }
        const Rodrigues rod = rodrigues_fwd(isJ ? th : Vec3{0.f, 0.f, 0.f});
        // ---- pointer-doubling down-sweep: after round r a lane is composed with 2^(r+1) ancestors ----
        Mat3 Rg = rod.R;                   // becomes the global rotation
        Vec3 pj = dj;                      // becomes the posed joint (without transl)
        auto round = [&](int r) __attribute__((always_inline)) {
            Mat3 Ra;
#pragma unroll
            for (int i = 0; i < 9; ++i) Ra.m[i] = bperm(anc_addr[r], Rg.m[i]);
            const Vec3 da = {bperm(anc_addr[r], pj.x), bperm(anc_addr[r], pj.y), bperm(anc_addr[r], pj.z)};
            pj = mul(Ra, pj) + da;
            Rg = mul(Ra, Rg);
        };
        // straight-line code for the two common depths (22 AMASS targets: 3 rounds; all 24 joints: 4) - a conditional round ends in
        // twelve register copies where its results join the skipped path
        if (a.num_rounds == 3) { round(0); round(1); round(2); }
        else if (a.num_rounds == 4) { round(0); round(1); round(2); round(3); }
        else {
#pragma unroll
            for (int r = 0; r < kMaxRounds; ++r)
                if (r < a.num_rounds) round(r);
        }
        // ---- e. joint loss, its gradient, subtree force / torque sums ------------------------------
        // (no branch on "this lane has a target": wconf = 0 there, so its loss and gradient vanish)
        Vec3 gj;
        float part = 0.f;                  // per-lane partial of the joint loss
        {
            const float ex = pj.x + tr.x - tgt.x, ey = pj.y + tr.y - tgt.y, ez = pj.z + tr.z - tgt.z;
            const float x2 = ex * ex, y2 = ey * ey, z2 = ez * ez;
            const float dx = s2 + x2, dy = s2 + y2, dz = s2 + z2;
            if (last) part = wconf * ((s2 * x2) / dx + (s2 * y2) / dy + (s2 * z2) / dz);
            // d gmof / d e = 2 e s^4 / (s^2 + e^2)^2
            const float k2 = 2.f * wconf * (s2 * s2);
            gj = {k2 * ex * fast_rcp(dx * dx), k2 * ey * fast_rcp(dy * dy), k2 * ez * fast_rcp(dz * dz)};
        }
        // subtree sums of g and p x g.  A subtree is the lane range [t, t + size_t) (DFS order), so its
        // sum is a difference of two inclusive prefix sums: five DPP steps per half-wave in double (no LDS
        // round trip), then ONE round of cross-lane fetches for the range ends.
        float sums[6];
        {
            // (lanes beyond the tree carry no target: wconf = 0 there, so g and p x g are exact zeros without a select - their
            //  transforms are identities over zero offsets, every factor is finite)
            const Vec3 pxg = cross(pj, gj);
            const float lo[3] = {gj.x, gj.y, gj.z};
            const float hi[3] = {pxg.x, pxg.y, pxg.z};
            if (PAIR) {
                // both halves carry a tree: six scans
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float v = i < 3 ? lo[i] : hi[i - 3];
                    const double scan = half_wave_inclusive_scan(v);
                    const double hi_end = bperm64(sub_end_addr, scan);  // prefix at the last lane of the subtree
                    const double lo_end = scan - (double)v;             // prefix just before its first lane (this lane)
                    sums[i] = (float)(hi_end - lo_end);                 // (lanes beyond the tree: window = the lane itself, an exact zero)
                }
            } else {
                // the two triples ride in the two 32-lane halves (g in lanes t, p x g in lanes 32 + t): three scans
                float w3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float x = lo[i], y = hi[i];
                    swap32(x, y);                                   // x: lanes t keep g, lanes 32 + t receive p x g of joint t
                    w3[i] = x;
                }
                float s3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double scan = half_wave_inclusive_scan(w3[i]);
                    const double hi_end = bperm64(sub_end_addr, scan);
                    const double lo_end = scan - (double)w3[i];
                    s3[i] = (float)(hi_end - lo_end);
                }
                // bring the p x g sums back to the joint's own lane
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    float x = s3[i], y = s3[i];
                    swap32(x, y);                                   // y: lanes t receive the value of lane 32 + t
                    sums[i] = s3[i];
                    sums[3 + i] = y;
                }
            }
        }
        const Vec3 aj = {sums[0], sums[1], sums[2]};
        const Vec3 tj = {sums[3], sums[4], sums[5]};
        // torque about this joint of every force below it, and the force itself, expressed in the parent
        // frame: Rg = Rgp R  =>  Rgp^T v = R (Rg^T v)  (at the root Rg = R, so Rgp = I falls out)
        const Vec3 torque = tj - cross(pj, aj);
        const Vec3 w = mul(rod.R, mulT(Rg, torque));
        const Vec3 gd = mul(rod.R, mulT(Rg, aj));     // dL/d(J_j - J_parent)
        // dL/dtheta_j = J_l(theta_j)^T w with the left Jacobian of the rotation vector,
        //   J_l^T w = (sin a / a) w + (1 - sin a / a) u (u.w) - ((1 - cos a) / a) u x w,
        // i.e. the pull-back of the perturbation R -> exp([dphi]x) R with dL = w . dphi.
        Vec3 gth;
        {
            const float a1 = rod.s * rod.inv_angle, a3 = (1.0f - rod.c) * rod.inv_angle;
            const float uw = rod.u.x * w.x + rod.u.y * w.y + rod.u.z * w.z;
            const float a2uw = (1.0f - a1) * uw;
            const Vec3 uxw = cross(rod.u, w);
            gth = {a1 * w.x + a2uw * rod.u.x - a3 * uxw.x, a1 * w.y + a2uw * rod.u.y - a3 * uxw.y, a1 * w.z + a2uw * rod.u.z - a3 * uxw.z};
        }
        float gb[16];
        {
            float dd[4 * DDQ];
            read_dd(dd);
            const Vec3 gz = {isJ ? gd.x : 0.f, isJ ? gd.y : 0.f, isJ ? gd.z : 0.f};     // one select per component instead of one per beta
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int kk = k < NBT ? k : 0;
                gb[k] = k < NBT ? gz.x * dd[kk] + gz.y * dd[NBT + kk] + gz.z * dd[2 * NBT + kk] : 0.f;
            }
        }
        // d joint-loss / d beta_k summed over the tree; which lane ends up with which k: see the helpers
        const float gbeta = PAIR ? butterfly16_half_sum(gb, lane) : butterfly16_sum(gb, lane);
        const int gk = PAIR ? (((lane >> 4) & 1) << 3 | ((lane >> 3) & 1) << 2 | ((lane >> 2) & 1) << 1 | ((lane >> 1) & 1)) : (lane >> 2);
        const bool gk_writer = PAIR ? (lane & 1) == 0 : (lane & 3) == 0;

        // ---- f. tree layout -> gradient strip ---------------------------------------------------------
        const float jloss = last ? (PAIR ? half_sum_fast(part) : wave_sum_fast(part)) : 0.f;
        if (isJ) { gs_t[thoff] = gth.x; gs_t[thoff + 1] = gth.y; gs_t[thoff + 2] = gth.z; }
        if (gk_writer && gk < NB) gs_t[XS_BETA + gk] = gbeta;
        // the root's subtree is the whole tree: its force sum is d/d transl
        if (tl == 0) { gs_t[XS_TRANSL] = aj.x; gs_t[XS_TRANSL + 1] = aj.y; gs_t[XS_TRANSL + 2] = aj.z; gs_t[XS - 1] = jloss; }
    };

    // ---- component role: y = D / scale - c, core part of the quadratic form per frame, publish both ----
    auto comp_consume = [&](const floatx4 (&yacc)[5], int comp) __attribute__((always_inline)) {
        const float inv_scale = a.inv_scale[comp];
        const float* cmu_c = cmu + comp * 2 * NC + 4 * cg;     // + 16 t: mu of this lane's rows; + NC: c
        float qp = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 th4 = *reinterpret_cast<const float4*>(cxs + XS_BODY + 16 * t + 4 * cg);
            const float4 mu4 = *reinterpret_cast<const float4*>(cmu_c + 16 * t);
            const float4 c4 = *reinterpret_cast<const float4*>(cmu_c + NC + 16 * t);
            float4 y;
            y.x = yacc[t][0] * inv_scale - c4.x;
            y.y = yacc[t][1] * inv_scale - c4.y;
            y.z = yacc[t][2] * inv_scale - c4.z;
            y.w = yacc[t][3] * inv_scale - c4.w;
            qp += (th4.x - mu4.x) * y.x + (th4.y - mu4.y) * y.y + (th4.z - mu4.z) * y.z + (th4.w - mu4.w) * y.w;
            // (columns beyond the workgroup's frames repeat the last slot's theta, so they hold the same values:
            //  every lane stores, no predicate)
            *reinterpret_cast<float4*>(yx + cslot * YX_STRIDE + comp * NC + 16 * t + 4 * cg) = y;
        }
        qp = pair_sum32(qp);
        qp = pair_sum16(qp);                             // summed over the four row groups g: all four hold the total
        qx[cslot * MG + comp] = qp;
        // rim rows: accumulator rows 64 + 4 g + i; g = 0 holds rows 64..67, g = 1 row 68 and three zero rows
        if (cg < 2) {
            float4 w;
            w.x = yacc[4][0] * inv_scale; w.y = yacc[4][1] * inv_scale; w.z = yacc[4][2] * inv_scale; w.w = yacc[4][3] * inv_scale;
            *reinterpret_cast<float4*>(wx + cslot * 64 + comp * 8 + 4 * cg) = w;
        }
    };

    // ---- row role, part 1 (needs y and q of every component, not the tree): arg-min component, its y in
    // row layout, and every prior's share of the gradient and of the loss ---------------------------------
    // rim rows 64..68 (lane (gm, gs < 5) finishes row 64 + gs of component gm; w = P_BA theta_A from the matrix cores):
    //   y_B = w + P_BB theta_B - c_B
    //   rim share of d^T P d = theta_B (P_BA d_A) + d_B y_B,   P_BA d_A = w - P_BA mu_A
    const int tBoff = XS_BODY + NC + (gs < NR ? gs : 0);
    float gp0[FW], gp1[FW], lossp[FW], bestv[FW];
    auto row_prior = [&](bool last) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < FW; ++h) {
        const int slot = slot0 + h;
        const float* xs = slots + slot * SLOT;
        float yA = 0.f, yBs = 0.f, best = 0.f;
        if (use_gmm) {
            const float4 t64 = *reinterpret_cast<const float4*>(xs + XS_BODY + NC);      // theta_B[0..3]
            const float t68 = xs[XS_BODY + NC + 4];
            float yBv, qrim;
            {
                float pbb[NR], cB, kB, muB;
#pragma unroll
                for (int k = 0; k < NR; ++k) pbb[k] = RC_IN_LDS ? rcl[lane * 8 + k] : pbb_r[k];
                cB = RC_IN_LDS ? rcl[lane * 8 + 5] : cB_r;
                kB = RC_IN_LDS ? rcl[lane * 8 + 6] : kB_r;
                muB = RC_IN_LDS ? rcl[lane * 8 + 7] : muB_r;
                const float wm = wx[slot * 64 + lane], tB = xs[tBoff];
                const float vB = pbb[0] * t64.x + pbb[1] * t64.y + pbb[2] * t64.z + pbb[3] * t64.w + pbb[4] * t68;
                const float y = wm + vB - cB;
                const float term = tB * (wm - kB) + (tB - muB) * y;
                yBv = gs < NR ? y : 0.f;
                qrim = group8_sum(gs < NR ? term : 0.f);         // every lane of group gm: rim share of component gm
            }
            const float q = qrim + qx[slot * MG + gm];           // lanes 8m..8m+7: d^T P d of component m
            float val = 0.5f * q + a.neg_log_nllw[gm < M ? gm : 0];
            val = gm < M ? val : __builtin_inff();               // (components beyond M never win)
            // arg-min over the eight lane groups: a wave-wide minimum (one DPP step inside the rows of 16, two permlane
            // swaps across them), then the lowest lane that holds it - the first minimum wins, as torch.min and the
            // eight-readlane compare chain this replaces (55 instructions per frame against 14)
            best = group8_wave_min(val);
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(val == best);
            const int mstar = hit ? (int)(__builtin_ctzll(hit) >> 3) : 0;
            // rows 0..63: core part from the component wave + the rim columns, P[l][64 + c] = P[64 + c][l]
            yA = yx[slot * YX_STRIDE + mstar * NC + lane];
            const float* rimf = lds + mstar * (NR * NC) + lane;
            yA += rimf[0 * NC] * t64.x + rimf[1 * NC] * t64.y + rimf[2 * NC] * t64.z + rimf[3 * NC] * t64.w + rimf[4 * NC] * t68;
            yBs = bperm((8 * mstar + (lane < NR ? lane : 0)) * 4, yBv);     // rows 64 + lane for lanes < 5
        }
        // Branch-free: every lane evaluates  c_y y + 2 c_q (x - ref)  with its own coefficients (set A: mixture
        // + preserve; set B: the same for body_pose[64..68], shape prior for betas, transl prior for transl,
        // zeros elsewhere), and the angle prior's exp() runs with sign 0 (-> 1, times 0) outside its four lanes.
        const float dA = x0[h] - pr0[h], dB = x1[h] - refB[h];
        const float eA = __expf(x0[h] * angA);
        const float ga = wpp2 * yA + 2.f * wpr2 * dA + (wa2 * 2.f * angA) * (eA * eA);
        const float gb = cyB * yBs + 2.f * cqB * dB;
        float part = 0.f;                  // per-lane partial of the prior losses
        if (last) part = wpr2 * dA * dA + (angA != 0.f ? wa2 * eA * eA : 0.f) + cqB * dB * dB;
        gp0[h] = ga; gp1[h] = gb; lossp[h] = part; bestv[h] = best;
        }  // frames of this wave
    };
    // ---- row role, part 2 (after the tree): add the joint term's gradient, Adam ------------------------------
    auto row_finish = [&](int it, bool last) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < FW; ++h) {
        const float* gs_r = slots + (slot0 + h) * SLOT + XS;
        // (a parameter that is not optimised gets gradient 0: Adam then leaves it and its state untouched, so the
        //  update below needs no branch)
        g0[h] = optA ? gs_r[offA] + gp0[h] : 0.f;
        g1[h] = optB ? gs_r[offB] + gp1[h] : 0.f;
        if (last) loss_total[h] = wave_sum_fast(lossp[h]) + wpp2 * bestv[h] + gs_r[XS - 1];

        // torch.optim.Adam, single-tensor path
        const float2 co = a.adam_coef[it];     // {lr / (1 - b1^t), sqrt(1 - b2^t)}
        const float inv_bc2 = fast_rcp(co.y);
        adam_update(x0[h], m0[h], v0[h], g0[h], co, inv_bc2, om_b1, a.beta2, om_b2, a.eps);
        adam_update(x1[h], m1[h], v1[h], g1[h], co, inv_bc2, om_b1, a.beta2, om_b2, a.eps);
        }  // frames of this wave
    };

    auto load_frags = [&](int comp, half8 (&ph)[4][2], half8 (&pl)[4][2]) __attribute__((always_inline)) {
        const half8* fi = reinterpret_cast<const half8*>(a.pa_frag32) + (size_t)comp * 4 * 4 * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ph[t][0] = fi[(t * 4 + 0) * 64 + lane];
            ph[t][1] = fi[(t * 4 + 1) * 64 + lane];
            pl[t][0] = fi[(t * 4 + 2) * 64 + lane];
            pl[t][1] = fi[(t * 4 + 3) * 64 + lane];
        }
    };

    // Two workgroup barriers per iteration in every role: [parameters published] and [gradients, y, q
    // published].  The roles of the split shape run their own loops, so that the tree waves carry no
    // component registers and the row waves no tree registers.
    if (SPLIT && wave >= NROW) {
        // tree waves: kinematics, joint loss, analytic backward.  They are the iteration's critical path and share their SIMD
        // with a row wave: raised priority, so that the row wave (and its MFMAs, which hold the SIMD for 16 cycles each) fills
        // the tree wave's stalls instead of competing for its issue slots
        // (wide shape, same-box A/B at 4096 frames: tree waves 3 / row waves 0 0.3815 ms; rows above trees 0.417; all equal 0.417 -
        //  the row waves issue few instructions and are starved either way, the tree waves are what the SIMD must keep fed)
        __builtin_amdgcn_s_setprio(3);
        const int steps = PAIR ? 1 : a.chain_len;
        // persistent L-BFGS (lb_mode 3, at most two frames per workgroup): the IDLE tree wave 4 + s is slot s's optimiser, its state -
        // 27 scalars, eight vectors, the history in LDS by ring slot - RESIDENT in registers and LDS for the whole fit (the state
        // arrays in global memory are only written at the end).  Between two more barriers per iteration it consumes the closure
        // result the row wave has just written (loss, gradient: global memory, same CU) and puts the next point into the parameter
        // arrays, which the row wave then reads back: a round is one iteration of this loop (~2.5 us) + ~1 us of optimiser, no
        // launch.  Behind the last step every frame parks at its accepted point.  (The tree pass keeps its ONE call site below: a
        // second inlined copy is contracted differently by the compiler and the closures stop being bit-identical.)
        LbfgsArgs la{};
        if constexpr (MODE == MODE_SPLIT_LBFGS) {
            if (lb_loop) la = a.lbv;
        }
        la.finalize = 0;
        lbfgs_dev::Frame fr(la);                                        // (unbound: touches no memory; bound below where an optimiser runs)
        if constexpr (MODE == MODE_SPLIT_LBFGS) {
            if (lb_loop) {
                double* lds_al = reinterpret_cast<double*>(lb_stage);
                float* lds_hist = reinterpret_cast<float*>(lb_stage + (size_t)2 * la.H * sizeof(double));
                fr.bind(lb_step ? lb_frame : 0, lane, lds_hist, lds_al, lb_pairs);
                fr.resident = true;
#pragma unroll
                for (int w = 0; w < lbfgs_dev::SV_HIST; ++w)
#pragma unroll
                    for (int k = 0; k < lbfgs_dev::EPL; ++k) fr.V[w][k] = 0.f;
            }
        }
        for (int step = 0; step < steps; ++step) {
            const int nit = step == 0 ? a.num_iters : a.chain_iters;
            if constexpr (MODE == MODE_SPLIT_LBFGS) {
                if (lb_loop && a.chain_len > 1) {             // sequence chain: a fresh optimiser per frame, on the frame's own parameter row
                    fr.restart(lb_frame * a.chain_len + step);
                    la.max_iter = step == 0 ? a.lbv.max_iter : a.lb_chain_max_iter;
                    la.max_eval = la.max_iter * 5 / 4;
                }
            }
            if (!PAIR && step > 0 && tk >= 0) {           // targets of this step's frame (sequence s, frame row s chain_len + step)
                const size_t ft = (size_t)f[0] * a.chain_len + step;
                const float* y = a.j3d + (ft * a.num_targets + tk) * 3;
                tgt = {y[0], y[1], y[2]};
                const float c = a.conf ? a.conf[(a.conf_per_frame ? ft * a.num_targets : 0) + tk] : 1.0f;
                wconf = (a.joint_w * a.joint_w) * (c * c);
            }
            K2B_FSTAMP_DECL;
            for (int it = 0; it < nit; ++it) {
                K2B_FSTAMP(0);
                __syncthreads();
                K2B_FSTAMP(1);
                if (do_tree) tree_pass(it == nit - 1 || lb_loop);
                K2B_FSTAMP(2);
                __syncthreads();
                K2B_FSTAMP(3);
                if constexpr (MODE == MODE_SPLIT_LBFGS) {
                    if (lb_loop) {
                        __syncthreads();
                        if (lb_step && it < nit - 1) {
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                            // (the closure's loss is a wave-uniform = SCALAR load, and the scalar cache is not coherent with the row wave's
                            //  vector store of a round ago: drop it)
                            __builtin_amdgcn_s_dcache_inv();
                            if (fr.s.phase != lbfgs_dev::PH_DONE) {
                                const float* gsrc = la.grad_in + (size_t)lb_frame * la.P;
#pragma unroll
                                for (int k = 0; k < lbfgs_dev::EPL; ++k) fr.GN[k] = fr.has(k) ? gsrc[lane + 64 * k] : 0.f;
                                const float* lp = la.loss_in + lb_frame;
                                asm volatile("" : "+v"(lp));      // (a per-lane address: a VECTOR load, through the L1 the row wave's store went through)
                                lbfgs_dev::lbfgs_consume(fr, (double)*lp);
                            }
                            // rounds ran out inside a line search: back to the accepted point (a finished frame parked itself)
                            if (it == nit - 2 && fr.s.phase != lbfgs_dev::PH_INIT && fr.s.phase != lbfgs_dev::PH_DONE) fr.park();
                            if (lane == 0) lb_done_cell[wave - NROW] = fr.s.phase == lbfgs_dev::PH_DONE ? 1 : 0;
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        }
                        __syncthreads();
                        // every optimiser of the workgroup is done: the closures left would evaluate the same points again - skip to
                        // the final one (the flags are read by every wave behind the same barrier: uniform)
                        const bool all_done = lb_done_cell[0] != 0 && lb_done_cell[1] != 0;
                        if (lb_step && (it == nit - 2 || (all_done && it < nit - 2))) fr.save();
                        if (all_done && it < nit - 2) it = nit - 2;
                    }
                }
            }
            K2B_FSTAMP_TREE_PRINT;
        }
        return;
    }
    if constexpr (MODE == MODE_WIDE) {
        // ---- row waves of the 16-wave shape: component `wave`, optimiser state of slots 2 wave and 2 wave + 1 -------------------
        // 128 registers per lane (four waves per SIMD), so this path is written for short live ranges instead of sharing the
        // lambdas above: set B of BOTH frames is packed into one register set (frame h in lanes 32 h + k: 19 of 32 lanes used
        // instead of 19 of 64 twice - one Adam update, one publish), the lo fragments and the rim fragments stream from LDS one
        // tile at a time, and a tile is consumed while the next one's products run.  Every value is computed by the same
        // operations in the same order as in the other shapes: results are bit-identical.
        const int hq = lane >> 5, kq = lane & 31;                      // packed set B: frame slot0 + hq, parameter kq
        const bool actQ = kq < nparamB, bodyQ = kq < NR, goQ = kq >= NR && kq < 8, betaQ = kq >= 8 && kq < 8 + NB;
        const bool translQ = actQ && !bodyQ && !goQ && !betaQ;
        const int offQ = bodyQ ? XS_BODY + NC + kq : (goQ ? kq - NR : (betaQ ? XS_BETA + (kq - 8) : XS_TRANSL + (kq - 8 - NB)));
        const int pQ = bodyQ ? 3 + NC + kq : (goQ ? kq - NR : (betaQ ? 3 + D + (kq - 8) : 3 + D + NB + (kq - 8 - NB)));
        const bool optQ = actQ && ((a.opt_mask >> (bodyQ ? 1 : (goQ ? 0 : (betaQ ? 2 : 3)))) & 1);
        const int fQ = hq ? f[1] : f[0];
        float xa[2], ma[2], va[2], pra[2];                             // set A of the two frames
        float xq, mq = 0.f, vq = 0.f, refQ;                            // set B, packed
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xa[h] = a.bp_in[(size_t)f[h] * D + lane];
            pra[h] = prsrc[(size_t)f[h] * D + lane];
            ma[h] = va[h] = 0.f;
        }
        xq = actQ ? *param_ptr(fQ, pQ, a.go_in, a.bp_in, a.be_in, a.tr_in) : 0.f;
        refQ = bodyQ ? prsrc[(size_t)fQ * D + NC + kq] : (translQ ? a.tr_prior[(size_t)fQ * 3 + (kq - 8 - NB)] : 0.f);
        const float cyQ = bodyQ ? wpp2 : 0.f;
        const float cqQ = bodyQ ? wpr2 : (betaQ ? ws2 : (translQ ? wt2 : 0.f));
        float* const xsQ = slots + (slot0 + hq) * SLOT;                // this lane's set-B strips
        const bool rowQ = slot0 + hq < F;                              // (a workgroup's last wave may carry one frame only)
        half8 ph[4][2];
        {
            half8 plq[4][2];
            load_frags(wave, ph, plq);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) plo[((wave * 4 + t) * 2 + ks) * 64 + lane] = plq[t][ks];
        }
        const half8* const plc = plo + (size_t)wave * 4 * 2 * 64 + lane;
        const half8* const rf = rimfrag + wave * kPriorRimFragEntries + ridx;
        const float inv_scale = a.inv_scale[wave];
        const float* const cmu_c = cmu + wave * 2 * NC + 4 * cg;
        float g0o[2] = {0.f, 0.f}, gqo = 0.f, losso[2] = {0.f, 0.f};
        K2B_FSTAMP_DECL;
        for (int it = 0; it < a.num_iters; ++it) {
            const bool last = it == a.num_iters - 1;
            const float2 co = a.adam_coef[it];         // requested a whole iteration ahead of its use (behind barrier 2 it was a
                                                       // global load with a full wait on the iteration's critical path)
            K2B_FSTAMP(0);
            // ---- publish -------------------------------------------------------------------------------------------------------
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (slot0 + h < F) {
                    float* xs = slots + (slot0 + h) * SLOT;
                    xs[offA] = xa[h];
                    if (use_gmm) {
                        _Float16* th_hi = reinterpret_cast<_Float16*>(xs + 2 * XS);
                        const _Float16 hh = (_Float16)xa[h];
                        th_hi[lane] = hh;
                        th_hi[NC + lane] = (_Float16)(xa[h] - (float)hh);
                    }
                }
            }
            if (actQ && rowQ) xsQ[offQ] = xq;
            K2B_FSTAMP(1);
            __syncthreads();
            K2B_FSTAMP(2);
            // ---- component `wave` for all 16 slots: products tile by tile, tile t - 1 consumed under the products of tile t ----
            if (use_gmm) {
                const half8 bh0 = *reinterpret_cast<const half8*>(cth_hi + 8 * cg);
                const half8 bh1 = *reinterpret_cast<const half8*>(cth_hi + 32 + 8 * cg);
                const half8 bl0 = *reinterpret_cast<const half8*>(cth_lo + 8 * cg);
                const half8 bl1 = *reinterpret_cast<const half8*>(cth_lo + 32 + 8 * cg);
                float qp = 0.f;
                floatx4 accp = {0.f, 0.f, 0.f, 0.f};
                auto consume_tile = [&](int t, const floatx4& acc) __attribute__((always_inline)) {
                    const float4 th4 = *reinterpret_cast<const float4*>(cxs + XS_BODY + 16 * t + 4 * cg);
                    const float4 mu4 = *reinterpret_cast<const float4*>(cmu_c + 16 * t);
                    const float4 c4 = *reinterpret_cast<const float4*>(cmu_c + NC + 16 * t);
                    float4 y;
                    y.x = acc[0] * inv_scale - c4.x;
                    y.y = acc[1] * inv_scale - c4.y;
                    y.z = acc[2] * inv_scale - c4.z;
                    y.w = acc[3] * inv_scale - c4.w;
                    qp += (th4.x - mu4.x) * y.x + (th4.y - mu4.y) * y.y + (th4.z - mu4.z) * y.z + (th4.w - mu4.w) * y.w;
                    *reinterpret_cast<float4*>(yx + cslot * YX_STRIDE + wave * NC + 16 * t + 4 * cg) = y;
                };
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const half8 l0 = plc[(2 * t) * 64], l1 = plc[(2 * t + 1) * 64];
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, bh1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bl0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bl1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][0], bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph[t][1], bh1, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t > 0) consume_tile(t - 1, accp);
                    __builtin_amdgcn_sched_barrier(0);
                    accp = acc;
                }
                {
                    const half8 rh0 = rf[0], rl0 = rf[21], rh1 = rf[42], rl1 = rf[63];
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl0, bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rl1, bh1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bl0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bl1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh0, bh0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(rh1, bh1, acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    consume_tile(3, accp);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" ::"v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1));
                    qp = pair_sum32(qp);
                    qp = pair_sum16(qp);
                    qx[cslot * MG + wave] = qp;
                    if (cg < 2) {
                        float4 w;
                        w.x = acc[0] * inv_scale; w.y = acc[1] * inv_scale; w.z = acc[2] * inv_scale; w.w = acc[3] * inv_scale;
                        *reinterpret_cast<float4*>(wx + cslot * 64 + wave * 8 + 4 * cg) = w;
                    }
                }
                K2B_FSTAMP(3);
                K2B_FSTAMP(4);
                if (lane == 0) __hip_atomic_fetch_add(row_sync, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int target = NROW * (it + 1);
                while (__hip_atomic_load(row_sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
            }
            K2B_FSTAMP(5);
            // ---- priors: per frame in the (component, rim row) lane layout, then the set-A gradient; set B once for both --------
            float gpa[2] = {0.f, 0.f}, lossp[2] = {0.f, 0.f}, bestv[2] = {0.f, 0.f};
            float yQ = 0.f;
            // per-lane rim constants (P_BB row, c_B, P_BA mu_A, mu_B): two 16-byte LDS reads per iteration, for both frames
            const float4 rc0 = *reinterpret_cast<const float4*>(rcl + lane * 8), rc1 = *reinterpret_cast<const float4*>(rcl + lane * 8 + 4);
            // (both frames unconditionally - a wave's second slot beyond the workgroup's frames reads strips nobody wrote and its
            //  results are dropped below: without the branch the two frames' chains of LDS round trips interleave, 4096 frames
            //  0.3610 against 0.3680 ms; hoisting the reads and terms that do not hang on the meeting above it, or dropping the same
            //  branch from the Adam and publish loops, adds nothing)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#ifdef K2B_WIDE_PRIORS_BRANCH
                if (slot0 + h >= F) continue;      // (A/B build: the form before this change)
#endif
                const int slot = slot0 + h;
                const float* xs = slots + slot * SLOT;
                float yA = 0.f, best = 0.f;
                if (use_gmm) {
                    const float4 t64 = *reinterpret_cast<const float4*>(xs + XS_BODY + NC);
                    const float t68 = xs[XS_BODY + NC + 4];
                    float yBv, qrim;
                    {
                        const float wm = wx[slot * 64 + lane], tB = xs[tBoff];
                        const float vB = rc0.x * t64.x + rc0.y * t64.y + rc0.z * t64.z + rc0.w * t64.w + rc1.x * t68;
                        const float y = wm + vB - rc1.y;
                        const float term = tB * (wm - rc1.z) + (tB - rc1.w) * y;
                        yBv = gs < NR ? y : 0.f;
                        qrim = group8_sum(gs < NR ? term : 0.f);
                    }
                    const float q = qrim + qx[slot * MG + gm];
                    float val = 0.5f * q + a.neg_log_nllw[gm < M ? gm : 0];
                    val = gm < M ? val : __builtin_inff();
                    best = group8_wave_min(val);
                    const unsigned long long hit = __builtin_amdgcn_ballot_w64(val == best);
                    const int mstar = hit ? (int)(__builtin_ctzll(hit) >> 3) : 0;
                    yA = yx[slot * YX_STRIDE + mstar * NC + lane];
                    const float* rimf = lds + mstar * (NR * NC) + lane;
                    yA += rimf[0 * NC] * t64.x + rimf[1 * NC] * t64.y + rimf[2 * NC] * t64.z + rimf[3 * NC] * t64.w + rimf[4 * NC] * t68;
                    const float yb = bperm((8 * mstar + (kq < NR ? kq : 0)) * 4, yBv);      // rows 64 + kq, for both halves
                    yQ = hq == h ? yb : yQ;
                }
                const float dA = xa[h] - pra[h];
                const float eA = __expf(xa[h] * angA);
                gpa[h] = wpp2 * yA + 2.f * wpr2 * dA + (wa2 * 2.f * angA) * (eA * eA);
                if (last) lossp[h] = wpr2 * dA * dA + (angA != 0.f ? wa2 * eA * eA : 0.f);
                bestv[h] = best;
            }
            const float dQ = xq - refQ;
            const float gpq = cyQ * yQ + 2.f * cqQ * dQ;
            if (last) {
                // the set-B share of a frame's prior loss joins the partial of the lane it sits on in the other shapes (lane kq)
                float pq0 = cqQ * dQ * dQ, pq1 = pq0;
                swap32(pq0, pq1);                      // pq1: lanes kq < 32 receive the value of lane 32 + kq (frame 1)
                lossp[0] += hq == 0 ? cqQ * dQ * dQ : 0.f;
                lossp[1] += hq == 0 ? pq1 : 0.f;
            }
            K2B_FSTAMP(6);
            __syncthreads();
            K2B_FSTAMP(7);
            // ---- joint gradient from the tree waves, Adam ------------------------------------------------------------------------
            const float inv_bc2 = fast_rcp(co.y);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (slot0 + h >= F) continue;
                const float* gs_r = slots + (slot0 + h) * SLOT + XS;
                const float g = optA ? gs_r[offA] + gpa[h] : 0.f;
                if (last) { g0o[h] = g; losso[h] = wave_sum_fast(lossp[h]) + wpp2 * bestv[h] + gs_r[XS - 1]; }
                adam_update(xa[h], ma[h], va[h], g, co, inv_bc2, om_b1, a.beta2, om_b2, a.eps);
            }
            {
                const float g = (optQ && rowQ) ? xsQ[XS + offQ] + gpq : 0.f;
                if (last) gqo = g;
                adam_update(xq, mq, vq, g, co, inv_bc2, om_b1, a.beta2, om_b2, a.eps);
            }
            K2B_FSTAMP(8);
        }
        K2B_FSTAMP_ROW_PRINT;
        // ---- results -------------------------------------------------------------------------------------------------------------
        // (frame numbers and addresses are formed again HERE, from a lane number the compiler cannot trace back: kept from the
        //  prologue they are a dozen 64-bit values that live in scratch across the whole loop)
        int lane_r = lane;
        asm volatile("" : "+v"(lane_r));
        const int P = 3 + D + NB + 3;
        const int hq_r = lane_r >> 5, kq_r = lane_r & 31;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int fr = blockIdx.x * F + slot0 + h;
            if (slot0 + h >= F || fr >= a.num_frames) continue;
            a.bp_out[(size_t)fr * D + lane_r] = xa[h];
            if (lane_r == 0 && a.loss_out) a.loss_out[fr] = losso[h];
            if (a.grad_out) a.grad_out[(size_t)fr * P + 3 + lane_r] = g0o[h];
        }
        {
            const int fr = blockIdx.x * F + slot0 + hq_r;
            const bool body = kq_r < NR, go = kq_r >= NR && kq_r < 8, beta = kq_r >= 8 && kq_r < 8 + NB;
            const int pq = body ? 3 + NC + kq_r : (go ? kq_r - NR : (beta ? 3 + D + (kq_r - 8) : 3 + D + NB + (kq_r - 8 - NB)));
            if (kq_r < nparamB && slot0 + hq_r < F && fr < a.num_frames) {
                float* dst = pq < 3 ? a.go_out + (size_t)fr * 3 + pq
                           : (pq < 3 + D ? a.bp_out + (size_t)fr * D + (pq - 3)
                           : (pq < 3 + D + NB ? a.be_out + (size_t)fr * NB + (pq - 3 - D) : a.tr_out + (size_t)fr * 3 + (pq - 3 - D - NB)));
                *dst = xq;
                if (a.grad_out) a.grad_out[(size_t)fr * P + pq] = gqo;
            }
        }
        return;
    }
    if (SPLIT) {
        // row waves: optimiser state of slot `wave`, mixture components `wave` and `wave + 4`
        // (the lo fragments of the second component are parked in LDS: 96 instead of 128 resident registers,
        //  which keeps this loop free of scratch spills)
        //  With eight row waves (round 4) a row wave carries ONE component: both halves in registers at 12 waves per CU (168
        //  registers), the lo half in LDS at 16 (128).
        constexpr bool LO_A_IN_LDS = PAIR || CPW == 1;
        half8 pa_h[4][2], pa_l[4][2], pb_h[CPW == 2 ? 4 : 1][2], pb_l[CPW == 2 ? 4 : 1][2];
        load_frags(wave, pa_h, pa_l);
        if constexpr (CPW == 2) load_frags(wave + 4, pb_h, pb_l);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (CPW == 2) plo[(((wave + 4) * 4 + t) * 2 + ks) * 64 + lane] = pb_l[t][ks];
                if (LO_A_IN_LDS) plo[((wave * 4 + t) * 2 + ks) * 64 + lane] = pa_l[t][ks];   // two frames of optimiser state: both
            }
        auto out_ptr_c = [&](size_t fr, int p) -> float* {
            if (p < 3) return a.go_out + fr * 3 + p;
            if (p < 3 + D) return a.bp_out + fr * D + (p - 3);
            if (p < 3 + D + NB) return a.be_out + fr * NB + (p - 3 - D);
            return a.tr_out + fr * 3 + (p - 3 - D - NB);
        };
        int git = 0;                 // iterations over all chain steps (the row waves' meeting counter runs on)
        const int steps = PAIR ? 1 : a.chain_len;
        for (int step = 0; step < steps; ++step) {
            const int nit = step == 0 ? a.num_iters : a.chain_iters;
            if (!PAIR && step > 0) { // (chain: FW == 1) start from the previous result, preserve it, fresh Adam state
                wpr2 = a.preserve_w * a.preserve_w;
                cqB = bodyB ? wpr2 : cqB;
                pr0[0] = x0[0];
                if (bodyB) { pr1[0] = x1[0]; refB[0] = x1[0]; }
                m0[0] = v0[0] = m1[0] = v1[0] = 0.f;
            }
            if constexpr (MODE == MODE_SPLIT_LBFGS) {
                if (lb_loop && chain && do_row && f_valid[0]) {   // the frame's start point into ITS row of the parameter arrays: the
                    const size_t fr_row = (size_t)f[0] * a.chain_len + step;     // optimiser reads it there (phase INIT) and moves it
                    *out_ptr_c(fr_row, pA) = x0[0];
                    if (actB) *out_ptr_c(fr_row, pB) = x1[0];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
        K2B_FSTAMP_DECL;
        for (int it = 0; it < nit; ++it, ++git) {
            const bool last = it == nit - 1;
            K2B_FSTAMP(0);
            if (do_row) publish();
            K2B_FSTAMP(1);
            __syncthreads();
            K2B_FSTAMP(2);
            floatx4 ya[5], yb[5];
            if (use_gmm) {
                comp_issue(pa_h, pa_l, ya, LO_A_IN_LDS ? wave : -1, wave);
                if constexpr (CPW == 2) comp_issue(pb_h, pb_l, yb, wave + 4, wave + 4);
            }
            K2B_FSTAMP(3);
            if (use_gmm) {
                comp_consume(ya, wave);
                if constexpr (CPW == 2) comp_consume(yb, wave + 4);
                // the four row waves hold all eight components between them: they meet at an LDS counter
                // (release after their y / q writes, acquire before reading the others') so that the
                // arg-min and the priors' gradient are done while the tree waves still work, and only
                // "add the joint gradient, Adam, publish" is left on the iteration's critical path
                K2B_FSTAMP(4);
                if (lane == 0) __hip_atomic_fetch_add(row_sync, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int target = NROW * (git + 1);
                while (__hip_atomic_load(row_sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
            }
            K2B_FSTAMP(5);
            if (do_row) row_prior(last || lb_loop);
            K2B_FSTAMP(6);
            __syncthreads();
            K2B_FSTAMP(7);
            if (do_row) row_finish(it, last || lb_loop);
            K2B_FSTAMP(8);
            if constexpr (MODE == MODE_SPLIT_LBFGS) {
                if (lb_loop) {       // closure result out, next point in (the optimiser runs on the slot's idle tree wave in between)
                    const int P_ = 3 + D + NB + 3;
                    if (do_row && f_valid[0] && !last) {
                        float* gdst = const_cast<float*>(a.lb_grad) + (size_t)f[0] * P_;
                        gdst[pA] = g0[0];
                        if (actB) gdst[pB] = g1[0];
                        if (lane == 0) const_cast<float*>(a.lb_loss)[f[0]] = loss_total[0];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    __syncthreads();
                    __syncthreads();
                    if (lb_done_cell[0] != 0 && lb_done_cell[1] != 0 && it < nit - 2) it = nit - 2;   // (as the tree waves: every optimiser is done)
                    if (do_row) {    // (also behind the last closure: whatever the zero-step Adam update made of a non-finite gradient, the
                                     //  result is the parked point in the parameter arrays)
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        if (chain) {
                            const size_t fr_row = (size_t)f[0] * a.chain_len + step;
                            x0[0] = *out_ptr_c(fr_row, pA);
                            x1[0] = actB ? *out_ptr_c(fr_row, pB) : 0.f;
                        } else {
                            x0[0] = *param_ptr(f[0], pA, a.go_in, a.bp_in, a.be_in, a.tr_in);
                            x1[0] = actB ? *param_ptr(f[0], pB, a.go_in, a.bp_in, a.be_in, a.tr_in) : 0.f;
                        }
                    }
                }
            }
        }
        K2B_FSTAMP_ROW_PRINT;
            if (!PAIR && chain && do_row && f_valid[0]) {     // this step's frame: row s chain_len + step
                const size_t fr = (size_t)f[0] * a.chain_len + step;
                *out_ptr_c(fr, pA) = x0[0];
                if (actB) *out_ptr_c(fr, pB) = x1[0];
                if (lane == 0 && a.loss_out) a.loss_out[fr] = loss_total[0];
            }
        }
        if (chain) return;
    } else {
        // paired: every wave carries component `wave` and the row and tree roles of its two slots
        half8 pa_h[4][2], pa_l[4][2];
        load_frags(wave, pa_h, pa_l);
        {                    // park the lo fragments in LDS (visible after the first barrier of the loop)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) plo[((wave * 4 + t) * 2 + ks) * 64 + lane] = pa_l[t][ks];
        }
        for (int it = 0; it < a.num_iters; ++it) {
            const bool last = it == a.num_iters - 1;
            if (do_row) publish();
            __syncthreads();
            floatx4 ya[5];
            if (use_gmm) comp_issue(pa_h, pa_l, ya, wave, wave);
            if (use_gmm) comp_consume(ya, wave);        // consumed before the tree, whose registers are then free
            if (do_tree) tree_pass(last);
            __syncthreads();
            if (do_row) {
                row_prior(last);
                row_finish(it, last);
            }
        }
    }
    if (!do_row) return;

    // ---- 4. results -----------------------------------------------------------------------------------
    auto out_ptr = [&](int fr, int p) -> float* {
        if (p < 3) return a.go_out + (size_t)fr * 3 + p;
        if (p < 3 + D) return a.bp_out + (size_t)fr * D + (p - 3);
        if (p < 3 + D + NB) return a.be_out + (size_t)fr * NB + (p - 3 - D);
        return a.tr_out + (size_t)fr * 3 + (p - 3 - D - NB);
    };
#pragma unroll
    for (int h = 0; h < FW; ++h) {
        if (!f_valid[h]) continue;
        const int fr = f[h];
        *out_ptr(fr, pA) = x0[h];
        if (actB) *out_ptr(fr, pB) = x1[h];
        if (lane == 0 && a.loss_out) a.loss_out[fr] = loss_total[h];
        if (a.grad_out) {
            const int P = 3 + D + NB + 3;
            a.grad_out[(size_t)fr * P + pA] = g0[h];
            if (actB) a.grad_out[(size_t)fr * P + pB] = g1[h];
        }
    }
}

// the same call restricted to frames [f0, f0 + n): every per-frame array advanced by f0 frames
static FitArgs frame_range(const FitArgs& a, int f0, int n) {
    FitArgs r = a;
    const size_t f = (size_t)f0, P = (size_t)(3 + D + a.num_betas + 3);
    r.num_frames = n;
    r.j3d = a.j3d + f * a.num_targets * 3;
    if (a.conf && a.conf_per_frame) r.conf = a.conf + f * a.num_targets;
    r.go_in = a.go_in + f * 3; r.bp_in = a.bp_in + f * D; r.be_in = a.be_in + f * a.num_betas; r.tr_in = a.tr_in + f * 3;
    if (a.preserve) r.preserve = a.preserve + f * D;
    if (a.tr_prior) r.tr_prior = a.tr_prior + f * 3;
    r.go_out = a.go_out + f * 3; r.bp_out = a.bp_out + f * D; r.be_out = a.be_out + f * a.num_betas; r.tr_out = a.tr_out + f * 3;
    if (a.loss_out) r.loss_out = a.loss_out + f;
    if (a.grad_out) r.grad_out = a.grad_out + f * P;
    return r;
}

hipError_t launch_fit_world(const FitArgs& a_in, hipStream_t stream) {
    if (a_in.num_frames <= 0) return hipSuccess;
    if (a_in.chain_len > 1) {
        // warm-start chains: num_frames sequences, one split-shape slot each (the chain is serial, so the shape that
        // gives one frame the most hardware is the one to use), up to four sequences per workgroup
        FitArgs a = a_in;
        int fpw = (a.num_frames + a.num_cus - 1) / a.num_cus;
        fpw = fpw < 1 ? 1 : (fpw > 4 ? 4 : fpw);
        if (a.lb_mode == 3 && fpw > 2) fpw = 2;                  // (the optimisers sit on the idle tree waves: two sequences per workgroup)
        a.frames_per_wg = fpw;
        const dim3 grid((a.num_frames + fpw - 1) / fpw), block(MAXW * 64);
        if (a.lb_mode == 3) {
            if (a.num_betas <= 10) hipLaunchKernelGGL((k2b_fit_world_kernel<10, MODE_SPLIT_LBFGS>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k2b_fit_world_kernel<16, MODE_SPLIT_LBFGS>), grid, block, 0, stream, a);
            return hipGetLastError();
        }
        if (a.lb_mode != 0) return hipErrorInvalidValue;
        if (a.num_betas <= 10) hipLaunchKernelGGL((k2b_fit_world_kernel<10, MODE_SPLIT>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((k2b_fit_world_kernel<16, MODE_SPLIT>), grid, block, 0, stream, a);
        return hipGetLastError();
    }
    // More frames than one full-width launch of the densest shape holds (16 per CU): such a launch runs in rounds of
    // num_cus workgroups, each as long as a full one however few workgroups it has (10 000 frames: 625 workgroups =
    // 2.4 rounds, paid as 3).  Launch the whole rounds first and the remainder on its own, in the shape that suits
    // ITS size (1 808 frames: split-paired, 0.6 of a paired round).  Frames are independent: results do not change.
    const int full = MAXS * a_in.num_cus;
    if (a_in.num_frames > full && a_in.num_frames % full != 0 && !a_in.force_shape) {
        const int head = (a_in.num_frames / full) * full;
        const hipError_t e = launch_fit_world(frame_range(a_in, 0, head), stream);
        if (e != hipSuccess) return e;
        return launch_fit_world(frame_range(a_in, head, a_in.num_frames - head), stream);
    }
    FitArgs a = a_in;
    a.chain_len = 1;
    // frame slots per workgroup: enough to cover the batch with one workgroup per CU.  Up to 4: split
    // (SIMDs would idle, so every frame gets two cooperating waves); up to 8: one wave per frame;
    // beyond: two frames per wave.
    int fpw = (a.num_frames + a.num_cus - 1) / a.num_cus;
    int mode = fpw <= 4 ? MODE_SPLIT : (fpw <= MAXW ? MODE_SPLIT_PAIRED : MODE_WIDE);
    // k2b_fit_config::debug_launch_shape forces a shape regardless of the batch size, so that the parity tests can
    // drive every shape with the small golden cases (1..3: the 8-wave shapes of rounds 1-3; 4: the 16-wave one)
    static const int forced[5] = {0, MODE_SPLIT, MODE_SPLIT_PAIRED, MODE_PAIRED, MODE_WIDE};
    auto cap_of = [](int m) { return (m == MODE_SPLIT || m == MODE_SPLIT_LBFGS) ? 4 : ((m == MODE_PAIRED || m == MODE_WIDE) ? MAXS : MAXW); };
    if (a.lb_mode != 0) {
        if (fpw > (a.lb_mode == 3 ? 2 : 4)) return hipErrorInvalidValue;     // the caller checks (k2b_api.hip: lbfgs_run)
        mode = MODE_SPLIT_LBFGS;
    } else if (a.force_shape) {
        mode = forced[a.force_shape];
        // small (test) batches fill the shape's slots; batches of a CU count or more keep one workgroup per CU where the shape can
        if (a.num_frames < a.num_cus) { fpw = cap_of(mode); if (fpw > a.num_frames) fpw = a.num_frames; }
    }
    const int cap = cap_of(mode);
    fpw = fpw < 1 ? 1 : (fpw > cap ? cap : fpw);
    a.frames_per_wg = fpw;
    const dim3 grid((a.num_frames + fpw - 1) / fpw);
#define K2B_LAUNCH(NBT_, MODE_) hipLaunchKernelGGL((k2b_fit_world_kernel<NBT_, MODE_>), grid, dim3(Shape<MODE_>::NWAVES * 64), 0, stream, a)
#define K2B_LAUNCH_NB(MODE_) do { if (a.num_betas <= 10) K2B_LAUNCH(10, MODE_); else K2B_LAUNCH(16, MODE_); } while (0)
    switch (mode) {
        case MODE_SPLIT: K2B_LAUNCH_NB(MODE_SPLIT); break;
        case MODE_SPLIT_PAIRED: K2B_LAUNCH_NB(MODE_SPLIT_PAIRED); break;
        case MODE_PAIRED: K2B_LAUNCH_NB(MODE_PAIRED); break;
        case MODE_SPLIT_LBFGS: K2B_LAUNCH_NB(MODE_SPLIT_LBFGS); break;
        default: K2B_LAUNCH_NB(MODE_WIDE); break;
    }
#undef K2B_LAUNCH_NB
#undef K2B_LAUNCH
    return hipGetLastError();
}

}  // namespace k2b
