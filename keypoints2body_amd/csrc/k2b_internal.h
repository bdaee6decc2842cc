// Internal declarations shared by the HIP translation units of libk2b.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "k2b_device.h"

namespace k2b {

constexpr int kFitJoints = 24;      // joints of the tree the fused fit kernel is built for (SMPL)
constexpr int kPriorDim = 69;       // 3 * (kFitJoints - 1)
constexpr int kPriorMaxGauss = 8;   // mixture components resident in LDS
constexpr int kMaxBetas = 16;       // shape coefficients of the 24-joint fused fit kernel
constexpr int kMaxShape = 32;       // shape coefficients (betas | expression) of the LBS kernels and the tree fit kernel
constexpr int kFitMaxWaves = 8;     // frames (waves) per workgroup
constexpr int kMaxJoints = 64;
constexpr int kMaxRounds = 4;       // pointer-doubling rounds: tree depth < 2^4
constexpr int kLaneTabStride = 8;   // ints per lane: joint, parent lane, anc[kMaxRounds], subtree size, depth
// LDS image of the prior: rim rows 64..68 over columns 0..63 as [m][5][64], then mu | c = P mu of rows 0..63 as [m][2][64]
constexpr int kPriorRimFragEntries = 4 * 21;   // per component: 4 fragments [k-step 2][hi | lo] of 20 live lanes + one zero entry, 16 B each
constexpr int kPriorImageFloats = 2 * 5 * 64 * 4 + kPriorMaxGauss * 2 * 64 + kPriorMaxGauss * kPriorRimFragEntries * 4;
// the 64 x 64 core of every component as MFMA A fragments (f16 hi / lo, see k2b_api.hip):
//   [m][tile 4][ks0 hi, ks1 hi, ks0 lo, ks1 lo][64 lanes][8 halfs]
constexpr int kPriorFrag32Halfs = kPriorMaxGauss * 4 * 4 * 64 * 8;

// Kernel arguments of the fused fit (passed by value).
// Arguments of the device L-BFGS (k2b_lbfgs.hip); defined here because the fused fit kernel carries a copy (FitArgs::lbv).
struct LbfgsArgs {
    int B, P, D, NB, H;                  // frames, parameters per frame (3 + D + NB + 3), pose / shape widths, history slots
    int max_iter, max_eval;
    double lr, tol_g, tol_c;             // lr, tolerance_grad, tolerance_change
    float *go, *bp, *be, *tr;            // the point to evaluate next, in the closure's own parameter arrays (in / out)
    const float *loss_in, *grad_in;      // closure result at that point: [B], [B][P]
    double* sd;                          // state: scalars, integers, vectors (lbfgs_state_bytes)
    int* si;
    float* sv;
    int finalize;                        // 1: no result consumed, every frame's accepted point -> parameter arrays
};

struct FitArgs {
    // model (device)
    // tree tables are indexed by LANE: lanes follow the DFS pre-order of the kinematic tree
    const float* dt;            // [64][3]     J_template[j] - J_template[parent]  (root: J_template[0])
    const float* dd;            // [64][3][16] same for J_dirs, zero padded
    const int* lane_tab;        // [64][kLaneTabStride]
    int num_rounds;             // pointer-doubling rounds needed by the targeted joints of this call
    int num_betas;
    // prior (device)
    const float* pa_image;      // LDS image, kPriorImageFloats floats (see k2b_api.hip)
    const void* pa_frag32;      // kPriorFrag32Halfs f16
    const float* row_const;     // [8][64] per-lane rim constants: P_BB row (5), (P mu)_B, (P_BA mu_A), mu_B
    const float* neg_log_nllw;  // [8]
    float inv_scale[kPriorMaxGauss];  // 1 / (power-of-two scale of component m's fragments)
    int num_gauss;
    int frames_per_wg;          // set by launch_fit_world
    // call (device unless noted)
    int num_frames, num_targets;
    int lane_target[kFitJoints];  // target index fitted by joint j, or -1
    const float* j3d;
    const float* conf;
    int conf_per_frame;
    const float *go_in, *bp_in, *be_in, *tr_in, *preserve, *tr_prior;
    float *go_out, *bp_out, *be_out, *tr_out, *loss_out, *grad_out;
    const float2* adam_coef;    // [num_iters] {lr / (1 - b1^t), sqrt(1 - b2^t)}
    int num_iters;
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float sigma, joint_w, pose_prior_w, angle_w, shape_w, preserve_w;
    int freeze_betas;
    int opt_mask;               // bit 0 global_orient, 1 body_pose, 2 betas, 3 transl
    float transl_prior_w;
    int angle_index[4];
    float angle_sign[4];
    int num_cus;
    int chain_len, chain_iters; // warm-start chain: num_frames SEQUENCES of chain_len frames each (1: independent frames)
    int force_shape;            // 0 = chosen by the batch size; 1..4 = split / split-paired / paired / wide (k2b_fit_config::debug_launch_shape)
    // L-BFGS step as a prologue of the closure's own launch (k2b_lbfgs.hip): 0 = none; 1 = every frame's row wave first consumes the
    // PREVIOUS launch's loss / gradient (lb->loss_in / grad_in) and writes the next point into the parameter arrays, then the launch
    // evaluates it; 2 = the finalise step (accepted points into the parameter arrays), then the evaluation.  Split shape only.
    int lb_mode;
    LbfgsArgs lbv;              // the optimiser's arguments (by value: no device copy to keep in step)
    // 3 = persistent: the whole fit in ONE launch (num_iters = rounds + 1 closures; at most two frames per workgroup); the row wave
    // writes each closure's result to lb_loss / lb_grad (= lb->loss_in / grad_in), lb_history = lb->H (host copies: no device read)
    const float *lb_loss, *lb_grad;
    int lb_history;
    int lb_chain_max_iter;      // lb_mode 3 with chain_len > 1 (the default sequence mode in ONE launch): max_iter of the follow-up frames (lbv: the first's)
};

hipError_t launch_fit_world(const FitArgs& a, hipStream_t stream);

// Kernel arguments of the fused fit for large trees (k2b_fit_tree.hip: 25..64 joints, SMPL-H / SMPL-X).
typedef _Float16 k2b_half;
struct FitTreeArgs {
    // model (device), indexed by LANE (DFS pre-order of the kinematic tree)
    const float* dt;            // [64][3]      rest offset from the parent at shape 0 (root: its rest joint)
    const float* dd;            // [64][3][32]  the same for the shape directions, zero padded
    const int* tab;             // [64][8]: joint, parent lane, subtree size, depth, prior source lane, prior source
                                //          component, first prior dimension of this joint (-1: none), unused
    const int* anc;             // [64][4]: lane of the ancestor 1, 2, 4, 8 levels up, or 63 (a non-joint lane: identity)
    int num_joints, num_shape, num_rounds;   // pointer-doubling rounds the targeted joints need: 2^rounds > their depth
    // prior (device): the mixture folded to its first prior_dims <= 64 dimensions (see k2b_fit_tree.hip)
    const k2b_half* pfrag;      // [8][4 tiles][ks0 hi, ks1 hi, ks0 lo, ks1 lo][64 lanes][8]: the 64 x 64 core of every (scaled) precision
                                // matrix as v_mfma_f32_16x16x32_f16 A fragments - the image k2b_fit.hip uses (k2b_prior::frag32)
    float inv_scale[8];         // per component: 1 / (power-of-two scale of its fragments)
    const float *ph, *pb, *pmu; // [M][64]          h = b - A mu, b, mu
    const float* pcl;           // [M]              0.5 c_m - log(nll weight)
    int num_gauss, prior_dims;
    // call
    int num_frames, num_targets;
    int lane_target[64];        // target index fitted by the joint of lane l, or -1
    const float* j3d;
    const float* conf;
    int conf_per_frame;
    const float *go_in, *bp_in, *be_in, *tr_in, *preserve;
    float *go_out, *bp_out, *be_out, *tr_out, *loss_out, *grad_out;
    const float2* adam_coef;    // [num_iters] {lr / (1 - b1^t), sqrt(1 - b2^t)}
    int num_iters;
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float sigma, joint_w, pose_prior_w, angle_w, shape_w, preserve_w;
    int freeze_betas, num_betas_prior;   // the shape prior and freeze_betas apply to the first num_betas_prior coefficients
    int opt_mask;               // bit 0 global_orient, 1 body_pose, 2 shape coefficients, 3 transl
    int angle_index[4];
    float angle_sign[4];
    int chain_len, chain_iters;  // warm-start chain: num_frames SEQUENCES of chain_len frames each (<= 1: independent frames)
    int comp_waves;              // set by launch_fit_tree: 4 = the mixture runs on four dedicated component waves (<= 4 frames per CU)
    int debug_shape;             // 0 = chosen by the batch size; 1 = force the plain shape, 2 = force the component-wave shape (tests)
};
hipError_t launch_fit_tree(const FitTreeArgs& a, hipStream_t stream);

// LBS operands are f16 hi/lo pairs in MFMA fragment order: [k-step][row][16 halfs].
constexpr float kPdScale = 256.0f;   // power-of-two scale of the vertex-GEMM B operand (keeps f16 lo terms normal)

// Element (k, row) inside piece number `frag` of a piece-ordered operand: a piece is the
// 1 KiB a wave moves for one 16-deep k-step of one 32-row tile, stored lane-linear:
// [h = (k >> 3) & 1][row & 31][k & 7]  (lane 32 h + row owns 16 contiguous bytes).
__host__ __device__ inline size_t frag_elem(size_t frag, int k, int row) {
    return ((frag * 2 + ((k >> 3) & 1)) * 32 + (row & 31)) * 8 + (k & 7);
}

// Pose set-up for LBS: per frame the relative transforms A_j (3x4) and the feature vector
// X = [vec(R_1..R_{J-1} - I) | beta | 1 | 1] as f16 hi/lo MFMA operands, plus the posed
// kinematic joints (+ transl).
struct PoseArgs {
    const float* j_basis_lane; // [3][1 + NB][64]: J_template (k' = 0) and J_dirs (k' = 1 + k) with the joint as the fastest index, zeros for lanes >= J
    const int* parents;        // [J]
    int num_joints, num_betas, num_out_joints;
    int num_frames, frames_padded;
    int k_steps_x;             // 16-deep k-steps of the vertex GEMM (features)
    const float *go, *bp, *be, *tr;  // tr may be null
    k2b_half *xh, *xl;         // pieces [k_steps_x][frames_padded / 32]
    k2b_half* a2;              // group layout of the tile kernel (see TileArgs)
    int a2_stream_order;       // 1: k-groups of an entry in the stream kernel's order hi.. | PAD | lo.. | ZERO (see StreamArgs)
    float* joints_out;         // [B][num_out_joints][3] (first J rows written) or null
    int xcd_frames;            // set by the launcher: 0 = workgroup b takes frame b; else XCD label b % 8 takes the frames [x, x + 1) * xcd_frames
};
hipError_t launch_pose_setup(const PoseArgs& a, hipStream_t stream);
int lbs_frames_padded(int num_frames);

// ---- tile kernel (k2b_lbs_tile_kernel): 128 frames x 128 vertices per workgroup, persistent ------------------
// Operands of the transform GEMM T = A . W^T in k-GROUPS of 8 joints over 16-row tiles (256 B = [16 rows][8 halfs]):
//   A  [16-frame tile][entry 12][NGP][16][8]   groups: hi_0..hi_{GA-1}, lo_0..lo_{GA-1}, PAD (translation terms), ZERO
//   W  [16-vertex tile][NGP][16][8]            groups: hi_0..hi_{GA-1}, lo_0..lo_{GA-1}, ONES ([1,1,1,0,...] per row), ZERO
//      (the ZERO group only ever meets zeros of A, so its first half per row is free to carry a tag: 1 + output joint of the vertex)
// GA = ceil(J / 8), NGP = 2 GA + 2 (a multiple of 4: whole 1 KiB pieces).  The three f16-split products (hi.hi + hi.lo +
// lo.hi) are ONE contraction over the concatenated group sequences  A: hi | hi | lo | PAD,  W: hi | lo | hi | ONES
// (3 GA + 1 groups, padded to a multiple of four with ZERO): ceil((3 GA + 1) / 4) MFMAs of depth 32 per output tile, no
// padding of J to 16, and the PAD x ONES position adds the frame's translation (three f16 terms) to the entries 3, 7, 11
// inside the GEMM.
constexpr int tile_groups_a(int J) { return (J + 7) / 8; }
constexpr int tile_ngp(int GA) { return 2 * GA + 2; }
struct TileArgs {
    const k2b_half *xh, *xl;     // X fragments [k_steps_x][f_tiles]          (pose set-up)
    const k2b_half *a2;          // A groups    [2 f_tiles][12][NGP][16][8]   (pose set-up)
    const k2b_half *pdh, *pdl;   // Pd fragments [k_steps_x][3][v_tiles]      (model)
    const k2b_half *w2;          // W groups    [2 v_tiles][NGP][16][8]       (model)
    int groups_a;                // GA
    int k_steps_x, f_tiles, v_tiles;
    int num_frames, num_out;
    float* out;                  // [B][out_stride][3], rows out_row0 .. out_row0 + num_out - 1
    int out_stride, out_row0;
    float* dump;                 // >= 64 x 3 floats: where lanes without a valid (frame, vertex) put their store
    // vertex-selected output joints (smplx's extra joints) written by the same launch: a vertex whose W row carries "1 + e" in
    // its padding group is stored a second time, to joints_out[frame][joints_row0 + e]; null: no joint copies
    float* joints_out;
    int joints_stride, joints_row0;   // J + E, J
    int num_wgs;                 // grid size (a multiple of 8), set by the launcher
};
hipError_t launch_skin_tiles(const TileArgs& a, int num_cus, hipStream_t stream);

// ---- stream kernel (k2b_lbs_stream.hip): the same tile for 17-24 joints and 7 pose k-steps, Pd global -> registers ------------
// Operands (1 KiB pieces = 64 lanes x 8 halfs in v_mfma_f32_16x16x32_f16 operand order: lane = row + 16 k-group):
//   X   as for the tile kernel
//   A   [16-frame tile][entry 12][fragment 2]: fragment 0 = k-groups hi_0 hi_1 hi_2 PAD, fragment 1 = lo_0 lo_1 lo_2 ZERO
//       (the pose set-up writes this group order with PoseArgs::a2_stream_order)
//   Pd  [k-step 7][16-vertex tile][coordinate 3][hi | lo]
//   W   [16-vertex tile][3]: [hi | ONES], [hi | tag], [lo | 0]   (tag: 1 + output joint of the vertex, first half of the group)
constexpr int kStreamKSteps = 7;
struct StreamArgs {
    const k2b_half *xh, *xl, *a2, *pd, *w;
    int f32_tiles;               // 32-frame tiles (frames padded / 32)
    int nv16;                    // 16-vertex tiles of the padded vertex set, a multiple of 8
    int num_frames, num_out;
    float* out;
    int out_stride, out_row0;
    float* dump;
    float* joints_out;           // see TileArgs
    int joints_stride, joints_row0;
    int num_wgs;
};
hipError_t launch_skin_stream(const StreamArgs& a, int num_cus, hipStream_t stream);
// 49-56 joints and 16 pose k-steps (SMPL-X), k2b_lbs_stream_x_kernel: the same operands with
//   A   [16-frame tile][entry 12][fragment 4]: hi_0-3 | hi_4-6 PAD | lo_0-3 | lo_4-6 ZERO   (a2_stream_order with GA = 7)
//   Pd  [k-step 16][16-vertex tile][coordinate 3][hi | lo]
//   W   [16-vertex tile][5]: hi_0-3, [hi_4-6 | ONES], lo_0-3, [lo_4-6 | 0], [hi_4-6 | tag]
constexpr int kStreamXKSteps = 16;
hipError_t launch_skin_stream_x(const StreamArgs& a, int num_cus, hipStream_t stream);
hipError_t launch_gather_joints(const float* verts, const int* ids, float* joints, int num_frames, int V, int J, int E,
                                hipStream_t stream);

// J x V contraction on the matrix cores: out[J][N] = j_regressor[J][V] . rhs[V][N].
hipError_t launch_jreg_contract(const float* j_regressor, const float* rhs, float* out, int J, int V, int N,
                                float* partial_ws, int num_splits, hipStream_t stream);

// Joint-loss term of vertex-selected joints and its gradient (slow path, k2b_vertex.hip).
struct VertexTermArgs {
    // model (device): smplx tensors as uploaded by k2b_model_create
    const float *v_template, *shapedirs, *posedirs, *lbs_weights, *j_template, *j_dirs;
    const int *parents, *extra_ids;
    int num_vertices, num_betas, num_joints;
    // call
    int num_frames, num_sel;
    int sel[32];                // index into extra_ids of every fitted vertex joint
    int sel_k[32];              // its column in targets / conf
    int num_targets;            // columns of targets / conf
    const float* targets;       // dev [B][num_targets][3]
    const float* conf;          // dev [num_targets] or [B][num_targets] (conf_per_frame) or null
    int conf_per_frame;
    float sigma, joint_w;
    const float *go, *bp, *be, *tr;
    float *loss_out, *grad_out; // dev [B], [B][3 + 69 + NB + 3] (grad_out may be null with the Adam tail)
    // Adam tail (grad_in != null): the launch adds its loss / gradient to those of the evaluate-only launch of the fused
    // kernel (loss_in, grad_in), applies the optimiser's membership mask and makes the frame's Adam step in place
    const float *loss_in, *grad_in;
    float *go_w, *bp_w, *be_w, *tr_w;   // the parameters again, writable (same buffers as go, bp, be, tr)
    float *adam_m, *adam_v;             // dev [B][P]
    const float2* adam_coef;            // this step's {lr / (1 - b1^t), sqrt(1 - b2^t)}
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    int opt_mask;
    int frozen_shape;                   // the first `frozen_shape` shape coefficients take no step (frozen betas beside a free expression)
};
hipError_t launch_vertex_term(const VertexTermArgs& a, hipStream_t stream);
hipError_t launch_adam(float* x, const float* g, float* m, float* v, long long n, float lr_over_bc1, float sqrt_bc2,
                       float one_minus_beta1, float beta2, float one_minus_beta2, float eps, hipStream_t stream);

// Raises a kernel's dynamic-LDS limit once PER DEVICE (the attribute is per device; a process-wide flag left the second
// GPU of a process without it, and two threads raced on it): one atomic flag per device ordinal and kernel instantiation.
template <class Kernel>
inline hipError_t ensure_dynamic_lds(Kernel kernel, std::atomic<unsigned long long>& done_mask, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);   // idempotent: a race repeats it
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}

// ---- device-resident L-BFGS (k2b_lbfgs.hip): one state machine per frame, one closure result consumed per step launch ----------
constexpr int kLbfgsMaxHistory = 100;    // torch.optim.LBFGS's default history_size
size_t lbfgs_state_bytes(int B, int P, int H, size_t* off_si, size_t* off_sv);
hipError_t launch_lbfgs_step(const LbfgsArgs& a, hipStream_t stream);
hipError_t launch_lbfgs_frame_prep(float* go, const float* sgo, float* bp, const float* sbp, float* be, const float* sbe, float* tr,
                                   const float* str, float* pres, int D, int NB, void* state, size_t state_bytes, hipStream_t stream);

// Geodesic angle (degrees) between n pairs of axis-angle rotations (evaluation metric, k2b_metrics.hip).
hipError_t launch_angular_error(const float* pred, const float* gt, float* out, long long n, hipStream_t stream);

}  // namespace k2b
