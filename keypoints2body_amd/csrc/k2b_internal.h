// Internal declarations shared by the HIP translation units of libk2b.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k2b_device.h"

namespace k2b {

constexpr int kFitJoints = 24;      // joints of the tree the fused fit kernel is built for (SMPL)
constexpr int kPriorDim = 69;       // 3 * (kFitJoints - 1)
constexpr int kPriorMaxGauss = 8;   // mixture components resident in LDS
constexpr int kMaxBetas = 16;
constexpr int kFitMaxWaves = 8;     // frames (waves) per workgroup
constexpr int kMaxJoints = 64;
constexpr int kMaxRounds = 4;       // pointer-doubling rounds: tree depth < 2^4
constexpr int kMaxWinBits = 5;      // subtree sizes < 2^5
constexpr int kLaneTabStride = 16;  // ints per lane: joint, parent lane, anc[kMaxRounds], win[kMaxWinBits]
// LDS image of the prior: rows 0..60 as [8][17][64][4] + [8][64], rows 61..68 as [8][9][64]
constexpr int kPriorImageFloats = kPriorMaxGauss * (17 * 256 + 64) + kPriorMaxGauss * 9 * 64;

// Kernel arguments of the fused fit (passed by value).
struct FitArgs {
    // model (device)
    // tree tables are indexed by LANE: lanes follow the DFS pre-order of the kinematic tree
    const float* dt;            // [64][3]     J_template[j] - J_template[parent]  (root: J_template[0])
    const float* dd;            // [64][3][16] same for J_dirs, zero padded
    const int* lane_tab;        // [64][kLaneTabStride]
    int num_rounds, num_win_bits;
    int num_betas;
    // prior (device)
    const float* pa_image;      // LDS image, kPriorImageFloats floats (see k2b_api.hip)
    const float* row_const;     // muA[8][64], cA[8][64], muB[64], cB[64]
    const float* neg_log_nllw;  // [8]
    int num_gauss;
    // call (device unless noted)
    int num_frames, num_targets;
    int lane_target[kFitJoints];  // target index fitted by joint j, or -1
    const float* j3d;
    const float* conf;
    int conf_per_frame;
    const float *go_in, *bp_in, *be_in, *tr_in, *preserve;
    float *go_out, *bp_out, *be_out, *tr_out, *loss_out, *grad_out;
    const float2* adam_coef;    // [num_iters] {lr / (1 - b1^t), sqrt(1 - b2^t)}
    int num_iters;
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float sigma, joint_w, pose_prior_w, angle_w, shape_w, preserve_w;
    int freeze_betas;
    int angle_index[4];
    float angle_sign[4];
    int num_cus;
};

hipError_t launch_fit_world(const FitArgs& a, hipStream_t stream);

// Pose set-up for LBS: per frame the relative transforms A_j (3x4), the pose feature
// vec(R_1..R_{J-1} - I) and the posed kinematic joints (+ transl).
struct PoseArgs {
    const float* j_template;   // [J][3]
    const float* j_dirs;       // [J][3][NB]
    const int* parents;        // [J]
    int num_joints, num_betas, num_out_joints;
    int num_frames;
    const float *go, *bp, *be, *tr;  // tr may be null
    float* A;                  // [B][J][12]
    float* feat;               // [B][9(J-1)]
    float* joints_out;         // [B][num_out_joints][3] (first J rows written)
};
hipError_t launch_pose_setup(const PoseArgs& a, hipStream_t stream);

struct SkinArgs {
    int num_vertices, num_joints, num_betas, num_pose_feats;
    const float* v_template;   // [V][3]
    const float* shapedirs;    // [V][3][NB]
    const float* posedirs;     // [P][3V]
    const float* lbs_weights;  // [V][J]
    const int* vertex_ids;     // optional subset [n_out] (null: all V vertices in order)
    int num_out;               // vertices produced per frame
    int num_frames;
    const float* be;           // [B][NB]
    const float* tr;           // [B][3] or null
    const float* A;            // [B][J][12]
    const float* feat;         // [B][P]
    float* out;                // [B][out_stride][3] rows out_row0 .. out_row0+num_out-1
    int out_stride, out_row0;
};
hipError_t launch_skin(const SkinArgs& a, hipStream_t stream);
int skin_bpad(int num_frames);  // row stride of the frame-minor A / feat workspaces

// J x V contraction on the matrix cores: out[J][N] = j_regressor[J][V] . rhs[V][N].
hipError_t launch_jreg_contract(const float* j_regressor, const float* rhs, float* out, int J, int V, int N,
                                float* partial_ws, int num_splits, hipStream_t stream);

}  // namespace k2b
