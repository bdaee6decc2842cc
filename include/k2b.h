/*
 * k2b.h — C ABI of the MI355X-native SMPLify-style fitting engine (libk2b.so).
 *
 * The reference (saifkhichi96/keypoints2body) is pure Python and has no FFI; these
 * entry points are what a binding for its hot path would call, one per Python-level
 * seam that SURVEY.md §8(b) names.  Each declaration cites the reference interface it
 * replaces (paths relative to /root/reference/keypoints2body).  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns K2B_OK (0) or a negative k2b_status; k2b_last_error()
 *     returns a thread-local, human-readable message for the last failure;
 *   - "dev" pointers are device (HBM) addresses, e.g. torch `tensor.data_ptr()` of a
 *     contiguous float32 tensor on the HIP device; "host" pointers are host memory;
 *   - all matrices are row-major (C order), float32 unless stated;
 *   - launches are stream-ordered on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream).  Calls that may block the host or synchronise the device:
 *     *_create / *_destroy always; k2b_lbs when its per-model workspace has to GROW (first
 *     call, or more frames than any earlier call: hipDeviceSynchronize + hipFree/hipMalloc;
 *     k2b_model_reserve() pre-sizes it so that later calls never do); k2b_fit_world on the
 *     FIRST use of an (iterations, lr, beta1, beta2) combination per model (blocking upload
 *     of the Adam bias table; at most 64 tables are kept, least recently used evicted after
 *     a stream sync).  k2b_vertex_term never synchronises: its joint selection travels by value in
 *     the kernel arguments;
 *   - buffers are caller-owned; inputs are never written; handles may be shared by
 *     threads as long as concurrent calls use different streams AND different
 *     workspaces (one workspace per handle: serialise calls on one handle).
 */
#ifndef K2B_H
#define K2B_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum k2b_status {
    K2B_OK = 0,
    K2B_ERR_INVALID_ARGUMENT = -1, /* -> Python ValueError (world_space.py:118-119, adapters.py:104-117) */
    K2B_ERR_UNSUPPORTED = -2,      /* -> Python NotImplementedError (api/frame.py:71-75) */
    K2B_ERR_HIP = -3,              /* -> Python RuntimeError: a HIP runtime call failed */
    K2B_ERR_NO_DEVICE = -4         /* -> Python RuntimeError: no gfx950 device / kernels not loadable */
} k2b_status;

typedef struct k2b_model k2b_model; /* body-model constants resident in HBM */
typedef struct k2b_prior k2b_prior; /* max-mixture pose prior resident in HBM */

/* Library / ABI version: (major << 16) | minor. */
uint32_t k2b_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char *k2b_last_error(void);

/* ---------------------------------------------------------------------------------
 * Body model.  Replaces `load_body_model` / `smplx.create(...).to(device)`
 * (api/model_factory.py:19-40) for the constants the path reads, and precomputes
 * J_template = J_regressor . v_template and J_dirs = J_regressor . shapedirs (the
 * dense J x V contraction, an fp32 MFMA kernel) so that the per-iteration kernel never
 * touches vertices (SURVEY.md §8a note N1).
 *
 * All pointers are HOST pointers; the arrays are copied.
 *   v_template  [V][3]        shapedirs [V][3][NB]      posedirs [9*(J-1)][3*V]
 *   j_regressor [J][V]        lbs_weights [V][J]        parents [J] (parents[0] = -1,
 *   extra_vertex_ids [E]      (output joints J..J+E-1 are these vertices)   parents[i] < i)
 * Limits: 2 <= J <= 64, 1 <= NB <= 32, E >= 0.  (24 joints with NB <= 16 run the 24-lane fused fit kernel,
 * everything else the tree kernel; the vertex kernel is built for 17-24 and 49-56 joints.)
 * ------------------------------------------------------------------------------- */
int k2b_model_create(k2b_model **out, int32_t num_vertices, int32_t num_joints, int32_t num_betas,
                     int32_t num_extra_joints, const float *v_template, const float *shapedirs,
                     const float *posedirs, const float *j_regressor, const float *lbs_weights,
                     const int32_t *parents, const int32_t *extra_vertex_ids);
void k2b_model_destroy(k2b_model *model);
/* Sizes: V, J, NB, E. */
int k2b_model_dims(const k2b_model *model, int32_t *num_vertices, int32_t *num_joints,
                   int32_t *num_betas, int32_t *num_extra_joints);
/* Pre-sizes the per-model LBS workspace for batches of up to `max_frames` frames, so that no later k2b_lbs
 * call on this model allocates or synchronises (see Conventions).  Synchronises the device if it has to grow. */
int k2b_model_reserve(k2b_model *model, int32_t max_frames);
/* Copies the precomputed J_template [J][3] and J_dirs [J][3][NB] to HOST buffers (either may be NULL). */
int k2b_model_joint_basis(const k2b_model *model, float *j_template, float *j_dirs);

/* ---------------------------------------------------------------------------------
 * Pose prior.  Replaces the buffers `MaxMixturePrior.__init__` registers
 * (core/prior.py:133-163): `means [M][D]`, `precisions [M][D][D]`, `nll_weights [M]`
 * (HOST pointers, float32, exactly those buffers).  D must equal 3*(J-1) of the model
 * it is used with.  The library symmetrises each precision (0.5 (P + P^T), the exact
 * gradient of the quadratic form) and stores -log(nll_weights).
 * ------------------------------------------------------------------------------- */
int k2b_prior_create(k2b_prior **out, int32_t num_gaussians, int32_t dim, const float *means,
                     const float *precisions, const float *nll_weights);
void k2b_prior_destroy(k2b_prior *prior);

/* ---------------------------------------------------------------------------------
 * Fit configuration.  Field-for-field the knobs of
 *   WorldSpaceFitter.__init__/fit_frame   (core/fitters/world_space.py:56-66, 93-103, 214)
 *   body_fitting_loss_3d                  (core/losses.py:24-38)
 *   torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8)  (world_space.py:249)
 * k2b_fit_config_default() fills the values the reference's world fitter uses.
 * ------------------------------------------------------------------------------- */
typedef struct k2b_fit_config {
    int32_t num_iters;          /* num_iters_first if seq_ind == 0 else num_iters_followup */
    double step_size;           /* Adam lr (config.py:30), 1e-2 (0 = evaluate only).  Adam numbers are doubles */
    double adam_beta1;          /* 0.9     because torch keeps them as Python floats and forms     */
    double adam_beta2;          /* 0.999   1-beta, beta**t and lr/(1-beta1**t) in double before    */
    double adam_eps;            /* 1e-8    rounding to float32 (torch/optim/adam.py)               */
    float sigma;                /* GMoF sigma (losses.py:33), 100 */
    float joint_loss_weight;    /* 600 (config.py:34) */
    float pose_prior_weight;    /* 4.78*1.5 (losses.py:34) */
    float angle_prior_weight;   /* 15.2 (losses.py:36) */
    float shape_prior_weight;   /* 5.0 (losses.py:35) */
    float pose_preserve_weight; /* 5.0 if seq_ind > 0 else 0 (world_space.py:211) */
    int32_t freeze_betas;       /* betas excluded from the optimiser (world_space.py:171,228-229) */
    int32_t conf_per_frame;     /* 0: conf is [K] shared by all frames (the reference's behaviour,
                                   world_space.py:161-164); 1: conf is [B][K] */
    int32_t angle_prior_index[4]; /* body-pose indices of the bending prior (losses.py:16): 52,55,9,12 */
    float angle_prior_sign[4];    /* (losses.py:17): +1,-1,-1,-1 */
    /* Parameter groups handed to the optimiser: bit 0 global_orient, 1 body_pose, 2 betas, 3 transl.
     * 15 = world fitter (world_space.py:215-229); 9 = stage 1 of the camera-space fitter
     * (camera_space.py:137-141: [global_orient, camera_translation]).  freeze_betas clears bit 2. */
    int32_t optimize_mask;
    /* w_t^2 * |transl - transl_prior_target|^2, the depth term of camera_fitting_loss_3d
     * (losses.py:70-93: depth_loss_weight 100; its broadcast over the 4 torso joints makes the
     * effective weight 200 in that path).  0 = off. */
    float transl_prior_weight;
    /* Debug / test knob: 0 = the launcher picks the launch shape from the batch size (the product setting);
     * 1, 2, 3, 4 force the split / split-paired / paired / wide (16-wave) shape of the fused kernel (and a single launch), so
     * that the parity tests can drive every shape with small cases.  Results do not depend on it.  The tree kernel of the larger
     * models has two shapes: 1 = plain (every wave a frame + its share of the mixture), 2 = four extra component waves. */
    int32_t debug_launch_shape;
    /* Larger models (SMPL-H / SMPL-X): `body_pose` holds ALL non-root joints (SMPL-X: body 63 | jaw, eyes 9 | hands
     * 90), `betas` all shape coefficients (betas | expression).  The mixture, the bending prior and the preserve term
     * see the first prior_pose_dims values of body_pose (0 = all of them; SMPL-X: 63, the mixture's remaining
     * dimensions fixed at 0); the shape prior and freeze_betas apply to the first num_betas_prior coefficients
     * (0 = all; SMPL-X: 10, the expression stays free as in world_space.py:137-151,215-229). */
    int32_t prior_pose_dims;
    int32_t num_betas_prior;
} k2b_fit_config;

void k2b_fit_config_default(k2b_fit_config *cfg);
/* sizeof(k2b_fit_config) as compiled into the library: lets a binding verify its struct mirror. */
uint32_t k2b_fit_config_size(void);

/* ---------------------------------------------------------------------------------
 * k2b_fit_world — the hot path.  Replaces the Adam branch of
 * `WorldSpaceFitter.fit_frame` (core/fitters/world_space.py:93-256) for a batch of B
 * independent frames: per iteration the SMPL joint forward (Rodrigues, kinematic chain,
 * J(beta)), `body_fitting_loss_3d` (core/losses.py:24-67) with `MaxMixturePrior`
 * (core/prior.py:182-195), the analytic backward and the Adam update, all iterations
 * fused in one launch (one or two wavefronts per frame, or two frames per wavefront, by batch size).
 *
 *   model_joint_index [K] HOST int32: model joint fitted to target k (the reference's
 *       smpl_index / target_model_indices, world_space.py:194-201); values must be
 *       distinct.  Indices >= J name smplx's vertex-selected "extra" joints (at
 *       most 32 of them, at least one kinematic joint beside them; any supported tree): the call then queues TWO
 *       launches per iteration on the stream - the fused kernel in evaluate-only mode and the
 *       vertex-term kernel with its Adam tail - with no host work in between; conf may be per
 *       frame there too.
 *   j3d  dev [B][K][3]   target joints (already gathered with corr_index)
 *   conf dev [K] or [B][K] (see conf_per_frame); NULL = ones
 *   *_in dev: initial global_orient [B][3], body_pose [B][3(J-1)], betas [B][NB], transl [B][3]
 *   preserve_pose dev [B][3(J-1)] or NULL (= body_pose_in, as world_space.py:159)
 *   transl_prior_target dev [B][3] or NULL (= transl_in): centre of the transl prior
 *   *_out dev: fitted parameters, same shapes (may alias the inputs)
 *   loss_out dev [B]: per-frame loss of the LAST iteration evaluated BEFORE its step
 *       (world_space.py:256); the reference's scalar is the sum over the batch
 *   grad_out dev [B][P] or NULL: d loss / d params at the last iteration, P = 3J + NB + 3,
 *       order [global_orient | body_pose | betas | transl] (debug / tests)
 * ------------------------------------------------------------------------------- */
int k2b_fit_world(const k2b_model *model, const k2b_prior *prior, const k2b_fit_config *cfg,
                  int32_t num_frames, int32_t num_targets, const int32_t *model_joint_index,
                  const float *j3d, const float *conf,
                  const float *global_orient_in, const float *body_pose_in, const float *betas_in,
                  const float *transl_in, const float *preserve_pose, const float *transl_prior_target,
                  float *global_orient_out, float *body_pose_out, float *betas_out, float *transl_out,
                  float *loss_out, float *grad_out, void *stream);

/* ---------------------------------------------------------------------------------
 * k2b_fit_sequence — the reference's sequence mode as ONE launch.  Replaces the frame loop of
 * `optimize_params_sequence` with `use_previous_frame_init=True` (api/sequence.py:214-281): frame 0 of
 * a sequence is fitted from the given start with cfg->num_iters iterations and no preserve term
 * (seq_ind == 0: world_space.py:159,211); every later frame starts from the previous frame's
 * RESULT, preserves that result's body pose with cfg->pose_preserve_weight (world_space.py:159) and
 * runs `followup_iters` iterations (world_space.py:214) with a fresh Adam state.  The chain of one
 * sequence is serial (one cooperating wavefront pair walks it); num_sequences chains run side by side.
 *
 *   j3d  dev [S][T][K][3], conf dev [K] or [S][T][K] (conf_per_frame)
 *   *_in dev [S][...]: start of frame 0 of every sequence
 *   *_out dev [S][T][...], loss_out dev [S][T]: every frame's fitted parameters / last-iteration loss
 * Kinematic targets only (vertex-selected joints: K2B_ERR_UNSUPPORTED - fit frame by frame); cfg->transl_prior_weight
 * must be 0.  24-joint models run the split shape of the fused kernel, larger trees (SMPL-X) the tree kernel.
 * ------------------------------------------------------------------------------- */
int k2b_fit_sequence(const k2b_model *model, const k2b_prior *prior, const k2b_fit_config *cfg,
                     int32_t num_sequences, int32_t frames_per_sequence, int32_t followup_iters,
                     int32_t num_targets, const int32_t *model_joint_index,
                     const float *j3d, const float *conf,
                     const float *global_orient_in, const float *body_pose_in, const float *betas_in,
                     const float *transl_in,
                     float *global_orient_out, float *body_pose_out, float *betas_out, float *transl_out,
                     float *loss_out, void *stream);

/* ---------------------------------------------------------------------------------
 * k2b_fit_world_lbfgs — the L-BFGS branch of the fitters, which is the reference's DEFAULT (core/config.py:29
 * `use_lbfgs=True`; world_space.py:231-247, camera_space.py:144-182,229-267): per frame
 *     torch.optim.LBFGS(params, max_iter=num_iters, lr=step_size, line_search_fn="strong_wolfe").step(closure)
 * followed by one more evaluation of the loss at the result (world_space.py:245-246).  The closure (loss and gradient at
 * the current parameters) is k2b_fit_world's evaluate-only launch under `cfg` (its num_iters / step_size are ignored; weights,
 * sigma, optimize_mask, freeze_betas, prior_pose_dims ... apply: parameters outside the optimiser get a zero gradient and
 * never move); the optimiser itself - two-loop recursion over up to `history_size` pairs (<= 0: torch's 100), strong-Wolfe
 * bracket / zoom with torch's cubic interpolation, tolerance_grad / tolerance_change / max_iter / max_eval = max_iter * 5 / 4
 * exits - is a state machine on the device, one independent instance per frame (k2b_lbfgs.hip).  The call only queues
 * launches on `stream`: no host synchronisation, nothing is read back.  max_eval + 2 rounds of [closure, optimiser step] + the
 * final evaluation, as ONE launch for at most two frames per CU (the rounds are iterations of the fused kernel's loop, the
 * optimiser runs on an idle wave of the workgroup), one launch per round up to four frames per CU (the step as a prologue
 * of the closure's launch), two launches per round beyond that and for the larger models; the result does not depend on which.  Arguments as k2b_fit_world; preserve_pose / transl_prior_target NULL = the initial body pose /
 * translation; *_out may alias *_in; loss_out dev [B] and grad_out dev [B][3 + 3(J-1) + NB + 3] (either may be NULL) receive
 * loss and gradient AT the result.  Frames never interact (torch couples the frames of a batch in one line search; the
 * reference only ever passes one frame).  At most 192 parameters per frame.
 * Line searches branch on rounding, so results agree with torch's statistically (and iterate by iterate with the float64
 * twin core/lbfgs_batched.py while rounding has not yet been amplified): see DESIGN.md.
 * ------------------------------------------------------------------------------- */
int k2b_fit_world_lbfgs(const k2b_model *model, const k2b_prior *prior, const k2b_fit_config *cfg,
                        int32_t num_frames, int32_t num_targets, const int32_t *model_joint_index,
                        const float *j3d, const float *conf,
                        const float *global_orient_in, const float *body_pose_in, const float *betas_in,
                        const float *transl_in, const float *preserve_pose, const float *transl_prior_target,
                        float *global_orient_out, float *body_pose_out, float *betas_out, float *transl_out,
                        float *loss_out, float *grad_out,
                        int32_t max_iter, int32_t history_size, double lr, double tolerance_grad,
                        double tolerance_change, void *stream);

/* ---------------------------------------------------------------------------------
 * k2b_fit_sequence_lbfgs — the reference's DEFAULT sequence mode in one call: the frame loop of
 * `optimize_params_sequence` with `use_previous_frame_init=True` (api/sequence.py:214-281, config.py:57) over the L-BFGS
 * branch (`use_lbfgs=True`, config.py:29; world_space.py:231-247).  ONE sequence of num_frames frames: frame 0 is fitted from
 * the given start (*_in: one row) with first_iters iterations and no preserve term (world_space.py:159,211); every later frame
 * starts from its predecessor's RESULT, preserves that result's body pose with cfg->pose_preserve_weight and runs
 * followup_iters iterations (world_space.py:214).  Every frame is one k2b_fit_world_lbfgs fit; the call only queues launches
 * (no host work between the frames) - ONE launch for the whole sequence where the fused kernel takes it (24-joint model,
 * kinematic targets: the frame loop runs inside the persistent launch), a launch or a few per frame otherwise; same bits.
 *   j3d dev [T][K][3]; conf dev [K], or [T][K] with cfg->conf_per_frame (each frame reads its own row, as the sequence API
 *   passes `conf[idx]`); *_out dev [T][...], loss_out dev [T] (may be NULL): every frame's result / loss at the result.
 * cfg->transl_prior_weight must be 0.
 * ------------------------------------------------------------------------------- */
int k2b_fit_sequence_lbfgs(const k2b_model *model, const k2b_prior *prior, const k2b_fit_config *cfg,
                           int32_t num_frames, int32_t num_targets, const int32_t *model_joint_index,
                           const float *j3d, const float *conf,
                           const float *global_orient_in, const float *body_pose_in, const float *betas_in,
                           const float *transl_in,
                           float *global_orient_out, float *body_pose_out, float *betas_out, float *transl_out,
                           float *loss_out, int32_t first_iters, int32_t followup_iters, int32_t history_size,
                           double lr, double tolerance_grad, double tolerance_change, void *stream);

/* ---------------------------------------------------------------------------------
 * k2b_lbs — full SMPL forward.  Replaces `self.smpl(**kwargs)` (smplx `SMPL.forward`,
 * call sites world_space.py:34,192,278; engine.py:114) for a batch:
 *   joints_out dev [B][J+E][3], vertices_out dev [B][V][3] (NULL: joints only; the E
 *   vertex-selected joints are then skinned alone).  transl may be NULL (no translation).
 * ------------------------------------------------------------------------------- */
int k2b_lbs(const k2b_model *model, int32_t num_frames, const float *global_orient,
            const float *body_pose, const float *betas, const float *transl,
            float *joints_out, float *vertices_out, void *stream);

/* Development: copies the first nbytes (<= 64 KiB) of the model's scratch row (where diagnostic builds leave their
 * in-kernel time stamps) to HOST memory; synchronises the device. */
int k2b_debug_read_dump(const k2b_model *model, void *host, int64_t nbytes);

/* ---------------------------------------------------------------------------------
 * Vertex-selected joints in the loss, stand-alone pieces.  `target_model_indices` of the reference
 * (world_space.py:198-201) may name smplx's "extra" joints, which are single mesh vertices
 * (index J + e, e < E).  k2b_fit_world handles them itself (see model_joint_index above); these
 * two entry points expose the pieces it is made of, for optimisers that run on the host (the
 * L-BFGS branch may call k2b_fit_world in evaluate-only mode instead) and for the parity tests.
 *
 * k2b_vertex_term: loss [B] and gradient [B][3 + 3(J-1) + NB + 3] (layout of grad_out above)
 *   of  joint_loss_weight^2 conf_e^2 sum_xyz gmof(vertex_e + transl - target_e)  over the E_sel
 *   selected extra joints; extra_index HOST int32 [E_sel] in [0, E); targets dev [B][E_sel][3];
 *   conf dev [E_sel] or NULL.  At most 32 selected joints per call.
 * k2b_adam_step: torch.optim.Adam single-tensor update of n floats for step t = 1, 2, ...
 *   (bias corrections formed in double like the fused kernel's table); m, v are the caller's
 *   state buffers (zero before the first step).
 * ------------------------------------------------------------------------------- */
int k2b_vertex_term(const k2b_model *model, int32_t num_frames, int32_t num_selected,
                    const int32_t *extra_index, const float *targets, const float *conf,
                    float sigma, float joint_loss_weight, const float *global_orient,
                    const float *body_pose, const float *betas, const float *transl,
                    float *loss_out, float *grad_out, void *stream);
int k2b_adam_step(int64_t n, float *params, const float *grad, float *m, float *v, int32_t step,
                  double step_size, double beta1, double beta2, double eps, void *stream);

/* ---------------------------------------------------------------------------------
 * k2b_angular_error_deg — the evaluation metric behind MPJAE.  Replaces
 * `compute_angular_error_deg` (reference cli/eval.py:131-140, with `rotvec_to_rotmat`
 * :88-128) for n pairs of axis-angle rotations:
 *   pred, gt dev [n][3]; err_deg_out dev [n] = geodesic angle in degrees, clipped like
 *   the reference (cos in [-1 + 1e-6, 1 - 1e-6]).  n == 0 is a no-op.
 * ------------------------------------------------------------------------------- */
int k2b_angular_error_deg(int64_t n, const float *pred_rotvec, const float *gt_rotvec,
                          float *err_deg_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* K2B_H */
