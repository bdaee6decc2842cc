// k2b_lbfgs.hip — device-resident L-BFGS with strong-Wolfe line search, one independent optimiser per frame.
//
// The reference's DEFAULT optimiser is `torch.optim.LBFGS(params, max_iter=num_iters, lr=step_size,
// line_search_fn="strong_wolfe").step(closure)` per frame (keypoints2body/core/fitters/world_space.py:231-247,
// core/fitters/camera_space.py:144-182,229-267; core/config.py:29 `use_lbfgs=True`).  Its closure - loss and gradient at the
// current parameters - is the fused fit kernel in evaluate-only mode.  Rounds 2-3 drove the optimiser from the host (torch's own
// class for one frame: an upload, a launch and a download per closure call, 6.9 ms per fit; a vectorised numpy restatement,
// core/lbfgs_batched.py, for batches).  Here the optimiser's state machine itself runs on the device:
//
//   * torch's algorithm (LBFGS.step, _strong_wolfe, _cubic_interpolate: two-loop recursion, bracket phase, zoom phase with its
//     insufficient-progress rule, the tolerance / max_iter / max_eval exits) restated as a per-frame state machine
//     INIT -> (BRACKET | ZOOM)* -> DONE that consumes ONE closure result per call and names the next point to evaluate;
//   * one wavefront per frame (k2b_lbfgs_device.h): vectors (<= 192 parameters) three elements per lane, in registers for the
//     whole step; state and history (2 H P floats) in global memory between steps, the history staged in LDS for the two-loop
//     recursion; inner products reduced over the wave in double (DPP scan), scalars in double where torch holds Python floats;
//     no FMA contraction (the code is inlined into three kernels and must round alike in each);
//   * the host only ENQUEUES a fixed number of [closure, step] rounds - max_eval + 2, the most any frame can need - without
//     reading anything back: frames that finish early idle at their final point.  No host synchronisation, no PCIe traffic.
//     THREE ways the rounds reach the device, by the batch size, bit-identical (tests/test_gpu_lbfgs.py):
//       - at most two frames per CU: ONE persistent launch of the fused fit kernel (k2b_fit.hip, lb_mode 3) - the closures are
//         iterations of its loop, the optimiser lives on an idle wave of the workgroup with its state resident; with chain_len
//         frames the same launch runs the whole warm-start sequence (k2b_fit_sequence_lbfgs);
//       - at most four: one launch per round, the step as a prologue of the closure's launch (lb_mode 1 / 2);
//       - beyond, and for the larger models: two launches per round - the closure's, and k2b_lbfgs_step_kernel below.
//
// CPU twin: keypoints2body_amd/core/lbfgs_batched.py (itself pinned to torch.optim.LBFGS iterate by iterate in float64,
// tests/test_lbfgs_batched.py); tests/test_gpu_lbfgs.py compares the two on the real closure.
#include "k2b_internal.h"
#include "k2b_lbfgs_device.h"

namespace k2b {

using namespace lbfgs_dev;

__global__ __launch_bounds__(64) void k2b_lbfgs_step_kernel(const LbfgsArgs a, int lds_pairs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lbfgs_lds[];
    lbfgs_step_frame(a, blockIdx.x, threadIdx.x, lbfgs_lds, lds_pairs);
}

// start of one frame of a warm-start sequence: parameters and preserve pose from their sources (the previous frame's result),
// the optimiser's scalars and integers cleared (phase INIT) - one launch instead of five copies and a memset
__global__ __launch_bounds__(256) void k2b_lbfgs_frame_prep_kernel(float* go, const float* sgo, float* bp, const float* sbp, float* be,
                                                                   const float* sbe, float* tr, const float* str, float* pres, int D, int NB,
                                                                   unsigned int* state, int state_words) {
    const int i = threadIdx.x;
    for (int e = i; e < D; e += 256) { const float v = sbp[e]; bp[e] = v; pres[e] = v; }
    for (int e = i; e < NB; e += 256) be[e] = sbe[e];
    if (i < 3) { go[i] = sgo[i]; tr[i] = str[i]; }
    for (int e = i; e < state_words; e += 256) state[e] = 0u;
}

hipError_t launch_lbfgs_frame_prep(float* go, const float* sgo, float* bp, const float* sbp, float* be, const float* sbe, float* tr,
                                   const float* str, float* pres, int D, int NB, void* state, size_t state_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(k2b_lbfgs_frame_prep_kernel, dim3(1), dim3(256), 0, stream, go, sgo, bp, sbp, be, sbe, tr, str, pres, D, NB,
                       reinterpret_cast<unsigned int*>(state), (int)(state_bytes / 4));
    return hipGetLastError();
}

size_t lbfgs_state_bytes(int B, int P, int H, size_t* off_si, size_t* off_sv) {
    size_t n = (size_t)B * (SD_RO + H) * sizeof(double);
    *off_si = n;
    n += (size_t)B * SI_COUNT * sizeof(int);
    n = (n + 15) / 16 * 16;
    *off_sv = n;
    n += (size_t)B * (SV_HIST + 2 * (size_t)H) * P * sizeof(float);
    return n;
}

hipError_t launch_lbfgs_step(const LbfgsArgs& a, hipStream_t stream) {
    if (a.B <= 0) return hipSuccess;
    if (a.P > 192 || a.H < 1 || a.H > kLbfgsMaxHistory) return hipErrorInvalidValue;
    // LDS: the alphas, then as many staged history pairs as a fit can collect (one per outer iteration) - within 48 KiB, so that
    // several frames share a CU; pairs beyond that are read from global memory
    const int PL = (a.P + 63) / 64 * 64;
    int pairs = a.H < a.max_iter ? a.H : a.max_iter;
    const size_t per_pair = (size_t)2 * PL * sizeof(float), head = (size_t)2 * a.H * sizeof(double);
    const size_t room = 48 * 1024 - head;
    if ((size_t)pairs * per_pair > room) pairs = (int)(room / per_pair);
    const size_t lds = head + (size_t)pairs * per_pair;
    hipLaunchKernelGGL(k2b_lbfgs_step_kernel, dim3(a.B), dim3(64), lds, stream, a, pairs);
    return hipGetLastError();
}

}  // namespace k2b
