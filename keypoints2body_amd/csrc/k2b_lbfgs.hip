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
//   * one wavefront per frame: vectors (<= 192 parameters) three elements per lane in global memory (history: 2 H P floats),
//     inner products reduced over the wave in double, scalars in double where torch holds Python floats;
//   * the host only ENQUEUES a fixed number of [evaluate-only fit launch, step launch] rounds - max_eval + 2, the most any frame
//     can need - without reading anything back: frames that finish early idle at their final point (their evaluations are
//     ignored).  No host synchronisation, no PCIe traffic, any number of frames per launch.
//
// CPU twin: keypoints2body_amd/core/lbfgs_batched.py (itself pinned to torch.optim.LBFGS iterate by iterate in float64,
// tests/test_lbfgs_batched.py); tests/test_gpu_lbfgs.py compares the two on the real closure.
#include "k2b_internal.h"

namespace k2b {

namespace {

enum { PH_INIT = 0, PH_BRACKET = 1, PH_ZOOM = 2, PH_DONE = 3 };
// per-frame scalars (double) and integers
enum { SD_LOSS, SD_PREV_LOSS, SD_HDIAG, SD_T, SD_T_PREV, SD_F_PREV, SD_GTD_PREV, SD_F0, SD_GTD0, SD_DNORM, SD_BT0, SD_BT1, SD_BF0, SD_BF1,
       SD_BG0, SD_BG1, SD_RO };                      // SD_RO .. SD_RO + H - 1: 1 / (y . s) of the history pairs
enum { SI_PHASE, SI_NOLD, SI_NITER, SI_EVALS, SI_LS_ITER, SI_MAX_LS, SI_LS_EVALS, SI_FIRST, SI_LOW, SI_INSUF, SI_HEAD, SI_COUNT };
// per-frame vectors (float [P] each), then Y [H][P] and S [H][P]
enum { SV_X, SV_G, SV_PREV_G, SV_D, SV_G_PREV, SV_G0, SV_BG0, SV_BG1, SV_HIST };

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// torch's _cubic_interpolate on doubles (bounds given, or the interval [min(x1, x2), max(x1, x2)])
__device__ double cubic(double x1, double f1, double g1, double x2, double f2, double g2, bool bounded, double lo, double hi) {
    if (!bounded) { lo = x1 <= x2 ? x1 : x2; hi = x1 <= x2 ? x2 : x1; }
    const double d1 = g1 + g2 - 3.0 * (f1 - f2) / (x1 - x2);
    const double sq = d1 * d1 - g1 * g2;
    if (!(sq >= 0.0)) return 0.5 * (lo + hi);
    const double d2 = sqrt(sq);
    double pos = x1 <= x2 ? x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2.0 * d2)) : x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2.0 * d2));
    // min(max(pos, lo), hi) with Python's rules: a NaN position stays NaN (every comparison with it is false)
    pos = lo > pos ? lo : pos;
    pos = hi < pos ? hi : pos;
    return pos;
}

// wave-uniform scalars of one frame: loaded into registers at the start of a step, written back by lane 0 at its end (never
// re-read from memory inside a step: a value lane 0 has just stored is not guaranteed visible to the other lanes' loads)
struct Scal {
    double loss, prev_loss, hdiag, t, t_prev, f_prev, gtd_prev, f0, gtd0, dnorm, bt[2], bf[2], bg[2];
    int phase, nold, niter, evals, ls_iter, max_ls, ls_evals, first, low, insuf, head;
};

struct Frame {
    const LbfgsArgs& a;
    const int f, lane, P, H;
    double* sd;
    int* si;
    float* sv;
    Scal s;
    __device__ Frame(const LbfgsArgs& a_, int f_, int lane_)
        : a(a_), f(f_), lane(lane_), P(a_.P), H(a_.H), sd(a_.sd + (size_t)f_ * (SD_RO + a_.H)), si(a_.si + (size_t)f_ * SI_COUNT),
          sv(a_.sv + (size_t)f_ * (size_t)(SV_HIST + 2 * a_.H) * a_.P) {
        s.loss = sd[SD_LOSS]; s.prev_loss = sd[SD_PREV_LOSS]; s.hdiag = sd[SD_HDIAG]; s.t = sd[SD_T]; s.t_prev = sd[SD_T_PREV];
        s.f_prev = sd[SD_F_PREV]; s.gtd_prev = sd[SD_GTD_PREV]; s.f0 = sd[SD_F0]; s.gtd0 = sd[SD_GTD0]; s.dnorm = sd[SD_DNORM];
        s.bt[0] = sd[SD_BT0]; s.bt[1] = sd[SD_BT1]; s.bf[0] = sd[SD_BF0]; s.bf[1] = sd[SD_BF1]; s.bg[0] = sd[SD_BG0]; s.bg[1] = sd[SD_BG1];
        s.phase = si[SI_PHASE]; s.nold = si[SI_NOLD]; s.niter = si[SI_NITER]; s.evals = si[SI_EVALS]; s.ls_iter = si[SI_LS_ITER];
        s.max_ls = si[SI_MAX_LS]; s.ls_evals = si[SI_LS_EVALS]; s.first = si[SI_FIRST]; s.low = si[SI_LOW]; s.insuf = si[SI_INSUF];
        s.head = si[SI_HEAD];
    }
    __device__ void save() const {
        if (lane != 0) return;
        sd[SD_LOSS] = s.loss; sd[SD_PREV_LOSS] = s.prev_loss; sd[SD_HDIAG] = s.hdiag; sd[SD_T] = s.t; sd[SD_T_PREV] = s.t_prev;
        sd[SD_F_PREV] = s.f_prev; sd[SD_GTD_PREV] = s.gtd_prev; sd[SD_F0] = s.f0; sd[SD_GTD0] = s.gtd0; sd[SD_DNORM] = s.dnorm;
        sd[SD_BT0] = s.bt[0]; sd[SD_BT1] = s.bt[1]; sd[SD_BF0] = s.bf[0]; sd[SD_BF1] = s.bf[1]; sd[SD_BG0] = s.bg[0]; sd[SD_BG1] = s.bg[1];
        si[SI_PHASE] = s.phase; si[SI_NOLD] = s.nold; si[SI_NITER] = s.niter; si[SI_EVALS] = s.evals; si[SI_LS_ITER] = s.ls_iter;
        si[SI_MAX_LS] = s.max_ls; si[SI_LS_EVALS] = s.ls_evals; si[SI_FIRST] = s.first; si[SI_LOW] = s.low; si[SI_INSUF] = s.insuf;
        si[SI_HEAD] = s.head;
    }
    // vectors: element e always belongs to lane e % 64, so a lane only ever reads back what it wrote itself
    __device__ float* vec(int which) const { return sv + (size_t)which * P; }
    __device__ float* hist_y(int slot) const { return sv + (size_t)(SV_HIST + slot) * P; }
    __device__ float* hist_s(int slot) const { return sv + (size_t)(SV_HIST + H + slot) * P; }

    // the parameter arrays the closure reads (kernel layout [global_orient | body_pose | betas | transl])
    __device__ float* eval_ptr(int e) const {
        if (e < 3) return a.go + (size_t)f * 3 + e;
        if (e < 3 + a.D) return a.bp + (size_t)f * a.D + (e - 3);
        if (e < 3 + a.D + a.NB) return a.be + (size_t)f * a.NB + (e - 3 - a.D);
        return a.tr + (size_t)f * 3 + (e - 3 - a.D - a.NB);
    }
    __device__ double dot(const float* u, const float* v) const {
        double acc = 0.0;
        for (int e = lane; e < P; e += 64) acc += (double)u[e] * (double)v[e];
        return wave_sum_d(acc);
    }
    __device__ void copy(float* dst, const float* src) const {
        for (int e = lane; e < P; e += 64) dst[e] = src[e];
    }
    // x_eval = x + t d (float arithmetic, as the tensors' dtype)
    __device__ void issue() {
        const float t = (float)s.t;
        const float *x = vec(SV_X), *d = vec(SV_D);
        for (int e = lane; e < P; e += 64) *eval_ptr(e) = x[e] + t * d[e];
        s.ls_evals += 1;
    }
    __device__ void park() {                     // a finished frame idles at its final point
        const float* x = vec(SV_X);
        for (int e = lane; e < P; e += 64) *eval_ptr(e) = x[e];
    }

    // ---- LBFGS.step: direction, step length and the first line-search evaluation of the next outer iteration ----------------
    __device__ void start_iteration() {
        s.niter += 1;
        float *g = vec(SV_G), *d = vec(SV_D), *prev_g = vec(SV_PREV_G);
        if (s.niter == 1) {
            double s1 = 0.0;
            for (int e = lane; e < P; e += 64) { d[e] = -g[e]; s1 += (double)fabsf(g[e]); }
            s1 = wave_sum_d(s1);
            const double inv = 1.0 / (double)(float)s1;                     // (the sum is a float32 tensor in torch)
            s.t = (inv < 1.0 ? inv : 1.0) * a.lr;
            s.nold = 0; s.head = 0; s.hdiag = 1.0;
        } else {
            // y = g - prev_g, s = d t: the pair joins the history if y . s > 1e-10 (a full ring drops its oldest pair)
            const float tf = (float)s.t;
            const bool full = s.nold == H;
            int slot = s.head + s.nold; slot = slot >= H ? slot - H : slot;   // (full: slot == head, the oldest pair's)
            // (a full ring: the candidate is formed in the bracket's scratch vectors first - the oldest pair must survive a rejected update)
            float *ty = full ? vec(SV_BG0) : hist_y(slot), *ts = full ? vec(SV_BG1) : hist_s(slot);
            double ys = 0.0, yy = 0.0;
            for (int e = lane; e < P; e += 64) {
                const float y = g[e] - prev_g[e], sv_ = d[e] * tf;
                ty[e] = y; ts[e] = sv_;
                ys += (double)y * (double)sv_; yy += (double)y * (double)y;
            }
            ys = wave_sum_d(ys); yy = wave_sum_d(yy);
            int new_slot = -1;
            double ro_new = 0.0;
            if (ys > 1e-10) {
                if (full) {
                    copy(hist_y(slot), ty); copy(hist_s(slot), ts);
                    s.head = s.head + 1 == H ? 0 : s.head + 1;
                } else {
                    s.nold += 1;
                }
                s.hdiag = ys / yy;
                new_slot = slot; ro_new = 1.0 / ys;
                if (lane == 0) sd[SD_RO + slot] = ro_new;
            }
            // two-loop recursion: q = -g; backward over the pairs, r = q Hdiag; forward
            float q[3];
            for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; q[k] = e < P ? -g[e] : 0.f; }
            double al[kLbfgsMaxHistory];
            for (int i = s.nold - 1; i >= 0; --i) {
                int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                const float* S = hist_s(sl); const float* Y = hist_y(sl);
                double p = 0.0;
                for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; if (e < P) p += (double)S[e] * (double)q[k]; }
                const double ro = sl == new_slot ? ro_new : sd[SD_RO + sl];
                const double ali = wave_sum_d(p) * ro;
                al[i] = ali;
                const float af = (float)ali;
                for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; if (e < P) q[k] = q[k] - af * Y[e]; }
            }
            const float hf = (float)s.hdiag;
            for (int k = 0; k < 3; ++k) q[k] = q[k] * hf;
            for (int i = 0; i < s.nold; ++i) {
                int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                const float* S = hist_s(sl); const float* Y = hist_y(sl);
                double p = 0.0;
                for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; if (e < P) p += (double)Y[e] * (double)q[k]; }
                const double ro = sl == new_slot ? ro_new : sd[SD_RO + sl];
                const double be = wave_sum_d(p) * ro;
                const float cf = (float)(al[i] - be);
                for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; if (e < P) q[k] = q[k] + cf * S[e]; }
            }
            for (int k = 0; k < 3; ++k) { const int e = lane + 64 * k; if (e < P) d[e] = q[k]; }
            s.t = a.lr;
        }
        // prev_g = g, prev_loss = loss; directional derivative
        double gtd = 0.0;
        float dn = 0.f;
        for (int e = lane; e < P; e += 64) { prev_g[e] = g[e]; gtd += (double)g[e] * (double)d[e]; dn = fmaxf(dn, fabsf(d[e])); }
        gtd = wave_sum_d(gtd);
        dn = wave_max_f(dn);
        s.prev_loss = s.loss;
        if (!(gtd <= -a.tol_c)) {                // "gtd > -tolerance_change" (NaN stops too)
            s.phase = PH_DONE;
            park();
            return;
        }
        // strong-Wolfe line search from x along d: first evaluation at the initial step
        copy(vec(SV_G0), g);
        copy(vec(SV_G_PREV), g);
        s.f0 = s.loss; s.gtd0 = gtd; s.dnorm = (double)dn;
        s.max_ls = a.max_eval - s.evals;
        s.t_prev = 0.0; s.f_prev = s.loss; s.gtd_prev = gtd;
        s.ls_iter = 0; s.ls_evals = 0; s.first = 1; s.insuf = 0;
        s.phase = PH_BRACKET;
        issue();
    }

    // ---- line search over (step t, loss fv, gradient gsrc there): take the step, run LBFGS.step's checks -----------------------
    __device__ void finish_line_search(double t, double fv, const float* gsrc) {
        float *x = vec(SV_X), *d = vec(SV_D), *g = vec(SV_G);
        const float tf = (float)t;
        float gm = 0.f, sm = 0.f;
        for (int e = lane; e < P; e += 64) {
            const float ge = gsrc[e];
            x[e] = x[e] + tf * d[e];
            g[e] = ge;
            gm = fmaxf(gm, fabsf(ge));
            sm = fmaxf(sm, fabsf(d[e] * tf));
        }
        gm = wave_max_f(gm); sm = wave_max_f(sm);
        s.t = t; s.loss = fv; s.evals += s.ls_evals;
        bool stop = s.niter >= a.max_iter || s.evals >= a.max_eval || (double)gm <= a.tol_g || (double)sm <= a.tol_c ||
                    fabs(fv - s.prev_loss) < a.tol_c;
        stop = stop || !(fabs(fv) <= 1.79e308);                      // not finite
        if (stop) {
            s.phase = PH_DONE;
            park();
            return;
        }
        start_iteration();
    }

    // ---- _strong_wolfe: zoom phase, loop head ---------------------------------------------------------------------------------
    __device__ void zoom_next() {
        const double width = fabs(s.bt[1] - s.bt[0]);
        if (s.ls_iter >= s.max_ls || width * s.dnorm < a.tol_c) {
            const int lo = s.low;
            finish_line_search(s.bt[lo], s.bf[lo], vec(lo ? SV_BG1 : SV_BG0));
            return;
        }
        double t = cubic(s.bt[0], s.bf[0], s.bg[0], s.bt[1], s.bf[1], s.bg[1], false, 0.0, 0.0);
        const double hi = s.bt[0] > s.bt[1] ? s.bt[0] : s.bt[1], lo = s.bt[0] < s.bt[1] ? s.bt[0] : s.bt[1];
        const double eps = 0.1 * (hi - lo);
        const double dmin = (hi - t) < (t - lo) ? (hi - t) : (t - lo);
        const bool near = dmin < eps;
        const bool move = near && (s.insuf || t >= hi || t <= lo);
        if (move) t = fabs(t - hi) < fabs(t - lo) ? hi - eps : lo + eps;
        s.insuf = (near && !move) ? 1 : 0;
        s.t = t;
        issue();
    }

    // ---- _strong_wolfe: bracket phase receives an evaluation ---------------------------------------------------------------------
    __device__ void bracket(double f_new, const float* g_new) {
        const double c1 = 1e-4, c2 = 0.9;
        s.ls_iter += s.first ? 0 : 1;                                 // the first evaluation precedes the loop
        s.first = 0;
        const double t = s.t, f0 = s.f0, gtd0 = s.gtd0;
        const double gtd_new = dot(g_new, vec(SV_D));
        if (s.ls_iter >= s.max_ls) {                                  // "ls_iter == max_ls": bracket = [0, t], no zoom
            const bool lower0 = f0 <= f_new;
            finish_line_search(lower0 ? 0.0 : t, lower0 ? f0 : f_new, lower0 ? vec(SV_G0) : g_new);
            return;
        }
        const bool armijo = (f_new > f0 + c1 * t * gtd0) || (s.ls_iter > 1 && f_new >= s.f_prev);
        const bool wolfe = !armijo && fabs(gtd_new) <= -c2 * gtd0;
        const bool uphill = !armijo && !wolfe && gtd_new >= 0.0;
        if (wolfe) { finish_line_search(t, f_new, g_new); return; }
        if (armijo || uphill) {                                       // bracket [t_prev, t] found: zoom
            copy(vec(SV_BG0), vec(SV_G_PREV));
            copy(vec(SV_BG1), g_new);
            s.bt[0] = s.t_prev; s.bt[1] = t; s.bf[0] = s.f_prev; s.bf[1] = f_new; s.bg[0] = s.gtd_prev; s.bg[1] = gtd_new;
            s.low = s.f_prev <= f_new ? 0 : 1;
            s.insuf = 0;
            s.phase = PH_ZOOM;
            zoom_next();
            return;
        }
        // extrapolate
        const double t_next = cubic(s.t_prev, s.f_prev, s.gtd_prev, t, f_new, gtd_new, true, t + 0.01 * (t - s.t_prev), t * 10.0);
        copy(vec(SV_G_PREV), g_new);
        s.t_prev = t; s.f_prev = f_new; s.gtd_prev = gtd_new; s.t = t_next;
        issue();
    }

    // ---- _strong_wolfe: zoom phase receives an evaluation ----------------------------------------------------------------------------
    __device__ void zoom_receive(double f_new, const float* g_new) {
        const double c1 = 1e-4, c2 = 0.9;
        s.ls_iter += 1;
        const double t = s.t, f0 = s.f0, gtd0 = s.gtd0;
        const double gtd_new = dot(g_new, vec(SV_D));
        const int low = s.low, high = 1 - low;
        const bool worse = (f_new > f0 + c1 * t * gtd0) || (f_new >= s.bf[low]);
        bool wolfe = false;
        if (worse) {                              // Armijo violated or not below the lowest point: the trial replaces the HIGH end
            s.bt[high] = t; s.bf[high] = f_new; s.bg[high] = gtd_new;
            copy(vec(high ? SV_BG1 : SV_BG0), g_new);
            s.low = s.bf[0] <= s.bf[1] ? 0 : 1;
        } else {
            wolfe = fabs(gtd_new) <= -c2 * gtd0;
            if (!wolfe && gtd_new * (s.bt[high] - s.bt[low]) >= 0.0) {     // the old low becomes the high end
                s.bt[high] = s.bt[low]; s.bf[high] = s.bf[low]; s.bg[high] = s.bg[low];
                copy(vec(high ? SV_BG1 : SV_BG0), vec(low ? SV_BG1 : SV_BG0));
            }
            s.bt[low] = t; s.bf[low] = f_new; s.bg[low] = gtd_new;
            copy(vec(low ? SV_BG1 : SV_BG0), g_new);
        }
        if (wolfe) { finish_line_search(t, f_new, g_new); return; }
        zoom_next();
    }
};

}  // namespace

// One call = one closure result consumed per frame.  `finalize`: no result is consumed; every frame's ACCEPTED point goes into the
// parameter arrays (frames still in a line search when the rounds run out fall back to it), for the final loss evaluation.
__global__ __launch_bounds__(64) void k2b_lbfgs_step_kernel(const LbfgsArgs a) {
    const int f = blockIdx.x, lane = threadIdx.x;
    Frame fr(a, f, lane);
    if (a.finalize) { if (fr.s.phase != PH_INIT) fr.park(); return; }
    if (fr.s.phase == PH_DONE) return;
    const double f_new = (double)a.loss_in[f];
    const float* g_new = a.grad_in + (size_t)f * a.P;
    if (fr.s.phase == PH_INIT) {
        // x = the start (already in the parameter arrays), first closure result
        float *x = fr.vec(SV_X), *g = fr.vec(SV_G);
        float gm = 0.f;
        for (int e = lane; e < a.P; e += 64) { x[e] = *fr.eval_ptr(e); g[e] = g_new[e]; gm = fmaxf(gm, fabsf(g_new[e])); }
        gm = wave_max_f(gm);
        fr.s.loss = f_new; fr.s.evals = 1; fr.s.niter = 0;
        if ((double)gm <= a.tol_g) fr.s.phase = PH_DONE;
        else fr.start_iteration();
    } else if (fr.s.phase == PH_BRACKET) {
        fr.bracket(f_new, g_new);
    } else {
        fr.zoom_receive(f_new, g_new);
    }
    fr.save();
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
    hipLaunchKernelGGL(k2b_lbfgs_step_kernel, dim3(a.B), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace k2b
