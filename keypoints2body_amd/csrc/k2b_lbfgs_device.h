// k2b_lbfgs_device.h - the per-frame L-BFGS state machine as device code: one wavefront per frame, one closure result consumed
// per call.  Included by k2b_lbfgs.hip (the stand-alone step kernel) and by k2b_fit.hip (the step as a prologue of the closure's
// own launch).  Algorithm, state layout and references: k2b_lbfgs.hip.
#pragma once
#include "k2b_internal.h"

// No FMA contraction in the optimiser: its code is inlined into three kernels (the step kernel, the closure's prologue, the
// persistent loop), and what the compiler fuses depends on the context - a one-ulp difference in a search direction is enough for
// the line searches to part ways.  Separate multiplies and adds are also what the float32 tensors of torch (and the numpy twin) do.
#pragma clang fp contract(off)

namespace k2b {
namespace lbfgs_dev {

enum { PH_INIT = 0, PH_BRACKET = 1, PH_ZOOM = 2, PH_DONE = 3 };
// per-frame scalars (double) and integers
enum { SD_LOSS, SD_PREV_LOSS, SD_HDIAG, SD_T, SD_T_PREV, SD_F_PREV, SD_GTD_PREV, SD_F0, SD_GTD0, SD_DNORM, SD_BT0, SD_BT1, SD_BF0, SD_BF1,
       SD_BG0, SD_BG1, SD_RO };                      // SD_RO .. SD_RO + H - 1: 1 / (y . s) of the history pairs
enum { SI_PHASE, SI_NOLD, SI_NITER, SI_EVALS, SI_LS_ITER, SI_MAX_LS, SI_LS_EVALS, SI_FIRST, SI_LOW, SI_INSUF, SI_HEAD, SI_COUNT };
// per-frame vectors (float [P] each), then Y [H][P] and S [H][P]
enum { SV_X, SV_G, SV_PREV_G, SV_D, SV_G_PREV, SV_G0, SV_BG0, SV_BG1, SV_HIST };

// sum over the wave in double, the same in every lane: an inclusive DPP scan (five row steps + the lower half's total into the
// upper half, k2b_lanes.h) and lane 63's value.  (Six rounds of `__shfl_xor` on doubles - twelve ds_bpermute round trips - were
// most of a step's time: the two-loop recursion runs two such reductions per history pair, each waiting for the one before.)
__device__ __forceinline__ double wave_sum_d(double v) {
    double s = v;
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
    K2B_DPP_ADD64(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
    K2B_DPP_ADD64(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef K2B_DPP_ADD64
    const long long bits = __builtin_bit_cast(long long, s);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// maximum over the wave of a NON-NEGATIVE value, the same in every lane (lanes a DPP step does not reach contribute 0)
__device__ __forceinline__ float wave_max_f(float v) {
#define K2B_DPP_MAX(ctrl, row_mask) \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, true)))
    K2B_DPP_MAX(0x111, 0xf); K2B_DPP_MAX(0x112, 0xf); K2B_DPP_MAX(0x114, 0xf); K2B_DPP_MAX(0x118, 0xf);
    K2B_DPP_MAX(0x142, 0xa); K2B_DPP_MAX(0x143, 0xc);
#undef K2B_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// torch's _cubic_interpolate on doubles (bounds given, or the interval [min(x1, x2), max(x1, x2)])
__device__ __forceinline__ double cubic(double x1, double f1, double g1, double x2, double f2, double g2, bool bounded, double lo, double hi) {
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
    double loss, prev_loss, hdiag, t, t_prev, f_prev, gtd_prev, f0, gtd0, dnorm, bt0, bt1, bf0, bf1, bg0, bg1;   // (no arrays: a runtime
                                                                                                                   //  index would put them in scratch)
    int phase, nold, niter, evals, ls_iter, max_ls, ls_evals, first, low, insuf, head;
};

constexpr int EPL = 3;                           // vector elements per lane (P <= 192): element e = lane + 64 k
typedef float Vec[EPL];

// A step used to walk through global memory: every vector operation a loop of loads and stores, every decision behind the
// round trip of the one before (9.6 us median, 12.8 us mean per step for one frame: more than the closure's launch).  Now the
// frame's eight state vectors and the closure's gradient are loaded ONCE, all loads in flight together, live in registers (three
// elements per lane) for the whole step and are written back once at its end; the history pairs the two-loop recursion needs are
// staged in LDS by one burst of loads.
struct Frame {
    const LbfgsArgs& a;
    int f, lane, P, H;
    int fp;                                      // row of the parameter arrays the closure evaluates (= f except in a sequence chain)
    double* sd;
    int* si;
    float* sv;
    Scal s;
    Vec V[SV_HIST];                              // SV_X .. SV_BG1
    Vec GN;                                      // the closure's gradient at the evaluated point
    float* lds_hist;                             // [pair][y | s][PL]: staged history (dynamic LDS), PL = 64 * ceil(P / 64)
    double* lds_al;                              // [H] alphas of the two-loop recursion, then [H] rho of the pairs (oldest first)
    int lds_pairs, PL;
    bool resident;                               // the object lives across rounds (persistent launch): LDS keeps the pairs BY RING SLOT, nothing is re-staged
    // (unbound: touches no memory - a kernel may declare the object and bind it only where an optimiser really runs)
    __device__ __forceinline__ explicit Frame(const LbfgsArgs& a_)
        : a(a_), f(0), lane(0), P(0), H(0), fp(0), sd(nullptr), si(nullptr), sv(nullptr), lds_hist(nullptr), lds_al(nullptr), lds_pairs(0), PL(0),
          resident(false) {}
    __device__ __forceinline__ Frame(const LbfgsArgs& a_, int f_, int lane_, float* lds_hist_, double* lds_al_, int lds_pairs_) : Frame(a_) {
        bind(f_, lane_, lds_hist_, lds_al_, lds_pairs_);
    }
    // frame f_ of the batch: pointers, then the 27 scalars of its state
    __device__ __forceinline__ void bind(int f_, int lane_, float* lds_hist_, double* lds_al_, int lds_pairs_) {
        f = f_; fp = f_; lane = lane_; P = a.P; H = a.H;
        sd = a.sd + (size_t)f_ * (SD_RO + a.H); si = a.si + (size_t)f_ * SI_COUNT;
        sv = a.sv + (size_t)f_ * (size_t)(SV_HIST + 2 * a.H) * a.P;
        lds_hist = lds_hist_; lds_al = lds_al_; lds_pairs = lds_pairs_;
        PL = (a.P + 63) / 64 * 64;
        s.loss = sd[SD_LOSS]; s.prev_loss = sd[SD_PREV_LOSS]; s.hdiag = sd[SD_HDIAG]; s.t = sd[SD_T]; s.t_prev = sd[SD_T_PREV];
        s.f_prev = sd[SD_F_PREV]; s.gtd_prev = sd[SD_GTD_PREV]; s.f0 = sd[SD_F0]; s.gtd0 = sd[SD_GTD0]; s.dnorm = sd[SD_DNORM];
        s.bt0 = sd[SD_BT0]; s.bt1 = sd[SD_BT1]; s.bf0 = sd[SD_BF0]; s.bf1 = sd[SD_BF1]; s.bg0 = sd[SD_BG0]; s.bg1 = sd[SD_BG1];
        s.phase = si[SI_PHASE]; s.nold = si[SI_NOLD]; s.niter = si[SI_NITER]; s.evals = si[SI_EVALS]; s.ls_iter = si[SI_LS_ITER];
        s.max_ls = si[SI_MAX_LS]; s.ls_evals = si[SI_LS_EVALS]; s.first = si[SI_FIRST]; s.low = si[SI_LOW]; s.insuf = si[SI_INSUF];
        s.head = si[SI_HEAD];
    }
    // a fresh optimiser on the resident object (the next frame of a sequence chain): phase INIT, empty history, vectors cleared
    __device__ __forceinline__ void restart(int param_row) {
        fp = param_row;
        s = Scal{};
#pragma unroll
        for (int w = 0; w < SV_HIST; ++w)
#pragma unroll
            for (int k = 0; k < EPL; ++k) V[w][k] = 0.f;
    }
    __device__ __forceinline__ bool has(int k) const { return lane + 64 * k < P; }
    // every state vector and the new gradient: 27 independent loads (a frame in phase INIT has no vectors yet: zeros)
    // (requested whatever the phase turns out to be - the scalars above are still in flight -; a frame in phase INIT has no vectors
    //  yet: what was read is dropped)
    __device__ __forceinline__ void load_vectors(const float* g_new) {
        Vec tmp[SV_HIST];
#pragma unroll
        for (int w = 0; w < SV_HIST; ++w)
#pragma unroll
            for (int k = 0; k < EPL; ++k) tmp[w][k] = has(k) ? sv[(size_t)w * P + lane + 64 * k] : 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) GN[k] = has(k) ? g_new[lane + 64 * k] : 0.f;
        const bool fresh = s.phase == PH_INIT;
#pragma unroll
        for (int w = 0; w < SV_HIST; ++w)
#pragma unroll
            for (int k = 0; k < EPL; ++k) V[w][k] = fresh ? 0.f : tmp[w][k];
    }
    __device__ __forceinline__ void save() const {
#pragma unroll
        for (int w = 0; w < SV_HIST; ++w)
#pragma unroll
            for (int k = 0; k < EPL; ++k)
                if (has(k)) sv[(size_t)w * P + lane + 64 * k] = V[w][k];
        if (lane != 0) return;
        sd[SD_LOSS] = s.loss; sd[SD_PREV_LOSS] = s.prev_loss; sd[SD_HDIAG] = s.hdiag; sd[SD_T] = s.t; sd[SD_T_PREV] = s.t_prev;
        sd[SD_F_PREV] = s.f_prev; sd[SD_GTD_PREV] = s.gtd_prev; sd[SD_F0] = s.f0; sd[SD_GTD0] = s.gtd0; sd[SD_DNORM] = s.dnorm;
        sd[SD_BT0] = s.bt0; sd[SD_BT1] = s.bt1; sd[SD_BF0] = s.bf0; sd[SD_BF1] = s.bf1; sd[SD_BG0] = s.bg0; sd[SD_BG1] = s.bg1;
        si[SI_PHASE] = s.phase; si[SI_NOLD] = s.nold; si[SI_NITER] = s.niter; si[SI_EVALS] = s.evals; si[SI_LS_ITER] = s.ls_iter;
        si[SI_MAX_LS] = s.max_ls; si[SI_LS_EVALS] = s.ls_evals; si[SI_FIRST] = s.first; si[SI_LOW] = s.low; si[SI_INSUF] = s.insuf;
        si[SI_HEAD] = s.head;
    }
    __device__ __forceinline__ float* hist_y(int slot) const { return sv + (size_t)(SV_HIST + slot) * P; }
    __device__ __forceinline__ float* hist_s(int slot) const { return sv + (size_t)(SV_HIST + H + slot) * P; }

    // the parameter arrays the closure reads (kernel layout [global_orient | body_pose | betas | transl])
    __device__ __forceinline__ float* eval_ptr(int e) const {
        if (e < 3) return a.go + (size_t)fp * 3 + e;
        if (e < 3 + a.D) return a.bp + (size_t)fp * a.D + (e - 3);
        if (e < 3 + a.D + a.NB) return a.be + (size_t)fp * a.NB + (e - 3 - a.D);
        return a.tr + (size_t)fp * 3 + (e - 3 - a.D - a.NB);
    }
    __device__ __forceinline__ double dot(const Vec& u, const Vec& v) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) acc += (double)u[k] * (double)v[k];       // (elements beyond P are zeros)
        return wave_sum_d(acc);
    }
    __device__ __forceinline__ static void copy(Vec& dst, const Vec& src) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) dst[k] = src[k];
    }
    // x_eval = x + t d (float arithmetic, as the tensors' dtype)
    __device__ __forceinline__ void issue() {
        const float t = (float)s.t;
#pragma unroll
        for (int k = 0; k < EPL; ++k)
            if (has(k)) *eval_ptr(lane + 64 * k) = V[SV_X][k] + t * V[SV_D][k];
        s.ls_evals += 1;
    }
    __device__ __forceinline__ void park() {                     // a finished frame idles at its final point
#pragma unroll
        for (int k = 0; k < EPL; ++k)
            if (has(k)) *eval_ptr(lane + 64 * k) = V[SV_X][k];
    }

    // ---- LBFGS.step: direction, step length and the first line-search evaluation of the next outer iteration ----------------
    __device__ __forceinline__ void start_iteration() {
        s.niter += 1;
        Vec &g = V[SV_G], &d = V[SV_D], &prev_g = V[SV_PREV_G];
        if (s.niter == 1) {
            double s1 = 0.0;
#pragma unroll
            for (int k = 0; k < EPL; ++k) { d[k] = -g[k]; s1 += (double)fabsf(g[k]); }
            s1 = wave_sum_d(s1);
            const double inv = 1.0 / (double)(float)s1;                     // (the sum is a float32 tensor in torch)
            s.t = (inv < 1.0 ? inv : 1.0) * a.lr;
            s.nold = 0; s.head = 0; s.hdiag = 1.0;
        } else {
            // y = g - prev_g, s = d t: the pair joins the history if y . s > 1e-10 (a full ring drops its oldest pair; the candidate
            // lives in registers, so a rejected update leaves the ring untouched)
            const float tf = (float)s.t;
            const bool full = s.nold == H;
            int slot = s.head + s.nold; slot = slot >= H ? slot - H : slot;   // (full: slot == head, the oldest pair's)
            Vec ty, ts;
            double ys = 0.0, yy = 0.0;
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                const float y = g[k] - prev_g[k], sv_ = d[k] * tf;
                ty[k] = y; ts[k] = sv_;
                ys += (double)y * (double)sv_; yy += (double)y * (double)y;
            }
            ys = wave_sum_d(ys); yy = wave_sum_d(yy);
            int new_slot = -1;
            double ro_new = 0.0;
            if (ys > 1e-10) {
                float *hy = hist_y(slot), *hs = hist_s(slot);
#pragma unroll
                for (int k = 0; k < EPL; ++k)
                    if (has(k)) { hy[lane + 64 * k] = ty[k]; hs[lane + 64 * k] = ts[k]; }
                if (full) s.head = s.head + 1 == H ? 0 : s.head + 1;
                else s.nold += 1;
                s.hdiag = ys / yy;
                new_slot = slot; ro_new = 1.0 / ys;
                if (lane == 0) sd[SD_RO + slot] = ro_new;
            }
            // the pairs of the recursion.  One step per launch: oldest first, one burst of loads into LDS (the new pair from registers),
            // rho beside them.  Resident object: LDS holds every pair since it was made, indexed by its ring slot - only the new pair
            // (and its rho) is written.
            const int nold = s.nold, staged = resident ? lds_pairs : (nold < lds_pairs ? nold : lds_pairs);
            double* const lds_ro = lds_al + H;
            if (resident) {
                if (new_slot >= 0) {
                    if (lane == 0) lds_ro[new_slot] = ro_new;
                    if (new_slot < lds_pairs) {
                        float* dst = lds_hist + (size_t)new_slot * 2 * PL;
#pragma unroll
                        for (int k = 0; k < EPL; ++k)
                            if (lane + 64 * k < PL) { dst[lane + 64 * k] = ty[k]; dst[PL + lane + 64 * k] = ts[k]; }
                    }
                }
            } else {
                for (int i = lane; i < nold; i += 64) {
                    int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                    lds_ro[i] = sl == new_slot ? ro_new : sd[SD_RO + sl];
                }
#pragma unroll 4
                for (int i = 0; i < staged; ++i) {
                    int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                    const float *Y = hist_y(sl), *S = hist_s(sl);
                    float* dst = lds_hist + (size_t)i * 2 * PL;
#pragma unroll
                    for (int k = 0; k < EPL; ++k) {
                        if (lane + 64 * k < PL) {
                            const bool in = has(k);
                            dst[lane + 64 * k] = sl == new_slot ? ty[k] : (in ? Y[lane + 64 * k] : 0.f);
                            dst[PL + lane + 64 * k] = sl == new_slot ? ts[k] : (in ? S[lane + 64 * k] : 0.f);
                        }
                    }
                }
            }
            // (LDS index of pair i / ring slot sl: its position in the staged burst, or the slot itself)
            auto lidx = [&](int i, int sl) { return resident ? sl : i; };
            auto pair_y = [&](int i, int sl, int k) -> float {
                const int li = lidx(i, sl);
                return li < staged ? lds_hist[(size_t)li * 2 * PL + lane + 64 * k] : (has(k) ? hist_y(sl)[lane + 64 * k] : 0.f);
            };
            auto pair_s = [&](int i, int sl, int k) -> float {
                const int li = lidx(i, sl);
                return li < staged ? lds_hist[(size_t)li * 2 * PL + PL + lane + 64 * k] : (has(k) ? hist_s(sl)[lane + 64 * k] : 0.f);
            };
            // two-loop recursion: q = -g; backward over the pairs, r = q Hdiag; forward
            Vec q;
#pragma unroll
            for (int k = 0; k < EPL; ++k) q[k] = -g[k];
            for (int i = nold - 1; i >= 0; --i) {
                int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                double p = 0.0;
#pragma unroll
                for (int k = 0; k < EPL; ++k) if (lane + 64 * k < PL) p += (double)pair_s(i, sl, k) * (double)q[k];
                const double ali = wave_sum_d(p) * lds_ro[lidx(i, sl)];
                if (lane == 0) lds_al[i] = ali;
                const float af = (float)ali;
#pragma unroll
                for (int k = 0; k < EPL; ++k) if (lane + 64 * k < PL) q[k] = q[k] - af * pair_y(i, sl, k);
            }
            const float hf = (float)s.hdiag;
#pragma unroll
            for (int k = 0; k < EPL; ++k) q[k] = q[k] * hf;
            for (int i = 0; i < nold; ++i) {
                int sl = s.head + i; sl = sl >= H ? sl - H : sl;
                double p = 0.0;
#pragma unroll
                for (int k = 0; k < EPL; ++k) if (lane + 64 * k < PL) p += (double)pair_y(i, sl, k) * (double)q[k];
                const double be = wave_sum_d(p) * lds_ro[lidx(i, sl)];
                const float cf = (float)(lds_al[i] - be);
#pragma unroll
                for (int k = 0; k < EPL; ++k) if (lane + 64 * k < PL) q[k] = q[k] + cf * pair_s(i, sl, k);
            }
            copy(d, q);
            s.t = a.lr;
        }
        // prev_g = g, prev_loss = loss; directional derivative
        double gtd = 0.0;
        float dn = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) { prev_g[k] = g[k]; gtd += (double)g[k] * (double)d[k]; dn = fmaxf(dn, fabsf(d[k])); }
        gtd = wave_sum_d(gtd);
        dn = wave_max_f(dn);
        s.prev_loss = s.loss;
        if (!(gtd <= -a.tol_c)) {                // "gtd > -tolerance_change" (NaN stops too)
            s.phase = PH_DONE;
            park();
            return;
        }
        // strong-Wolfe line search from x along d: first evaluation at the initial step
        copy(V[SV_G0], g);
        copy(V[SV_G_PREV], g);
        s.f0 = s.loss; s.gtd0 = gtd; s.dnorm = (double)dn;
        s.max_ls = a.max_eval - s.evals;
        s.t_prev = 0.0; s.f_prev = s.loss; s.gtd_prev = gtd;
        s.ls_iter = 0; s.ls_evals = 0; s.first = 1; s.insuf = 0;
        s.phase = PH_BRACKET;
        issue();
    }

    // ---- line search over (step t, loss fv, gradient gsrc there): take the step, run LBFGS.step's checks -----------------------
    __device__ __forceinline__ void finish_line_search(double t, double fv, const Vec& gsrc_in) {
        Vec gsrc;
        copy(gsrc, gsrc_in);                     // (it may be one of the vectors overwritten below)
        Vec &x = V[SV_X], &d = V[SV_D], &g = V[SV_G];
        const float tf = (float)t;
        float gm = 0.f, sm = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            const float ge = gsrc[k];
            x[k] = x[k] + tf * d[k];
            g[k] = ge;
            gm = fmaxf(gm, fabsf(ge));
            sm = fmaxf(sm, fabsf(d[k] * tf));
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
    __device__ __forceinline__ void zoom_next() {
        const double width = fabs(s.bt1 - s.bt0);
        if (s.ls_iter >= s.max_ls || width * s.dnorm < a.tol_c) {
            const int lo = s.low;
            Vec gl;
#pragma unroll
            for (int k = 0; k < EPL; ++k) gl[k] = lo ? V[SV_BG1][k] : V[SV_BG0][k];
            finish_line_search(lo ? s.bt1 : s.bt0, lo ? s.bf1 : s.bf0, gl);     // (selects, not indexed: the scalars stay in registers)
            return;
        }
        double t = cubic(s.bt0, s.bf0, s.bg0, s.bt1, s.bf1, s.bg1, false, 0.0, 0.0);
        const double hi = s.bt0 > s.bt1 ? s.bt0 : s.bt1, lo = s.bt0 < s.bt1 ? s.bt0 : s.bt1;
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
    __device__ __forceinline__ void bracket(double f_new) {
        const double c1 = 1e-4, c2 = 0.9;
        s.ls_iter += s.first ? 0 : 1;                                 // the first evaluation precedes the loop
        s.first = 0;
        const double t = s.t, f0 = s.f0, gtd0 = s.gtd0;
        const double gtd_new = dot(GN, V[SV_D]);
        if (s.ls_iter >= s.max_ls) {                                  // "ls_iter == max_ls": bracket = [0, t], no zoom
            const bool lower0 = f0 <= f_new;
            Vec gl;
#pragma unroll
            for (int k = 0; k < EPL; ++k) gl[k] = lower0 ? V[SV_G0][k] : GN[k];
            finish_line_search(lower0 ? 0.0 : t, lower0 ? f0 : f_new, gl);
            return;
        }
        const bool armijo = (f_new > f0 + c1 * t * gtd0) || (s.ls_iter > 1 && f_new >= s.f_prev);
        const bool wolfe = !armijo && fabs(gtd_new) <= -c2 * gtd0;
        const bool uphill = !armijo && !wolfe && gtd_new >= 0.0;
        if (wolfe) { finish_line_search(t, f_new, GN); return; }
        if (armijo || uphill) {                                       // bracket [t_prev, t] found: zoom
            copy(V[SV_BG0], V[SV_G_PREV]);
            copy(V[SV_BG1], GN);
            s.bt0 = s.t_prev; s.bt1 = t; s.bf0 = s.f_prev; s.bf1 = f_new; s.bg0 = s.gtd_prev; s.bg1 = gtd_new;
            s.low = s.f_prev <= f_new ? 0 : 1;
            s.insuf = 0;
            s.phase = PH_ZOOM;
            zoom_next();
            return;
        }
        // extrapolate
        const double t_next = cubic(s.t_prev, s.f_prev, s.gtd_prev, t, f_new, gtd_new, true, t + 0.01 * (t - s.t_prev), t * 10.0);
        copy(V[SV_G_PREV], GN);
        s.t_prev = t; s.f_prev = f_new; s.gtd_prev = gtd_new; s.t = t_next;
        issue();
    }

    // ---- _strong_wolfe: zoom phase receives an evaluation ----------------------------------------------------------------------------
    __device__ __forceinline__ void zoom_receive(double f_new) {
        const double c1 = 1e-4, c2 = 0.9;
        s.ls_iter += 1;
        const double t = s.t, f0 = s.f0, gtd0 = s.gtd0;
        const double gtd_new = dot(GN, V[SV_D]);
        const int low = s.low, high = 1 - low;
#define K2B_GET2(n, i) ((i) ? s.n##1 : s.n##0)
#define K2B_SET2(n, i, x) do { const double k2b_x = (x); s.n##1 = (i) ? k2b_x : s.n##1; s.n##0 = (i) ? s.n##0 : k2b_x; } while (0)   /* value selects: a conditional store becomes an indexed one, i.e. scratch */
        const bool worse = (f_new > f0 + c1 * t * gtd0) || (f_new >= K2B_GET2(bf, low));
        bool wolfe = false;
        auto set_end = [&](int which, const Vec& src) {               // bracket gradient `which` (0 / 1) := src
            Vec tmp;
            copy(tmp, src);
#pragma unroll
            for (int k = 0; k < EPL; ++k) { V[SV_BG1][k] = which ? tmp[k] : V[SV_BG1][k]; V[SV_BG0][k] = which ? V[SV_BG0][k] : tmp[k]; }
        };
        if (worse) {                              // Armijo violated or not below the lowest point: the trial replaces the HIGH end
            K2B_SET2(bt, high, t); K2B_SET2(bf, high, f_new); K2B_SET2(bg, high, gtd_new);
            set_end(high, GN);
            s.low = s.bf0 <= s.bf1 ? 0 : 1;
        } else {
            wolfe = fabs(gtd_new) <= -c2 * gtd0;
            if (!wolfe && gtd_new * (K2B_GET2(bt, high) - K2B_GET2(bt, low)) >= 0.0) {     // the old low becomes the high end
                K2B_SET2(bt, high, K2B_GET2(bt, low)); K2B_SET2(bf, high, K2B_GET2(bf, low)); K2B_SET2(bg, high, K2B_GET2(bg, low));
                Vec gl;
#pragma unroll
                for (int k = 0; k < EPL; ++k) gl[k] = low ? V[SV_BG1][k] : V[SV_BG0][k];
                set_end(high, gl);
            }
            K2B_SET2(bt, low, t); K2B_SET2(bf, low, f_new); K2B_SET2(bg, low, gtd_new);
            set_end(low, GN);
        }
#undef K2B_GET2
#undef K2B_SET2
        if (wolfe) { finish_line_search(t, f_new, GN); return; }
        zoom_next();
    }
};

// One call = one closure result consumed per frame.  `finalize`: no result is consumed; every frame's ACCEPTED point goes into the
// parameter arrays (frames still in a line search when the rounds run out fall back to it), for the final loss evaluation.
// lds = 2 H doubles (alphas, rho) followed by lds_pairs staged history pairs of 2 x 64 ceil(P / 64) floats, private to the wave.
// one closure result (loss f_new, gradient in fr.GN) through the state machine
__device__ __forceinline__ void lbfgs_consume(Frame& fr, double f_new) {
    const LbfgsArgs& a = fr.a;
    const int lane = fr.lane;
    if (fr.s.phase == PH_INIT) {
        // x = the start (already in the parameter arrays), first closure result
        float gm = 0.f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            fr.V[SV_X][k] = fr.has(k) ? *fr.eval_ptr(lane + 64 * k) : 0.f;
            fr.V[SV_G][k] = fr.GN[k];
            gm = fmaxf(gm, fabsf(fr.GN[k]));
        }
        gm = wave_max_f(gm);
        fr.s.loss = f_new; fr.s.evals = 1; fr.s.niter = 0;
        if ((double)gm <= a.tol_g) fr.s.phase = PH_DONE;
        else fr.start_iteration();
    } else if (fr.s.phase == PH_BRACKET) {
        fr.bracket(f_new);
    } else if (fr.s.phase == PH_ZOOM) {
        fr.zoom_receive(f_new);
    }
}

__device__ __forceinline__ void lbfgs_step_frame(const LbfgsArgs& a, int f, int lane, unsigned char* lds, int lds_pairs) {
    double* lds_al = reinterpret_cast<double*>(lds);
    float* lds_hist = reinterpret_cast<float*>(lds + (size_t)2 * a.H * sizeof(double));
    Frame fr(a, f, lane, lds_hist, lds_al, lds_pairs);
    fr.load_vectors(a.grad_in + (size_t)f * a.P);
    if (a.finalize) { if (fr.s.phase != PH_INIT) fr.park(); return; }
    if (fr.s.phase == PH_DONE) return;
    lbfgs_consume(fr, (double)a.loss_in[f]);
    fr.save();
}

// bytes of LDS a frame's step wants for `pairs` staged history pairs
__host__ __device__ inline size_t lbfgs_lds_bytes(int H, int P, int pairs) {
    return (size_t)2 * H * sizeof(double) + (size_t)pairs * 2 * ((P + 63) / 64 * 64) * sizeof(float);
}

}  // namespace lbfgs_dev
}  // namespace k2b

#pragma clang fp contract(fast)      // (the translation units are built with -ffp-contract=fast)
