// k2b_api.hip — the C ABI of libk2b.so (include/k2b.h): handles, host-side table
// preparation, argument validation and kernel launches.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/k2b.h"
#include "k2b_internal.h"


#ifndef K2B_LBS_STREAM
#define K2B_LBS_STREAM 1      // 0: development builds that keep the tile kernel for 17-24 joint models (A/B timing)
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(K2B_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

template <typename T>
hipError_t upload(T** dst, const T* src, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace

struct k2b_model {
    int V = 0, J = 0, NB = 0, E = 0, P = 0;
    float *v_template = nullptr, *shapedirs = nullptr, *posedirs = nullptr, *j_regressor = nullptr,
          *lbs_weights = nullptr;
    int *parents = nullptr, *extra_ids = nullptr;
    float *j_template = nullptr, *j_dirs = nullptr;          // device
    float* j_basis_lane = nullptr;                           // device: [3][1 + NB][64 lanes] = template | directions, lane = joint (pose set-up)
    std::vector<float> h_j_template, h_j_dirs;               // host copies
    // fused-fit tables (J == 24 only)
    bool fit_ok = false;
    std::string fit_why;
    float *dt = nullptr, *dd = nullptr;
    int* tree = nullptr;
    std::vector<int> depth;                                  // depth of every joint (root 0)
    // LBS B operands (f16 hi/lo, MFMA fragment order) for the whole mesh and for the E extra-joint vertices
    struct VertexSet {
        k2b::k2b_half *pdh = nullptr, *pdl = nullptr;
        k2b::k2b_half* w2 = nullptr;                         // W in the tile kernel's group layout (k2b_internal.h, TileArgs)
        k2b::k2b_half *spd = nullptr, *sw = nullptr;         // stream kernel's Pd / W (k2b_internal.h, StreamArgs), or null
        int v_tiles = 0, num = 0, nv16 = 0;
    } mesh, extra;
    bool joints_in_mesh = false;                             // every extra joint's vertex is tagged in mesh.w2 (no gather launch)
    bool stream = false;                                     // 17-24 joints and 7 pose k-steps: the stream kernel skins this model
    bool stream_x = false;                                   // 49-56 joints and 16 pose k-steps (SMPL-X): k2b_lbs_stream_x_kernel
    // tables of the tree fit kernel (any J <= 64), lane order = DFS pre-order
    float *tt_dt = nullptr, *tt_dd = nullptr;
    int *tt_tab = nullptr, *tt_anc = nullptr;
    int tt_prior_dims = -1;                                  // what the prior columns of tt_tab currently describe (guarded by mu)
    std::vector<int> tt_lane_of;                             // lane of every joint
    int groups_a = 0;                                        // GA = ceil(J / 8)
    k2b::k2b_half* wsA2 = nullptr;                           // per-frame A operand of the tile kernel
    float* dump = nullptr;                                   // 64 x 3 floats: store target of lanes outside the batch
    int k_steps_x = 0;
    // LBS per-frame operand workspace (grow-only)
    k2b::k2b_half *wsXh = nullptr, *wsXl = nullptr;
    int ws_bpad = 0;
    // Adam coefficient tables, one per (iters, lr, b1, b2); at most kMaxAdamTables, least recently used evicted
    struct AdamTable { float2* dev; uint64_t last_use; };
    std::map<std::tuple<int, double, double, double>, AdamTable> adam_tables;
    uint64_t adam_clock = 0;
    std::mutex mu;
};

struct k2b_prior {
    int M = 0, D = 0;
    float *pa_image = nullptr, *row_const = nullptr, *nlw = nullptr;
    k2b::k2b_half* frag32 = nullptr;
    float inv_scale[k2b::kPriorMaxGauss] = {};
    // host copies (symmetrised precisions in double, means, nll weights) and the mixture folded to its first Dv
    // dimensions for the tree fit kernel, built on first use per Dv
    std::vector<double> Ps, mu;
    std::vector<float> nllw;
    struct Folded { float *pA = nullptr, *ph = nullptr, *pb = nullptr, *pmu = nullptr, *pcl = nullptr; };
    std::map<int, Folded> folded;
    std::mutex mu_lock;
};

extern "C" {

uint32_t k2b_version(void) { return (1u << 16) | 0u; }
const char* k2b_last_error(void) { return g_err.c_str(); }

uint32_t k2b_fit_config_size(void) { return (uint32_t)sizeof(k2b_fit_config); }

void k2b_fit_config_default(k2b_fit_config* c) {
    if (!c) return;
    c->num_iters = 30;           // FrameOptimizeConfig.num_iters_first (core/config.py:32)
    c->step_size = 1e-2;
    c->adam_beta1 = 0.9;
    c->adam_beta2 = 0.999;
    c->adam_eps = 1e-8;
    c->sigma = 100.0f;
    c->joint_loss_weight = 600.0f;
    c->pose_prior_weight = (float)(4.78 * 1.5);
    c->angle_prior_weight = 15.2f;
    c->shape_prior_weight = 5.0f;
    c->pose_preserve_weight = 0.0f;
    c->freeze_betas = 0;
    c->conf_per_frame = 0;
    const int idx[4] = {52, 55, 9, 12};
    const float sg[4] = {1.f, -1.f, -1.f, -1.f};
    for (int i = 0; i < 4; ++i) { c->angle_prior_index[i] = idx[i]; c->angle_prior_sign[i] = sg[i]; }
    c->optimize_mask = 15;
    c->transl_prior_weight = 0.0f;
    c->debug_launch_shape = 0;
    c->prior_pose_dims = 0;
    c->num_betas_prior = 0;
}

int k2b_model_create(k2b_model** out, int32_t V, int32_t J, int32_t NB, int32_t E, const float* v_template,
                     const float* shapedirs, const float* posedirs, const float* j_regressor,
                     const float* lbs_weights, const int32_t* parents, const int32_t* extra_vertex_ids) {
    if (!out) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: out is NULL");
    *out = nullptr;
    if (V <= 0 || J < 2 || J > k2b::kMaxJoints || NB < 1 || NB > k2b::kMaxShape || E < 0)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: bad sizes V=%d J=%d NB=%d E=%d (need 2<=J<=64, 1<=NB<=32)", V, J, NB, E);
    if (!v_template || !shapedirs || !posedirs || !j_regressor || !lbs_weights || !parents || (E > 0 && !extra_vertex_ids))
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: NULL constant array");
    if (parents[0] >= 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: parents[0] must be -1 (root)");
    for (int j = 1; j < J; ++j)
        if (parents[j] < 0 || parents[j] >= j)
            return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: parents[%d]=%d must be in [0,%d)", j, parents[j], j);
    for (int e = 0; e < E; ++e)
        if (extra_vertex_ids[e] < 0 || extra_vertex_ids[e] >= V)
            return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_create: extra_vertex_ids[%d]=%d out of range", e, extra_vertex_ids[e]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(K2B_ERR_NO_DEVICE, "k2b_model_create: no HIP device visible (this engine has no CPU path)");

    k2b_model* m = new k2b_model;
    // released (device buffers included) on every early return below; handed to the caller at the end
    struct Guard {
        k2b_model* m;
        ~Guard() { if (m) k2b_model_destroy(m); }
    } guard{m};
    m->V = V; m->J = J; m->NB = NB; m->E = E; m->P = 9 * (J - 1);
    HIP_TRY(upload(&m->v_template, v_template, (size_t)V * 3));
    HIP_TRY(upload(&m->shapedirs, shapedirs, (size_t)V * 3 * NB));
    HIP_TRY(upload(&m->posedirs, posedirs, (size_t)m->P * 3 * V));
    HIP_TRY(upload(&m->j_regressor, j_regressor, (size_t)J * V));
    HIP_TRY(upload(&m->lbs_weights, lbs_weights, (size_t)V * J));
    HIP_TRY(upload(&m->parents, parents, (size_t)J));
    HIP_TRY(upload(&m->extra_ids, extra_vertex_ids, (size_t)E));

    // LBS operands: B side of the two GEMMs, f16 hi/lo in fragment order (k2b_lbs.hip)
    {
        const int P = m->P;
        int KX = ((P + NB + 2 + 31) / 32) * 2;         // even: the kernels stage 32-deep slices
        m->groups_a = k2b::tile_groups_a(J);
        // 49-56 joints with fewer features than SMPL-X (SMPL-H: 477 -> 15 k-steps): one all-zero k-step more buys the stream kernel
        if (K2B_LBS_STREAM && m->groups_a == 7 && KX < 2 * k2b::kStreamXKSteps) KX = 2 * k2b::kStreamXKSteps;
        m->k_steps_x = KX;
        m->stream = K2B_LBS_STREAM && m->groups_a == 3 && KX == 2 * k2b::kStreamKSteps;
        m->stream_x = K2B_LBS_STREAM && m->groups_a == 7 && KX == 2 * k2b::kStreamXKSteps;
        HIP_TRY(hipMalloc((void**)&m->dump, 64 * 1024));     // 64 x 3 floats used; the rest is room for diagnostic builds
        // tag[i] = 1 + e when vertex i of the set is the vertex of output joint J + e (mesh set only), else 0
        auto build = [&](k2b_model::VertexSet& vs, const std::vector<int>& ids, const std::vector<int>& tag) -> int {
            const int n = (int)ids.size();
            vs.num = n;
            vs.v_tiles = (n + 31) / 32;
            const int vp = vs.v_tiles * 32;
            std::vector<k2b::k2b_half> pdh((size_t)KX * 3 * vp * 16, (k2b::k2b_half)0.f), pdl(pdh.size(), (k2b::k2b_half)0.f);
            for (int i = 0; i < n; ++i) {
                const int v = ids[i];
                for (int c = 0; c < 3; ++c) {
                    auto put = [&](int k, k2b::k2b_half hi, k2b::k2b_half lo) {
                        const size_t o = k2b::frag_elem(((size_t)(k >> 4) * 3 + c) * vs.v_tiles + (i >> 5), k, i);
                        pdh[o] = hi;
                        pdl[o] = lo;
                    };
                    auto split = [&](int k, float x) -> float {   // returns what two f16 terms leave over
                        const float xs = x * k2b::kPdScale;
                        const k2b::k2b_half hi = (k2b::k2b_half)xs;
                        const k2b::k2b_half lo = (k2b::k2b_half)(xs - (float)hi);
                        put(k, hi, lo);
                        return xs - (float)hi - (float)lo;
                    };
                    for (int k = 0; k < P; ++k) split(k, posedirs[(size_t)k * 3 * V + 3 * v + c]);
                    for (int k = 0; k < NB; ++k) split(P + k, shapedirs[((size_t)v * 3 + c) * NB + k]);
                    const float rest = split(P + NB, v_template[(size_t)v * 3 + c]);
                    // the template is metre-scale: keep its third term as an extra K row (feature = 1)
                    const k2b::k2b_half rh = (k2b::k2b_half)rest;
                    put(P + NB + 1, rh, (k2b::k2b_half)(rest - (float)rh));
                }
            }
            // tile-kernel layout of W: [16-vertex tile][hi groups | lo groups | ONES | ZERO][16 rows][8 joints]
            const int GA = k2b::tile_groups_a(J), NGP = k2b::tile_ngp(GA), v16 = vs.v_tiles * 2;
            std::vector<k2b::k2b_half> w2((size_t)v16 * NGP * 128, (k2b::k2b_half)0.f);
            for (int t = 0; t < v16; ++t)
                for (int r = 0; r < 16; ++r) {
                    const int i = t * 16 + r;
                    k2b::k2b_half* rowp = w2.data() + ((size_t)t * NGP * 16 + r) * 8;
                    if (i < n)
                        for (int j = 0; j < J; ++j) {
                            const float w = lbs_weights[(size_t)ids[i] * J + j];
                            const k2b::k2b_half hi = (k2b::k2b_half)w;
                            rowp[(size_t)(j >> 3) * 128 + (j & 7)] = hi;
                            rowp[(size_t)(GA + (j >> 3)) * 128 + (j & 7)] = (k2b::k2b_half)(w - (float)hi);
                        }
                    for (int k = 0; k < 3; ++k) rowp[(size_t)(2 * GA) * 128 + k] = (k2b::k2b_half)1.f;   // ONES: picks up the PAD terms
                    if (i < n && !tag.empty() && tag[i]) rowp[(size_t)(2 * GA + 1) * 128] = (k2b::k2b_half)(float)tag[i];   // ZERO group: joint tag
                }
            hipError_t e;
            if (m->stream || m->stream_x) {
                // stream kernels: Pd [k-step][16-vertex tile][coord][hi | lo] and W [16-vertex tile][3 or 5 fragments], 1 KiB pieces in
                // MFMA operand order (lane = row + 16 k-group, 8 halfs); vertex tiles padded to whole 128-vertex groups
                const int nv16 = (n + 127) / 128 * 8, SK = KX / 2, NWF = m->stream ? 3 : 5;
                vs.nv16 = nv16;
                std::vector<k2b::k2b_half> spd((size_t)SK * nv16 * 6 * 512, (k2b::k2b_half)0.f), sw((size_t)nv16 * NWF * 512, (k2b::k2b_half)0.f);
                for (int i = 0; i < n; ++i) {
                    const int v16 = i >> 4, r = i & 15;
                    for (int c = 0; c < 3; ++c)
                        for (int k = 0; k < KX * 16; ++k) {      // the split values already sit in pdh / pdl: same k, same scale
                            const size_t src = k2b::frag_elem(((size_t)(k >> 4) * 3 + c) * vs.v_tiles + (i >> 5), k, i);
                            const size_t dst = ((((size_t)(k >> 5) * nv16 + v16) * 3 + c) * 2) * 512 + (size_t)((((k >> 3) & 3) * 16 + r) * 8 + (k & 7));
                            spd[dst] = pdh[src];
                            spd[dst + 512] = pdl[src];
                        }
                    k2b::k2b_half* wt = sw.data() + (size_t)v16 * NWF * 512;
                    auto at = [&](int frag, int group, int k) -> k2b::k2b_half& { return wt[(size_t)frag * 512 + (size_t)((group * 16 + r) * 8 + k)]; };
                    for (int j = 0; j < J; ++j) {
                        const float w = lbs_weights[(size_t)ids[i] * J + j];
                        const k2b::k2b_half hi = (k2b::k2b_half)w, lo = (k2b::k2b_half)(w - (float)hi);
                        const int gj = j >> 3, kj = j & 7;
                        if (m->stream) { at(0, gj, kj) = hi; at(1, gj, kj) = hi; at(2, gj, kj) = lo; }
                        else if (gj < 4) { at(0, gj, kj) = hi; at(2, gj, kj) = lo; }
                        else { at(1, gj - 4, kj) = hi; at(4, gj - 4, kj) = hi; at(3, gj - 4, kj) = lo; }
                    }
                    // last group of the fragment that meets the PAD group of A: ONES; of the one that meets ZERO: the joint tag
                    for (int k = 0; k < 3; ++k) at(m->stream ? 0 : 1, 3, k) = (k2b::k2b_half)1.f;
                    if (!tag.empty() && tag[i]) at(m->stream ? 1 : 4, 3, 0) = (k2b::k2b_half)(float)tag[i];
                }
                if ((e = upload(&vs.spd, spd.data(), spd.size())) != hipSuccess) return (int)e;
                if ((e = upload(&vs.sw, sw.data(), sw.size())) != hipSuccess) return (int)e;
            }
            if (m->stream || m->stream_x) return 0;          // the tile kernel's images stay on the host (SMPL-X: 64 MB less per GPU)
            if ((e = upload(&vs.w2, w2.data(), w2.size())) != hipSuccess) return (int)e;
            if ((e = upload(&vs.pdh, pdh.data(), pdh.size())) != hipSuccess) return (int)e;
            if ((e = upload(&vs.pdl, pdl.data(), pdl.size())) != hipSuccess) return (int)e;
            return 0;
        };
        std::vector<int> all(V), ex(extra_vertex_ids, extra_vertex_ids + E), tag(V, 0);
        for (int v = 0; v < V; ++v) all[v] = v;
        // an output joint rides in the W image of its vertex (k2b_lbs.hip) - unless two joints share a vertex or the index
        // does not fit an f16 integer, in which case the gather launch stays
        m->joints_in_mesh = E > 0 && E <= 1024;
        for (int e = 0; e < E && m->joints_in_mesh; ++e) {
            if (tag[extra_vertex_ids[e]]) m->joints_in_mesh = false;
            tag[extra_vertex_ids[e]] = e + 1;
        }
        if (!m->joints_in_mesh) std::fill(tag.begin(), tag.end(), 0);
        if (build(m->mesh, all, tag) != 0 || (E > 0 && build(m->extra, ex, std::vector<int>()) != 0))
            return fail(K2B_ERR_HIP, "k2b_model_create: uploading LBS operands failed");
    }

    // J x V contraction on the matrix cores
    {
        const int splits = 32;
        const int Jp = (J + 15) / 16 * 16, Np = (3 * NB + 15) / 16 * 16;
        float* ws = nullptr;
        HIP_TRY(hipMalloc((void**)&ws, (size_t)splits * Jp * Np * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&m->j_template, (size_t)J * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&m->j_dirs, (size_t)J * 3 * NB * sizeof(float)));
        HIP_TRY(k2b::launch_jreg_contract(m->j_regressor, m->v_template, m->j_template, J, V, 3, ws, splits, nullptr));
        HIP_TRY(k2b::launch_jreg_contract(m->j_regressor, m->shapedirs, m->j_dirs, J, V, 3 * NB, ws, splits, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        m->h_j_template.resize((size_t)J * 3);
        m->h_j_dirs.resize((size_t)J * 3 * NB);
        HIP_TRY(hipMemcpy(m->h_j_template.data(), m->j_template, m->h_j_template.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(m->h_j_dirs.data(), m->j_dirs, m->h_j_dirs.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(ws));
        // the same numbers lane-major for the pose set-up kernel (lane = joint): one of its loads touches 1-2 cache lines instead
        // of one per joint
        std::vector<float> lane((size_t)3 * (1 + NB) * 64, 0.f);
        for (int j = 0; j < J; ++j)
            for (int c = 0; c < 3; ++c) {
                lane[((size_t)c * (1 + NB)) * 64 + j] = m->h_j_template[j * 3 + c];
                for (int k = 0; k < NB; ++k) lane[((size_t)c * (1 + NB) + 1 + k) * 64 + j] = m->h_j_dirs[((size_t)j * 3 + c) * NB + k];
            }
        HIP_TRY(hipMalloc((void**)&m->j_basis_lane, lane.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(m->j_basis_lane, lane.data(), lane.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    // tables of the fused fit kernel: lanes follow the DFS pre-order of the tree, so that every
    // subtree is a contiguous lane range
    bool ok = (J == k2b::kFitJoints) && NB <= k2b::kMaxBetas;
    if (!ok) m->fit_why = "the 24-lane fused fit kernel is built for the 24-joint SMPL tree with <= 16 betas";
    std::vector<std::vector<int>> children(J);
    std::vector<int> depth(J, 0);
    int maxd = 0;
    for (int j = 1; j < J; ++j) {
        children[parents[j]].push_back(j);
        depth[j] = depth[parents[j]] + 1;
        maxd = depth[j] > maxd ? depth[j] : maxd;
    }
    std::vector<int> order, lane_of(J, -1), size(J, 1);
    {
        std::vector<int> stack{0};
        while (!stack.empty()) {
            const int j = stack.back();
            stack.pop_back();
            lane_of[j] = (int)order.size();
            order.push_back(j);
            for (auto it = children[j].rbegin(); it != children[j].rend(); ++it) stack.push_back(*it);
        }
        for (int j = J - 1; j >= 1; --j) size[parents[j]] += size[j];
    }
    int rounds = 0;
    while ((1 << rounds) < maxd + 1) ++rounds;
    if (ok && (rounds > k2b::kMaxRounds || J > 32)) { ok = false; m->fit_why = "tree too deep / large for the fused fit kernel"; }
    m->depth = depth;
    std::vector<int> tab((size_t)64 * k2b::kLaneTabStride, -1);
    std::vector<float> dt((size_t)64 * 3, 0.f), dd((size_t)64 * 3 * k2b::kMaxBetas, 0.f);
    if (ok) {
        for (int l = 0; l < J; ++l) {
            const int j = order[l], p = parents[j];
            int* t = tab.data() + (size_t)l * k2b::kLaneTabStride;
            t[0] = j;
            t[1] = p >= 0 ? lane_of[p] : -1;
            // ancestor lane 2^r levels up (pointer doubling), -1 once past the root
            int anc = t[1];
            for (int r = 0; r < k2b::kMaxRounds; ++r) {
                t[2 + r] = anc;
                for (int s = 0; s < (1 << r) && anc >= 0; ++s) {   // advance 2^r more levels
                    const int aj = order[anc];
                    anc = parents[aj] >= 0 ? lane_of[parents[aj]] : -1;
                }
            }
            t[2 + k2b::kMaxRounds] = size[j];        // subtree = lanes [l, l + size)
            t[3 + k2b::kMaxRounds] = depth[j];
            for (int c = 0; c < 3; ++c) {
                dt[l * 3 + c] = m->h_j_template[j * 3 + c] - (p >= 0 ? m->h_j_template[p * 3 + c] : 0.f);
                for (int k = 0; k < NB; ++k)
                    dd[(l * 3 + c) * k2b::kMaxBetas + k] =
                        m->h_j_dirs[(j * 3 + c) * NB + k] - (p >= 0 ? m->h_j_dirs[(p * 3 + c) * NB + k] : 0.f);
            }
        }
    }
    HIP_TRY(upload(&m->dt, dt.data(), dt.size()));
    HIP_TRY(upload(&m->dd, dd.data(), dd.size()));
    HIP_TRY(upload(&m->tree, tab.data(), tab.size()));
    m->fit_ok = ok;
    {   // tree fit kernel: [64][8] lane table (prior columns filled per call), rest offsets and their shape directions
        std::vector<int> tt((size_t)64 * 8, -1);
        std::vector<float> tdt((size_t)64 * 3, 0.f), tdd((size_t)64 * 3 * k2b::kMaxShape, 0.f);
        for (int l = 0; l < 64; ++l) { tt[l * 8 + 1] = 0; tt[l * 8 + 2] = 1; tt[l * 8 + 3] = 1000; }
        for (int l = 0; l < J; ++l) {
            const int j = order[l], p = parents[j];
            tt[l * 8 + 0] = j; tt[l * 8 + 1] = p >= 0 ? lane_of[p] : 0; tt[l * 8 + 2] = size[j]; tt[l * 8 + 3] = depth[j];
            for (int c = 0; c < 3; ++c) {
                tdt[l * 3 + c] = m->h_j_template[j * 3 + c] - (p >= 0 ? m->h_j_template[p * 3 + c] : 0.f);
                for (int k = 0; k < NB; ++k)
                    tdd[(l * 3 + c) * k2b::kMaxShape + k] =
                        m->h_j_dirs[(j * 3 + c) * NB + k] - (p >= 0 ? m->h_j_dirs[(p * 3 + c) * NB + k] : 0.f);
            }
        }
        // ancestors 1, 2, 4, 8 levels up (pointer doubling); 63 = "none": a lane that is no joint and holds the identity
        std::vector<int> tanc((size_t)64 * 4, 63);
        if (J <= 63)
            for (int l = 0; l < J; ++l) {
                int aj = order[l];
                for (int r = 0, dist = 0; r < 4; ++r) {
                    for (; dist < (1 << r) && aj >= 0; ++dist) aj = parents[aj];
                    tanc[l * 4 + r] = aj >= 0 ? lane_of[aj] : 63;
                }
            }
        m->tt_lane_of = lane_of;
        HIP_TRY(upload(&m->tt_anc, tanc.data(), tanc.size()));
        HIP_TRY(upload(&m->tt_dt, tdt.data(), tdt.size()));
        HIP_TRY(upload(&m->tt_dd, tdd.data(), tdd.size()));
        HIP_TRY(upload(&m->tt_tab, tt.data(), tt.size()));
    }
    guard.m = nullptr;
    *out = m;
    return K2B_OK;
}

void k2b_model_destroy(k2b_model* m) {
    if (!m) return;
    (void)hipDeviceSynchronize();
    float* fl[] = {m->v_template, m->shapedirs, m->posedirs, m->j_regressor, m->lbs_weights, m->j_template,
                   m->j_dirs, m->dt, m->dd, m->j_basis_lane};
    for (float* p : fl) if (p) (void)hipFree(p);
    k2b::k2b_half* hl[] = {m->mesh.pdh, m->mesh.pdl, m->extra.pdh, m->extra.pdl, m->wsXh, m->wsXl, m->mesh.w2, m->extra.w2, m->wsA2,
                           m->mesh.spd, m->mesh.sw, m->extra.spd, m->extra.sw};
    if (m->dump) (void)hipFree(m->dump);
    if (m->tt_dt) (void)hipFree(m->tt_dt);
    if (m->tt_dd) (void)hipFree(m->tt_dd);
    if (m->tt_tab) (void)hipFree(m->tt_tab);
    if (m->tt_anc) (void)hipFree(m->tt_anc);
    for (k2b::k2b_half* p : hl) if (p) (void)hipFree(p);
    if (m->parents) (void)hipFree(m->parents);
    if (m->extra_ids) (void)hipFree(m->extra_ids);
    if (m->tree) (void)hipFree(m->tree);
    for (auto& kv : m->adam_tables) (void)hipFree(kv.second.dev);
    delete m;
}

int k2b_debug_read_dump(const k2b_model* m, void* host, int64_t nbytes) {
    if (!m || !host || nbytes < 0 || nbytes > 64 * 1024) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_debug_read_dump: bad arguments");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host, m->dump, (size_t)nbytes, hipMemcpyDeviceToHost));
    return K2B_OK;
}

int k2b_model_dims(const k2b_model* m, int32_t* V, int32_t* J, int32_t* NB, int32_t* E) {
    if (!m) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_dims: model is NULL");
    if (V) *V = m->V;
    if (J) *J = m->J;
    if (NB) *NB = m->NB;
    if (E) *E = m->E;
    return K2B_OK;
}

int k2b_model_joint_basis(const k2b_model* m, float* j_template, float* j_dirs) {
    if (!m) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_joint_basis: model is NULL");
    if (j_template) memcpy(j_template, m->h_j_template.data(), m->h_j_template.size() * sizeof(float));
    if (j_dirs) memcpy(j_dirs, m->h_j_dirs.data(), m->h_j_dirs.size() * sizeof(float));
    return K2B_OK;
}

int k2b_prior_create(k2b_prior** out, int32_t M, int32_t D, const float* means, const float* precisions,
                     const float* nll_weights) {
    if (!out) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_prior_create: out is NULL");
    *out = nullptr;
    if (!means || !precisions || !nll_weights) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_prior_create: NULL array");
    if (D != k2b::kPriorDim) return fail(K2B_ERR_UNSUPPORTED, "k2b_prior_create: dim=%d, the fit kernel is built for 69-D body poses", D);
    if (M < 1 || M > k2b::kPriorMaxGauss)
        return fail(K2B_ERR_UNSUPPORTED, "k2b_prior_create: num_gaussians=%d, supported 1..%d", M, k2b::kPriorMaxGauss);
    for (int m = 0; m < M; ++m)
        if (!(nll_weights[m] >= 0.f)) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_prior_create: nll_weights[%d] must be >= 0", m);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(K2B_ERR_NO_DEVICE, "k2b_prior_create: no HIP device visible (this engine has no CPU path)");

    constexpr int MG = k2b::kPriorMaxGauss;
    // symmetrised precisions and c_m = P_m mu_m, in double
    std::vector<double> Ps((size_t)M * D * D), c((size_t)M * D, 0.0);
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j)
                Ps[((size_t)m * D + i) * D + j] =
                    0.5 * ((double)precisions[((size_t)m * D + i) * D + j] + (double)precisions[((size_t)m * D + j) * D + i]);
    for (int m = 0; m < M; ++m)
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
            for (int j = 0; j < D; ++j) s += Ps[((size_t)m * D + i) * D + j] * (double)means[m * D + j];
            c[(size_t)m * D + i] = s;
        }
    auto P = [&](int m, int i, int j) -> float { return (float)Ps[((size_t)m * D + i) * D + j]; };

    // LDS image: rim rows 64..68 over the core columns, then mu | c of the core rows (k2b_internal.h)
    constexpr int NC = 64, NR = 5;
    std::vector<float> pa((size_t)k2b::kPriorImageFloats, 0.f);
    float* cmu = pa.data() + 2 * NR * 64 * 4;
    std::vector<float> rc((size_t)8 * 64, 0.f), nlw(MG, 0.f);
    for (int m = 0; m < M; ++m) {
        for (int cc = 0; cc < NR; ++cc)
            for (int col = 0; col < NC; ++col)
                pa[((size_t)m * NR + cc) * NC + col] = P(m, NC + cc, col);       // [m][rim row][column]: lane = column, conflict-free 4-byte reads
        for (int r = 0; r < NC; ++r) {
            cmu[(size_t)m * 2 * NC + r] = means[m * D + r];
            cmu[(size_t)m * 2 * NC + NC + r] = (float)c[(size_t)m * D + r];
        }
        // per-lane constants of the rim rows: lane 8m + s, s < 5 <-> row 64 + s of component m
        for (int s = 0; s < NR; ++s) {
            const int l = 8 * m + s;
            for (int k = 0; k < NR; ++k) rc[(size_t)k * 64 + l] = P(m, NC + s, NC + k);
            rc[(size_t)5 * 64 + l] = (float)c[(size_t)m * D + NC + s];
            double kb = 0.0;
            for (int j = 0; j < NC; ++j) kb += Ps[((size_t)m * D + NC + s) * D + j] * (double)means[m * D + j];
            rc[(size_t)6 * 64 + l] = (float)kb;
            rc[(size_t)7 * 64 + l] = means[m * D + NC + s];
        }
        nlw[m] = -logf(nll_weights[m]);      // a weight that underflowed to 0 gives +inf, as torch.log does in the reference: never the arg-min
    }
    // the 64 x 64 core as MFMA A fragments (v_mfma_f32_16x16x32_f16: lane l holds row l & 15,
    // k = 8 (l >> 4) + j), two f16 terms per entry, scaled by a power of two per component so that the
    // largest entry sits near 2^13 (hi never overflows, lo stays normal for every entry that matters)
    std::vector<k2b::k2b_half> f32((size_t)k2b::kPriorFrag32Halfs, (k2b::k2b_half)0.f);
    k2b_prior* p = new k2b_prior;
    p->M = M; p->D = D;
    p->Ps = Ps;
    p->mu.assign(means, means + (size_t)M * D);
    p->nllw.assign(nll_weights, nll_weights + M);
    for (int m = 0; m < MG; ++m) p->inv_scale[m] = 1.0f;
    for (int m = 0; m < M; ++m) {
        double maxabs = 0.0;
        for (int row = 0; row < D; ++row)          // core rows and the rim rows 64..68 (their fragments share the scale)
            for (int col = 0; col < NC; ++col) maxabs = std::max(maxabs, std::fabs(Ps[((size_t)m * D + row) * D + col]));
        int e = 0;
        if (maxabs > 0.0 && std::isfinite(maxabs)) e = 13 - (int)std::ceil(std::log2(maxabs));
        e = std::max(-100, std::min(100, e));
        const double scale = std::ldexp(1.0, e);
        p->inv_scale[m] = (float)std::ldexp(1.0, -e);
        for (int t = 0; t < 4; ++t)
            for (int l = 0; l < 64; ++l) {
                const int row = 16 * t + (l & 15), g = l >> 4;
                for (int ks = 0; ks < 2; ++ks)
                    for (int j = 0; j < 8; ++j) {
                        const float x = (float)(Ps[((size_t)m * D + row) * D + 32 * ks + 8 * g + j] * scale);
                        const k2b::k2b_half hi = (k2b::k2b_half)x;
                        f32[((((size_t)m * 4 + t) * 4 + ks) * 64 + l) * 8 + j] = hi;
                        f32[((((size_t)m * 4 + t) * 4 + 2 + ks) * 64 + l) * 8 + j] = (k2b::k2b_half)(x - (float)hi);
                    }
            }
        // rim rows 64..68 over the core columns as a FIFTH row tile (rows 69..79 are zero), kept compact in the LDS image:
        // fragment f = 2 ks + (hi | lo), entry kg * 5 + row = the eight halfs of MFMA lane (row, k-group kg); entry 20 = zeros,
        // read by the lanes of the tile's empty rows (k2b_fit.hip, comp_issue)
        k2b::k2b_half* rf = reinterpret_cast<k2b::k2b_half*>(pa.data() + 2 * NR * 64 * 4 + MG * 2 * NC) + (size_t)m * k2b::kPriorRimFragEntries * 8;
        for (int ks = 0; ks < 2; ++ks)
            for (int g = 0; g < 4; ++g)
                for (int row = 0; row < NR; ++row)
                    for (int j = 0; j < 8; ++j) {
                        const float x = (float)(Ps[((size_t)m * D + NC + row) * D + 32 * ks + 8 * g + j] * scale);
                        const k2b::k2b_half hi = (k2b::k2b_half)x;
                        rf[((size_t)(2 * ks) * 21 + g * 5 + row) * 8 + j] = hi;
                        rf[((size_t)(2 * ks + 1) * 21 + g * 5 + row) * 8 + j] = (k2b::k2b_half)(x - (float)hi);
                    }
    }
    // upload; on any failure the partially built handle is released before the error is returned
    auto upload_all = [&]() -> hipError_t {
        hipError_t e;
        if ((e = upload(&p->pa_image, pa.data(), pa.size())) != hipSuccess) return e;
        if ((e = upload(&p->row_const, rc.data(), rc.size())) != hipSuccess) return e;
        if ((e = upload(&p->nlw, nlw.data(), nlw.size())) != hipSuccess) return e;
        if ((e = hipMalloc((void**)&p->frag32, f32.size() * sizeof(k2b::k2b_half))) != hipSuccess) return e;
        return hipMemcpy(p->frag32, f32.data(), f32.size() * sizeof(k2b::k2b_half), hipMemcpyHostToDevice);
    };
    if (const hipError_t e = upload_all(); e != hipSuccess) {
        k2b_prior_destroy(p);
        return fail(K2B_ERR_HIP, "k2b_prior_create: upload failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return K2B_OK;
}

void k2b_prior_destroy(k2b_prior* p) {
    if (!p) return;
    (void)hipDeviceSynchronize();
    float* fl[] = {p->pa_image, p->row_const, p->nlw};
    for (float* q : fl) if (q) (void)hipFree(q);
    if (p->frag32) (void)hipFree(p->frag32);
    for (auto& kv : p->folded) {
        float* fl2[] = {kv.second.pA, kv.second.ph, kv.second.pb, kv.second.pmu, kv.second.pcl};
        for (float* q : fl2) if (q) (void)hipFree(q);
    }
    delete p;
}

namespace {
// Adam bias terms in double, exactly as torch/optim/adam.py computes them in Python floats; cached per model
int adam_table(k2b_model* model, const k2b_fit_config* cfg, hipStream_t stream, float2** out) {
    std::lock_guard<std::mutex> lk(model->mu);
    const auto key = std::make_tuple((int)cfg->num_iters, cfg->step_size, cfg->adam_beta1, cfg->adam_beta2);
    auto it = model->adam_tables.find(key);
    if (it != model->adam_tables.end()) {
        it->second.last_use = ++model->adam_clock;
        *out = it->second.dev;
        return K2B_OK;
    }
    std::vector<float2> h(cfg->num_iters);
    const double lr = cfg->step_size, b1 = cfg->adam_beta1, b2 = cfg->adam_beta2;
    for (int t = 1; t <= cfg->num_iters; ++t) {
        const double bc1 = 1.0 - std::pow(b1, (double)t), bc2 = 1.0 - std::pow(b2, (double)t);
        h[t - 1] = make_float2((float)(lr / bc1), (float)std::sqrt(bc2));
    }
    constexpr size_t kMaxAdamTables = 64;
    if (model->adam_tables.size() >= kMaxAdamTables) {       // evict the least recently used table
        auto victim = model->adam_tables.begin();
        for (auto jt = model->adam_tables.begin(); jt != model->adam_tables.end(); ++jt)
            if (jt->second.last_use < victim->second.last_use) victim = jt;
        HIP_TRY(hipStreamSynchronize(stream));               // a launch in flight may still read it
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(victim->second.dev));
        model->adam_tables.erase(victim);
    }
    float2* coef = nullptr;
    HIP_TRY(upload(&coef, h.data(), h.size()));
    model->adam_tables[key] = {coef, ++model->adam_clock};
    *out = coef;
    return K2B_OK;
}

// the mixture restricted to its first Dv dimensions, the others fixed at 0 (k2b_fit_tree.hip)
int folded_prior(k2b_prior* p, int Dv, k2b_prior::Folded* out) {
    std::lock_guard<std::mutex> lk(p->mu_lock);
    auto it = p->folded.find(Dv);
    if (it != p->folded.end()) { *out = it->second; return K2B_OK; }
    const int M = p->M, D = p->D;
    constexpr int MG = k2b::kPriorMaxGauss;
    std::vector<float> A((size_t)MG * 16 * 64 * 4, 0.f), h((size_t)MG * 64, 0.f), b((size_t)MG * 64, 0.f), mu((size_t)MG * 64, 0.f),
        cl(MG, 3.0e38f);                                     // components beyond M: never the arg-min
    for (int m = 0; m < M; ++m) {
        auto P = [&](int i, int j) { return p->Ps[((size_t)m * D + i) * D + j]; };
        double c = 0.0;
        for (int k = Dv; k < D; ++k)
            for (int l = Dv; l < D; ++l) c += p->mu[(size_t)m * D + k] * P(k, l) * p->mu[(size_t)m * D + l];   // d_c = -mu_c
        for (int i = 0; i < Dv; ++i) {
            double bi = 0.0, Amu = 0.0;
            for (int k = Dv; k < D; ++k) bi -= P(i, k) * p->mu[(size_t)m * D + k];
            for (int j = 0; j < Dv; ++j) {
                A[(((size_t)m * 16 + (j >> 2)) * 64 + i) * 4 + (j & 3)] = (float)P(i, j);
                Amu += P(i, j) * p->mu[(size_t)m * D + j];
            }
            b[(size_t)m * 64 + i] = (float)bi;
            h[(size_t)m * 64 + i] = (float)(bi - Amu);
            mu[(size_t)m * 64 + i] = (float)p->mu[(size_t)m * D + i];
        }
        cl[m] = (float)(0.5 * c) - logf(p->nllw[m]);         // a weight that underflowed to 0 gives +inf: never the arg-min
    }
    k2b_prior::Folded f;
    HIP_TRY(upload(&f.pA, A.data(), A.size()));
    HIP_TRY(upload(&f.ph, h.data(), h.size()));
    HIP_TRY(upload(&f.pb, b.data(), b.size()));
    HIP_TRY(upload(&f.pmu, mu.data(), mu.size()));
    HIP_TRY(upload(&f.pcl, cl.data(), cl.size()));
    p->folded[Dv] = f;
    *out = f;
    return K2B_OK;
}

extern "C++" {
template <class Args, class Launch>
int fit_world_vertex_joints(k2b_model* model, const k2b_fit_config* cfg, Args a, const float* tr_prior_src, int frozen_shape,
                            const std::vector<int>& vsel, const std::vector<int>& vcol, hipStream_t stream, Launch launch_eval);
}

// large trees (SMPL-H / SMPL-X), or a prior over a prefix of the body pose: k2b_fit_tree.hip
int fit_tree(k2b_model* model, k2b_prior* prior, const k2b_fit_config* cfg, int prior_dims, int32_t B, int32_t K,
             const int32_t* model_joint_index, const float* j3d, const float* conf, const float* go_in, const float* bp_in,
             const float* be_in, const float* tr_in, const float* preserve, const float* tr_prior, float* go_out, float* bp_out,
             float* be_out, float* tr_out, float* loss_out, float* grad_out, void* stream, int chain_len = 1, int chain_iters = 0) {
    const int J = model->J, NB = model->NB;
    if (prior_dims < 3 || prior_dims > 64 || prior_dims % 3 != 0 || prior_dims > prior->D || prior_dims > 3 * (J - 1))
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: the tree kernel takes a prior over the first 3..63 body-pose dimensions "
                    "(a multiple of 3, at most the mixture's %d), got %d", prior->D, prior_dims);
    if (cfg->transl_prior_weight != 0.0f || tr_prior)
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: the translation prior (camera-space fitter) is not built for %d-joint models", J);
    if (B < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_frames=%d", B);
    if (K < 1 || K > J + model->E) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_targets=%d out of range", K);
    if (!model_joint_index) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: model_joint_index is NULL");
    if (cfg->num_iters < 1 || cfg->num_iters > (1 << 20)) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_iters=%d", cfg->num_iters);
    if (!(cfg->step_size >= 0.0) || !(cfg->adam_beta1 >= 0.0 && cfg->adam_beta1 < 1.0) || !(cfg->adam_beta2 >= 0.0 && cfg->adam_beta2 < 1.0))
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: bad Adam hyper-parameters");
    if (B == 0) return K2B_OK;
    if (!j3d || !go_in || !bp_in || !be_in || !tr_in || !go_out || !bp_out || !be_out || !tr_out)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: NULL parameter / target buffer (init transl is required, world_space.py:118-119)");
    k2b::FitTreeArgs a{};
    for (int l = 0; l < 64; ++l) a.lane_target[l] = -1;
    int maxd = 0, num_kinematic = 0;
    std::vector<int> vsel, vcol;             // vertex-selected joints: index into the model's extra joints, target column
    for (int k = 0; k < K; ++k) {
        const int j = model_joint_index[k];
        if (j < 0 || j >= J + model->E) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: model_joint_index[%d]=%d out of range", k, j);
        if (j >= J) {                        // vertex-selected joint: its term comes from k2b_vertex_term_kernel
            if ((int)vsel.size() >= 32) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: more than 32 vertex-selected joints among the targets");
            vsel.push_back(j - J);
            vcol.push_back(k);
            continue;
        }
        ++num_kinematic;
        const int l = model->tt_lane_of[j];
        if (a.lane_target[l] >= 0) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: joint %d is targeted twice", j);
        a.lane_target[l] = k;
        maxd = model->depth[j] > maxd ? model->depth[j] : maxd;
    }
    for (int i = 0; i < 4; ++i) {
        const int ai = cfg->angle_prior_index[i];
        if (ai < 0 || ai >= prior_dims) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: angle_prior_index[%d]=%d must be in [0,%d)", i, ai, prior_dims);
        a.angle_index[i] = ai;
        a.angle_sign[i] = cfg->angle_prior_sign[i];
    }
    k2b_prior::Folded f;
    if (const int rc = folded_prior(prior, prior_dims, &f); rc != K2B_OK) return rc;
    float2* coef = nullptr;
    {   // a chain's follow-up frames restart the optimiser: one table long enough for both counts, each reads its prefix
        k2b_fit_config tc = *cfg;
        if (chain_len > 1 && chain_iters > tc.num_iters) tc.num_iters = chain_iters;
        if (const int rc = adam_table(model, &tc, (hipStream_t)stream, &coef); rc != K2B_OK) return rc;
    }
    {   // prior columns of the lane table (depend on prior_dims): uploaded when the value changes.  The state lives in the
        // model (a per-thread cache keyed by the handle's address would go stale when a handle is destroyed and another one
        // created at the same address, or when two threads use one model with different values); launches of one model with
        // DIFFERENT prior_pose_dims must not be in flight on different streams at once.
        std::lock_guard<std::mutex> lk(model->mu);
        if (model->tt_prior_dims != prior_dims) {
            std::vector<int> cols((size_t)64 * 3, -1);
            for (int i = 0; i < prior_dims; ++i) {               // prior dimension i = component i % 3 of joint 1 + i / 3
                cols[i * 3 + 0] = model->tt_lane_of[1 + i / 3];
                cols[i * 3 + 1] = i % 3;
            }
            for (int j = 1; j < J; ++j)
                if (3 * (j - 1) + 2 < prior_dims) cols[model->tt_lane_of[j] * 3 + 2] = 3 * (j - 1);
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            for (int l = 0; l < 64; ++l)
                HIP_TRY(hipMemcpy(model->tt_tab + l * 8 + 4, cols.data() + l * 3, 3 * sizeof(int), hipMemcpyHostToDevice));
            model->tt_prior_dims = prior_dims;
        }
    }
    a.dt = model->tt_dt; a.dd = model->tt_dd; a.tab = model->tt_tab; a.anc = model->tt_anc;
    a.num_joints = J; a.num_shape = NB;
    a.num_rounds = 0;
    while ((1 << a.num_rounds) < maxd + 1) ++a.num_rounds;
    if (J > 63 || a.num_rounds > 4) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: the tree kernel takes up to 63 joints and depth 15");
    a.pfrag = prior->frag32; a.ph = f.ph; a.pb = f.pb; a.pmu = f.pmu; a.pcl = f.pcl;
    for (int m = 0; m < k2b::kPriorMaxGauss; ++m) a.inv_scale[m] = prior->inv_scale[m];
    a.num_gauss = prior->M; a.prior_dims = prior_dims;
    a.num_frames = B; a.num_targets = K;
    a.j3d = j3d; a.conf = conf; a.conf_per_frame = cfg->conf_per_frame ? 1 : 0;
    a.go_in = go_in; a.bp_in = bp_in; a.be_in = be_in; a.tr_in = tr_in; a.preserve = preserve;
    a.go_out = go_out; a.bp_out = bp_out; a.be_out = be_out; a.tr_out = tr_out; a.loss_out = loss_out; a.grad_out = grad_out;
    a.adam_coef = coef; a.num_iters = cfg->num_iters;
    a.one_minus_beta1 = (float)(1.0 - cfg->adam_beta1);
    a.beta2 = (float)cfg->adam_beta2; a.one_minus_beta2 = (float)(1.0 - cfg->adam_beta2);
    a.eps = (float)cfg->adam_eps;
    a.sigma = cfg->sigma; a.joint_w = cfg->joint_loss_weight; a.pose_prior_w = cfg->pose_prior_weight;
    a.angle_w = cfg->angle_prior_weight; a.shape_w = cfg->shape_prior_weight; a.preserve_w = cfg->pose_preserve_weight;
    a.freeze_betas = cfg->freeze_betas ? 1 : 0;
    a.num_betas_prior = cfg->num_betas_prior > 0 ? (cfg->num_betas_prior < NB ? cfg->num_betas_prior : NB) : NB;
    a.opt_mask = cfg->optimize_mask & 15;
    a.chain_len = chain_len > 1 ? chain_len : 1;
    a.chain_iters = chain_iters;
    if (cfg->debug_launch_shape < 0 || cfg->debug_launch_shape > 4)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: debug_launch_shape=%d must be 0..4", cfg->debug_launch_shape);
    a.debug_shape = cfg->debug_launch_shape <= 2 ? cfg->debug_launch_shape : 0;   // tree kernel: 1 = plain, 2 = component waves
    if (!vsel.empty()) {
        if (num_kinematic == 0)
            return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: at least one kinematic joint (model index < %d) must be among the targets", J);
        if (chain_len > 1)
            return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_sequence: vertex-selected joints are not built into the chain (fit frame by frame)");
        return fit_world_vertex_joints(model, cfg, a, nullptr, cfg->freeze_betas ? a.num_betas_prior : 0, vsel, vcol, (hipStream_t)stream,
                                       [&](const k2b::FitTreeArgs& e) { return k2b::launch_fit_tree(e, (hipStream_t)stream); });
    }
    HIP_TRY(k2b::launch_fit_tree(a, (hipStream_t)stream));
    return K2B_OK;
}
}  // namespace

namespace {
// Adam fit with vertex-selected joints among the targets (world_space.py:198-201 with indices >= J).  The fused kernel
// fits kinematic joints only, so every iteration is two launches queued back to back: the fused kernel in evaluate-only
// mode (kinematic targets + every prior -> loss, gradient) and the vertex-term kernel with its Adam tail (vertex targets,
// sum of the gradients, the optimiser step in place).  `a` is the fused launch fully set up for the caller's buffers.
extern "C++" {
template <class Args, class Launch>
int fit_world_vertex_joints(k2b_model* model, const k2b_fit_config* cfg, Args a, const float* tr_prior_src, int frozen_shape,
                            const std::vector<int>& vsel, const std::vector<int>& vcol, hipStream_t stream, Launch launch_eval) {
    const int B = a.num_frames, NB = model->NB, D = 3 * (model->J - 1), P = 3 + D + NB + 3;
    const int iters = cfg->num_iters;
    float2 *coef = nullptr, *coef_eval = nullptr;
    if (const int rc = adam_table(model, cfg, stream, &coef); rc != K2B_OK) return rc;
    {
        k2b_fit_config ec = *cfg;
        ec.num_iters = 1;
        ec.step_size = 0.0;
        if (const int rc = adam_table(model, &ec, stream, &coef_eval); rc != K2B_OK) return rc;
    }
    // stream-ordered scratch: gradient and loss of the evaluate launch, Adam state, copies of the preserve pose and the
    // translation prior's centre (their defaults are the INITIAL parameters, which the in-place steps overwrite)
    const size_t n_g = (size_t)B * P, n_all = 3 * n_g + B + (size_t)B * D + (size_t)B * 3;
    float* ws = nullptr;
    HIP_TRY(hipMallocAsync((void**)&ws, n_all * sizeof(float), stream));
    float *gbuf = ws, *mbuf = ws + n_g, *vbuf = ws + 2 * n_g, *lbuf = ws + 3 * n_g, *pres = lbuf + B, *trp = pres + (size_t)B * D;
    auto cleanup = [&](int rc) { (void)hipFreeAsync(ws, stream); return rc; };
#define K2B_TRY_WS(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); return cleanup(fail(K2B_ERR_HIP, "k2b_fit_world: HIP call failed in the vertex-joint path")); } } while (0)
    K2B_TRY_WS(hipMemsetAsync(mbuf, 0, 2 * n_g * sizeof(float), stream));
    K2B_TRY_WS(hipMemcpyAsync(pres, a.preserve ? a.preserve : a.bp_in, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    K2B_TRY_WS(hipMemcpyAsync(trp, tr_prior_src ? tr_prior_src : a.tr_in, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    const struct { const float* src; float* dst; size_t n; } cp[] = {
        {a.go_in, a.go_out, (size_t)B * 3}, {a.bp_in, a.bp_out, (size_t)B * D}, {a.be_in, a.be_out, (size_t)B * NB}, {a.tr_in, a.tr_out, (size_t)B * 3}};
    for (const auto& c : cp)
        if (c.src != c.dst) K2B_TRY_WS(hipMemcpyAsync(c.dst, c.src, c.n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    float* user_grad = a.grad_out;
    float* user_loss = a.loss_out;
    a.go_in = a.go_out; a.bp_in = a.bp_out; a.be_in = a.be_out; a.tr_in = a.tr_out;
    a.preserve = pres;
    if constexpr (std::is_same<Args, k2b::FitArgs>::value) a.tr_prior = trp;
    a.adam_coef = coef_eval; a.num_iters = 1;
    a.loss_out = lbuf; a.grad_out = gbuf;

    k2b::VertexTermArgs v{};
    v.v_template = model->v_template; v.shapedirs = model->shapedirs; v.posedirs = model->posedirs; v.lbs_weights = model->lbs_weights;
    v.j_template = model->j_template; v.j_dirs = model->j_dirs; v.parents = model->parents; v.extra_ids = model->extra_ids;
    v.num_vertices = model->V; v.num_betas = NB; v.num_joints = model->J;
    v.frozen_shape = frozen_shape;
    v.num_frames = B; v.num_sel = (int)vsel.size();
    for (size_t e = 0; e < vsel.size(); ++e) { v.sel[e] = vsel[e]; v.sel_k[e] = vcol[e]; }
    v.num_targets = a.num_targets; v.targets = a.j3d; v.conf = a.conf; v.conf_per_frame = a.conf_per_frame;
    v.sigma = a.sigma; v.joint_w = a.joint_w;
    v.go = a.go_out; v.bp = a.bp_out; v.be = a.be_out; v.tr = a.tr_out;
    v.go_w = a.go_out; v.bp_w = a.bp_out; v.be_w = a.be_out; v.tr_w = a.tr_out;
    float* loss_sink = user_loss ? user_loss : lbuf;
    v.loss_in = lbuf; v.grad_in = gbuf;
    v.adam_m = mbuf; v.adam_v = vbuf;
    v.one_minus_beta1 = a.one_minus_beta1; v.beta2 = a.beta2; v.one_minus_beta2 = a.one_minus_beta2; v.eps = a.eps;
    v.opt_mask = a.opt_mask;
    for (int it = 0; it < iters; ++it) {
        K2B_TRY_WS(launch_eval(a));
        v.adam_coef = coef + it;
        const bool last = it == iters - 1;
        v.loss_out = last ? loss_sink : lbuf;        // (lbuf: read and written by the same lane)
        v.grad_out = last ? user_grad : nullptr;
        K2B_TRY_WS(k2b::launch_vertex_term(v, stream));
    }
#undef K2B_TRY_WS
    return cleanup(K2B_OK);
}
}  // extern "C++"
}  // namespace

namespace {
int fit_world_impl(const k2b_model* model_c, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t B, int32_t K,
                   const int32_t* model_joint_index, const float* j3d, const float* conf, const float* go_in,
                   const float* bp_in, const float* be_in, const float* tr_in, const float* preserve,
                   const float* tr_prior, float* go_out,
                   float* bp_out, float* be_out, float* tr_out, float* loss_out, float* grad_out, void* stream,
                   int chain_len, int chain_iters, int lb_mode = 0, const k2b::LbfgsArgs* lb_host = nullptr, int lb_chain_max_iter = 0) {
    k2b_model* model = const_cast<k2b_model*>(model_c);
    if (!model || !prior || !cfg) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: model, prior and cfg are required");
    const int pose_dims_all = 3 * (model->J - 1);
    const int prior_dims = cfg->prior_pose_dims > 0 ? cfg->prior_pose_dims : (prior->D < pose_dims_all ? prior->D : pose_dims_all);
    const bool small_tree = model->fit_ok && prior->D == pose_dims_all && prior_dims == pose_dims_all &&
                            (cfg->num_betas_prior == 0 || cfg->num_betas_prior == model->NB);
    if (chain_len > 1 && (chain_iters < 1 || chain_iters > (1 << 20)))
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence: followup_iters=%d", chain_iters);
    if (chain_len > 1 && (preserve || tr_prior || grad_out || cfg->transl_prior_weight != 0.0f))
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_sequence: no explicit preserve pose, translation prior or gradient output in a chain");
    if (!small_tree)
        return fit_tree(model, const_cast<k2b_prior*>(prior), cfg, prior_dims, B, K, model_joint_index, j3d, conf, go_in, bp_in, be_in, tr_in,
                        preserve, tr_prior, go_out, bp_out, be_out, tr_out, loss_out, grad_out, stream, chain_len, chain_iters);
    if (B < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_frames=%d", B);
    if (K < 1 || K > model->J + model->E) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_targets=%d out of range", K);
    if (!model_joint_index) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: model_joint_index is NULL");
    if (cfg->num_iters < 1 || cfg->num_iters > (1 << 20)) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: num_iters=%d", cfg->num_iters);
    if (!(cfg->step_size >= 0.0) || !(cfg->adam_beta1 >= 0.0 && cfg->adam_beta1 < 1.0) || !(cfg->adam_beta2 >= 0.0 && cfg->adam_beta2 < 1.0))
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: bad Adam hyper-parameters");
    if (B == 0) return K2B_OK;
    if (!j3d || !go_in || !bp_in || !be_in || !tr_in || !go_out || !bp_out || !be_out || !tr_out)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: NULL parameter / target buffer (init transl is required, world_space.py:118-119)");

    k2b::FitArgs a{};
    int lane_target[k2b::kFitJoints];
    std::vector<int> vsel, vcol;             // vertex-selected joints: index into the model's extra joints, target column
    int num_kinematic = 0;
    for (int j = 0; j < k2b::kFitJoints; ++j) lane_target[j] = -1;
    for (int k = 0; k < K; ++k) {
        const int j = model_joint_index[k];
        if (j < 0 || j >= model->J + model->E) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: model_joint_index[%d]=%d out of range", k, j);
        if (j >= model->J) {                 // vertex-selected joint: its term comes from k2b_vertex_term_kernel
            if ((int)vsel.size() >= 32) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: more than 32 vertex-selected joints among the targets");
            vsel.push_back(j - model->J);
            vcol.push_back(k);
            continue;
        }
        if (lane_target[j] >= 0) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: joint %d is targeted twice", j);
        lane_target[j] = k;
        ++num_kinematic;
    }
    if (!vsel.empty() && num_kinematic == 0)
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: at least one kinematic joint (model index < %d) must be among the targets", model->J);
    if (!vsel.empty() && chain_len > 1)
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_sequence: vertex-selected joints are not built into the chain (fit frame by frame)");
    for (int i = 0; i < 4; ++i) {
        const int ai = cfg->angle_prior_index[i];
        if (ai < 0 || ai >= 64) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: angle_prior_index[%d]=%d must be in [0,64)", i, ai);
        a.angle_index[i] = ai;
        a.angle_sign[i] = cfg->angle_prior_sign[i];
    }

    float2* coef = nullptr;
    {   // a chain's follow-up frames restart the optimiser: one table long enough for both counts, each reads its prefix
        k2b_fit_config tc = *cfg;
        if (chain_len > 1 && chain_iters > tc.num_iters) tc.num_iters = chain_iters;
        if (const int rc = adam_table(model, &tc, (hipStream_t)stream, &coef); rc != K2B_OK) return rc;
    }

    a.dt = model->dt; a.dd = model->dd; a.lane_tab = model->tree;
    // global transforms are only needed down to the deepest targeted joint: 2^rounds > its depth
    {
        int maxd = 0;
        for (int k = 0; k < K; ++k)
            if (model_joint_index[k] < model->J) maxd = model->depth[model_joint_index[k]] > maxd ? model->depth[model_joint_index[k]] : maxd;
        int rounds = 0;
        while ((1 << rounds) < maxd + 1) ++rounds;
        a.num_rounds = rounds;
    }
    a.num_betas = model->NB;
    a.pa_image = prior->pa_image; a.row_const = prior->row_const; a.neg_log_nllw = prior->nlw;
    a.pa_frag32 = prior->frag32;
    for (int m = 0; m < k2b::kPriorMaxGauss; ++m) a.inv_scale[m] = prior->inv_scale[m];
    a.frames_per_wg = 0;
    a.num_gauss = prior->M;
    a.num_frames = B; a.num_targets = K;
    memcpy(a.lane_target, lane_target, sizeof lane_target);
    a.j3d = j3d; a.conf = conf; a.conf_per_frame = cfg->conf_per_frame ? 1 : 0;
    a.go_in = go_in; a.bp_in = bp_in; a.be_in = be_in; a.tr_in = tr_in; a.preserve = preserve;
    a.tr_prior = tr_prior ? tr_prior : tr_in;
    a.go_out = go_out; a.bp_out = bp_out; a.be_out = be_out; a.tr_out = tr_out;
    a.loss_out = loss_out; a.grad_out = grad_out;
    a.adam_coef = coef; a.num_iters = cfg->num_iters;
    // 1 - beta is formed in double (Python float) and only then rounded to the tensor dtype
    a.one_minus_beta1 = (float)(1.0 - cfg->adam_beta1);
    a.beta2 = (float)cfg->adam_beta2; a.one_minus_beta2 = (float)(1.0 - cfg->adam_beta2);
    a.eps = (float)cfg->adam_eps;
    a.sigma = cfg->sigma; a.joint_w = cfg->joint_loss_weight; a.pose_prior_w = cfg->pose_prior_weight;
    a.angle_w = cfg->angle_prior_weight; a.shape_w = cfg->shape_prior_weight; a.preserve_w = cfg->pose_preserve_weight;
    a.freeze_betas = cfg->freeze_betas ? 1 : 0;
    a.opt_mask = (cfg->optimize_mask & 15) & (cfg->freeze_betas ? ~4 : ~0);
    a.transl_prior_w = cfg->transl_prior_weight;
    if (cfg->debug_launch_shape < 0 || cfg->debug_launch_shape > 4)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world: debug_launch_shape=%d must be 0..4", cfg->debug_launch_shape);
    a.force_shape = cfg->debug_launch_shape;
    a.chain_len = chain_len > 1 ? chain_len : 1;
    a.chain_iters = chain_iters;
    a.num_cus = device_cus();
    a.lb_mode = lb_mode;
    if (lb_host) { a.lbv = *lb_host; a.lb_loss = lb_host->loss_in; a.lb_grad = lb_host->grad_in; a.lb_history = lb_host->H; }
    a.lb_chain_max_iter = lb_chain_max_iter;
    if (lb_mode != 0 && (!vsel.empty() || (chain_len > 1 && lb_mode != 3) || !lb_host))
        return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_world: the fused L-BFGS step needs kinematic targets and independent frames");
    if (!vsel.empty())
        return fit_world_vertex_joints(model, cfg, a, a.tr_prior, 0, vsel, vcol, (hipStream_t)stream,
                                       [&](const k2b::FitArgs& e) { return k2b::launch_fit_world(e, (hipStream_t)stream); });
    HIP_TRY(k2b::launch_fit_world(a, (hipStream_t)stream));
    return K2B_OK;
}
}  // namespace

int k2b_fit_world(const k2b_model* model, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t B, int32_t K,
                  const int32_t* model_joint_index, const float* j3d, const float* conf, const float* go_in,
                  const float* bp_in, const float* be_in, const float* tr_in, const float* preserve,
                  const float* tr_prior, float* go_out,
                  float* bp_out, float* be_out, float* tr_out, float* loss_out, float* grad_out, void* stream) {
    return fit_world_impl(model, prior, cfg, B, K, model_joint_index, j3d, conf, go_in, bp_in, be_in, tr_in, preserve, tr_prior,
                          go_out, bp_out, be_out, tr_out, loss_out, grad_out, stream, 1, 0);
}

int k2b_fit_sequence(const k2b_model* model, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t num_sequences,
                     int32_t frames_per_sequence, int32_t followup_iters, int32_t K, const int32_t* model_joint_index,
                     const float* j3d, const float* conf, const float* go_in, const float* bp_in, const float* be_in,
                     const float* tr_in, float* go_out, float* bp_out, float* be_out, float* tr_out, float* loss_out,
                     void* stream) {
    if (frames_per_sequence < 1) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence: frames_per_sequence=%d", frames_per_sequence);
    if ((int64_t)num_sequences * frames_per_sequence > (int64_t)1 << 30)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence: %d x %d frames", num_sequences, frames_per_sequence);
    if (frames_per_sequence == 1) {           // a chain of one: the first-frame fit (no preserve term)
        k2b_fit_config c1 = *cfg;
        c1.pose_preserve_weight = 0.0f;
        return fit_world_impl(model, prior, &c1, num_sequences, K, model_joint_index, j3d, conf, go_in, bp_in, be_in, tr_in, nullptr,
                              nullptr, go_out, bp_out, be_out, tr_out, loss_out, nullptr, stream, 1, 0);
    }
    return fit_world_impl(model, prior, cfg, num_sequences, K, model_joint_index, j3d, conf, go_in, bp_in, be_in, tr_in, nullptr, nullptr,
                          go_out, bp_out, be_out, tr_out, loss_out, nullptr, stream, frames_per_sequence, followup_iters);
}

namespace {
// One device-driven L-BFGS fit of B frames whose start is ALREADY in the parameter arrays (go, bp, be, tr: in place): state
// cleared, max_eval + 2 rounds of [closure, step], finalise, loss (+ gradient) at the result.  `ws` = lbfgs_ws_bytes() of
// stream-ordered scratch, `pres` / `trp` = preserve pose / translation prior centre (device, outside the parameter arrays).
struct LbfgsWs { unsigned char* base; size_t off_si, off_sv, n_state; float *gbuf, *lbuf, *gbuf2, *lbuf2; bool state_cleared; };   // (state_cleared: the caller did)
size_t lbfgs_ws_layout(int B, int P, int H, LbfgsWs* w) {
    w->n_state = (k2b::lbfgs_state_bytes(B, P, H, &w->off_si, &w->off_sv) + 15) / 16 * 16;
    // two closure-result buffers (the fused rounds alternate)
    const size_t n_res = (2 * ((size_t)B * P + B) * sizeof(float) + 15) / 16 * 16;
    return w->n_state + n_res;
}
// pointers into the workspace laid out above; returns the first byte behind it (the caller's own scratch)
float* lbfgs_ws_assign(LbfgsWs* w, unsigned char* ws, int B, int P) {
    w->base = ws;
    w->gbuf = reinterpret_cast<float*>(ws + w->n_state);
    w->lbuf = w->gbuf + (size_t)B * P;
    w->gbuf2 = w->lbuf + B;
    w->lbuf2 = w->gbuf2 + (size_t)B * P;
    const size_t n_res = (2 * ((size_t)B * P + B) * sizeof(float) + 15) / 16 * 16;
    return reinterpret_cast<float*>(ws + w->n_state + n_res);
}
int lbfgs_run(const k2b_model* model_c, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t B, int32_t K,
              const int32_t* model_joint_index, const float* j3d, const float* conf, const float* pres, const float* trp,
              float* go, float* bp, float* be, float* tr, float* loss_out, float* grad_out, int max_iter, int H, double lr,
              double tol_g, double tol_c, LbfgsWs& w, void* stream_v) {
    hipStream_t stream = (hipStream_t)stream_v;
    const int NB = model_c->NB, D = 3 * (model_c->J - 1), P = 3 + D + NB + 3;
    const int max_eval = max_iter * 5 / 4;                               // torch's default
    if (!w.state_cleared) HIP_TRY(hipMemsetAsync(w.base, 0, w.off_sv, stream));   // scalars and integers: phase INIT (vectors are written before they are read)
    w.state_cleared = false;
    k2b_fit_config ec = *cfg;
    ec.num_iters = 1;
    ec.step_size = 0.0;                                                  // evaluate-only: the closure
    auto closure = [&](float* loss, float* grad) {
        return fit_world_impl(model_c, prior, &ec, B, K, model_joint_index, j3d, conf, go, bp, be, tr, pres, trp,
                              go, bp, be, tr, loss, grad, stream_v, 1, 0);
    };
    k2b::LbfgsArgs la{};
    la.B = B; la.P = P; la.D = D; la.NB = NB; la.H = H;
    la.max_iter = max_iter; la.max_eval = max_eval;
    la.lr = lr; la.tol_g = tol_g; la.tol_c = tol_c;
    la.go = go; la.bp = bp; la.be = be; la.tr = tr;
    la.loss_in = w.lbuf; la.grad_in = w.gbuf;
    la.sd = reinterpret_cast<double*>(w.base); la.si = reinterpret_cast<int*>(w.base + w.off_si); la.sv = reinterpret_cast<float*>(w.base + w.off_sv);
    const int rounds = max_eval + 2;
    // One launch per round where the fused kernel takes the frames (24-joint model, the prior over the whole pose, kinematic
    // targets only, at most four frames per CU): launch r = [step on the result of launch r - 1 | closure]; the last launch =
    // [finalise | closure] with the caller's outputs.  Two result buffers alternate (a launch reads the one its predecessor
    // wrote while writing the other).  Otherwise two launches per round.
    // (development: K2B_LBFGS_SCHEME = 1 forces two launches per round, 2 the fused rounds wherever they apply; tools/dev_lbfgs_schemes.py)
    static const int scheme_env = [] { const char* e = getenv("K2B_LBFGS_SCHEME"); return e ? atoi(e) : 0; }();
    bool fused = scheme_env != 1 && model_c->fit_ok && (B + device_cus() - 1) / device_cus() <= 4;
    if (fused) {
        const int pose_dims_all = 3 * (model_c->J - 1);
        const int prior_dims = cfg->prior_pose_dims > 0 ? cfg->prior_pose_dims : (prior->D < pose_dims_all ? prior->D : pose_dims_all);
        fused = prior->D == pose_dims_all && prior_dims == pose_dims_all && (cfg->num_betas_prior == 0 || cfg->num_betas_prior == model_c->NB);
        for (int k = 0; k < K && fused; ++k) fused = model_joint_index[k] >= 0 && model_joint_index[k] < model_c->J;
    }
    if (fused) {
        // the optimiser's arguments travel in the launch's own arguments: [0] reads result buffer A, [1] reads B
        k2b::LbfgsArgs both[2] = {la, la};
        both[0].loss_in = w.lbuf;  both[0].grad_in = w.gbuf;
        both[1].loss_in = w.lbuf2; both[1].grad_in = w.gbuf2;
        auto fused_launch = [&](int mode, int read_sel, float* loss, float* grad) {
            return fit_world_impl(model_c, prior, &ec, B, K, model_joint_index, j3d, conf, go, bp, be, tr, pres, trp,
                                  go, bp, be, tr, loss, grad, stream_v, 1, 0, mode, &both[read_sel]);
        };
        if (scheme_env != 2 && (B + device_cus() - 1) / device_cus() <= 2) {
            // at most two frames per CU: the whole fit is ONE persistent launch - rounds closures, each followed by its step on an
            // idle wave of the workgroup, the finalise pass and the closure at the result (k2b_fit.hip, lb_mode 3)
            k2b_fit_config pc = ec;
            pc.num_iters = rounds + 1;
            return fit_world_impl(model_c, prior, &pc, B, K, model_joint_index, j3d, conf, go, bp, be, tr, pres, trp,
                                  go, bp, be, tr, loss_out ? loss_out : w.lbuf2, grad_out, stream_v, 1, 0, 3, &both[0]);
        }
        // launch 0: closure only, writes A; launch r >= 1 reads (r - 1) & 1 and writes r & 1
        if (const int rc = closure(w.lbuf, w.gbuf); rc != K2B_OK) return rc;
        for (int r = 1; r <= rounds; ++r) {
            const int rd = (r - 1) & 1, wr = r & 1;
            if (r < rounds) {
                if (const int rc = fused_launch(1, rd, wr ? w.lbuf2 : w.lbuf, wr ? w.gbuf2 : w.gbuf); rc != K2B_OK) return rc;
            } else {
                // the last step consumes result r - 1, then the finalise launch parks every frame and evaluates the result
                if (const int rc = fused_launch(1, rd, wr ? w.lbuf2 : w.lbuf, wr ? w.gbuf2 : w.gbuf); rc != K2B_OK) return rc;
                return fused_launch(2, wr, loss_out ? loss_out : (rd ? w.lbuf2 : w.lbuf), grad_out);
            }
        }
    }
    for (int r = 0; r < rounds; ++r) {
        if (const int rc = closure(w.lbuf, w.gbuf); rc != K2B_OK) return rc;
        HIP_TRY(k2b::launch_lbfgs_step(la, stream));
    }
    la.finalize = 1;
    HIP_TRY(k2b::launch_lbfgs_step(la, stream));
    // loss (and gradient) at the result (world_space.py:245-246 evaluates the loss once more behind the optimiser)
    return closure(loss_out ? loss_out : w.lbuf, grad_out);
}
int lbfgs_check(const k2b_model* model, const k2b_prior* prior, const k2b_fit_config* cfg, int max_iter, int* history_size, double lr,
                const char* who) {
    if (!model || !prior || !cfg) return fail(K2B_ERR_INVALID_ARGUMENT, "%s: model, prior and cfg are required", who);
    if (max_iter < 1 || max_iter > 10000) return fail(K2B_ERR_INVALID_ARGUMENT, "%s: max_iter=%d", who, max_iter);
    if (*history_size <= 0) *history_size = k2b::kLbfgsMaxHistory;
    if (*history_size > k2b::kLbfgsMaxHistory)
        return fail(K2B_ERR_UNSUPPORTED, "%s: history_size=%d (at most %d)", who, *history_size, k2b::kLbfgsMaxHistory);
    if (!(lr > 0.0)) return fail(K2B_ERR_INVALID_ARGUMENT, "%s: lr must be positive", who);
    if (3 + 3 * (model->J - 1) + model->NB + 3 > 192)
        return fail(K2B_ERR_UNSUPPORTED, "%s: %d parameters per frame (at most 192)", who, 3 + 3 * (model->J - 1) + model->NB + 3);
    return K2B_OK;
}
}  // namespace

// L-BFGS branch of the fitters on the device (world_space.py:231-247, camera_space.py:144-182,229-267): per frame
// torch.optim.LBFGS(max_iter, lr, line_search_fn="strong_wolfe").step(closure), the closure = this library's evaluate-only fit
// launch, the optimiser = k2b_lbfgs.hip's state machine.  Only launches are queued: max_eval + 2 rounds of [closure, step], then
// the accepted points go back into the parameter arrays and one more closure launch leaves the final loss (+ gradient).
int k2b_fit_world_lbfgs(const k2b_model* model_c, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t B, int32_t K,
                        const int32_t* model_joint_index, const float* j3d, const float* conf, const float* go_in,
                        const float* bp_in, const float* be_in, const float* tr_in, const float* preserve, const float* tr_prior,
                        float* go_out, float* bp_out, float* be_out, float* tr_out, float* loss_out, float* grad_out,
                        int32_t max_iter, int32_t history_size, double lr, double tolerance_grad, double tolerance_change,
                        void* stream_v) {
    if (const int rc = lbfgs_check(model_c, prior, cfg, max_iter, &history_size, lr, "k2b_fit_world_lbfgs"); rc != K2B_OK) return rc;
    if (B < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world_lbfgs: num_frames=%d", B);
    if (B == 0) return K2B_OK;
    if (!go_in || !bp_in || !be_in || !tr_in || !go_out || !bp_out || !be_out || !tr_out)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_world_lbfgs: NULL parameter buffer");
    hipStream_t stream = (hipStream_t)stream_v;
    const int NB = model_c->NB, D = 3 * (model_c->J - 1), P = 3 + D + NB + 3;
    const int H = history_size < max_iter ? history_size : max_iter;      // (a fit makes at most max_iter - 1 pairs)
    // stream-ordered workspace: optimiser state, closure results, the preserve pose and the translation prior's centre (their
    // defaults are the INITIAL parameters, which the parameter arrays stop holding after the first step)
    LbfgsWs w{};
    const size_t n_opt = lbfgs_ws_layout(B, P, H, &w);
    unsigned char* ws = nullptr;
    HIP_TRY(hipMallocAsync((void**)&ws, n_opt + ((size_t)B * D + (size_t)B * 3) * sizeof(float), stream));
    auto cleanup = [&](int rc) { (void)hipFreeAsync(ws, stream); return rc; };
#define K2B_TRY_WS(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); return cleanup(fail(K2B_ERR_HIP, "k2b_fit_world_lbfgs: HIP call failed")); } } while (0)
    float *pres = lbfgs_ws_assign(&w, ws, B, P), *trp = pres + (size_t)B * D;
    K2B_TRY_WS(hipMemcpyAsync(pres, preserve ? preserve : bp_in, (size_t)B * D * sizeof(float), hipMemcpyDeviceToDevice, stream));
    K2B_TRY_WS(hipMemcpyAsync(trp, tr_prior ? tr_prior : tr_in, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToDevice, stream));
    const struct { const float* src; float* dst; size_t n; } cp[] = {
        {go_in, go_out, (size_t)B * 3}, {bp_in, bp_out, (size_t)B * D}, {be_in, be_out, (size_t)B * NB}, {tr_in, tr_out, (size_t)B * 3}};
    for (const auto& c : cp)
        if (c.src != c.dst) K2B_TRY_WS(hipMemcpyAsync(c.dst, c.src, c.n * sizeof(float), hipMemcpyDeviceToDevice, stream));
#undef K2B_TRY_WS
    return cleanup(lbfgs_run(model_c, prior, cfg, B, K, model_joint_index, j3d, conf, pres,
                             (tr_prior || cfg->transl_prior_weight != 0.0f) ? trp : nullptr,   // (the tree kernel has no translation prior)
                             go_out, bp_out, be_out, tr_out, loss_out, grad_out, max_iter, H, lr, tolerance_grad, tolerance_change, w, stream_v));
}

// The reference's DEFAULT sequence mode in one call: the frame loop of optimize_params_sequence with use_previous_frame_init=True
// (api/sequence.py:214-281) over the L-BFGS branch (world_space.py:231-247).  Frame 0 is fitted from the given start with
// first_iters iterations and no preserve term; every later frame starts from its predecessor's RESULT, preserves that result's
// body pose with cfg->pose_preserve_weight (world_space.py:159,211) and runs followup_iters iterations; each frame is one
// device-driven L-BFGS fit (k2b_fit_world_lbfgs) and only launches are queued - no host work between the frames.
int k2b_fit_sequence_lbfgs(const k2b_model* model_c, const k2b_prior* prior, const k2b_fit_config* cfg, int32_t T, int32_t K,
                           const int32_t* model_joint_index, const float* j3d, const float* conf, const float* go_in,
                           const float* bp_in, const float* be_in, const float* tr_in, float* go_out, float* bp_out, float* be_out,
                           float* tr_out, float* loss_out, int32_t first_iters, int32_t followup_iters, int32_t history_size, double lr,
                           double tolerance_grad, double tolerance_change, void* stream_v) {
    if (const int rc = lbfgs_check(model_c, prior, cfg, first_iters, &history_size, lr, "k2b_fit_sequence_lbfgs"); rc != K2B_OK) return rc;
    if (followup_iters < 1 || followup_iters > 10000) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence_lbfgs: followup_iters=%d", followup_iters);
    if (T < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence_lbfgs: frames=%d", T);
    if (cfg->transl_prior_weight != 0.0f) return fail(K2B_ERR_UNSUPPORTED, "k2b_fit_sequence_lbfgs: no translation prior in a chain");
    if (T == 0) return K2B_OK;
    if (!j3d || !go_in || !bp_in || !be_in || !tr_in || !go_out || !bp_out || !be_out || !tr_out)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_fit_sequence_lbfgs: NULL parameter / target buffer");
    hipStream_t stream = (hipStream_t)stream_v;
    const int NB = model_c->NB, D = 3 * (model_c->J - 1), P = 3 + D + NB + 3;
    const int it_max = first_iters > followup_iters ? first_iters : followup_iters;
    const int Hmax = history_size < it_max ? history_size : it_max;
    LbfgsWs w{};
    const size_t n_opt = lbfgs_ws_layout(1, P, Hmax, &w);
    unsigned char* ws = nullptr;
    HIP_TRY(hipMallocAsync((void**)&ws, n_opt + (size_t)D * sizeof(float), stream));
    auto cleanup = [&](int rc) { (void)hipFreeAsync(ws, stream); return rc; };
#define K2B_TRY_WS(expr) do { if ((expr) != hipSuccess) { (void)hipGetLastError(); return cleanup(fail(K2B_ERR_HIP, "k2b_fit_sequence_lbfgs: HIP call failed")); } } while (0)
    float* pres = lbfgs_ws_assign(&w, ws, 1, P);
    k2b_fit_config fc = *cfg;
    // The whole sequence in ONE launch where the fused kernel takes it (24-joint model, the prior over the whole pose, kinematic
    // targets): the frame loop runs inside the persistent launch - per frame a fresh optimiser on the workgroup's idle wave, the
    // start and the preserve pose from the predecessor's result in registers (k2b_fit.hip: chain + lb_mode 3).
    {
        static const int scheme_env = [] { const char* e = getenv("K2B_LBFGS_SCHEME"); return e ? atoi(e) : 0; }();
        const int pose_dims_all = 3 * (model_c->J - 1);
        const int prior_dims = cfg->prior_pose_dims > 0 ? cfg->prior_pose_dims : (prior->D < pose_dims_all ? prior->D : pose_dims_all);
        bool one = scheme_env == 0 && T > 1 && model_c->fit_ok && prior->D == pose_dims_all && prior_dims == pose_dims_all &&
                   (cfg->num_betas_prior == 0 || cfg->num_betas_prior == model_c->NB) && cfg->transl_prior_weight == 0.0f;
        for (int k = 0; k < K && one; ++k) one = model_joint_index[k] >= 0 && model_joint_index[k] < model_c->J;
        if (one) {
            const int me_first = first_iters * 5 / 4, me_follow = followup_iters * 5 / 4;
            k2b::LbfgsArgs la{};
            la.B = 1; la.P = P; la.D = D; la.NB = NB; la.H = Hmax;
            la.max_iter = first_iters; la.max_eval = me_first;
            la.lr = lr; la.tol_g = tolerance_grad; la.tol_c = tolerance_change;
            la.go = go_out; la.bp = bp_out; la.be = be_out; la.tr = tr_out;       // rows t of the outputs: frame t's point
            la.loss_in = w.lbuf; la.grad_in = w.gbuf;
            la.sd = reinterpret_cast<double*>(w.base); la.si = reinterpret_cast<int*>(w.base + w.off_si); la.sv = reinterpret_cast<float*>(w.base + w.off_sv);
            K2B_TRY_WS(hipMemsetAsync(w.base, 0, w.off_sv, stream));
            k2b_fit_config pc = fc;
            pc.num_iters = me_first + 3;                                          // rounds + the closure at the result
            pc.step_size = 0.0;
            return cleanup(fit_world_impl(model_c, prior, &pc, 1, K, model_joint_index, j3d, conf, go_in, bp_in, be_in, tr_in, nullptr, nullptr,
                                          go_out, bp_out, be_out, tr_out, loss_out, nullptr, stream_v, T, me_follow + 3, 3, &la, followup_iters));
        }
    }
    for (int t = 0; t < T; ++t) {
        float *go = go_out + (size_t)t * 3, *bp = bp_out + (size_t)t * D, *be = be_out + (size_t)t * NB, *tr = tr_out + (size_t)t * 3;
        const float *sgo = t ? go - 3 : go_in, *sbp = t ? bp - D : bp_in, *sbe = t ? be - NB : be_in, *str = t ? tr - 3 : tr_in;
        // start of this frame = the start given (frame 0) or the previous frame's result; its body pose is also what is preserved
        // (one small launch: the four parameter rows, the preserve pose and the cleared optimiser state)
        K2B_TRY_WS(k2b::launch_lbfgs_frame_prep(go, sgo, bp, sbp, be, sbe, tr, str, pres, D, NB, w.base, w.off_sv, stream));
        w.state_cleared = true;
        fc.pose_preserve_weight = t ? cfg->pose_preserve_weight : 0.0f;
        const int iters = t ? followup_iters : first_iters;
        const int H = history_size < iters ? history_size : iters;
        const float* cf = conf ? conf + (cfg->conf_per_frame ? (size_t)t * K : 0) : nullptr;
        k2b_fit_config one = fc;
        one.conf_per_frame = 0;                                   // (one frame per fit: its row of a per-frame array is a shared row)
        if (const int rc = lbfgs_run(model_c, prior, &one, 1, K, model_joint_index, j3d + (size_t)t * K * 3, cf, pres, nullptr, go, bp, be, tr,
                                     loss_out ? loss_out + t : nullptr, nullptr, iters, H, lr, tolerance_grad, tolerance_change, w, stream_v);
            rc != K2B_OK) return cleanup(rc);
    }
#undef K2B_TRY_WS
    return cleanup(K2B_OK);
}

namespace {
// grow-only per-model workspace of the per-frame LBS operands (caller holds m->mu)
int reserve_lbs_workspace(k2b_model* m, int bpad) {
    if (bpad <= m->ws_bpad) return K2B_OK;
    HIP_TRY(hipDeviceSynchronize());
    k2b::k2b_half** ws[] = {&m->wsXh, &m->wsXl, &m->wsA2};
    for (auto w : ws) { if (*w) HIP_TRY(hipFree(*w)); *w = nullptr; }
    m->ws_bpad = 0;
    const size_t nx = (size_t)m->k_steps_x * bpad * 16;
    const size_t na2 = (size_t)(bpad / 16) * 12 * k2b::tile_ngp(m->groups_a) * 128;
    HIP_TRY(hipMalloc((void**)&m->wsA2, na2 * sizeof(k2b::k2b_half)));
    HIP_TRY(hipMemset(m->wsA2, 0, na2 * sizeof(k2b::k2b_half)));    // PAD / ZERO groups and padding frames stay zero
    HIP_TRY(hipMalloc((void**)&m->wsXh, nx * sizeof(k2b::k2b_half)));
    HIP_TRY(hipMalloc((void**)&m->wsXl, nx * sizeof(k2b::k2b_half)));
    // rows of padding frames are never written by the set-up kernel: keep them finite
    HIP_TRY(hipMemset(m->wsXh, 0, nx * sizeof(k2b::k2b_half)));
    HIP_TRY(hipMemset(m->wsXl, 0, nx * sizeof(k2b::k2b_half)));
    m->ws_bpad = bpad;
    return K2B_OK;
}
}  // namespace

extern "C" int k2b_model_reserve(k2b_model* m, int32_t max_frames) {
    if (!m) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_reserve: model is NULL");
    if (max_frames < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_model_reserve: max_frames=%d", max_frames);
    std::lock_guard<std::mutex> lk(m->mu);
    return reserve_lbs_workspace(m, k2b::lbs_frames_padded(max_frames));
}

int k2b_lbs(const k2b_model* model_c, int32_t B, const float* go, const float* bp, const float* be, const float* tr,
            float* joints_out, float* vertices_out, void* stream_v) {
    k2b_model* m = const_cast<k2b_model*>(model_c);
    if (!m) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_lbs: model is NULL");
    if (B < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_lbs: num_frames=%d", B);
    if (B == 0) return K2B_OK;
    if (!go || !bp || !be) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_lbs: NULL parameter buffer");
    if (!joints_out && !vertices_out) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_lbs: no output requested");
    hipStream_t stream = (hipStream_t)stream_v;
    const int bpad = k2b::lbs_frames_padded(B);
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (const int rc = reserve_lbs_workspace(m, bpad); rc != K2B_OK) return rc;
    }
    k2b::PoseArgs pa{};
    pa.j_basis_lane = m->j_basis_lane; pa.parents = m->parents;
    pa.num_joints = m->J; pa.num_betas = m->NB; pa.num_out_joints = m->J + m->E;
    pa.num_frames = B; pa.frames_padded = bpad; pa.k_steps_x = m->k_steps_x;
    pa.go = go; pa.bp = bp; pa.be = be; pa.tr = tr;
    if (m->groups_a != 3 && m->groups_a != 7)
        return fail(K2B_ERR_UNSUPPORTED, "k2b_lbs: %d joints; the vertex kernel is built for 17-24 (SMPL) and 49-56 (SMPL-H / SMPL-X) joints", m->J);
    pa.xh = m->wsXh; pa.xl = m->wsXl; pa.a2 = m->wsA2; pa.joints_out = joints_out;
    pa.a2_stream_order = (m->stream || m->stream_x) ? 1 : 0;
    HIP_TRY(k2b::launch_pose_setup(pa, stream));
    auto skin = [&](const k2b_model::VertexSet& vs, float* out, int stride, int row0, float* joint_copies) -> hipError_t {
        if (m->stream || m->stream_x) {
            k2b::StreamArgs sa{};
            sa.xh = m->wsXh; sa.xl = m->wsXl; sa.a2 = m->wsA2; sa.pd = vs.spd; sa.w = vs.sw;
            sa.f32_tiles = bpad / 32; sa.nv16 = vs.nv16;
            sa.num_frames = B; sa.num_out = vs.num; sa.out = out; sa.out_stride = stride; sa.out_row0 = row0;
            sa.dump = m->dump;
            sa.joints_out = joint_copies; sa.joints_stride = m->J + m->E; sa.joints_row0 = m->J;
            return m->stream ? k2b::launch_skin_stream(sa, device_cus(), stream) : k2b::launch_skin_stream_x(sa, device_cus(), stream);
        }
        k2b::TileArgs ta{};
        ta.xh = m->wsXh; ta.xl = m->wsXl; ta.a2 = m->wsA2; ta.pdh = vs.pdh; ta.pdl = vs.pdl; ta.w2 = vs.w2;
        ta.groups_a = m->groups_a; ta.k_steps_x = m->k_steps_x; ta.f_tiles = bpad / 32; ta.v_tiles = vs.v_tiles;
        ta.num_frames = B; ta.num_out = vs.num; ta.out = out; ta.out_stride = stride; ta.out_row0 = row0;
        ta.dump = m->dump;
        ta.joints_out = joint_copies; ta.joints_stride = m->J + m->E; ta.joints_row0 = m->J;
        return k2b::launch_skin_tiles(ta, device_cus(), stream);
    };
    if (vertices_out) {
        // the mesh launch also writes the vertex-selected joints (their vertices are tagged in the W image)
        const bool copies = joints_out && m->E > 0 && m->joints_in_mesh;
        HIP_TRY(skin(m->mesh, vertices_out, m->V, 0, copies ? joints_out : nullptr));
        if (joints_out && m->E > 0 && !copies)
            HIP_TRY(k2b::launch_gather_joints(vertices_out, m->extra_ids, joints_out, B, m->V, m->J, m->E, stream));
    } else if (joints_out && m->E > 0) {
        HIP_TRY(skin(m->extra, joints_out, m->J + m->E, m->J, nullptr));
    }
    return K2B_OK;
}

int k2b_vertex_term(const k2b_model* model_c, int32_t B, int32_t E_sel, const int32_t* extra_index, const float* targets,
                    const float* conf, float sigma, float joint_loss_weight, const float* go, const float* bp, const float* be,
                    const float* tr, float* loss_out, float* grad_out, void* stream_v) {
    k2b_model* m = const_cast<k2b_model*>(model_c);
    if (!m) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_vertex_term: model is NULL");
    if (B < 0 || E_sel < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_vertex_term: negative size");
    if (B == 0 || E_sel == 0) return K2B_OK;
    if (m->J > 64 || m->NB > 32) return fail(K2B_ERR_UNSUPPORTED, "k2b_vertex_term: %d joints / %d shape coefficients, at most 64 / 32", m->J, m->NB);
    if (E_sel > 32) return fail(K2B_ERR_UNSUPPORTED, "k2b_vertex_term: %d vertex-selected joints, at most 32 per call", E_sel);
    if (!extra_index || !targets || !go || !bp || !be || !tr || !loss_out || !grad_out)
        return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_vertex_term: NULL buffer");
    for (int e = 0; e < E_sel; ++e)
        if (extra_index[e] < 0 || extra_index[e] >= m->E)
            return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_vertex_term: extra_index[%d]=%d outside [0,%d)", e, extra_index[e], m->E);
    hipStream_t stream = (hipStream_t)stream_v;
    k2b::VertexTermArgs a{};
    a.v_template = m->v_template; a.shapedirs = m->shapedirs; a.posedirs = m->posedirs; a.lbs_weights = m->lbs_weights;
    a.j_template = m->j_template; a.j_dirs = m->j_dirs; a.parents = m->parents; a.extra_ids = m->extra_ids;
    a.num_vertices = m->V; a.num_betas = m->NB; a.num_joints = m->J;
    a.num_frames = B; a.num_sel = E_sel; a.targets = targets; a.conf = conf;
    for (int e = 0; e < E_sel; ++e) { a.sel[e] = extra_index[e]; a.sel_k[e] = e; }
    a.num_targets = E_sel;
    a.sigma = sigma; a.joint_w = joint_loss_weight;
    a.go = go; a.bp = bp; a.be = be; a.tr = tr; a.loss_out = loss_out; a.grad_out = grad_out;
    HIP_TRY(k2b::launch_vertex_term(a, stream));
    return K2B_OK;
}

int k2b_adam_step(int64_t n, float* params, const float* grad, float* mbuf, float* vbuf, int32_t step, double step_size,
                  double beta1, double beta2, double eps, void* stream_v) {
    if (n < 0 || step < 1) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_adam_step: n=%lld step=%d", (long long)n, step);
    if (n == 0) return K2B_OK;
    if (!params || !grad || !mbuf || !vbuf) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_adam_step: NULL buffer");
    const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
    HIP_TRY(k2b::launch_adam(params, grad, mbuf, vbuf, (long long)n, (float)(step_size / bc1), (float)std::sqrt(bc2),
                             (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (hipStream_t)stream_v));
    return K2B_OK;
}

int k2b_angular_error_deg(int64_t n, const float* pred_rotvec, const float* gt_rotvec, float* err_deg_out, void* stream_v) {
    if (n < 0) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_angular_error_deg: n=%lld must be >= 0", (long long)n);
    if (n == 0) return K2B_OK;
    if (!pred_rotvec || !gt_rotvec || !err_deg_out) return fail(K2B_ERR_INVALID_ARGUMENT, "k2b_angular_error_deg: NULL buffer");
    if (n > (int64_t)0x7fffffff * 256) return fail(K2B_ERR_UNSUPPORTED, "k2b_angular_error_deg: n=%lld exceeds one launch", (long long)n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(K2B_ERR_NO_DEVICE, "k2b_angular_error_deg: no HIP device visible (this engine has no CPU path)");
    HIP_TRY(k2b::launch_angular_error(pred_rotvec, gt_rotvec, err_deg_out, (long long)n, (hipStream_t)stream_v));
    return K2B_OK;
}

}  // extern "C"
