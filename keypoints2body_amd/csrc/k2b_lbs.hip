// k2b_lbs.hip — full SMPL forward (pose set-up + vertex skinning) for gfx950.
//
// Replaces the body-model call `self.smpl(**kwargs)` of the reference
// (keypoints2body/core/fitters/world_space.py:34,192,278), i.e. smplx's SMPL.forward /
// lbs(): shape blend, pose-corrective blend, kinematic chain, linear blend skinning,
// vertex-selected extra joints, translation.  CPU twin: oracle/smpl_torch.py.
//
// The two contractions of LBS are GEMMs over (frames x vertices):
//   v_posed[f][v][c] = sum_k X[f][k] * Pd[k][v][c]      K = 9(J-1) + NB + 2   (224 for SMPL)
//        X = [vec(R_1..R_{J-1} - I) | beta | 1 | 1],  Pd = [posedirs ; shapedirs ; v_template ; residual]
//   T[f][v][e]       = sum_j W[v][j] * A[f][j][e]       K = J (24), e = 12 entries of the 3x4 transform
// and the result is v[f][v] = T[f][v] [v_posed; 1] + transl[f].
//
// Both run on the matrix cores.  fp32-input MFMA issues at the fp32 VALU rate, so operands are
// split into two f16 terms (x = hi + lo, ~22 mantissa bits) and each product takes three
// v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate): 3/16 of the fp32 MFMA time
// at fp32-level accuracy (|error| ~ 1e-6 m on metre-scale vertices; tests/test_gpu_parity.py).
// (A single f16 term for the pose-corrective rows of Pd - two products instead of three - was priced in
//  round 3: the dropped x_hi * Pd_lo product is 6.4e-6 m at the synthetic model's posedirs magnitude,
//  over the 5e-6 gate)
//
// MFMA orientation D[frame][vertex]: a lane of the accumulator holds four frames of ONE vertex, so the
// skinning epilogue (T applied to v_posed) is a per-lane computation with no cross-lane traffic and the
// three coordinates of a (frame, vertex) pair leave as ONE 12-byte store.  Operands are stored in exactly
// the order the 64 lanes of a wave consume them, in 1 KiB pieces, so a piece moves global -> LDS as one
// linear copy (global_load_lds_dwordx4) and is read back with conflict-free ds_read_b128.
#include <hip/hip_fp16.h>

#include "k2b_internal.h"

namespace k2b {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
struct __attribute__((packed, aligned(4))) float3v { float x, y, z; };

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Timing-only diagnostic builds (no stores / no fills / no MFMAs / L2-resident stores / s_memtime stamps) live outside this file:
// tools/build_lbs_variants.sh compiles it with -DK2B_LBS_DIAG_HEADER=<tools/lbs_diag.h>, which redefines the hooks below.
#ifdef K2B_LBS_DIAG_HEADER
#include K2B_LBS_DIAG_HEADER
#else
#define K2B_DIAG_SKIP_FILL(lq) false              // true: this slice's fills are not issued
#define K2B_DIAG_STORE(f, ok) ((void)0)           // may redirect the frame row / validity of a store
#define K2B_DIAG_MFMA(x, y, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0)
#define K2B_DIAG_STAMP_DECL ((void)0)
#define K2B_DIAG_STAMP(slice, k) ((void)0)
#define K2B_DIAG_TILE_DONE ((void)0)
#define K2B_DIAG_KERNEL_END ((void)0)
#endif
#ifndef K2B_PSTAMP                                 // phase stamps of the pose set-up kernel (tools/dev_pose_stamps.py)
#define K2B_PSTAMP_DECL ((void)0)
#define K2B_PSTAMP(i) ((void)0)
#define K2B_PSTAMP_END ((void)0)
#endif

// ---------------------------------------------------------------------------------------------
// Pose set-up: one 64-lane workgroup per frame, lane j = joint j.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxXSteps = 32;       // 16-deep k-steps of the feature vector this kernel can stage (SMPL: 14, SMPL-X: 32)

// NBC = capacity of the shape loop (10 / 16 / 20 / 32: the unrolled J(beta) loads and products are a third of this
// kernel's instructions at 32), GA = ceil(J / 8) (compile-time divisors in the A2 store loop)
template <int NBC, int GA>
__global__ __launch_bounds__(64) void k2b_pose_setup_kernel(const PoseArgs a) {
    __shared__ float sR[kMaxJoints][9];
    __shared__ float sd[kMaxJoints][3];
    __shared__ int spar[kMaxJoints];
    // Workgroups go round-robin over the 8 XCDs, and the 16 / 32 frames that share a 256-byte group of A (a 1 KiB piece of X)
    // write 16 bytes of it each: with consecutive frames on consecutive XCDs every L2 holds two of a line's sixteen pieces and
    // writes them back as partial lines.  XCD label x = block % 8 therefore takes the CONTIGUOUS frames [x per, (x + 1) per),
    // per a multiple of 32: a line is completed inside one L2.
    const int f = a.xcd_frames ? (int)(blockIdx.x & 7) * a.xcd_frames + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (f >= a.num_frames) return;                       // (whole workgroup, before any barrier)
    const int j = threadIdx.x;
    const int J = a.num_joints, NB = a.num_betas;
    const bool act = j < J;
    const int bp = a.frames_padded;
    K2B_PSTAMP_DECL;
    K2B_PSTAMP(0);

    // Every load of this prologue is issued before the first use (fixed trip counts, predicated on k < NB): the kernel is
    // one dependent chain per frame, and a loop of NB load -> fma round trips to L2 was a third of its time.
    __shared__ float sJ[kMaxJoints][3];
    Vec3 th = {0.f, 0.f, 0.f};
    Vec3 Jj = {0.f, 0.f, 0.f};
    int par = -1;
    if (act) {
        const float* src = j == 0 ? a.go + (size_t)f * 3 : a.bp + (size_t)f * 3 * (J - 1) + 3 * (j - 1);
        th = {src[0], src[1], src[2]};
        par = a.parents[j];
        float beta[NBC];
#pragma unroll
        for (int k = 0; k < NBC; ++k) beta[k] = k < NB ? a.be[(size_t)f * NB + k] : 0.f;
        float e[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float dir[NBC];
            const float* tab = a.j_basis_lane + (size_t)c * (1 + NB) * 64 + j;     // lane-major: a load = 1-2 cache lines for the wave
#pragma unroll
            for (int k = 0; k < NBC; ++k) dir[k] = k < NB ? tab[(1 + k) * 64] : 0.f;
            float s = tab[0];
#pragma unroll
            for (int k = 0; k < NBC; ++k) s += dir[k] * beta[k];      // (k >= NB adds exact zeros: same sum for every capacity)
            e[c] = s;
        }
        Jj = {e[0], e[1], e[2]};
        sJ[j][0] = e[0]; sJ[j][1] = e[1]; sJ[j][2] = e[2];
        spar[j] = par;
    }
    K2B_PSTAMP(1);
    const Rodrigues rod = rodrigues_fwd(th);
    if (act) {
        for (int i = 0; i < 9; ++i) sR[j][i] = rod.R.m[i];
    }
    __syncthreads();
    K2B_PSTAMP(2);
    if (act) {      // offset from the parent's rest joint (the parent's J(beta) comes from its own lane)
        const Vec3 Jp = par >= 0 ? Vec3{sJ[par][0], sJ[par][1], sJ[par][2]} : Vec3{0.f, 0.f, 0.f};
        const Vec3 d = Jj - Jp;
        sd[j][0] = d.x; sd[j][1] = d.y; sd[j][2] = d.z;
    }
    __syncthreads();

    // The f16 hi/lo operands of this frame are staged in LDS in k order and leave as 16-byte chunks: inside a
    // fragment the eight k values of one (h, frame) pair are contiguous (frag_elem), so a frame owns 4 chunks
    // per X k-step and 8 per (entry, joint k-step) of A.  (Element-wise 2-byte stores - 1200 per frame - made
    // this kernel store-issue bound.)
    __shared__ __attribute__((aligned(16))) _Float16 sXh[kMaxXSteps * 16], sXl[kMaxXSteps * 16];
    __shared__ __attribute__((aligned(16))) _Float16 sAh[12][kMaxJoints], sAl[12][kMaxJoints];
    __shared__ __attribute__((aligned(16))) _Float16 sTp[3][8];      // translation as three f16 terms (PAD group of the tile kernel)
    auto put_x = [&](int k, float x) {
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        sXh[k] = hi;
        sXl[k] = lo;
    };
    const int P = 9 * (J - 1);
    if (act && j > 0) {
        for (int i = 0; i < 9; ++i) put_x((j - 1) * 9 + i, rod.R.m[i] - ((i % 4 == 0) ? 1.f : 0.f));
    }
    for (int k = P + j; k < a.k_steps_x * 16; k += 64) {   // betas, the two constant-1 features, zero padding
        const float x = k < P + NB ? a.be[(size_t)f * NB + (k - P)] : (k < P + NB + 2 ? 1.f : 0.f);
        put_x(k, x);
    }

    K2B_PSTAMP(3);
    // global transform: compose towards the root
    Mat3 Rg = rod.R;
    Vec3 pg = {0.f, 0.f, 0.f};
    if (act) {
        pg = {sd[j][0], sd[j][1], sd[j][2]};
        for (int anc = par; anc >= 0; anc = spar[anc]) {
            Mat3 Ra;
            for (int i = 0; i < 9; ++i) Ra.m[i] = sR[anc][i];
            const Vec3 da = {sd[anc][0], sd[anc][1], sd[anc][2]};
            pg = mul(Ra, pg) + da;
            Rg = mul(Ra, Rg);
        }
    }
    K2B_PSTAMP(4);
    // A_j = [Rg | pg - Rg J_j] (k = joint; zero rows for the padded joints J .. 16 k_steps_a - 1)
    {
        const Vec3 rj = mul(Rg, Jj);
        const float At[12] = {Rg.m[0], Rg.m[1], Rg.m[2], pg.x - rj.x, Rg.m[3], Rg.m[4], Rg.m[5], pg.y - rj.y,
                              Rg.m[6], Rg.m[7], Rg.m[8], pg.z - rj.z};
        for (int e = 0; e < 12; ++e) {
            _Float16 hi = (_Float16)0.f, lo = (_Float16)0.f;
            if (act) split_f16(At[e], hi, lo);
            sAh[e][j] = hi;
            sAl[e][j] = lo;
        }
    }
    __syncthreads();
    K2B_PSTAMP(5);
    {
        const int tiles = bp >> 5, tile = f >> 5;
        const int nx = 2 * a.k_steps_x;                    // chunks per X array
        for (int c = j; c < 2 * nx; c += 64) {
            const int lo = c >= nx, kc = lo ? c - nx : c, ks = kc >> 1, h = kc & 1;
            const uint4 v = *reinterpret_cast<const uint4*>((lo ? sXl : sXh) + ks * 16 + 8 * h);
            *reinterpret_cast<uint4*>((lo ? a.xl : a.xh) + frag_elem((size_t)ks * tiles + tile, 8 * h, f)) = v;
        }
    }
    K2B_PSTAMP(6);
    if (a.a2) {
        // group layout of the tile kernel: [16-frame tile][entry][hi groups | lo groups | PAD | ZERO][row 16][8]; the PAD group of
        // the entries 3, 7, 11 carries the translation as three f16 terms (hi + mid + lo: 33 bits), ZERO is never written
        if (j < 3) {
            const float t = a.tr ? a.tr[(size_t)f * 3 + j] : 0.f;
            const _Float16 t0 = (_Float16)t, t1 = (_Float16)(t - (float)t0), t2 = (_Float16)(t - (float)t0 - (float)t1);
            sTp[j][0] = t0; sTp[j][1] = t1; sTp[j][2] = t2;
            for (int i = 3; i < 8; ++i) sTp[j][i] = (_Float16)0.f;
        }
        __syncthreads();
        constexpr int NGP = 2 * GA + 2;
        const int tile = f >> 4, row = f & 15;
        k2b_half* base = a.a2 + ((size_t)tile * 12 * NGP * 16 + row) * 8;
        for (int c = j; c < 12 * 2 * GA + 3; c += 64) {
            uint4 v;
            size_t grp;             // (entry, group) index
            if (c < 12 * 2 * GA) {
                const int e = c / (2 * GA), r = c % (2 * GA), lo = r >= GA, g = lo ? r - GA : r;
                v = *reinterpret_cast<const uint4*>((lo ? &sAl[e][0] : &sAh[e][0]) + 8 * g);
                grp = (size_t)e * NGP + (a.a2_stream_order ? (lo ? GA + 1 + g : g) : r);
            } else {
                const int r3 = c - 12 * 2 * GA;
                v = *reinterpret_cast<const uint4*>(&sTp[r3][0]);
                grp = (size_t)(4 * r3 + 3) * NGP + (a.a2_stream_order ? GA : 2 * GA);
            }
            *reinterpret_cast<uint4*>(base + grp * 16 * 8) = v;
        }
    }
    K2B_PSTAMP(7);
    if (!act) return;
    if (a.joints_out) {
        float* o = a.joints_out + ((size_t)f * a.num_out_joints + j) * 3;
        const float tx = a.tr ? a.tr[(size_t)f * 3] : 0.f, ty = a.tr ? a.tr[(size_t)f * 3 + 1] : 0.f,
                    tz = a.tr ? a.tr[(size_t)f * 3 + 2] : 0.f;
        o[0] = pg.x + tx; o[1] = pg.y + ty; o[2] = pg.z + tz;
    }
    K2B_PSTAMP_END;
}

constexpr int kFragHalfs = 512;      // one piece = 64 lanes x 8 halfs = 1 KiB

// ---------------------------------------------------------------------------------------------
// Tile kernel: workgroup = 8 waves = 128 frames x 128 vertices, persistent; v_mfma_f32_16x16x32_f16 throughout.
//
//   wave w: vertices [32 (w & 3), +32) (two 16-vertex tiles), frames [64 (w >> 2), +64) (four 16-frame tiles):
//   8 accumulator tiles of 16 x 16 per coordinate, 96 registers in the pose phase.
//
// Per tile the operands stream through a TWO-slot ring of 64 KiB (global_load_lds_dwordx4, lane-linear 1 KiB pieces) as a
// sequence of slices, one workgroup barrier per slice of 72 MFMAs per wave; the fills of slice q + 1 are issued at the top
// of slice q.  Workgroups are persistent: the slice sequence runs on across tile boundaries, so the next tile's first
// slice arrives under the present tile's last one and its stores.
//   pose slice = one 32-deep k-step:  X 4 x 32 frames x 2 k-halves x hi/lo (16 pieces) + Pd 4 x 32 vertices x 3 coords
//                                     x 2 k-halves x hi/lo (48 pieces): 72 MFMAs per wave (8 tiles x 3 coords x 3 products)
//   transform slice u (u-th 16-frame tile of every wave), EPS entries in d-major order (e = 4 r + d):
//                                     EPS x 2 frame tiles x NGP groups; EPS x NKT x 2 MFMAs per wave
//   out_r += T_e * v_posed_d  (d < 3)   |   out_r += T_e  (d = 3; the translation arrives through the PAD x ONES position)
// The transform phase works on ONE 16-frame tile at a time: 24 accumulator registers of outputs + 8 of T beside the 96 of
// v_posed (which die tile by tile), so the three coordinates of a (frame, vertex) pair meet in registers and leave as ONE
// 12-byte store - no LDS parking, no spills.  Bytes into LDS per 32 x 32 sub-tile: 41 KiB (128 x 64 kernel: 60 KiB).
// ---------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int kTileSlotBytes = 64 * 1024;
#ifndef K2B_TILE_AHEAD
#define K2B_TILE_AHEAD(NKT) ((NKT) <= 3)
#endif
#ifndef K2B_TILE_CHUNK
#define K2B_TILE_CHUNK 8      // frame groups per L2 chunk of the tile walk
#endif


__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the tiles of one workgroup: XCD label x = block % 8 owns a contiguous range of (frame chunk, vertex group) items,
// a chunk = 8 frame groups (1024 frames: their per-frame operands, 2.3 MB, stay in the XCD's L2 while the vertex
// groups stream past); inside the range the frame group runs fastest, the XCD's workgroups take every nx-th tile.
struct TileWalk {
    int vgroups, fgroups, item_lo, item_hi, nx;
    int t;                      // index into the XCD's tile sequence (item-major, 8 frame slots per item)
    int fg, vg;
    bool valid;
    __device__ void init(const TileArgs& a, int block, int nblocks) {
        vgroups = (a.v_tiles + 3) >> 2; fgroups = (a.f_tiles + 3) >> 2;
        const int items = ((fgroups + K2B_TILE_CHUNK - 1) / K2B_TILE_CHUNK) * vgroups, x = block & 7;
        item_lo = (int)((long long)items * x / 8); item_hi = (int)((long long)items * (x + 1) / 8);
        nx = nblocks >> 3;
        t = (block >> 3) - nx;
        next();
    }
    __device__ void next() {
        for (;;) {
            t += nx;
            const int item = item_lo + t / K2B_TILE_CHUNK;
            if (item >= item_hi) { valid = false; return; }
            const int c = item / vgroups;
            fg = c * K2B_TILE_CHUNK + t % K2B_TILE_CHUNK; vg = item - c * vgroups;
            if (fg < fgroups) { valid = true; return; }
        }
    }
};

__device__ __forceinline__ floatx4 tile_mfma(half8 x, half8 y, floatx4 c) { return K2B_DIAG_MFMA(x, y, c); }

template <int GA, int EPS>
__global__ __launch_bounds__(512) void k2b_lbs_tile_kernel(const TileArgs a) {
    constexpr int NGP = tile_ngp(GA);                 // 256-byte groups per (entry, 16-frame tile) and per 16-vertex tile of W
    constexpr int LSEQ = 3 * GA + 1;                  // k-groups of the concatenated contraction
    constexpr int NKT = (LSEQ + 3) / 4;               // MFMAs (32-deep k-steps) per entry and output tile
    constexpr int NTS = 12 / EPS;                     // transform slices per 16-frame tile
    constexpr int TP = EPS * 2 * (NGP / 4);           // pieces of a transform slice
    constexpr int WP = 8 * (NGP / 4);                 // pieces of the resident W image (8 tiles of 16 vertices)
    static_assert(12 % EPS == 0 && NGP % 4 == 0 && TP % 4 == 0 && WP % 4 == 0, "piece counts must divide over 8 waves");
    static_assert(TP * 1024 <= kTileSlotBytes, "transform slice does not fit a ring slot");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [2 slots][64 KiB] | W image [8][NGP][256 B]
    unsigned char* const wimg = lds + 2 * kTileSlotBytes;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vt = wave & 3, fpair = wave >> 2;
    const int g = lane >> 4, row = lane & 15;         // MFMA 16x16x32 operand lane: row of the tile, k-group of the step
    const int ftiles = a.f_tiles, vtiles = a.v_tiles, KX = a.k_steps_x >> 1;   // 32-deep k-steps (k_steps_x is even)
    const int f16tiles = ftiles * 2;

    // The per-lane LDS offsets are re-derived from the lane id at the start of each phase (behind an opaque copy, so
    // that they are not hoisted out of the tile loop): eight registers less alive across the other phase.
    auto opaque_lane = [&]() { int l = lane; asm volatile("" : "+v"(l)); return l; };
    const int lane8 = lane * 8;                       // halfs: this lane's 16 bytes of a 1 KiB piece

    // ---- roles -----------------------------------------------------------------------------------------------------
    // Waves w and w + 4 share a SIMD.  Everything that occupies a wave's instruction stream without feeding the matrix
    // pipe - issuing the LDS-DMA fills (~75 cycles each), issuing stores against a busy store path (hundreds of cycles
    // each) - is placed so that the SIMD partner is in its MFMA stream meanwhile (first version: all eight waves did
    // both at the same points of a slice, and the matrix pipe idled 40 % of every slice):
    //   loader waves 0-3:  top of slice: the stores of the PREVIOUS slice's outputs (held in 24 registers across the
    //                      barrier), then ALL fills of the next slice (16 / 12 pieces); MFMAs; counted wait; barrier
    //   waves 4-7:         MFMAs first (they start at once after the barrier); their stores at the end of the slice, when
    //                      the loaders are in their MFMAs; no fills, so they never wait on the vector-memory counter.
    const bool loader = wave < 4;

    // ---- loader: a cursor one slice ahead of the consumer (all quantities wave-uniform, i.e. scalar registers) --------
    // The source of every piece is  base pointer + 32-bit element offset + lane x 16 B.  What depends on the wave only is
    // worked out once, what depends on the tile once per tile; per slice a piece costs an add or two (the first version
    // redid the whole index arithmetic, divisions included, per piece: 900 cycles of scalar work per slice and wave).
    TileWalk lw;
    lw.init(a, blockIdx.x, a.num_wgs);
    int ls = 0;                   // slice of the loader's tile to issue next
    int lq = 0;                   // global slice counter of the loader (slot = lq & 1)
    const int spt = KX + 4 * NTS; // slices per tile
    const int lwv = wave & 3;     // loader wave index: piece p = lwv + 4 i
    const k2b_half* const xbase = (lwv & 1) ? a.xl : a.xh;        // hi / lo is the parity of the piece number
    const k2b_half* const pbase = (lwv & 1) ? a.pdl : a.pdh;
    // element offsets of this wave's 16 pose pieces at the loader's k-step: lane i of ONE vector register holds piece i
    // (read back with v_readlane; sixteen scalar registers instead pushed the kernel into scalar spills and s_load
    // re-materialisation, whose lgkmcnt traffic forces full drains of the LDS reads)
    unsigned vpoff = 0;
    const unsigned vstride = lane < 4 ? 2u * ftiles * kFragHalfs : 2u * 3u * vtiles * kFragHalfs;   // per 32-deep k-step
    auto issue = [&]() {
        if (!lw.valid) return;
        if (K2B_DIAG_SKIP_FILL(lq)) { ++lq; if (++ls == spt) { ls = 0; lw.next(); } return; }
        unsigned char* slot = lds + (lq & 1) * kTileSlotBytes;
        if (ls < KX) {
            if (ls == 0) {        // new tile: offsets of the pieces at k-step 0 (lane i: piece p = lwv + 4 i)
                const int ol = opaque_lane();             // (re-derived per tile: hoisted out of the tile loop these lane constants cost registers)
                const int p = lwv + 4 * (ol & 15);
                const int kh = p >> 3, ft = (p >> 1) & 3;                                            // X: [k-half][frame tile][hi/lo]
                const int i2 = p - 16, kh2 = i2 / 24, r24 = i2 - kh2 * 24, v4 = r24 / 6, c = (r24 - v4 * 6) >> 1;   // Pd: [k-half][vertex tile][coord][hi/lo]
                const int ftc = lw.fg * 4 + ft < ftiles ? lw.fg * 4 + ft : ftiles - 1;
                const int vtc = lw.vg * 4 + v4 < vtiles ? lw.vg * 4 + v4 : vtiles - 1;
                vpoff = ol < 4 ? (unsigned)(kh * ftiles + ftc) * kFragHalfs : (unsigned)((kh2 * 3 + c) * vtiles + vtc) * kFragHalfs;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)vpoff, i);
                const k2b_half* src = (i < 4 ? xbase : pbase) + o;
                __builtin_amdgcn_global_load_lds(src + lane8, slot + (lwv + 4 * i) * 1024, 16, 0, 0);
            }
            vpoff += vstride;
        } else {
            const int tsl = ls - KX, u = tsl / NTS, ts = tsl - u * NTS;
            unsigned fo[2];       // the u-th 16-frame tile of either wave pair
#pragma unroll
            for (int fsel = 0; fsel < 2; ++fsel) {
                int f16 = (lw.fg * 4 + 2 * fsel) * 2 + u;
                f16 = f16 < f16tiles ? f16 : f16tiles - 1;
                fo[fsel] = (unsigned)f16 * (12u * NGP * 128u);
            }
#pragma unroll
            for (int i = 0; i < TP / 4; ++i) {
                const int p = lwv + 4 * i;                               // piece: [entry in slice][frame tile of the slice][NGP / 4 pieces]
                const int ei = p / (2 * (NGP / 4)), rem = p - ei * (2 * (NGP / 4)), fsel = rem / (NGP / 4), pc = rem - fsel * (NGP / 4);
                const int nseq = ts * EPS + ei, d = nseq / 3, r = nseq - 3 * d, e = 4 * r + d;
                const unsigned o = (fsel ? fo[1] : fo[0]) + (unsigned)((e * NGP + 4 * pc) * 128);
                __builtin_amdgcn_global_load_lds(a.a2 + o + lane8, slot + p * 1024, 16, 0, 0);
            }
        }
        ++lq;
        if (++ls == spt) { ls = 0; lw.next(); }
    };

    // ---- consumer ------------------------------------------------------------------------------------------------
    TileWalk cw;
    cw.init(a, blockIdx.x, a.num_wgs);
    if (!cw.valid) return;        // whole workgroup: no tile
    int q = 0;                    // global slice counter of the consumer
    if (loader) issue();
    wait_vmcnt<0>();
    wg_barrier();

    auto rd = [&](const unsigned char* base, int off) -> half8 { return *reinterpret_cast<const half8*>(base + off); };
    const float inv_scale = 1.0f / kPdScale;
    float* const dump = a.dump + lane * 3;
    K2B_DIAG_STAMP_DECL;
    auto stamp = [&](int slice, int k) { K2B_DIAG_STAMP(slice, k); };

    // one 12-byte store per (frame, vertex) of a 16-frame unit: frame fbase0 + 4 g + i, 16-vertex tile v
    auto emit_stores = [&](const floatx4 (&o)[2][3], int fbase0, int vg) {
        const int v0 = (vg * 4 + vt) * 32 + row;         // this lane's vertex in 16-vertex tile 0 (tile 1: + 16)
        float* const orow = a.out + ((size_t)a.out_row0 + v0) * 3;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int f = fbase0 + 4 * g + i;
                bool ok = f < a.num_frames && v0 + 16 * v < a.num_out;
                K2B_DIAG_STORE(f, ok);
                float3v x;
                x.x = o[v][0][i]; x.y = o[v][1][i]; x.z = o[v][2][i];
                float* dst = ok ? orow + ((size_t)f * a.out_stride + 16 * v) * 3 : dump;
                *reinterpret_cast<float3v*>(dst) = x;
            }
    };
    // Every wave stores a unit's outputs at the end of the unit.  (Round 2 let the loader waves hold theirs across the barrier and
    // store them behind the next slice's fills; with the joint copies in the kernel those 24 registers push both instantiations
    // into scratch - the SMPL-X one spilled 25 registers at the tile boundaries - and without the hold the launch is 3-7 % faster
    // for SMPL and 7 % for SMPL-X: round 3 measurements, tools/build_lbs_variants.sh hold:"-DK2B_TILE_HOLD=1" no longer exists.)
    bool stored = false;          // loader waves: eight stores of this slice are in flight behind its fills
    auto slice_top = [&]() {      // loader waves only: the fills of the next slice
        issue();
        stored = false;
    };
    auto loader_wait = [&]() {    // the fills have landed; the eight stores behind them may fly on (joint copies are younger still:
                                  // "at most 8 outstanding" keeps meaning "every fill has landed")
        if (stored) wait_vmcnt<8>(); else wait_vmcnt<0>();
    };

    while (cw.valid) {
        floatx4 vp[4][2][3];      // [16-frame tile][16-vertex tile][coordinate]
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c) vp[f][v][c] = floatx4{0.f, 0.f, 0.f, 0.f};

        // ---- pose phase: v_posed * kPdScale = X . Pd ---------------------------------------------------------
        // lane offsets inside a pose slot: piece (k-half, ...) = two 512-byte groups over 32 rows ([group][row32][8])
        const int pl_ = opaque_lane(), pg_ = pl_ >> 4, pr_ = pl_ & 15;
        const int lx = (pg_ >> 1) * (8 * 1024) + (pg_ & 1) * 512 + pr_ * 16;     // X pieces:  [k-half][frame tile 4][hi/lo]
        const int lp = (pg_ >> 1) * (24 * 1024) + (pg_ & 1) * 512 + pr_ * 16;    // Pd pieces: 16 KiB + [k-half][vertex tile 4][coord 3][hi/lo]
        for (int ks = 0; ks < KX; ++ks) {
            stamp(ks, 0);
            if (loader) {
                if (ks == 0) {    // resident W image of this tile's vertex group (single buffer: every read of the previous
                                  // tile's image lies before the barrier that ended its last slice)
#pragma unroll
                    for (int i = 0; i < WP / 4; ++i) {
                        const int p = lwv + 4 * i, v16 = p / (NGP / 4), pc = p - v16 * (NGP / 4);
                        int vt16 = cw.vg * 8 + v16;
                        vt16 = vt16 < 2 * vtiles ? vt16 : 2 * vtiles - 1;
                        const k2b_half* src = a.w2 + ((size_t)vt16 * NGP + 4 * pc) * 128;
                        __builtin_amdgcn_global_load_lds(src + lane8, wimg + p * 1024, 16, 0, 0);
                    }
                }
                slice_top();
            }
            stamp(ks, 1);
            const unsigned char* slot = lds + (q & 1) * kTileSlotBytes;
            const unsigned char* xb = slot + lx, *pb = slot + 16 * 1024 + lp;
            half8 xf[4][2];       // [16-frame tile][hi/lo]
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    xf[f][hl] = rd(xb, ((2 * fpair + (f >> 1)) * 2 + hl) * 1024 + (f & 1) * 256);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                half8 pf[2][2];   // [16-vertex tile][hi/lo]
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) pf[v][hl] = rd(pb, (vt * 6 + c * 2 + hl) * 1024 + v * 256);
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        vp[f][v][c] = tile_mfma(xf[f][0], pf[v][0], vp[f][v][c]);
                        vp[f][v][c] = tile_mfma(xf[f][0], pf[v][1], vp[f][v][c]);
                        vp[f][v][c] = tile_mfma(xf[f][1], pf[v][0], vp[f][v][c]);
                    }
            }
            stamp(ks, 2);
            if (loader) loader_wait();        // the next slice (and, in the first slice, the W image) has landed
            stamp(ks, 4);
            wg_barrier();
            stamp(ks, 5);
            ++q;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int c = 0; c < 3; ++c) vp[f][v][c] *= inv_scale;

        // ---- transform phase: one 16-frame tile (u) at a time --------------------------------------------------------
        // the W fragments of this wave's two 16-vertex tiles stay in registers for the whole phase: re-read per entry they
        // made the transform slices LDS-bandwidth-bound (108 KiB per wave and slice; with them resident 36 KiB)
        // group offsets (inside an entry / a vertex tile) of this lane for the k-steps of the transform GEMM
        int offA[NKT], offW[NKT];
        {
            const int tl_ = opaque_lane(), tg_ = tl_ >> 4, tr_ = tl_ & 15;
#pragma unroll
            for (int i = 0; i < NKT; ++i) {
                auto seqA = [](int p) { return p < GA ? p : p < 2 * GA ? p - GA : p < 3 * GA ? p - GA : p == 3 * GA ? 2 * GA : 2 * GA + 1; };
                auto seqW = [](int p) { return p < GA ? p : p < 2 * GA ? p : p < 3 * GA ? p - 2 * GA : p == 3 * GA ? 2 * GA : 2 * GA + 1; };
                const int pa = tg_ == 0 ? seqA(4 * i) : tg_ == 1 ? seqA(4 * i + 1) : tg_ == 2 ? seqA(4 * i + 2) : seqA(4 * i + 3);
                const int pw = tg_ == 0 ? seqW(4 * i) : tg_ == 1 ? seqW(4 * i + 1) : tg_ == 2 ? seqW(4 * i + 2) : seqW(4 * i + 3);
                offA[i] = pa * 256 + tr_ * 16;
                offW[i] = pw * 256 + tr_ * 16;
            }
        }
        const unsigned char* wb = wimg + (2 * vt) * NGP * 256;
        half8 wf[2][NKT];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int k = 0; k < NKT; ++k) wf[v][k] = rd(wb, v * NGP * 256 + offW[k]);
        // vertex-selected extra joints ride in the W image: first half of the (otherwise all-zero) padding group of a vertex's row =
        // 1 + index of the output joint that IS this vertex (0: none).  It multiplies zeros of A in the GEMM.  A wave whose 32
        // vertices hold such a vertex stores that vertex a second time, into the joints array (rare: 21 of 6890 for SMPL) -
        // which replaces the gather launch that used to follow this kernel.
        int jrow[2];
        bool has_joint = false;
        if (a.joints_out) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                jrow[v] = (int)(float)*reinterpret_cast<const _Float16*>(wb + v * NGP * 256 + (2 * GA + 1) * 256 + (opaque_lane() & 15) * 16);
            }
            has_joint = __builtin_amdgcn_ballot_w64(jrow[0] != 0 || jrow[1] != 0) != 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            floatx4 out[2][3];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int r = 0; r < 3; ++r) out[v][r] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ts = 0; ts < NTS; ++ts) {
                stamp(KX + u * NTS + ts, 0);
                if (loader) slice_top();
                stamp(KX + u * NTS + ts, 1);
                const unsigned char* slot = lds + (q & 1) * kTileSlotBytes;
                // software pipeline over the entries: the A fragments of entry n + 1 are requested before the MFMAs of
                // entry n, and entry n - 1 is folded into the outputs while the matrix pipe works on entry n (left to the
                // compiler, every entry paid an LDS round trip and an MFMA drain: 2.6-4.2 k cycles for 72 MFMAs)
                // (large trees: 2 x NKT fragment registers of look-ahead do not fit beside the resident W fragments - the SMPL-X
                //  instantiation spilled 24 registers, and every scratch reload is a vmcnt(0) wait in the middle of the request
                //  stream - so there the fragments of an entry are read right before its MFMAs; the SIMD partner covers the wait)
                constexpr bool AHEAD = K2B_TILE_AHEAD(NKT);
                half8 af[AHEAD ? 2 : 1][NKT];
                floatx4 t[2][2];
                if (AHEAD) {
#pragma unroll
                    for (int k = 0; k < NKT; ++k) af[0][k] = rd(slot + fpair * NGP * 256, offA[k]);
                }
#pragma unroll
                for (int ei = 0; ei <= EPS; ++ei) {
                    if (AHEAD && ei + 1 < EPS) {
#pragma unroll
                        for (int k = 0; k < NKT; ++k) af[(ei + 1) & 1][k] = rd(slot + ((ei + 1) * 2 + fpair) * NGP * 256, offA[k]);
                    }
                    if (!AHEAD && ei < EPS) {
#pragma unroll
                        for (int k = 0; k < NKT; ++k) af[0][k] = rd(slot + (ei * 2 + fpair) * NGP * 256, offA[k]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (ei < EPS) {
#pragma unroll
                        for (int v = 0; v < 2; ++v) t[ei & 1][v] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < NKT; ++k)
#pragma unroll
                            for (int v = 0; v < 2; ++v) t[ei & 1][v] = tile_mfma(af[AHEAD ? (ei & 1) : 0][k], wf[v][k], t[ei & 1][v]);
                    }
                    if (ei > 0) {
                        const int nseq = ts * EPS + ei - 1, d = nseq / 3, r = nseq % 3;
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            if (d < 3) out[v][r] += t[(ei - 1) & 1][v] * vp[u][v][d];
                            else out[v][r] += t[(ei - 1) & 1][v];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                stamp(KX + u * NTS + ts, 2);
                if (ts == NTS - 1) {
                    const int fbase0 = (cw.fg * 4 + 2 * fpair) * 32 + u * 16;
                    if (has_joint) {
#pragma unroll
                        for (int v = 0; v < 2; ++v)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int f = fbase0 + 4 * g + i;
                                if (jrow[v] != 0 && f < a.num_frames) {
                                    float3v x;
                                    x.x = out[v][0][i]; x.y = out[v][1][i]; x.z = out[v][2][i];
                                    *reinterpret_cast<float3v*>(a.joints_out + ((size_t)f * a.joints_stride + a.joints_row0 + jrow[v] - 1) * 3) = x;
                                }
                            }
                    }
                    emit_stores(out, fbase0, cw.vg);
                    stored = true;               // (loaders: these eight stores are younger than the slice's fills)
                }
                stamp(KX + u * NTS + ts, 3);
                if (loader) loader_wait();
                stamp(KX + u * NTS + ts, 4);
                wg_barrier();
                stamp(KX + u * NTS + ts, 5);
                ++q;
            }
        }
        cw.next();
        K2B_DIAG_TILE_DONE;
    }
    wait_vmcnt<0>();
    K2B_DIAG_KERNEL_END;
}

// joints J..J+E-1 := vertices[extra ids]: only when an output joint's vertex cannot ride in the W image (two joints on one vertex)
__global__ void k2b_gather_joints_kernel(const float* __restrict__ verts, const int* __restrict__ ids, float* joints,
                                         int num_frames, int V, int J, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_frames * E) return;
    const int f = i / E, e = i % E;
    const float* s = verts + ((size_t)f * V + ids[e]) * 3;
    float* d = joints + ((size_t)f * (J + E) + J + e) * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

int lbs_frames_padded(int num_frames) { return (num_frames + 31) / 32 * 32; }

hipError_t launch_pose_setup(const PoseArgs& a_in, hipStream_t stream) {
    if (a_in.num_frames <= 0) return hipSuccess;
    if (a_in.k_steps_x > kMaxXSteps || a_in.num_joints > kMaxJoints || a_in.num_betas > kMaxShape) return hipErrorInvalidValue;
    const int GA = tile_groups_a(a_in.num_joints);
    PoseArgs a = a_in;
#ifndef K2B_POSE_XCD
#define K2B_POSE_XCD 1
#endif
    a.xcd_frames = (K2B_POSE_XCD && a.num_frames > 256) ? (a.num_frames + 255) / 256 * 32 : 0;
    const dim3 grid(a.xcd_frames ? 8 * a.xcd_frames : a.num_frames), block(64);
#define K2B_POSE(NBC_, GA_) hipLaunchKernelGGL((k2b_pose_setup_kernel<NBC_, GA_>), grid, block, 0, stream, a)
    if (GA == 3) { if (a.num_betas <= 10) K2B_POSE(10, 3); else if (a.num_betas <= 16) K2B_POSE(16, 3); else K2B_POSE(32, 3); }
    else if (GA == 7) { if (a.num_betas <= 10) K2B_POSE(10, 7); else if (a.num_betas <= 20) K2B_POSE(20, 7); else K2B_POSE(32, 7); }
    else return hipErrorInvalidValue;                      // 17..24 joints (SMPL) or 49..56 (SMPL-H / SMPL-X)
#undef K2B_POSE
    return hipGetLastError();
}

hipError_t launch_skin_tiles(const TileArgs& a_in, int num_cus, hipStream_t stream) {
    if (a_in.num_frames <= 0 || a_in.num_out <= 0) return hipSuccess;
    TileArgs a = a_in;
    const int vgroups = (a.v_tiles + 3) / 4, fgroups = (a.f_tiles + 3) / 4;
    const long long tiles = (long long)vgroups * fgroups;
    int wgs = num_cus < 8 ? 8 : num_cus / 8 * 8;           // one persistent workgroup per CU, a multiple of the 8 XCD labels
    if (tiles < wgs) wgs = (int)((tiles + 7) / 8 * 8);
    a.num_wgs = wgs;
    const int GA = a.groups_a;
    const size_t lds = (size_t)2 * kTileSlotBytes + (size_t)8 * tile_ngp(GA) * 256;
    if (a.k_steps_x & 1) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
#define K2B_TILE(GA_, EPS_)                                                                                          \
    do {                                                                                                             \
        static std::atomic<unsigned long long> lds_set{0};                                                           \
        e = ensure_dynamic_lds(k2b_lbs_tile_kernel<GA_, EPS_>, lds_set, lds);                                        \
        if (e != hipSuccess) return e;                                                                               \
        hipLaunchKernelGGL((k2b_lbs_tile_kernel<GA_, EPS_>), dim3(wgs), dim3(512), lds, stream, a);                  \
    } while (0)
    if (GA == 3) K2B_TILE(3, 12);
    else if (GA == 7) K2B_TILE(7, 6);
    else return hipErrorInvalidValue;                      // 17..24 joints (SMPL) or 49..56 (SMPL-X)
#undef K2B_TILE
    return hipGetLastError();
}

hipError_t launch_gather_joints(const float* verts, const int* ids, float* joints, int num_frames, int V, int J, int E,
                                hipStream_t stream) {
    if (num_frames <= 0 || E <= 0) return hipSuccess;
    const int n = num_frames * E;
    hipLaunchKernelGGL(k2b_gather_joints_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, verts, ids, joints,
                       num_frames, V, J, E);
    return hipGetLastError();
}

}  // namespace k2b
