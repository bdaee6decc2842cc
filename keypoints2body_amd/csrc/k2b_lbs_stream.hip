// k2b_lbs_stream.hip — vertex skinning for 17-24 joint models (SMPL), round-3 design for gfx950.
//
// Same arithmetic as the tile kernel of k2b_lbs.hip (two GEMMs over frames x vertices on v_mfma_f32_16x16x32_f16, operands
// as f16 hi/lo pairs, v = T [v_posed; 1] + t; reference seam: the final forward of world_space.py:258-278, smplx's
// SMPL.forward), and the same 128 frames x 128 vertices per persistent workgroup.  What changed is how the operands travel:
//
//   * a wave owns 16 vertices x 128 frames (eight 16 x 16 accumulator tiles per coordinate).  Its share of the big
//     B operand Pd (posedirs | shapedirs | template, 48 of the 64 KiB a 32-deep k-step needs) is private to it, so it goes
//     global -> REGISTERS with plain 16-byte loads, two k-steps ahead, and never touches LDS or a barrier;
//   * the frame-side operand X of the WHOLE tile (7 k-steps x 16 KiB) is resident in LDS: the pose phase has NO barrier at all,
//     waves drift apart and the stores of one wave overlap the matrix work of another;
//   * X of the next tile is fetched by LDS-DMA during the transform phase, one k-step per 16-frame unit, the transform
//     operand A in a two-slot ring of 24 KiB units: one workgroup barrier per unit, 8 per tile (tile kernel: 11, and 64 KiB
//     of fills behind each);
//   * per entry of the 3 x 4 transform the two A fragments [hi | t] and [lo | 0] are read ONCE and meet three resident W
//     fragments [hi | 1], [hi | tag], [lo | 0]: two LDS reads for three MFMAs.
// Every vector-memory wait is a counted s_waitcnt with a compile-time count (the issue order of a wave is fixed; loads, LDS-DMA
// and stores retire in order), so the loads the wave does not need yet stay in flight.
//
// LDS: 7 x 16 KiB (X) + 2 x 24 KiB (A units) = 160 KiB.  Built for 7 k-steps (9 (J - 1) + NB + 2 <= 224: SMPL with up to 15
// shape coefficients); everything else runs the tile kernel.
#include <hip/hip_fp16.h>

#include <type_traits>

#include "k2b_internal.h"

namespace k2b {

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) float3v { float x, y, z; };

constexpr int SKX = kStreamKSteps;                 // 32-deep k-steps of the pose GEMM
constexpr int kXBytes = SKX * 16 * 1024;
constexpr int kUnitBytes = 24 * 1024;              // A operand of one 16-frame unit: 12 entries x 2 fragments
// diagnostics live in tools/lbs_diag.h (tools/build_lbs_variants.sh); the product build leaves the hooks empty
#ifdef K2B_LBS_DIAG_HEADER
#include K2B_LBS_DIAG_HEADER
#endif
#ifndef K2B_SDIAG_BEGIN
#define K2B_SDIAG_BEGIN ((void)0)
#define K2B_SDIAG_END ((void)0)
#endif
#ifndef K2B_SDIAG_STAMP
#define K2B_SDIAG_STAMP(i) ((void)0)
#define K2B_SDIAG_TILE ((void)0)
#endif
#ifndef K2B_SX_SKIP
#define K2B_SX_SKIP 0          // timing-only builds of the SMPL-X kernel (tools/build_lbs_variants.sh name:"-DK2B_SX_SKIP=n"), a bit mask:
#endif                         // 1 no stores, 2 no fills behind the prologue, 4 no Pd loads in the loop, 8 no barriers in the loop,
                               // 16 no LDS reads, 32 Pd of vertex group 0 for every tile (always L2-resident)
                               // (results are wrong on purpose; they answer "what does this part cost")
#ifndef K2B_SXDIAG_STAMP
#define K2B_SXDIAG_BEGIN ((void)0)
#define K2B_SXDIAG_STAMP(i) ((void)0)
#define K2B_SXDIAG_TILE ((void)0)
#define K2B_SXDIAG_STORES 1
#endif
constexpr int CHUNK = 8;                           // frame groups per L2 chunk of the tile walk (as the tile kernel)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// 16-byte load global -> register through a scalar base and a per-lane byte offset; the result is only valid behind a
// counted wait that names the register (pd_ready / below)
template <int OFF>
__device__ __forceinline__ void gload16(half8& dst, unsigned lane_off, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(lane_off), "s"(sbase), "n"(OFF) : "memory");
}

// 16-byte LDS read whose completion the code waits for itself (counted lgkmcnt waits that name the registers): the
// compiler's own waits are lgkmcnt(0) in front of the first use, which stalls a wave on every fragment it has just requested
template <int OFF>
__device__ __forceinline__ void lread16(half8& dst, unsigned addr) {
    if constexpr (K2B_SX_SKIP & 16) asm volatile("" : "=v"(dst) : "v"(addr));
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// 12-byte store through a scalar row base and a per-lane 32-bit byte offset (the compiler's own code adds 64-bit vector addresses
// per store; nobody waits for a store except the end of the kernel)
typedef float float3r __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void gstore12(unsigned lane_off, float3r d, const void* sbase) {
    asm volatile("global_store_dwordx3 %0, %1, %2" ::"v"(lane_off), "v"(d), "s"(sbase) : "memory");
}
#define K2B_LDS_READY2(N, b) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory")

struct Walk {                  // (frame group, vertex group) tiles of this workgroup, XCD-aware: see TileWalk in k2b_lbs.hip
    int vgroups, fgroups, item_lo, item_hi, nx;
    __device__ void init(int vgroups_, int fgroups_, int block, int nblocks) {
        vgroups = vgroups_; fgroups = fgroups_;
        const int items = ((fgroups + CHUNK - 1) / CHUNK) * vgroups, x = block & 7;
        item_lo = (int)((long long)items * x / 8); item_hi = (int)((long long)items * (x + 1) / 8);
        nx = nblocks >> 3;
    }
    // the tile behind sequence index t (advanced to the next valid one), or fg = -1 when the workgroup's sequence is exhausted
    __device__ void next(int& t, int& fg, int& vg) const {
        for (;;) {
            t += nx;
            const int item = item_lo + t / CHUNK;
            if (item >= item_hi) { fg = -1; vg = 0; return; }
            const int c = item / vgroups;
            fg = c * CHUNK + t % CHUNK; vg = item - c * vgroups;
            if (fg < fgroups) return;
        }
    }
};

}  // namespace

__global__ __launch_bounds__(512) void k2b_lbs_stream_kernel(const StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [7][16 KiB] X | [2][24 KiB] A units
    unsigned char* const aslots = lds + kXBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = lane & 15, g = lane >> 4;                 // MFMA operand lane: row of the 16-row tile, k-group
    const int f32tiles = a.f32_tiles, f16tiles = 2 * f32tiles, nv16 = a.nv16;
    const unsigned lane16 = (unsigned)lane * 16u;

    Walk walk;
    walk.init(nv16 >> 3, (f32tiles + 3) >> 2, blockIdx.x, a.num_wgs);
    int wt = (blockIdx.x >> 3) - walk.nx, cfg, cvg;          // sequence index, frame group and vertex group of the current tile
    walk.next(wt, cfg, cvg);
    if (cfg < 0) return;
    K2B_SDIAG_BEGIN;

    // ---- issue helpers (all addresses wave-uniform + lane x 16 B) ----------------------------------------------------------
    // X of k-step ks: 16 pieces [k-half 2][32-frame tile 4][hi | lo]; wave w moves (k-half w >> 2, frame tile w & 3), hi and lo
    auto issue_x = [&](int fg, int ks) {
        const int kh = wave >> 2, ft = wave & 3;
        int ftc = fg * 4 + ft;
        ftc = ftc < f32tiles ? ftc : f32tiles - 1;
        const size_t o = ((size_t)(2 * ks + kh) * f32tiles + ftc) * 512 + lane * 8;
        unsigned char* dst = lds + ks * 16384 + (kh * 8 + ft * 2) * 1024;
        __builtin_amdgcn_global_load_lds(a.xh + o, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a.xl + o, dst + 1024, 16, 0, 0);
    };
    // A of one 16-frame unit: 24 contiguous pieces [entry 12][fragment 2]; wave w moves pieces 3 w .. 3 w + 2
    auto issue_a = [&](int f16, int slot) {
        f16 = f16 < f16tiles ? f16 : f16tiles - 1;
        const k2b_half* src = a.a2 + ((size_t)f16 * 24 + 3 * wave) * 512 + lane * 8;
        unsigned char* dst = aslots + slot * kUnitBytes + 3 * wave * 1024;
#pragma unroll
        for (int i = 0; i < 3; ++i) __builtin_amdgcn_global_load_lds(src + i * 512, dst + i * 1024, 16, 0, 0);
    };
    // Pd of this wave's 16 vertices for k-step ks: 6 consecutive KiB [coordinate 3][hi | lo]
    auto load_pd = [&](half8 (&buf)[3][2], int vg, int ks) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.pd + ((size_t)ks * nv16 + ((K2B_SX_SKIP & 32) ? 0 : vg) * 8 + wave) * 6 * 512) + 3072;
        gload16<-3072>(buf[0][0], lane16, base); gload16<-2048>(buf[0][1], lane16, base);
        gload16<-1024>(buf[1][0], lane16, base); gload16<0>(buf[1][1], lane16, base);
        gload16<1024>(buf[2][0], lane16, base);  gload16<2048>(buf[2][1], lane16, base);
    };
#define K2B_PD_READY(N, b)                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]) : "n"(N) : "memory")

    const int lx = (g >> 1) * 8192 + (g & 1) * 512 + row * 16;      // lane part of an X fragment address inside a k-step
    const unsigned lds0 = (unsigned)(uintptr_t)lds;                  // LDS byte address of the X region
    const unsigned lxa = lds0 + lx, lxb = lxa + 65536;               // (the 16-bit offset field reaches four k-steps)
    const int la = g * 256 + row * 16;                               // lane part of an A fragment address inside a piece
    const float inv_scale = 1.0f / kPdScale;

    // ---- prologue: everything the first tile needs ------------------------------------------------------------------------
    half8 pb0[3][2], pb1[3][2], pb2[3][2];      // Pd buffers: k-step ks lives in buffer ks % 3
    half8 wf[3];                                // W fragments of this wave's 16 vertices: [hi | 1], [hi | tag], [lo | 0]
#pragma unroll
    for (int ks = 0; ks < SKX; ++ks) issue_x(cfg, ks);            // 14 fills
    issue_a(cfg * 8, 0);                                          // 3 fills
    load_pd(pb0, cvg, 0);                                         // 6 loads
    load_pd(pb1, cvg, 1);                                         // 6 loads
    K2B_PD_READY(0, pb0);                                           // everything of the prologue has landed (once per launch)
    K2B_PD_READY(0, pb1);
    wg_barrier();

    while (cfg >= 0) {
        int nt = wt, nxf, nxv;
        walk.next(nt, nxf, nxv);
        const int nfg = nxf >= 0 ? nxf : cfg, nvg = nxf >= 0 ? nxv : cvg;   // (no next tile: the same addresses again, so that the
                                                                             //  counted waits keep their counts)
        K2B_SDIAG_STAMP(0);
        floatx4 vp[8][3];         // [16-frame tile][coordinate]; the first k-step starts every accumulator from zero

        // ---- pose phase: v_posed * kPdScale = X . Pd, no barrier -------------------------------------------------------------
        // X fragments (hi | lo of one 16-frame tile), two buffers: tile q + 1 of the phase's 56 is requested before the nine MFMAs of
        // tile q, and a counted wait leaves those two reads in flight
        half8 xq[2][2];
        auto xread = [&](half8 (&dst)[2], auto ksc, auto fc) {
            constexpr int ks = decltype(ksc)::value, f = decltype(fc)::value;
            constexpr int off = (ks & 3) * 16384 + (f >> 1) * 2048 + (f & 1) * 256;
            const unsigned base = ks < 4 ? lxa : lxb;
            lread16<off>(dst[0], base); lread16<off + 1024>(dst[1], base);
        };
        auto kstep = [&](auto ksc, const half8 (&pd)[3][2]) {
            constexpr int ks = decltype(ksc)::value;
            auto tile = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                half8 (&cur)[2] = xq[f & 1];
                if constexpr (f < 7) xread(xq[(f + 1) & 1], ksc, std::integral_constant<int, (f + 1) & 7>{});
                else if constexpr (ks < SKX - 1) xread(xq[0], std::integral_constant<int, (ks + 1) % SKX>{}, std::integral_constant<int, 0>{});
                if constexpr (f < 7 || ks < SKX - 1) K2B_LDS_READY2(2, cur); else K2B_LDS_READY2(0, cur);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (ks == 0) vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][0], vp[f][c], 0, 0, 0);
                    vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][1], vp[f][c], 0, 0, 0);
                    vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[1], pd[c][0], vp[f][c], 0, 0, 0);
                }
            };
            tile(std::integral_constant<int, 0>{}); tile(std::integral_constant<int, 1>{}); tile(std::integral_constant<int, 2>{});
            tile(std::integral_constant<int, 3>{}); tile(std::integral_constant<int, 4>{}); tile(std::integral_constant<int, 5>{});
            tile(std::integral_constant<int, 6>{}); tile(std::integral_constant<int, 7>{});
        };
        xread(xq[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        // younger than the awaited loads at each wait (in issue order): see the table in DESIGN.md 4.2
        // k-steps 0 and 1 were loaded during the previous tile's pose phase, in front of A(unit 1): the wait of the previous tile's
        // unit 1 has covered them already (whether or not that tile stored), so nothing is left to wait for here - vmcnt(63), the
        // counter's ceiling, only ties the registers to this point and does NOT wait for the transform phase's last stores, whose
        // acknowledgements take thousands of cycles when every CU writes at once (stores and loads share the counter and retire
        // in order).  (First tile: the prologue waited for everything.)
        K2B_PD_READY(63, pb0);
        load_pd(pb2, cvg, 2);                        // Pd two k-steps ahead
        kstep(std::integral_constant<int, 0>{}, pb0);
        K2B_SDIAG_STAMP(1);
        K2B_PD_READY(63, pb1);                         // (likewise)
        load_pd(pb0, cvg, 3);
        kstep(std::integral_constant<int, 1>{}, pb1);
        K2B_SDIAG_STAMP(2);
        K2B_PD_READY(6, pb2);
        load_pd(pb1, cvg, 4);
        kstep(std::integral_constant<int, 2>{}, pb2);
        K2B_SDIAG_STAMP(3);
        K2B_PD_READY(6, pb0);
        load_pd(pb2, cvg, 5);
        kstep(std::integral_constant<int, 3>{}, pb0);
        K2B_SDIAG_STAMP(4);
        K2B_PD_READY(6, pb1);
        load_pd(pb0, cvg, 6);
        kstep(std::integral_constant<int, 4>{}, pb1);
        K2B_SDIAG_STAMP(5);
        K2B_PD_READY(6, pb2);
        load_pd(pb1, nvg, 0);                          // the next tile's first two k-steps ride through the transform phase
        kstep(std::integral_constant<int, 5>{}, pb2);
        K2B_SDIAG_STAMP(6);
        K2B_PD_READY(6, pb0);
        {                                              // W fragments of this tile's vertices (needed behind the pose phase: 12 registers less until here)
            const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w + ((size_t)cvg * 8 + wave) * 3 * 512);
            gload16<0>(wf[0], lane16, wbase); gload16<1024>(wf[1], lane16, wbase); gload16<2048>(wf[2], lane16, wbase);
        }
        load_pd(pb2, nvg, 1);
        kstep(std::integral_constant<int, 6>{}, pb0);
        K2B_SDIAG_STAMP(7);
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { vp[f][c][i] *= inv_scale; asm volatile("" : "+v"(vp[f][c][i])); }

        // ---- transform phase: one 16-frame unit at a time ---------------------------------------------------------------------
        // this wave's vertex, and the output joint it may be (tag = 1 + index, first half of the padding group of fragment 1)
        const int v = (cvg * 8 + wave) * 16 + row;
        const bool okv = v < a.num_out;
        int jrow = 0;
        bool has_joint = false;
        const size_t row_bytes = (size_t)a.out_stride * 12;                     // one frame of the output
        // a store address = wave-uniform row base (tile's first frame + u 16 + i, scalar arithmetic) + this lane's 32-bit offset
        // (its vertex, and the frames 4 g .. of its k-group): global_store with a scalar base, no 64-bit vector arithmetic per store
        unsigned char* const tbase = reinterpret_cast<unsigned char*>(a.out) + (size_t)(cfg * 128) * row_bytes;
        const unsigned voff = (unsigned)(((size_t)a.out_row0 + v) * 12 + (size_t)(4 * g) * row_bytes);
        const bool tile_full = cfg * 128 + 127 < a.num_frames && (cvg * 8 + wave) * 16 + 15 < a.num_out;   // wave-uniform
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // A of this unit has landed (each wave waits for its own pieces, the barrier collects them); what may stay in flight is
            // younger: the previous unit's 2 X fills + 4 stores (unit 0: the 6 Pd loads of the next tile's k-step 1; unit 7: stores only,
            // because the last X fills must be visible to every wave before the next pose phase)
            K2B_SDIAG_STAMP(8 + 3 * u);
            // The stores of a unit are part of the count ONLY where every one of them is issued: in the predicated path a wave whose
            // lanes are all beyond the mesh or the batch skips the instruction (s_cbranch_execz), so a wave of a partial tile counts
            // the fills alone (tile_full is wave-uniform: a scalar branch).  Fewer younger operations than N would leave the awaited
            // fills in flight; more (the rare joint copies) only make the wait stricter.
            if (u == 0) asm volatile("s_waitcnt vmcnt(6)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2])::"memory");   // (W: loaded at k-step 6, in front of the next tile's k-step 1)
            else if (u == 7) { if (tile_full) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
            else { if (tile_full) wait_vmcnt<6>(); else wait_vmcnt<2>(); }
            wg_barrier();
            K2B_SDIAG_STAMP(9 + 3 * u);
            if (u == 0 && a.joints_out) {              // (W fragments are long there: they are older than everything waited for)
                const float tg = (float)wf[1][0];      // lanes g == 3 hold the padding group of their row
                jrow = (int)__shfl(tg, 48 + row, 64);
                has_joint = __builtin_amdgcn_ballot_w64(jrow != 0) != 0;
            }
            // behind the barrier the other slot and (after unit 0) the X region are free: next unit's A, next tile's X k-step u
            if (u < 7) { issue_a(cfg * 8 + u + 1, (u + 1) & 1); issue_x(nfg, u); }
            else issue_a(nfg * 8, 0);
            floatx4 out[3] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
            // entries in d-major order (n -> d = n / 3, r = n % 3, entry 4 r + d); fragments of entry n + 1 requested before the
            // MFMAs of entry n, entry n - 1 folded into the outputs while the matrix pipe works on entry n
            half8 af[2][2];
            floatx4 t[2];
            const unsigned sa = lds0 + kXBytes + (u & 1) * kUnitBytes + la;
            lread16<0>(af[0][0], sa); lread16<1024>(af[0][1], sa);
            auto entry = [&](auto nc) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n + 1 < 12) {
                    constexpr int e1 = 4 * ((n + 1) % 3) + (n + 1) / 3;
                    lread16<e1 * 2048>(af[(n + 1) & 1][0], sa); lread16<e1 * 2048 + 1024>(af[(n + 1) & 1][1], sa);
                }
                if constexpr (n + 1 < 12) K2B_LDS_READY2(2, af[n & 1]); else if constexpr (n < 12) K2B_LDS_READY2(0, af[n & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (n < 12) {
                    floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[n & 1][0], wf[0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[n & 1][1], wf[1], acc, 0, 0, 0);
                    t[n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[n & 1][0], wf[2], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);     // the fold of entry n - 1 behind all three MFMAs of entry n: no hazard no-ops on t
                if constexpr (n > 0) {
                    constexpr int d = (n - 1) / 3, r = (n - 1) % 3;
                    // element by element: written on the 4-vectors this becomes v_pk_fma_f32 / v_pk_add_f32, which issue at well under
                    // half the rate of the scalar forms beside MFMAs (MI355X_MICROARCH.md) - and the transform phase is VALU-issue-bound
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (d < 3) out[r][i] = __builtin_fmaf(t[(n - 1) & 1][i], vp[u][d][i], out[r][i]);
                        else out[r][i] += t[(n - 1) & 1][i];
                        asm volatile("" : "+v"(out[r][i]));   // fold NOW (left to itself the compiler keeps all twelve T tiles, 48 registers, for
                    }                                         // the end) and do not re-pack
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            entry(std::integral_constant<int, 0>{}); entry(std::integral_constant<int, 1>{}); entry(std::integral_constant<int, 2>{});
            entry(std::integral_constant<int, 3>{}); entry(std::integral_constant<int, 4>{}); entry(std::integral_constant<int, 5>{});
            entry(std::integral_constant<int, 6>{}); entry(std::integral_constant<int, 7>{}); entry(std::integral_constant<int, 8>{});
            entry(std::integral_constant<int, 9>{}); entry(std::integral_constant<int, 10>{}); entry(std::integral_constant<int, 11>{});
            entry(std::integral_constant<int, 12>{});
            K2B_SDIAG_STAMP(10 + 3 * u);
            // one 12-byte store per (frame, vertex): lane (vertex row, g) holds frames 4 g .. 4 g + 3 of the unit
            const int fbase = (cfg * 8 + u) * 16 + 4 * g;
            if (tile_full) {                           // every (frame, vertex) of the tile exists: lane base + a wave-uniform row offset
#pragma unroll
                for (int i = 0; i < 4; ++i) gstore12(voff, float3r{out[0][i], out[1][i], out[2][i]}, tbase + (size_t)(u * 16 + i) * row_bytes);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = fbase + i;
                    if (okv && f < a.num_frames) gstore12(voff, float3r{out[0][i], out[1][i], out[2][i]}, tbase + (size_t)(u * 16 + i) * row_bytes);
                }
            }
            if (has_joint) {                           // rare (21 of 6890 vertices): the vertex again, into the joints array
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = fbase + i;
                    if (jrow != 0 && okv && f < a.num_frames) {
                        float3v x;
                        x.x = out[0][i]; x.y = out[1][i]; x.z = out[2][i];
                        *reinterpret_cast<float3v*>(a.joints_out + ((size_t)f * a.joints_stride + a.joints_row0 + jrow - 1) * 3) = x;
                    }
                }
            }
        }
        // buffers 1 and 2 hold the next tile's k-steps 0 and 1 (landed long ago: the waits of the units covered them); the empty
        // statement pins the copies behind this point - the compiler takes an asm load's result for ready at once
        asm volatile("" : "+v"(pb1[0][0]), "+v"(pb1[0][1]), "+v"(pb1[1][0]), "+v"(pb1[1][1]), "+v"(pb1[2][0]), "+v"(pb1[2][1]),
                          "+v"(pb2[0][0]), "+v"(pb2[0][1]), "+v"(pb2[1][0]), "+v"(pb2[1][1]), "+v"(pb2[2][0]), "+v"(pb2[2][1]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) { pb0[c][h] = pb1[c][h]; pb1[c][h] = pb2[c][h]; }
        wt = nt; cfg = nxf; cvg = nxv;
        K2B_SDIAG_TILE;
    }
    wait_vmcnt<0>();
    K2B_SDIAG_END;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k2b_lbs_stream_x_kernel: the same design for 49-56 joints and 16 pose k-steps (SMPL-X: 9 x 54 + 20 + 2 = 508 features).
//
// What does not carry over from the kernel above is the resident X: 16 k-steps x 16 KiB do not fit, so X walks through a FOUR-slot
// ring of one k-step each, three k-steps ahead, and the pose phase has one barrier per k-step (the barrier at the top of k-step
// ks publishes X(ks + 1), which every wave has waited for itself, and frees the slot of k-step ks - 1 for X(ks + 3)).  Pd still
// goes global -> registers two k-steps ahead, private to the wave.  The transform contraction over 55 joints is
//   A fragments per entry (1 KiB each):  H0 = hi groups 0-3,  H1 = hi 4-6 | PAD (translation terms),  L0 = lo 0-3,  L1 = lo 4-6 | ZERO
//   W fragments per 16-vertex tile:      Wh0, Wh1 | ONES, Wl0, Wl1 | 0, Wh1 | tag          (resident in registers for the phase)
//   T = H0.Wh0 + H1.[Wh1|ONES] + H0.Wl0 + H1.[Wl1|0] + L0.Wh0 + L1.[Wh1|tag]            four LDS reads for six MFMAs
// LDS: 4 x 16 KiB (X ring) + 2 x 48 KiB (A units) = 160 KiB.
//
// Vector-memory operations of a wave in issue order (loads, LDS-DMA fills and stores retire in order; every wait is counted):
//   pose phase, top of k-step ks:  [wait] [barrier, ks = 1..14]  X(ks + 3) x 2 (ks = 1..12)   W x 5 (ks = 15)   Pd(ks + 2) x 6
//   transform, top of unit u:      [wait] [barrier]  A(u + 1) x 6   X(next tile, u) x 2 (u < 4)   ... stores x 4 (full tile)
// Pd(16), Pd(17) are the next tile's k-steps 0 and 1.  Younger than what a wait needs:
//   k-step 2..13: X(ks + 2), Pd(ks + 1) = 8;  14, 15: one Pd = 6;  0, 1: nothing to wait for (covered by the waits of the units)
//   unit 0: Pd(17) = 6 (needs W);  units 1..4: X x 2 + stores x 4 = 6 (partial tile: 2);  units 5..7: stores = 4 (partial: 0)
constexpr int XKS = kStreamXKSteps;
constexpr int kXRingBytes = 4 * 16 * 1024;
constexpr int kUnitXBytes = 48 * 1024;             // A operand of one 16-frame unit: 12 entries x 4 fragments
#ifndef K2B_STREAMX_CHUNK
#define K2B_STREAMX_CHUNK 8
#endif
#define K2B_LDS_READY4(N, b) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory")

namespace {
struct WalkX {                 // as Walk, with its own chunk length
    int vgroups, fgroups, item_lo, item_hi, nx;
    __device__ void init(int vgroups_, int fgroups_, int block, int nblocks) {
        vgroups = vgroups_; fgroups = fgroups_;
        const int items = ((fgroups + K2B_STREAMX_CHUNK - 1) / K2B_STREAMX_CHUNK) * vgroups, x = block & 7;
        item_lo = (int)((long long)items * x / 8); item_hi = (int)((long long)items * (x + 1) / 8);
        nx = nblocks >> 3;
    }
    __device__ void next(int& t, int& fg, int& vg) const {
        for (;;) {
            t += nx;
            const int item = item_lo + t / K2B_STREAMX_CHUNK;
            if (item >= item_hi) { fg = -1; vg = 0; return; }
            const int c = item / vgroups;
            fg = c * K2B_STREAMX_CHUNK + t % K2B_STREAMX_CHUNK; vg = item - c * vgroups;
            if (fg < fgroups) return;
        }
    }
};
}  // namespace

__global__ __launch_bounds__(512) void k2b_lbs_stream_x_kernel(const StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [4][16 KiB] X ring | [2][48 KiB] A units
    unsigned char* const aslots = lds + kXRingBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = lane & 15, g = lane >> 4;
    const int f32tiles = a.f32_tiles, f16tiles = 2 * f32tiles, nv16 = a.nv16;
    const unsigned lane16 = (unsigned)lane * 16u;

    WalkX walk;
    walk.init(nv16 >> 3, (f32tiles + 3) >> 2, blockIdx.x, a.num_wgs);
    int wt = (blockIdx.x >> 3) - walk.nx, cfg, cvg;
    walk.next(wt, cfg, cvg);
    if (cfg < 0) return;
    K2B_SXDIAG_BEGIN;

    // X of k-step ks into ring slot ks & 3: wave w moves (k-half w >> 2, frame tile w & 3), hi and lo
    auto issue_x = [&](int fg, int ks) {
        const int kh = wave >> 2, ft = wave & 3;
        int ftc = fg * 4 + ft;
        ftc = ftc < f32tiles ? ftc : f32tiles - 1;
        const size_t o = ((size_t)(2 * ks + kh) * f32tiles + ftc) * 512 + lane * 8;
        unsigned char* dst = lds + (ks & 3) * 16384 + (kh * 8 + ft * 2) * 1024;
        __builtin_amdgcn_global_load_lds(a.xh + o, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(a.xl + o, dst + 1024, 16, 0, 0);
    };
    // A of one 16-frame unit: 48 contiguous pieces [entry 12][fragment 4]; wave w moves pieces 6 w .. 6 w + 5
    auto issue_a = [&](int f16, int slot) {
        f16 = f16 < f16tiles ? f16 : f16tiles - 1;
        const k2b_half* src = a.a2 + ((size_t)f16 * 48 + 6 * wave) * 512 + lane * 8;
        unsigned char* dst = aslots + slot * kUnitXBytes + 6 * wave * 1024;
#pragma unroll
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds(src + i * 512, dst + i * 1024, 16, 0, 0);
    };
    auto load_pd = [&](half8 (&buf)[3][2], int vg, int ks) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.pd + ((size_t)ks * nv16 + ((K2B_SX_SKIP & 32) ? 0 : vg) * 8 + wave) * 6 * 512) + 3072;
        gload16<-3072>(buf[0][0], lane16, base); gload16<-2048>(buf[0][1], lane16, base);
        gload16<-1024>(buf[1][0], lane16, base); gload16<0>(buf[1][1], lane16, base);
        gload16<1024>(buf[2][0], lane16, base);  gload16<2048>(buf[2][1], lane16, base);
    };

    const int lx = (g >> 1) * 8192 + (g & 1) * 512 + row * 16;      // lane part of an X fragment address inside a ring slot
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned lxa = lds0 + lx;
    const int la = g * 256 + row * 16;                               // lane part of an A fragment address inside a piece
    const float inv_scale = 1.0f / kPdScale;

    half8 pb0[3][2], pb1[3][2], pb2[3][2];      // Pd buffers: k-step ks lives in buffer ks % 3 (16 % 3 = 1: the next tile's k-steps 0 and 1
                                                // land in buffers 1 and 2 and are renamed at the end of the tile)
    half8 wf[5];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) issue_x(cfg, ks);
    issue_a(cfg * 8, 0);
    load_pd(pb0, cvg, 0);
    load_pd(pb1, cvg, 1);
    K2B_PD_READY(0, pb0);
    K2B_PD_READY(0, pb1);
    wg_barrier();

    while (cfg >= 0) {
        int nt = wt, nxf, nxv;
        walk.next(nt, nxf, nxv);
        const int nfg = nxf >= 0 ? nxf : cfg, nvg = nxf >= 0 ? nxv : cvg;
        floatx4 vp[8][3];

        // ---- pose phase ------------------------------------------------------------------------------------------------------
        half8 xq[2][2];
        auto xread = [&](half8 (&dst)[2], auto ksc, auto fc) {
            constexpr int ks = decltype(ksc)::value, f = decltype(fc)::value;
            constexpr int off = (ks & 3) * 16384 + (f >> 1) * 2048 + (f & 1) * 256;
            lread16<off>(dst[0], lxa); lread16<off + 1024>(dst[1], lxa);
        };
        auto kstep = [&](auto ksc, const half8 (&pd)[3][2]) {
            constexpr int ks = decltype(ksc)::value;
            auto tile = [&](auto fc) {
                constexpr int f = decltype(fc)::value;
                half8 (&cur)[2] = xq[f & 1];
                // (the first fragment of k-step ks + 1 is read at the end of k-step ks: the barrier at the top of ks published it)
                if constexpr (f < 7) xread(xq[(f + 1) & 1], ksc, std::integral_constant<int, (f + 1) & 7>{});
                else if constexpr (ks < XKS - 1) xread(xq[0], std::integral_constant<int, ks + 1>{}, std::integral_constant<int, 0>{});
                if constexpr (f < 7 || ks < XKS - 1) K2B_LDS_READY2(2, cur); else K2B_LDS_READY2(0, cur);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (ks == 0) vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][0], vp[f][c], 0, 0, 0);
                    vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], pd[c][1], vp[f][c], 0, 0, 0);
                    vp[f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[1], pd[c][0], vp[f][c], 0, 0, 0);
                }
            };
            tile(std::integral_constant<int, 0>{}); tile(std::integral_constant<int, 1>{}); tile(std::integral_constant<int, 2>{});
            tile(std::integral_constant<int, 3>{}); tile(std::integral_constant<int, 4>{}); tile(std::integral_constant<int, 5>{});
            tile(std::integral_constant<int, 6>{}); tile(std::integral_constant<int, 7>{});
        };
        // top of k-step ks: wait, barrier, fills and loads as in the table above, then the 72 MFMAs; cur = buffer ks % 3, nxt = (ks + 2) % 3
        auto step = [&](auto ksc, half8 (&cur)[3][2], half8 (&nxt)[3][2]) {
            constexpr int ks = decltype(ksc)::value;
            K2B_SXDIAG_STAMP(ks);
            if constexpr (ks >= 2 && ks <= 13) K2B_PD_READY(8, cur);
            else if constexpr (ks >= 14) K2B_PD_READY(6, cur);
            else K2B_PD_READY(63, cur);                 // (the counter's ceiling: ties the registers to this point, waits for nothing)
            if constexpr (ks >= 1 && ks <= 14) if (!(K2B_SX_SKIP & 8)) wg_barrier();
            if constexpr (ks >= 1 && ks <= 12 && !(K2B_SX_SKIP & 2)) issue_x(cfg, ks + 3);
            if constexpr (ks == XKS - 1) {             // W fragments of this tile's vertices, needed behind the pose phase
                const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w + ((size_t)cvg * 8 + wave) * 5 * 512) + 2048;   // (13-bit signed offsets)
                gload16<-2048>(wf[0], lane16, wbase); gload16<-1024>(wf[1], lane16, wbase); gload16<0>(wf[2], lane16, wbase);
                gload16<1024>(wf[3], lane16, wbase); gload16<2048>(wf[4], lane16, wbase);
            }
            if constexpr (!(K2B_SX_SKIP & 4)) { if constexpr (ks + 2 < XKS) load_pd(nxt, cvg, ks + 2); else load_pd(nxt, nvg, ks + 2 - XKS); }
            kstep(ksc, cur);
        };
        xread(xq[0], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 0>{}, pb0, pb2);   step(std::integral_constant<int, 1>{}, pb1, pb0);
        step(std::integral_constant<int, 2>{}, pb2, pb1);   step(std::integral_constant<int, 3>{}, pb0, pb2);
        step(std::integral_constant<int, 4>{}, pb1, pb0);   step(std::integral_constant<int, 5>{}, pb2, pb1);
        step(std::integral_constant<int, 6>{}, pb0, pb2);   step(std::integral_constant<int, 7>{}, pb1, pb0);
        step(std::integral_constant<int, 8>{}, pb2, pb1);   step(std::integral_constant<int, 9>{}, pb0, pb2);
        step(std::integral_constant<int, 10>{}, pb1, pb0);  step(std::integral_constant<int, 11>{}, pb2, pb1);
        step(std::integral_constant<int, 12>{}, pb0, pb2);  step(std::integral_constant<int, 13>{}, pb1, pb0);
        step(std::integral_constant<int, 14>{}, pb2, pb1);  step(std::integral_constant<int, 15>{}, pb0, pb2);
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { vp[f][c][i] *= inv_scale; asm volatile("" : "+v"(vp[f][c][i])); }

        // ---- transform phase ---------------------------------------------------------------------------------------------------
        const int v = (cvg * 8 + wave) * 16 + row;
        const bool okv = v < a.num_out;
        int jrow = 0;
        bool has_joint = false;
        const size_t row_bytes = (size_t)a.out_stride * 12;
        unsigned char* const tbase = reinterpret_cast<unsigned char*>(a.out) + (size_t)(cfg * 128) * row_bytes;
        const unsigned voff = (unsigned)(((size_t)a.out_row0 + v) * 12 + (size_t)(4 * g) * row_bytes);
        const bool tile_full = cfg * 128 + 127 < a.num_frames && (cvg * 8 + wave) * 16 + 15 < a.num_out;   // wave-uniform
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            K2B_SXDIAG_STAMP(16 + 3 * u);
            if (u == 0) asm volatile("s_waitcnt vmcnt(6)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4])::"memory");
            else if (u <= 4) { if (tile_full) wait_vmcnt<6>(); else wait_vmcnt<2>(); }
            else { if (tile_full) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
            if (!(K2B_SX_SKIP & 8)) wg_barrier();
            K2B_SXDIAG_STAMP(17 + 3 * u);
            if (u == 0 && a.joints_out) {
                const float tg = (float)wf[4][0];      // lanes g == 3 hold the tag group of their row
                jrow = (int)__shfl(tg, 48 + row, 64);
                has_joint = __builtin_amdgcn_ballot_w64(jrow != 0) != 0;
            }
            // behind the barrier the other A slot is free, and (from unit 0 on) the whole X ring: next unit's A, next tile's X k-step u
            if (!(K2B_SX_SKIP & 2)) {
                if (u < 7) issue_a(cfg * 8 + u + 1, (u + 1) & 1); else issue_a(nfg * 8, 0);
                if (u < 4) issue_x(nfg, u);
            }
            floatx4 out[3] = {floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}, floatx4{0.f, 0.f, 0.f, 0.f}};
            half8 af[2][4];
            floatx4 t[2];
            const unsigned sa = lds0 + kXRingBytes + (u & 1) * kUnitXBytes + la;
            lread16<0>(af[0][0], sa); lread16<1024>(af[0][1], sa); lread16<2048>(af[0][2], sa); lread16<3072>(af[0][3], sa);
            auto entry = [&](auto nc) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n + 1 < 12) {
                    constexpr int e1 = 4 * ((n + 1) % 3) + (n + 1) / 3;
                    lread16<e1 * 4096>(af[(n + 1) & 1][0], sa); lread16<e1 * 4096 + 1024>(af[(n + 1) & 1][1], sa);
                    lread16<e1 * 4096 + 2048>(af[(n + 1) & 1][2], sa); lread16<e1 * 4096 + 3072>(af[(n + 1) & 1][3], sa);
                }
                if constexpr (n + 1 < 12) K2B_LDS_READY4(4, af[n & 1]); else if constexpr (n < 12) K2B_LDS_READY4(0, af[n & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (n < 12) {
                    const half8 (&f)[4] = af[n & 1];
                    floatx4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[0], wf[0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[1], wf[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[0], wf[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[1], wf[3], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[2], wf[0], acc, 0, 0, 0);
                    t[n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[3], wf[4], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (n > 0) {
                    constexpr int d = (n - 1) / 3, r = (n - 1) % 3;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (d < 3) out[r][i] = __builtin_fmaf(t[(n - 1) & 1][i], vp[u][d][i], out[r][i]);
                        else out[r][i] += t[(n - 1) & 1][i];
                        asm volatile("" : "+v"(out[r][i]));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            entry(std::integral_constant<int, 0>{}); entry(std::integral_constant<int, 1>{}); entry(std::integral_constant<int, 2>{});
            entry(std::integral_constant<int, 3>{}); entry(std::integral_constant<int, 4>{}); entry(std::integral_constant<int, 5>{});
            entry(std::integral_constant<int, 6>{}); entry(std::integral_constant<int, 7>{}); entry(std::integral_constant<int, 8>{});
            entry(std::integral_constant<int, 9>{}); entry(std::integral_constant<int, 10>{}); entry(std::integral_constant<int, 11>{});
            entry(std::integral_constant<int, 12>{});
            K2B_SXDIAG_STAMP(18 + 3 * u);
            const int fbase = (cfg * 8 + u) * 16 + 4 * g;
            if (K2B_SX_SKIP & 1) {
                asm volatile("" ::"v"(out[0]), "v"(out[1]), "v"(out[2]));
            } else if (tile_full) {
#pragma unroll
                for (int i = 0; i < 4; ++i) gstore12(voff, float3r{out[0][i], out[1][i], out[2][i]}, tbase + (size_t)(K2B_SXDIAG_STORES ? u * 16 + i : i) * row_bytes);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = fbase + i;
                    if (okv && f < a.num_frames) gstore12(voff, float3r{out[0][i], out[1][i], out[2][i]}, tbase + (size_t)(u * 16 + i) * row_bytes);
                }
            }
            if (has_joint) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = fbase + i;
                    if (jrow != 0 && okv && f < a.num_frames) {
                        float3v x;
                        x.x = out[0][i]; x.y = out[1][i]; x.z = out[2][i];
                        *reinterpret_cast<float3v*>(a.joints_out + ((size_t)f * a.joints_stride + a.joints_row0 + jrow - 1) * 3) = x;
                    }
                }
            }
        }
        asm volatile("" : "+v"(pb1[0][0]), "+v"(pb1[0][1]), "+v"(pb1[1][0]), "+v"(pb1[1][1]), "+v"(pb1[2][0]), "+v"(pb1[2][1]),
                          "+v"(pb2[0][0]), "+v"(pb2[0][1]), "+v"(pb2[1][0]), "+v"(pb2[1][1]), "+v"(pb2[2][0]), "+v"(pb2[2][1]));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) { pb0[c][h] = pb1[c][h]; pb1[c][h] = pb2[c][h]; }
        K2B_SXDIAG_STAMP(40);
        wt = nt; cfg = nxf; cvg = nxv;
        K2B_SXDIAG_TILE;
    }
    wait_vmcnt<0>();
}

hipError_t launch_skin_stream(const StreamArgs& a_in, int num_cus, hipStream_t stream) {
    if (a_in.num_frames <= 0 || a_in.num_out <= 0) return hipSuccess;
    StreamArgs a = a_in;
    if ((a.nv16 & 7) || a.f32_tiles <= 0) return hipErrorInvalidValue;
    const long long tiles = (long long)(a.nv16 >> 3) * ((a.f32_tiles + 3) >> 2);
    int wgs = num_cus < 8 ? 8 : num_cus / 8 * 8;           // one persistent workgroup per CU, a multiple of the 8 XCD labels
    if (tiles < wgs) wgs = (int)((tiles + 7) / 8 * 8);
    a.num_wgs = wgs;
    const size_t lds = (size_t)kXBytes + 2 * kUnitBytes;
    static std::atomic<unsigned long long> lds_set{0};
    const hipError_t e = ensure_dynamic_lds(k2b_lbs_stream_kernel, lds_set, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k2b_lbs_stream_kernel, dim3(wgs), dim3(512), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_skin_stream_x(const StreamArgs& a_in, int num_cus, hipStream_t stream) {
    if (a_in.num_frames <= 0 || a_in.num_out <= 0) return hipSuccess;
    StreamArgs a = a_in;
    if ((a.nv16 & 7) || a.f32_tiles <= 0) return hipErrorInvalidValue;
    const long long tiles = (long long)(a.nv16 >> 3) * ((a.f32_tiles + 3) >> 2);
    int wgs = num_cus < 8 ? 8 : num_cus / 8 * 8;
    if (tiles < wgs) wgs = (int)((tiles + 7) / 8 * 8);
    a.num_wgs = wgs;
    const size_t lds = (size_t)kXRingBytes + 2 * kUnitXBytes;
    static std::atomic<unsigned long long> lds_set{0};
    const hipError_t e = ensure_dynamic_lds(k2b_lbs_stream_x_kernel, lds_set, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k2b_lbs_stream_x_kernel, dim3(wgs), dim3(512), lds, stream, a);
    return hipGetLastError();
}

}  // namespace k2b
