// Timing-only diagnostics of the LBS tile kernel (k2b_lbs.hip), kept OUT of the shipped translation unit: the kernel calls the
// K2B_DIAG_* hooks, which are no-ops in the product build; tools/build_lbs_variants.sh compiles k2b_lbs.hip with
//   -DK2B_LBS_DIAG_HEADER='"<repo>/tools/lbs_diag.h"' -DK2B_TILE_DIAG=<n>
// into tools/libk2b_<name>.so.  Results of these builds are WRONG on purpose; they answer "what does this part cost".
//   K2B_TILE_DIAG 1: every store goes to the dump row     2: only the first slice is ever filled (stale LDS afterwards)
//   3: no MFMAs (fills, LDS reads, barriers, stores only)  5: stores land in the rows of the first 32 frames only (an
//   L2-resident footprint: no HBM write stream)            6: s_memtime stamps of every slice of one tile (tools/dev_lbs_stamps.py)
#pragma once
#ifndef K2B_TILE_DIAG
#define K2B_TILE_DIAG 0
#endif

#if K2B_TILE_DIAG == 2
#define K2B_DIAG_SKIP_FILL(lq) ((lq) > 0)
#else
#define K2B_DIAG_SKIP_FILL(lq) false
#endif

#if K2B_TILE_DIAG == 5
#define K2B_DIAG_STORE(f, ok) do { (f) &= 31; (ok) = true; } while (0)
#elif K2B_TILE_DIAG == 1
#define K2B_DIAG_STORE(f, ok) do { (f) = 0; (ok) = false; } while (0)
#else
#define K2B_DIAG_STORE(f, ok) ((void)0)
#endif

#if K2B_TILE_DIAG == 3
// operands stay live (their LDS reads are kept), no matrix instruction
#define K2B_DIAG_MFMA(x, y, c) ([&] { asm volatile("" ::"v"(x), "v"(y)); return (c); }())
#else
#define K2B_DIAG_MFMA(x, y, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0)
#endif

#if K2B_TILE_DIAG == 6
// stamps of the workgroup's THIRD tile, all 8 waves, 8 per slice, kept in the unused 16 KiB of LDS behind the W image (no
// vector-memory traffic, so the counted waits are undisturbed) and copied out at the end by the blocks 0 and 77
#define K2B_DIAG_STAMP_DECL                                                                     \
    unsigned* const diag_stamps = reinterpret_cast<unsigned*>(wimg + 8 * NGP * 256);             \
    int diag_tile_no = 0
#define K2B_DIAG_STAMP(slice, k)                                                                \
    do { if (diag_tile_no == 2 && lane == 0) diag_stamps[(wave * 32 + (slice)) * 8 + (k)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define K2B_DIAG_TILE_DONE ++diag_tile_no
#define K2B_DIAG_KERNEL_END                                                                     \
    do {                                                                                        \
        if (blockIdx.x == 0 || blockIdx.x == 77) {                                              \
            wg_barrier();                                                                       \
            unsigned* dst = reinterpret_cast<unsigned*>(a.dump) + 1024 + (blockIdx.x ? 2048 : 0); \
            for (int i = threadIdx.x; i < 8 * 32 * 8; i += 512) dst[i] = diag_stamps[i];        \
        }                                                                                       \
    } while (0)
#else
#define K2B_DIAG_STAMP_DECL ((void)0)
#define K2B_DIAG_STAMP(slice, k) ((void)0)
#define K2B_DIAG_TILE_DONE ((void)0)
#define K2B_DIAG_KERNEL_END ((void)0)
#endif

// ---- stream kernel (k2b_lbs_stream.hip) -------------------------------------------------------------------------------------
//   K2B_STREAM_DIAG 1: in-kernel clock.  Lane 0 of wave 0 of every workgroup stamps s_memtime (shader cycles) and s_memrealtime
//   (100 MHz) at the start and the end of its tile loop into the model's scratch row: [1024 + 4 block .. + 3] dwords = d cycles,
//   d realtime ticks, tiles, 0 (tools/dev_lbs_clock.py: clock = d cycles / d ticks x 100 MHz, median over workgroups).
//   K2B_STREAM_DIAG 2: s_memtime stamps of the workgroup's THIRD tile, every wave, 32 points (0 tile start, 1-7 after each pose
//   k-step, then per unit u: 8 + 3u before the counted wait, 9 + 3u behind the barrier, 10 + 3u MFMAs issued, before the
//   stores): blocks 0 and 77 write [1024 + 2048 b' + (wave 32 + i)] dwords (tools/dev_lbs_sstamps.py).
#if K2B_STREAM_DIAG == 2
#define K2B_SDIAG_BEGIN int sd_tile = 0
#define K2B_SDIAG_STAMP(i)                                                                     \
    do { if (sd_tile == 2 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 77))                \
             reinterpret_cast<unsigned*>(a.dump)[1024 + (blockIdx.x ? 2048 : 0) + wave * 32 + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define K2B_SDIAG_TILE ++sd_tile
#define K2B_SDIAG_END ((void)0)
#endif
#if K2B_STREAM_DIAG == 1
#define K2B_SDIAG_BEGIN                                                                        \
    const unsigned long long sd_t0 = __builtin_amdgcn_s_memtime(), sd_r0 = __builtin_amdgcn_s_memrealtime()
#define K2B_SDIAG_END                                                                          \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 256) {                                            \
            unsigned* d = reinterpret_cast<unsigned*>(a.dump) + 1024 + 4 * blockIdx.x;         \
            d[0] = (unsigned)(__builtin_amdgcn_s_memtime() - sd_t0);                           \
            d[1] = (unsigned)(__builtin_amdgcn_s_memrealtime() - sd_r0);                       \
        }                                                                                      \
    } while (0)
#endif

// ---- SMPL-X stream kernel (k2b_lbs_stream_x_kernel) -----------------------------------------------------------------------------
//   K2B_STREAMX_DIAG 2: s_memtime stamps of the workgroup's SECOND tile, every wave, 41 points (ks = 0..15: top of pose k-step ks,
//   then per unit u: 16 + 3u before the counted wait, 17 + 3u behind the barrier, 18 + 3u MFMAs issued, before the stores; 40 tile
//   end): blocks 0 and 77 write [1024 + 2048 b' + wave 64 + i] dwords (tools/dev_lbs_xstamps.py)
//   K2B_STREAMX_DIAG 5: the stores of a full tile land in the rows of the tile's first 4 frames only (an L2-resident footprint)
#if K2B_STREAMX_DIAG == 2
#define K2B_SXDIAG_BEGIN int sd_tile = 0
#define K2B_SXDIAG_STAMP(i)                                                                    \
    do { if (sd_tile == 1 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 77))                \
             reinterpret_cast<unsigned*>(a.dump)[1024 + (blockIdx.x ? 2048 : 0) + wave * 64 + (i)] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define K2B_SXDIAG_TILE ++sd_tile
#define K2B_SXDIAG_STORES 1
#endif
#if K2B_STREAMX_DIAG == 5
#define K2B_SXDIAG_BEGIN ((void)0)
#define K2B_SXDIAG_STAMP(i) ((void)0)
#define K2B_SXDIAG_TILE ((void)0)
#define K2B_SXDIAG_STORES 0
#endif

// ---- pose set-up kernel (k2b_pose_setup_kernel) ---------------------------------------------------------------------------------
//   K2B_POSE_STAMPS 1: s_memtime at nine points of every frame's workgroup (0 start, 1 parameters loaded + J(beta), 2 Rodrigues +
//   barrier, 3 offsets + barrier + X staged, 4 chain composed, 5 A staged + barrier, 6 X stored, 7 A stored, 8 all stores
//   acknowledged); lane 0 writes the differences to the start as floats over the frame's first three output joints
//   (tools/dev_pose_stamps.py reads them; the joints of such a build are wrong on purpose)
#if K2B_POSE_STAMPS
#define K2B_PSTAMP_DECL unsigned long long pst[10]
#define K2B_PSTAMP(i) pst[i] = __builtin_amdgcn_s_memtime()
#define K2B_PSTAMP_END                                                                         \
    do {                                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        K2B_PSTAMP(8);                                                                         \
        if (j == 0 && a.joints_out)                                                            \
            for (int i = 0; i < 9; ++i) a.joints_out[(size_t)f * a.num_out_joints * 3 + i] = (float)(pst[i] - pst[0]); \
    } while (0)
#endif
