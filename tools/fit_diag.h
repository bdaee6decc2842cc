// Diagnostics of the fused fit kernel (k2b_fit.hip), kept out of the shipped translation unit: tools/build_fit_stamps.sh compiles
// k2b_fit.hip with -DK2B_FIT_DIAG_HEADER=<this file>.  s_memtime stamps of iteration 50 in both role loops of the split shapes, printed
// with device printf by one workgroup behind the loop (tools/dev_fit_once.py FRAMES - 1 fstamp shows them):
//   tree wave: 0 before barrier 1 | 1 behind it | 2 behind the tree pass | 3 behind barrier 2
//   row wave:  0 | 1 published | 2 behind barrier 1 | 3 both components' MFMAs issued | 4 consumed | 5 met the other row waves |
//              6 arg-min + priors | 7 behind barrier 2 | 8 Adam
#pragma once
#define K2B_FSTAMP_DECL unsigned fst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define K2B_FSTAMP(i) do { if (it == 50) fst[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#define K2B_FSTAMP_TREE_PRINT                                                                                                    \
    do { if (blockIdx.x == 3 && lane == 0) printf("tree wave %d: wait at barrier 1 %u | tree pass %u | wait at barrier 2 %u\n", wave, \
                                                   fst[1] - fst[0], fst[2] - fst[1], fst[3] - fst[2]); } while (0)
#define K2B_FSTAMP_ROW_PRINT                                                                                                     \
    do { if (blockIdx.x == 3 && lane == 0)                                                                                       \
        printf("row wave %d: publish %u | wait at barrier 1 %u | MFMA issue %u | consume %u | meet %u | arg-min + priors %u | wait at barrier 2 %u | Adam %u\n", \
               wave, fst[1] - fst[0], fst[2] - fst[1], fst[3] - fst[2], fst[4] - fst[3], fst[5] - fst[4], fst[6] - fst[5], fst[7] - fst[6], fst[8] - fst[7]); } while (0)
