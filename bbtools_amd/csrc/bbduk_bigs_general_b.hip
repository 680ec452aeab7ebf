// bbduk_bigs_general_b.hip -- bbduk_bigs_every_kernel for the GENERAL family over a big-layout map: ksplit, ktrim=n, ktrim=rl, findbestmatch
// (see bbduk_bigs_general.hip).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_general_b_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                                 const int64_t n, const int64_t totalBases, const int paired,
                                 int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                                 int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, MODE != BBDUK_MODE_FBM, true, true, 2, true, 6>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs_general_b(int mode) {
    if (mode == BBDUK_MODE_KSPLIT) return bbduk_bigs_general_b_kernel<BBDUK_MODE_KSPLIT>;
    if (mode == BBDUK_MODE_KMASK) return bbduk_bigs_general_b_kernel<BBDUK_MODE_KMASK>;
    if (mode == BBDUK_MODE_KTRIM_TIPS) return bbduk_bigs_general_b_kernel<BBDUK_MODE_KTRIM_TIPS>;
    return bbduk_bigs_general_b_kernel<BBDUK_MODE_FBM>;
}
