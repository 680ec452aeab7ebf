// bbduk_bigs_every_b.hip -- bbduk_bigs_every_kernel for ksplit, ktrim=n, ktrim=rl and findbestmatch (see bbduk_bigs_every.hip): the operators whose
// large maps had no big-layout form until round 5 (cache-resident at 60-230 Gbases/s from 10^6 keys, refused beyond 2^29 buckets).  The scan and
// the verification are bbduk_bigs.inc's; what each mode reads out of the exact hit plane is wave_body's (BIGS && EVERY).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_every_b_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                               const int64_t n, const int64_t totalBases, const int paired,
                               int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                               int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    // (SHORT: these modes' kernels are instantiated with it whether or not mink is set -- P.useShort decides at run time; findbestmatch has no short k-mers)
    wave_body<MODE, MODE != BBDUK_MODE_FBM, true, false, 2, true, 6>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs_every_b(int mode) {
    if (mode == BBDUK_MODE_KSPLIT) return bbduk_bigs_every_b_kernel<BBDUK_MODE_KSPLIT>;
    if (mode == BBDUK_MODE_KMASK) return bbduk_bigs_every_b_kernel<BBDUK_MODE_KMASK>;
    if (mode == BBDUK_MODE_KTRIM_TIPS) return bbduk_bigs_every_b_kernel<BBDUK_MODE_KTRIM_TIPS>;
    return bbduk_bigs_every_b_kernel<BBDUK_MODE_FBM>;
}
