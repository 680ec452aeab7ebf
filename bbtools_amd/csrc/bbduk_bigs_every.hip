// bbduk_bigs_every.hip -- bbduk_bigs_every_kernel: the every-hit scans over a big-layout map (ktrim=l; kfilter with maxbadkmers > 0, mkf, mcf), i.e.
// wave_body with the stream scan of bbduk_bigs.inc and the exact hit plane behind it (SHAPE = 6: every candidate of the scan is verified with
// lanes = candidates, the mode's facts are read out of the plane one lane per read).  DESIGN 4.13 "Every-hit scans".  A translation unit of its
// own for the parallel build; ksplit, ktrim=n, ktrim=rl and findbestmatch: bbduk_bigs_every_b.hip.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE, bool SHORT>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_every_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                             const int64_t n, const int64_t totalBases, const int paired,
                             int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                             int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, SHORT, true, false, 2, true, 6>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs_every_b(int mode);
batch_kernel_t bbduk_pick_bigs_every(int mode, bool useShort) {
    if (mode == BBDUK_MODE_KTRIM_L) return useShort ? bbduk_bigs_every_kernel<BBDUK_MODE_KTRIM_L, true> : bbduk_bigs_every_kernel<BBDUK_MODE_KTRIM_L, false>;
    if (mode == BBDUK_MODE_KFILTER) return bbduk_bigs_every_kernel<BBDUK_MODE_KFILTER, false>;
    return bbduk_pick_bigs_every_b(mode);
}
