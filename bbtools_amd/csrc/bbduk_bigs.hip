// bbduk_bigs.hip -- bbduk_bigs_kernel: the first-hit kfilter scan over a big-layout map with the 32-bit line function, i.e. wave_body with the
// stream scan of bbduk_bigs.inc (SHAPE = 5).  DESIGN 4.10 "The stream scan over minimizer lines".  One instantiation per input format; what
// forbidNs changes (the windows that see an undefined base) is decided at run time in the rare exact paths.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int FMT>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                       const int64_t n, const int64_t totalBases, const int paired,
                       int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                       int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<BBDUK_MODE_KFILTER, false, true, false, FMT, true, 5>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs(bool packed) { return packed ? bbduk_bigs_kernel<1> : bbduk_bigs_kernel<0>; }
