// bbduk_bigs.hip -- bbduk_bigs_kernel: the first-hit kfilter scan over a big-layout map with the 32-bit line function, i.e. wave_body with the
// stream scan of bbduk_bigs.inc (SHAPE = 5).  DESIGN 4.10 "The stream scan over minimizer lines".  One instantiation per input format; what
// forbidNs changes (the windows that see an undefined base) is decided at run time in the rare exact paths.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE, bool SHORT, int FMT>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                       const int64_t n, const int64_t totalBases, const int paired,
                       int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                       int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, SHORT, true, false, FMT, true, 5>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
// kfilter (maxbadkmers = 0): one kernel per input format; ktrim=r (the large-map path of the trimming mode, round 4): with and without the short
// k-mers of mink, input format decided per launch
batch_kernel_t bbduk_pick_bigs(int mode, bool useShort, bool packed) {
    if (mode == BBDUK_MODE_KTRIM_R) return useShort ? bbduk_bigs_kernel<BBDUK_MODE_KTRIM_R, true, 2> : bbduk_bigs_kernel<BBDUK_MODE_KTRIM_R, false, 2>;
    return packed ? bbduk_bigs_kernel<BBDUK_MODE_KFILTER, false, 1> : bbduk_bigs_kernel<BBDUK_MODE_KFILTER, false, 0>;
}
