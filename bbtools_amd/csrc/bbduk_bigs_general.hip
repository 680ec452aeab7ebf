// bbduk_bigs_general.hip -- bbduk_bigs_every_kernel for the GENERAL family (restrictleft / restrictright, skipr1 / skipr2, qskip, speed, rcomp=f) over a
// big-layout map, round 5: the flags only decide WHICH positions of a read are looked up (the plane of valid positions, bigs_valid_plane), cut the windows
// at a span start inside the read (looked up exactly by worker lanes, bigs_value_at's spanLo), gate a key (speed) or pick its strand (rcomp=f) -- the
// stream scan over the minimizer lines and the exact hit plane behind it are unchanged.  First-hit scans (ktrim=r, plain kfilter) take this every-hit form
// too: one kernel per mode.  kfilter / ktrim=r / ktrim=l here, the other modes in bbduk_bigs_general_b.hip (parallel build).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_general_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                               const int64_t n, const int64_t totalBases, const int paired,
                               int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                               int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, MODE != BBDUK_MODE_KFILTER, true, true, 2, true, 6>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs_general_b(int mode);
batch_kernel_t bbduk_pick_bigs_general(int mode) {
    if (mode == BBDUK_MODE_KFILTER) return bbduk_bigs_general_kernel<BBDUK_MODE_KFILTER>;
    if (mode == BBDUK_MODE_KTRIM_R) return bbduk_bigs_general_kernel<BBDUK_MODE_KTRIM_R>;
    if (mode == BBDUK_MODE_KTRIM_L) return bbduk_bigs_general_kernel<BBDUK_MODE_KTRIM_L>;
    return bbduk_pick_bigs_general_b(mode);
}
