// bbduk_bigs_kbig.hip -- bbduk_bigs_every_kernel for k > 31 (countSetKmersBig: runs of matching 31-mers, BBDukProcessorS.java:1727-1800) over a
// big-layout map, plain and with the GENERAL flags (round 5: until then k > 31 kept the cache-resident layout at any map size and was refused beyond
// 2^29 buckets).  The scan and the verification are bbduk_bigs.inc's; the run state machine over the exact hit plane is wave_body's (KBIG && BIGS).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <bool GENERAL>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_bigs_kbig_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                            const int64_t n, const int64_t totalBases, const int paired,
                            int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                            int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<BBDUK_MODE_KBIG, false, true, GENERAL, 2, true, 6>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}
batch_kernel_t bbduk_pick_bigs_kbig(bool general) { return general ? bbduk_bigs_kbig_kernel<true> : bbduk_bigs_kbig_kernel<false>; }
