// bbduk_stream_every.hip -- bbduk_stream_every_kernel for ktrim=l and for kfilter with a threshold (maxbadkmers > 0, mkf, mcf): wave_body
// with the stream scan and the exact hit plane behind it (SHAPE = 4; bbduk_stream_scan.inc: stream_every_verify).  DESIGN 4.0.
// Input format decided per launch (FMT = 2); ksplit and ktrim=n: bbduk_stream_every_b.hip.
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE, bool SHORT, bool FORBIDN, bool GENERAL>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_stream_every_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                               const int64_t n, const int64_t totalBases, const int paired,
                               int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                               int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, SHORT, FORBIDN, GENERAL, 2, false, 4>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

batch_kernel_t bbduk_pick_stream_every_b(int mode, bool forbidN, bool general);
batch_kernel_t bbduk_pick_stream_every(int mode, bool useShort, bool forbidN, bool general) {
    if (mode == BBDUK_MODE_KSPLIT || mode == BBDUK_MODE_KMASK || mode == BBDUK_MODE_FBM) return bbduk_pick_stream_every_b(mode, forbidN, general);
    if (mode == BBDUK_MODE_KFILTER) {
        if (general) return bbduk_stream_every_kernel<BBDUK_MODE_KFILTER, true, true, true>;
        return forbidN ? bbduk_stream_every_kernel<BBDUK_MODE_KFILTER, false, true, false> : bbduk_stream_every_kernel<BBDUK_MODE_KFILTER, false, false, false>;
    }
    // (ktrim=r as an every-hit scan, round 6: handles whose query-side expansion is tabulated -- launch_batch)
    if (general) return mode == BBDUK_MODE_KTRIM_R ? bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_R, true, true, true> : bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_L, true, true, true>;
    if (useShort) return forbidN ? bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_L, true, true, false> : bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_L, true, false, false>;
    return forbidN ? bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_L, false, true, false> : bbduk_stream_every_kernel<BBDUK_MODE_KTRIM_L, false, false, false>;
}
