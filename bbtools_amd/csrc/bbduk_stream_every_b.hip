// bbduk_stream_every_b.hip -- bbduk_stream_every_kernel for ksplit, ktrim=n and findbestmatch (see bbduk_stream_every.hip).
#include "bbduk_device.inc"
#include "bbduk_kernels.h"

template <int MODE, bool FORBIDN, bool GENERAL>
__global__ __launch_bounds__(BLOCK_THREADS)
void bbduk_stream_every_b_kernel(const KParams P, const uint8_t* __restrict__ bases, const int64_t* __restrict__ offsets,
                                 const int64_t n, const int64_t totalBases, const int paired,
                                 int32_t* __restrict__ outA, int32_t* __restrict__ outId, uint8_t* __restrict__ outFlags,
                                 int64_t* __restrict__ counters, const int* __restrict__ slowFlag) {
    wave_body<MODE, MODE != BBDUK_MODE_FBM, FORBIDN, GENERAL, 2, false, 4>(P, bases, offsets, n, totalBases, paired, outA, outId, outFlags, counters, slowFlag);
}

batch_kernel_t bbduk_pick_stream_every_b(int mode, bool forbidN, bool general) {
    if (mode == BBDUK_MODE_FBM) {
        if (general) return bbduk_stream_every_b_kernel<BBDUK_MODE_FBM, true, true>;
        return forbidN ? bbduk_stream_every_b_kernel<BBDUK_MODE_FBM, true, false> : bbduk_stream_every_b_kernel<BBDUK_MODE_FBM, false, false>;
    }
    if (mode == BBDUK_MODE_KSPLIT) {
        if (general) return bbduk_stream_every_b_kernel<BBDUK_MODE_KSPLIT, true, true>;
        return forbidN ? bbduk_stream_every_b_kernel<BBDUK_MODE_KSPLIT, true, false> : bbduk_stream_every_b_kernel<BBDUK_MODE_KSPLIT, false, false>;
    }
    if (general) return bbduk_stream_every_b_kernel<BBDUK_MODE_KMASK, true, true>;
    return forbidN ? bbduk_stream_every_b_kernel<BBDUK_MODE_KMASK, true, false> : bbduk_stream_every_b_kernel<BBDUK_MODE_KMASK, false, false>;
}
